"""xhistogram_amd — the xhistogram binning-reduction hot path, native to AMD MI355X (gfx950).

``xhistogram_amd.core.histogram`` is a drop-in for ``xhistogram.core.histogram``; the fused
digitize -> joint index -> scatter-add kernel lives in ``libxhist_amd.so`` (csrc/, C ABI in
include/xhist_amd.h).  There is no CPU implementation in this package.
"""

from .core import histogram  # noqa: F401

__version__ = "0.1.0"
__all__ = ["histogram"]
