"""Run under an interpreter that HAS dask (the image's /opt/conda/bin/python3.9; the default
python lacks it).  `lazy`: graph construction only (no GPU needed) — restates the reference's
test_histogram_shape / test_histogram_dask laziness and TypeError rules (test_core.py:231-313,
fixtures.py:8-17).  `compute`: chunked results on the GPU vs numpy (test_chunking.py:8-146)."""
import os
import sys
from itertools import combinations

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dask  # noqa: E402
import dask.array as dsa  # noqa: E402

from xhistogram_amd.core import histogram  # noqa: E402


def forbidden(shape, chunks=None, dtype=float):
    def boom():
        raise ValueError("Triggered forbidden computation")

    a = dsa.from_delayed(dask.delayed(boom)(), shape, dtype)
    return a.rechunk(chunks) if chunks is not None else a


def lazy():
    shape = (10, 15, 12, 20)
    b = forbidden(shape, chunks=(1,) + shape[1:])
    bins = np.linspace(-4, 4, 27)
    for bs in (None, 5, "auto"):
        c, _ = histogram(b, bins=bins, block_size=bs)
        assert c.shape == (26,) and isinstance(c, dsa.Array)
        for axis in list(range(4)) + list(range(-1, -5, -1)):
            c, _ = histogram(b, bins=bins, axis=axis, block_size=bs)
            s = list(shape)
            del s[axis]
            assert c.shape == tuple(s) + (26,) and isinstance(c, dsa.Array)
        for i, j in combinations(range(4), 2):
            c, _ = histogram(b, bins=bins, axis=(i, j), block_size=bs)
            assert c.shape == tuple(shape[k] for k in range(4) if k not in (i, j)) + (26,)
    for axis in [(0, 1, 2, 3), (3, 2, 0, 1)]:
        assert histogram(b, bins=bins, axis=axis)[0].shape == (26,)
    c, _ = histogram(b, b, bins=[bins, bins], weights=forbidden(shape), density=True)
    assert c.shape == (26, 26) and isinstance(c, dsa.Array)
    for bad in (10, "auto"):
        for args, kw in (((b,), {}), ((np.zeros(shape),), {"weights": forbidden(shape)})):
            try:
                histogram(*args, bins=bad, **kw)
            except TypeError:
                pass
            else:
                raise AssertionError("dask + non-array bins must raise TypeError")
    layers = list(histogram(b, bins=bins, axis=(1, 2))[0].dask.layers)
    assert any(k.startswith("bincount") for k in layers) and any(k.startswith("sum") for k in layers), layers
    print("LAZY-OK")


def compute():
    rng = np.random.default_rng(0)
    bins_a, bins_b = np.linspace(-4, 4, 9), np.linspace(-4, 4, 10)
    a, b, w = rng.standard_normal((10, 12)), rng.standard_normal((10, 12)), rng.uniform(0, 1, (10, 12))
    for xc in (1, 3, 10):
        for yc in (2, 12):
            da = dsa.from_array(a, chunks=(xc, yc))
            h, _ = histogram(da, bins=bins_a)
            np.testing.assert_array_equal(h.compute(), np.histogram(a, bins=bins_a)[0])
            h, _ = histogram(da, bins=bins_a, axis=0)
            np.testing.assert_array_equal(h.compute(), np.stack([np.histogram(a[:, j], bins=bins_a)[0] for j in range(12)]))
            db = dsa.from_array(b, chunks=(xc + 1, yc + 1))  # unaligned with da
            h, _ = histogram(da, db, bins=[bins_a, bins_b])
            np.testing.assert_array_equal(h.compute(), np.histogram2d(a.ravel(), b.ravel(), bins=[bins_a, bins_b])[0])
            dw = dsa.from_array(w, chunks=(xc + 1, yc + 1))
            h, _ = histogram(da, bins=bins_a, weights=dw)
            np.testing.assert_allclose(h.compute(), np.histogram(a, bins=bins_a, weights=w)[0], rtol=1e-6)
            h, _ = histogram(da, bins=bins_a, weights=dw, density=True, axis=1)
            want = np.stack([np.histogram(a[i], bins=bins_a, weights=w[i], density=True)[0] for i in range(10)])
            np.testing.assert_allclose(h.compute(), want, rtol=1e-6)
    t = rng.standard_normal((8, 36, 72)).astype(np.float32)  # C4 miniature: chunks on time
    h, _ = histogram(dsa.from_array(t, chunks=(2, 36, 72)), bins=np.linspace(-4, 4, 51), axis=[1, 2])
    np.testing.assert_array_equal(h.compute(scheduler="threads"), np.stack([np.histogram(t[i], bins=np.linspace(-4, 4, 51))[0] for i in range(8)]))
    print("COMPUTE-OK")


if __name__ == "__main__":
    {"lazy": lazy, "compute": compute}[sys.argv[1]]()
