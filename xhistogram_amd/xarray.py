"""xarray API of the MI355X-native histogram: label handling around ``core.histogram``.

Drop-in for ``xhistogram.xarray.histogram`` (reference: /root/reference/xhistogram/xarray.py:13-201)
with the same signature, output dims / coords / name and errors.  This module does no arithmetic
besides the bin centres; the data arrays (numpy, dask, or GPU-resident) go straight to
:func:`xhistogram_amd.core.histogram`.
"""

from __future__ import annotations

from .core import histogram as _core_histogram
from .core import histogram_two_weights as _core_histogram_two_weights

__all__ = ["histogram"]


def _xr():
    try:
        import xarray
    except ImportError as e:  # the reference hard-imports xarray at module import (xarray.py:5)
        raise ImportError("xhistogram_amd.xarray.histogram needs the xarray package") from e
    return xarray


def histogram(*args, bins=None, range=None, dim=None, weights=None, density=False, block_size="auto",
              keep_coords=False, bin_dim_suffix="_bin"):
    """Histogram applied along specified dimensions.

    Parameters (identical to the reference, xarray.py:24-100)
    ----------
    args : xarray.DataArray objects
        Input data; N arguments give an N-dimensional histogram.  All must be named and alignable
        (``join="exact"``); they are broadcast against each other by dimension name.
    bins, range, weights, density, block_size
        As in :func:`xhistogram_amd.core.histogram`.  ``weights`` is a DataArray whose dims are a
        subset of the data's — or (an extension: the reference's TODO at xarray.py:106) a PAIR of
        such DataArrays, binned in one pass over the data; the result is then a pair of
        DataArrays (mean of ``A`` in the bins = ``h[0] / h[1]`` for ``weights=(A * w, w)``).
    dim : tuple of strings, optional
        Dimensions to histogram over; default all.
    keep_coords : bool
        Carry over coordinates compatible with the output dims.
    bin_dim_suffix : str
        Output bin dimensions are named ``<arg name> + bin_dim_suffix``.

    Returns
    -------
    xarray.DataArray named ``histogram_<name0>_<name1>…`` with dims = kept dims + bin dims and the
    bin midpoints as coordinates of the bin dims (carrying the inputs' attrs).
    """
    xr = _xr()
    data_args = list(args)
    n_data = len(data_args)
    for a in data_args:  # xarray.py:109-117
        if not isinstance(a, xr.DataArray):
            raise TypeError(
                "xhistogram.xarray.histogram accepts only xarray.DataArray objects but a %s was provided" % type(a).__name__
            )
    for a in data_args:
        assert a.name is not None, "all arrays must have a name"

    operands = list(data_args)
    if not keep_coords:  # coordinates only get in the way of alignment (xarray.py:119-123)
        operands = [a.reset_coords(drop=True) for a in operands]
    pair = isinstance(weights, (tuple, list))
    if pair:
        if len(weights) != 2:
            raise ValueError("weights must be one DataArray or a pair of them")
        if density:
            raise ValueError("density is not defined for a pair of weights")
    w_ops = list(weights) if pair else ([] if weights is None else [weights])
    operands.extend(w.reset_coords(drop=True) for w in w_ops)
    operands = list(xr.align(*operands, join="exact"))  # xarray.py:126
    first = operands[0]
    first_coords = first.coords

    # broadcast by name: union of dims in first-seen order, missing dims inserted with length 1
    # (the core broadcasts them without copying) — xarray.py:133-150
    dims_order = []
    for a in operands:
        for d in a.dims:
            if d not in dims_order:
                dims_order.append(d)
    lined_up = []
    for a in operands:
        missing = [d for d in dims_order if d not in a.dims]
        if missing:
            a = a.expand_dims({d: 1 for d in missing})
        if tuple(a.dims) != tuple(dims_order):
            a = a.transpose(*dims_order)
        lined_up.append(a)
    arrays = [a.data for a in lined_up]
    w_data = [arrays.pop() for _ in w_ops][::-1]

    if dim is not None:  # xarray.py:157-162
        kept_dims = [d for d in dims_order if d not in dim]
        axis = [lined_up[0].get_axis_num(d) for d in dim]
    else:
        kept_dims = []
        axis = None

    if pair:
        *h_all, edges = _core_histogram_two_weights(
            *arrays, weights=tuple(w_data), bins=bins, range=range, axis=axis, block_size=block_size
        )
    else:
        h_data, edges = _core_histogram(
            *arrays, weights=w_data[0] if w_data else None, bins=bins, range=range, axis=axis, density=density,
            block_size=block_size
        )
        h_all = [h_data]

    # output labels (xarray.py:174-201)
    bin_dims = [a.name + bin_dim_suffix for a in operands[:n_data]]
    out_dims = kept_dims + bin_dims
    coords = {name: first[name] for name in kept_dims if name in first_coords}
    for name, e, a in zip(bin_dims, edges, operands):
        coords[name] = ((name,), 0.5 * (e[:-1] + e[1:]), a.attrs)
    if keep_coords:
        for c in first_coords:
            if c not in coords and set(first[c].dims).issubset(out_dims):
                coords[c] = first[c]
    out_name = "_".join(["histogram"] + [a.name for a in operands[:n_data]])
    out = tuple(xr.DataArray(h, dims=out_dims, coords=coords, name=out_name) for h in h_all)
    return out if pair else out[0]
