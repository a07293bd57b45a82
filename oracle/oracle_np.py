"""CPU oracle for the xhistogram binning-reduction hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain numpy, the algorithm of the reference's hot path
(`/root/reference/xhistogram/core.py`).  It is the *checker* for the HIP path, never the
product: only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it.  Nothing under ``xhistogram_amd/`` imports this module.

Parity pinning
--------------
The arithmetic of the reference path lives in numpy's C routines ``searchsorted`` and
``bincount`` (third-party; the reference pins ``numpy>=1.17`` in setup.py:23, and was run here
under numpy 1.26.4).  This restatement is pinned two ways:

* against golden vectors produced by importing the reference itself in the build container
  (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; checked by
  ``tests/test_oracle_golden.py``), and
* against ``np.histogram`` / ``np.histogram2d`` / ``np.histogramdd``, which is what the
  reference's own tests use as their oracle (test_core.py:25-228).

Functions and the reference lines they follow
---------------------------------------------
``digitize_inclusive``      core.py:163-174  (searchsorted side="right" + right-edge fix-up)
``joint_index``             core.py:176-183  (ravel_multi_index over (E_d+1)-sized axes, C order)
``row_bincount``            core.py:73-83    (row-offset trick + bincount, weights cast to f64)
``bincount_rows``           core.py:137-194  (_bincount_2d_vectorized: the whole hot path + trim)
``block_adapter``           core.py:197-247  (_bincount: N-D block -> [rows, cols] -> N-D)
``histogram``               core.py:250-466  (public numpy API, numpy branch only)
``density_normalise``       core.py:444-462  (with the intended outer product for D >= 3)

A second, deliberately naive definition (``bincount_rows_definitional``) uses python loops
and ``bisect`` so that tiny cases are checked without numpy's searchsorted at all.
"""

from __future__ import annotations

import bisect
from functools import reduce
from operator import mul

import numpy as np


# --------------------------------------------------------------------------------------
# L1: digitize -> joint index -> bincount -> trim                       (core.py:137-194)
# --------------------------------------------------------------------------------------
def digitize_inclusive(a, edges):
    """Bin index of every sample against ``edges`` with numpy.histogram's edge rule.

    Follows core.py:163-174.  Returned codes (core.py:157-162):
    0 -> a < edges[0];  i -> edges[i-1] <= a < edges[i];  E-1 also holds a == edges[-1];
    E -> a > edges[-1] or NaN.
    """
    edges = np.asarray(edges)
    a = np.asarray(a)
    code = np.searchsorted(edges, a, side="right")
    code = np.array(code, dtype=np.int64, copy=True)
    hit_last = a == edges[-1]
    code[hit_last] -= 1
    return code


def joint_index(codes, n_edges):
    """Row-major flat index over axes of length E_d + 1; first input is the slowest axis.

    Follows core.py:176-183 (ravel_multi_index over ``hist_shapes``), written as a Horner
    recurrence so that it does not depend on numpy's ravel_multi_index.
    """
    flat = np.zeros_like(codes[0], dtype=np.int64)
    for code, e in zip(codes, n_edges):
        flat = flat * (e + 1) + code
    return flat


def row_bincount(flat, n_internal, weights=None):
    """Per-row bincount of a 2-D index array; unweighted -> int64, weighted -> float64.

    Follows core.py:73-83: every row's indices are shifted into their own span of
    ``n_internal`` slots and a single bincount is taken.  numpy casts any real weight dtype
    to float64 inside bincount and rejects complex.
    """
    rows = flat.shape[0]
    shifted = flat + (np.arange(rows, dtype=np.int64) * n_internal)[:, None]
    if weights is None:
        out = np.bincount(shifted.reshape(-1), minlength=rows * n_internal)
    else:
        out = np.bincount(
            shifted.reshape(-1), weights=np.asarray(weights).reshape(-1), minlength=rows * n_internal
        )
    return out.reshape(rows, n_internal)


def bincount_rows(samples, edges, weights=None):
    """The hot path: independent D-dimensional histogram of every row of [M, C] blocks.

    Follows core.py:137-194 (``_bincount_2d_vectorized``).  ``samples`` is a list of D arrays of
    identical shape [M, C]; ``edges`` a list of D 1-D edge arrays; ``weights`` None or [M, C].
    Returns [M, nb_0, ..., nb_{D-1}] (contiguous; the reference returns a strided view of the
    same values).  ``block_size`` is not a parameter: the reference's row blocking
    (core.py:86-134) never changes the result.
    """
    samples = [np.asarray(s) for s in samples]
    edges = [np.asarray(e) for e in edges]
    if len(samples) != len(edges):
        raise ValueError("one edge array per sample array")
    shape = samples[0].shape
    for s, e in zip(samples, edges):
        if s.ndim != 2 or e.ndim != 1 or s.shape != shape:
            raise AssertionError("samples must be equal-shape 2-D arrays, edges 1-D")  # core.py:146-151
    if weights is not None and np.asarray(weights).shape != shape:
        raise AssertionError("weights must match the sample shape")

    n_edges = [len(e) for e in edges]
    codes = [digitize_inclusive(s, e) for s, e in zip(samples, edges)]
    flat = joint_index(codes, n_edges)
    n_internal = reduce(mul, [e + 1 for e in n_edges], 1)
    full = row_bincount(flat, n_internal, weights)
    full = full.reshape((shape[0],) + tuple(e + 1 for e in n_edges))
    core = (slice(None),) + tuple(slice(1, -1) for _ in n_edges)  # core.py:189-192
    return np.ascontiguousarray(full[core])


def bincount_rows_definitional(samples, edges, weights=None):
    """Loop-and-bisect statement of the same contract (tiny inputs only).

    bin k (0-based, real bins only) holds x with edges[k] <= x < edges[k+1], the last bin also
    x == edges[-1]; NaN, x < edges[0] and x > edges[-1] are dropped; a sample is dropped if it is
    dropped in ANY dimension.  Weighted sums accumulate in float64 in sample order.
    """
    samples = [np.asarray(s) for s in samples]
    edge_lists = [list(np.asarray(e)) for e in edges]
    rows, cols = samples[0].shape
    nb = [len(e) - 1 for e in edge_lists]
    if weights is None:
        out = np.zeros((rows,) + tuple(nb), dtype=np.int64)
    else:
        out = np.zeros((rows,) + tuple(nb), dtype=np.float64)
        weights = np.asarray(weights)
    for r in range(rows):
        for c in range(cols):
            where = []
            for s, e, n in zip(samples, edge_lists, nb):
                x = s[r, c]
                if x != x or x < e[0] or x > e[-1]:
                    where = None
                    break
                k = bisect.bisect_right(e, x) - 1
                where.append(min(k, n - 1))
            if where is None:
                continue
            if weights is None:
                out[(r,) + tuple(where)] += 1
            else:
                out[(r,) + tuple(where)] += np.float64(weights[r, c])
    return out


# --------------------------------------------------------------------------------------
# L2: block adapter                                                     (core.py:197-247)
# --------------------------------------------------------------------------------------
def normalise_axis(axis, ndim):
    """core.py:341-352: None or ints (negative allowed) -> list of non-negative python ints."""
    if axis is None:
        return None
    out = []
    for ax in np.atleast_1d(axis):
        ax = int(ax)
        if ax < 0:
            ax += ndim
        if not 0 <= ax < ndim:
            raise AssertionError("axis must be less than ndim")
        out.append(ax)
    return out


def to_rows_cols(a, axis):
    """core.py:211-227: reduced axes moved last (in the order given), flattened to [M, C]."""
    a = np.asarray(a)
    if axis is None or set(axis) == set(range(a.ndim)):
        return a.reshape(1, -1)
    moved = np.moveaxis(a, axis, tuple(range(-len(axis), 0)))
    keep = moved.shape[: moved.ndim - len(axis)]
    return moved.reshape(int(np.prod(keep, dtype=np.int64)), -1)


def block_adapter(arrays, edges, weights=None, axis=None):
    """core.py:197-247 (``_bincount``): returns kept-axes shape (1 for reduced axes) + bin dims."""
    a0 = np.asarray(arrays[0])
    full = axis is None or set(axis) == set(range(a0.ndim))
    if full:
        kept = (1,) * a0.ndim
    else:
        kept = tuple(1 if i in axis else a0.shape[i] for i in range(a0.ndim))
    s2d = [to_rows_cols(a, axis) for a in arrays]
    w2d = None if weights is None else to_rows_cols(weights, axis)
    counts = bincount_rows(s2d, edges, w2d)
    return counts.reshape(kept + counts.shape[1:])


# --------------------------------------------------------------------------------------
# L3: public numpy API (numpy branch)                                   (core.py:250-466)
# --------------------------------------------------------------------------------------
def format_bins(bins, n):
    """core.py:37-48."""
    if bins is None:
        raise ValueError("bins must be provided")
    if isinstance(bins, (int, str, np.ndarray)):
        bins = n * [bins]
    if len(bins) != n:
        raise ValueError("The number of bin definitions doesn't match the number of args")
    return bins


def format_range(range_, n):
    """core.py:51-70."""
    from collections.abc import Iterable

    if range_ is None:
        return n * [None]
    nested = all(isinstance(i, Iterable) for i in range_)
    if len(range_) == 2 and not nested:
        return n * [range_]
    if len(range_) == n:
        if all(len(x) == 2 for x in range_):
            return range_
        raise ValueError("range should be (lower, upper) or a list of such tuples, one per arg")
    raise ValueError("The number of ranges doesn't match the number of args")


def density_normalise(counts, edges):
    """core.py:444-462; D >= 3 uses the outer product the reference intended (its
    ``np.prod(np.ix_(...))`` at core.py:454 fails on numpy >= 1.24)."""
    widths = [np.diff(e) for e in edges]
    d = len(edges)
    area = widths[0]
    for w in widths[1:]:
        area = np.multiply.outer(area, w)
    bin_axes = tuple(range(-d, 0))
    totals = counts.sum(axis=bin_axes)
    totals = totals.reshape(totals.shape + d * (1,))
    with np.errstate(divide="ignore", invalid="ignore"):
        return counts / area / totals


def histogram(*args, bins=None, range=None, axis=None, weights=None, density=False, block_size="auto"):
    """numpy branch of core.py:250-466.  ``block_size`` is accepted and does not change results."""
    n = len(args)
    a0 = np.asarray(args[0])
    axis = normalise_axis(axis, a0.ndim)
    arrays = [np.asarray(a) for a in args]
    if weights is not None:
        arrays.append(np.asarray(weights))
    arrays = np.broadcast_arrays(*arrays)  # core.py:366
    w = arrays[-1] if weights is not None else None
    bins = format_bins(bins, n)
    range = format_range(range, n)
    edges = [
        np.histogram_bin_edges(a, bins=b, range=r, weights=w)  # core.py:383-388
        for a, b, r in zip(arrays, bins, range)
    ]
    drop = tuple(axis) if axis is not None else tuple(np.arange(arrays[0].ndim))
    counts = block_adapter(list(arrays[:n]), edges, w, axis).squeeze(tuple(int(i) for i in drop))
    if density:
        return density_normalise(counts, edges), edges
    return counts, edges


# --------------------------------------------------------------------------------------
# Throughput leg used by bench.py's cpu_baseline (mirrors the reference's dask-threaded path:
# per-chunk hot path + sum over chunks, core.py:429-439)
# --------------------------------------------------------------------------------------
def chunked_threaded(samples, edges, weights=None, chunk=5_000_000, threads=1):
    from concurrent.futures import ThreadPoolExecutor

    n = samples[0].shape[-1]
    spans = [(i, min(i + chunk, n)) for i in np.arange(0, n, chunk)]

    def one(span):
        lo, hi = span
        s = [x[lo:hi].reshape(1, -1) for x in samples]
        w = None if weights is None else weights[lo:hi].reshape(1, -1)
        return bincount_rows(s, edges, w)

    if threads <= 1:
        parts = [one(s) for s in spans]
    else:
        with ThreadPoolExecutor(threads) as ex:
            parts = list(ex.map(one, spans))
    return reduce(np.add, parts)
