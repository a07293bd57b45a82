"""Generate golden input/output vectors by RUNNING THE REFERENCE in the build container.

Run (build container only; the reference never travels to the GPU box):

    PYTHONPATH=/root/reference /opt/conda/bin/python3.9 tests/golden/make_golden.py

Writes ``tests/golden/hotpath.npz`` (direct calls of the reference's
``_bincount_2d_vectorized``, core.py:137-194), ``tests/golden/core.npz`` (calls of the public
``xhistogram.core.histogram``, core.py:250-466, numpy and dask branches) and
``tests/golden/manifest.json`` (non-array parameters + interpreter/numpy/dask versions).
The fixtures are data only: inputs and the outputs the reference produced for them.

f32 note: under numpy < 2 the ``a == b[-1]`` test of core.py:171 is evaluated
in float32 for float32 data; the build follows numpy >= 2 / NEP 50 (float64 compare).  All
float32 cases here use a last edge that is exactly representable in float32, where both agree.
"""

import json
import os
import sys

import numpy as np

import xhistogram.core as ref  # the reference (PYTHONPATH=/root/reference)

try:
    import dask
    import dask.array as dsa
except ImportError:  # pragma: no cover
    dsa = None

HERE = os.path.dirname(os.path.abspath(__file__))

SPECIALS = np.array(
    [np.nan, -np.inf, np.inf, -0.0, 0.0, 1.0, 2.0, 4.0, 3.9999999999999996, 4.000000000000001, -1e-300, 0.5],
    dtype=np.float64,
)


def hotpath_cases():
    """name -> (list of [M,C] arrays, list of edges, weights or None)."""
    out = {}
    r = np.random.default_rng(11)
    out["f64_uniform_1row"] = ([r.standard_normal((1, 4096))], [np.linspace(-4, 4, 101)], None)
    out["f64_uniform_rows"] = ([r.standard_normal((7, 513))], [np.linspace(-3, 3, 31)], None)
    e = np.sort(r.uniform(-4, 4, 33))
    e[0], e[-1] = -4.0, 4.0
    out["f64_nonuniform"] = ([r.standard_normal((3, 1000)) * 2], [e], None)
    out["f64_weighted"] = ([r.standard_normal((4, 777))], [np.linspace(-4, 4, 51)], r.uniform(0, 1, (4, 777)))
    out["specials_edges_0124"] = ([SPECIALS.reshape(1, -1)], [np.array([0.0, 1.0, 2.0, 4.0])], None)
    out["specials_weighted_nan"] = (
        [SPECIALS.reshape(1, -1)],
        [np.array([0.0, 1.0, 2.0, 4.0])],
        np.array([[np.nan, 1, 1, 2, 3, 4, np.nan, 5, 6, np.nan, 7, 8.5]], dtype=np.float64),
    )
    out["duplicate_edges"] = (
        [np.array([[0.0, 0.5, 1.0, 1.0, 1.5, 2.0, 2.0, -1.0, 3.0]])],
        [np.array([0.0, 1.0, 1.0, 2.0, 2.0])],
        None,
    )
    out["right_edge_all_ones"] = ([np.ones((5, 20))], [np.array([0.0, 0.5, 1.0])], None)
    out["f32_data"] = (
        [r.standard_normal((3, 2048)).astype(np.float32)],
        [np.linspace(-4, 4, 51)],
        None,
    )
    out["f32_data_f32_weights"] = (
        [r.standard_normal((2, 999)).astype(np.float32)],
        [np.linspace(-4, 4, 11)],
        r.uniform(0, 2, (2, 999)).astype(np.float32),
    )
    out["i32_data"] = ([r.integers(-20, 20, (3, 500)).astype(np.int32)], [np.linspace(-10, 10, 21)], None)
    out["i64_data"] = ([r.integers(-20, 20, (2, 500)).astype(np.int64)], [np.arange(-10.5, 11.0, 1.0)], None)
    out["u8_data"] = ([r.integers(0, 255, (2, 300)).astype(np.uint8)], [np.linspace(0, 255, 18)], None)
    out["bool_weights"] = ([r.standard_normal((2, 400))], [np.linspace(-2, 2, 9)], r.integers(0, 2, (2, 400)).astype(bool))
    out["int_weights"] = ([r.standard_normal((2, 400))], [np.linspace(-2, 2, 9)], r.integers(-3, 9, (2, 400)).astype(np.int64))
    a = r.standard_normal((5, 640))
    b = r.standard_normal((5, 640))
    out["2d_uniform"] = ([a, b], [np.linspace(-4, 4, 10), np.linspace(-4, 4, 11)], None)
    a2, b2 = a.copy(), b.copy()
    a2[0, :7] = np.nan
    b2[1, 3:9] = np.nan
    b2[0, 5] = np.inf
    out["2d_nan_one_arg"] = ([a2, b2], [np.linspace(-4, 4, 10), np.linspace(-4, 4, 11)], r.uniform(0, 1, (5, 640)))
    ea = np.sort(r.uniform(-4, 4, 17))
    eb = np.sort(r.uniform(-4, 4, 23))
    out["2d_nonuniform"] = ([a, b], [ea, eb], None)
    c = r.standard_normal((5, 640))
    out["3d"] = ([a, b, c], [np.linspace(-4, 4, 10), np.linspace(-4, 4, 11), np.linspace(-3, 3, 7)], None)
    d = r.standard_normal((5, 640))
    out["4d_weighted"] = (
        [a, b, c, d],
        [np.linspace(-4, 4, 8), np.linspace(-4, 4, 9), np.linspace(-4, 4, 10), np.linspace(-4, 4, 11)],
        r.uniform(0, 1, (5, 640)),
    )
    out["i64_datetime_like"] = (
        [np.array([[959817600000000000 + k * 86400000000000 for k in range(5)]], dtype=np.int64)],
        [np.array([915148800000000000, 946684800000000000, 978307200000000000], dtype=np.int64)],
        None,
    )
    out["big_int64_vs_int_edges"] = (
        [np.array([[2**60, 2**60 + 1, 2**60 + 2, 2**60 + 3, -(2**62)]], dtype=np.int64)],
        [np.array([2**60, 2**60 + 2, 2**60 + 3], dtype=np.int64)],
        None,
    )
    out["one_sample"] = ([np.array([[0.25]])], [np.array([0.0, 1.0])], None)
    out["empty_cols"] = ([np.zeros((3, 0))], [np.linspace(0, 1, 5)], None)
    out["single_bin"] = ([r.uniform(-1, 2, (2, 100))], [np.array([0.0, 1.0])], None)
    out["wide_bins_1k"] = ([r.standard_normal((1, 20000))], [np.linspace(-4, 4, 1025)], None)
    return out


def core_cases():
    """name -> dict(args, kwargs) for the public API."""
    r = np.random.default_rng(23)
    out = {}
    x = r.standard_normal((5, 20))
    xn = x.copy()
    xn.ravel()[r.choice(x.size, 20, replace=False)] = np.nan
    b9 = np.linspace(-4, 4, 10)
    for dens in (False, True):
        for ax in (None, 1, 0, -1, (0, 1)):
            for nm, data in (("x", x), ("xnan", xn)):
                key = "1d_%s_axis%s_dens%d" % (nm, str(ax).replace(" ", ""), dens)
                out[key] = dict(args=[data], kw=dict(bins=b9, axis=ax, density=dens, block_size=None))
    for bs in (None, 1, 2, "auto"):
        out["blocksize_%s" % bs] = dict(args=[x], kw=dict(bins=b9, axis=1, block_size=bs))
    out["weights_full"] = dict(args=[x], kw=dict(bins=b9, axis=1, weights=r.uniform(0, 1, x.shape), block_size=None))
    out["weights_bcast_row"] = dict(args=[x], kw=dict(bins=b9, axis=1, weights=2 * np.ones((1, 20)), block_size=None))
    out["weights_bcast_col"] = dict(args=[x], kw=dict(bins=b9, axis=1, weights=r.uniform(0, 1, (5, 1)), block_size=None))
    out["weights_density"] = dict(args=[x], kw=dict(bins=b9, axis=1, weights=r.uniform(0, 1, x.shape), density=True, block_size=None))
    out["bins_int"] = dict(args=[x], kw=dict(bins=10, block_size=None))
    out["bins_int_range"] = dict(args=[x], kw=dict(bins=12, range=(-2, 2), block_size=None))
    out["bins_int_axis_range"] = dict(args=[xn], kw=dict(bins=7, range=(-3, 3), axis=0, block_size=None))
    y = r.standard_normal((5, 20))
    out["2d_bins_list"] = dict(args=[x, y], kw=dict(bins=[b9, np.linspace(-4, 4, 11)], block_size=None))
    out["2d_bins_int_ranges"] = dict(args=[x, y], kw=dict(bins=[5, 6], range=[(-2, 2), (-3, 3)], block_size=None))
    out["2d_bcast_args"] = dict(args=[r.standard_normal(20), y], kw=dict(bins=[b9, np.linspace(-4, 4, 11)], block_size=None))
    out["2d_density_axis1"] = dict(args=[x, y], kw=dict(bins=[b9, np.linspace(-4, 4, 11)], axis=1, density=True, block_size=None))
    z4 = r.standard_normal((3, 4, 5, 6))
    b27 = np.linspace(-4, 4, 27)
    for ax in ((0,), (2,), (1, 3), (3, 1), (0, 1, 2), (0, 1, 2, 3), (3, 2, 0, 1), (-1,), (-2, 0)):
        out["4dshape_axis%s" % "_".join(map(str, ax))] = dict(args=[z4], kw=dict(bins=b27, axis=ax, block_size=None))
    out["4dshape_weights_axis13"] = dict(
        args=[z4], kw=dict(bins=b27, axis=(1, 3), weights=r.uniform(0, 1, z4.shape), block_size=None)
    )
    e = np.zeros((4, 6))
    e[1:] = r.uniform(0.1, 0.9, (3, 6))
    e[0] = 7.0  # row entirely out of range -> density NaN row
    out["density_empty_row"] = dict(args=[e], kw=dict(bins=np.linspace(0, 1, 5), axis=1, density=True, block_size=None))
    out["f32_axis"] = dict(args=[r.standard_normal((6, 50)).astype(np.float32)], kw=dict(bins=np.linspace(-4, 4, 51), axis=1, block_size=None))
    return out


def dask_cases():
    r = np.random.default_rng(37)
    out = {}
    a = r.standard_normal((10, 12))
    b = r.standard_normal((10, 12))
    w = r.uniform(0, 1, (10, 12))
    ea, eb = np.linspace(-4, 4, 9), np.linspace(-4, 4, 10)
    out["dask_full_reduce"] = dict(args=[a], chunks=[(3, 5)], kw=dict(bins=ea))
    out["dask_axis0"] = dict(args=[a], chunks=[(3, 5)], kw=dict(bins=ea, axis=0))
    out["dask_axis1_weights_unaligned"] = dict(args=[a], chunks=[(3, 5)], wchunks=(4, 6), weights=w, kw=dict(bins=ea, axis=1))
    out["dask_2d_unaligned"] = dict(args=[a, b], chunks=[(2, 3), (3, 4)], kw=dict(bins=[ea, eb]))
    out["dask_2d_density"] = dict(args=[a, b], chunks=[(5, 12), (5, 12)], kw=dict(bins=[ea, eb], density=True))
    t = r.standard_normal((8, 6, 10)).astype(np.float32)
    out["dask_time_chunks_c4_mini"] = dict(args=[t], chunks=[(2, 6, 10)], kw=dict(bins=np.linspace(-4, 4, 51), axis=[1, 2]))
    return out


def jsonable(v):
    if isinstance(v, np.ndarray):
        return {"__array__": True}
    if isinstance(v, (list, tuple)):
        return [jsonable(i) for i in v]
    if isinstance(v, (np.integer,)):
        return int(v)
    return v


def signatures():
    """the reference's public / hot-path signatures as text (north_star: "exact signatures"): core functions through
    inspect; xarray.histogram from the SOURCE with ast — the module cannot be imported here (no xarray in any interpreter)"""
    import ast
    import inspect

    out = {"core.%s" % n: str(inspect.signature(getattr(ref, n))) for n in ("histogram", "_bincount", "_bincount_2d_vectorized")}
    src = os.path.join(os.path.dirname(ref.__file__), "xarray.py")
    for node in ast.parse(open(src).read()).body:
        if isinstance(node, ast.FunctionDef) and node.name == "histogram":
            out["xarray.histogram"] = "(" + ast.unparse(node.args) + ")"
    return out


def main():
    if "--signatures-only" in sys.argv:  # refresh manifest["signatures"] without rewriting the .npz archives
        path = os.path.join(HERE, "manifest.json")
        manifest = json.load(open(path))
        manifest["signatures"] = signatures()
        with open(path, "w") as f:
            json.dump(manifest, f, indent=1, sort_keys=True)
        print("wrote", len(manifest["signatures"]), "signatures")
        return
    manifest = {
        "python": sys.version.split()[0],
        "numpy": np.__version__,
        "dask": getattr(dask, "__version__", None) if dsa is not None else None,
        "reference": "xgcm/xhistogram @ /root/reference (xhistogram/core.py)",
        "hotpath": {},
        "core": {},
        "dask_cases": {},
        "signatures": signatures(),
    }
    hp = {}
    for name, (samples, edges, weights) in hotpath_cases().items():
        got = ref._bincount_2d_vectorized(*samples, bins=edges, weights=weights)
        for i, s in enumerate(samples):
            hp["%s/s%d" % (name, i)] = s
        for i, e in enumerate(edges):
            hp["%s/e%d" % (name, i)] = e
        if weights is not None:
            hp["%s/w" % name] = weights
        hp["%s/out" % name] = np.ascontiguousarray(got)
        manifest["hotpath"][name] = {"D": len(samples), "weighted": weights is not None, "out_dtype": str(got.dtype)}
    np.savez_compressed(os.path.join(HERE, "hotpath.npz"), **hp)

    co = {}
    for name, case in core_cases().items():
        kw = dict(case["kw"])
        h, edges = ref.histogram(*case["args"], **kw)
        for i, a in enumerate(case["args"]):
            co["%s/a%d" % (name, i)] = a
        meta = {}
        for k, v in kw.items():
            if isinstance(v, np.ndarray):
                co["%s/kw_%s" % (name, k)] = v
                meta[k] = {"__array__": True}
            elif k == "bins" and isinstance(v, list) and isinstance(v[0], np.ndarray):
                for i, e in enumerate(v):
                    co["%s/kw_bins%d" % (name, i)] = e
                meta[k] = {"__array_list__": len(v)}
            else:
                meta[k] = jsonable(v)
        co["%s/h" % name] = np.ascontiguousarray(h)
        for i, e in enumerate(edges):
            co["%s/edges%d" % (name, i)] = e
        manifest["core"][name] = {"n_args": len(case["args"]), "kw": meta, "h_dtype": str(h.dtype), "h_shape": list(h.shape)}

    if dsa is not None:
        for name, case in dask_cases().items():
            kw = dict(case["kw"])
            dargs = [dsa.from_array(a, chunks=c) for a, c in zip(case["args"], case["chunks"])]
            if "weights" in case:
                kw["weights"] = dsa.from_array(case["weights"], chunks=case["wchunks"])
            h, edges = ref.histogram(*dargs, **kw)
            h = np.asarray(h.compute())
            for i, a in enumerate(case["args"]):
                co["%s/a%d" % (name, i)] = a
            if "weights" in case:
                co["%s/kw_weights" % name] = case["weights"]
            bins = case["kw"]["bins"]
            meta = {k: jsonable(v) for k, v in case["kw"].items() if k != "bins"}
            if isinstance(bins, list):
                for i, e in enumerate(bins):
                    co["%s/kw_bins%d" % (name, i)] = e
                meta["bins"] = {"__array_list__": len(bins)}
            else:
                co["%s/kw_bins" % name] = bins
                meta["bins"] = {"__array__": True}
            if "weights" in case:
                meta["weights"] = {"__array__": True}
            co["%s/h" % name] = np.ascontiguousarray(h)
            manifest["dask_cases"][name] = {
                "n_args": len(case["args"]),
                "kw": meta,
                "chunks": [list(c) for c in case["chunks"]],
                "wchunks": list(case.get("wchunks", [])),
                "h_dtype": str(h.dtype),
                "h_shape": list(h.shape),
            }
    np.savez_compressed(os.path.join(HERE, "core.npz"), **co)
    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("wrote", len(manifest["hotpath"]), "hot-path,", len(manifest["core"]), "core,", len(manifest["dask_cases"]), "dask cases")


if __name__ == "__main__":
    main()
