"""Randomised differential soak of the public entry points against the numpy oracle (development tool;
the test-suite runs seeded subsets of the same space).  python tools/soak.py [seconds] [seed] [max columns]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from oracle import oracle_np as onp
from xhistogram_amd import core

trace = open(os.environ["SOAK_TRACE"], "w") if os.environ.get("SOAK_TRACE") else None
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
big_cols = int(sys.argv[3]) if len(sys.argv) > 3 else 200_000  # upper bound of the last axis in the "big" third of the cases


def edges_for(rng, kind, nb, lo=-3.0, hi=3.0):
    if kind == "linspace":
        return np.linspace(lo, hi, nb + 1)
    if kind == "random":
        e = np.sort(rng.uniform(lo, hi, nb + 1))
        e[0], e[-1] = lo, hi
        return e
    if kind == "geom":
        return lo + np.concatenate([[0.0], np.cumsum(np.geomspace(1e-6, 1.0, nb))]) * (hi - lo) / np.geomspace(1e-6, 1.0, nb).sum()
    if kind == "logspace":  # geometric edges: the float-bit-pattern bucket grid of the packed entries (round 4)
        return np.geomspace(10.0 ** float(rng.integers(-6, 0)), hi * float(rng.choice([1.0, 3.0, 1e3])), nb + 1)
    if kind == "symlog":  # edges on both sides of zero, logarithmic away from it
        h = max(1, nb // 2)
        pos = np.geomspace(10.0 ** float(rng.integers(-5, -1)), hi, h)
        e = np.concatenate([-pos[::-1], [0.0], pos]) if rng.random() < 0.5 else np.concatenate([-pos[::-1] * float(rng.choice([1.0, 0.3])), pos])
        return e
    if kind == "int":
        return np.arange(-nb // 2, nb - nb // 2 + 1).astype(np.int64)
    raise ValueError(kind)


def one(seed):
    rng = np.random.default_rng(seed)
    ndim = int(rng.integers(1, 4))
    big = rng.random() < 0.3
    shape = tuple(int(rng.integers(1, 7)) for _ in range(ndim - 1)) + (int(rng.integers(1, big_cols if big else 3000)),)
    if rng.random() < 0.3:
        shape = tuple(rng.permutation(shape))
    if rng.random() < 0.08:  # many short rows (the flat / row-per-lane kernels)
        shape = (int(rng.integers(4096, 12000)), int(rng.integers(1, 1200)))
        ndim = 2
    d = int(rng.choice([1, 1, 1, 2, 2, 3]))
    dtype = rng.choice(["f64", "f32", "i32", "i64", "u8", "f16"])
    # a third of the joint histograms mix dtypes (the mixed-dtype vector kernels / the generic family)
    dtypes = [str(rng.choice(["f64", "f32", "i32", "i16", "u8", "f16"])) if (d > 1 and rng.random() < 0.35) else str(dtype) for _ in range(d)]
    kinds = ["int" if dt in ("i32", "i64", "u8", "i16") and rng.random() < 0.5 else str(rng.choice(["linspace", "linspace", "random", "random", "geom", "logspace", "symlog"])) for dt in dtypes]
    nbmax = {1: 70_000, 2: 400, 3: 50}[d]
    nbs = [int(rng.choice([1, 3, 17, 100, int(rng.integers(1, nbmax))])) for _ in range(d)]
    edges = [edges_for(rng, k, nb) for k, nb in zip(kinds, nbs)]
    npdts = {"f64": np.float64, "f32": np.float32, "i32": np.int32, "i64": np.int64, "u8": np.uint8, "f16": np.float16, "i16": np.int16}
    args = []
    for dt in dtypes:
        npdt = npdts[dt]
        a = rng.standard_normal(shape) * 2
        if npdt in (np.int32, np.int64, np.int16):
            a = np.round(a * 20)
        elif npdt == np.uint8:
            a = np.abs(np.round(a * 20)) % 256
        a = a.astype(npdt)
        if a.dtype.kind == "f" and a.size > 8 and rng.random() < 0.4:
            # some samples ON an edge, or next to one in the sample's own precision (the exact-redo path of the packed entries,
            # the float32 thresholds, the right-edge rule)
            e = edges[len(args)].astype(np.float64)
            k = int(rng.integers(1, max(2, a.size // 4)))
            pick = rng.choice(e, size=k).astype(npdt)
            if rng.random() < 0.5:
                pick = np.nextafter(pick, npdt(rng.choice([-np.inf, np.inf])))
            a.reshape(-1)[rng.integers(0, a.size, k)] = pick
        if a.dtype.kind == "f" and a.size > 3 and rng.random() < 0.5:
            a.reshape(-1)[rng.integers(0, a.size, 3)] = [np.nan, np.inf, -np.inf]
        args.append(a)
    axes = [None] + [tuple(c) for r in range(1, ndim + 1) for c in __import__("itertools").combinations(range(ndim), r)]
    axis = axes[int(rng.integers(0, len(axes)))]
    wkind = rng.choice(["none", "none", "full", "bcast", "scalar"])
    w = None
    if wkind == "full":
        wdt = rng.choice(["f64", "f64", "f32", "i32", "bool", "f16"])
        w = rng.uniform(-1, 2, shape)
        w = {"f64": lambda v: v, "f32": lambda v: v.astype(np.float32), "i32": lambda v: np.round(v * 4).astype(np.int32),
             "bool": lambda v: v > 0.5, "f16": lambda v: v.astype(np.float16)}[str(wdt)](w)
    elif wkind == "bcast":
        wshape = tuple(n if rng.random() < 0.5 else 1 for n in shape)
        w = rng.uniform(0, 2, wshape)
    elif wkind == "scalar":
        w = np.full((1,) * ndim, 0.75)
    kept = 1 if axis is None else int(np.prod([n for i, n in enumerate(shape) if i not in axis]))
    if kept * int(np.prod(nbs)) > 4_000_000:
        return None  # keep outputs (and the oracle's temporaries) small
    density = bool(rng.random() < 0.25)
    resident = bool(rng.random() < 0.6) and not any(a.dtype == np.float16 and False for a in args)
    two = w is not None and w.dtype.kind == "f" and w.dtype.itemsize >= 4 and rng.random() < 0.25 and not density
    bins = edges if d > 1 else edges[0]
    desc = dict(seed=seed, shape=shape, d=d, dtype=dtypes, wdtype=None if w is None else str(w.dtype), kinds=kinds, nbs=nbs, axis=axis, wkind=wkind, density=density, resident=resident, two=two)
    try:
        want = onp.histogram(*args, bins=bins, weights=w, axis=axis, density=density)[0]
    except (NotImplementedError, TypeError, ValueError):
        return None
    conv = (lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()) if resident else (lambda a: a)
    if resident and rng.random() < 0.35:  # device-resident without torch: DeviceArray in, numpy out
        from xhistogram_amd.devicearray import DeviceArray
        conv, resident = DeviceArray.from_numpy, False
        desc["devicearray"] = True
    # kernel-family overrides on the plan these edges map to (reset afterwards)
    plan, override = None, None
    if rng.random() < 0.5:
        override = [("partition", 1), ("slices", 1), ("arith", 1), ("lanes", 1), ("lanes", -1), ("force_generic", 1), ("force_global", 1),
                    ("lds_copies", 1), ("slices", -1), ("arith", -1), ("fused", -1), ("partition", 1),
                    ("partition", 1, "route_spl", 8), ("partition", 1, "route_spl", 4), ("partition", 1, "route_spl", 4)][int(rng.integers(0, 15))]
        try:
            dom, cedges, _ = core._compare_domain([a.dtype for a in args], edges)
            plan = core._get_plan(cedges, dom, 0)
            for i in range(0, len(override), 2):
                plan.set_param(override[i], override[i + 1])
        except (NotImplementedError, TypeError):
            plan = None
    desc["override"] = override
    if trace:
        trace.write("  %r\n" % (desc,))
        trace.flush()
    try:
        return _compare(args, bins, w, axis, density, resident, two, conv, want, desc, rng)
    finally:
        if plan is not None:
            for i in range(0, len(override), 2):
                plan.set_param(override[i], 0)


def _compare(args, bins, w, axis, density, resident, two, conv, want, desc, rng):
    try:
        if two:
            w2 = rng.uniform(0, 1, w.shape)
            ha, hb, _ = core.histogram_two_weights(*[conv(a) for a in args], bins=bins, weights=(conv(w), conv(w2)), axis=axis)
            got = ha
            gotb = hb.cpu().numpy() if resident else hb
            wantb = onp.histogram(*args, bins=bins, weights=w2, axis=axis)[0]
            if not np.allclose(gotb, wantb, rtol=1e-6, atol=1e-9, equal_nan=True):
                return dict(desc, where="second weights")
        else:
            got = core.histogram(*[conv(a) for a in args], bins=bins, weights=None if w is None else conv(w), axis=axis, density=density)[0]
    except NotImplementedError:
        return None
    got = got.cpu().numpy() if resident else got
    if got.shape != np.shape(want):
        return dict(desc, where="shape %s vs %s" % (got.shape, np.shape(want)))
    if w is None and not density:
        ok = np.array_equal(got, want)
    else:
        ok = np.allclose(got, want, rtol=1e-6, atol=1e-9, equal_nan=True)
    return None if ok else dict(desc, where="values", maxdiff=float(np.nanmax(np.abs(np.asarray(got, dtype=np.float64) - want))))


t0 = time.time()
n = 0
seed = seed0
while time.time() - t0 < budget:
    if trace:  # a GPU memory fault kills the process: the last seed written is the culprit (or the one before it)
        trace.write("%d\n" % seed)
        trace.flush()
    bad = one(seed)
    if trace:
        torch.cuda.synchronize()
    if bad is not None:
        print("MISMATCH", bad, flush=True)
        sys.exit(1)
    n += 1
    seed += 1
print("soak ok: %d random cases in %.0f s (seeds %d..%d)" % (n, time.time() - t0, seed0, seed - 1))
