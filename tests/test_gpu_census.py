"""Every kernel instantiation the library can dispatch is compared with the oracle (VERDICT r5 "next" #2).

The dispatch surface of libxhist_amd.so is a cross product — sample dtype x weight dtype x inputs x digitize form x histogram
home x row shape — of which the other GPU tests, written around BASELINE's configs and the reference's known answers, select a
third.  This module walks that product on purpose: for every (dtype, weights, inputs) it builds the smallest inputs that make
the pickers of xhist_exec_device.hip.h choose each digitize form (through the EDGES: one, two, three, four or many edges per
bucket of the plan's grid, np.linspace edges, geometric edges) and each histogram home (through the BIN COUNT and the documented
plan parameters that the C ABI exposes: force_global, partition, fused, slices, lanes, flat_rows, pack, arith, route_spl), runs
the call through the C ABI on resident data and compares with oracle_np.bincount_rows — what the reference computes with
searchsorted + ravel_multi_index + bincount (/root/reference/xhistogram/core.py:163-183, :73-83).  Samples sit on edges, next to
edges, outside the range and on NaN; the column count leaves a ragged last tile.

With XHIST_AMD_KERNEL_LOG set (tests/conftest.py sets it for `-m gpu` runs) the library logs the symbol of every kernel it
picks; tests/test_zz_gpu_census_total.py (the last-sorted module) then holds the session's log against the host stubs of the
shared object: an instantiation that nothing selected fails the run — it has to be given a case here or leave the library.
The product of the float families (`_cases`, ~9 200 cases) is run in full only in discovery mode; the suite runs the set cover of
it that tests/golden/census_cases.json names (tools/census_cover.py), plus the special families below, which are always complete.
"""
import os

import numpy as np
import pytest

from conftest import assert_hist_equal
from oracle import oracle_np as onp
from test_gpu_parity import _dev, xh  # noqa: F401  (xh: the module fixture)

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F64, F32 = np.float64, np.float32


# ---------------------------------------------------------------------------------------------------------------------
# edges that decide the digitize form
# ---------------------------------------------------------------------------------------------------------------------
def _grid_buckets(n_edges):
    """buckets of the plan's fine (uint16) table over [e0, eL]: xhist_plan.hip.h build (K = 2 * min(4096, max(8, pow2(4 E))))"""
    k = 8
    while k < min(4 * n_edges, 1 << 20):
        k *= 2
    return 2 * min(4096, max(8, k))


def edges_of(kind, nb, lo=-4.0, hi=4.0, seed=0):
    """nb bins on [lo, hi] whose edges put 1 / 2 / 3 / 4 / many edges into some bucket of the plan's grid ("k1" ... "k4",
    "crowd"), np.linspace edges ("lin": the arithmetic forms), or geometric ones ("geom": the float-bits grid)"""
    rng = np.random.default_rng(seed)
    if kind == "lin":
        return np.linspace(lo, hi, nb + 1)
    if kind == "geom":  # (four decades on the positive axis, whatever lo / hi say: a linear grid cannot separate these)
        return np.geomspace(1e-2, 1e2, nb + 1)
    per = {"k1": 1, "k2": 2, "k3": 3, "k4": 4, "crowd": 7}[kind]
    extra = per - 1
    base_n = nb + 1 - extra
    assert base_n >= 3, (kind, nb)
    base = np.linspace(lo, hi, base_n)
    # not arithmetic: every interior edge moved by a small fraction of the bin width (still one per bucket)
    width = (hi - lo) / (base_n - 1)
    base[1:-1] += rng.uniform(-0.2, 0.2, base_n - 2) * min(width, (hi - lo) / _grid_buckets(nb + 1))
    if extra:  # `per` edges inside ONE bucket of the grid: between 1/4 and 3/4 of the bucket that holds the middle edge
        j = base_n // 2
        bw = (hi - lo) / _grid_buckets(nb + 1)
        b0 = lo + (np.floor((base[j] - lo) / bw) + 0.25) * bw
        cluster = b0 + (bw / (2.0 * per)) * np.arange(per)
        base = np.sort(np.concatenate([np.delete(base, j), cluster]))
        base = base[np.concatenate([[True], np.diff(base) > 0])]
        while len(base) < nb + 1:  # (a neighbour fell into the cluster's bucket and was merged: put an edge back further out)
            gaps = np.diff(base)
            g = int(np.argmax(gaps))
            base = np.insert(base, g + 1, 0.5 * (base[g] + base[g + 1]))
        base = base[: nb + 1] if len(base) > nb + 1 else base
        base[0], base[-1] = lo, hi
    assert len(base) == nb + 1 and np.all(np.diff(base) > 0)
    return base


def samples_for(edges_list, n_rows, n_cols, dtype, seed):
    """[n_rows, n_cols] per input: normal samples spread over the range, then samples ON edges (the right edge included),
    on their float neighbours, outside the range, NaN and +-inf"""
    rng = np.random.default_rng(seed)
    out = []
    for d, e in enumerate(edges_list):
        lo, hi = e[0], e[-1]
        x = rng.uniform(lo - 0.05 * (hi - lo), hi + 0.05 * (hi - lo), (n_rows, n_cols))
        flat = x.reshape(-1)
        k = flat.size
        idx = rng.permutation(k)
        on = idx[: k // 16]
        flat[on] = e[rng.integers(0, len(e), on.size)]
        nxt = idx[k // 16: k // 12]
        flat[nxt] = np.nextafter(e[rng.integers(0, len(e), nxt.size)], np.inf if d % 2 else -np.inf)
        flat[idx[k // 12: k // 12 + max(1, k // 200)]] = np.nan
        flat[idx[-3:]] = [np.inf, -np.inf, hi]
        out.append(x.astype(dtype))
    return out


# ---------------------------------------------------------------------------------------------------------------------
# the cases
# ---------------------------------------------------------------------------------------------------------------------
def _split_bins(total, D):
    """D bin counts whose product is about `total` (distinct, so a transposed index would show)"""
    if D == 1:
        return [total]
    if D == 2:
        a = max(2, int(round(total ** 0.5 * 1.25)))
        return [a, max(2, total // a)]
    a = max(2, int(round(total ** (1 / 3) * 1.3)))
    b = max(2, int(round(total ** (1 / 3))))
    return [a, b, max(2, total // (a * b))]


KINDS = ("lin", "k1", "k2", "k3", "k4", "crowd", "geom")
_MIN_NB = {"lin": 2, "k1": 3, "k2": 4, "k3": 5, "k4": 6, "crowd": 9, "geom": 4}

# home -> total bins per number of inputs for a (float64-weighted, counting) histogram, [rows, cols], plan parameters
# (sizes: the smallest that reach the home; partitioned homes need n_cols >= 4 with "partition" = 1)
_BIG_W, _BIG_U = {1: 40_000, 2: 40_000, 3: 40_000}, {1: 90_000, 2: 90_000, 3: 90_000}
_LDS = {1: 96, 2: 400, 3: 2_000}
_LANE = {1: 40, 2: 110, 3: 220}
HOMES = {
    "lds": dict(bins=(_LDS, _LDS), shape=(3, 20_011), params={}),
    "lds_many_copies": dict(bins=({1: 24, 2: 30, 3: 60}, {1: 24, 2: 30, 3: 60}), shape=(1, 50_003), params={}),
    "packed16": dict(bins=(None, {1: 50_000, 2: 50_000, 3: 50_000}), shape=(2, 30_011), params={}),  # unweighted only: uint16 counters two per word
    "global": dict(bins=(_LDS, _LDS), shape=(2, 10_007), params={"force_global": 1}),
    "global_big": dict(bins=({1: 60_000, 2: 60_000, 3: 60_000}, {1: 120_000, 2: 120_000, 3: 120_000}), shape=(1, 20_011), params={"partition": -1, "slices": -1}),
    "sliced": dict(bins=(_BIG_W, _BIG_U), shape=(2, 20_011), params={"slices": 1}),
    "route": dict(bins=(_BIG_W, _BIG_U), shape=(1, 40_013), params={"partition": 1}),
    "route_rows": dict(bins=(_BIG_W, _BIG_U), shape=(3, 20_011), params={"partition": 1}),
    "route_rows_spl4": dict(bins=(_BIG_W, _BIG_U), shape=(3, 20_011), params={"partition": 1, "route_spl": 4}),
    "route_spl4": dict(bins=(_BIG_W, _BIG_U), shape=(1, 40_013), params={"partition": 1, "route_spl": 4}),
    "route_exact": dict(bins=(_BIG_W, None), shape=(1, 40_013), params={"partition": 1, "records48": -1}),  # float64 weights as full records
    "three_pass": dict(bins=(_BIG_W, _BIG_U), shape=(1, 40_013), params={"partition": 1, "fused": -1}),
    "lanes": dict(bins=(_LANE, _LANE), shape=(3_001, 37), params={"lanes": 1}),
    "lanes_wide": dict(bins=({1: 600, 2: 600, 3: 700}, {1: 600, 2: 600, 3: 700}), shape=(2_003, 53), params={"lanes": 1}),
    "lanes_long": dict(bins=(None, _LANE), shape=(40, 66_001), params={"lanes": 1}),  # rows beyond 65535 samples: uint32 counter columns
    "flat_rows": dict(bins=({1: 50, 2: 90, 3: 90}, {1: 50, 2: 90, 3: 90}), shape=(5_003, 181), params={"flat_rows": 1}),
    # reductions over a LEADING axis: the rows are the contiguous direction (row stride 1) and the row-per-lane kernels take the
    # view as it lies; beyond 65535 samples per row the counts use uint32 columns
    "lanes_lead": dict(bins=(_LANE, _LANE), shape=(301, 1_009), params={}, lead=True),
    "lanes_lead_long": dict(bins=(None, _LANE), shape=(48, 66_001), params={}, lead=True),
    # BASELINE C4's shape class (>= 64 rows of >= 2^19 float32 samples, counts): tiles twice as long
    "long_rows_f32": dict(bins=(None, {1: 50}), shape=(64, 1 << 19), params={}, only_st=F32, only_kinds=("lin", "k1", "k2")),
}
# one input: where the edges of a histogram near the LDS capacity still fit it decides between LDS copies, packed uint16
# counters, bin slices and the partitioned passes — a ladder of bin counts around those borders
for _nb in (12_000, 14_000, 16_000, 17_000, 20_000, 24_000, 30_000, 36_000):
    for _tag, _pp in (("", {}), ("_part", {"partition": 1}), ("_3pass", {"partition": 1, "fused": -1}), ("_sliced", {"slices": 1})):
        HOMES["d1_%d%s" % (_nb, _tag)] = dict(bins=({1: _nb}, {1: _nb}), shape=(1, 30_011), params=_pp)
FORM_PARAMS = {  # digitize-form overrides tried for every (kind, home): {} = the pickers' own choice
    "lin": ({}, {"arith": 1}, {"arith": -1}, {"arith32": 1}, {"arith": -1, "pack": 1}),
    "k1": ({}, {"pack": 1}),
    "k2": ({}, {"pack": -1}, {"pack": 1}),
    "k3": ({}, {"pack": -1}, {"pack": 1}),
    "k4": ({}, {"pack": -1}, {"pack": 1}),
    "crowd": ({}, {"pack": -1}, {"pack": 1}),
    "geom": ({}, {"pack": -1}, {"pack": 1}),
}


def _cases(st, wt, D):
    weighted = wt is not None
    for home, h in HOMES.items():
        per_d = h["bins"][0 if weighted else 1]
        if per_d is None or D not in per_d:
            continue
        for kind in KINDS:
            nbs = _split_bins(per_d[D], D)
            if min(nbs) < _MIN_NB[kind]:
                continue
            if home.startswith("d1_") and kind == "geom":
                continue
            if h.get("only_st") is not None and st is not h["only_st"]:
                continue
            if h.get("only_kinds") is not None and kind not in h["only_kinds"]:
                continue
            for form in FORM_PARAMS[kind]:
                if "arith32" in form and st is not F32:
                    continue
                yield dict(home=home, kind=kind, nbs=nbs, shape=h["shape"], params=dict(h["params"], **form), lead=bool(h.get("lead")))


_KNOBS = ("block_threads", "grid_blocks", "force_global", "force_generic", "fused", "records48", "exchange", "route_spl", "flat_rows",
          "min_parts", "route_pool_pct", "partition", "lanes", "slices", "route_grid", "acc_grid", "pack", "arith", "arith32", "lds_copies",
          "exchange_budget_ms", "exchange_arrive_us", "exchange_min_pct")


def _fresh(plan):
    """every tuning key back to its default: plans are cached by their edges, and np.linspace / geometric edges are shared with
    other tests of the session — what a case selects must not depend on what ran before it"""
    for k in _KNOBS:
        plan.set_param(k, 0)
    return plan


def _dev_lead(a):
    """[rows, cols] on the GPU with the ROWS as the contiguous direction (strides (1, rows)): what a reduction over a leading
    axis of a C-ordered array hands to the hot path"""
    import torch

    return torch.as_tensor(np.ascontiguousarray(a.T)).cuda().T


def _run_case(core, st, wt, D, case, seed):
    import torch

    edges = [edges_of(case["kind"], nb, seed=seed + d) for d, nb in enumerate(case["nbs"])]
    n_rows, n_cols = case["shape"]
    xs = samples_for(edges, n_rows, n_cols, st, seed)
    rng = np.random.default_rng(seed + 99)
    w = None if wt is None else rng.uniform(0.25, 2.0, (n_rows, n_cols)).astype(wt)
    # the oracle compares in float64 against float64 edges (numpy's promotion); float32 weights are added as float64
    want = onp.bincount_rows([x.astype(F64) for x in xs], edges, None if w is None else w.astype(F64))
    put = _dev_lead if case.get("lead") else _dev
    xd = [put(x) for x in xs]
    wd = None if w is None else put(w)
    dts = [core._np_dtype_of(s) for s in xd]
    cmp_domain, conv, _ = core._compare_domain(dts, edges)
    plan = _fresh(core._get_plan(conv, cmp_domain, 0))
    for k, v in case["params"].items():
        plan.set_param(k, v)
    try:
        got = core._bincount_2d_vectorized(*xd, bins=edges, weights=wd)
        torch.cuda.synchronize()
        desc = plan.describe()
    finally:
        for k in case["params"]:
            plan.set_param(k, 0)
    assert_hist_equal(got.cpu().numpy(), want, weighted=wt is not None), (case, desc)
    return desc


_ST = {"f64": F64, "f32": F32}
_WT = {"none": None, "f32": F32, "f64": F64}


def _key(st, wt, D, case):
    return "%s|%s|%d|%s|%s|%s" % (st, wt, D, case["home"], case["kind"], ",".join("%s=%d" % kv for kv in sorted(case["params"].items())))


# Which of the ~3900 cases of the product are RUN: the smallest set that still selects every kernel the whole product selects
# (greedy set cover over a discovery run: XHIST_CENSUS_DISCOVER=<file> XHIST_AMD_KERNEL_LOG_ALL=1 runs them all and writes which
# case picked which kernels; tools/census_cover.py reduces that to tests/golden/census_cases.json).  Without the file: all of them.
_COVER = os.path.join(ROOT, "tests", "golden", "census_cases.json")


def _selected_keys():
    if os.environ.get("XHIST_CENSUS_DISCOVER") or not os.path.exists(_COVER):
        return None
    import json

    return set(json.load(open(_COVER))["cases"])


def _log_lines():
    path = os.environ.get("XHIST_AMD_KERNEL_LOG")
    if not path or not os.path.exists(path):
        return []
    with open(path) as f:
        return f.read().splitlines()


@pytest.mark.parametrize("D", [1, 2, 3])
@pytest.mark.parametrize("wt", ["none", "f32", "f64"])
@pytest.mark.parametrize("st", ["f64", "f32"])
def test_dispatch_surface_float_samples(xh, st, wt, D):
    """float64 / float32 samples x {counts, float32 weights, float64 weights} x 1-3 inputs: every digitize form in every
    histogram home, each result against the oracle"""
    keep = _selected_keys()
    discover = os.environ.get("XHIST_CENSUS_DISCOVER")
    n = 0
    for i, case in enumerate(_cases(_ST[st], _WT[wt], D)):
        key = _key(st, wt, D, case)
        if keep is not None and key not in keep:
            continue
        before = len(_log_lines()) if discover else 0
        _run_case(xh, _ST[st], _WT[wt], D, case, seed=1000 * D + 17 * i)
        if discover:
            import json

            with open(discover, "a") as f:
                f.write(json.dumps({"case": key, "kernels": sorted(set(_log_lines()[before:]))}) + "\n")
        n += 1
    assert n > 0


# ---------------------------------------------------------------------------------------------------------------------
# the other kernel families: small integer / half samples, the exact int64 domain, dtype mixtures, two weights, generic
# ---------------------------------------------------------------------------------------------------------------------
def _run_general(core, samples, edges, w, params, weighted, lead=False):
    """samples: list of [rows, cols] numpy arrays (any dtype); compares with the oracle in float64 / exact int64"""
    import torch

    put = _dev_lead if lead else _dev
    xd = [put(x) for x in samples]
    wd = None if w is None else put(w)
    dts = [core._np_dtype_of(t) for t in xd]
    cmp_domain, conv, _ = core._compare_domain(dts, edges)
    plan = _fresh(core._get_plan(conv, cmp_domain, 0))
    for k, v in params.items():
        plan.set_param(k, v)
    try:
        got = core._bincount_2d_vectorized(*xd, bins=edges, weights=wd)
        torch.cuda.synchronize()
    finally:
        for k in params:
            plan.set_param(k, 0)
    want = onp.bincount_rows(samples, edges, w)
    assert_hist_equal(got.cpu().numpy(), want, weighted)


@pytest.mark.parametrize("dtype", ["int32", "int64", "int16", "uint8", "float16"])
def test_dispatch_surface_small_samples(xh, dtype):
    """one input of a small integer / half dtype (compared in float64 like numpy does): counts or float64 weights, histogram in
    LDS or behind memory-side atomics, one edge per bucket or a binary search; rows streamed or one row per lane"""
    rng = np.random.default_rng(7)
    dt = np.dtype(dtype)
    for kind in ("k1", "crowd", "lin"):
        for home, shape, params in (("lds", (3, 20_011), {}), ("global", (2, 10_007), {"force_global": 1}), ("lanes_lead", (301, 1_009), {}),
                                    ("lanes_lead_long", (48, 66_001), {})):
            for weighted in (False, True):
                nb = 60
                e = edges_of(kind, nb, lo=0.0 if dt.kind == "u" else -100.0, hi=200.0 if dt.kind == "u" else 100.0, seed=3)
                if dt.kind in "iu":
                    x = rng.integers(-20 if dt.kind == "u" else -120, 220 if dt.kind == "u" else 120, shape).clip(np.iinfo(dt).min, np.iinfo(dt).max).astype(dt)
                else:
                    x = rng.uniform(-110, 110, shape).astype(dt)
                    x.reshape(-1)[::97] = np.nan
                w = rng.uniform(0.5, 1.5, shape) if weighted else None
                _run_general(xh, [x], [e], w, params, weighted, lead=home.startswith("lanes_lead"))


def test_dispatch_surface_int64_domain(xh):
    """int64 samples against INTEGER edges: compared exactly in int64 (values beyond 2^53 included)"""
    rng = np.random.default_rng(8)
    for home, shape, params in (("lds", (3, 20_011), {}), ("global", (2, 10_007), {"force_global": 1})):
        for weighted in (False, True):
            base = (1 << 60)
            e = base + np.sort(rng.choice(100_000, 65, replace=False)).astype(np.int64)
            x = base + rng.integers(-1000, 101_000, shape).astype(np.int64)
            w = rng.uniform(0.5, 1.5, shape) if weighted else None
            _run_general(xh, [x], [e], w, params, weighted)


def test_dispatch_surface_generic_integer_domains(xh):
    """the generic family's exact-integer variants: TWO int64 inputs against integer edges (compared in int64), an int64 input
    next to a float64 one (a compare domain per input), edge arrays too long for LDS; histogram in LDS or behind memory-side atomics"""
    rng = np.random.default_rng(10)
    base = 1 << 58
    shape = (3, 20_011)
    ei = base + np.sort(rng.choice(50_000, 41, replace=False)).astype(np.int64)
    ej = base + np.sort(rng.choice(50_000, 33, replace=False)).astype(np.int64)
    ef = edges_of("k2", 24, lo=-3.0, hi=3.0, seed=1)
    xi = base + rng.integers(-500, 50_500, shape).astype(np.int64)
    xj = base + rng.integers(-500, 50_500, shape).astype(np.int64)
    xf = rng.uniform(-3.3, 3.3, shape)
    big = base + np.sort(rng.choice(4_000_000, 30_001, replace=False)).astype(np.int64)  # 30001 int64 edges: 240 KB, beyond LDS
    xb = base + rng.integers(-5_000, 4_005_000, shape).astype(np.int64)
    bigf = np.sort(rng.uniform(-3, 3, 30_001))
    for weighted in (False, True):
        w = rng.uniform(0.5, 1.5, shape) if weighted else None
        for params in ({}, {"force_global": 1}):
            _run_general(xh, [xi, xj], [ei, ej], w, params, weighted)        # int64 domain, two inputs
            _run_general(xh, [xi, xf], [ei, ef], w, params, weighted)        # per-input domains
            _run_general(xh, [xf, xj], [ef, ej], w, params, weighted)
        _run_general(xh, [xb], [big], w, {}, weighted)                       # tables that do not fit LDS: int64 domain
        _run_general(xh, [xb, xf], [big, ef], w, {}, weighted)               # ... per-input domains
        _run_general(xh, [xf * 1.0], [bigf], w, {"force_generic": 1}, weighted)  # ... float64 domain


@pytest.mark.parametrize("D", [1, 2, 3])
def test_dispatch_surface_mixed_dtypes(xh, D):
    """inputs of different dtypes / integer weights (consumed as float64 by the MIXED vector kernels; the generic family where
    those have no variant): every digitize form they take, weighted and not"""
    rng = np.random.default_rng(9 + D)
    dts = [np.float32, np.int32, np.float64][:D] if D > 1 else [np.float32]
    for kind in ("lin", "k1", "k2", "k3", "crowd"):
        for params in ({}, {"arith": -1}, {"arith": 1}, {"force_generic": 1}, {"force_global": 1}, {"force_generic": 1, "force_global": 1}):
            for wkind in (None, np.int32, np.float64):
                nbs = _split_bins({1: 96, 2: 400, 3: 2_000}[D], D)
                if min(nbs) < _MIN_NB[kind]:
                    continue
                edges = [edges_of(kind, nb, lo=-50.0, hi=50.0, seed=d) for d, nb in enumerate(nbs)]
                shape = (3, 20_011)
                xs = []
                for d in range(D):
                    v = rng.uniform(-55, 55, shape)
                    xs.append(np.rint(v).astype(dts[d]) if np.dtype(dts[d]).kind == "i" else v.astype(dts[d]))
                w = None if wkind is None else (rng.integers(1, 5, shape).astype(wkind) if np.dtype(wkind).kind == "i" else rng.uniform(0.5, 2, shape))
                if D == 1 and wkind is None:
                    continue  # (homogeneous: the float product above)
                _run_general(xh, xs, edges, w, params, w is not None)


def test_dispatch_surface_odds_and_ends(xh):
    """what the products above do not reach by construction: ONE input of a dtype that has no vector kernel of its own (uint16:
    the MIXED kernels take it, unweighted, as float64) on every digitize form, and the three-pass partitioned mode with more than
    128 partitions and float32 weights (records in groups of four)"""
    rng = np.random.default_rng(12)
    shape = (3, 20_011)
    x16 = rng.integers(0, 1200, shape).astype(np.uint16)
    x8 = rng.integers(-128, 128, shape).astype(np.int8)
    xb = rng.integers(0, 2, shape).astype(np.bool_)
    for kind in ("lin", "k1", "k2", "crowd"):
        for x, lo, hi in ((x16, 100.0, 1100.0), (x8, -100.0, 100.0), (xb, -0.5, 1.5)):
            e = edges_of(kind, 96 if x is not xb else 12, lo=lo, hi=hi, seed=2)
            for params in ({}, {"arith": 1}, {"arith": -1}):
                _run_general(xh, [x], [e], None, params, False)
            # the same from HOST memory (numpy in, numpy out: the dtype reaches the library as it is — torch has no uint16 /
            # uint32 arithmetic to hand it over resident), plus uint32
            for xin in ((x, x.astype(np.uint32)) if x is x16 else (x,)):
                got = xh._bincount_2d_vectorized(xin, bins=[e], weights=None)
                assert_hist_equal(got, onp.bincount_rows([xin], [e], None), False)
    nb = 1_600  # 1600 x 1600 float64 sums = 157 partitions of 2^14 bins
    edges = [np.linspace(-4, 4, nb + 1), edges_of("k1", nb, seed=5)]
    xs = samples_for(edges, 1, 40_013, F32, 3)
    w = rng.uniform(0.5, 1.5, (1, 40_013)).astype(F32)
    _run_general(xh, [x.astype(F64) for x in xs], edges, w.astype(F64), {"partition": 1}, True)          # float64: groups of four as well
    xd32 = [x for x in xs]
    import torch

    dts = [np.dtype(F32)] * 2
    cmp_domain, conv, _ = xh._compare_domain(dts, edges)
    plan = xh._get_plan(conv, cmp_domain, 0)
    plan.set_param("partition", 1)
    try:
        got = xh._bincount_2d_vectorized(*[_dev(x) for x in xd32], bins=edges, weights=_dev(w))
        torch.cuda.synchronize()
    finally:
        plan.set_param("partition", 0)
    assert_hist_equal(got.cpu().numpy(), onp.bincount_rows([x.astype(F64) for x in xs], edges, w.astype(F64)), True)
    # counts with more than 128 partitions (2100 x 2100 bins = 135 partitions of 2^15): three passes, records in groups of four
    nb = 2_100
    edges = [np.linspace(-4, 4, nb + 1), edges_of("k1", nb, seed=6)]
    xs = samples_for(edges, 1, 40_013, F64, 4)
    _run_general(xh, xs, edges, None, {"partition": 1}, False)
    # ONE column taken out of a wider array (a [rows, 1] view with a column stride): no vector kernel of the homogeneous family
    # takes a strided column, the MIXED one does (n_cols == 1) — unweighted, on a binary search, one and two edges per bucket
    base = torch.as_tensor(rng.uniform(-4.4, 4.4, (6_007, 3))).cuda()
    col = base[:, 1:2]
    assert col.stride() == (3, 1) and col.shape == (6_007, 1)
    for kind in ("k1", "k2", "crowd", "lin"):
        e = edges_of(kind, 96, seed=9)
        got = xh._bincount_2d_vectorized(col, bins=[e], weights=None)
        torch.cuda.synchronize()
        assert_hist_equal(got.cpu().numpy(), onp.bincount_rows([col.cpu().numpy()], [e], None), False)


@pytest.mark.parametrize("weighted", [False, True])
def test_padded_layout_at_the_lds_capacity_border(xh, weighted):
    """one float32 input on np.linspace edges takes the float32 arithmetic digitize with a PADDED LDS layout (32 slots in
    front, 32 behind).  Within 64 bins of the LDS capacity the padded histogram does not fit while the unpadded one would: the
    picker must not hand the padded kernel an unpadded allocation there (it now keeps the digitize it had).  Bin counts on both
    sides of both borders, against the oracle."""
    rng = np.random.default_rng(13)
    cap = (160 * 1024) // (8 if weighted else 4)
    x = rng.uniform(-4.2, 4.2, (1, 300_007)).astype(F32)
    x[0, ::211] = np.nan
    w = rng.uniform(0.5, 1.5, x.shape).astype(F32) if weighted else None
    for nb in (cap - 80, cap - 64, cap - 50, cap - 33, cap - 32, cap - 20, cap - 1):
        e = np.linspace(-4.0, 4.0, nb + 1)
        for params in ({}, {"arith32": 1}):
            _run_general(xh, [x], [e], w, params, weighted)


def test_dispatch_surface_one_long_float64_row(xh):
    """one float64 row of >= 2^29 samples without weights takes tiles twice as long (the headline's 8 B/sample variant; one or two
    edges per bucket).  Too long for the oracle in a test: the whole row must equal the sum of its two halves — which are short
    enough for the ordinary kernels the rest of this module holds to the oracle — and the total the in-range count torch finds."""
    import torch

    n = (1 << 29) + 4_099
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    x = torch.empty((1, n), dtype=torch.float64, device="cuda").normal_(generator=g)
    x[0, ::1_000_003] = float("nan")
    for kind in ("k1", "k2"):
        e = edges_of(kind, 100, seed=3)
        x[0, 5::2_000_003] = float(e[17])  # on an edge
        x[0, 7::3_000_017] = float(e[-1])  # the right edge counts
        whole = xh._bincount_2d_vectorized(x, bins=[e], weights=None)
        h = n // 2
        a = xh._bincount_2d_vectorized(x[:, :h], bins=[e], weights=None)
        b = xh._bincount_2d_vectorized(x[:, h:], bins=[e], weights=None)
        torch.cuda.synchronize()
        assert torch.equal(whole, a + b)
        inside = int(((x >= float(e[0])) & (x <= float(e[-1]))).sum().item())
        assert int(whole.sum().item()) == inside


@pytest.mark.parametrize("D", [1, 2, 3])
@pytest.mark.parametrize("wt", ["f32", "f64"])
@pytest.mark.parametrize("st", ["f64", "f32"])
def test_dispatch_surface_two_weights(xh, st, wt, D):
    """two weight arrays in one pass (histogram_two_weights: the TODO at /root/reference/xhistogram/xarray.py:106): each of the
    pair equals the oracle's single-weight histogram, for one and two edges per bucket"""
    import torch

    rng = np.random.default_rng(11 * D)
    for kind in ("lin", "k1", "k2"):
        nbs = _split_bins({1: 96, 2: 400, 3: 900}[D], D)
        edges = [edges_of(kind, nb, seed=d) for d, nb in enumerate(nbs)]
        shape = (2, 20_011)
        xs = samples_for(edges, shape[0], shape[1], _ST[st], 5)
        wa, wb = (rng.uniform(0.5, 2.0, shape).astype(_WT[wt]) for _ in range(2))
        for params in ({}, {"arith": -1}):
            xd = [_dev(x) for x in xs]
            dts = [xh._np_dtype_of(t) for t in xd]
            cmp_domain, conv, _ = xh._compare_domain(dts, edges)
            plan = xh._get_plan(conv, cmp_domain, 0)
            for k, v in params.items():
                plan.set_param(k, v)
            try:
                ha, hb, _ = xh.histogram_two_weights(*xd, bins=edges, axis=1, weights=(_dev(wa), _dev(wb)))
                torch.cuda.synchronize()
            finally:
                for k in params:
                    plan.set_param(k, 0)
            x64 = [x.astype(F64) for x in xs]
            assert_hist_equal(ha.cpu().numpy(), onp.bincount_rows(x64, edges, wa.astype(F64)), True)
            assert_hist_equal(hb.cpu().numpy(), onp.bincount_rows(x64, edges, wb.astype(F64)), True)
