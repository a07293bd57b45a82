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
picks; `test_zz_every_dispatchable_kernel_was_compared` (last test of the last-sorted GPU module) then holds the log against
the host stubs of the shared object: an instantiation that nothing selected fails the run — it has to be given a case here or
leave the library.
"""
import itertools
import os
import subprocess
import sys

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
    if kind == "geom":
        e = np.geomspace(1e-3, hi - lo + 1e-3, nb + 1) - 1e-3 + lo
        e[0], e[-1] = lo, hi
        return e
    per = {"k1": 1, "k2": 2, "k3": 3, "k4": 4, "crowd": 7}[kind]
    extra = per - 1
    base_n = nb + 1 - extra
    assert base_n >= 3, (kind, nb)
    base = np.linspace(lo, hi, base_n)
    # not arithmetic: every interior edge moved by a small fraction of the bin width (still one per bucket)
    width = (hi - lo) / (base_n - 1)
    base[1:-1] += rng.uniform(-0.2, 0.2, base_n - 2) * min(width, (hi - lo) / _grid_buckets(nb + 1))
    if extra:
        j = base_n // 2
        delta = (hi - lo) / _grid_buckets(nb + 1) / 16.0
        cluster = base[j] + delta * np.arange(1, extra + 1)
        base = np.sort(np.concatenate([base, cluster]))
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

# home -> (total bins for a float64-weighted / counting histogram, [rows, cols], plan parameters)
# (sizes: the smallest that reach the home; partitioned homes need n_cols >= 4 with "partition" = 1)
HOMES = {
    "lds": dict(bins=(96, 96), shape=(3, 20_011), params={}),
    "lds_many_copies": dict(bins=(24, 24), shape=(1, 50_003), params={}),
    "packed16": dict(bins=(None, 50_000), shape=(2, 30_011), params={}),          # unweighted only: uint16 counters two per word
    "global": dict(bins=(96, 96), shape=(2, 10_007), params={"force_global": 1}),
    "global_big": dict(bins=(60_000, 120_000), shape=(1, 20_011), params={"partition": -1, "slices": -1}),
    "sliced": dict(bins=(40_000, 90_000), shape=(2, 20_011), params={"slices": 1}),
    "route": dict(bins=(40_000, 90_000), shape=(1, 40_013), params={"partition": 1}),
    "route_rows": dict(bins=(40_000, 90_000), shape=(3, 20_011), params={"partition": 1}),
    "route_spl4": dict(bins=(40_000, 90_000), shape=(1, 40_013), params={"partition": 1, "route_spl": 4}),
    "route_exact": dict(bins=(40_000, None), shape=(1, 40_013), params={"partition": 1, "records48": -1}),  # float64 weights as full records
    "three_pass": dict(bins=(40_000, 90_000), shape=(1, 40_013), params={"partition": 1, "fused": -1}),
    "lanes": dict(bins=(40, 40), shape=(3_001, 37), params={"lanes": 1}),
    "lanes_wide": dict(bins=(600, 600), shape=(2_003, 53), params={"lanes": 1}),   # uint16 counter columns where unweighted
    "flat_rows": dict(bins=(50, 50), shape=(5_003, 181), params={"flat_rows": 1}),
}
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
        total = h["bins"][0 if weighted else 1]
        if total is None:
            continue
        for kind in KINDS:
            nbs = _split_bins(total, D)
            if kind != "lin" and min(nbs) < 9:
                continue
            for form in FORM_PARAMS[kind]:
                if "arith32" in form and st is not F32:
                    continue
                yield dict(home=home, kind=kind, nbs=nbs, shape=h["shape"], params=dict(h["params"], **form))


def _run_case(core, st, wt, D, case, seed):
    import torch

    edges = [edges_of(case["kind"], nb, seed=seed + d) for d, nb in enumerate(case["nbs"])]
    n_rows, n_cols = case["shape"]
    xs = samples_for(edges, n_rows, n_cols, st, seed)
    rng = np.random.default_rng(seed + 99)
    w = None if wt is None else rng.uniform(0.25, 2.0, (n_rows, n_cols)).astype(wt)
    # the oracle compares in float64 against float64 edges (numpy's promotion); float32 weights are added as float64
    want = onp.bincount_rows([x.astype(F64) for x in xs], edges, None if w is None else w.astype(F64))
    xd = [_dev(x) for x in xs]
    wd = None if w is None else _dev(w)
    dts = [core._np_dtype_of(s) for s in xd]
    cmp_domain, conv, _ = core._compare_domain(dts, edges)
    plan = core._get_plan(conv, cmp_domain, 0)
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
