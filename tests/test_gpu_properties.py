"""Property-based GPU parity (the reference's test_chunking_hypotheses.py idea, widened): random
shapes, axes, dtypes, edge layouts and kernel-family overrides against the oracle; plus
thread-safety of one plan under concurrent callers (dask's threaded scheduler does this)."""
import threading

import numpy as np
import pytest

from conftest import assert_hist_equal
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
hyp = pytest.importorskip("hypothesis")
from hypothesis import given, settings, HealthCheck  # noqa: E402
import hypothesis.strategies as st  # noqa: E402


@pytest.fixture(scope="module")
def xh():
    from xhistogram_amd import _native, core

    assert _native.device_count() >= 1
    return core


def _edges(draw, kind, n):
    if kind == "uniform":
        lo = draw(st.floats(-5, 0))
        return np.linspace(lo, lo + draw(st.floats(0.5, 9)), n + 1)
    if kind == "random":
        e = np.sort(np.array(draw(st.lists(st.floats(-4, 4, allow_nan=False, width=32), min_size=n + 1, max_size=n + 1))))
        return e
    if kind == "duplicates":
        e = np.sort(np.round(np.array(draw(st.lists(st.floats(-3, 3), min_size=n + 1, max_size=n + 1))), 0))
        return e
    # geometric: tiny and huge bins together -> crowded buckets -> binary-search family
    return np.concatenate([[-4.0], -4.0 + np.cumsum(np.geomspace(1e-9, 4.0, n))])


@st.composite
def cases(draw):
    d = draw(st.integers(1, 3))
    ndim = draw(st.integers(1, 3))
    shape = tuple(draw(st.integers(1, 9)) for _ in range(ndim - 1)) + (draw(st.integers(1, 300)),)
    axis_choices = [None] + [tuple(c) for r in range(1, ndim + 1) for c in __import__("itertools").combinations(range(ndim), r)]
    axis = draw(st.sampled_from(axis_choices))
    dtype = draw(st.sampled_from([np.float64, np.float32, np.int32, np.int64]))
    kinds = [draw(st.sampled_from(["uniform", "random", "duplicates", "geometric"])) for _ in range(d)]
    edges = [_edges(draw, k, draw(st.integers(1, 12))) for k in kinds]
    weighted = draw(st.booleans())
    wshape = draw(st.sampled_from(["full", "row", "scalar"])) if weighted else None
    density = draw(st.booleans())
    seed = draw(st.integers(0, 2**31 - 1))
    resident = draw(st.booleans())
    override = draw(st.sampled_from([None, "force_global", "force_generic", "lds_copies", "arith", "arith"]))
    return d, shape, axis, dtype, edges, wshape, density, seed, resident, override


@settings(max_examples=300, deadline=None, suppress_health_check=list(HealthCheck))
@given(cases())
def test_random_cases_match_oracle(xh, case):
    d, shape, axis, dtype, edges, wshape, density, seed, resident, override = case
    rng = np.random.default_rng(seed)
    if np.dtype(dtype).kind == "f":
        args = [(rng.standard_normal(shape) * 2).astype(dtype) for _ in range(d)]
        for a in args:
            flat = a.reshape(-1)
            flat[rng.integers(0, flat.size, max(1, flat.size // 17))] = rng.choice([np.nan, np.inf, -np.inf, 4.0, -4.0, 0.0])
    else:
        args = [rng.integers(-5, 6, shape).astype(dtype) for _ in range(d)]
    w = None
    if wshape == "full":
        w = rng.uniform(0, 2, shape)
    elif wshape == "row":
        w = rng.uniform(0, 2, shape[-1:])
    elif wshape == "scalar":
        w = np.float64(0.75) * np.ones((1,) * len(shape))
    kw = dict(bins=edges if d > 1 else edges[0], axis=axis, density=density)
    try:
        want, _ = onp.histogram(*args, weights=w, **kw)
    except NotImplementedError:
        return
    if resident:
        targs = [torch.as_tensor(a).cuda() for a in args]
        tw = None if w is None else torch.as_tensor(np.asarray(w)).cuda()
    else:
        targs, tw = args, w
    plan = None
    if override:
        dts = [xh._np_dtype_of(a) for a in targs]
        try:
            dom, conv, _ = xh._compare_domain(dts, edges)
        except NotImplementedError:
            return
        plan = xh._get_plan(conv, dom, 0)
        plan.set_param(override, 1)
    try:
        got, _ = xh.histogram(*targs, weights=tw, **kw)
    except NotImplementedError:
        return
    finally:
        if plan is not None:
            plan.set_param(override, 0)
    got = got.cpu().numpy() if resident else got
    assert_hist_equal(got, want, weighted=(w is not None) or density)


def test_one_plan_many_threads(xh):
    """ctypes releases the GIL: concurrent executes on one cached plan must not interfere"""
    rng = np.random.default_rng(77)
    edges = np.linspace(-4, 4, 41)
    data = [rng.standard_normal((3, 50_000 + 1000 * i)) for i in range(8)]
    want = [onp.histogram(a, bins=edges, axis=1)[0] for a in data]
    results = [None] * len(data)
    errors = []

    def work(i):
        try:
            for _ in range(5):
                s = torch.cuda.Stream()
                with torch.cuda.stream(s):
                    h, _ = xh.histogram(torch.as_tensor(data[i]).cuda(), bins=edges, axis=1)
                    hn, _ = xh.histogram(data[i], bins=edges, axis=1, weights=np.ones_like(data[i]))
                s.synchronize()
                results[i] = (h.cpu().numpy(), hn)
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(data))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for (h, hn), w in zip(results, want):
        np.testing.assert_array_equal(h, w)
        np.testing.assert_array_equal(hn, w.astype(np.float64))


@st.composite
def large_cases(draw):
    shape = draw(st.sampled_from([(5000, 37), (4100, 300), (7, 9000), (3, 40, 500), (300, 20, 7), (64, 64, 64), (2, 70001), (9000, 3, 5)]))
    ndim = len(shape)
    axis_choices = [None] + [tuple(c) for r in range(1, ndim + 1) for c in __import__("itertools").combinations(range(ndim), r)]
    axis = draw(st.sampled_from(axis_choices))
    dtype = draw(st.sampled_from([np.float64, np.float32, np.int32, np.uint8, np.float16, np.int64]))
    d = draw(st.integers(1, 2))
    kinds = [draw(st.sampled_from(["uniform", "random", "geometric"])) for _ in range(d)]
    edges = [_edges(draw, k, draw(st.integers(2, 40))) for k in kinds]
    wkind = draw(st.sampled_from([None, "f64", "f32", "bcast"]))
    seed = draw(st.integers(0, 2**31 - 1))
    return shape, axis, dtype, edges, wkind, seed


@settings(max_examples=120, deadline=None, suppress_health_check=list(HealthCheck))
@given(large_cases())
def test_random_large_device_cases_match_oracle(xh, case):
    """bigger blocks: full vector tiles with ragged tails, the row-per-lane kernels (fused and
    transposing), grouped-row views of middle-axis reductions, integer / half samples"""
    shape, axis, dtype, edges, wkind, seed = case
    rng = np.random.default_rng(seed)
    d = len(edges)
    if np.dtype(dtype).kind == "f":
        args = [(rng.standard_normal(shape) * 2).astype(dtype) for _ in range(d)]
        args[0].reshape(-1)[:: max(1, args[0].size // 13)] = np.nan
    elif dtype == np.uint8:
        args = [rng.integers(0, 9, shape).astype(dtype) for _ in range(d)]
    else:
        args = [rng.integers(-5, 6, shape).astype(dtype) for _ in range(d)]
    w = None
    if wkind == "f64":
        w = rng.uniform(0, 2, shape)
    elif wkind == "f32":
        w = rng.uniform(0, 2, shape).astype(np.float32)
    elif wkind == "bcast":
        w = rng.uniform(0, 2, (1,) * (len(shape) - 1) + shape[-1:])
    kw = dict(bins=edges if d > 1 else edges[0], axis=axis)
    want, _ = onp.histogram(*args, weights=w, **kw)
    targs = [torch.as_tensor(a).cuda() for a in args]
    tw = None if w is None else torch.as_tensor(w).cuda()
    got, _ = xh.histogram(*targs, weights=tw, **kw)
    assert_hist_equal(got.cpu().numpy(), want, weighted=w is not None)


def test_more_than_2_31_elements(xh):
    """indices are 64-bit everywhere: 2.6e9 float32 samples (10.4 GB) as one row, as rows x columns,
    over the leading axis (row-per-lane kernels) and weighted, against sums of sub-range histograms"""
    dev = torch.device("cuda", 0)
    free, _ = torch.cuda.mem_get_info(dev)
    n = 2_600_000_000
    if free < 30 * 2**30:
        pytest.skip("needs 30 GB of free device memory")
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    x = torch.empty(n, dtype=torch.float32, device=dev).uniform_(-1.1, 1.1, generator=g)
    e = np.linspace(-1, 1, 101)
    inside = int(((x >= -1) & (x <= 1)).sum().item())
    h, _ = xh.histogram(x, bins=e)
    cuts = (0, 900_000_000, 2_200_000_003, n)
    parts = sum(xh.histogram(x[a:b], bins=e)[0] for a, b in zip(cuts, cuts[1:]))
    assert torch.equal(h, parts) and int(h.sum().item()) == inside
    x2 = x[: 5 * 500_000_000].view(5, 500_000_000)
    h2, _ = xh.histogram(x2, bins=e, axis=1)
    assert torch.equal(h2[4], xh.histogram(x2[4], bins=e)[0])
    x3 = x.view(2600, 1_000_000)
    e3 = np.linspace(-1, 1, 11)
    h3, _ = xh.histogram(x3, bins=e3, axis=0)
    assert int(h3.sum().item()) == inside
    assert torch.equal(h3[999_999], xh.histogram(x3[:, 999_999].contiguous(), bins=e3)[0])
    hb, eb = xh.histogram(x, bins=64)  # device min / max over the whole array
    assert int(hb.sum().item()) == n
    w = torch.empty(n, dtype=torch.float32, device=dev).uniform_(0, 1, generator=g)
    hw, _ = xh.histogram(x, bins=e, weights=w)
    hwp = sum(xh.histogram(x[a:b], bins=e, weights=w[a:b])[0] for a, b in zip(cuts, cuts[1:]))
    assert float((hw - hwp).abs().max() / hw.abs().max()) < 1e-12


@st.composite
def mixed_cases(draw):
    """joint histograms whose inputs have DIFFERENT dtypes, weights of any dtype, histograms from a few bins to beyond LDS
    (the mixed-dtype vector kernels, the generic family, packed / sliced / partitioned modes with their overrides)"""
    d = draw(st.integers(1, 3))
    rows = draw(st.integers(1, 4))
    cols = draw(st.sampled_from([1, 3, 4, 5, 63, 64, 257, 1000, 4096, 20_011]))
    dtypes = [draw(st.sampled_from([np.float64, np.float32, np.float16, np.int32, np.int16, np.uint8, np.int64])) for _ in range(d)]
    big = d == 2 and draw(st.booleans())
    nbs = [draw(st.integers(150, 420)) if big else draw(st.integers(1, 40)) for _ in range(d)]
    kinds = [draw(st.sampled_from(["uniform", "random"])) for _ in range(d)]
    wdtype = draw(st.sampled_from([None, None, np.float64, np.float32, np.float16, np.int32, np.bool_, np.uint8]))
    axis = draw(st.sampled_from([None, 1, 0]))
    seed = draw(st.integers(0, 2**31 - 1))
    resident = draw(st.booleans())
    override = draw(st.sampled_from([None, None, ("force_generic", 1), ("partition", 1), ("fused", -1), ("slices", 1), ("slices", -1), ("arith", 1), ("lanes", -1)]))
    return d, rows, cols, dtypes, nbs, kinds, wdtype, axis, seed, resident, override


@settings(max_examples=250, deadline=None, suppress_health_check=list(HealthCheck))
@given(mixed_cases())
def test_mixed_dtypes_and_big_histograms_match_oracle(xh, case):
    d, rows, cols, dtypes, nbs, kinds, wdtype, axis, seed, resident, override = case
    rng = np.random.default_rng(seed)
    args, edges = [], []
    for dt, nb, kind in zip(dtypes, nbs, kinds):
        if np.dtype(dt).kind == "f":
            a = (rng.standard_normal((rows, cols)) * 2).astype(dt)
            a.reshape(-1)[rng.integers(0, a.size, max(1, a.size // 23))] = rng.choice([np.nan, np.inf, -np.inf, 3.0, -3.0])
            lo, hi = -3.0, 3.0
        elif dt == np.uint8:
            a = rng.integers(0, 256, (rows, cols)).astype(dt)
            lo, hi = 10.0, 250.0
        else:
            a = rng.integers(-60, 61, (rows, cols)).astype(dt)
            lo, hi = -50.0, 50.0
        args.append(a)
        e = np.linspace(lo, hi, nb + 1) if kind == "uniform" else np.sort(np.concatenate([[lo, hi], rng.uniform(lo, hi, nb - 1)]))
        edges.append(e)
    w = None
    if wdtype is not None:
        w = rng.uniform(0, 3, (rows, cols))
        w = (w > 1.5) if wdtype == np.bool_ else (np.round(w * 3).astype(wdtype) if np.dtype(wdtype).kind in "iu" else w.astype(wdtype))
    kw = dict(bins=edges if d > 1 else edges[0], axis=axis)
    want, _ = onp.histogram(*args, weights=w, **kw)
    if resident:
        targs = [torch.as_tensor(a).cuda() for a in args]
        tw = None if w is None else torch.as_tensor(w).cuda()
    else:
        targs, tw = args, w
    plan = None
    if override:
        dom, conv, _ = xh._compare_domain([np.dtype(np.float64)] * d if False else [xh._np_dtype_of(a) for a in targs], edges)
        plan = xh._get_plan(conv, dom, 0)
        plan.set_param(*override)
    try:
        got, _ = xh.histogram(*targs, weights=tw, **kw)
    finally:
        if plan is not None:
            plan.set_param(override[0], 0)
    got = got.cpu().numpy() if resident else got
    assert_hist_equal(got, want, weighted=w is not None)


def test_host_route_calls_survive_allocation_churn_in_another_thread(xh):
    """12 threads of one-sample host-route calls while another thread allocates and frees device memory and creates /
    destroys plans: with staging from hipMallocAsync this put a sample in the wrong bin about once in 10^4 calls
    (profiles/r02_n_race_probe.txt); scratch now comes from the library's own allocator"""
    from xhistogram_amd import _native

    ea, eb = np.linspace(-4, 4, 9), np.linspace(-4, 4, 10)
    plan = _native.Plan([ea, eb], 0, 0)
    stop = threading.Event()
    bad, errors = [0], []

    def churn():
        k = 0
        while not stop.is_set():
            t = _native.DeviceBuffer(0, 1 << 20)
            t.close()
            if k % 8 == 0:
                _native.Plan([ea + 1e-6 * (k % 64), eb], 0, 0).close()
            k += 1

    def worker(seed):
        try:
            r = np.random.default_rng(seed)
            for _ in range(400):
                x, y = r.standard_normal(1), r.standard_normal(1)
                out = np.empty(plan.bins_shape, dtype=np.int64)
                xv = [_native.make_view(x.ctypes.data, _native.F64, 0, 0), _native.make_view(y.ctypes.data, _native.F64, 0, 0)]
                plan.execute(xv, None, 1, 1, out.ctypes.data, False, _native.MEM_HOST)
                if not np.array_equal(out, np.histogram2d(x, y, bins=[ea, eb])[0].astype(np.int64)):
                    bad[0] += 1
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    c = threading.Thread(target=churn)
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(12)]
    c.start()
    [t.start() for t in ts]
    [t.join() for t in ts]
    stop.set()
    c.join()
    assert not errors, errors
    assert bad[0] == 0, "%d of 4800 calls put their sample in the wrong bin" % bad[0]


def test_one_shot_entry_point_evicts_its_plan_cache_safely(xh):
    """xhist_bincount_rows keeps at most 64 plans and drops them all when full: threads that are still executing a
    dropped plan hold their own reference (the cache used to destroy plans in use)"""
    import ctypes as C

    from xhistogram_amd import _native

    lib = _native.load()
    rng = np.random.default_rng(5)
    x = rng.standard_normal(20_000)
    errors, bad = [], [0]

    def worker(k):
        try:
            for it in range(60):
                e = np.linspace(-4, 4 + 1e-3 * (k * 60 + it), 33)  # 480 distinct edge sets over 8 threads: many evictions
                out = np.empty(32, dtype=np.int64)
                view = (_native.XhistArray * 1)(_native.make_view(x.ctypes.data, _native.F64, x.size, 1))
                eptr = (C.c_void_p * 1)(e.ctypes.data)
                elen = (C.c_int64 * 1)(e.size)
                rc = lib.xhist_bincount_rows(0, 1, view, None, 1, x.size, eptr, elen, _native.CMP_F64, C.c_void_p(out.ctypes.data),
                                             _native.I64, _native.MEM_HOST, 0, None)
                if rc != 0:
                    errors.append((rc, lib.xhist_last_error()))
                    return
                if not np.array_equal(out, np.histogram(x, bins=e)[0]):
                    bad[0] += 1
        except Exception as exc:  # pragma: no cover
            errors.append(repr(exc))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors[:2]
    assert bad[0] == 0
    assert lib.xhist_shutdown() == 0
