"""GPU parity tests proper: the HIP path (through the C ABI) against the committed golden vectors
(outputs of the reference itself) and against the CPU oracle on seeded inputs.

Bar (north_star): unweighted int64 counts bit-exact; float64 weighted / density within 1e-6
relative (tests/conftest.py::assert_hist_equal).  Run with ``pytest -m gpu`` on an MI355X.
"""
import os

import numpy as np
import pytest

from conftest import MANIFEST, assert_hist_equal
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def xh():
    from xhistogram_amd import _native, core

    _native.load()
    assert _native.device_count() >= 1, "no MI355X visible: GPU tests must not pass on a fallback"
    assert "gfx950" in _native.device_info(0)["name"]
    return core


def _dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def _plan_for(core, samples, edges):
    dts = [core._np_dtype_of(s) for s in samples]
    cmp_domain, conv, _ = core._compare_domain(dts, edges)
    return core._get_plan(conv, cmp_domain, 0)


def _run(core, samples, edges, w, resident, **params):
    """hot path on host (numpy) or device-resident (torch) inputs with optional plan tuning"""
    if resident:
        if any(s.dtype.kind in "mM" or s.dtype in (np.uint16, np.uint32, np.uint64) for s in samples):
            pytest.skip("dtype has no torch equivalent")
        samples = [_dev(s) for s in samples]
        w = None if w is None else _dev(w)
    plan = _plan_for(core, samples, edges)
    for k, v in params.items():
        plan.set_param(k, v)
    try:
        out = core._bincount_2d_vectorized(*samples, bins=edges, weights=w)
        desc = plan.describe() if samples[0].shape[0] * samples[0].shape[1] and plan.n_bins else ""
    finally:
        for k in params:
            plan.set_param(k, 0)
    if resident:
        out = out.cpu().numpy()
    return out, desc


# ---------------------------------------------------------------------------------------------
# golden vectors (reference outputs)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("resident", [False, True], ids=["host", "device"])
@pytest.mark.parametrize("name", sorted(MANIFEST["hotpath"]))
def test_hotpath_golden(xh, golden, name, resident):
    samples, edges, w, want = golden.hotpath_case(name)
    got, _ = _run(xh, samples, edges, w, resident)
    assert got.dtype == want.dtype
    assert_hist_equal(got, want, weighted=w is not None)


@pytest.mark.parametrize("mode", ["force_global", "force_generic", "lds_copies1", "pack"])
@pytest.mark.parametrize("name", sorted(MANIFEST["hotpath"]))
def test_hotpath_golden_all_kernel_families(xh, golden, name, mode):
    samples, edges, w, want = golden.hotpath_case(name)
    params = {"lds_copies": 1} if mode == "lds_copies1" else {mode: 1}  # ("pack": packed bucket entries wherever the plan has them)
    got, desc = _run(xh, samples, edges, w, True, **params)
    if desc:
        if mode == "force_global":
            assert "hist=global" in desc, desc
        if mode == "force_generic":
            assert "family=generic" in desc, desc
    assert_hist_equal(got, want, weighted=w is not None)


@pytest.mark.parametrize("name", sorted(MANIFEST["core"]))
def test_public_api_golden(xh, golden, name):
    args, kw, want, meta = golden.core_case(name)
    got, edges = xh.histogram(*args, **kw)
    assert got.shape == tuple(meta["h_shape"])
    assert str(got.dtype) == meta["h_dtype"]
    assert_hist_equal(got, want, weighted=("weights" in kw) or kw.get("density", False))
    for i, e in enumerate(edges):
        np.testing.assert_array_equal(e, golden.core["%s/edges%d" % (name, i)])


@pytest.mark.parametrize("name", sorted(MANIFEST["core"]))
def test_public_api_golden_device_resident(xh, golden, name):
    args, kw, want, meta = golden.core_case(name)
    targs = [_dev(a) for a in args]
    tkw = dict(kw)
    if "weights" in kw:
        tkw["weights"] = _dev(kw["weights"])
    got, edges = xh.histogram(*targs, **tkw)
    assert isinstance(got, torch.Tensor) and got.is_cuda
    got = got.cpu().numpy()
    assert got.shape == tuple(meta["h_shape"])
    assert_hist_equal(got, want, weighted=("weights" in kw) or kw.get("density", False))
    for i, e in enumerate(edges):
        np.testing.assert_array_equal(e, golden.core["%s/edges%d" % (name, i)])


@pytest.mark.parametrize("name", sorted(MANIFEST["dask_cases"]))
def test_dask_cases_unchunked_equivalence(xh, golden, name):
    """reference dask-branch outputs (blockwise + sum) == our single-launch result"""
    args, kw, want, meta = golden.core_case(name, "dask_cases")
    got, _ = xh.histogram(*args, **kw)
    assert_hist_equal(got, want, weighted=("weights" in kw) or kw.get("density", False))


# ---------------------------------------------------------------------------------------------
# seeded comparisons with the oracle at sizes it finishes in seconds (BASELINE configs, scaled)
# ---------------------------------------------------------------------------------------------
def _nonuniform_edges(rng, n):
    e = np.sort(rng.uniform(-4, 4, n))
    e[0], e[-1] = -4.0, 4.0
    return e


CONFIGS = {
    # name: (D, n, dtype, edges builder, weighted)
    "C1_1d_f64_100bins": (1, 1_000_000, np.float64, lambda r: [np.linspace(-4, 4, 101)], False),
    "C2_1d_f64_100bins_weighted": (1, 4_000_003, np.float64, lambda r: [np.linspace(-4, 4, 101)], True),
    "C3_2d_256x256_nonuniform": (2, 2_000_000, np.float64, lambda r: [_nonuniform_edges(r, 257), _nonuniform_edges(r, 257)], False),
    "C5_2d_1024x1024_weighted": (2, 2_000_000, np.float64, lambda r: [np.linspace(-4, 4, 1025)] * 2, True),
    "f32_50bins": (1, 3_000_001, np.float32, lambda r: [np.linspace(-4, 4, 51)], False),
    "3d_f32_weighted": (3, 1_000_000, np.float32, lambda r: [np.linspace(-3, 3, 13), np.linspace(-3, 3, 9), np.linspace(-3, 3, 17)], True),
}


@pytest.mark.parametrize("resident", [False, True], ids=["host", "device"])
@pytest.mark.parametrize("cfg", sorted(CONFIGS))
def test_configs_vs_oracle(xh, cfg, resident):
    d, n, dt, mk, weighted = CONFIGS[cfg]
    import zlib
    rng = np.random.default_rng(zlib.crc32(cfg.encode()))
    samples = [rng.standard_normal((1, n)).astype(dt) for _ in range(d)]
    edges = mk(rng)
    w = rng.uniform(0, 1, (1, n)) if weighted else None
    want = onp.bincount_rows(samples, edges, w)
    got, _ = _run(xh, samples, edges, w, resident)
    assert_hist_equal(got, want, weighted)


def test_c4_rows_f32(xh):
    """C4 miniature: (T, lat*lon) f32 rows, 50 bins, reduced over the trailing axes"""
    rng = np.random.default_rng(4)
    x = rng.standard_normal((16, 72, 144)).astype(np.float32)
    edges = np.linspace(-4, 4, 51)
    want, _ = onp.histogram(x, bins=edges, axis=(1, 2))
    got, _ = xh.histogram(x, bins=edges, axis=(1, 2))
    np.testing.assert_array_equal(got, want)
    got_t, _ = xh.histogram(_dev(x), bins=edges, axis=(1, 2))
    np.testing.assert_array_equal(got_t.cpu().numpy(), want)


@pytest.mark.parametrize("dt", [np.float64, np.float32, np.float16, np.int64, np.int32, np.int16, np.int8, np.uint8, np.bool_])
def test_sample_dtypes(xh, dt):
    rng = np.random.default_rng(5)
    if np.dtype(dt).kind == "f":
        x = (rng.standard_normal((3, 5000)) * 3).astype(dt)
    elif np.dtype(dt).kind == "b":
        x = rng.integers(0, 2, (3, 5000)).astype(dt)
    else:
        info = np.iinfo(dt)
        x = rng.integers(max(info.min, -100), min(info.max, 100), (3, 5000)).astype(dt)
    edges = [np.linspace(-10, 10, 41)]
    want = onp.bincount_rows([x], edges)
    for resident in (False, True):
        got, _ = _run(xh, [x], edges, None, resident)
        np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("wdt", [np.float64, np.float32, np.int64, np.int32, np.uint8, np.bool_, np.float16])
def test_weight_dtypes(xh, wdt):
    rng = np.random.default_rng(6)
    x = rng.standard_normal((2, 7000))
    w = rng.integers(0, 5, (2, 7000)).astype(wdt)
    edges = [np.linspace(-3, 3, 25)]
    want = onp.bincount_rows([x], edges, w)
    for resident in (False, True):
        got, _ = _run(xh, [x], edges, w, resident)
        assert_hist_equal(got, want, True)


def test_strided_and_broadcast_views_device(xh):
    """row-broadcast, col-broadcast, transposed and sliced device views need no copies"""
    rng = np.random.default_rng(7)
    x = rng.standard_normal((6, 4000))
    edges = np.linspace(-3, 3, 31)
    w_row = rng.uniform(0, 1, (1, 4000))
    w_col = rng.uniform(0, 1, (6, 1))
    for w in (w_row, w_col):
        want, _ = onp.histogram(x, bins=edges, axis=1, weights=w)
        got, _ = xh.histogram(_dev(x), bins=edges, axis=1, weights=_dev(w))
        assert_hist_equal(got.cpu().numpy(), want, True)
        got_h, _ = xh.histogram(x, bins=edges, axis=1, weights=w)
        assert_hist_equal(got_h, want, True)
    # reduce over the leading axis: rows have stride 1, columns stride 4000
    want, _ = onp.histogram(x, bins=edges, axis=0)
    got, _ = xh.histogram(_dev(x), bins=edges, axis=0)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    np.testing.assert_array_equal(xh.histogram(x, bins=edges, axis=0)[0], want)
    # unaligned slice (element offset 1) falls off the vector-load family but must stay exact
    xs = _dev(x)[:, 1:3998]
    want, _ = onp.histogram(x[:, 1:3998], bins=edges, axis=1)
    np.testing.assert_array_equal(xh.histogram(xs, bins=edges, axis=1)[0].cpu().numpy(), want)


def test_bins_int_on_device_matches_numpy_edges(xh):
    rng = np.random.default_rng(8)
    x = rng.standard_normal(100_000)
    want, we = onp.histogram(x, bins=37)
    got, ge = xh.histogram(_dev(x), bins=37)
    np.testing.assert_array_equal(ge[0], we[0])
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    got, ge = xh.histogram(_dev(x.astype(np.float32)), bins=12, range=(-2, 2))
    want, we = onp.histogram(x.astype(np.float32), bins=12, range=(-2, 2))
    np.testing.assert_array_equal(ge[0], we[0])
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    with pytest.raises(ValueError):  # numpy: autodetected range of [nan, nan] is not finite
        xn = x.copy()
        xn[5] = np.nan
        xh.histogram(_dev(xn), bins=10)


def test_block_size_never_changes_results(xh):
    rng = np.random.default_rng(9)
    x = rng.standard_normal((9, 3000))
    edges = np.linspace(-4, 4, 10)
    base, _ = xh.histogram(x, bins=edges, axis=1, block_size=None)
    for bs in (1, 2, 4, 100, "auto"):
        np.testing.assert_array_equal(xh.histogram(x, bins=edges, axis=1, block_size=bs)[0], base)


# ---------------------------------------------------------------------------------------------
# reference known-answer tests, restated (test_core.py:72-113, test_xarray.py:38-67 idea)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("block_size", [None, 1, 2])
def test_right_edge(xh, block_size):  # test_core.py:95-113
    data = np.ones((5, 20))
    bins = np.array([0, 0.5, 1])
    h, _ = xh.histogram(data, bins=bins, axis=1, block_size=block_size)
    assert h.shape == (5, 2)
    np.testing.assert_array_equal(h.sum(axis=0), np.histogram(data, bins=bins)[0])
    np.testing.assert_array_equal(xh.histogram(data, bins=bins, block_size=block_size)[0], np.histogram(data, bins=bins)[0])


@pytest.mark.parametrize("block_size", [None, 1, 2, "auto"])
def test_weights_twice_counts_exact(xh, block_size):  # test_core.py:72-92
    rng = np.random.default_rng(10)
    data = rng.standard_normal((5, 20))
    bins = np.linspace(-4, 4, 10)
    h, _ = xh.histogram(data, bins=bins, axis=1, block_size=block_size)
    h_w, _ = xh.histogram(data, bins=bins, axis=1, weights=2 * np.ones_like(data), block_size=block_size)
    np.testing.assert_array_equal(2 * h, h_w)
    h_b, _ = xh.histogram(data, bins=bins, axis=1, weights=2 * np.ones((1, 20)), block_size=block_size)
    np.testing.assert_array_equal(2 * h, h_b)


def test_all_ones_known_answer(xh):  # test_xarray.py:38-67 on the numpy API
    for shape in ((7,), (4, 5), (2, 3, 4), (2, 3, 4, 5)):
        ones = np.ones(shape)
        h, _ = xh.histogram(ones, bins=np.array([0.0, 0.9, 1.1, 2.0]))
        np.testing.assert_array_equal(h, [0, ones.size, 0])


def test_vs_numpy_histogramdd_density(xh):  # test_core.py:160-228
    rng = np.random.default_rng(12)
    a, b, c = (rng.standard_normal((5, 20)) for _ in range(3))
    a.ravel()[rng.choice(100, 20, replace=False)] = np.nan
    ba, bb, bc = np.linspace(-4, 4, 10), np.linspace(-4, 4, 11), np.linspace(-4, 4, 10)
    h, _ = xh.histogram(a, b, c, bins=[ba, bb, bc], density=True)
    want = np.histogramdd((a.ravel(), b.ravel(), c.ravel()), bins=[ba, bb, bc], density=True)[0]
    np.testing.assert_allclose(h, want, rtol=1e-6)
    areas = np.einsum("i,j,k", np.diff(ba), np.diff(bb), np.diff(bc))
    np.testing.assert_allclose(np.sum(h * areas), 1.0, rtol=1e-9)


def test_datetime64(xh):  # test_core.py:365-382
    data = np.array(["2000-06-0%d" % d for d in range(1, 6)], dtype="datetime64[ns]")
    bins = np.array([np.datetime64("1999-01-01"), np.datetime64("2000-01-01"), np.datetime64("2001-01-01")])
    h = xh.histogram(data, bins=bins)[0]
    np.testing.assert_array_equal(h, np.histogram(data.view("i8"), bins=bins.astype("datetime64[ns]").view("i8"))[0])
    np.testing.assert_array_equal(h, [0, 5])


def test_errors_mirror_reference(xh):
    x = np.zeros((3, 4))
    with pytest.raises(ValueError):
        xh.histogram(x, bins=None)
    with pytest.raises(ValueError):
        xh.histogram(x, x, bins=[np.linspace(0, 1, 3)])
    with pytest.raises(ValueError):
        xh.histogram(x, bins=np.array([0.0, 2.0, 1.0]))  # numpy: bins must increase monotonically
    with pytest.raises(ValueError):
        xh.histogram(x, bins=5, range=[(0, 1), (0, 1)])
    with pytest.raises(TypeError):
        xh.histogram(x, bins="auto", weights=np.ones_like(x))
    with pytest.raises(TypeError):
        xh.histogram(x, bins=np.linspace(0, 1, 3), weights=np.ones_like(x) * 1j)
    with pytest.raises(AssertionError):
        xh.histogram(x, bins=np.linspace(0, 1, 3), axis=2)


def test_empty_inputs(xh):
    edges = np.linspace(0, 1, 5)
    h, _ = xh.histogram(np.zeros((3, 0)), bins=edges, axis=1)
    np.testing.assert_array_equal(h, np.zeros((3, 4), dtype=np.int64))
    h, _ = xh.histogram(torch.zeros((3, 0), dtype=torch.float64, device="cuda"), bins=edges, axis=1)
    np.testing.assert_array_equal(h.cpu().numpy(), np.zeros((3, 4), dtype=np.int64))
    h, _ = xh.histogram(np.zeros((0, 7)), bins=edges, axis=1)
    assert h.shape == (0, 4)


# ---------------------------------------------------------------------------------------------
# full-size, size-independent properties (BASELINE C2 / C3 sizes; the oracle would take minutes)
# ---------------------------------------------------------------------------------------------
def _bench():
    """bench.py's in-process checker (torch ops only), shared with the full-size tests"""
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench

    return bench


@pytest.fixture(scope="module")
def big():
    n = 1_000_000_000
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    x = torch.empty(n, dtype=torch.float64, device="cuda")
    x.normal_(generator=g)
    w = torch.empty(n, dtype=torch.float64, device="cuda")
    w.uniform_(generator=g)
    yield x, w
    del x, w
    torch.cuda.empty_cache()


def test_full_size_c2_properties(xh, big):
    x, w = big
    n = x.numel()
    edges = np.linspace(-4, 4, 101)
    h, _ = xh.histogram(x, bins=edges)
    h2, _ = xh.histogram(x, bins=edges)
    assert h.dtype == torch.int64
    assert torch.equal(h, h2), "int64 counts must be deterministic run to run"
    in_range = int(((x >= -4.0) & (x <= 4.0)).sum().item())
    assert int(h.sum().item()) == in_range
    # linearity over a split of the sample axis (what the dask sum / multi-GPU all-reduce relies on)
    k = 400_000_123
    ha, _ = xh.histogram(x[:k], bins=edges)
    hb, _ = xh.histogram(x[k:], bins=edges)
    assert torch.equal(ha + hb, h)
    # oracle on a 2e6 prefix, exact
    pre = x[:2_000_000].cpu().numpy().reshape(1, -1)
    np.testing.assert_array_equal(xh.histogram(x[:2_000_000], bins=edges)[0].cpu().numpy(), onp.bincount_rows([pre], [edges])[0])
    # weighted: total weight of in-range samples, and weights == 2 gives exactly 2 x counts
    hw, _ = xh.histogram(x, bins=edges, weights=w)
    total = float(w[(x >= -4.0) & (x <= 4.0)].sum().item())
    assert abs(float(hw.sum().item()) - total) <= 1e-9 * total
    # per-bin check against a torch f64 reference of the same op (bucketize == searchsorted right)
    e_t = torch.as_tensor(edges, device="cuda")
    idx = torch.bucketize(x, e_t, right=True)
    idx = torch.where(x == edges[-1], idx - 1, idx)
    ref_counts = torch.bincount(idx, minlength=102)[1:101]
    assert torch.equal(ref_counts, h)
    ref_w = torch.bincount(idx, weights=w, minlength=102)[1:101]
    torch.testing.assert_close(hw, ref_w, rtol=1e-6, atol=0)
    assert n == 1_000_000_000


def test_full_size_c3_properties(xh, big):
    x, y = big  # second array reused as the other coordinate after an affine map to [-4, 4)
    y = y * 8.0 - 4.0
    rng = np.random.default_rng(1)
    ea, eb = _nonuniform_edges(rng, 257), _nonuniform_edges(rng, 257)
    h, _ = xh.histogram(x, y, bins=[ea, eb])
    assert h.shape == (256, 256) and h.dtype == torch.int64
    in_range = int(((x >= -4.0) & (x <= 4.0) & (y >= -4.0) & (y <= 4.0)).sum().item())
    assert int(h.sum().item()) == in_range
    # marginals equal the 1-D histograms of each coordinate restricted to the other's range
    hx, _ = xh.histogram(x[(y >= -4.0) & (y <= 4.0)], bins=ea)
    assert torch.equal(h.sum(dim=1), hx)
    hy, _ = xh.histogram(y[(x >= -4.0) & (x <= 4.0)], bins=eb)
    assert torch.equal(h.sum(dim=0), hy)
    m = 3_000_000
    want = onp.bincount_rows([x[:m].cpu().numpy().reshape(1, -1), y[:m].cpu().numpy().reshape(1, -1)], [ea, eb])[0]
    np.testing.assert_array_equal(xh.histogram(x[:m], y[:m], bins=[ea, eb])[0].cpu().numpy(), want)
    # VERDICT r4 "next" #3: the same per-bin cross-check C2 gets, at all 10^9 pairs — an independent restatement in torch ops
    # (bucketize + last-edge rule + joint index + bincount; bench.torch_reference, itself pinned to the oracle by
    # tests/test_bench_reference.py), every one of the 65 536 bins
    ref = _bench().torch_reference(torch, [x, y], None, [ea, eb], 1, x.numel(), False)
    assert torch.equal(ref.reshape(256, 256), h)
    # samples ON edges at full size: a slice of x overwritten with edge values of both precisions' neighbours
    xe = x.clone()
    k = 1 << 20
    xe[:k] = torch.as_tensor(np.resize(np.concatenate([ea, np.nextafter(ea, -np.inf), np.nextafter(ea, np.inf)]), k), device="cuda")
    he, _ = xh.histogram(xe, y, bins=[ea, eb])
    assert torch.equal(_bench().torch_reference(torch, [xe, y], None, [ea, eb], 1, x.numel(), False).reshape(256, 256), he)
    del xe, ref


# ---------------------------------------------------------------------------------------------
# packed-uint16 LDS mode (mid-size joint histograms): wrap bookkeeping must be exact
# ---------------------------------------------------------------------------------------------
def test_packed16_mode_wraps_are_exact(xh):
    rng = np.random.default_rng(21)
    edges = [np.linspace(0, 256, 257), np.linspace(0, 256, 257)]  # 65536 bins -> packed16 in LDS
    n = 6_000_000
    # heavy hitters: one even bin, its odd neighbour (same LDS word), the very last bin, plus noise
    x = np.full(n, 10.5)
    y = np.full(n, 20.5)          # flat = 10*256+20 (even)
    y[1::3] = 21.5                # flat+1 (odd, same word): carries from the low half land here
    x[2::7], y[2::7] = 255.5, 255.5  # last bin (odd half of the last word)
    m = rng.integers(0, n, 200_000)
    x[m] = rng.uniform(-5, 260, m.size)
    y[m] = rng.uniform(-5, 260, m.size)
    samples = [x.reshape(1, -1), y.reshape(1, -1)]
    want = onp.bincount_rows(samples, edges)
    got, desc = _run(xh, samples, edges, None, True)
    assert "hist=packed16" in desc, desc
    np.testing.assert_array_equal(got, want)
    assert want.max() > 3 * 65536  # the test really wrapped 16-bit halves several times
    # odd number of bins: the last word has an unused high half
    e1 = [np.linspace(0, 1, 14_002)]  # 14001 bins + 128 KB of edge/bucket tables: uint32 no longer fits -> packed16
    z = rng.uniform(-0.1, 1.1, (1, 3_000_000))
    z[0, ::2] = 0.999999  # last bin, wraps
    z[0, 1::4] = 0.0
    got, desc = _run(xh, [z], e1, None, True)
    assert "hist=packed16" in desc, desc
    np.testing.assert_array_equal(got, onp.bincount_rows([z], e1))


def test_f32_threshold_domain_is_exact_at_edges(xh):
    """float32 samples are compared in float32 against thresholds thr = min{f32 >= edge}; probe
    every float32 neighbour of every edge, where a rounding slip would move a sample"""
    edges = np.linspace(-4, 4, 51)  # most edges are not float32-representable
    e32 = edges.astype(np.float32)
    pts = np.concatenate([e32, np.nextafter(e32, np.float32(np.inf)), np.nextafter(e32, np.float32(-np.inf)),
                          np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 4.0, -4.0], dtype=np.float32)])
    x = np.tile(pts, 300).astype(np.float32).reshape(3, -1)
    want = onp.bincount_rows([x], [edges])
    for resident in (False, True):
        got, desc = _run(xh, [x], [edges], None, resident, arith32=-1)  # (left alone, np.linspace edges take the float32 arithmetic: scan=9)
        np.testing.assert_array_equal(got, want)
    assert "cmp=f32thr" in desc, desc
    got, desc = _run(xh, [x], [edges], None, True)
    assert "scan=9" in desc, desc
    np.testing.assert_array_equal(got, want)
    # last edge not representable in float32: no float32 sample can equal it
    e2 = np.array([0.0, 0.1, 0.30000000000000004])
    x2 = np.array([[0.3, 0.30000001192092896, 0.29999998211860657, 0.1, 0.0]], dtype=np.float32)
    np.testing.assert_array_equal(_run(xh, [x2], [e2], None, True)[0], onp.bincount_rows([x2], [e2]))
    # huge / infinite edges
    e3 = np.array([-1e300, -1.0, 1.0, 1e300])
    x3 = np.array([[-np.inf, -3.4e38, -1.0, 0.0, 1.0, 3.4e38, np.inf, np.nan]], dtype=np.float32)
    np.testing.assert_array_equal(_run(xh, [x3], [e3], None, True)[0], onp.bincount_rows([x3], [e3]))


def test_more_than_2_32_samples_int64_indexing(xh):
    """> 2^32 samples in one row (BASELINE C5 is 4e9 in total): 64-bit indexing, uint32 LDS
    counters flushed per workgroup; known answer from a periodic pattern"""
    n = (1 << 32) + 12_345_678
    pat = torch.tensor([-5.0, -3.5, -0.5, 0.5, 0.5, 3.5, 4.0, float("nan")], dtype=torch.float32, device="cuda")
    x = pat.repeat((n + 7) // 8)[:n]
    edges = np.array([-4.0, -1.0, 0.0, 1.0, 4.0])
    h, _ = xh.histogram(x, bins=edges)
    full, rem = divmod(n, 8)
    want = np.array([1, 1, 2, 2], dtype=np.int64) * full
    tail = pat[:rem].cpu().numpy()
    want += onp.bincount_rows([tail.reshape(1, -1)], [edges])[0]
    np.testing.assert_array_equal(h.cpu().numpy(), want)
    del x
    torch.cuda.empty_cache()


def test_leading_axis_reduction_large_device_view(xh):
    """dim='time' of (time, y, x): columns are strided; must equal the oracle (and not crawl)"""
    rng = np.random.default_rng(31)
    t = rng.standard_normal((300, 64, 80)).astype(np.float32)
    edges = np.linspace(-4, 4, 51)
    want, _ = onp.histogram(t, bins=edges, axis=0)
    got, _ = xh.histogram(_dev(t), bins=edges, axis=0)
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    got, _ = xh.histogram(_dev(t), bins=edges, axis=(0, 2))
    np.testing.assert_array_equal(got.cpu().numpy(), onp.histogram(t, bins=edges, axis=(0, 2))[0])


# ---------------------------------------------------------------------------------------------
# dense short rows streamed flat (hist_flat_rows)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(5000, 365), (4097, 20), (9000, 1), (4097, 3), (7777, 64), (5001, 255), (5003, 256), (4099, 800),
                                   (6000, 100), (4096, 127)])
@pytest.mark.parametrize("edges_kind", ["linspace", "uneven", "many"])
def test_dense_short_rows_streamed_flat(xh, shape, dtype, edges_kind):
    """Many dense short rows of one unweighted input (histogram over the last axis of (time x lat, lon), say) are streamed as
    ONE contiguous array, a sample's row being its position divided by the row length (hist_flat_rows, xhist_lanes.hip.h):
    row lengths around the vector width and the copy thresholds, a row count that is no multiple of the rows per workgroup,
    an array whose size is no multiple of the 16-byte vector, NaN / out-of-range / right-edge samples."""
    rng = np.random.default_rng(shape[1] + len(edges_kind))
    edges = {"linspace": np.linspace(-3, 3, 51), "uneven": np.sort(rng.uniform(-3, 3, 38)), "many": np.linspace(-4, 4, 401)}[edges_kind]
    x = rng.standard_normal(shape).astype(dtype)
    x[::7, ::5] = np.nan
    x[3, :] = edges[-1] if float(dtype(edges[-1])) == edges[-1] else 0.0
    x[5, :] = 100.0
    x[-1, :] = -0.5
    want = onp.bincount_rows([x], [edges], None)
    got, desc = _run(xh, [x], [edges], None, True)
    assert "family=flat_rows" in desc, desc
    np.testing.assert_array_equal(got, want)
    got, desc = _run(xh, [x], [edges], None, True, flat_rows=-1)  # (and the kernels it replaced)
    assert "family=flat_rows" not in desc, desc
    np.testing.assert_array_equal(got, want)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("dims,weights", [(d, w) for d in (1, 2) for w in (None, "f32", "f64", "both_signs_and_nan")
                                          if not (d == 1 and w is None)])  # (one input without weights: test_dense_short_rows_streamed_flat)
@pytest.mark.parametrize("shape", [(5000, 365), (4097, 20), (9000, 1), (4097, 3), (7777, 64), (5001, 255), (4099, 800)])
def test_dense_short_rows_streamed_flat_weights_and_joint(xh, shape, dims, weights, dtype):
    """hist_flat_rows with weights (float64 sums in LDS, one copy) and with two inputs (joint bins): the reference's per-row
    result, NaN weights poisoning only their own bin, NaN weights on dropped samples discarded (core.py:73-83)."""
    rng = np.random.default_rng(shape[1] * 3 + dims)
    xs = [rng.standard_normal(shape).astype(dtype) for _ in range(dims)]
    xs[0][::7, ::5] = np.nan
    xs[-1][5, :] = 100.0
    w = {None: None, "f32": rng.uniform(0, 1, shape).astype(np.float32), "f64": rng.uniform(0, 1, shape),
         "both_signs_and_nan": rng.standard_normal(shape)}[weights]
    if weights == "both_signs_and_nan":
        w[::11, ::3] = np.nan  # some on dropped samples (x is NaN at [::77, ::15]), some on binned ones
    bins = [np.linspace(-3, 3, 31), np.sort(rng.uniform(-3, 3, 12))][:dims]
    want = onp.bincount_rows(xs, bins, w)
    got, desc = _run(xh, xs, bins, w, True)
    assert "family=flat_rows" in desc and "D=%d" % dims in desc, desc
    assert_hist_equal(got, want, w is not None)
    got, desc = _run(xh, xs, bins, w, True, flat_rows=-1)
    assert "family=flat_rows" not in desc, desc
    assert_hist_equal(got, want, w is not None)


def test_dense_short_rows_flat_needs_alignment_and_falls_back(xh):
    """the flat kernel reads aligned 16-byte vectors: a view that starts 4 bytes into an allocation takes the older kernels"""
    rng = np.random.default_rng(8)
    m, c = 5000, 100
    edges = np.linspace(-3, 3, 51)
    flat = torch.as_tensor(rng.standard_normal(m * c + 1).astype(np.float32)).cuda()
    x = flat[1:].view(m, c)
    assert x.data_ptr() % 16 == 4
    got, _ = xh.histogram(x, bins=edges, axis=1)
    plan = _plan_for(xh, [x], [edges])
    assert "family=flat_rows" not in plan.describe(), plan.describe()
    np.testing.assert_array_equal(got.cpu().numpy(), onp.bincount_rows([x.cpu().numpy()], [edges], None))
    got, _ = xh.histogram(flat[4:4 + (m - 1) * c].view(m - 1, c), bins=edges, axis=1)  # 16 bytes in: aligned again
    assert "family=flat_rows" in plan.describe(), plan.describe()
    np.testing.assert_array_equal(got.cpu().numpy(), onp.bincount_rows([flat[4:4 + (m - 1) * c].view(m - 1, c).cpu().numpy()], [edges], None))


def test_dense_short_rows_flat_in_row_blocks(xh):
    """block_size cuts the rows into separate launches on one output (core.py:86-134): blocks that start on a 16-byte
    boundary stream flat, the others take the older kernels — same result either way"""
    rng = np.random.default_rng(10)
    edges = np.linspace(-3, 3, 41)
    for shape, dtype, bs in [((20000, 40), np.float64, 5000), ((20004, 33), np.float32, 5001), ((16384, 100), np.float32, 4096)]:
        x = rng.standard_normal(shape).astype(dtype)
        w = rng.uniform(0, 1, shape)
        want = onp.histogram(x, bins=edges, axis=1)[0]
        got = xh.histogram(_dev(x), bins=edges, axis=1, block_size=bs)[0]
        np.testing.assert_array_equal(got.cpu().numpy(), want)
        wantw = onp.histogram(x, bins=edges, axis=1, weights=w)[0]
        gotw = xh.histogram(_dev(x), bins=edges, axis=1, weights=_dev(w), block_size=bs)[0]
        assert_hist_equal(gotw.cpu().numpy(), wantw, True)


@pytest.mark.parametrize("weighted", [False, True])
def test_accumulate_into_an_existing_output_short_rows_and_leading_axis(xh, weighted):
    """xhist_plan_execute(accumulate=1) adds to what the output holds (the C ABI's contract for callers that bin a stream in
    pieces): through the flat short-row kernel and through the row-per-lane kernels, whose plain-store flush must turn into
    adds — twice the same call gives twice the histogram."""
    from xhistogram_amd import _native

    rng = np.random.default_rng(12)
    edges = np.linspace(-3, 3, 41)
    plan = xh._get_plan([edges], _native.CMP_F64, 0)
    stream = torch.cuda.current_stream().cuda_stream
    for shape, lead in [((5000, 100), False), ((300, 6000), True)]:
        x = rng.standard_normal(shape).astype(np.float32)
        w = rng.uniform(0, 1, shape).astype(np.float32) if weighted else None
        xt, wt = _dev(x), (None if w is None else _dev(w))
        if lead:  # rows are the contiguous direction: reduce over the leading axis
            rows, cols, rs, cs = shape[1], shape[0], 1, shape[1]
            want = onp.histogram(x, bins=edges, axis=0, weights=w)[0]
        else:
            rows, cols, rs, cs = shape[0], shape[1], shape[1], 1
            want = onp.histogram(x, bins=edges, axis=1, weights=w)[0]
        out = torch.zeros((rows, 40), dtype=torch.float64 if weighted else torch.int64, device="cuda")
        xv = [_native.make_view(xt.data_ptr(), _native.F32, rs, cs)]
        wv = _native.make_view(wt.data_ptr(), _native.F32, rs, cs) if weighted else None
        for k in (1, 2, 3):
            plan.execute(xv, wv, rows, cols, out.data_ptr(), weighted, _native.MEM_DEVICE, accumulate=True, stream=stream)
            torch.cuda.synchronize()
            assert ("family=lanes" if lead else "family=flat_rows") in plan.describe(), plan.describe()
            assert_hist_equal(out.cpu().numpy(), k * want, weighted)


def test_dense_short_rows_flat_any_length_when_forced(xh):
    """"flat_rows" = 1: any row length below 65536 (the multiply-high row index, several iterations per row)"""
    rng = np.random.default_rng(9)
    edges = np.linspace(-3, 3, 51)
    for shape in [(4096, 5000), (4100, 1023), (5000, 16385)]:
        x = rng.standard_normal(shape).astype(np.float32)
        got, desc = _run(xh, [x], [edges], None, True, flat_rows=1)
        assert "family=flat_rows" in desc, desc
        np.testing.assert_array_equal(got, onp.bincount_rows([x], [edges], None))


# ---------------------------------------------------------------------------------------------
# partitioned multi-pass mode (histograms beyond LDS, BASELINE C5)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("weighted", [False, True])
@pytest.mark.parametrize("n", [5_000_000, 4_194_304 + 777, 1023])
def test_partitioned_mode_c5_shape(xh, weighted, n):
    rng = np.random.default_rng(41 + n % 7)
    x = rng.standard_normal((1, n))
    y = rng.standard_normal((1, n)) * 1.5
    x[0, ::1001] = np.nan
    y[0, 5::997] = 4.0  # right edge of the last bin
    w = rng.uniform(0, 1, (1, n)) if weighted else None
    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    want = onp.bincount_rows([x, y], edges, w)
    got, desc = _run(xh, [x, y], edges, w, True, partition=1)
    assert "hist=partitioned" in desc, desc
    assert_hist_equal(got, want, weighted)
    got2, desc2 = _run(xh, [x, y], edges, w, True, partition=-1)
    assert "hist=global" in desc2, desc2
    assert_hist_equal(got2, want, weighted)


def test_partitioned_mode_3d_f32_nonuniform(xh):
    rng = np.random.default_rng(43)
    n = 3_000_000
    s = [rng.standard_normal((1, n)).astype(np.float32) for _ in range(3)]
    edges = [_nonuniform_edges(rng, 129), np.linspace(-4, 4, 129), _nonuniform_edges(rng, 65)]
    want = onp.bincount_rows(s, edges)
    got, desc = _run(xh, s, edges, None, True, partition=1)
    assert "hist=partitioned" in desc, desc
    np.testing.assert_array_equal(got, want)
    w = rng.uniform(0, 2, (1, n)).astype(np.float32)
    got, desc = _run(xh, s, edges, w, True, partition=1)
    assert "hist=partitioned" in desc, desc
    assert_hist_equal(got, onp.bincount_rows(s, edges, w), True)


def test_partitioned_mode_skewed_everything_in_one_bin(xh):
    """worst case for the slot reservation: one partition receives every sample"""
    n = 6_000_000
    x = np.full((1, n), 0.123)
    y = np.full((1, n), -2.5)
    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    got, desc = _run(xh, [x, y], edges, None, True, partition=1)
    assert "hist=partitioned" in desc, desc
    assert got.sum() == n and got.max() == n
    np.testing.assert_array_equal(got, onp.bincount_rows([x, y], edges))


@pytest.mark.parametrize("weighted", [False, True])
def test_partitioned_mode_more_than_128_partitions(xh, weighted):
    """> 128 partitions: records leave the scatter pass in groups of 4 instead of 8"""
    rng = np.random.default_rng(47)
    n = 2_500_003
    nb = 1500 if weighted else 2100  # 2.25 M float64 bins / 4.41 M uint32 bins = 138 / 135 partitions
    x = rng.standard_normal((1, n))
    y = rng.uniform(-4.2, 4.2, (1, n))
    w = rng.uniform(-1, 1, (1, n)) if weighted else None
    edges = [np.linspace(-4, 4, nb + 1), np.linspace(-4, 4, nb + 1)]
    want = onp.bincount_rows([x, y], edges, w)
    got, desc = _run(xh, [x, y], edges, w, True, partition=1)
    assert "hist=partitioned" in desc and "group=4" in desc, desc
    assert_hist_equal(got, want, weighted)


@pytest.mark.parametrize("case", ["w_f32_200x200", "u_f64_300x300", "w_1d_50000_arith", "u_3d_f32", "rows_40"])
def test_bin_slices_for_histograms_of_a_few_times_lds(xh, case):
    """histograms of a few times the LDS capacity: S launches, each keeping one slice of the bins in LDS"""
    rng = np.random.default_rng(46)
    n = 400_000
    if case == "w_f32_200x200":
        s = [rng.standard_normal((2, n)).astype(np.float32), (rng.standard_normal((2, n)) * 1.2).astype(np.float32)]
        edges = [np.linspace(-4, 4, 201), _nonuniform_edges(rng, 201)]
        w = rng.uniform(0, 1, (2, n)).astype(np.float32)
    elif case == "u_f64_300x300":
        s = [rng.standard_normal((1, n)), rng.uniform(-4.5, 4.5, (1, n))]
        edges = [np.linspace(-4, 4, 301), np.linspace(-4, 4, 301)]
        w = None
    elif case == "w_1d_50000_arith":
        edges = [np.linspace(-3, 3, 50_001)]
        s = [_edge_torture(edges[0], rng, n)]
        w = rng.uniform(-1, 1, s[0].shape)
    elif case == "u_3d_f32":
        s = [rng.standard_normal((3, n)).astype(np.float32) for _ in range(3)]
        edges = [np.linspace(-3, 3, 61), np.linspace(-3, 3, 51), np.linspace(-3, 3, 41)]  # 122400 bins
        w = None
    else:  # many rows: not a shape the partitioned mode takes
        s = [rng.standard_normal((40, 30_000)), rng.standard_normal((40, 30_000))]
        edges = [np.linspace(-4, 4, 181), np.linspace(-4, 4, 161)]
        w = rng.uniform(0, 1, (40, 30_000))
    want = onp.bincount_rows(s, edges, w)
    # (test-sized inputs are small enough for memory-side atomics to win: the mode is asked for)
    got, desc = _run(xh, s, edges, w, True, slices=1)
    n_slices = int(desc.split("slices=")[1].split()[0])
    assert "family=fast" in desc and n_slices >= 2, desc
    assert_hist_equal(got, want, w is not None)
    assert_hist_equal(_run(xh, s, edges, w, False, slices=1)[0], want, w is not None)  # host route
    assert_hist_equal(_run(xh, s, edges, w, True)[0], want, w is not None)             # whatever the library picks
    got, desc = _run(xh, s, edges, w, True, slices=-1)                                 # and without slices
    assert "slices=1" in desc or "partitioned" in desc, desc
    assert_hist_equal(got, want, w is not None)


def test_partitioned_mode_a_few_long_rows(xh):
    """one beyond-LDS joint histogram per row (e.g. per time step): the partitioned mode row by row"""
    rng = np.random.default_rng(49)
    rows, n = 3, 1_200_000
    x = rng.standard_normal((rows, n))
    y = rng.standard_normal((rows, n)) * 1.3
    w = rng.uniform(0, 1, (rows, n))
    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    for ww in (None, w):
        got, desc = _run(xh, [x, y], edges, ww, True, partition=1)
        assert "hist=partitioned" in desc, desc
        assert_hist_equal(got, onp.bincount_rows([x, y], edges, ww), ww is not None)
    # rows of a strided parent (every second row) and weights broadcast along the rows
    xs = _dev(np.repeat(x, 2, axis=0))[::2]
    ys = _dev(np.repeat(y, 2, axis=0))[::2]
    wb = _dev(w[:1]).expand(rows, n)
    plan = _plan_for(xh, [xs, ys], edges)
    plan.set_param("partition", 1)
    try:
        got = xh._bincount_2d_vectorized(xs, ys, bins=edges, weights=wb).cpu().numpy()
        assert "hist=partitioned" in plan.describe()
    finally:
        plan.set_param("partition", 0)
    assert_hist_equal(got, onp.bincount_rows([x, y], edges, np.broadcast_to(w[:1], (rows, n))), True)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("weighted", [False, True])
def test_partitioned_mode_several_rows_in_one_pass(xh, dtype, weighted):
    """rows that follow each other at one stride go through the routing pass several at a time ((row, partition) is the
    partition): more rows than one pass takes, a ragged row length, a row stride above the row length, weights per row /
    broadcast over the rows / of both signs"""
    rng = np.random.default_rng(63)
    rows, n = 23, 70_001
    x = rng.standard_normal((rows, n + 5)).astype(dtype)[:, :n]  # row stride n + 5
    y = (rng.standard_normal((rows, n)) * 1.3).astype(dtype)
    x[3, ::501] = np.nan
    y[7, 5::499] = 3.0  # right edge of the last bin
    edges = [np.linspace(-3, 3, 301), np.linspace(-3, 3, 301)]  # 90 000 bins: 6 partitions (weighted) per row
    for w in ([None] if not weighted else [rng.uniform(0, 1, (rows, n)), np.broadcast_to(rng.uniform(0, 1, (1, n)), (rows, n)),
                                          rng.standard_normal((rows, n))]):
        want = onp.bincount_rows([x, y], edges, w)
        got, desc = _run(xh, [x, y], edges, w, True, partition=1)
        assert "hist=partitioned" in desc and "rows_per_pass=1 " not in desc, desc
        assert_hist_equal(got, want, weighted)
    # two rows, C5-sized histogram (64 partitions per row: exactly two rows per pass), three rows = one pass + one row
    e5 = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    for r in (2, 3):
        xs, ys = rng.standard_normal((r, 300_003)).astype(dtype), rng.standard_normal((r, 300_003)).astype(dtype)
        ws = rng.uniform(0, 1, xs.shape) if weighted else None
        got, desc = _run(xh, [xs, ys], e5, ws, True, partition=1)
        assert "hist=partitioned" in desc, desc
        assert_hist_equal(got, onp.bincount_rows([xs, ys], e5, ws), weighted)


def test_partitioned_mode_tiny_and_ragged_inputs(xh):
    """forced partitioned mode on inputs of a few records: carried records and padding only"""
    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    rng = np.random.default_rng(48)
    for n in (4, 5, 7, 8, 9, 8191, 8192, 8193, 16389):
        x = rng.standard_normal((1, n))
        y = rng.standard_normal((1, n))
        w = rng.uniform(0, 1, (1, n))
        for ww in (None, w):
            got, desc = _run(xh, [x, y], edges, ww, True, partition=1)
            assert "hist=partitioned" in desc, desc
            assert_hist_equal(got, onp.bincount_rows([x, y], edges, ww), ww is not None)


def test_partitioned_mode_packed_records_one_sign(xh):
    """float64 weights of one sign travel as 8-byte records (36 mantissa bits + the bin code): within 2^-37 per weight of the
    full-float64 records, and both inside the 1e-6 contract against the oracle"""
    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    rng = np.random.default_rng(60)
    n = 3_000_001
    x, y = rng.standard_normal((1, n)), rng.standard_normal((1, n))
    for w in (rng.uniform(0, 1, (1, n)), -rng.uniform(0, 1e300, (1, n)), rng.uniform(0, 1, (1, n)) * 1e-300,
              np.abs(rng.standard_normal((1, n))).astype(np.float64) + 0.0):
        want = onp.bincount_rows([x, y], edges, w)
        packed, desc = _run(xh, [x, y], edges, w, True, partition=1)
        assert "records=packed48" in desc, desc
        exact, desc = _run(xh, [x, y], edges, w, True, partition=1, records48=-1)
        assert "records=u16+f64" in desc, desc
        assert_hist_equal(packed, want, True)
        assert_hist_equal(exact, want, True)
        np.testing.assert_allclose(packed, exact, rtol=2.0 ** -36, atol=0)


def test_partitioned_mode_packed_records_fall_back_on_mixed_signs(xh):
    """weights of both signs: the packed attempt is discarded ON THE GPU and the exact pass of the same call produces the result
    (weights that cancel would turn 2^-37 per weight into anything per bin); the plan remembers and later calls go straight
    to exact records"""
    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    rng = np.random.default_rng(61)
    n = 2_500_003
    x, y = rng.standard_normal((1, n)), rng.standard_normal((1, n))
    w = rng.standard_normal((1, n)) * 1e6
    w[0, 1::2] = -w[0, ::2][: w[0, 1::2].size] * (1 + 1e-9)  # pairs that nearly cancel wherever they share a bin
    want = onp.bincount_rows([x, y], edges, w)
    exact, _ = _run(xh, [x, y], edges, w, True, partition=1, records48=-1)
    first, desc = _run(xh, [x, y], edges, w, True, partition=1, records48=0)  # (setting the knob forgets earlier calls)
    assert "records=packed48(+exact" in desc, desc
    np.testing.assert_allclose(first, exact, rtol=1e-9, atol=1e-3)  # same records, another order of the atomics
    scale = np.abs(w).max()
    np.testing.assert_allclose(first, want, rtol=1e-6, atol=1e-9 * scale)
    # one negative weight among positive ones is enough
    w2 = rng.uniform(0, 1, (1, n))
    w2[0, 12345] = -0.25
    got, _ = _run(xh, [x, y], edges, w2, True, partition=1, records48=0)
    ex2, _ = _run(xh, [x, y], edges, w2, True, partition=1, records48=-1)
    np.testing.assert_allclose(got, ex2, rtol=1e-12, atol=0)
    # the plan remembers: without touching the knob, the next call does not try packed records
    plan = _plan_for(xh, [_dev(x), _dev(y)], edges)
    plan.set_param("partition", 1)
    try:
        xh._bincount_2d_vectorized(_dev(x), _dev(y), bins=edges, weights=_dev(w2))
        torch.cuda.synchronize()
        again = xh._bincount_2d_vectorized(_dev(x), _dev(y), bins=edges, weights=_dev(w2)).cpu().numpy()
        assert "records=u16+f64" in plan.describe(), plan.describe()
    finally:
        plan.set_param("partition", 0)
        plan.set_param("records48", 0)
    np.testing.assert_allclose(again, ex2, rtol=1e-12, atol=0)


def test_partitioned_mode_packed_records_special_weights(xh):
    """NaN poisons its own bin only (whatever its payload), infinities stay infinities, the largest finite weight does not round
    up to infinity, -0.0 and denormals are harmless, a NaN weight on a dropped sample is dropped with it"""
    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    rng = np.random.default_rng(62)
    n = 1_000_003
    x, y = rng.standard_normal((1, n)), rng.standard_normal((1, n))
    w = rng.uniform(0, 1, (1, n))
    w[0, 10] = np.nan
    w[0, 11] = np.frombuffer(np.uint64(0x7FF0000000000001).tobytes(), np.float64)[0]  # NaN whose payload sits in the low 16 bits
    w[0, 12] = np.inf
    w[0, 13] = np.finfo(np.float64).max
    w[0, 14] = 5e-324
    w[0, 15] = 0.0
    x[0, 16], w[0, 16] = 9.0, np.nan  # out of range: dropped, weight and all
    x[0, 17], w[0, 17] = np.nan, -1.0  # dropped: its sign does not count
    want = onp.bincount_rows([x, y], edges, w)
    got, desc = _run(xh, [x, y], edges, w, True, partition=1, records48=0)
    assert "records=packed48" in desc, desc
    assert np.array_equal(np.isnan(got), np.isnan(want)) and np.array_equal(np.isinf(got), np.isinf(want))
    assert np.isnan(got).sum() == len({(np.searchsorted(edges[0], x[0, i], "right"), np.searchsorted(edges[1], y[0, i], "right")) for i in (10, 11)})
    fin = np.isfinite(want)
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-6, atol=0)


@pytest.mark.parametrize("resident", [False, True], ids=["host", "device"])
def test_per_input_compare_domains_datetime_and_int64_next_to_floats(xh, resident):
    """a time axis (datetime64 / int64, compared exactly) against a float axis, as numpy does per argument"""
    rng = np.random.default_rng(58)
    n = 200_000
    t0 = np.datetime64("2001-03-04T00:00:00", "s")
    t = (t0 + rng.integers(-10_000, 400 * 86400, (2, n)).astype("timedelta64[s]")).astype("datetime64[s]")
    t_edges = np.arange(np.datetime64("2001-03-01"), np.datetime64("2002-03-02"), np.timedelta64(1, "D"))  # datetime64[D]
    v = rng.standard_normal((2, n))
    v_edges = np.linspace(-3, 3, 25)
    w = rng.uniform(0, 1, (2, n))
    for samples, edges in (([t, v], [t_edges, v_edges]), ([v, t], [v_edges, t_edges])):
        for ww in (None, w):
            want = onp.bincount_rows(samples, edges, ww)
            if resident and any(s.dtype.kind == "M" for s in samples):
                # torch has no datetime64: the int64 view (in the common unit) is what a torch user holds
                common = np.result_type(t.dtype, t_edges.dtype)
                samples_t = [s.astype(common).view(np.int64) if s.dtype.kind == "M" else s for s in samples]
                edges_t = [e.astype(common).view(np.int64) if e.dtype.kind == "M" else e for e in edges]
                got, desc = _run(xh, samples_t, edges_t, ww, True)
            else:
                got, desc = _run(xh, samples, edges, ww, resident)
            assert_hist_equal(got, want, ww is not None)
    if not resident:  # the public entry point: datetime64 unit alignment + per-input domains + axis handling
        h, _ = xh.histogram(t, v, bins=[t_edges, v_edges], axis=1, weights=w)
        assert_hist_equal(h, onp.histogram(t, v, bins=[t_edges, v_edges], axis=1, weights=w)[0], True)
    # int64 values beyond 2^53 (float64 would merge neighbours) next to a float input
    big = (1 << 60) + rng.integers(0, 40, (1, 50_000))
    big_edges = (1 << 60) + np.arange(0, 41, 2)
    f = rng.uniform(0, 1, (1, 50_000)).astype(np.float32)
    f_edges = np.linspace(0, 1, 5)
    got, desc = _run(xh, [big, f], [big_edges, f_edges], None, resident)
    assert "cmp=per-input" in desc, desc
    np.testing.assert_array_equal(got, onp.bincount_rows([big, f], [big_edges, f_edges]))
    got = _run(xh, [f, big, f], [f_edges, big_edges, f_edges], None, resident)[0]
    np.testing.assert_array_equal(got, onp.bincount_rows([f, big, f], [f_edges, big_edges, f_edges]))


def test_unsigned_64_bit_samples_and_edges(xh):
    """uint64 data against uint64 edges (numpy compares them as uint64): values on both sides of 2^63"""
    rng = np.random.default_rng(57)
    base = np.uint64((1 << 63) - 20)
    x = (base + rng.integers(0, 60, (2, 40_000)).astype(np.uint64)).astype(np.uint64)
    x[0, :5] = [0, 1, np.iinfo(np.uint64).max, (1 << 63) - 1, 1 << 63]
    edges = [(base + np.arange(0, 50, 3).astype(np.uint64)).astype(np.uint64)]
    assert edges[0][0] < (1 << 63) < edges[0][-1]
    want = onp.bincount_rows([x], edges)
    got, desc = _run(xh, [x], edges, None, False)
    assert "cmp=i64" in desc, desc
    np.testing.assert_array_equal(got, want)
    w = rng.uniform(0, 1, x.shape)
    assert_hist_equal(_run(xh, [x], edges, w, False)[0], onp.bincount_rows([x], edges, w), True)
    # smaller unsigned samples against uint64 edges, next to a float input
    s8 = rng.integers(0, 256, (2, 40_000)).astype(np.uint8)
    e8 = [np.arange(0, 300, 7).astype(np.uint64)]
    np.testing.assert_array_equal(_run(xh, [s8], e8, None, False)[0], onp.bincount_rows([s8], e8))
    f = rng.standard_normal((2, 40_000))
    ef = np.linspace(-2, 2, 6)
    got, desc = _run(xh, [f, x], [ef, edges[0]], None, False)
    assert "cmp=per-input" in desc, desc
    np.testing.assert_array_equal(got, onp.bincount_rows([f, x], [ef, edges[0]]))
    h, _ = xh.histogram(x, bins=edges[0], axis=1)
    np.testing.assert_array_equal(h, onp.histogram(x, bins=edges[0], axis=1)[0])


@pytest.mark.parametrize("resident", [False, True], ids=["host", "device"])
def test_two_weight_arrays_in_one_pass(xh, resident):
    """histogram_two_weights == two weighted histograms; device-resident float inputs share one pass"""
    rng = np.random.default_rng(56)
    conv = _dev if resident else (lambda a: a)

    def check(args, bins, wa, wb, axis=None, expect_fused=None):
        ha, hb, edges = xh.histogram_two_weights(*[conv(a) for a in args], bins=bins, weights=(conv(wa), conv(wb)), axis=axis)
        if resident:
            ha, hb = ha.cpu().numpy(), hb.cpu().numpy()
        ra, _ = onp.histogram(*args, bins=bins, weights=wa, axis=axis)
        rb, _ = onp.histogram(*args, bins=bins, weights=wb, axis=axis)
        assert ha.shape == ra.shape and hb.shape == rb.shape
        assert_hist_equal(ha, ra, True)
        assert_hist_equal(hb, rb, True)
        if expect_fused is not None and resident:
            plan = _plan_for(xh, [conv(a) for a in args], [np.asarray(e) for e in edges])
            assert ("weights=2" in plan.describe()) == expect_fused, plan.describe()

    x = rng.standard_normal(700_001)
    x[::1001] = np.nan
    a = rng.uniform(-2, 3, x.shape)
    w = rng.uniform(0, 1, x.shape)
    e = np.linspace(-4, 4, 101)
    check([x], e, a * w, w, expect_fused=True)                                   # the mean-in-bins idiom, C2 shape
    check([x.astype(np.float32)], e, (a * w).astype(np.float32), w.astype(np.float32), expect_fused=True)
    y = rng.standard_normal(x.shape)
    check([x, y], [np.linspace(-4, 4, 33), _nonuniform_edges(rng, 41)], a, w, expect_fused=True)
    t = rng.standard_normal((5, 7, 3001))
    wt = rng.uniform(0, 1, t.shape)
    check([t], np.linspace(-3, 3, 21), wt, wt * t, axis=2)                          # rows
    check([t], np.linspace(-3, 3, 21), wt, wt * t, axis=(1, 2))
    check([t], np.linspace(-3, 3, 21), wt[:1], wt, axis=0)                          # broadcast first weights, leading axis
    # no fused kernel: integer samples, a histogram beyond LDS, different weight dtypes -> two passes, same answer
    check([rng.integers(0, 50, 100_000).astype(np.int32)], np.arange(51), w[:100_000], a[:100_000], expect_fused=False)
    check([x[:200_000], y[:200_000]], [np.linspace(-4, 4, 301)] * 2, a[:200_000], w[:200_000], expect_fused=False)
    check([x[:100_000]], e, a[:100_000].astype(np.float32), w[:100_000], expect_fused=False)
    with pytest.raises(ValueError):
        xh.histogram_two_weights(conv(x), bins=e, weights=(conv(w),))


@pytest.mark.parametrize("dt", [np.float64, np.float32, np.int32])
def test_device_min_max_feeding_integer_bins(xh, dt):
    """bins=int without range on device data: the GPU min/max (vectorised for contiguous floats, generic
    otherwise) must give numpy's edges bit for bit, NaN-propagating like numpy"""
    from xhistogram_amd import _native

    rng = np.random.default_rng(55)
    for n in (1, 7, 4095, 4096, 4097, 1_000_003):
        x = (rng.standard_normal(n) * 100).astype(dt)
        if n > 10:
            x[-3] = 12345 if dt == np.int32 else 1e6   # maximum in the ragged tail
            x[1] = -54321 if dt == np.int32 else -1e7
        v = _native.make_view(_dev(x).data_ptr(), _native.dtype_tag(np.dtype(dt)), n, 1)
        t = _dev(x)
        v = _native.make_view(t.data_ptr(), _native.dtype_tag(np.dtype(dt)), n, 1)
        lo, hi = _native.minmax(v, 1, n, _native.MEM_DEVICE)
        assert lo == float(x.min()) and hi == float(x.max()), (n, lo, hi)
        h, e = xh.histogram(t, bins=13)
        hr, er = np.histogram(x, bins=13)
        np.testing.assert_array_equal(e[0], er)
        np.testing.assert_array_equal(h.cpu().numpy(), hr)
    if dt != np.int32:
        x = rng.standard_normal(100_000).astype(dt)
        x[99_999] = np.nan
        t = _dev(x)
        lo, hi = _native.minmax(_native.make_view(t.data_ptr(), _native.dtype_tag(np.dtype(dt)), x.size, 1), 1, x.size, _native.MEM_DEVICE)
        assert np.isnan(lo) and np.isnan(hi)
        x[99_999] = np.inf
        x[5] = -np.inf
        t = _dev(x)
        lo, hi = _native.minmax(_native.make_view(t.data_ptr(), _native.dtype_tag(np.dtype(dt)), x.size, 1), 1, x.size, _native.MEM_DEVICE)
        assert lo == -np.inf and hi == np.inf
        # a strided (non-contiguous) view takes the generic kernel
        s2 = _dev(rng.standard_normal((300, 40)).astype(dt))[:, ::3]
        lo, hi = _native.minmax(_native.make_view(s2.data_ptr(), _native.dtype_tag(np.dtype(dt)), s2.stride(0), s2.stride(1)), 300, s2.shape[1], _native.MEM_DEVICE)
        assert lo == float(s2.min()) and hi == float(s2.max())


def test_weights_that_span_a_small_slab_of_device_data(xh):
    """cos(lat) / cell-area weights against (time, lat, lon) on the GPU: only the (lat, lon) slab they span is
    written out and the weighted kernels read it with stride 0 along every other axis (core._weights_slab);
    reduced axes outside the slab are summed afterwards.  Must equal the reference's materialised-weights result."""
    rng = np.random.default_rng(154)
    t = rng.standard_normal((48, 36, 72)).astype(np.float32)
    t[0, 3, :5] = np.nan
    u = rng.standard_normal(t.shape).astype(np.float32)
    e = np.linspace(-3, 3, 31)
    w_lat = np.cos(np.linspace(-1.4, 1.4, 36)).reshape(1, 36, 1)
    w_area = rng.uniform(0, 2, (36, 72))
    w_lon = rng.uniform(0, 2, (72,)).astype(np.float32)
    w_time = rng.uniform(0, 2, (48, 1, 1))
    w_bad = w_lat.copy()
    w_bad[0, 5, 0] = np.nan
    w_bad[0, 9, 0] = np.inf
    cases = [(w_lat, (1, 2), True), (w_lat, None, True), (w_lat, (0, 1), True), (w_lat, 1, True), (w_lat, (0, 1, 2), True),
             (w_area, (1, 2), True), (w_area, None, True), (w_lon, (1, 2), True), (w_lon, 2, True), (w_lon, (0, 2), False),
             (w_bad, (1, 2), True), (w_time, (1, 2), False), (w_lat, 2, False), (w_lat, (0, 2), False),
             (rng.uniform(0, 1, t.shape), (1, 2), False), (np.float64(0.5) * np.ones((1, 1, 1)), None, False)]
    for w, axis, applies in cases:
        for args, bins in (([t], e), ([t, u], [np.linspace(-3, 3, 9), np.linspace(-3, 3, 7)])):
            want, _ = onp.histogram(*args, bins=bins, weights=w, axis=axis)
            got, _ = xh.histogram(*[_dev(a) for a in args], bins=bins, weights=_dev(w), axis=axis)
            assert got.shape == want.shape and got.dtype == torch.float64
            assert_hist_equal(got.cpu().numpy(), want, True)
        drop = list(range(3)) if axis is None else [axis] if isinstance(axis, int) else list(axis)
        part = xh._weights_slab([_dev(t)], _dev(w), drop, [e], None, "torch")
        assert (part is not None) == applies, (w.shape, axis)
    # weights that arrive broadcast already (a stride-0 view of full shape: what the xarray wrapper hands over)
    got, _ = xh.histogram(_dev(t), bins=e, weights=_dev(w_lat).expand(t.shape), axis=(1, 2), density=True)
    assert_hist_equal(got.cpu().numpy(), onp.histogram(t, bins=e, weights=w_lat, axis=(1, 2), density=True)[0], True)


@pytest.mark.parametrize("resident", [False, True], ids=["host", "device"])
def test_weights_broadcast_along_reduced_axes(xh, resident):
    """cos(lat)-style weights: constant along some reduced axes -> counted unweighted over those axes,
    weights applied to the counts; must equal the reference's materialised-weights result"""
    rng = np.random.default_rng(54)
    conv = _dev if resident else (lambda a: a)
    t = rng.standard_normal((6, 40, 700))
    t[0, 3, :5] = np.nan
    e = np.linspace(-3, 3, 31)

    def check(args, wts, axis, bins=e, density=False):
        want, _ = onp.histogram(*args, bins=bins, weights=wts, axis=axis, density=density)
        got, _ = xh.histogram(*[conv(a) for a in args], bins=bins, weights=conv(wts), axis=axis, density=density)
        got = got.cpu().numpy() if resident else got
        assert got.shape == want.shape and got.dtype == np.float64
        assert_hist_equal(got, want, True)

    w_lat = np.cos(np.linspace(-1.4, 1.4, 40)).reshape(1, 40, 1)
    for axis in ((1, 2), None, (2,), (0, 2), (0, 1, 2)):
        check([t], w_lat, axis)
    check([t], w_lat, (1, 2), density=True)
    check([t], rng.uniform(0, 2, (6, 1, 1)), (1, 2))          # one weight per kept row
    check([t], rng.uniform(0, 2, (700,)), (1, 2))             # varies along the LAST axis only: counts over the middle one
    check([t], np.float64(0.25) * np.ones((1, 1, 1)), None)   # a scalar weight
    check([t.astype(np.float32)], w_lat.astype(np.float32), (1, 2))
    check([t], np.broadcast_to(w_lat, t.shape), (1, 2))    # already broadcast: a stride-0 view of full shape
    if resident:
        got, _ = xh.histogram(_dev(t), bins=e, weights=_dev(w_lat).expand(t.shape), axis=(1, 2))
        assert_hist_equal(got.cpu().numpy(), onp.histogram(t, bins=e, weights=w_lat, axis=(1, 2))[0], True)
    u = rng.standard_normal(t.shape)
    check([t, u], w_lat, (1, 2), bins=[np.linspace(-3, 3, 9), np.linspace(-3, 3, 7)])
    # NaN / inf weights: only bins that received a sample of that weight are affected
    w_bad = w_lat.copy()
    w_bad[0, 5, 0] = np.nan
    w_bad[0, 9, 0] = np.inf
    narrow = t.copy()
    narrow[:, 5, :] = 0.1    # latitude 5 only ever hits one bin
    want, _ = onp.histogram(narrow, bins=e, weights=w_bad, axis=(1, 2))
    got, _ = xh.histogram(conv(narrow), bins=e, weights=conv(w_bad), axis=(1, 2))
    got = got.cpu().numpy() if resident else got
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_array_equal(np.isinf(got), np.isinf(want))
    ok = np.isfinite(want)
    np.testing.assert_allclose(got[ok], want[ok], rtol=1e-6)


@pytest.mark.parametrize("resident", [False, True], ids=["host", "device"])
def test_non_adjacent_reduced_axes_in_two_steps(xh, resident):
    """axis=(0, 2) of a 3-D array: histogram over the last block of adjacent axes, then sum the rest"""
    rng = np.random.default_rng(53)
    conv = _dev if resident else (lambda a: a)
    t = rng.standard_normal((9, 30, 700))
    q = rng.standard_normal((5, 8, 6, 300)).astype(np.float32)
    e = np.linspace(-3, 3, 21)
    for arr, axis in ((t, (0, 2)), (q, (0, 3)), (q, (1, 3)), (q, (0, 1, 3)), (q, (0, 2, 3))):
        want, _ = onp.histogram(arr, bins=e, axis=axis)
        got, _ = xh.histogram(conv(arr), bins=e, axis=axis)
        got = got.cpu().numpy() if resident else got
        assert got.dtype == np.int64
        np.testing.assert_array_equal(got, want)
        w = rng.uniform(0, 1, arr.shape)
        want, _ = onp.histogram(arr, bins=e, axis=axis, weights=w, density=True)
        got, _ = xh.histogram(conv(arr), bins=e, axis=axis, weights=conv(w), density=True)
        got = got.cpu().numpy() if resident else got
        assert_hist_equal(got, want, True)


@pytest.mark.parametrize("weighted", [False, True])
def test_int64_domain_vector_kernels(xh, weighted):
    """int64 / datetime64 samples against integer edges: exact int64 comparison on the vector kernels,
    including values float64 cannot tell apart"""
    rng = np.random.default_rng(51)
    base = (1 << 60) + 7
    for n in (1, 3, 1023, 200_003):
        x = base + rng.integers(-50, 250, (2, n))
        x[0, 0] = np.iinfo(np.int64).min
        x[-1, -1] = np.iinfo(np.int64).max
        edges = [base + np.sort(rng.choice(np.arange(-40, 240), 37, replace=False))]
        edges[0][-1] = edges[0][-2] + 1  # neighbours that are one apart
        w = rng.uniform(0, 1, x.shape) if weighted else None
        got, desc = _run(xh, [x], edges, w, True)
        assert "family=fast" in desc and "cmp=i64" in desc, desc
        assert_hist_equal(got, onp.bincount_rows([x], edges, w), weighted)
        assert_hist_equal(_run(xh, [x], edges, w, False)[0], onp.bincount_rows([x], edges, w), weighted)
    t0 = np.datetime64("1999-12-31T23:59:00", "s")
    t = (t0 + rng.integers(0, 200_000, (3, 50_001)).astype("timedelta64[s]")).astype("datetime64[s]")
    te = np.arange(np.datetime64("2000-01-01"), np.datetime64("2000-01-04"), np.timedelta64(6, "h"))
    w = rng.uniform(0, 1, t.shape) if weighted else None
    got, desc = _run(xh, [t], [te], w, False)
    assert "family=fast" in desc, desc
    assert_hist_equal(got, onp.bincount_rows([t], [te], w), weighted)


def test_small_integer_samples_with_integer_edges_take_the_vector_kernels(xh):
    """bins=np.arange(257) on uint8 / int16 / int32 data: exact in float64, so no int64 generic family"""
    rng = np.random.default_rng(59)
    for dt, lo, hi in ((np.uint8, 0, 256), (np.int16, -300, 300), (np.int32, -70000, 70000)):
        x = rng.integers(lo, hi, (1, 500_001)).astype(dt)
        edges = [np.arange(max(lo, -200), min(hi, 257) + 1)]  # int64 edges
        got, desc = _run(xh, [x], edges, None, True)
        assert "family=fast" in desc and "cmp=f64" in desc, desc
        np.testing.assert_array_equal(got, onp.bincount_rows([x], edges))
    # int64 samples keep the exact int64 comparison
    x = rng.integers(-5, 300, (1, 10_000)).astype(np.int64)
    got, desc = _run(xh, [x], [np.arange(257)], None, True)
    assert "cmp=i64" in desc, desc
    np.testing.assert_array_equal(got, onp.bincount_rows([x], [np.arange(257)]))


# ---------------------------------------------------------------------------------------------
# arithmetic edges (bins=int / np.linspace): table-free digitize when the edge tables would not fit
# ---------------------------------------------------------------------------------------------
def _edge_torture(edges, rng, n_random, dtype=np.float64):
    """samples that sit on, just below and just above every edge, plus the usual specials"""
    e = np.asarray(edges, dtype=np.float64)
    parts = [e, np.nextafter(e, -np.inf), np.nextafter(e, np.inf), rng.uniform(e[0] - 1, e[-1] + 1, n_random),
             np.array([np.nan, np.inf, -np.inf, e[0], e[-1], 0.0, -0.0])]
    if dtype == np.float32:  # float32 neighbours of the float32-rounded edges
        f = e.astype(np.float32)
        parts += [f.astype(np.float64), np.nextafter(f, np.float32(-np.inf)).astype(np.float64),
                  np.nextafter(f, np.float32(np.inf)).astype(np.float64)]
    x = np.concatenate(parts).astype(dtype)
    rng.shuffle(x)
    return x.reshape(1, -1)


@pytest.mark.parametrize("nb,weighted,dt,want_hist", [
    (30000, False, np.float64, "hist=lds"),       # 120 KB of uint32 counters; the edge table alone would be 240 KB
    (60000, False, np.float64, "hist=packed16"),  # uint16 counters, table-free
    (12000, True, np.float64, "hist=lds"),        # 96 KB of float64 sums + 96 KB of edges would not fit together
    (20000, False, np.float32, "hist=lds"),       # float32 samples compared in float64 (the tables would fit, with 3 edges per bucket)
    (15000, True, np.float32, "hist=lds"),        # 120 KB of float64 sums + 60 KB of float32 thresholds would not fit
])
def test_arithmetic_edges_large_1d(xh, nb, weighted, dt, want_hist):
    rng = np.random.default_rng(61 + nb)
    edges = [np.linspace(-3.7, 5.1, nb + 1)]
    x = _edge_torture(edges[0], rng, 400_000, dt)
    w = rng.uniform(-1, 2, x.shape) if weighted else None
    got, desc = _run(xh, [x], edges, w, True, arith32=-1)
    assert "family=fast" in desc and "scan=5" in desc and want_hist in desc, desc
    assert_hist_equal(got, onp.bincount_rows([x], edges, w), weighted)
    if dt == np.float32:  # left alone, float32 samples on these edges are decided in float32 arithmetic (round 5)
        got, desc = _run(xh, [x], edges, w, True)
        assert "family=fast" in desc and "scan=9" in desc and want_hist in desc, desc
        assert_hist_equal(got, onp.bincount_rows([x], edges, w), weighted)
    got, desc = _run(xh, [x], edges, w, False)  # host route
    assert_hist_equal(got, onp.bincount_rows([x], edges, w), weighted)


@pytest.mark.parametrize("lo,hi", [(-4.0, 4.0), (0.0, 1.0), (-1e-300, 3e-300), (-1e300, 1e300), (1e6, 1e6 + 1.0),
                                   (-123.456, -123.0), (0.1, 0.7), (2.0 ** -1074 * 64, 2.0 ** -1074 * 64000)])
@pytest.mark.parametrize("nb", [1, 2, 7, 100, 1001])
def test_arithmetic_digitize_forced_small_bin_counts(xh, lo, hi, nb):
    """the table-free digitize on shapes where it is not the automatic choice: every edge, its two
    neighbours, out-of-range values and specials, for several magnitudes of e_0 and step"""
    rng = np.random.default_rng(nb)
    edges = [np.linspace(lo, hi, nb + 1)]
    if np.any(np.diff(edges[0]) < 0):
        pytest.skip("np.linspace itself is not monotone here (subnormal step)")
    x = _edge_torture(edges[0], rng, 0)
    x = np.concatenate([x, rng.uniform(lo - (hi - lo) * 0.1, hi + (hi - lo) * 0.1, (1, 5000))], axis=1)
    want = onp.bincount_rows([x], edges)
    got, desc = _run(xh, [x], edges, None, True, arith=1)
    np.testing.assert_array_equal(got, want, err_msg=desc)
    if (lo, hi) == (-4.0, 4.0):
        assert "scan=5" in desc, desc
    w = rng.uniform(0, 1, x.shape)
    assert_hist_equal(_run(xh, [x], edges, w, True, arith=1)[0], onp.bincount_rows([x], edges, w), True)


def test_arithmetic_digitize_forced_2d_3d_f32(xh):
    rng = np.random.default_rng(71)
    n = 200_000
    e2 = [np.linspace(-3, 3, 41), np.linspace(0, 10, 1001)]
    x = rng.standard_normal((1, n))
    y = _edge_torture(e2[1], rng, n - 3 * 1001 - 7)
    got, desc = _run(xh, [x, y], e2, None, True, arith=1)
    assert "scan=5" in desc, desc
    np.testing.assert_array_equal(got, onp.bincount_rows([x, y], e2))
    e3 = [np.linspace(-3, 3, 13), np.linspace(-3, 3, 9), np.linspace(-3, 3, 17)]
    s3 = [rng.standard_normal((4, n // 4)).astype(np.float32) for _ in range(3)]
    w = rng.uniform(0, 1, s3[0].shape).astype(np.float32)
    got, desc = _run(xh, s3, e3, w, True, arith=1)
    assert "scan=5" in desc, desc
    assert_hist_equal(got, onp.bincount_rows(s3, e3, w), True)


def test_arithmetic_edges_partitioned_count_pass(xh):
    """60000 weighted bins: histogram and edge tables both beyond LDS -> partitioned, table-free count pass"""
    rng = np.random.default_rng(67)
    edges = [np.linspace(0.0, 1.0, 60001)]
    x = _edge_torture(edges[0], rng, 600_000)
    w = rng.uniform(0, 1, x.shape)
    got, desc = _run(xh, [x], edges, w, True, partition=1)
    assert "hist=partitioned" in desc and "scan=5" in desc, desc
    assert_hist_equal(got, onp.bincount_rows([x], edges, w), True)


def test_arithmetic_edges_not_assumed(xh):
    """edges that only LOOK uniform keep the table path (and stay exact)"""
    rng = np.random.default_rng(68)
    cases = {
        "cumsum": np.cumsum(np.full(30001, 0.1)) - 7.0,            # accumulated rounding: not j*step + e_0
        "perturbed": np.linspace(-4, 4, 30001),
        "coarse_ulp": np.linspace(1e15, 1e15 + 3000.0, 30001),     # step 0.1 vs ulp 0.125: not resolved
    }
    cases["perturbed"] = cases["perturbed"].copy()
    cases["perturbed"][12345] = np.nextafter(cases["perturbed"][12345], np.inf)
    for name, e in cases.items():
        e = np.maximum.accumulate(e)
        x = _edge_torture(e, rng, 50_000)
        got, desc = _run(xh, [x], [e], None, True)
        assert "scan=5" not in desc, (name, desc)
        np.testing.assert_array_equal(got, onp.bincount_rows([x], [e]), err_msg=name)


def test_arithmetic_edges_2d_beyond_lds_tables(xh):
    """two inputs whose edge tables (2 x 6000 edges = 96 KB + bucket tables) crowd the histogram out of LDS"""
    rng = np.random.default_rng(69)
    edges = [np.linspace(-2, 2, 9), np.linspace(-4, 4, 6001)]
    n = 300_000
    x = rng.standard_normal((1, n))
    y = _edge_torture(edges[1], rng, n - 3 * 6001 - 7)
    assert y.shape == x.shape
    got, desc = _run(xh, [x, y], edges, None, True)
    assert "family=fast" in desc, desc
    np.testing.assert_array_equal(got, onp.bincount_rows([x, y], edges))


def _route_tile(desc, sample_bytes, weight_bytes, dims, forced_spl):
    """tile length the routing pass must report: 1024 x 8 samples only where route_long_tile_ok (csrc/xhist_pick.hip.h)
    lets the long-tile kernel exist, else 1024 x 4 — whatever "route_spl" asked for"""
    import re

    arith = int(re.search(r"scan=(\d+)", desc).group(1)) == 5
    long_ok = 2 * (dims * sample_bytes + weight_bytes) <= (32 if arith else 24) and not (sample_bytes == 8 and weight_bytes == 8)
    return 1024 * (8 if (long_ok and forced_spl != 4) else 4)


@pytest.mark.parametrize("block,spl", [(1024, 4), (1024, 8)])
@pytest.mark.parametrize("lo,hi,nb", [(-4.0, 4.0, 1024), (0.0, 1.0, 1500), (-1e-300, 3e-300, 1100), (-1e300, 1e300, 1200),
                                      (1e6, 1e6 + 1.0, 1030), (-123.456, -123.0, 2000), (0.1, 0.7, 1111)])
def test_partitioned_mode_arithmetic_digitize_on_and_next_to_every_edge(xh, lo, hi, nb, block, spl):
    """The routing pass decides bins of arithmetic edges by arithmetic alone unless a sample is within delta bins of an
    edge (bin_arith_fast, xhist_kernels.hip.h): a sample on every edge and on both floating-point neighbours of it, in
    both dimensions, must land where searchsorted puts it (core.py:163-174) — for several magnitudes of e_0 and step and
    every workgroup size and tile length of the pass."""
    rng = np.random.default_rng(nb + block + spl)
    edges = [np.linspace(lo, hi, nb + 1), np.linspace(-2.0, 6.0, 1025)]
    x = _edge_torture(edges[0], rng, 20_000)
    y = _edge_torture(edges[1], rng, x.shape[1] - 3 * 1025 - 7)
    assert x.shape == y.shape
    rng.shuffle(y[0])
    want = onp.bincount_rows([x, y], edges)
    got, desc = _run(xh, [x, y], edges, None, True, partition=1, arith=1, route_spl=spl)
    assert "hist=partitioned" in desc and "route=fused" in desc and "scan=5" in desc, desc
    assert "tile=%d block=%d " % (block * spl, block) in desc, desc
    np.testing.assert_array_equal(got, want, err_msg=desc)
    w = rng.uniform(0.5, 1.5, x.shape)
    got, desc = _run(xh, [y, x], edges[::-1], w, True, partition=1, arith=1, route_spl=spl)
    assert "scan=5" in desc and "tile=%d block=%d " % (block * 4, block) in desc, desc  # (float64 + float64 weights: never the long tile)
    assert_hist_equal(got, onp.bincount_rows([y, x], edges[::-1], w), True)


@pytest.mark.parametrize("geometry", ["auto", (1024, 4), (1024, 8)], ids=str)
@pytest.mark.parametrize("edges_kind", ["linspace", "uneven"])
@pytest.mark.parametrize("weights", ["none", "f32", "f64_one_sign", "f64_both_signs"])
@pytest.mark.parametrize("dims,dtype", [(1, np.float64), (2, np.float64), (3, np.float64), (1, np.float32), (2, np.float32), (3, np.float32)])
def test_partitioned_mode_routing_geometries(xh, dims, dtype, weights, edges_kind, geometry):
    """The routing pass exists per tile length (route_geom_for, xhist_exec_device.hip.h: 1024 threads x 8 samples where the
    registers allow, else 1024 x 4): both, forced where they exist, and the automatic choice
    give the reference's histogram for every dtype combination — ragged tail, NaNs, right-edge samples and a tile-sized
    run of one value included."""
    rng = np.random.default_rng(dims * 7 + len(weights))
    n = 300_000 + 8191 + 3
    nb = {1: 200_000, 2: 300, 3: 48}[dims]
    e = np.linspace(-4, 4, nb + 1)
    if edges_kind == "uneven":
        e = np.sort(e + rng.uniform(-0.4, 0.4, e.size) * (e[1] - e[0]))
        if dims == 1:
            e = e[:: 20]  # (tables of uneven edges have to fit the LDS for the fast family: 10^4 bins, not partitioned)
    edges = [e] * dims
    xs = [rng.standard_normal((1, n)).astype(dtype) * (1.0 + 0.3 * d) for d in range(dims)]
    xs[0][0, ::499] = np.nan
    if float(dtype(e[-1])) == e[-1]:
        xs[-1][0, 7::1013] = e[-1]  # right edge of the last bin
    for x in xs:
        x[0, 100_000:100_000 + 9000] = 0.25  # more than a tile of one (bin, partition)
    w = {"none": None, "f32": rng.uniform(0, 1, (1, n)).astype(np.float32), "f64_one_sign": rng.uniform(0, 1, (1, n)),
         "f64_both_signs": rng.standard_normal((1, n))}[weights]
    params = {} if geometry == "auto" else {"route_spl": geometry[1]}
    want = onp.bincount_rows(xs, edges, w)
    got, desc = _run(xh, xs, edges, w, True, partition=1, **params)
    assert "hist=partitioned" in desc or (dims == 1 and edges_kind == "uneven"), desc
    if geometry != "auto" and "route=fused" in desc and "scan=7" not in desc and "scan=8" not in desc:  # (packed entries may trade the long tile for LDS)
        wb = 0 if w is None else w.dtype.itemsize
        tile = _route_tile(desc, np.dtype(dtype).itemsize, wb, dims, geometry[1])
        assert "tile=%d block=1024 " % tile in desc or (geometry[1] == 8 and "tile=4096 block=1024 " in desc), desc
    assert_hist_equal(got, want, w is not None)


@pytest.mark.parametrize("min_parts", [0, 1, 5, 64, 128])
@pytest.mark.parametrize("rows", [1, 3, 12])
@pytest.mark.parametrize("weights", ["none", "f32", "f64"])
def test_partitioned_mode_finer_partitions(xh, weights, rows, min_parts):
    """A histogram of a few LDS capacities is cut into 16+ partitions of 2^11 ... 2^15 bins instead of the two or three
    that its size asks for (part_geometry, xhist_exec_device.hip.h: a handful of partitions serialises the routing pass's
    rank counters): every partition size gives the reference's result, with one row and with rows sharing a pass."""
    rng = np.random.default_rng(rows * 10 + len(weights))
    n = 150_001
    edges = [np.linspace(-3, 3, 541), np.linspace(-2, 2, 261) ** 3]
    x, y = rng.standard_normal((rows, n)), rng.standard_normal((rows, n)) * 2.0
    x[:, ::311] = np.nan
    w = {"none": None, "f32": rng.uniform(0, 1, (rows, n)).astype(np.float32), "f64": rng.uniform(0, 1, (rows, n))}[weights]
    want = onp.bincount_rows([x, y], edges, w)
    got, desc = _run(xh, [x, y], edges, w, True, partition=1, min_parts=min_parts)
    assert "hist=partitioned" in desc, desc
    per_part = int(desc.split("bins_per_part=")[1].split()[0])
    n_bins = 540 * 260
    if min_parts == 1:
        assert per_part == (1 << 14 if w is not None else 1 << 15), desc
    elif min_parts == 128:
        # never finer than 2^11 bins per partition; rows that fit one pass with 8+ partitions each are left in it
        assert per_part == 1 << 11 or rows > 1, desc
    elif min_parts == 0 and weights != "f64":
        assert -(-n_bins // per_part) >= 8 and (rows > 1 or per_part < (1 << 14 if w is not None else 1 << 15)), desc
    assert_hist_equal(got, want, w is not None)


def test_partitioned_mode_small_float64_weighted_calls_take_three_passes(xh):
    """left to itself, a single row of fewer than 1.5 x 10^7 float64 samples with float64 weights goes through count + prefix +
    scatter (9-15 % faster there than the routing pass, whose fixed costs show); more samples, other dtypes, or "fused" = 1
    take the one-pass route.  Same histogram either way."""
    rng = np.random.default_rng(72)
    n = 4_500_000
    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    x, y = rng.standard_normal((1, n)), rng.standard_normal((1, n))
    w = rng.uniform(0, 1, (1, n))
    want = onp.bincount_rows([x, y], edges, w)
    got, desc = _run(xh, [x, y], edges, w, True)
    assert "hist=partitioned" in desc and "route=fused" not in desc, desc
    assert_hist_equal(got, want, True)
    got, desc = _run(xh, [x, y], edges, w, True, fused=1)
    assert "hist=partitioned" in desc and "route=fused" in desc, desc
    assert_hist_equal(got, want, True)
    got, desc = _run(xh, [x.astype(np.float32), y.astype(np.float32)], edges, w.astype(np.float32), True)
    assert "route=fused" in desc, desc
    assert_hist_equal(got, onp.bincount_rows([x.astype(np.float32), y.astype(np.float32)], edges, w.astype(np.float32)), True)


@pytest.mark.parametrize("weighted", [False, True])
def test_partitioned_mode_finer_partitions_three_pass_route(xh, weighted):
    """the count + prefix + scatter form of the partitioned mode ("fused" = -1) takes the finer partitions as well"""
    rng = np.random.default_rng(71)
    n = 400_003
    edges = [np.linspace(-3, 3, 541), np.linspace(-2, 2, 261)]
    x, y = rng.standard_normal((1, n)), rng.standard_normal((1, n))
    w = rng.uniform(0, 1, (1, n)) if weighted else None
    want = onp.bincount_rows([x, y], edges, w)
    for mp in (1, 0, 64):
        got, desc = _run(xh, [x, y], edges, w, True, partition=1, fused=-1, min_parts=mp)
        assert "hist=partitioned" in desc and "route=fused" not in desc, desc
        assert_hist_equal(got, want, weighted)


@pytest.mark.parametrize("weights", ["none", "one_sign", "both_signs", "f32"])
@pytest.mark.parametrize("pct", [2, 10, 40])
def test_partitioned_mode_chunk_pool_runs_dry(xh, weights, pct):
    """VERDICT r2 "weak" #5: the routing pass no longer traps when its chunk pool is exhausted — what finds no chunk is
    added to the output directly (exact); a packed-record pass reports "both signs" instead, so the exact pass redoes the
    call.  The pool is cut to a few percent of its size to force the path."""
    rng = np.random.default_rng(91 + pct)
    n = 3_000_000
    x, y = rng.standard_normal((1, n)), rng.standard_normal((1, n))
    x[0, ::977] = np.nan
    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    w = {"none": None, "one_sign": rng.uniform(0, 1, (1, n)), "both_signs": rng.standard_normal((1, n)),
         "f32": rng.uniform(0, 1, (1, n)).astype(np.float32)}[weights]
    want = onp.bincount_rows([x, y], edges, w)
    got, desc = _run(xh, [x, y], edges, w, True, partition=1, route_pool_pct=pct, records48=0)
    assert "hist=partitioned" in desc and "route=fused" in desc, desc
    assert_hist_equal(got, want, w is not None)
    plan = _plan_for(xh, [_dev(x), _dev(y)], edges)
    torch.cuda.synchronize()
    if pct <= 10:  # (40 % of the worst-case size still holds what 3*10^6 samples need: the path is not forced there)
        assert "pool_dry=1" in plan.describe(), plan.describe()
    # and with the pool back at full size the same plan runs clean again
    plan.set_param("route_pool_pct", 100)
    got, desc = _run(xh, [x, y], edges, w, True, partition=1)
    assert_hist_equal(got, want, w is not None)
    torch.cuda.synchronize()
    assert "pool_dry=1" not in plan.describe(), plan.describe()
    plan.set_param("route_pool_pct", 0)


@pytest.mark.parametrize("weighted", [False, True])
def test_partitioned_mode_chunk_pool_runs_dry_with_several_rows_per_pass(xh, weighted):
    """the pool-dry path addresses the output by (row, partition): several rows in one routing pass, pool cut to 3 %"""
    rng = np.random.default_rng(64)
    rows, n = 9, 90_001
    x, y = rng.standard_normal((rows, n)), rng.standard_normal((rows, n)) * 1.2
    edges = [np.linspace(-3, 3, 301), np.linspace(-3, 3, 301)]
    w = rng.uniform(0, 1, (rows, n)) if weighted else None
    got, desc = _run(xh, [x, y], edges, w, True, partition=1, route_pool_pct=3)
    assert "hist=partitioned" in desc and "rows_per_pass=1 " not in desc, desc
    assert_hist_equal(got, onp.bincount_rows([x, y], edges, w), weighted)
    torch.cuda.synchronize()
    plan = _plan_for(xh, [_dev(x), _dev(y)], edges)
    assert "pool_dry=1" in plan.describe(), plan.describe()
    plan.set_param("route_pool_pct", 100)
    plan.set_param("route_pool_pct", 0)


@pytest.mark.parametrize("dt", [np.float64, np.float32, np.int32])
@pytest.mark.parametrize("name", ["sqrt", "sturges", "rice", "scott"])
def test_bin_estimators_on_device_resident_data_without_a_host_copy(xh, name, dt, monkeypatch):
    """f-1 (VERDICT r2 "next" #7): np.histogram_bin_edges' cheap estimators from ONE reduction on the GPU (xhist_moments):
    the edges are bit-identical to numpy's (core.py:383-388) and the tensor never visits the host"""
    rng = np.random.default_rng(7)
    a = (rng.standard_normal((3, 50_001)) * 3 + 1).astype(dt) if np.dtype(dt).kind == "f" else rng.integers(-40, 90, (3, 50_001)).astype(dt)
    t = _dev(a)
    monkeypatch.setattr(torch.Tensor, "cpu", lambda self, *args, **kw: (_ for _ in ()).throw(AssertionError("the data went to the host")))
    for r in (None, (-2.0, 5.5), (0, 30)):
        want = np.histogram_bin_edges(a, bins=name, range=r)
        got = xh._device_bin_edges(t, name, r, False)
        assert got.dtype == want.dtype
        np.testing.assert_array_equal(got, want, err_msg=str((name, dt, r)))
    tv = t[:, ::3]  # a strided view: the reduction takes strides
    np.testing.assert_array_equal(xh._device_bin_edges(tv, name, None, False), np.histogram_bin_edges(a[:, ::3], bins=name))
    monkeypatch.undo()
    h, edges = xh.histogram(t, bins=name)
    want_h, want_e = np.histogram(a, bins=name)
    np.testing.assert_array_equal(edges[0] if isinstance(edges, (list, tuple)) else edges, want_e)
    np.testing.assert_array_equal(h.cpu().numpy(), want_h)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("name", ["fd", "auto"])
def test_quartile_estimators_on_device_without_a_host_copy(xh, name, dt):
    """"fd" and "auto" need the data only through n, min, max and the quartiles (numpy/lib/_histograms_impl.py): four order
    statistics, found exactly on the GPU by narrowing histograms (core._device_order_statistics), and np.percentile's own
    interpolation — edges bit-identical to numpy's for sizes from 1 element up, ties, constant data, heavy tails, values
    packed into a few ulps; the tensor itself never goes to the host (only counts and the last interval's few values do)."""
    rng = np.random.default_rng(17)
    cases = {
        "one": np.array([2.5]), "two": np.array([1.0, 4.0]), "three": np.array([3.0, -1.0, 8.0]), "ten": rng.standard_normal(10),
        "normal": rng.standard_normal(100_001) * 3 + 1, "uniform": rng.uniform(-5, 9, 250_000),
        "ties": np.round(rng.standard_normal(300_000) * 4), "constant": np.full(5000, 1.25),
        "mostly_constant": np.where(rng.random(200_000) < 0.9, 2.0, rng.standard_normal(200_000)),
        "lognormal": rng.lognormal(0, 3, 400_000), "near_1e6": 1e6 + rng.standard_normal(150_000) * 1e-3,
        "few_ulps": 1.0 + rng.integers(0, 7, 100_000) * np.finfo(dt).eps, "big": rng.standard_normal(3_000_000),
        "two_d": rng.standard_normal((300, 1001)),
    }
    for label, a in cases.items():
        a = a.astype(dt)
        t = _dev(a)
        try:
            want = np.histogram_bin_edges(a, bins=name)
        except ValueError as err:  # ("Too many bins for data range": values a few ulps apart) — the same error, then
            with pytest.raises(ValueError, match=str(err)[:20]):
                xh._device_quartile_edges(t, name, None, np.dtype(dt), False)
            continue
        n_before = t.numel()
        got = xh._device_quartile_edges(t, name, None, np.dtype(dt), False)
        assert got is not None, label
        assert got.dtype == want.dtype, label
        np.testing.assert_array_equal(got, want, err_msg=label)
        assert t.numel() == n_before
    # the public entry point takes it; ranges, integer data and DeviceArrays still go through numpy on a host copy
    a = cases["normal"].astype(dt)
    h, edges = xh.histogram(_dev(a), bins=name)
    want_h, want_e = np.histogram(a, bins=name)
    np.testing.assert_array_equal(edges[0] if isinstance(edges, (list, tuple)) else edges, want_e)
    np.testing.assert_array_equal(h.cpu().numpy(), want_h)
    for r in ((-1.0, 1.0), (0, 30), (2.0, 2.0), (50.0, 60.0), (-100, 100)):  # a range: the selector sees the data cut to it (possibly none of it)
        got = xh._device_quartile_edges(_dev(a), name, r, np.dtype(dt), False)
        assert got is not None, r
        np.testing.assert_array_equal(got, np.histogram_bin_edges(a, bins=name, range=r), err_msg=str(r))
    np.testing.assert_array_equal(xh._device_bin_edges(_dev(a), name, (-1.0, 1.0), False), np.histogram_bin_edges(a, bins=name, range=(-1.0, 1.0)))
    with pytest.raises(ValueError):
        xh._device_quartile_edges(_dev(a), name, (3.0, 1.0), np.dtype(dt), False)  # numpy's "max must be larger than min"
    with pytest.raises(ValueError):
        xh._device_bin_edges(_dev(np.array([1.0, np.nan, 2.0], dtype=dt)), name, None, False)  # numpy's own error for non-finite data


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_doane_and_stone_estimators_on_device(xh, dt):
    """"doane" from float64 reductions on the device (declining when the bin count hangs on numpy's own summation order),
    "stone" from this library's histograms of 1 ... max(100, sqrt(n)) uniform bins (exact counts: numpy's very numbers)."""
    import warnings

    rng = np.random.default_rng(19)
    cases = {"one": np.array([2.5]), "two": np.array([1.0, 4.0]), "three": np.array([3.0, -1.0, 8.0]), "normal": rng.standard_normal(60_001) * 3 + 1,
             "skewed": rng.lognormal(0, 1, 80_000), "uniform": rng.uniform(-5, 9, 50_000), "ties": np.round(rng.standard_normal(40_000) * 4),
             "constant": np.full(3000, 1.25), "bimodal": np.concatenate([rng.normal(-3, 0.5, 30_000), rng.normal(4, 1.5, 20_000)])}
    declined = 0
    for label, a in cases.items():
        a = a.astype(dt)
        for name in ("doane", "stone"):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                want = np.histogram_bin_edges(a, bins=name)
                got = xh._device_doane_stone_edges(_dev(a), name, None, np.dtype(dt), False)
            if got is None:  # (allowed for "doane" only: a tie, or nearly constant data)
                assert name == "doane", label
                declined += 1
                continue
            assert got.dtype == want.dtype, (label, name)
            np.testing.assert_array_equal(got, want, err_msg="%s %s" % (label, name))
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                np.testing.assert_array_equal(xh._device_bin_edges(_dev(a), name, None, False), want)
    assert declined <= 2
    a = cases["skewed"].astype(dt)
    for r in ((0.5, 3.0), (0, 30), (2.0, 2.0), (500.0, 600.0)):  # a range: the selector sees the data cut to it (possibly none of it)
        for name in ("doane", "stone"):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                want = np.histogram_bin_edges(a, bins=name, range=r)
                got = xh._device_doane_stone_edges(_dev(a), name, r, np.dtype(dt), False)
            if got is None:
                assert name == "doane", r
                continue
            np.testing.assert_array_equal(got, want, err_msg="%s %s" % (r, name))
    big = _dev(rng.standard_normal(20_000_000).astype(dt))
    assert xh._device_doane_stone_edges(big, "stone", None, np.dtype(dt), False) is None  # 4472 candidates: left to numpy


@pytest.mark.parametrize("dt", [np.int32, np.int64, np.uint8, np.int16])
@pytest.mark.parametrize("name", ["fd", "auto"])
def test_quartile_estimators_of_integer_data_on_device(xh, name, dt):
    _integer_estimators(xh, name, dt, xh._device_quartile_edges)


@pytest.mark.parametrize("dt", [np.int32, np.int64, np.uint8])
@pytest.mark.parametrize("name", ["doane", "stone"])
def test_doane_and_stone_of_integer_data_on_device(xh, name, dt):
    _integer_estimators(xh, name, dt, xh._device_doane_stone_edges)


def _integer_estimators(xh, name, dt, fn):
    """integer data: order statistics by the same search (float64 edges tell integers below 2^53 apart), np.percentile's
    interpolation in float64, numpy's width >= 1 rule for integer data"""
    rng = np.random.default_rng(21)
    lo, hi = (0, 250) if dt == np.uint8 else (-3000, 9000)
    for a in (rng.integers(lo, hi, 100_003), rng.integers(lo, lo + 3, 5000), np.array([7, 7, 7, 7]), rng.integers(lo, hi, 11), np.array([5])):
        a = a.astype(dt)
        for r in (None, (lo + 10, hi - 20)):
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                want = np.histogram_bin_edges(a, bins=name, range=r)
                got = fn(_dev(a), name, r, np.dtype(dt), False)
            if got is None and name == "doane":
                continue  # (a tie at the ceil, or nearly constant data: numpy decides)
            assert got is not None and got.dtype == want.dtype, (len(a), r)
            np.testing.assert_array_equal(got, want, err_msg=str((len(a), r)))


def test_order_statistics_on_device(xh):
    """every rank of small arrays and scattered ranks of big ones, against np.sort"""
    rng = np.random.default_rng(18)
    for dt in (np.float64, np.float32):
        for a in (rng.standard_normal(1000), np.round(rng.standard_normal(5000) * 2), rng.lognormal(0, 5, 200_000), rng.standard_normal(2_000_000)):
            a = a.astype(dt)
            srt = np.sort(a)
            ranks = list(range(len(a))) if len(a) <= 5000 else [0, 1, 17, len(a) // 4, len(a) // 2, len(a) - 2, len(a) - 1] + list(rng.integers(0, len(a), 20))
            if len(ranks) > 60:
                ranks = ranks[::len(ranks) // 60]
            got = xh._device_order_statistics(_dev(a).reshape(-1), ranks, float(srt[0]), float(srt[-1]), len(a))
            assert got is not None
            for r in ranks:
                assert got[int(r)] == float(srt[int(r)]), (dt, len(a), r)


def test_bin_estimators_that_need_the_data_still_work(xh):
    """"doane" / "stone" (third moments, a search over bin counts) — and "fd" / "auto" of integer data: numpy's implementation on a
    host copy, same edges"""
    rng = np.random.default_rng(8)
    a = rng.standard_normal(20_000)
    ah = rng.standard_normal(5000).astype(np.float16)
    for name in ("fd", "doane"):  # float16: numpy on a host copy
        np.testing.assert_array_equal(xh._device_bin_edges(_dev(ah), name, None, False), np.histogram_bin_edges(ah, bins=name))
    for name in ("fd", "auto", "doane", "stone"):
        np.testing.assert_array_equal(xh._device_bin_edges(_dev(a), name, None, False), np.histogram_bin_edges(a, bins=name))
    with pytest.raises(TypeError):
        xh._device_bin_edges(_dev(a), "sqrt", None, True)  # weighted data: numpy's own TypeError


def test_scratch_cache_shrinks_to_what_recent_calls_use(xh):
    """VERDICT r2 "weak" #8: the library's caching allocator keeps what the largest RECENT call held (so C5's record streams
    are allocated once), not a fixed half of the device: after a call that needed hundreds of MB, a run of small calls hands
    the memory back to the driver (a torch process sharing the GPU gets it)."""
    from xhistogram_amd import _native

    if os.environ.get("XHIST_AMD_POOL_KEEP_GB"):
        pytest.skip("the keep limit is fixed by the environment")
    rng = np.random.default_rng(5)
    n = 40_000_000
    x, y, w = (torch.as_tensor(rng.standard_normal(n)).cuda() for _ in range(3))
    w = w.abs()
    edges = [np.linspace(-4, 4, 1025)] * 2
    small = rng.standard_normal(20_000)
    e1 = np.linspace(-4, 4, 33)
    h, _ = xh.histogram(x, y, bins=edges, weights=w)
    torch.cuda.synchronize()
    big = _native.scratch_stats(0)
    assert big["cached"] > 200 << 20 and big["limit"] >= big["cached"], big  # the record streams of 4*10^7 samples stay cached ...
    h2, _ = xh.histogram(x, y, bins=edges, weights=w)
    torch.cuda.synchronize()
    again = _native.scratch_stats(0)
    assert again["cached"] <= big["cached"] + (8 << 20), (big, again)  # ... and serve the next such call: nothing new is allocated
    for _ in range(700):  # host-route calls stage through a few hundred KB of scratch each: more than two windows of 256 frees
        xh.histogram(small, bins=e1)
    torch.cuda.synchronize()
    after = _native.scratch_stats(0)
    assert after["limit"] == 64 << 20 and after["cached"] <= 64 << 20, (big, after)
    np.testing.assert_allclose(h.cpu().numpy(), h2.cpu().numpy(), rtol=1e-9)


# ---------------------------------------------------------------------------------------------
# row-per-lane kernels: leading-axis reductions and many short rows
# ---------------------------------------------------------------------------------------------
def _describe_last(xh, samples, edges):
    return _plan_for(xh, samples, edges).describe()


@pytest.mark.parametrize("dt", [np.float32, np.float64])
@pytest.mark.parametrize("weighted", [False, True])
def test_lanes_leading_axis_reduction(xh, dt, weighted):
    rng = np.random.default_rng(51)
    t = (rng.standard_normal((301, 37, 53)) * 1.5).astype(dt)  # rows = 37*53 = 1961 (not a multiple of 256)
    t[5, 3, 7] = np.nan
    t[:, 0, 0] = 4.0
    w = rng.uniform(0, 1, t.shape) if weighted else None
    edges = np.linspace(-4, 4, 51)
    want, _ = onp.histogram(t, bins=edges, axis=0, weights=w)
    got, _ = xh.histogram(_dev(t), bins=edges, axis=0, weights=None if w is None else _dev(w))
    desc = _describe_last(xh, [_dev(t[0])], [edges])
    assert "family=lanes" in desc and "transpose=0" in desc, desc
    assert_hist_equal(got.cpu().numpy(), want, weighted)
    # weights broadcast along the reduced axis (one weight per kept position) and along kept axes
    if weighted:
        for wshape in ((1, 37, 53), (301, 1, 1)):
            wb = rng.uniform(0, 1, wshape)
            want, _ = onp.histogram(t, bins=edges, axis=0, weights=wb)
            got, _ = xh.histogram(_dev(t), bins=edges, axis=0, weights=_dev(wb))
            assert_hist_equal(got.cpu().numpy(), want, True)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("weights", [None, "f32", "f64"])
@pytest.mark.parametrize("shape,axis", [((300, 40, 50), 0), ((700, 33, 9), 0), ((130, 5000), 0), ((40, 90, 37), 1), ((2000, 3, 5), 0),
                                        ((64, 20, 1030), (0, 1))])
def test_lanes_leading_axis_with_more_bins_than_lane_private_columns_hold(xh, shape, axis, weights, dtype):
    """Reductions over a leading axis give every ROW (a grid point) one LDS counter column; with a column per LANE the
    histogram is n_bins x 257 words, which a joint histogram (20 x 23 bins) or a few hundred weighted bins do not fit — the
    generic family took those (7.8 ms for (1825, 360, 720) float32 pairs over `time`).  Now the lane groups of a row share
    its column and the workgroup takes fewer rows (execute_lanes: "shared")."""
    rng = np.random.default_rng(len(shape) * 7 + shape[0])
    xs = [rng.standard_normal(shape).astype(dtype) for _ in range(2)]
    xs[0][::7] = np.nan
    xs[1][..., ::5] = 100.0
    w = None if weights is None else rng.uniform(0, 1, shape).astype(np.float32 if weights == "f32" else np.float64)
    bins = [np.linspace(-3, 3, 21), np.sort(rng.uniform(-3, 3, 24))]
    want = onp.histogram(*xs, bins=bins, axis=axis, weights=w)[0]
    got, _ = xh.histogram(*[_dev(a) for a in xs], bins=bins, axis=axis, weights=None if w is None else _dev(w))
    desc = _plan_for(xh, [_dev(a) for a in xs], bins).describe()
    assert "family=lanes" in desc and "(shared)" in desc, desc
    assert_hist_equal(got.cpu().numpy(), want, w is not None)
    # one input, 300 weighted bins: 617 KB of lane-private float64 columns
    e1 = np.linspace(-3, 3, 301)
    w1 = rng.uniform(0, 1, shape)
    want = onp.histogram(xs[0], bins=e1, axis=axis, weights=w1)[0]
    got, _ = xh.histogram(_dev(xs[0]), bins=e1, axis=axis, weights=_dev(w1))
    desc = _plan_for(xh, [_dev(xs[0])], [e1]).describe()
    assert "family=lanes" in desc and "(shared)" in desc, desc
    assert_hist_equal(got.cpu().numpy(), want, True)


@pytest.mark.parametrize("shape", [(5000, 20), (4097, 365), (70_000, 7), (4096, 384), (300_001, 33)])
def test_lanes_many_short_rows(xh, shape):
    rng = np.random.default_rng(52)
    x = rng.standard_normal(shape)
    edges = np.linspace(-3, 3, 25)
    want, _ = onp.histogram(x, bins=edges, axis=1)
    got, _ = xh.histogram(_dev(x), bins=edges, axis=1)
    desc = _describe_last(xh, [_dev(x[:1])], [edges])
    assert "family=flat_rows" in desc, desc  # dense rows: streamed as one contiguous array
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    plan = _plan_for(xh, [_dev(x[:1])], [edges])
    plan.set_param("flat_rows", -1)
    try:
        got, _ = xh.histogram(_dev(x), bins=edges, axis=1)
        desc = plan.describe()
    finally:
        plan.set_param("flat_rows", 0)
    assert "family=lanes" in desc and "transpose=fused" in desc, desc  # (its predecessor) one pass: load, turn in LDS, count
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    xf = x.astype(np.float32)[:, ::-1].copy()
    xf[::7, 0] = np.nan
    np.testing.assert_array_equal(xh.histogram(_dev(xf), bins=edges, axis=1)[0].cpu().numpy(), onp.histogram(xf, bins=edges, axis=1)[0])
    sl = _dev(x)[:, 1:]  # row stride != number of columns, unaligned row starts
    np.testing.assert_array_equal(xh.histogram(sl, bins=edges, axis=1)[0].cpu().numpy(), onp.histogram(x[:, 1:], bins=edges, axis=1)[0])
    w = rng.uniform(0, 1, shape).astype(np.float32)
    want, _ = onp.histogram(x, bins=edges, axis=1, weights=w, density=True)
    got, _ = xh.histogram(_dev(x), bins=edges, axis=1, weights=_dev(w), density=True)
    desc = _describe_last(xh, [_dev(x[:1])], [edges])
    assert "family=flat_rows" in desc and "weighted=1" in desc, desc  # dense rows with weights: streamed flat as well
    assert_hist_equal(got.cpu().numpy(), want, True)
    plan.set_param("flat_rows", -1)
    try:
        got, _ = xh.histogram(_dev(x), bins=edges, axis=1, weights=_dev(w), density=True)
        desc = plan.describe()
    finally:
        plan.set_param("flat_rows", 0)
    if shape[1] <= 80:
        assert "family=lanes" in desc and "transpose=1" in desc, desc  # (before) weighted, very short rows: transposed scratch + lanes
    else:
        assert "family=fast" in desc and "direct_store=1" in desc, desc  # (before) one 64-thread workgroup per row, plain-store flush
    assert_hist_equal(got.cpu().numpy(), want, True)
    # host route reaches the same kernels through the staged copy
    np.testing.assert_array_equal(xh.histogram(x, bins=edges, axis=1)[0], onp.histogram(x, bins=edges, axis=1)[0])


def test_lanes_2d_joint_nonuniform_and_binary_search_tables(xh):
    rng = np.random.default_rng(53)
    a = rng.standard_normal((6000, 33)).astype(np.float32)
    b = rng.standard_normal((6000, 33)).astype(np.float32)
    ea = _nonuniform_edges(rng, 9)
    eb = np.concatenate([[-4.0], -4.0 + np.cumsum(np.geomspace(1e-9, 4.0, 11))])  # crowded buckets -> SCAN 0
    want, _ = onp.histogram(a, b, bins=[ea, eb], axis=1)
    got, _ = xh.histogram(_dev(a), _dev(b), bins=[ea, eb], axis=1)
    desc = _describe_last(xh, [_dev(a[:1]), _dev(b[:1])], [ea, eb])
    assert "family=flat_rows" in desc and "scan=0" in desc, desc
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    plan = _plan_for(xh, [_dev(a[:1]), _dev(b[:1])], [ea, eb])
    plan.set_param("flat_rows", -1)
    try:
        got, _ = xh.histogram(_dev(a), _dev(b), bins=[ea, eb], axis=1)
        desc = plan.describe()
    finally:
        plan.set_param("flat_rows", 0)
    assert "family=lanes" in desc, desc
    np.testing.assert_array_equal(got.cpu().numpy(), want)


def test_lanes_few_rows_split_columns_and_accumulate(xh):
    """few rows: columns are split over workgroups and partial tiles are added atomically"""
    rng = np.random.default_rng(54)
    t = rng.standard_normal((50_000, 3, 5))  # reduce axis 0: 15 rows of 50000 strided columns
    edges = np.linspace(-4, 4, 33)
    want, _ = onp.histogram(t, bins=edges, axis=0)
    got, _ = xh.histogram(_dev(t), bins=edges, axis=0)
    desc = _describe_last(xh, [_dev(t[0])], [edges])
    assert "family=lanes" in desc and "direct_store=0" in desc, desc
    np.testing.assert_array_equal(got.cpu().numpy(), want)
    # block_size row blocking goes through separate launches on the same output
    x = rng.standard_normal((9000, 40))
    np.testing.assert_array_equal(
        xh.histogram(_dev(x), bins=edges, axis=1, block_size=2500)[0].cpu().numpy(), onp.histogram(x, bins=edges, axis=1)[0]
    )


def test_host_route_leading_axis_reduction_is_staged_as_it_lies(xh):
    """numpy (time, y, x) reduced over time: the [rows, cols] view has row stride 1; the library
    copies the rectangle untransposed and runs the row-per-lane kernel on the transposed view"""
    rng = np.random.default_rng(61)
    t = rng.standard_normal((500, 40, 33)).astype(np.float32)
    edges = np.linspace(-4, 4, 41)
    v = xh._rows_cols(t, [0], False)
    assert v.strides == (4, 40 * 33 * 4) and np.shares_memory(v, t)
    ptr, tag, rs, cs, _ir, _os, keep = xh._strided_view(v, "numpy")
    assert (rs, cs) == (1, 40 * 33) and ptr == t.ctypes.data  # no host copy
    np.testing.assert_array_equal(xh.histogram(t, bins=edges, axis=0)[0], onp.histogram(t, bins=edges, axis=0)[0])
    w = rng.uniform(0, 1, t.shape)
    assert_hist_equal(xh.histogram(t, bins=edges, axis=0, weights=w)[0], onp.histogram(t, bins=edges, axis=0, weights=w)[0], True)
    sub = t[:, 3:17, 5:]  # non-contiguous in both kept axes: falls back to a host copy, still exact
    np.testing.assert_array_equal(xh.histogram(sub, bins=edges, axis=0)[0], onp.histogram(sub, bins=edges, axis=0)[0])
    big = rng.standard_normal((3, 50_000_000 // 3)).astype(np.float32)  # column chunks of the staged route
    np.testing.assert_array_equal(xh.histogram(big, bins=edges, axis=0)[0][:5], onp.histogram(big[:, :5], bins=edges, axis=0)[0])


# ---------------------------------------------------------------------------------------------
# corner cases of the contract
# ---------------------------------------------------------------------------------------------
def test_degenerate_edges_and_values(xh):
    x = np.array([[0.0, 1.0, -1.0, np.nan, np.inf, -np.inf, 5e-324, -5e-324, 1.7976931348623157e308, 0.5]])
    cases = [
        np.array([0.0, 0.0]),                      # zero-width single bin: only x == 0 lands (right edge inclusive)
        np.array([-np.inf, 0.0, np.inf]),          # infinite outer edges
        np.array([5e-324, 1e-310, 1.0]),           # denormal edges
        np.array([-1e308, 1e308]),                 # span overflows to inf
        np.array([0.0, 0.5, 0.5, 0.5, 1.0]),       # repeated interior edge: empty bins
        np.array([1.0, 1.0, 1.0]),                 # all edges equal
        np.linspace(-1, 1, 3),
    ]
    for e in cases:
        want = onp.bincount_rows([x], [e])
        for resident in (False, True):
            got, _ = _run(xh, [x], [e], None, resident)
            np.testing.assert_array_equal(got, want, err_msg=str(e))
        wts = np.arange(1.0, x.size + 1).reshape(1, -1)
        got, _ = _run(xh, [x], [e], wts, True)
        assert_hist_equal(got, onp.bincount_rows([x], [e], wts), True)


def test_single_edge_means_zero_bins(xh):
    x = np.random.default_rng(0).standard_normal((3, 10))
    got, _ = _run(xh, [x], [np.array([0.0])], None, True)
    assert got.shape == (3, 0)
    got, _ = _run(xh, [x, x], [np.array([0.0]), np.linspace(-1, 1, 4)], None, False)
    assert got.shape == (3, 0, 3)


def test_all_samples_dropped_and_all_in_last_bin(xh):
    edges = [np.linspace(0, 1, 11)]
    nan = np.full((2, 10_000), np.nan)
    np.testing.assert_array_equal(_run(xh, [nan], edges, None, True)[0], np.zeros((2, 10), dtype=np.int64))
    ones = np.ones((2, 10_000))
    want = np.zeros((2, 10), dtype=np.int64)
    want[:, -1] = 10_000
    np.testing.assert_array_equal(_run(xh, [ones], edges, None, True)[0], want)
    # density of an empty row is NaN, like the reference
    h, _ = xh.histogram(_dev(nan), bins=edges[0], axis=1, density=True)
    assert np.isnan(h.cpu().numpy()).all()


def test_eight_inputs_is_the_limit(xh):
    rng = np.random.default_rng(9)
    s = [rng.standard_normal((2, 500)) for _ in range(8)]
    e = [np.linspace(-3, 3, 3)] * 8
    np.testing.assert_array_equal(_run(xh, s, e, None, True)[0], onp.bincount_rows(s, e))
    with pytest.raises(NotImplementedError):
        _run(xh, s + [s[0]], e + [e[0]], None, True)


def test_float32_denormal_samples_and_edges(xh):
    """the float32 threshold domain must compare denormals exactly (no flush-to-zero)"""
    tiny = np.array([1e-45, 3e-45, 1e-41, 1e-39, 1.1754942e-38, 1.17549435e-38, 0.0, -1e-45, -1e-39], dtype=np.float32)
    x = np.tile(tiny, 50).reshape(2, -1)
    for e in (np.array([0.0, 1e-44, 1e-40, 1.17549435e-38, 1.0]), np.array([-1e-40, -1e-45, 0.0, 2e-45, 1e-38]),
              np.array([1e-45, 2e-45, 3e-45, 4e-45])):
        want = onp.bincount_rows([x], [e])
        for resident in (False, True):
            np.testing.assert_array_equal(_run(xh, [x], [e], None, resident)[0], want, err_msg=str(e))
        np.testing.assert_array_equal(_run(xh, [x], [e], None, True, force_generic=1)[0], want)


# ---------------------------------------------------------------------------------------------
# grouped-row views: reduced axes between kept axes (ABI v2), no moveaxis+reshape copy
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("resident", [False, True], ids=["host", "device"])
@pytest.mark.parametrize("dt", [np.float32, np.float64, np.int32])
def test_middle_axis_reductions_run_uncopied(xh, dt, resident):
    rng = np.random.default_rng(71)
    t = (rng.standard_normal((23, 31, 300)) * 2).astype(dt)  # inner dim 300 >= a wave: lanes coalesce
    edges = np.linspace(-4, 4, 33)
    w = rng.uniform(0, 1, t.shape)
    for axis in ([1], [0], [2], [0, 1], [1, 2]):
        want, _ = onp.histogram(t, bins=edges, axis=axis)
        got, _ = xh.histogram(_maybe(t, resident), bins=edges, axis=axis)
        np.testing.assert_array_equal(_tonp(got), want, err_msg=str(axis))
        want, _ = onp.histogram(t, bins=edges, axis=axis, weights=w)
        got, _ = xh.histogram(_maybe(t, resident), bins=edges, axis=axis, weights=_maybe(w, resident))
        assert_hist_equal(_tonp(got), want, True)
    if dt != np.int32 and resident:
        desc = _plan_for(xh, [_dev(t[0])], [edges]).describe()
        assert "family=" in desc


def _maybe(a, resident):
    return _dev(a) if resident else a


def _tonp(a):
    return a.cpu().numpy() if hasattr(a, "cpu") else np.asarray(a)


@pytest.mark.parametrize("resident", [False, True], ids=["host", "device"])
def test_grouped_rows_4d_two_inputs_broadcast_weights(xh, resident):
    rng = np.random.default_rng(72)
    a = rng.standard_normal((5, 7, 11, 70))
    b = rng.standard_normal((5, 7, 11, 70))
    ea, eb = np.linspace(-3, 3, 7), _nonuniform_edges(rng, 6)
    for axis, wshape in (([1, 2], (5, 1, 1, 70)), ([2], (1, 7, 11, 1)), ([1], (5, 7, 11, 70)), ([0, 1, 2], (1, 1, 1, 70))):
        w = rng.uniform(0, 1, wshape)
        want, _ = onp.histogram(a, b, bins=[ea, eb], axis=axis, weights=w)
        got, _ = xh.histogram(_maybe(a, resident), _maybe(b, resident), bins=[ea, eb], axis=axis, weights=_maybe(w, resident))
        assert_hist_equal(_tonp(got), want, True)
    # sliced (non-contiguous) inputs: still described by strides
    sl = a[:, ::2, :, 3:]
    want, _ = onp.histogram(sl, bins=ea, axis=[1])
    np.testing.assert_array_equal(_tonp(xh.histogram(_maybe(a, resident)[:, ::2, :, 3:], bins=ea, axis=[1])[0]), want)
    # second input in a different memory order than the first: pairing of samples must survive
    bf = np.asfortranarray(b)
    want, _ = onp.histogram(a, bf, bins=[ea, eb], axis=[1, 2])
    np.testing.assert_array_equal(_tonp(xh.histogram(a, bf, bins=[ea, eb], axis=[1, 2])[0]), want)


def test_int64_samples_round_to_double_like_numpy(xh):
    """int64 data against float64 edges is compared AFTER rounding to double (numpy's promotion):
    2^53 + 1 is not representable and lands on the edge 2^53"""
    big = 2**53
    x = np.array([[big - 1, big, big + 1, big + 2, big + 3, -big - 1, 2**62, -(2**62)]], dtype=np.int64)
    e = np.array([-float(2**63), -float(big), float(big), float(big + 2), float(2**63)])
    want = onp.bincount_rows([x], [e])
    for resident in (False, True):
        for params in ({}, {"force_generic": 1}):
            if params and not resident:
                continue
            np.testing.assert_array_equal(_run(xh, [x], [e], None, resident, **params)[0], want)
    xs = np.tile(x, (1, 4000))  # long enough for the vector kernel's full tiles
    np.testing.assert_array_equal(_run(xh, [xs], [e], None, True)[0], onp.bincount_rows([xs], [e]))
    # unsigned bytes and half floats, ragged lengths around the 16- and 8-element vectors
    rng = np.random.default_rng(81)
    for n in (1, 15, 16, 17, 4095, 4097, 70001):
        u = rng.integers(0, 256, (2, n)).astype(np.uint8)
        eu = np.linspace(0, 255, 18)
        np.testing.assert_array_equal(_run(xh, [u], [eu], None, True)[0], onp.bincount_rows([u], [eu]))
        h = (rng.standard_normal((2, n)) * 2).astype(np.float16)
        h[0, 0] = np.nan
        eh = np.linspace(-4, 4, 17)
        np.testing.assert_array_equal(_run(xh, [h], [eh], None, True)[0], onp.bincount_rows([h], [eh]))
        w = rng.uniform(0, 1, (2, n))
        assert_hist_equal(_run(xh, [u], [eu], w, True)[0], onp.bincount_rows([u], [eu], w), True)


def test_execute_is_hip_graph_capturable(xh):
    """memset + kernel on the caller's stream: a device-resident execute can be captured into a
    hipGraph and replayed (eager launches are already ~10 us, so this is a property, not a speed-up)"""
    from xhistogram_amd import _native

    edges = np.linspace(-4, 4, 101)
    rng = np.random.default_rng(91)
    xs = [rng.standard_normal(200_000) for _ in range(2)]
    x = _dev(xs[0])
    plan = xh._get_plan([edges], _native.CMP_F64, 0)
    out = torch.zeros(100, dtype=torch.int64, device="cuda")
    xv = [_native.make_view(x.data_ptr(), _native.F64, x.numel(), 1)]
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        plan.execute(xv, None, 1, x.numel(), out.data_ptr(), False, _native.MEM_DEVICE, stream=s.cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        plan.execute(xv, None, 1, x.numel(), out.data_ptr(), False, _native.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
    for data in (xs[1], xs[0]):  # replay on new contents of the same buffer
        x.copy_(_dev(data))
        out.fill_(-1)
        g.replay()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), onp.bincount_rows([data.reshape(1, -1)], [edges])[0])


def test_counts_do_not_depend_on_launch_geometry(xh):
    """any lost or duplicated atomic would show as a count mismatch between geometries
    (the determinism check that stands in for a race detector)"""
    from xhistogram_amd import _native

    rng = np.random.default_rng(95)
    n = 3_000_017
    x = rng.standard_normal((1, n))
    y = rng.standard_normal((1, n))
    for samples, edges in (([x], [np.linspace(-4, 4, 101)]), ([x, y], [np.linspace(-4, 4, 65), _nonuniform_edges(rng, 49)])):
        want = onp.bincount_rows(samples, edges)
        for block in (64, 256, 1024):
            for grid in (1, 7, 256, 5000):
                for copies in (0, 1, 8):
                    got, _ = _run(xh, samples, edges, None, True, block_threads=block, grid_blocks=grid, lds_copies=copies)
                    np.testing.assert_array_equal(got, want, err_msg="block=%d grid=%d copies=%d" % (block, grid, copies))


def test_tables_too_large_for_lds_are_read_through_l2(xh):
    """50001 edges (400 KB of tables) that are NOT arithmetic: the generic family reads the tables
    from global memory and accumulates with global atomics (np.linspace edges of that size take the
    table-free vector kernel); 65536+ edges per dimension are refused with a clear message"""
    rng = np.random.default_rng(97)
    x = rng.standard_normal((2, 300_000)) * 2
    e = np.linspace(-5, 5, 50_001)
    got, desc = _run(xh, [x], [e], None, True)
    assert "family=fast" in desc and "scan=5" in desc, desc
    np.testing.assert_array_equal(got, onp.bincount_rows([x], [e]))
    e = e.copy()
    e[777] = np.nextafter(e[777], np.inf)
    got, desc = _run(xh, [x], [e], None, True)
    assert "family=generic" in desc and "hist=global" in desc, desc
    np.testing.assert_array_equal(got, onp.bincount_rows([x], [e]))
    w = rng.uniform(0, 1, x.shape)
    assert_hist_equal(_run(xh, [x], [e], w, False)[0], onp.bincount_rows([x], [e], w), True)


@pytest.mark.parametrize("weighted", [False, True])
def test_more_than_65535_edges_per_dimension(xh, weighted):
    """no 16-bit table fields any more: arithmetic edges run table-free on the vector kernels
    (in bin slices here, the partitioned mode when forced), anything else binary-searches
    the whole edge array in the generic family"""
    rng = np.random.default_rng(98)
    e = np.linspace(-5, 5, 100_001)
    x = np.concatenate([rng.standard_normal((1, 200_000)) * 2, e[None, ::7], np.nextafter(e, -np.inf)[None, ::11],
                        np.array([[np.nan, np.inf, -np.inf, -5.0, 5.0]])], axis=1)
    w = rng.uniform(0, 1, x.shape) if weighted else None
    want = onp.bincount_rows([x], [e], w)
    got, desc = _run(xh, [x], [e], w, True)
    assert "family=fast" in desc and "scan=5" in desc, desc  # (2 x 10^5 samples: memory-side atomics win)
    assert_hist_equal(got, want, weighted)
    got, desc = _run(xh, [x], [e], w, True, slices=1)
    assert "family=fast" in desc and "scan=5" in desc and "slices=1" not in desc, desc  # 100000 bins: LDS bin slices
    assert_hist_equal(got, want, weighted)
    got, desc = _run(xh, [x], [e], w, True, partition=1)
    assert "hist=partitioned" in desc and "scan=5" in desc, desc
    assert_hist_equal(got, want, weighted)
    assert_hist_equal(_run(xh, [x], [e], w, False)[0], want, weighted)  # host route
    # not arithmetic: generic family, binary search over 70001 edges read through L2
    e2 = np.sort(rng.uniform(-5, 5, 70_001))
    x2 = np.concatenate([rng.standard_normal((3, 20_000)) * 2, np.tile(e2[None, ::4][:, :5000], (3, 1))], axis=1)
    w2 = rng.uniform(0, 1, x2.shape) if weighted else None
    got, desc = _run(xh, [x2], [e2], w2, True)
    assert "family=generic" in desc, desc
    assert_hist_equal(got, onp.bincount_rows([x2], [e2], w2), weighted)
    # two inputs, one of them with a huge edge array
    e3 = [np.linspace(0, 1, 4), e2]
    y2 = rng.uniform(-0.1, 1.1, x2.shape)
    got, desc = _run(xh, [y2, x2], e3, w2, True)
    assert_hist_equal(got, onp.bincount_rows([y2, x2], e3, w2), weighted)


@pytest.mark.parametrize("dtype,k", [(np.float32, 2), (np.float64, 3), (np.float64, 32), (np.int32, 17), (np.uint8, 32), (np.float16, 5)])
def test_few_rows_along_the_contiguous_direction(dtype, k):
    """(N, k) reduced over its leading axis, k small (columns of a table): the row-per-lane kernels split
    their 256 lanes into 256 / R groups of R >= k rows, every group on its own columns; with and without
    weights, dense and padded column stride, device-resident and host arrays."""
    from xhistogram_amd import _native, core

    rng = np.random.default_rng(k)
    n = (1 << 22) // k + 1237
    raw = rng.standard_normal((n, k + 3)) * (30 if np.dtype(dtype).kind in "iu" else 1)
    if np.dtype(dtype).kind == "u":
        raw = np.abs(raw)
    x_pad = raw.astype(dtype)
    edges = np.linspace(-60, 60, 41) if np.dtype(dtype).kind in "iu" else np.linspace(-3, 3, 41)
    w = rng.uniform(0, 1, (n, k))
    want = onp.histogram(x_pad[:, :k], bins=edges, axis=0)[0]
    wantw = onp.histogram(x_pad[:, :k], bins=edges, axis=0, weights=w)[0]
    for xd in (_dev(np.ascontiguousarray(x_pad[:, :k])), _dev(x_pad)[:, :k]):  # column stride k, then k + 3
        got = core.histogram(xd, bins=edges, axis=0)[0]
        desc = core._get_plan([edges], _native.CMP_F64, 0).describe()
        assert "family=lanes" in desc, desc
        assert_hist_equal(got.cpu().numpy(), want, weighted=False)
        gotw = core.histogram(xd, bins=edges, axis=0, weights=_dev(w))[0]
        assert_hist_equal(gotw.cpu().numpy(), wantw, weighted=True)
    got_host = core.histogram(x_pad[:, :k], bins=edges, axis=0)[0]  # numpy in: staged as it lies
    assert_hist_equal(got_host, onp.histogram(x_pad[:, :k], bins=edges, axis=0)[0], weighted=False)
    # more rows than one workgroup holds, non-uniform edges (binary search for the integer / half kernels)
    x2 = (rng.standard_normal((5000, 300)) * (30 if np.dtype(dtype).kind in "iu" else 1)).astype(dtype)
    if np.dtype(dtype).kind == "u":
        x2 = np.abs(x2)
    e2 = np.unique(np.round(_nonuniform_edges(rng, 41) * (15 if np.dtype(dtype).kind in "iu" else 0.75), 3))
    w2 = rng.uniform(0, 1, x2.shape)
    got2 = core.histogram(_dev(x2), bins=e2, axis=0)[0]
    assert "family=lanes" in core._get_plan([e2], _native.CMP_F64, 0).describe()
    assert_hist_equal(got2.cpu().numpy(), onp.histogram(x2, bins=e2, axis=0)[0], weighted=False)
    got2w = core.histogram(_dev(x2), bins=e2, axis=0, weights=_dev(w2))[0]
    assert_hist_equal(got2w.cpu().numpy(), onp.histogram(x2, bins=e2, axis=0, weights=w2)[0], weighted=True)
    if k != 3:
        return
    # a per-row broadcast weight stays a broadcast
    wrow = rng.uniform(0, 1, (n, 1))
    gotb = core.histogram(_dev(x_pad)[:, :k], bins=edges, axis=0, weights=_dev(wrow))[0]
    assert_hist_equal(gotb.cpu().numpy(), onp.histogram(x_pad[:, :k], bins=edges, axis=0, weights=np.broadcast_to(wrow, (n, k)))[0], weighted=True)


@pytest.mark.parametrize("case", ["f32_x_f64", "i32_x_i32", "u8_x_f32_w_int", "i32_1d_50000_bins", "i64_x_f64_int_edges"])
def test_odd_dtype_mixtures_with_histograms_beyond_lds(case):
    """device-resident inputs only the generic family takes, histogram beyond its LDS: promoted to float64 on
    the device where the comparison runs in float64 anyway (core._promote_for_big_histograms), exact otherwise"""
    from xhistogram_amd import _native, core

    rng = np.random.default_rng(len(case))
    n = 300_000
    e = np.linspace(-4, 4, 257)
    w = None
    if case == "f32_x_f64":
        xs, es = [rng.standard_normal(n).astype(np.float32), rng.standard_normal(n)], [e, _nonuniform_edges(rng, 257)]
    elif case == "i32_x_i32":
        xs, es = [rng.integers(-150, 150, n).astype(np.int32), rng.integers(-150, 150, n).astype(np.int32)], [np.arange(-128, 129), np.arange(-128, 129) - 0.5]
    elif case == "u8_x_f32_w_int":
        xs, es = [rng.integers(0, 256, n).astype(np.uint8), rng.standard_normal(n).astype(np.float32)], [np.arange(257), e]
        w = rng.integers(0, 5, n).astype(np.int16)
    elif case == "i32_1d_50000_bins":
        xs, es = [rng.integers(-30000, 30000, n).astype(np.int32)], [np.arange(-25000, 25001)]
    else:  # an exact int64 axis next to a float axis: stays in its domains
        xs, es = [rng.integers(-2**40, 2**40, n), rng.standard_normal(n)], [np.linspace(-2**40, 2**40, 257).astype(np.int64), e]
    want = onp.histogram(*xs, bins=es, weights=w)[0]
    got = core.histogram(*[_dev(x) for x in xs], bins=es, weights=None if w is None else _dev(w))[0]
    assert_hist_equal(got.cpu().numpy(), want, weighted=w is not None)
    got_host = core.histogram(*xs, bins=es, weights=w)[0]
    assert_hist_equal(got_host, want, weighted=w is not None)


# ---------------------------------------------------------------------------------------------
# the short cut for plain device-resident calls must be indistinguishable from the general path
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_resident_short_cut_equals_the_general_path(xh, dtype):
    rng = np.random.default_rng(8)
    x = rng.standard_normal((6, 7, 500)).astype(dtype)
    y = rng.standard_normal((6, 7, 500)).astype(dtype)
    w = rng.uniform(0, 1, x.shape)
    x[0, 0, :5] = np.nan
    e1, e2 = np.linspace(-3, 3, 31), np.sort(rng.uniform(-3, 3, 12))
    xt, yt, wt = _dev(x), _dev(y), _dev(w)
    cases = [
        ((xt,), dict(bins=e1)),
        ((xt,), dict(bins=e1, axis=2)),
        ((xt,), dict(bins=e1, axis=(1, 2), weights=wt)),
        ((xt, yt), dict(bins=[e1, e2], axis=(1, 2), weights=wt, density=True)),
        ((xt, yt), dict(bins=[e1, e2], density=True)),
        ((xt,), dict(bins=e1.astype(np.float32), axis=-1)),
        ((xt,), dict(bins=e1, axis=0)),                                     # leading axes: rows are the contiguous direction
        ((xt,), dict(bins=e1, axis=(0, 1), weights=wt)),
        ((xt, yt), dict(bins=[e1, e2], axis=(1, 0), weights=wt, density=True)),
    ]
    for args, kw in cases:
        assert xh._resident_fast_path(args, kw["bins"], None, xh._normalise_axis(kw.get("axis"), 3), kw.get("weights"), kw.get("density", False), "auto") is not None
        fast, fe = xh.histogram(*args, **kw)
        slow, se = xh.histogram(*args, block_size=1 << 40, **kw)  # an explicit block size: the general path
        oargs = [a.cpu().numpy() for a in args]
        okw = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in kw.items()}
        want, _ = onp.histogram(*oargs, **okw)
        assert fast.dtype == slow.dtype and tuple(fast.shape) == tuple(slow.shape) == np.asarray(want).shape
        assert_hist_equal(fast.cpu().numpy(), want, weighted="weights" in kw or kw.get("density", False))
        assert_hist_equal(slow.cpu().numpy(), want, weighted="weights" in kw or kw.get("density", False))
        for a, b in zip(fe, se):
            np.testing.assert_array_equal(a, b)
    # what the short cut must leave alone
    assert xh._resident_fast_path((xt[:, :, ::2],), e1, None, None, None, False, "auto") is None      # not contiguous
    assert xh._resident_fast_path((xt,), 10, None, None, None, False, "auto") is None                 # integer bins
    assert xh._resident_fast_path((xt,), e1, None, [1], None, False, "auto") is None                  # an axis in the middle
    assert xh._resident_fast_path((xt,), e1, None, [0, 2], None, False, "auto") is None               # axes apart
    assert xh._resident_fast_path((xt,), e1, None, None, wt[0], False, "auto") is None                # broadcast weights
    assert xh._resident_fast_path((xt.to(torch.float16),), e1, None, None, None, False, "auto") is None
    with pytest.raises(ValueError):
        xh.histogram(xt, bins=np.array([0.0, 2.0, 1.0]))  # numpy's monotonicity check still runs
    # in-place edits of an edge array are seen (the cache is keyed on the bytes)
    e = np.linspace(-1, 1, 5)
    h1, _ = xh.histogram(xt, bins=e)
    e[:] = np.linspace(-2, 2, 5)
    h2, _ = xh.histogram(xt, bins=e)
    np.testing.assert_array_equal(h2.cpu().numpy(), onp.histogram(x, bins=np.linspace(-2, 2, 5))[0])
    assert int(h2.sum()) > int(h1.sum())


def test_host_inputs_of_mixed_dtypes_with_a_big_histogram_take_the_vector_kernels(xh):
    """numpy inputs: float32 x float64 samples, int32 weights, 300 x 300 bins (beyond LDS) — converted to float64
    on the host (exact: numpy promotes the same way) instead of 2.5e10/s memory-side atomics of the generic family"""
    from xhistogram_amd import _native

    rng = np.random.default_rng(12)
    n = 400_000
    x = rng.standard_normal(n).astype(np.float32)
    y = rng.standard_normal(n)
    w = rng.integers(-3, 9, n).astype(np.int32)
    e = [np.linspace(-4, 4, 301), np.sort(rng.uniform(-4, 4, 301))]
    got, _ = xh.histogram(x, y, bins=e, weights=w)
    want, _ = onp.histogram(x, y, bins=e, weights=w)
    assert_hist_equal(got, want, weighted=True)
    desc = xh._get_plan(e, _native.CMP_F64, xh._host_device()).describe()
    assert "family=fast" in desc, desc
    got, _ = xh.histogram(x, y, bins=e)
    np.testing.assert_array_equal(got, onp.histogram(x, y, bins=e)[0])


@pytest.mark.parametrize("resident", [False, True], ids=["host", "device"])
def test_mixed_dtype_vector_kernels_against_the_oracle(xh, resident):
    """float32 next to float64, integers in a joint histogram, integer / bool / half weights: the MIXED variant of the
    vector kernels (per-array vector loads, consumed as float64) — every ragged length, rows, arithmetic and random edges"""
    from xhistogram_amd import _native

    rng = np.random.default_rng(31)
    for n in (1, 3, 4, 5, 1023, 4096, 70_001):
        x = rng.standard_normal((3, n)).astype(np.float32)
        y = rng.standard_normal((3, n))
        k = rng.integers(-40, 40, (3, n)).astype(np.int16)
        u = rng.integers(0, 255, (3, n)).astype(np.uint8)
        b = rng.integers(0, 2, (3, n)).astype(bool)
        wi = rng.integers(-5, 9, (3, n)).astype(np.int32)
        wh = rng.uniform(0, 2, (3, n)).astype(np.float16)
        x[0, 0] = np.nan
        e_lin, e_rnd, e_int, e_u8 = np.linspace(-3, 3, 25), np.sort(rng.uniform(-3, 3, 18)), np.arange(-41.0, 42.0, 3), np.linspace(0, 255, 9)
        cases = [
            ([x, y], [e_lin, e_rnd], None), ([x, y], [e_rnd, e_lin], wi), ([k, u], [e_int, e_u8], None),
            ([y], [e_lin], wi), ([x], [e_rnd], b), ([y, k, x], [e_lin, e_int, e_rnd], wh), ([u, y], [e_u8, e_rnd], wi),
        ]
        for samples, edges, w in cases:
            got, desc = _run(xh, samples, edges, w, resident)
            want = onp.bincount_rows(samples, edges, w)
            assert_hist_equal(got, want, weighted=w is not None)
            if n == 4096:  # (rows of 1- / 2-byte elements must be dword-aligned for the vector loads: 70001 is not)
                assert "mixed-dtypes" in desc, desc
            got_g, desc_g = _run(xh, samples, edges, w, True, force_generic=1)
            assert "generic" in desc_g and "mixed" not in desc_g, desc_g
            assert_hist_equal(got_g, want, weighted=w is not None)


@pytest.mark.parametrize("resident", [False, True], ids=["host", "device"])
def test_table_columns_with_many_bins_are_gathered_into_rows(xh, resident):
    """(n, K) tables histogrammed over their leading axis with more bins than the row-per-lane kernels hold: the library
    gathers the (strided) columns into dense rows and runs the vector kernels on them — whole tables, column slices of a
    wider table, with weights, beyond-LDS histograms"""
    rng = np.random.default_rng(41)
    n = 30_011
    for dtype, K in ((np.float32, 4), (np.float64, 6), (np.int16, 8)):
        t = (rng.standard_normal((n, K)) * (1 if dtype != np.int16 else 50)).astype(dtype)
        w = rng.uniform(0, 1, (n, K))
        if dtype != np.int16:
            t[7, 1] = np.nan
        lo, hi = (-4, 4) if dtype != np.int16 else (-200, 200)
        for nb in (3000, 150_000):
            e = np.linspace(lo, hi, nb + 1)
            for view, wv in ((t, None), (t[:, 1:4], None), (t, w), (t[:, :3], w[:, :3])):
                a = _dev(view) if resident else view
                if resident and not view.flags.c_contiguous:
                    a = _dev(t)[:, (1 if view.shape[1] == 3 and wv is None else 0):(4 if wv is None else 3)]
                ww = None if wv is None else (_dev(w)[:, : view.shape[1]] if resident else wv)
                got, _ = xh.histogram(a, bins=e, axis=0, weights=ww)
                got = got.cpu().numpy() if resident else got
                want, _ = onp.histogram(np.ascontiguousarray(view), bins=e, axis=0, weights=None if wv is None else np.ascontiguousarray(wv))
                assert_hist_equal(got, want, weighted=wv is not None)


def test_one_pass_routing_with_few_partitions_and_many_tiles(xh):
    """3 partitions, 1024-record chunks, ~40 tiles per workgroup: every tile asks the workgroup's id stock for several
    chunks per partition (found by tools/soak.py: the pool was once sized without the ids a stock range drops when it
    runs out — a memory fault).  Exact against the three-pass route (an independent implementation) and the in-range count."""
    from xhistogram_amd import _native

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(77)
    n = 120_000_000
    a = [torch.empty(n, dtype=torch.float32, device=dev).normal_(generator=g) for _ in range(3)]
    edges = [np.linspace(-3, 3, 48), np.linspace(-3, 3, 18) ** 3 / 9.0, np.linspace(-3, 3, 101)]
    plan = xh._get_plan(edges, _native.CMP_F64, 0)
    plan.set_param("partition", 1)  # (left alone, 12 B/sample and 2 bin slices would win the cost model)
    plan.set_param("min_parts", 1)  # (and the bins would be cut into 16+ partitions)
    try:
        h, _ = xh.histogram(*a, bins=edges)
        assert "route=fused" in plan.describe() and "parts=3" in plan.describe(), plan.describe()
        plan.set_param("fused", -1)
        h3, _ = xh.histogram(*a, bins=edges)
        assert "route=fused" not in plan.describe() and "partitioned" in plan.describe(), plan.describe()
    finally:
        plan.set_param("fused", 0)
        plan.set_param("partition", 0)
        plan.set_param("min_parts", 0)
    assert torch.equal(h, h3)
    inside = torch.ones(n, dtype=torch.bool, device=dev)
    for x, e in zip(a, edges):
        inside &= (x >= float(e[0])) & (x <= float(e[-1]))
    assert int(h.sum()) == int(inside.sum())
    m = 1_000_000
    hp, _ = xh.histogram(*[x[:m] for x in a], bins=edges)
    np.testing.assert_array_equal(hp.cpu().numpy(), onp.histogram(*[x[:m].cpu().numpy() for x in a], bins=edges)[0])


# ---------------------------------------------------------------------------------------------
# packed bucket entries (count_le_pack): float64 samples on non-uniform edges, float32 pre-compare + exact redo
def _pack_torture(edges, rng, n_random):
    """what the float32 pre-compare cannot decide, and everything around it: every edge, its float64 neighbours, the float32
    image of every edge and ITS float64 and float32 neighbours, the rounding boundaries between float32 values next to an
    edge, samples scattered within one float32 ulp of every edge, plus randoms and the usual specials"""
    e = np.asarray(edges, dtype=np.float64)
    with np.errstate(over="ignore"):
        f32 = e.astype(np.float32)
    f = f32.astype(np.float64)
    fu = np.nextafter(f32, np.float32(np.inf)).astype(np.float64)
    fd = np.nextafter(f32, np.float32(-np.inf)).astype(np.float64)
    ulp = np.where(np.isfinite(fu - fd), (fu - fd) / 2, 0.0)
    mids = [(f + fu) / 2, (f + fd) / 2]
    parts = [e, np.nextafter(e, -np.inf), np.nextafter(e, np.inf), f, np.nextafter(f, -np.inf), np.nextafter(f, np.inf), fu, fd]
    for m in mids:
        parts += [m, np.nextafter(m, -np.inf), np.nextafter(m, np.inf)]
    parts += [np.repeat(e, 8) + rng.uniform(-1.5, 1.5, e.size * 8) * np.repeat(ulp, 8),
              rng.uniform(e[0] - 0.1 * (e[-1] - e[0]), e[-1] + 0.1 * (e[-1] - e[0]), n_random),
              np.array([np.nan, np.inf, -np.inf, e[0], e[-1], 0.0, -0.0, 1e-10, -1e-10, 1e-40, -1e-40, 5e-324, -5e-324]),
              0.5 * np.abs(e[e != 0]).min() * np.array([1.0, -1.0, 1.0 - 1e-9, -1.0 + 1e-9])]
    x = np.concatenate(parts)
    rng.shuffle(x)
    return x.reshape(1, -1)


def _sorted_uniform(rng, n, lo, hi):
    e = np.sort(rng.uniform(lo, hi, n))
    e[0], e[-1] = lo, hi
    return e


_PACK_EDGES = {
    "c3_like_257": lambda rng: _sorted_uniform(rng, 257, -4.0, 4.0),
    "geometric_300": lambda rng: np.geomspace(1e-3, 50.0, 300),
    "geometric_2001": lambda rng: np.geomspace(1e-3, 50.0, 2001),
    "logspace_negative": lambda rng: -np.logspace(3, -2, 400),
    "symlog": lambda rng: np.concatenate([-np.geomspace(50.0, 1e-3, 200), [0.0], np.geomspace(1e-3, 50.0, 200)]),
    "symlog_no_zero": lambda rng: np.concatenate([-np.geomspace(9.0, 1e-5, 120), np.geomspace(3e-4, 700.0, 333)]),
    "negative_129": lambda rng: _sorted_uniform(rng, 129, -1000.0, -999.0),
    "few_5": lambda rng: np.array([-1.0, -0.25, 0.1, 0.7, 3.0]),
    "two_edges": lambda rng: np.array([0.3, 0.7]),
    "big_magnitude": lambda rng: _sorted_uniform(rng, 65, 1.0e30, 3.0e30),
    "small_magnitude": lambda rng: _sorted_uniform(rng, 65, 1.0e-30, 3.0e-30),
    # float32 cannot tell these edges apart / hold them at all: the packed set must not be offered, results stay exact
    "beyond_float32": lambda rng: _sorted_uniform(rng, 33, 1.0e300, 3.0e300),
    "below_float32": lambda rng: _sorted_uniform(rng, 33, 1.0e-300, 3.0e-300),
    "one_float32_ulp": lambda rng: 1.0 + np.sort(rng.uniform(0, 1e-8, 40)),
    "duplicates": lambda rng: np.sort(np.concatenate([rng.uniform(-2, 2, 50), [0.5] * 4, [-1.0] * 2])),
    "cluster_of_five": lambda rng: np.sort(np.concatenate([rng.uniform(-2, 2, 60), 0.123 + np.arange(5) * 1e-9])),
}
_PACK_NOT_OFFERED = {"beyond_float32", "below_float32", "one_float32_ulp", "duplicates", "cluster_of_five"}
_PACK_EITHER = {"two_edges"}
# float-bits buckets (scan=8); "symlog": edges on both sides of zero — magnitudes below the smallest |edge| are lifted to it and the
# empty binades around zero cut out of the key space
_PACK_KEY_MAP = {"geometric_300", "geometric_2001", "logspace_negative", "symlog", "symlog_no_zero"}


@pytest.mark.parametrize("weights", [None, "f64", "f32"])
@pytest.mark.parametrize("kind", sorted(_PACK_EDGES))
def test_packed_bucket_entries_1d_on_and_around_every_edge(xh, kind, weights):
    rng = np.random.default_rng(sum(map(ord, kind)))
    edges = [_PACK_EDGES[kind](rng)]
    x = _pack_torture(edges[0], rng, 100_000)
    w = None if weights is None else rng.uniform(-1, 2, x.shape).astype(np.float64 if weights == "f64" else np.float32)
    want = onp.bincount_rows([x], edges, w)
    got, desc = _run(xh, [x], edges, w, True, pack=1)
    assert_hist_equal(got, want, w is not None)
    offered = "scan=6" in desc or "scan=7" in desc or "scan=8" in desc
    if kind in _PACK_NOT_OFFERED:
        assert not offered, desc
    elif kind not in _PACK_EITHER:
        assert ("scan=8" in desc or offered) and "family=fast" in desc, desc
    if kind in _PACK_KEY_MAP:
        assert "scan=8" in desc, desc
        got, desc = _run(xh, [x], edges, w, True)  # and it is the automatic choice where the alternative is a binary search
        assert "scan=8" in desc, desc
        assert_hist_equal(got, want, w is not None)
    got, _ = _run(xh, [x], edges, w, True, pack=-1)
    assert_hist_equal(got, want, w is not None)


@pytest.mark.parametrize("weights", [None, "f64"])
@pytest.mark.parametrize("dims", [2, 3])
def test_packed_bucket_entries_joint(xh, dims, weights):
    """the automatic choice for joint histograms of float64 samples on non-uniform edges (BASELINE C3's shape)"""
    rng = np.random.default_rng(400 + dims)
    nb = ([257, 257] if weights is None else [65, 90]) if dims == 2 else ([33, 17, 41] if weights is None else [17, 9, 21])  # (weighted: float64 sums that still fit LDS)
    edges = [_sorted_uniform(rng, k, -4.0, 4.0) for k in nb]
    cols = [_pack_torture(e, rng, 60_000) for e in edges]
    n = min(c.shape[1] for c in cols)
    samples = [c[:, :n].copy() for c in cols]
    for d in range(1, dims):  # every edge-adjacent sample of one input meets ordinary samples of the others
        samples[d] = np.roll(samples[d], 7919 * d, axis=1)
    w = None if weights is None else rng.uniform(0, 1, samples[0].shape)
    want = onp.bincount_rows(samples, edges, w)
    got, desc = _run(xh, samples, edges, w, True)
    assert ("scan=6" in desc or "scan=7" in desc) and "family=fast" in desc, desc
    if dims == 2 and weights is None:
        assert "hist=packed16" in desc, desc
    assert_hist_equal(got, want, w is not None)
    got, desc = _run(xh, samples, edges, w, True, pack=-1)
    assert "scan=6" not in desc and "scan=7" not in desc, desc
    assert_hist_equal(got, want, w is not None)
    got, _ = _run(xh, samples, edges, w, False)  # host route
    assert_hist_equal(got, want, w is not None)


def test_packed_bucket_entries_many_rows_and_ragged_tiles(xh):
    rng = np.random.default_rng(431)
    edges = [_sorted_uniform(rng, 40, -3.0, 3.0), _sorted_uniform(rng, 23, -3.0, 3.0)]
    for shape in [(7, 10_001), (300, 777), (1, 3), (2, 4097)]:
        x, y = rng.standard_normal(shape), rng.standard_normal(shape)
        x[:, ::5] = rng.choice(edges[0], size=x[:, ::5].shape)  # plenty of samples ON an edge: the exact redo runs in most wavefronts
        got, desc = _run(xh, [x, y], edges, None, True)
        np.testing.assert_array_equal(got, onp.bincount_rows([x, y], edges), err_msg=desc)


@pytest.mark.parametrize("name", ["sqrt", "sturges", "rice", "scott", "fd", "auto", "doane", "stone"])
def test_bin_estimators_cut_float32_data_to_the_range_in_float32(xh, name):
    """numpy's keep mask compares float32 data with the range bounds ROUNDED TO float32 (NEP 50), so elements equal
    to float32(lo) < lo are kept — data clipped to 0.7 with range=(0.7, 1.0) has many of them"""
    rng = np.random.default_rng(11)
    a = np.clip(rng.uniform(0.0, 1.2, 40_000), 0.7, 1.1).astype(np.float32)
    assert float(np.float32(0.7)) < 0.7 and (a == np.float32(0.7)).sum() > 1000
    t = _dev(a)
    # (NumPy float64 / int64 scalars as bounds make numpy compare in float64 — those keep their value)
    for r in ((0.7, 1.0), (0.7, 0.7), (0.3, 1.1), (np.float64(0.7), 1.0), (np.float32(0.7), np.float64(1.0)), (0.7, np.float64(1.1)),
              (np.float64(0.0), np.float32(1.1))):
        try:
            want = np.histogram_bin_edges(a, bins=name, range=r)
        except ValueError:  # numpy's own "stone" trips over elements its float32 keep-mask lets in below a float64 first edge
            assert name == "stone"
            continue
        got = xh._device_bin_edges(t, name, r, False)
        np.testing.assert_array_equal(got, want, err_msg=str((name, r)))


@pytest.mark.parametrize("name", ["sqrt", "sturges", "rice", "scott"])
def test_moment_estimators_leave_huge_64_bit_integers_to_numpy(xh, name):
    """int64 magnitudes of 2^53 and more are not exact in the float64 reduction; numpy's integer arithmetic is"""
    rng = np.random.default_rng(12)
    a = (2 ** 60 + rng.integers(0, 2 ** 40, 5000)).astype(np.int64)  # (a span float64 edges can still resolve: numpy itself refuses less)
    t = _dev(a)
    want = np.histogram_bin_edges(a, bins=name)
    got = xh._device_bin_edges(t, name, None, False)
    np.testing.assert_array_equal(got, want)
    assert xh._device_estimator_edges(t, name, None, np.dtype(np.int64), False) is None  # declined, not guessed
    b = rng.integers(-(2 ** 40), 2 ** 40, 5000).astype(np.int64)  # ordinary 64-bit integers stay on the device
    assert xh._device_estimator_edges(_dev(b), name, None, np.dtype(np.int64), False) is not None
    np.testing.assert_array_equal(xh._device_bin_edges(_dev(b), name, None, False), np.histogram_bin_edges(b, bins=name))


def test_packed_bucket_entries_joint_of_a_log_axis_and_a_linear_one(xh):
    """the map is chosen per dimension: geometric edges (float-bits buckets) next to uneven linear ones"""
    rng = np.random.default_rng(440)
    edges = [np.geomspace(1e-4, 10.0, 100), _sorted_uniform(rng, 80, -4.0, 4.0)]  # (float64 sums of 99 x 79 bins still fit LDS)
    cols = [_pack_torture(e, rng, 80_000) for e in edges]
    n = min(c.shape[1] for c in cols)
    x, y = cols[0][:, :n].copy(), np.roll(cols[1][:, :n], 4099, axis=1)
    for w in (None, rng.uniform(0, 1, x.shape)):
        got, desc = _run(xh, [x, y], edges, w, True)
        assert "scan=8" in desc and "family=fast" in desc, desc
        assert_hist_equal(got, onp.bincount_rows([x, y], edges, w), w is not None)


@pytest.mark.parametrize("weights", [None, "f64", "f32"])
@pytest.mark.parametrize("kind", sorted(_PACK_EDGES))
def test_packed_bucket_entries_float32_samples(xh, kind, weights):
    """float32 samples: thresholds = smallest float32 >= e_j (> e_last for the last edge), exact without a redo path; the
    samples are the float32 values on and around every edge's float32 neighbourhood"""
    rng = np.random.default_rng(7 + sum(map(ord, kind)))
    edges = [_PACK_EDGES[kind](rng)]
    with np.errstate(over="ignore", invalid="ignore"):
        x = _pack_torture(edges[0], rng, 100_000).astype(np.float32)
    w = None if weights is None else rng.uniform(-1, 2, x.shape).astype(np.float64 if weights == "f64" else np.float32)
    want = onp.bincount_rows([x], edges, w)
    got, desc = _run(xh, [x], edges, w, True, pack=1)
    assert_hist_equal(got, want, w is not None)
    offered = any("scan=%d" % k in desc for k in (6, 7, 8))
    if kind in _PACK_KEY_MAP:
        assert "scan=8" in desc and "f32thr" in desc, desc
    elif kind not in _PACK_NOT_OFFERED and kind not in _PACK_EITHER and kind not in ("big_magnitude", "small_magnitude"):
        assert offered and "f32thr" in desc, desc
    got, _ = _run(xh, [x], edges, w, True, pack=-1)
    assert_hist_equal(got, want, w is not None)
    got, _ = _run(xh, [x], edges, w, False)  # host route, automatic choice
    assert_hist_equal(got, want, w is not None)


@pytest.mark.parametrize("weights", [None, "f32"])
@pytest.mark.parametrize("dims", [2, 3])
def test_packed_bucket_entries_float32_joint(xh, dims, weights):
    rng = np.random.default_rng(500 + dims)
    nb = ([257, 257] if weights is None else [65, 90]) if dims == 2 else ([33, 17, 41] if weights is None else [17, 9, 21])
    edges = [_sorted_uniform(rng, k, -4.0, 4.0) for k in nb]
    edges[-1] = np.geomspace(1e-3, 4.0, nb[-1])  # one logarithmic axis: the general kernels
    cols = [_pack_torture(e, rng, 60_000).astype(np.float32) for e in edges]
    n = min(c.shape[1] for c in cols)
    samples = [np.roll(c[:, :n], 7919 * d, axis=1).copy() for d, c in enumerate(cols)]
    w = None if weights is None else rng.uniform(0, 1, samples[0].shape).astype(np.float32)
    want = onp.bincount_rows(samples, edges, w)
    got, desc = _run(xh, samples, edges, w, True)
    assert "scan=8" in desc and "f32thr" in desc and "family=fast" in desc, desc
    assert_hist_equal(got, want, w is not None)
    got, desc = _run(xh, samples, edges, w, True, pack=-1)
    assert "scan=8" not in desc, desc
    assert_hist_equal(got, want, w is not None)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("weights", [None, "f32"])
@pytest.mark.parametrize("shape,axis,family", [((300, 40, 50), 0, "lanes"), ((700, 33, 9), 0, "lanes"), ((40, 90, 37), 1, "lanes"),
                                               ((5000, 365), 1, "flat_rows"), ((4097, 20), 1, "flat_rows"), ((70_000, 7), 1, None)])
def test_packed_bucket_entries_in_the_row_per_lane_and_flat_rows_kernels(xh, shape, axis, family, weights, dtype):
    """geometric edges (a binary search in the round-3 tables) through the kernels that digitize sample by sample: packed
    entries on the float-bit-pattern grid, samples on and around the edges in the data's own precision"""
    rng = np.random.default_rng(600 + len(shape) + axis)
    edges = np.geomspace(1e-3, 8.0, 41)
    a = np.abs(rng.standard_normal(shape) * 2).astype(dtype)
    flat = a.reshape(-1)
    k = flat.size // 5
    pick = rng.choice(edges, size=k).astype(dtype)
    pick[::3] = np.nextafter(pick[::3], dtype(np.inf))
    pick[1::3] = np.nextafter(pick[1::3], dtype(-np.inf))
    flat[rng.integers(0, flat.size, k)] = pick
    flat[:3] = [np.nan, np.inf, -1.0]
    w = None if weights is None else rng.uniform(0, 1, shape).astype(np.float32)
    want, _ = onp.histogram(a, bins=edges, axis=axis, weights=w)
    got, _ = xh.histogram(_dev(a), bins=edges, axis=axis, weights=None if w is None else _dev(w))
    desc = _describe_last(xh, [_dev(a.reshape(1, -1)[:, :4])], [edges])
    if family:
        assert "family=%s" % family in desc and "scan=8" in desc, desc
    assert_hist_equal(got.cpu().numpy(), want, w is not None)
    plan = _plan_for(xh, [_dev(a.reshape(1, -1)[:, :4])], [edges])
    plan.set_param("pack", -1)
    try:
        got, _ = xh.histogram(_dev(a), bins=edges, axis=axis, weights=None if w is None else _dev(w))
        assert "scan=8" not in plan.describe(), plan.describe()
    finally:
        plan.set_param("pack", 0)
    assert_hist_equal(got.cpu().numpy(), want, w is not None)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("weights", ["none", "f32", "f64_one_sign", "f64_both_signs"])
@pytest.mark.parametrize("kind", ["geometric", "random", "symlog"])
def test_partitioned_mode_routes_with_packed_bucket_entries(xh, kind, weights, dtype):
    """histograms beyond LDS on non-uniform edges: the one-pass routing pass digitizes with the packed entries (general
    variant) instead of a binary search / a 3-edge scan; samples on and around every edge"""
    rng = np.random.default_rng(700 + len(kind))
    if kind == "geometric":
        edges = [np.geomspace(1e-3, 5.0, 301), np.geomspace(1e-2, 9.0, 281)]
    elif kind == "symlog":
        edges = [np.concatenate([-np.geomspace(5.0, 1e-3, 150), [0.0], np.geomspace(1e-3, 5.0, 150)]), np.geomspace(1e-2, 9.0, 281)]
    else:
        edges = [_sorted_uniform(rng, 257, -4.0, 4.0), _sorted_uniform(rng, 300, -4.0, 4.0)]
    cols = [_pack_torture(e, rng, 150_000) for e in edges]
    n = min(c.shape[1] for c in cols)
    with np.errstate(over="ignore", invalid="ignore"):
        samples = [np.roll(c[:, :n], 7919 * d, axis=1).astype(dtype) for d, c in enumerate(cols)]
    w = {"none": None, "f32": rng.uniform(0, 1, (1, n)).astype(np.float32), "f64_one_sign": rng.uniform(0, 1, (1, n)),
         "f64_both_signs": rng.uniform(-1, 1, (1, n))}[weights]
    want = onp.bincount_rows(samples, edges, w)
    got, desc = _run(xh, samples, edges, w, True, partition=1)
    assert "hist=partitioned" in desc and "route=fused" in desc and "scan=8" in desc, desc
    assert_hist_equal(got, want, w is not None)
    got, desc = _run(xh, samples, edges, w, True, partition=1, pack=-1)
    assert "scan=8" not in desc, desc
    assert_hist_equal(got, want, w is not None)


# ---------------------------------------------------------------------------------------------
# float32 samples on arithmetic edges, digitized in float32 arithmetic (bin_arith32_fast, scan=9; round 5)
# ---------------------------------------------------------------------------------------------
def _f32_boundary_torture(edges, rng, n_random):
    """float32 samples on every float32 bin boundary (the smallest float32 >= e_j), on its float32 predecessor and successor,
    two ulps either side, plus randoms well beyond the range and the specials"""
    e = np.asarray(edges, dtype=np.float64)
    b = e.astype(np.float32)
    b = np.where(b.astype(np.float64) < e, np.nextafter(b, np.float32(np.inf)), b).astype(np.float32)
    parts = [b]
    for k in (1, 2):
        lo, hi = b.copy(), b.copy()
        for _ in range(k):
            lo, hi = np.nextafter(lo, np.float32(-np.inf)), np.nextafter(hi, np.float32(np.inf))
        parts += [lo, hi]
    span = float(e[-1] - e[0])
    parts += [rng.uniform(e[0] - 0.2 * span, e[-1] + 0.2 * span, n_random).astype(np.float32),
              np.array([np.nan, np.inf, -np.inf, 0.0, -0.0, 3.0e38, -3.0e38, 1e-45, e[0], e[-1]], dtype=np.float32)]
    x = np.concatenate(parts).astype(np.float32)
    rng.shuffle(x)
    return x.reshape(1, -1)


@pytest.mark.parametrize("mode", ["auto", "forced", "off"])
@pytest.mark.parametrize("lo,hi,nb", [(-4.0, 4.0, 50), (-4.0, 4.0, 100), (0.0, 1.0, 1000), (0.1, 0.7, 333), (-123.456, -123.0, 40),
                                      (1e6, 1e6 + 64.0, 64), (-1e-3, 3e-3, 77), (-3.7, 5.1, 20000), (250.0, 330.0, 7)])
def test_float32_arithmetic_digitize_on_and_next_to_every_boundary(xh, lo, hi, nb, mode):
    """BASELINE C4's digitize (float32 samples, np.linspace edges): decided by one float32 fma unless the sample is within
    delta bins of a float32 bin boundary; a sample on every boundary, on its neighbours and two ulps away must land where
    searchsorted puts the float64 image of the sample (core.py:163-174, numpy >= 2 compare rules)"""
    rng = np.random.default_rng(nb)
    edges = [np.linspace(lo, hi, nb + 1)]
    x = _f32_boundary_torture(edges[0], rng, 300_000)
    params = {"auto": {}, "forced": {"arith32": 1}, "off": {"arith32": -1}}[mode]
    want = onp.bincount_rows([x], edges)
    got, desc = _run(xh, [x], edges, None, True, **params)
    if mode == "off":
        assert "scan=9" not in desc, desc
    elif lo == 1e6:  # bins of 16 float32 ulps: the predecessor of a boundary sits 1/16 of a bin below it, delta = 1/8: not offered
        assert "scan=9" not in desc, desc
    else:
        assert "scan=9" in desc, desc
    np.testing.assert_array_equal(got, want, err_msg=desc)
    w = rng.uniform(0.5, 1.5, x.shape).astype(np.float32)
    got, desc = _run(xh, [x], edges, w, True, **params)
    assert_hist_equal(got, onp.bincount_rows([x], edges, w), True)
    # many rows (C4's shape in small): the long-tile variant and the row-wise geometry
    xr = _f32_boundary_torture(edges[0], rng, 150_000)
    cols = xr.shape[1] // 70
    xr = np.ascontiguousarray(xr[0, : 70 * cols].reshape(70, cols))
    got, desc = _run(xh, [xr], edges, None, True, **params)
    np.testing.assert_array_equal(got, onp.bincount_rows([xr], edges), err_msg=desc)
    if (lo, hi, nb) == (-4.0, 4.0, 50):  # C4's own shape class: >= 64 rows of >= 2^19 samples take 32 samples per lane and tile
        xl = np.tile(_f32_boundary_torture(edges[0], rng, (1 << 19) + 4321), (64, 1))
        for r in range(64):
            xl[r] = np.roll(xl[r], 977 * r)
        xl[5, ::3] = np.nan
        got, desc = _run(xh, [xl], edges, None, True, **params)
        assert "unroll=8" in desc and (mode == "off" or "scan=9" in desc), desc
        np.testing.assert_array_equal(got, onp.bincount_rows([xl], edges), err_msg=desc)


@pytest.mark.parametrize("weights", ["none", "f32", "f64"])
def test_float32_arithmetic_digitize_joint_histograms(xh, weights):
    """the same digitize per dimension of a 2-D / 3-D joint histogram of float32 samples (LDS-resident histograms)"""
    rng = np.random.default_rng(17)
    for dims, nbs in ((2, (50, 30)), (3, (12, 9, 7))):
        edges = [np.linspace(-2.0 - d, 3.0 + d, nb + 1) for d, nb in enumerate(nbs)]
        n = 200_000
        xs = [_f32_boundary_torture(e, rng, n)[:, :n] for e in edges]
        w = {"none": None, "f32": rng.uniform(0, 2, (1, n)).astype(np.float32), "f64": rng.standard_normal((1, n))}[weights]
        got, desc = _run(xh, xs, edges, w, True, arith32=1)
        assert "scan=9" in desc, desc
        assert_hist_equal(got, onp.bincount_rows(xs, edges, w), w is not None)


def test_float32_arithmetic_digitize_is_not_offered_for_bins_float32_cannot_resolve(xh):
    """bins narrower than a few float32 ulps: plan creation measures delta >= 1/8 (or boundaries that coincide) and the plan
    keeps the threshold tables / the float64 arithmetic — results stay exact"""
    rng = np.random.default_rng(3)
    edges = [np.linspace(1000.0, 1000.01, 401)]  # 2.5e-5 per bin at magnitude 1000: float32 ulp is 6.1e-5
    x = _f32_boundary_torture(edges[0], rng, 100_000)
    got, desc = _run(xh, [x], edges, None, True, arith32=1)
    assert "scan=9" not in desc, desc
    np.testing.assert_array_equal(got, onp.bincount_rows([x], edges))
