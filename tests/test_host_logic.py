"""Host-side logic of xhistogram_amd.core that needs no GPU: argument formatters (the reference's
tables, test_core.py:316-362), the [rows, cols] arrangement (views, no copies), compare-domain
selection (numpy's promotion rules), strided views handed to the C ABI."""
import numpy as np
import pytest

from oracle import oracle_np as onp
from xhistogram_amd import _native, core

bins_int, bins_str, bins_arr, range_ = 10, "auto", np.linspace(-4, 4, 10), (0, 1)


@pytest.mark.parametrize(
    "bins_in,n,expected",
    [
        (bins_int, 1, [bins_int]), (bins_str, 1, [bins_str]), (bins_arr, 1, [bins_arr]), ([bins_int], 1, [bins_int]),
        (bins_int, 2, 2 * [bins_int]), (bins_str, 2, 2 * [bins_str]), (bins_arr, 2, 2 * [bins_arr]),
        ([bins_int, bins_str, bins_arr], 3, [bins_int, bins_str, bins_arr]),
        ([bins_arr], 2, None), (None, 1, None), ([bins_arr, bins_arr], 1, None),
    ],
)
def test_format_bins(bins_in, n, expected):  # test_core.py:316-340
    if expected is None:
        with pytest.raises((ValueError, TypeError)):
            core._ensure_correctly_formatted_bins(bins_in, n)
    else:
        got = core._ensure_correctly_formatted_bins(bins_in, n)
        assert len(got) == len(expected) and all(g is e or g == e for g, e in zip(got, expected))


@pytest.mark.parametrize(
    "range_in,n,expected",
    [(range_, 1, [range_]), (range_, 2, [range_, range_]), ([range_, range_], 2, [range_, range_]),
     ([(range_[0],)], 1, None), ([range_], 2, None), ([range_, range_], 1, None)],
)
def test_format_range(range_in, n, expected):  # test_core.py:343-362
    if expected is None:
        with pytest.raises(ValueError):
            core._ensure_correctly_formatted_range(range_in, n)
    else:
        assert core._ensure_correctly_formatted_range(range_in, n) == expected
    assert core._ensure_correctly_formatted_range(None, 3) == [None, None, None]


def test_rows_cols_matches_reference_layout_without_copies():
    x = np.arange(3 * 4 * 5 * 6, dtype=np.float64).reshape(3, 4, 5, 6)
    for axis in ([3], [2, 3], [0], [1, 3], [3, 1], [0, 1, 2], [2, 0]):
        got = core._rows_cols(x, axis, False)
        np.testing.assert_array_equal(got, onp.to_rows_cols(x, axis))
    assert np.shares_memory(core._rows_cols(x, [2, 3], False), x)  # trailing axes: a view (C4's case)
    assert np.shares_memory(core._rows_cols(x, None, True), x)
    w = np.broadcast_to(np.arange(6.0), (3, 4, 5, 6))  # stride-0 weights stay stride-0
    v = core._rows_cols(w, [3], False)
    assert v.shape == (60, 6) and v.strides == (0, 8)
    ptr, tag, rs, cs, _ir, _os, keep = core._strided_view(v, "numpy")
    assert (tag, rs, cs) == (_native.F64, 0, 1) and ptr == w.ctypes.data
    col = np.broadcast_to(np.arange(5.0)[:, None], (5, 7))  # column broadcast: cs == 0
    assert core._strided_view(col, "numpy")[2:4] == (1, 0)
    neg = x[0, 0][:, ::-1]  # negative stride: one contiguous copy
    ptr, tag, rs, cs, _ir, _os, keep = core._strided_view(neg, "numpy")
    assert (rs, cs) == (6, 1) and not np.shares_memory(keep, x)
    np.testing.assert_array_equal(keep, neg)


def test_compare_domain_follows_numpy_promotion():
    f64, f32, i64, i32, u8 = (np.dtype(t) for t in (np.float64, np.float32, np.int64, np.int32, np.uint8))
    e_f = np.linspace(0, 1, 3)
    e_i = np.array([0, 5, 10])
    assert core._compare_domain([f64], [e_f])[0] == _native.CMP_F64
    assert core._compare_domain([f32], [e_f])[0] == _native.CMP_F64
    assert core._compare_domain([i64], [e_f])[0] == _native.CMP_F64  # numpy rounds int64 to float64 here too
    assert core._compare_domain([f32], [e_i])[0] == _native.CMP_F64
    dom, conv, _ = core._compare_domain([i64], [e_i])
    assert dom == _native.CMP_I64 and conv[0].dtype == np.int64
    # <= 32-bit integer samples against integer edges within +-2^53: exact in float64, taken there
    dom, conv, _ = core._compare_domain([i32, u8], [e_i, e_i.astype(np.int16)])
    assert dom == _native.CMP_F64 and all(c.dtype == np.float64 for c in conv)
    assert core._compare_domain([i32, i64], [e_i, e_i])[0] == _native.CMP_I64       # a 64-bit input keeps int64
    assert core._compare_domain([i32], [np.array([0, (1 << 53) + 1])])[0] == _native.CMP_I64  # edge not exact in float64
    assert core._compare_domain([i32, f64], [e_i.astype(np.int32), e_f])[0] == _native.CMP_F64  # small ints are exact in f64
    # a 64-bit integer (or datetime) input next to a float one: per-input domains
    dom, conv, _ = core._compare_domain([i64, f64], [e_i, e_f])
    assert dom == (_native.CMP_PER_DIM | 0b01) and conv[0].dtype == np.int64 and conv[1].dtype == np.float64
    dom, conv, common = core._compare_domain([f32, np.dtype("datetime64[s]")], [e_f, np.array(["2000-01-01", "2001-01-01"], dtype="datetime64[D]")])
    assert dom == (_native.CMP_PER_DIM | 0b10) and common[1] == np.dtype("datetime64[s]") and conv[1].dtype == np.int64
    t = np.array(["2000-01-01", "2001-01-01"], dtype="datetime64[D]")
    dom, conv, common = core._compare_domain([np.dtype("datetime64[ns]")], [t])
    assert dom == _native.CMP_I64 and common[0] == np.dtype("datetime64[ns]")
    assert conv[0][0] == np.datetime64("2000-01-01", "ns").astype(np.int64)
    with pytest.raises(TypeError):
        core._compare_domain([np.dtype(np.complex128)], [e_f])
    with pytest.raises(TypeError):
        core._compare_domain([np.dtype("datetime64[ns]")], [e_f])
    # unsigned 64-bit on both sides: the int64 domain with the sign bit flipped
    dom, conv, _ = core._compare_domain([np.dtype(np.uint64)], [e_i.astype(np.uint64)])
    assert dom == (_native.CMP_I64 | _native.CMP_UNSIGNED) and conv[0].dtype == np.uint64
    assert core._compare_domain([u8, f64], [e_i.astype(np.uint64), e_f])[0] == (_native.CMP_PER_DIM | 0b01 | _native.CMP_UNSIGNED)
    with pytest.raises(NotImplementedError):  # one signedness per plan
        core._compare_domain([np.dtype(np.uint64), i64], [e_i.astype(np.uint64), e_i])


def test_axis_and_argument_errors_raise_before_compute():
    x = np.zeros((3, 4))
    with pytest.raises(AssertionError):
        core.histogram(x, bins=np.linspace(0, 1, 3), axis=2)
    with pytest.raises(ValueError):
        core.histogram(x, bins=None)
    with pytest.raises(ValueError):
        core.histogram(x, x, bins=[np.linspace(0, 1, 3)])
    with pytest.raises(ValueError):
        core.histogram(x, bins=np.array([0.0, 2.0, 1.0]))
    with pytest.raises(TypeError):
        core.histogram(x, bins="auto", weights=np.ones_like(x))


def test_default_device_env(monkeypatch):
    monkeypatch.delenv("XHIST_AMD_DEVICE", raising=False)
    monkeypatch.delenv("LOCAL_RANK", raising=False)
    assert core.default_device() == 0
    monkeypatch.setenv("LOCAL_RANK", "3")
    assert core.default_device() == 3
    monkeypatch.setenv("XHIST_AMD_DEVICE", "5")
    assert core.default_device() == 5


def test_collapse_describes_the_reference_layout_by_strides():
    """_collapse must address exactly the elements of the reference's moveaxis+reshape block
    (core.py:211-227), row by row; the order inside a row is free (histograms do not care)"""
    rng = np.random.default_rng(3)
    x = rng.standard_normal((3, 4, 5, 6))
    raw = x.reshape(-1)
    for axis in ([0], [1], [2], [3], [1, 2], [0, 1], [2, 3], [0, 1, 2], [0, 1, 2, 3]):
        full = len(axis) == 4
        d = core._collapse(x, axis, full, core._reduced_order(x, axis))
        assert d is not None, axis
        m, c, rs, cs, ir, os_ = d
        r = np.arange(m)
        off = (r // ir) * os_ + (r % ir) * rs if ir else r * rs
        got = raw[off[:, None] + np.arange(c)[None, :] * cs]
        want = onp.to_rows_cols(x, None if full else axis)
        np.testing.assert_array_equal(np.sort(got, axis=1), np.sort(want, axis=1))
    assert core._collapse(x, [1], False, [1])[4:] == (30, 120)  # middle axis: grouped rows, no copy
    for axis in ([0, 3], [1, 3], [0, 2]):  # reduced axes not adjacent in memory: the copying route
        assert core._collapse(x, axis, False, core._reduced_order(x, axis)) is None
    w = np.broadcast_to(rng.standard_normal((1, 1, 5, 1)), x.shape)  # broadcast weights stay stride-0
    assert core._collapse(w, [2], False, [2]) == (72, 5, 0, 1, 0, 0)
    assert core._collapse(w, [3], False, [3]) == (60, 6, 1, 0, 5, 0)
    f = np.asfortranarray(x)  # the reduced order follows the FIRST array's memory order
    assert core._reduced_order(f, [1, 2]) == [2, 1]
    assert core._collapse(x[:, ::-1], [3], False, [3]) is None  # negative stride
    v = core._view_of(x, core._collapse(x, [1], False, [1]), "numpy")
    assert v[0] == x.ctypes.data and v[2:6] == (1, 30, 30, 120)
    # a 2-stride host view the staging copy cannot take row-by-row is passed as one group
    y = x[:, :, :, ::2]
    d = core._collapse(y, [3], False, [3])
    assert d == (60, 3, 6, 2, 0, 0) and core._view_of(y, d, "numpy")[4:6] == (60, 0)


def _oracle_bincount(*all_arrays, weights=False, axis=None, bins=None, density=None, block_size=None):
    """stand-in for core._bincount with the same contract, computed by the oracle (CPU): lets the
    host-side rewrites around it run without a GPU"""
    arrays = list(all_arrays)
    w = arrays.pop() if weights else None
    nd = arrays[0].ndim
    ax = tuple(range(nd)) if axis is None else tuple(int(a) for a in axis)
    h, _ = onp.histogram(*arrays, bins=bins if len(arrays) > 1 else bins[0], weights=w, axis=ax)
    kept = tuple(1 if i in ax else arrays[0].shape[i] for i in range(nd))
    return np.asarray(h).reshape(kept + tuple(len(b) - 1 for b in bins))


def test_host_side_rewrites_around_the_block_adapter(monkeypatch):
    """weights constant along reduced axes -> counts then weights; non-adjacent reduced axes -> two steps.
    Both only re-arrange calls of the block adapter, so with the adapter replaced by the oracle they can be
    checked on the CPU against the reference semantics (materialised weights, moveaxis + reshape)."""
    calls = []

    def spy(*a, **k):
        calls.append((k.get("weights"), tuple(k.get("axis") or ())))
        return _oracle_bincount(*a, **k)

    monkeypatch.setattr(core, "_bincount", spy)
    rng = np.random.default_rng(3)
    t = rng.standard_normal((5, 12, 300))
    t[1, 2, :4] = np.nan
    e = np.linspace(-3, 3, 13)
    w_lat = np.cos(np.linspace(-1, 1, 12)).reshape(1, 12, 1)
    for axis in ((1, 2), None, (0, 2), (2,)):
        calls.clear()
        got, _ = core.histogram(t, bins=e, weights=w_lat, axis=axis)
        want, _ = onp.histogram(t, bins=e, weights=w_lat, axis=axis)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
        assert calls and calls[0][0] is False, calls           # the samples were counted unweighted
    calls.clear()
    got, _ = core.histogram(t, bins=e, weights=np.broadcast_to(w_lat, t.shape), axis=(1, 2), density=True)
    np.testing.assert_allclose(got, onp.histogram(t, bins=e, weights=w_lat, axis=(1, 2), density=True)[0], rtol=1e-12)
    assert calls[0] == (False, (2,)), calls                    # a stride-0 view counts as a size-1 axis
    # NaN weight on a latitude nobody... every latitude has samples here, so the NaN must show where it lands
    w_nan = w_lat.copy()
    w_nan[0, 3, 0] = np.nan
    got, _ = core.histogram(t, bins=e, weights=w_nan, axis=(1, 2))
    want, _ = onp.histogram(t, bins=e, weights=w_nan, axis=(1, 2))
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    # full-size weights: not rewritten
    calls.clear()
    w_full = rng.uniform(0, 1, t.shape)
    got, _ = core.histogram(t, bins=e, weights=w_full, axis=(1, 2))
    np.testing.assert_allclose(got, onp.histogram(t, bins=e, weights=w_full, axis=(1, 2))[0], rtol=1e-12)
    assert calls == [(True, (1, 2))], calls
    # non-adjacent reduced axes: last adjacent block through the adapter, the rest summed
    calls.clear()
    got, _ = core.histogram(t, bins=e, axis=(0, 2))
    np.testing.assert_array_equal(got, onp.histogram(t, bins=e, axis=(0, 2))[0])
    assert got.dtype == np.int64 and calls == [(False, (2,))], calls
    calls.clear()
    q = rng.standard_normal((3, 4, 5, 200))
    wq = rng.uniform(0, 1, q.shape)
    got, _ = core.histogram(q, bins=e, axis=(0, 2, 3), weights=wq)
    assert calls == [(True, (2, 3))], calls
    np.testing.assert_allclose(got, onp.histogram(q, bins=e, axis=(0, 2, 3), weights=wq)[0], rtol=1e-12)


def test_overlapping_rows_and_unaligned_fields_take_the_copying_route():
    """Layouts the pitched staging copy cannot express: sliding windows (rows overlap in memory)
    and a float64 field of a packed record whose byte stride is no multiple of 8."""
    x = np.arange(40.0)
    win = np.lib.stride_tricks.sliding_window_view(x, 8)  # (33, 8), strides (8, 8) bytes
    d = core._collapse(win, [1], False, [1])
    assert d == (33, 8, 1, 1, 0, 0)
    assert core._view_of(win, d, "numpy") is None
    d0 = core._collapse(win, [0], False, [0])  # histogram over the window START axis: same overlap, transposed
    assert core._view_of(win, d0, "numpy") is None
    ok = x.reshape(5, 8)
    assert core._view_of(ok, core._collapse(ok, [1], False, [1]), "numpy") is not None
    rec = np.zeros(7, dtype=np.dtype([("a", "<f8"), ("b", "<i4")]))  # itemsize 12
    rec["a"] = np.arange(7.0)
    col = rec["a"].reshape(7, 1)
    ptr, tag, rs, cs, _, _, keep = core._strided_view(col, "numpy")
    assert keep is not col and keep.flags.c_contiguous and (rs, cs) == (1, 1)
    np.testing.assert_array_equal(keep.ravel(), np.arange(7.0))


def test_bin_width_estimators_restated_from_moments_match_numpy(monkeypatch):
    """VERDICT r2 "next" #7 (f-1): bins="sqrt" | "sturges" | "rice" | "scott" on device-resident data come from one fused
    count / min / max / mean / M2 reduction (xhist_moments) instead of a host copy of the array.  Here the reduction is
    a numpy double, so what is pinned is the host logic around it: the edges must be BIT-identical to
    np.histogram_bin_edges (core.py:383-388) — same dtype, same errors — for float64 / float32 / integer data, with and
    without a range, or the function must decline (None = numpy decides on a host copy)."""
    from xhistogram_amd import _native, core

    cur = {}

    def fake_moments(view, nr, nc, lo, hi, want_m2, dev, stream):
        x = cur["a"].ravel().astype(np.float64)
        if lo is not None:
            x = x[(x >= lo) & (x <= hi)]
        if x.size == 0:
            return 0, np.inf, -np.inf, np.nan, np.nan
        return x.size, x.min(), x.max(), x.mean(), ((x - x.mean()) ** 2).sum()

    monkeypatch.setattr(_native, "moments", fake_moments)
    monkeypatch.setattr(core, "_strided_view", lambda flat, backend: (0, 0, 0, 1, 0, 0, None))

    class Resident:
        def __init__(self, a):
            self.size, self.device, self.shape = a.size, 0, (1, a.size)

        def reshape(self, *shape):
            return self

    monkeypatch.setattr(core, "_is_devarr", lambda a: isinstance(a, Resident))
    rng = np.random.default_rng(0)
    declined = checked = 0
    for dt in (np.float64, np.float32, np.int32, np.uint8, np.int64):
        for trial in range(40):
            n = int(rng.integers(1, 5000))
            if np.dtype(dt).kind == "f":
                a = (rng.standard_normal(n) * rng.uniform(0.1, 100)).astype(dt)
            else:
                a = rng.integers(0 if np.dtype(dt).kind == "u" else -50, 200, n).astype(dt)
            if trial % 7 == 0:
                a[:] = a[0]
            if trial % 13 == 5 and np.dtype(dt).kind == "f":
                a[0] = np.nan
            cur["a"] = a
            for name in ("sqrt", "sturges", "rice", "scott"):
                for r in (None, (-1.0, 2.5), (0, 100), (5, 5), (np.float32(0.1), np.float32(7.3)), (2, 1)):
                    try:
                        want = np.histogram_bin_edges(a, bins=name, range=r)
                    except Exception as exc:  # noqa: BLE001 - compared below
                        want = (type(exc), str(exc))
                    try:
                        got = core._device_estimator_edges(Resident(a), name, r, np.dtype(dt), True)
                    except Exception as exc:  # noqa: BLE001
                        got = (type(exc), str(exc))
                    if got is None:
                        declined += 1
                        continue
                    checked += 1
                    if isinstance(want, tuple) or isinstance(got, tuple):
                        assert want == got, (dt, name, r)
                    else:
                        assert want.dtype == got.dtype and want.shape == got.shape, (dt, name, r, want.shape, got.shape)
                        np.testing.assert_array_equal(got, want, err_msg=str((dt, name, r)))
    assert checked > 10 * declined  # declining (constant data under "scott", ties) is the exception
    for other in ("fd", "auto", "doane", "stone"):
        assert core._device_estimator_edges(Resident(np.zeros(3)), other, None, np.dtype("f8"), True) is None


def test_moments_of_shards_combine_to_the_moments_of_the_whole():
    """sharded inputs (multigpu): every GPU reduces its shard to (n, min, max, mean, M2); the host combines them"""
    from xhistogram_amd import core

    rng = np.random.default_rng(1)
    x = rng.standard_normal(10_000) * 7 + 3
    cuts = [0, 1, 1, 4000, 4001, 10_000]
    parts = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        p = x[lo:hi]
        parts.append((p.size, p.min() if p.size else np.inf, p.max() if p.size else -np.inf, p.mean() if p.size else np.nan,
                      ((p - p.mean()) ** 2).sum() if p.size else np.nan))
    n, mn, mx, mean, m2 = core.combine_moments(parts)
    assert n == x.size and mn == x.min() and mx == x.max()
    np.testing.assert_allclose(mean, x.mean(), rtol=1e-13)
    np.testing.assert_allclose(m2, ((x - x.mean()) ** 2).sum(), rtol=1e-12)
    assert core.combine_moments([(0, np.inf, -np.inf, np.nan, np.nan)])[0] == 0
    nan_part = (3, np.nan, np.nan, np.nan, np.nan)
    assert np.isnan(core.combine_moments([parts[0], nan_part, parts[3]])[1])



def test_public_signatures_are_the_references():
    """north_star: "exact signatures".  tests/golden/manifest.json holds str(inspect.signature(...)) of the reference's
    histogram / _bincount / _bincount_2d_vectorized (core.py:250-258, :197-199, :137-139) and xarray.histogram's
    (xarray.py:13-23, from its source), written by make_golden.py in the build container"""
    import inspect
    import json
    import os

    from xhistogram_amd import xarray as xh_xarray

    want = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "manifest.json")))["signatures"]
    got = {"core.%s" % n: str(inspect.signature(getattr(core, n))) for n in ("histogram", "_bincount", "_bincount_2d_vectorized")}
    got["xarray.histogram"] = str(inspect.signature(xh_xarray.histogram))
    assert got == want


def test_range_cut_follows_numpy_promotion_per_bound():
    """float32 data meets a Python-float bound in float32 (weak scalar) but an np.float64 / np.int64 bound in
    float64 — the number of elements numpy's `keep` mask leaves is the number inside core._range_cut's bounds"""
    rng = np.random.default_rng(3)
    a = np.clip(rng.uniform(0.0, 1.2, 1000), 0.7, 1.1).astype(np.float32)
    for r in ((0.7, 1.0), (np.float64(0.7), 1.0), (np.float32(0.7), 1.0), (0.7, np.float64(1.0)), (np.int64(0), 1.1), (np.float16(0.7), np.float32(1.1))):
        keep = int(((a >= r[0]) & (a <= r[1])).sum())  # numpy's own compare, promotion included
        lo, hi = core._range_cut(r, np.dtype(np.float32))
        assert int(((a.astype(np.float64) >= lo) & (a.astype(np.float64) <= hi)).sum()) == keep, r
    assert core._range_cut((np.float64(0.7), 1.0), np.dtype(np.float32))[0] == 0.7
    assert core._range_cut((0.7, 1.0), np.dtype(np.float32))[0] == float(np.float32(0.7))
    assert core._range_cut((0.7, 1.0), np.dtype(np.float64)) == (0.7, 1.0)
