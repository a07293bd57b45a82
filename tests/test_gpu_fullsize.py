"""BASELINE.json configs C4 and C5 at their FULL size on one MI355X (both fit its 288 GB): the oracle cannot
run 4*10^9 samples, so parity at this size goes through size-independent properties (run-to-run equality of
integer counts, row sums = in-range counts, total = in-range weight, marginals against 1-D histograms) plus
an oracle comparison of a prefix / a few rows computed by the SAME full-size call."""
import numpy as np
import pytest

from conftest import assert_hist_equal
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def xh():
    from xhistogram_amd import _native, core

    assert _native.device_count() >= 1
    return core


def _free_gb():
    free, _total = torch.cuda.mem_get_info(0)
    return free / 2**30


def test_c4_full_size_time_lat_lon(xh):
    """(3650, 720, 1440) f32 = 15.1 GB, 50 bins over lat / lon (C4 on ONE GPU)"""
    if _free_gb() < 40:
        pytest.skip("needs 40 GB of free device memory")
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    T, LAT, LON = 3650, 720, 1440
    x = torch.empty((T, LAT, LON), dtype=torch.float32, device=dev).normal_(generator=g)
    x[17, 3, :100] = float("nan")
    x[3649, 719, 1439] = 4.0  # the right edge belongs to the last bin
    edges = np.linspace(-4.0, 4.0, 51)
    h, _ = xh.histogram(x, bins=edges, axis=(1, 2))
    assert tuple(h.shape) == (T, 50) and h.dtype == torch.int64
    # every time step: the counts add up to the samples inside [-4, 4] (NaN and out-of-range dropped)
    inside = torch.zeros(T, dtype=torch.int64, device=dev)
    for t0 in range(0, T, 365):
        blk = x[t0:t0 + 365]
        inside[t0:t0 + 365] = ((blk >= -4.0) & (blk <= 4.0)).sum(dim=(1, 2))
    assert torch.equal(h.sum(dim=1), inside)
    # run-to-run: integer counts do not depend on scheduling
    h2, _ = xh.histogram(x, bins=edges, axis=(1, 2))
    assert torch.equal(h, h2)
    # rows against the oracle (first, last, the NaN row, shard boundaries of an 8-way split)
    rows = [0, 17, 455, 456, 457, 1824, 3648, 3649]
    want = onp.histogram(x[rows].cpu().numpy(), bins=edges, axis=(1, 2))[0]
    np.testing.assert_array_equal(h[rows].cpu().numpy(), want)
    # the same rows through a time-chunked call (what a rank of the 8-GPU job holds)
    hs, _ = xh.histogram(x[456:912], bins=edges, axis=(1, 2))
    assert torch.equal(hs, h[456:912])
    del x, h, h2
    torch.cuda.empty_cache()


def test_c5_full_size_4e9_samples_weighted_density(xh):
    """4*10^9 samples (x, y, w float64 = 96 GB), 1024 x 1024 bins, weighted, density (C5 on ONE GPU): the
    exchange mode and the partitioned multi-pass mode beyond 2^32 samples"""
    if _free_gb() < 240:
        pytest.skip("needs 240 GB of free device memory")
    from xhistogram_amd import _native

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    n = 4_000_000_000
    x = torch.empty(n, dtype=torch.float64, device=dev).normal_(generator=g)
    y = torch.empty(n, dtype=torch.float64, device=dev).normal_(generator=g)
    w = torch.empty(n, dtype=torch.float64, device=dev).uniform_(generator=g)
    # (VERDICT r4 "next" #3: y is NOT clamped any more — both inputs drop ~6e-5 of the samples on their own, and y carries
    # NaNs, infinities and far-out values in three stretches, one of them beyond index 2^32)
    for i0 in (12_345, 2_000_000_001, n - 70_000):
        y[i0:i0 + 20_000] = float("nan")
        y[i0 + 20_000:i0 + 40_000] = 11.5
        y[i0 + 40_000:i0 + 50_000] = float("-inf")
    x[n - 1] = 4.0
    y[n - 1] = 3.5
    x[n - 2] = 0.25
    y[n - 2] = 4.0  # the right edge of y belongs to its last bin
    e = np.linspace(-4.0, 4.0, 1025)
    counts, _ = xh.histogram(x, y, bins=[e, e], weights=w)
    plan = xh._get_plan([e, e], _native.CMP_F64, 0)
    assert "partitioned" in plan.describe(), plan.describe()
    assert tuple(counts.shape) == (1024, 1024) and counts.dtype == torch.float64
    # which kernels: the exchange mode is offered and its probe finds ~94 % of these samples in its window, so it took
    # the call above (the next call's description carries what the GPU reported); the classic passes on the same 96 GB agree
    assert "exchange=if the probe" in plan.describe(), plan.describe()
    aborts_before = int(plan.describe().split("exchange_aborts=")[1].split()[0])  # (other tests of this process may have made the mode give up on purpose)
    torch.cuda.synchronize()
    plan.set_param("exchange", -1)
    try:
        classic, _ = xh.histogram(x, y, bins=[e, e], weights=w)
        desc = plan.describe()
    finally:
        plan.set_param("exchange", 0)
    assert "exchange=no" in desc and int(desc.split("exchange_window_ppm_before=")[1].split()[0]) >= 880_000, desc
    assert int(desc.split("exchange_aborts=")[1].split()[0]) == aborts_before, desc  # the 96 GB call did not give up
    torch.testing.assert_close(counts, classic, rtol=2.0 ** -34, atol=0)  # (both round the weights to 36 mantissa bits, then add in float64)
    del classic
    # total = weight of the samples inside the range of BOTH inputs (float64 sums: 1e-6 relative is the contract; observed ~1e-12)
    # both marginals = the 1-D weighted histograms of one input over the samples the OTHER one keeps (LDS kernel family: an
    # independent path); the weights of dropped samples are zeroed in a copy rather than masked out (no 32 GB gathers)
    tot = torch.zeros((), dtype=torch.float64, device=dev)
    wx = w.clone()  # weights of the samples y keeps
    wy = w.clone()  # ... x keeps
    for i0 in range(0, n, 500_000_000):
        sl = slice(i0, i0 + 500_000_000)
        okx = (x[sl] >= -4.0) & (x[sl] <= 4.0)
        oky = (y[sl] >= -4.0) & (y[sl] <= 4.0)
        wx[sl] = torch.where(oky, wx[sl], torch.zeros((), dtype=torch.float64, device=dev))
        wy[sl] = torch.where(okx, wy[sl], torch.zeros((), dtype=torch.float64, device=dev))
        tot += torch.where(okx & oky, w[sl], torch.zeros((), dtype=torch.float64, device=dev)).sum()
        del okx, oky
    np.testing.assert_allclose(float(counts.sum()), float(tot), rtol=1e-9)
    hx, _ = xh.histogram(x, bins=e, weights=wx)
    np.testing.assert_allclose(counts.sum(dim=1).cpu().numpy(), hx.cpu().numpy(), rtol=1e-9)
    del wx
    hy, _ = xh.histogram(y, bins=e, weights=wy)
    np.testing.assert_allclose(counts.sum(dim=0).cpu().numpy(), hy.cpu().numpy(), rtol=1e-9)
    del wy, hx, hy
    torch.cuda.empty_cache()
    # the whole against an independent restatement in torch ops over ALL 4*10^9 samples, bin by bin (bench.torch_reference:
    # bucketize + last-edge rule + joint index + bincount in 2^27-sample pieces; pinned to the oracle on the CPU)
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    import bench

    ref = bench.torch_reference(torch, [x, y], w, [e, e], 1, n, True).reshape(1024, 1024)
    torch.testing.assert_close(counts, ref, rtol=1e-6, atol=0)  # (packed 48-bit records: <= 2^-37 per weight; observed ~1e-11)
    assert bool(((ref == 0) == (counts == 0)).all())
    del ref
    # the last sample (index 4e9 - 1 > 2^32) landed in the last x bin, at y = 3.5
    by = int(np.searchsorted(e, 3.5, side="right") - 1)
    assert float(counts[1023, by]) > 0
    # density: integrates to one
    dens, _ = xh.histogram(x, y, bins=[e, e], weights=w, density=True)
    area = (8.0 / 1024) ** 2
    np.testing.assert_allclose(float(dens.sum()) * area, 1.0, rtol=1e-9)
    np.testing.assert_allclose((dens * float(counts.sum()) * area).cpu().numpy(), counts.cpu().numpy(), rtol=1e-9)
    # linearity: the halves add up to the whole (each half is itself beyond 2^31 samples)
    ha, _ = xh.histogram(x[: n // 2], y[: n // 2], bins=[e, e], weights=w[: n // 2])
    hb, _ = xh.histogram(x[n // 2:], y[n // 2:], bins=[e, e], weights=w[n // 2:])
    np.testing.assert_allclose((ha + hb).cpu().numpy(), counts.cpu().numpy(), rtol=1e-9)
    # a 3*10^6 prefix through the same (forced) mode against the oracle
    m = 3_000_000
    plan.set_param("partition", 1)
    try:
        hp, _ = xh.histogram(x[:m], y[:m], bins=[e, e], weights=w[:m])
        assert "partitioned" in plan.describe()
    finally:
        plan.set_param("partition", 0)
    want = onp.histogram(x[:m].cpu().numpy(), y[:m].cpu().numpy(), bins=[e, e], weights=w[:m].cpu().numpy())[0]
    assert_hist_equal(hp.cpu().numpy(), want, weighted=True)
    # unweighted counts at full size: exact, run-to-run identical, total = samples in range
    del w, ha, hb, dens
    torch.cuda.empty_cache()
    c1, _ = xh.histogram(x, y, bins=[e, e])
    c2, _ = xh.histogram(x, y, bins=[e, e])
    assert c1.dtype == torch.int64 and torch.equal(c1, c2)
    inside = 0
    for i0 in range(0, n, 500_000_000):
        xs, ys = x[i0:i0 + 500_000_000], y[i0:i0 + 500_000_000]
        inside += int(((xs >= -4.0) & (xs <= 4.0) & (ys >= -4.0) & (ys <= 4.0)).sum())
    assert int(c1.sum()) == inside
    assert torch.equal(bench.torch_reference(torch, [x, y], None, [e, e], 1, n, False).reshape(1024, 1024), c1)
    del x, y, c1, c2
    torch.cuda.empty_cache()
