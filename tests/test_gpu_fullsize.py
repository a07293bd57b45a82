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
    partitioned multi-pass mode beyond 2^32 samples"""
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
    y.clamp_(-3.999, 3.999)  # y never drops a sample: the x-marginal needs no mask (x still drops ~6e-5 of them)
    x[n - 1] = 4.0
    y[n - 1] = 3.5
    e = np.linspace(-4.0, 4.0, 1025)
    counts, _ = xh.histogram(x, y, bins=[e, e], weights=w)
    plan = xh._get_plan([e, e], _native.CMP_F64, 0)
    assert "partitioned" in plan.describe(), plan.describe()
    assert tuple(counts.shape) == (1024, 1024) and counts.dtype == torch.float64
    # total = weight of the samples inside the range (float64 sums: 1e-6 relative is the contract; observed ~1e-12)
    tot = torch.zeros((), dtype=torch.float64, device=dev)
    for i0 in range(0, n, 500_000_000):
        xs, ws = x[i0:i0 + 500_000_000], w[i0:i0 + 500_000_000]
        tot += torch.where((xs >= -4.0) & (xs <= 4.0), ws, torch.zeros((), dtype=torch.float64, device=dev)).sum()
    np.testing.assert_allclose(float(counts.sum()), float(tot), rtol=1e-9)
    # marginal over y = the 1-D weighted histogram of x (LDS kernel family: an independent path)
    hx, _ = xh.histogram(x, bins=e, weights=w)
    np.testing.assert_allclose(counts.sum(dim=1).cpu().numpy(), hx.cpu().numpy(), rtol=1e-9)
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
        xs = x[i0:i0 + 500_000_000]
        inside += int(((xs >= -4.0) & (xs <= 4.0)).sum())
    assert int(c1.sum()) == inside
    del x, y, c1, c2
    torch.cuda.empty_cache()


@pytest.mark.parametrize("signs", ["one", "both"])
def test_partitioned_mode_sub_batches_on_two_streams_give_the_serial_result(xh, signs):
    """the "overlap" form of the partitioned mode (routing pass of piece k + 1 under the adding-up pass of piece k, two record
    pools in turn, fork / join by events; measured slower and off by default — DESIGN 4.2) must stay CORRECT: same
    histogram as the serial form, for packed records (one sign) and for the exact redo (both signs)"""
    if _free_gb() < 20:
        pytest.skip("needs 20 GB of free device memory")
    from xhistogram_amd import _native

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    n = 150_000_000
    x = torch.empty(n, dtype=torch.float64, device=dev).normal_(generator=g)
    y = torch.empty(n, dtype=torch.float64, device=dev).normal_(generator=g)
    w = torch.empty(n, dtype=torch.float64, device=dev).uniform_(generator=g)
    if signs == "both":
        w -= 0.5
    edges = [np.linspace(-4.0, 4.0, 1025)] * 2
    plan = xh._get_plan(edges, _native.CMP_F64, 0)
    want, _ = xh.histogram(x, y, bins=edges, weights=w)
    assert "pieces=1" in plan.describe(), plan.describe()
    try:
        for pieces, cus in ((4, 48), (7, 32)):
            plan.set_param("overlap", pieces)
            plan.set_param("overlap_cus", cus)
            got, _ = xh.histogram(x, y, bins=edges, weights=w)
            desc = plan.describe()
            assert "pieces=%d" % pieces in desc and "second stream" in desc, desc
            torch.testing.assert_close(got, want, rtol=1e-9, atol=1e-6 if signs == "both" else 0.0)
    finally:
        plan.set_param("overlap", 0)
        plan.set_param("overlap_cus", 0)
    # a prefix against the oracle through the same overlapped call shape would need >= 6.7e7 samples: the sum stands in
    assert abs(float(want.sum()) - float(w[(x >= -4) & (x <= 4) & (y >= -4) & (y <= 4)].sum())) <= 1e-6 * float(w.abs().sum())
