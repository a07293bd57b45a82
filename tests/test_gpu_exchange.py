"""GPU parity of the exchange mode of the partitioned path (xhistogram_amd/csrc/xhist_exchange.hip.h): beyond-LDS histograms
of float64 samples with float64 weights whose records never go through HBM — one persistent workgroup per compute unit keeps
rows of a window of the histogram in LDS, records travel through rings inside each XCD.

What the reference computes for these calls is the digitize -> joint index -> bincount of
/root/reference/xhistogram/core.py:163-183 and :73-83; the oracle restates it.  The mode is forced here ("exchange" = 1:
any size, any window coverage) so that sizes the oracle finishes in seconds reach the kernel; by default it is taken from
2^25 samples on when the window the probe picks holds 88 % of them.
"""
import numpy as np
import pytest

from conftest import assert_hist_equal
from oracle import oracle_np as onp
from test_gpu_parity import _dev, _plan_for, _run, xh  # noqa: F401  (xh: the module fixture)

pytestmark = pytest.mark.gpu


def _exchange(xh, samples, edges, w, **more):
    got, desc = _run(xh, samples, edges, w, True, partition=1, exchange=1, **more)
    assert "exchange=forced" in desc, desc
    return got, desc


@pytest.mark.parametrize("n", [4, 4095, 4096, 4097, 1_000_003, 3_000_001])
def test_exchange_mode_c5_shape_against_the_oracle(xh, n):
    """BASELINE C5's shape (two N(0,1) inputs, U[0,1) weights, 1024 x 1024 bins on [-4, 4]) at sizes around the 4096-sample
    tile: ragged tails, one partial tile, several tiles per workgroup"""
    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    rng = np.random.default_rng(500 + n % 97)
    x, y = rng.standard_normal((1, n)), rng.standard_normal((1, n))
    w = rng.uniform(0, 1, (1, n))
    want = onp.bincount_rows([x, y], edges, w)
    got, _ = _exchange(xh, [x, y], edges, w)
    assert_hist_equal(got, want, True)
    classic, desc = _run(xh, [x, y], edges, w, True, partition=1, exchange=-1)
    assert "exchange=no" in desc, desc
    np.testing.assert_allclose(got, classic, rtol=2.0 ** -34, atol=0)  # (both round the weights to 36 mantissa bits, then add in float64)


def test_exchange_mode_specials_and_edges(xh):
    """NaN / +-inf samples, samples ON edges (the right edge included, core.py:170-173), NaN weights, -0.0: what the digitize of
    the classic path is tested with, through the exchange kernel's copy of it"""
    edges = [np.linspace(-2, 2, 641), np.linspace(0, 1, 513)]
    rng = np.random.default_rng(7)
    n = 600_001
    x = rng.uniform(-2.2, 2.2, (1, n))
    y = rng.uniform(-0.1, 1.1, (1, n))
    x[0, ::101] = np.nan
    y[0, ::103] = np.inf
    x[0, ::107] = -np.inf
    x[0, 5::211] = edges[0][rng.integers(0, 641, x[0, 5::211].size)]
    y[0, 7::223] = edges[1][rng.integers(0, 513, y[0, 7::223].size)]
    x[0, 11::227] = 2.0  # the right edge counts (last bin)
    y[0, 13::229] = 1.0
    x[0, 17::233] = -0.0
    w = rng.uniform(0.5, 1.5, (1, n))
    want = onp.bincount_rows([x, y], edges, w)
    got, _ = _exchange(xh, [x, y], edges, w)
    assert_hist_equal(got, want, True)
    w[0, 1000] = np.nan  # a NaN weight poisons its bin, like np.bincount's
    want = onp.bincount_rows([x, y], edges, w)
    got, _ = _exchange(xh, [x, y], edges, w)
    assert_hist_equal(got, want, True)


@pytest.mark.parametrize("case", ["1d", "3d", "3d_striped", "short_rows", "fits_window"])
def test_exchange_mode_other_shapes(xh, case):
    """one input (rows are 256-bin pieces), three inputs (rows = the first two dimensions), rows of 40 bins (many rows per
    workgroup), and a histogram that fits the window (no probe, no side adds)"""
    rng = np.random.default_rng(11)
    n = 1_500_007
    if case == "1d":
        edges = [np.linspace(-5, 5, 300_001)]
        samples = [rng.standard_normal((1, n)) * 1.5]
    elif case == "3d":
        edges = [np.linspace(-3, 3, 65), np.linspace(-3, 3, 49), np.linspace(0, 1, 401)]
        samples = [rng.standard_normal((1, n)), rng.standard_normal((1, n)), rng.uniform(-0.05, 1.05, (1, n))]
    elif case == "3d_striped":  # rows = 32 x 32 bins of the first two inputs: the heavy rows repeat every 32 (the rotated owner map's case)
        edges = [np.linspace(-4, 4, 33), np.linspace(-4, 4, 33), np.linspace(-4, 4, 1025)]
        samples = [rng.standard_normal((1, n)), rng.standard_normal((1, n)), rng.standard_normal((1, n))]
    elif case == "short_rows":
        edges = [np.linspace(-3, 3, 20_001), np.linspace(0, 1, 41)]
        samples = [rng.standard_normal((1, n)), rng.uniform(0, 1, (1, n))]
    else:
        edges = [np.linspace(-3, 3, 401), np.linspace(-3, 3, 501)]
        samples = [rng.standard_normal((1, n)), rng.standard_normal((1, n))]
    w = -rng.uniform(0, 3, (1, n))  # one sign, the negative one
    want = onp.bincount_rows(samples, edges, w)
    got, _ = _exchange(xh, samples, edges, w)
    assert_hist_equal(got, want, True)


def test_exchange_mode_samples_outside_the_window(xh):
    """uniform samples over a histogram twice the window: half of the records go to the side copy with memory-side atomics
    (slow, exact) — the forced mode takes the call anyway"""
    edges = [np.linspace(0, 1, 1025), np.linspace(0, 1, 1025)]
    rng = np.random.default_rng(12)
    n = 2_000_003
    x, y = rng.uniform(0, 1, (1, n)), rng.uniform(0, 1, (1, n))
    w = rng.uniform(0, 1, (1, n))
    want = onp.bincount_rows([x, y], edges, w)
    got, _ = _exchange(xh, [x, y], edges, w)
    assert_hist_equal(got, want, True)


def test_exchange_mode_one_owner_gets_everything(xh):
    """all samples in ONE histogram row: every tile sends 4096 records to one ring of 512 — the run goes out in pieces as the
    owner catches up (no deadlock, no loss)"""
    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    rng = np.random.default_rng(13)
    n = 1_000_001
    x = np.full((1, n), 0.00123)
    y = rng.standard_normal((1, n))
    w = rng.uniform(0, 1, (1, n))
    want = onp.bincount_rows([x, y], edges, w)
    got, _ = _exchange(xh, [x, y], edges, w)
    assert_hist_equal(got, want, True)


def test_exchange_mode_mixed_signs_fall_back_to_exact_records(xh):
    """weights of both signs: the exchange kernel reports them, its merge does not run, and the exact routing + adding-up passes
    of the same call produce the result"""
    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    rng = np.random.default_rng(14)
    n = 2_000_003
    x, y = rng.standard_normal((1, n)), rng.standard_normal((1, n))
    w = rng.standard_normal((1, n))
    want = onp.bincount_rows([x, y], edges, w)
    exact, _ = _run(xh, [x, y], edges, w, True, partition=1, records48=-1)
    got, _ = _exchange(xh, [x, y], edges, w, records48=0)
    np.testing.assert_allclose(got, exact, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9 * np.abs(w).max())


def test_exchange_mode_deadline_hands_the_call_to_the_classic_passes(xh):
    """a deadline that has passed before the kernel starts: the first workgroup that has to wait gives up, everybody leaves, the mode is off for this call and the classic packed passes queued behind
    take it — a hang (workgroups that are not all resident, a placement other than 32 per XCD) costs time, never results"""
    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    rng = np.random.default_rng(15)
    n = 3_000_001
    x, y = rng.standard_normal((1, n)), rng.standard_normal((1, n))
    w = rng.uniform(0, 1, (1, n))
    want = onp.bincount_rows([x, y], edges, w)
    got, _ = _exchange(xh, [x, y], edges, w, exchange_budget_ms=-1, records48=0)
    assert_hist_equal(got, want, True)
    # the next call notices the abort, forgets the "both signs" note the redo left and keeps the plan off the mode ...
    again, desc = _run(xh, [x, y], edges, w, True, partition=1)
    assert "exchange=no" in desc and "records=packed48" in desc, desc
    assert_hist_equal(again, want, True)
    # ... until the knob is touched
    plan = _plan_for(xh, [_dev(x), _dev(y)], edges)
    plan.set_param("exchange", 0)
    again, desc = _run(xh, [x, y], edges, w, True, partition=1, exchange=1)
    assert "exchange=forced" in desc, desc
    assert_hist_equal(again, want, True)


def test_exchange_mode_is_chosen_by_the_probe(xh):
    """default settings at 2^25 samples: N(0,1) samples put 94 % into the window -> the mode takes the call; uniform samples
    put 47 % there -> the classic packed passes take it.  Same results either way."""
    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    n = 1 << 25
    g = torch_gen(16)
    import torch

    for dist in ("normal", "uniform"):
        x = torch.empty((1, n), dtype=torch.float64, device="cuda")
        y = torch.empty((1, n), dtype=torch.float64, device="cuda")
        if dist == "normal":
            x.normal_(generator=g), y.normal_(generator=g)
        else:
            x.uniform_(-4, 4, generator=g), y.uniform_(-4, 4, generator=g)
        w = torch.empty((1, n), dtype=torch.float64, device="cuda").uniform_(generator=g)
        plan = _plan_for(xh, [x, y], edges)
        plan.set_param("partition", 1)
        try:
            auto = xh._bincount_2d_vectorized(x, y, bins=edges, weights=w)
            torch.cuda.synchronize()
            xh._bincount_2d_vectorized(x, y, bins=edges, weights=w)  # (its description carries what the GPU reported for the call before)
            desc = plan.describe()
            assert "exchange=if the probe" in desc, desc
            ppm = int(desc.split("exchange_window_ppm_before=")[1].split()[0])
            assert (ppm >= 880_000) == (dist == "normal"), (dist, ppm)
            plan.set_param("exchange", -1)
            classic = xh._bincount_2d_vectorized(x, y, bins=edges, weights=w)
        finally:
            plan.set_param("exchange", 0)
            plan.set_param("partition", 0)
        torch.testing.assert_close(auto, classic, rtol=2.0 ** -34, atol=0)
        assert float(auto.sum()) > 0


def torch_gen(seed):
    import torch

    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    return g


def test_exchange_mode_keeps_its_kernels_from_meeting_on_two_streams(xh):
    """the exchange kernel wants every compute unit (256 persistent workgroups that wait for one another): a second call on
    ANOTHER stream while the first one's kernel is still running takes the classic passes instead of sharing the chip with it"""
    import torch

    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    n = 1 << 28  # (the first call's kernel runs ~1.8 ms: the second call's look at its event comes well inside that)
    g = torch_gen(17)
    x = torch.empty((1, n), dtype=torch.float64, device="cuda").normal_(generator=g)
    y = torch.empty((1, n), dtype=torch.float64, device="cuda").normal_(generator=g)
    w = torch.empty((1, n), dtype=torch.float64, device="cuda").uniform_(generator=g)
    plan = _plan_for(xh, [x, y], edges)
    plan.set_param("partition", 1)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    try:
        for s in (s1, s2, s1, s2):  # (both streams have their scratch memory before the pair that counts: a fresh hipMalloc waits for the GPU)
            with torch.cuda.stream(s):
                xh._bincount_2d_vectorized(x, y, bins=edges, weights=w)
            torch.cuda.synchronize()
        with torch.cuda.stream(s1):
            a = xh._bincount_2d_vectorized(x, y, bins=edges, weights=w)
            d1 = plan.describe()
        with torch.cuda.stream(s2):
            b = xh._bincount_2d_vectorized(x, y, bins=edges, weights=w)
            d2 = plan.describe()
        torch.cuda.synchronize()
    finally:
        plan.set_param("partition", 0)
    assert "exchange=if the probe" in d1, d1
    assert "exchange=no" in d2, d2
    torch.testing.assert_close(a, b, rtol=2.0 ** -34, atol=0)
