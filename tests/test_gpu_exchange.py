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


def _c5_inputs(n, seed):
    import torch

    g = torch_gen(seed)
    x = torch.empty((1, n), dtype=torch.float64, device="cuda").normal_(generator=g)
    y = torch.empty((1, n), dtype=torch.float64, device="cuda").normal_(generator=g)
    w = torch.empty((1, n), dtype=torch.float64, device="cuda").uniform_(generator=g)
    return x, y, w


def _note(desc, key):
    return int(desc.split(key + "=")[1].split()[0])


def test_exchange_mode_views_that_are_only_8_byte_aligned(xh):
    """x[1:], y[1:], w[1:]: the kernel's 16-byte lane loads start on an odd element (the vector type is declared
    8-byte aligned, as part_route's is)"""
    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    rng = np.random.default_rng(21)
    n = 1_000_002
    x, y, w = rng.standard_normal(n), rng.standard_normal(n), rng.uniform(0, 1, n)
    want = onp.bincount_rows([x[None, 1:], y[None, 1:]], edges, w[None, 1:])
    xd, yd, wd = _dev(x), _dev(y), _dev(w)
    assert xd[1:].data_ptr() % 16 == 8
    plan = _plan_for(xh, [xd[None, 1:], yd[None, 1:]], edges)
    plan.set_param("partition", 1)
    plan.set_param("exchange", 1)
    try:
        got = xh._bincount_2d_vectorized(xd[None, 1:], yd[None, 1:], bins=edges, weights=wd[None, 1:])
        assert "exchange=forced" in plan.describe()
    finally:
        plan.set_param("exchange", 0)
        plan.set_param("partition", 0)
    assert_hist_equal(got.cpu().numpy(), want, True)


def test_exchange_mode_with_a_compute_unit_held_by_another_stream_fails_fast(xh):
    """VERDICT r5 "next" #3: a kernel of somebody else's that holds compute units on a second stream (xhist_debug_hold_cus: what an
    RCCL kernel of a collective, or another process, does) keeps some of the 256 persistent workgroups out.  The arrival handshake notices within its 200 us — nothing consumed, nothing
    produced — and the classic passes queued behind take the call: the result is the histogram, the call costs little more than
    a classic call (not the 0.5 s deadline), it is not counted as an abort in flight and the plan stays on the mode."""
    import time

    import torch

    from xhistogram_amd import _native

    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    n = 1 << 26
    x, y, w = _c5_inputs(n, 31)
    plan = _plan_for(xh, [x, y], edges)
    plan.set_param("partition", 1)
    side = torch.cuda.Stream()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

    def timed(e0, e1):
        e0.record()
        out = xh._bincount_2d_vectorized(x, y, bins=edges, weights=w)
        e1.record()
        return out

    try:
        plan.set_param("exchange", -1)
        for _ in range(3):
            classic = xh._bincount_2d_vectorized(x, y, bins=edges, weights=w)
        torch.cuda.synchronize()
        classic = timed(ev[0], ev[1])
        torch.cuda.synchronize()
        t_classic = ev[0].elapsed_time(ev[1])
        plan.set_param("exchange", 0)
        for _ in range(3):
            free = xh._bincount_2d_vectorized(x, y, bins=edges, weights=w)  # (the mode takes these: N(0,1) samples)
        torch.cuda.synchronize()
        d0 = plan.describe()
        assert "exchange=if the probe" in d0, d0
        aborts0, misses0 = _note(d0, "exchange_aborts"), _note(d0, "exchange_arrival_misses")
        # eight idle workgroups with 96 KB of LDS each, 0.3 s long: no 158 KB workgroup fits beside one (torch.cuda._sleep's single
        # wavefront holds no LDS and does NOT keep the kernel's workgroup off its compute unit — measured)
        _native.debug_hold_cus(8, 96 * 1024, 300_000, stream=side.cuda_stream)
        time.sleep(0.02)  # (they have started)
        held = timed(ev[2], ev[3])
        t_host = time.perf_counter()
        ev[3].synchronize()
        assert not side.query(), "the holding kernel ended before the histogram call did: nothing was held"
        t_held = ev[2].elapsed_time(ev[3])
        torch.cuda.synchronize()
        del t_host
        xh._bincount_2d_vectorized(x, y, bins=edges, weights=w)  # (its description carries what the GPU reported for the call before)
        d1 = plan.describe()
        again = xh._bincount_2d_vectorized(x, y, bins=edges, weights=w)
        torch.cuda.synchronize()
        d2 = plan.describe()
    finally:
        plan.set_param("exchange", 0)
        plan.set_param("partition", 0)
    torch.testing.assert_close(held, classic, rtol=2.0 ** -34, atol=0)
    torch.testing.assert_close(again, classic, rtol=2.0 ** -34, atol=0)
    torch.testing.assert_close(free, classic, rtol=2.0 ** -34, atol=0)
    assert _note(d1, "exchange_arrival_misses") == misses0 + 1, (d0, d1)
    assert _note(d1, "exchange_aborts") == aborts0 and _note(d2, "exchange_aborts") == aborts0, (d0, d1, d2)
    assert "exchange=if the probe" in d1 and "exchange=if the probe" in d2, (d1, d2)  # the plan stays on the mode
    assert t_held <= 2.0 * t_classic + 0.3, (t_held, t_classic)  # (ms; 0.2 of them are the handshake's patience)


def test_exchange_mode_next_to_a_matmul_stream(xh):
    """the same with a stream of torch matmuls beside the calls: whichever way a call goes — every workgroup resident, or one
    kept out and the classic passes instead — the results are the classic ones and no call waits for a deadline"""
    import torch

    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    n = 1 << 25
    x, y, w = _c5_inputs(n, 32)
    plan = _plan_for(xh, [x, y], edges)
    plan.set_param("partition", 1)
    a = torch.randn(4096, 4096, device="cuda", dtype=torch.float32)
    side = torch.cuda.Stream()
    try:
        plan.set_param("exchange", -1)
        classic = xh._bincount_2d_vectorized(x, y, bins=edges, weights=w)
        torch.cuda.synchronize()
        plan.set_param("exchange", 0)
        xh._bincount_2d_vectorized(x, y, bins=edges, weights=w)
        torch.cuda.synchronize()
        aborts0 = _note(plan.describe(), "exchange_aborts")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        outs = []
        e0.record()
        for _ in range(12):
            with torch.cuda.stream(side):
                for _ in range(4):
                    a @ a
            outs.append(xh._bincount_2d_vectorized(x, y, bins=edges, weights=w))
        e1.record()
        torch.cuda.synchronize()
        xh._bincount_2d_vectorized(x, y, bins=edges, weights=w)
        d = plan.describe()
    finally:
        plan.set_param("exchange", 0)
        plan.set_param("partition", 0)
    for o in outs:
        torch.testing.assert_close(o, classic, rtol=2.0 ** -34, atol=0)
    assert _note(d, "exchange_aborts") == aborts0, d
    assert e0.elapsed_time(e1) < 12 * 60.0, e0.elapsed_time(e1)  # (ms: nowhere near 12 deadlines of 500)


def test_exchange_mode_is_admitted_again_after_an_abort_in_flight(xh):
    """an abort in flight (here: a deadline that has already passed, "exchange_budget_ms" = -1) keeps the plan off the mode for
    its next 16 eligible calls — not for good — and the 17th takes it again"""
    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    n = 1 << 25
    x, y, w = _c5_inputs(n, 33)
    plan = _plan_for(xh, [x, y], edges)
    plan.set_param("partition", 1)
    plan.set_param("exchange", 0)  # (forgets what earlier tests left)
    import torch

    try:
        plan.set_param("exchange_budget_ms", -1)
        first = xh._bincount_2d_vectorized(x, y, bins=edges, weights=w)
        assert "exchange=if the probe" in plan.describe()
        torch.cuda.synchronize()
        plan.set_param("exchange_budget_ms", 0)
        seen = []
        for _ in range(18):
            out = xh._bincount_2d_vectorized(x, y, bins=edges, weights=w)
            torch.cuda.synchronize()
            seen.append("exchange=no" in plan.describe())
        torch.testing.assert_close(out, first, rtol=2.0 ** -34, atol=0)
    finally:
        plan.set_param("exchange_budget_ms", 0)
        plan.set_param("exchange", 0)
        plan.set_param("partition", 0)
    assert seen[:16] == [True] * 16 and seen[16:] == [False, False], seen


# ---- exact records through the rings (round 6): the reference's float64 adds of unrounded weights, any sign mixture ------------
def _exchange_exact(xh, samples, edges, w, **more):
    got, desc = _run(xh, samples, edges, w, True, partition=1, exchange=1, records48=-1, **more)
    assert "exchange=forced" in desc and "exchange_records=exact12" in desc, desc
    return got, desc


@pytest.mark.parametrize("n", [4, 4095, 4097, 1_000_003, 3_000_001])
@pytest.mark.parametrize("signs", ["one", "both"])
def test_exchange_mode_exact_records_c5_shape(xh, n, signs):
    """"records48" = -1: the weight travels whole (the packed word + its 16 low bits in a second ring, each with the lap tag) —
    weights of one sign and of both, at sizes around the tile; the result is the classic exact passes' to the last bits of a
    float64 sum taken in another order, and the oracle's within the contract"""
    edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
    rng = np.random.default_rng(600 + n % 89)
    x, y = rng.standard_normal((1, n)), rng.standard_normal((1, n))
    w = rng.uniform(0, 1, (1, n)) if signs == "one" else rng.standard_normal((1, n))
    want = onp.bincount_rows([x, y], edges, w)
    got, _ = _exchange_exact(xh, [x, y], edges, w)
    classic, desc = _run(xh, [x, y], edges, w, True, partition=1, exchange=-1, records48=-1)
    assert "exchange=no" in desc, desc
    np.testing.assert_allclose(got, classic, rtol=1e-12, atol=1e-12 * np.abs(w).max())
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9 * np.abs(w).max())


def test_exchange_mode_exact_records_keep_every_bit_of_a_weight(xh):
    """every sample in a bin of its own column, weights whose 16 low mantissa bits are all set / alternate / are the only bits
    that differ: a sum of ONE weight per bin must come back bit for bit (a packed record would round it); NaN, infinities and
    subnormal weights included"""
    nb = 1024
    edges = [np.linspace(0, nb, nb + 1), np.linspace(0, nb, nb + 1)]
    rng = np.random.default_rng(61)
    n = 200_000
    cells = rng.choice(nb * nb, n, replace=False)
    x = (cells // nb + 0.5)[None, :].astype(np.float64)
    y = (cells % nb + 0.5)[None, :].astype(np.float64)
    bits = rng.integers(0, 1 << 62, n, dtype=np.int64) | 0xFFFF
    bits[::3] ^= 0xAAAA
    w = bits.view(np.float64).copy()
    w[::1009] = np.nan
    w[5::2003] = np.inf
    w[7::3001] = -np.inf
    w[11::4001] = 5e-324
    w = w[None, :]
    got, _ = _exchange_exact(xh, [x, y], edges, w)
    want = np.zeros(nb * nb)
    want[cells] = w[0]
    np.testing.assert_array_equal(got.reshape(-1).view(np.int64)[~np.isnan(want)], want.view(np.int64)[~np.isnan(want)])
    assert np.array_equal(np.isnan(got.reshape(-1)), np.isnan(want))


@pytest.mark.parametrize("case", ["1d", "3d", "one_owner", "outside"])
def test_exchange_mode_exact_records_other_shapes(xh, case):
    rng = np.random.default_rng(62)
    n = 1_200_007
    if case == "1d":
        edges = [np.linspace(-5, 5, 300_001)]
        samples = [rng.standard_normal((1, n)) * 1.5]
    elif case == "3d":
        edges = [np.linspace(-3, 3, 65), np.linspace(-3, 3, 49), np.linspace(0, 1, 401)]
        samples = [rng.standard_normal((1, n)), rng.standard_normal((1, n)), rng.uniform(-0.05, 1.05, (1, n))]
    elif case == "one_owner":  # every tile sends 4096 records to one ring of 512: out in pieces
        edges = [np.linspace(-4, 4, 1025), np.linspace(-4, 4, 1025)]
        samples = [np.full((1, n), 0.00123), rng.standard_normal((1, n))]
    else:  # uniform samples over twice the window: half of them beside it
        edges = [np.linspace(0, 1, 1025), np.linspace(0, 1, 1025)]
        samples = [rng.uniform(0, 1, (1, n)), rng.uniform(0, 1, (1, n))]
    w = rng.standard_normal((1, n))
    want = onp.bincount_rows(samples, edges, w)
    got, _ = _exchange_exact(xh, samples, edges, w)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9 * np.abs(w).max())
