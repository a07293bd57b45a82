"""The in-call multi-GPU path on the GPU box.  The test box has ONE MI355X, so what runs here is the real
machinery at group size 1 — the per-GPU host thread, the library's device buffers, a one-rank RCCL communicator
created from that thread, scatter / Sharded / histogram — plus whatever the box offers beyond that: with two or
more GPUs visible the same tests use them all (and say so in their output; with one GPU they report
`multi-GPU legs: skipped, 1 GPU visible`)."""
import numpy as np
import pytest

from conftest import assert_hist_equal
from oracle import oracle_np as onp

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def mg():
    from xhistogram_amd import _native, multigpu

    assert _native.device_count() >= 1
    return multigpu


def _devices(mg):
    devs = mg.visible_devices()
    print("multi-GPU legs: %s" % ("running on %d GPUs" % len(devs) if len(devs) > 1 else "skipped, 1 GPU visible (group size 1 only)"))
    return devs


def test_device_buffer_roundtrip_and_add():
    from xhistogram_amd import _native

    a = np.arange(1000, dtype=np.int64)
    b = np.full(1000, 7, dtype=np.int64)
    ba, bb = _native.DeviceBuffer(0, a.nbytes), _native.DeviceBuffer(0, b.nbytes)
    ba.upload(a)
    bb.upload(b)
    ba.add(bb, 1000, _native.I64)
    out = np.empty_like(a)
    ba.download(out)
    np.testing.assert_array_equal(out, a + 7)
    f = np.linspace(0, 1, 1000)
    ba.upload(f)
    bb.upload(f * 2)
    ba.add(bb, 1000, _native.F64)
    fo = np.empty_like(f)
    ba.download(fo)
    np.testing.assert_allclose(fo, f * 3, rtol=1e-15)
    ba.close()
    bb.close()


def test_host_partials_through_rccl_from_the_gpu_threads(mg):
    devs = _devices(mg)
    group = mg.group_for(devs)
    rng = np.random.default_rng(0)
    parts = [rng.integers(0, 1000, (3, 100)).astype(np.int64) for _ in devs]
    got = mg._allreduce_host_partials(group, parts)
    np.testing.assert_array_equal(got, sum(parts))
    fparts = [rng.uniform(0, 1, (1024, 64)) for _ in devs]
    np.testing.assert_allclose(mg._allreduce_host_partials(group, fparts), sum(fparts), rtol=1e-12)
    assert group.comms()[0].rccl_version() > 0


@pytest.mark.parametrize("exchange", ["host", "rccl"])
def test_host_inputs_sharded_over_the_visible_gpus(mg, exchange):
    """numpy inputs cut into shards, one per GPU thread (explicit device list: the size threshold is for the
    automatic policy); with one GPU the call declines (None) and core.histogram takes the single-GPU route"""
    from xhistogram_amd import core

    devs = _devices(mg)
    rng = np.random.default_rng(1)
    x, w = rng.standard_normal(2_000_003), rng.uniform(0, 1, 2_000_003)
    e = np.linspace(-4, 4, 101)
    kw = dict(weights=True, axis=None, bins=[e], density=False, block_size="auto")
    got = mg.host_sharded_counts([x, w], w, 1, True, False, (0,), [e], kw, devices=devs, exchange=exchange)
    if len(devs) < 2:
        assert got is None
        got = core._counts_one_device([x, w], w, 1, True, False, (0,), [e], kw, "numpy")
    assert_hist_equal(got.reshape(-1), onp.histogram(x, bins=e, weights=w)[0], weighted=True)


def test_scatter_and_histogram_device_resident_shards(mg):
    devs = _devices(mg)
    rng = np.random.default_rng(2)
    x = rng.standard_normal((64, 9, 1000)).astype(np.float32)
    w = rng.uniform(0, 1, x.shape)
    e = np.linspace(-4, 4, 51)
    for shard_axis, axis in ((0, (1, 2)), (2, (1, 2)), (0, None)):
        xs = mg.scatter(x, devs, axis=shard_axis)
        ws = mg.scatter(w, devs, axis=shard_axis)
        assert [p.device.index for p in xs.parts] == devs[: len(xs.parts)]
        h, _ = mg.histogram(xs, bins=e, axis=axis)
        np.testing.assert_array_equal(h.cpu().numpy(), onp.histogram(x, bins=e, axis=axis)[0])
        hw, _ = mg.histogram(xs, bins=e, axis=axis, weights=ws, density=True)
        assert_hist_equal(hw.cpu().numpy(), onp.histogram(x, bins=e, axis=axis, weights=w, density=True)[0], weighted=True)
    xs = mg.scatter(x, devs, axis=0)
    h, edges = mg.histogram(xs, bins=20)  # integer bins: global min / max over the shards
    want, wedges = onp.histogram(x, bins=20)
    np.testing.assert_array_equal(edges[0], wedges[0])
    np.testing.assert_array_equal(h.cpu().numpy(), want)
    for ex in ("p2p", "rccl"):
        h, _ = mg.histogram(xs, bins=e, exchange=ex)
        np.testing.assert_array_equal(h.cpu().numpy(), onp.histogram(x, bins=e)[0])


def test_dask_style_concurrent_blocks_on_the_host_route(mg):
    """what dask's threaded scheduler does to the block adapter: many threads, each one block, each under
    multigpu.block_device(); every thread stages on its own stream (hipStreamPerThread)"""
    import threading

    from xhistogram_amd import core

    rng = np.random.default_rng(3)
    blocks = [rng.standard_normal((4, 50_000)).astype(np.float32) for _ in range(16)]
    e = np.linspace(-4, 4, 51)
    out = [None] * len(blocks)

    def work(k):
        out[k] = core._bincount_spread(blocks[k], weights=False, axis=[1], bins=[e], density=False, block_size="auto")

    ts = [threading.Thread(target=work, args=(k,)) for k in range(len(blocks))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    for k, b in enumerate(blocks):
        np.testing.assert_array_equal(out[k].reshape(4, 50), onp.histogram(b, bins=e, axis=1)[0])
    assert all(v == 0 for v in mg._inflight.values())
