"""Run by tests/test_gpu_alias_two_devices.py in a process of its own with XHIST_AMD_DEVICE_ALIAS=0,0: the library then shows
TWO logical devices that are both HIP device 0, and everything the multi-GPU code keys by device runs for real on a one-GPU
box — two host threads (multigpu.DeviceGroup), two plans per set of edges, two staging streams, device buffers of two
"GPUs", the block -> GPU assignment — with real kernels, against the numpy oracle.  What cannot run this way is RCCL itself
(it refuses two ranks on one GPU): the all-reduce is replaced by an add on the shared GPU, the code around it is real.
Prints one "ok <name>" line per check; any failure raises.
"""
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert os.environ.get("XHIST_AMD_DEVICE_ALIAS") == "0,0"

import torch  # noqa: E402

from oracle import oracle_np as onp  # noqa: E402
from xhistogram_amd import _native, core, multigpu  # noqa: E402


def close(got, want, weighted):
    if weighted:
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=0, equal_nan=True)
    else:
        np.testing.assert_array_equal(got, want)


def main():
    assert _native.device_count() == 2 and _native.physical_device(1) == 0
    assert _native.device_info(1)["name"].startswith("gfx950")
    multigpu.set_devices([0, 1])
    rng = np.random.default_rng(0)
    e = np.linspace(-4, 4, 101)

    # ---- two plans for one set of edges (the cache is keyed by LOGICAL device) -------------------------------
    cmp_domain, conv, _ = core._compare_domain([np.dtype("f8")], [e])
    p0, p1 = core._get_plan(conv, cmp_domain, 0), core._get_plan(conv, cmp_domain, 1)
    assert p0 is not p1 and p0.device == 0 and p1.device == 1
    print("ok two plan-cache keys")

    # ---- device-resident shards: scatter -> one thread per logical GPU -> exchange ----------------------------
    threads_seen = set()
    orig = core._counts_one_device

    def spy(*a, **k):
        threads_seen.add((threading.current_thread().name, getattr(core._tls, "device", None)))
        return orig(*a, **k)

    core._counts_one_device = spy
    try:
        x = rng.standard_normal((64, 9, 4000)).astype(np.float32)
        w = rng.uniform(0, 1, x.shape)
        e2 = np.linspace(-4, 4, 51)
        for shard_axis, axis in ((0, (1, 2)), (2, (1, 2)), (0, None)):
            xs, ws = multigpu.scatter(x, [0, 1], axis=shard_axis), multigpu.scatter(w, [0, 1], axis=shard_axis)
            assert len(xs.parts) == 2 and all(p.device.index == 0 for p in xs.parts) and xs.devices == [0, 1]
            h, _ = multigpu.histogram(xs, bins=e2, axis=axis, exchange="p2p")
            close(h.cpu().numpy(), onp.histogram(x, bins=e2, axis=axis)[0], False)
            hw, _ = multigpu.histogram(xs, bins=e2, axis=axis, weights=ws, density=True, exchange="p2p")
            close(hw.cpu().numpy(), onp.histogram(x, bins=e2, axis=axis, weights=w, density=True)[0], True)
        xs = multigpu.scatter(x, [0, 1], axis=0)
        h, edges = multigpu.histogram(xs, bins=20, exchange="p2p")  # integer bins: global min / max over the shards
        want, wedges = onp.histogram(x, bins=20)
        np.testing.assert_array_equal(edges[0], wedges[0])
        close(h.cpu().numpy(), want, False)
        for name in ("sqrt", "sturges", "rice", "scott"):  # bin estimators from the two GPUs' moments (xhist_moments on each)
            h, edges = multigpu.histogram(xs, bins=name, exchange="p2p")
            want, wedges = np.histogram(x, bins=name)
            np.testing.assert_array_equal(edges[0], wedges)
            close(h.cpu().numpy(), want, False)
    finally:
        core._counts_one_device = orig
    assert {d for _, d in threads_seen} == {0, 1} and len({t for t, _ in threads_seen}) == 2, threads_seen
    assert p0.describe() or True
    print("ok device-resident shards, p2p exchange, two GPU threads:", sorted(threads_seen))

    # a big joint histogram (the routing pass, 2 x 512-thread workgroups per CU) from both threads AT ONCE on one physical GPU
    n = 6_000_000
    bx, by, bw = rng.standard_normal(n), rng.standard_normal(n), rng.uniform(0, 1, n)
    eb = [np.linspace(-4, 4, 1025)] * 2
    sx, sy, sw = (multigpu.scatter(a, [0, 1], axis=0) for a in (bx, by, bw))
    hb, _ = multigpu.histogram(sx, sy, bins=eb, weights=sw, exchange="p2p")
    close(hb.cpu().numpy(), onp.histogram(bx, by, bins=eb, weights=bw)[0], True)
    print("ok concurrent partitioned-mode calls of two logical devices")

    # ---- host (numpy) shards: one staging pipeline per logical GPU, partials added on the host ----------------
    xh, wh = rng.standard_normal(9_000_003), rng.uniform(0, 1, 9_000_003)
    kw = dict(weights=True, axis=None, bins=[e], density=False, block_size="auto")
    got = multigpu.host_sharded_counts([xh, wh], wh, 1, True, False, (0,), [e], kw, devices=[0, 1], exchange="host")
    assert got is not None
    close(got.reshape(-1), onp.histogram(xh, bins=e, weights=wh)[0], True)
    rows = rng.standard_normal((40, 300_000)).astype(np.float32)  # kept-axis shards: disjoint rows, concatenated
    kw = dict(weights=False, axis=[1], bins=[e], density=False, block_size="auto")
    got = multigpu.host_sharded_counts([rows], None, 1, False, False, (1,), [e], kw, devices=[0, 1], exchange="host")
    close(np.asarray(got).reshape(40, 100), onp.histogram(rows, bins=e, axis=1)[0], False)
    h, _ = core.histogram(xh, bins=e, weights=wh)  # the public call takes the same route above 32 MiB per shard
    close(h, onp.histogram(xh, bins=e, weights=wh)[0], True)
    print("ok host shards over two logical GPUs")

    # ---- dask-style blocks: block -> least busy GPU, partials stay on "their" GPU, reduce_partials --------------
    blocks = [rng.standard_normal((4, 1, 50_000)).astype(np.float32) for _ in range(12)]
    parts = [None] * len(blocks)

    def work(k):
        parts[k] = core._bincount_partial(blocks[k], weights=False, axis=[1, 2], bins=[e], density=False, block_size="auto")

    ts = [threading.Thread(target=work, args=(k,)) for k in range(len(blocks))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(isinstance(p, _native.DevicePartial) for p in parts)
    assert {p.device for p in parts} == {0, 1}, [p.device for p in parts]  # both logical GPUs took blocks

    def allreduce_on_the_shared_gpu(sums, count, tag):  # stands in for RCCL: both buffers live on HIP device 0
        assert [s.device for s in sums] == [0, 1]
        sums[0].buf.add(sums[1].buf, count, tag)
        sums[0].buf.synchronize()

    for attempt in range(2):  # (a re-run of the reduce task sees untouched inputs)
        total = multigpu.reduce_partials([parts], drop_axes=(1, 2), out_dtype="<i8", _allreduce=allreduce_on_the_shared_gpu)
        want = sum(onp.histogram(b, bins=e, axis=(1, 2))[0] for b in blocks)
        close(total, want, False)
    assert all(v == 0 for v in multigpu._inflight.values())
    print("ok dask-style blocks on two logical GPUs + reduce_partials (exchange stubbed)")

    # ---- pickling: a partial that crosses a process boundary arrives through host memory -----------------------
    import copy
    import pickle

    from xhistogram_amd.devicearray import DeviceArray

    p = parts[0]
    q = pickle.loads(pickle.dumps(p))
    assert q.buf.ptr != p.buf.ptr and q.shape == p.shape and q.device == p.device
    np.testing.assert_array_equal(q.to_numpy(), p.to_numpy())
    r = copy.deepcopy(p)
    assert r.buf.ptr not in (p.buf.ptr, q.buf.ptr)
    np.testing.assert_array_equal(r.to_numpy(), p.to_numpy())
    a = DeviceArray.from_numpy(np.arange(24.0).reshape(2, 3, 4), 1)
    v = a[:, 1:, ::2]
    b = pickle.loads(pickle.dumps(v))
    assert b.device == 1 and b.ptr != v.ptr and b.is_contiguous()
    np.testing.assert_array_equal(b.to_numpy(), np.arange(24.0).reshape(2, 3, 4)[:, 1:, ::2])
    print("ok pickle / deepcopy of DevicePartial and DeviceArray")


if __name__ == "__main__":
    main()
    print("ALL OK")
