"""The N > 1 path on CPU: world_size-2 `gloo` process groups run the sharding / collective logic of
xhistogram_amd.distributed with the rank-local compute swapped for the oracle (test double; the
product default is the HIP path and is covered by the gpu-marked test at the bottom)."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_local(arrays, has_weights, axis, edges, block_size):
    from oracle import oracle_np as onp

    arrs = [a.numpy() if hasattr(a, "numpy") else np.asarray(a) for a in arrays]
    w = arrs.pop() if has_weights else None
    return onp.block_adapter(arrs, edges, w, axis)


CASES = {
    # name: (shape, n_args, kwargs, shard_axis, use_torch)
    "full_reduce_1d_weighted": ((1001,), 1, dict(bins=np.linspace(-4, 4, 101), weights=True), 0, True),
    "full_reduce_2d_hist": ((37, 41), 2, dict(bins=[np.linspace(-4, 4, 10), np.linspace(-3, 3, 8)]), 0, False),
    "reduced_axis_shard_keep_rows": ((6, 101), 1, dict(bins=np.linspace(-4, 4, 21), axis=1), 1, True),
    "kept_axis_shard_c4_like": ((7, 8, 9), 1, dict(bins=np.linspace(-4, 4, 51), axis=(1, 2)), 0, True),
    "kept_axis_shard_middle": ((5, 7, 6), 1, dict(bins=np.linspace(-4, 4, 11), axis=(0, 2)), 1, False),
    "bins_int_global_minmax": ((1001,), 1, dict(bins=17), 0, True),
    "bins_int_range_density": ((64, 33), 1, dict(bins=9, range=(-2, 2), density=True, axis=1, weights=True), 1, False),
    "density_full": ((500,), 1, dict(bins=np.linspace(-4, 4, 13), density=True), 0, True),
    # bin estimators from per-rank moments (n, min, max, mean, M2), combined after ONE all-gather of five numbers per rank
    "bins_scott_from_rank_moments": ((2001,), 1, dict(bins="scott"), 0, True),
    "bins_sturges_range_kept_rows": ((40, 101), 1, dict(bins="sturges", range=(-2, 2.5), axis=1), 1, False),
    "bins_sqrt_rice_two_args": ((33, 57), 2, dict(bins=["sqrt", "rice"]), 0, True),
}


def _worker(rank, world, port, case, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import oracle_np as onp
        from xhistogram_amd import distributed as xd

        shape, n_args, kw, shard_axis, use_torch = CASES[case]
        rng = np.random.default_rng(99)
        full = [rng.standard_normal(shape) for _ in range(n_args)]
        kw = dict(kw)
        wfull = rng.uniform(0, 1, shape) if kw.pop("weights", False) else None
        lo, hi = xd.shard_bounds(shape[shard_axis], world, rank)
        sl = [slice(None)] * len(shape)
        sl[shard_axis] = slice(lo, hi)
        conv = (lambda a: torch.from_numpy(np.ascontiguousarray(a))) if use_torch else (lambda a: a)
        mine = [conv(a[tuple(sl)]) for a in full]
        wmine = None if wfull is None else conv(wfull[tuple(sl)])
        with xd._hooks(local=_oracle_local):  # (module-private test hook: the oracle is the rank-local compute on CPU)
            h, edges = xd.histogram(*mine, weights=wmine, shard_axis=shard_axis, **kw)
        want, wedges = onp.histogram(*full, weights=wfull, **kw)
        h = h.numpy() if hasattr(h, "numpy") else np.asarray(h)
        ok = h.shape == want.shape and np.allclose(h, want, rtol=1e-12, atol=0, equal_nan=True)
        ok = ok and all(np.array_equal(a, b) for a, b in zip(edges, wedges))
        if want.dtype.kind in "iu":
            ok = ok and h.dtype.kind in "iu" and np.array_equal(h, want)
        q.put((rank, bool(ok), "" if ok else "shape %s vs %s" % (h.shape, want.shape)))
    except Exception as e:  # pragma: no cover
        import traceback

        q.put((rank, False, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", sorted(CASES))
def test_world2_gloo(case):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, case, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, msg in res:
        assert ok, "rank %d: %s" % (rank, msg)


def test_shard_bounds_cover_everything():
    from xhistogram_amd.distributed import shard_bounds

    for n in (0, 1, 7, 8, 3650, 10**9):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1


@pytest.mark.gpu
def test_single_rank_nccl_group_uses_hip_path():
    """world_size 1 on a real GPU: the default rank-local compute is the HIP kernel and the
    RCCL all-reduce / all-gather paths run end to end"""
    sys.path.insert(0, ROOT)
    from oracle import oracle_np as onp
    from xhistogram_amd import distributed as xd

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rng = np.random.default_rng(5)
        x = rng.standard_normal((16, 5000)).astype(np.float32)
        w = rng.uniform(0, 1, (16, 5000))
        edges = np.linspace(-4, 4, 51)
        xt, wt = torch.as_tensor(x).cuda(), torch.as_tensor(w).cuda()
        h, _ = xd.histogram(xt, bins=edges, axis=1, shard_axis=0)  # kept-axis shards -> gather
        np.testing.assert_array_equal(h.cpu().numpy(), onp.histogram(x, bins=edges, axis=1)[0])
        h, _ = xd.histogram(xt, bins=edges, weights=wt, shard_axis=1)  # reduced-axis shards -> all-reduce
        np.testing.assert_allclose(h.cpu().numpy(), onp.histogram(x, bins=edges, weights=w)[0], rtol=1e-6)
        h, e = xd.histogram(xt, bins=23)  # global min/max on device
        want, we = onp.histogram(x, bins=23)
        np.testing.assert_array_equal(e[0], we[0])
        np.testing.assert_array_equal(h.cpu().numpy(), want)
    finally:
        dist.destroy_process_group()


def _native_cases(xd, onp, comm, rank, world, dev):
    """the three exchange shapes through the C ABI's own RCCL communicator, data sharded by rank"""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((16, 5000)).astype(np.float32)
    w = rng.uniform(0, 1, (16, 5000))
    edges = np.linspace(-4, 4, 51)

    def shard(a, axis):
        lo, hi = xd.shard_bounds(a.shape[axis], world, rank)
        sl = [slice(None)] * a.ndim
        sl[axis] = slice(lo, hi)
        return torch.as_tensor(np.ascontiguousarray(a[tuple(sl)])).to(dev)

    h, _ = xd.histogram(shard(x, 0), bins=edges, axis=1, shard_axis=0, group=comm)  # kept-axis shards -> all-gather
    np.testing.assert_array_equal(h.cpu().numpy(), onp.histogram(x, bins=edges, axis=1)[0])
    h, _ = xd.histogram(shard(x, 1), bins=edges, weights=shard(w, 1), shard_axis=1, group=comm)  # -> all-reduce (f64)
    np.testing.assert_allclose(h.cpu().numpy(), onp.histogram(x, bins=edges, weights=w)[0], rtol=1e-6)
    h, e = xd.histogram(shard(x, 1), bins=23, shard_axis=1, group=comm)  # min / max all-reduce, then int64 all-reduce
    want, we = onp.histogram(x, bins=23)
    np.testing.assert_array_equal(e[0], we[0])
    np.testing.assert_array_equal(h.cpu().numpy(), want)
    h, _ = xd.histogram(x[:, : 100 * (rank + 1)], bins=edges, shard_axis=1, group=comm)  # numpy in (host route)
    assert isinstance(h, np.ndarray)


@pytest.mark.gpu
def test_single_rank_native_comm():
    """xhist_comm_* (RCCL dlopen-ed by libxhist_amd.so) end to end on one GPU: id, create, all-reduce of
    int64 / float64, min / max, all-gather, destroy"""
    sys.path.insert(0, ROOT)
    from oracle import oracle_np as onp
    from xhistogram_amd import _native
    from xhistogram_amd import distributed as xd

    torch.cuda.set_device(0)
    comm = _native.Comm(0, 0, 1, _native.comm_unique_id())
    try:
        assert comm.rccl_version() > 20000
        t = torch.arange(1000, dtype=torch.int64, device="cuda")
        comm.allreduce(t.data_ptr(), t.numel(), _native.I64, _native.REDUCE_SUM, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(t.cpu(), torch.arange(1000, dtype=torch.int64))
        with pytest.raises(ValueError):
            comm.allreduce(t.data_ptr(), t.numel(), _native.U8, _native.REDUCE_SUM)
        _native_cases(xd, onp, comm, 0, 1, torch.device("cuda", 0))
    finally:
        comm.close()


def _native_worker(rank, world, id_bytes, q):
    sys.path.insert(0, ROOT)
    try:
        from oracle import oracle_np as onp
        from xhistogram_amd import _native
        from xhistogram_amd import distributed as xd

        torch.cuda.set_device(rank)
        comm = _native.Comm(rank, rank, world, id_bytes)
        try:
            _native_cases(xd, onp, comm, rank, world, torch.device("cuda", rank))
        finally:
            comm.close()
        q.put((rank, True, ""))
    except Exception:  # pragma: no cover
        import traceback

        q.put((rank, False, traceback.format_exc()))


@pytest.mark.gpu
def test_world2_native_comm():
    """two processes, two GPUs, no torch.distributed anywhere: the id travels through a pipe"""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    sys.path.insert(0, ROOT)
    from xhistogram_amd import _native

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ident = _native.comm_unique_id()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_native_worker, args=(r, 2, ident, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, msg in res:
        assert ok, "rank %d: %s" % (rank, msg)


def test_native_comm_fails_loudly_without_a_gpu():
    sys.path.insert(0, ROOT)
    from xhistogram_amd import _native

    if _native.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no HIP device|not available"):
        _native.comm_unique_id()
    with pytest.raises(RuntimeError, match="not available"):
        _native.Comm(0, 0, 1, b"\0" * _native.COMM_ID_BYTES)
    with pytest.raises(ValueError):
        _native.Comm(0, 0, 1, b"short")
    with pytest.raises(ValueError):
        _native.Comm(0, 3, 2, b"\0" * _native.COMM_ID_BYTES)


@pytest.mark.gpu
def test_native_comm_deadline_through_the_python_shim():
    """the same from Python, in a process of its own: Comm(world_size=2) with one rank raises RuntimeError inside the
    deadline, and the communicator-less process goes on to run a single-rank exchange"""
    import subprocess
    import sys

    code = (
        "import os, time, torch\n"
        "from xhistogram_amd import _native\n"
        "_native.load()\n"
        "t0 = time.time()\n"
        "try:\n"
        "    _native.Comm(0, 0, 2, _native.comm_unique_id())\n"
        "    print('NO ERROR')\n"
        "except RuntimeError as e:\n"
        "    print('RAISED after %.1f s: %s' % (time.time() - t0, e))\n"
        "c = _native.Comm(0, 0, 1, _native.comm_unique_id())\n"
        "t = torch.arange(8, dtype=torch.int64, device='cuda')\n"
        "c.allreduce(t.data_ptr(), 8, _native.I64, _native.REDUCE_SUM, torch.cuda.current_stream().cuda_stream)\n"
        "c.wait(torch.cuda.current_stream().cuda_stream)\n"
        "print('SUM', int(t.sum()))\n"
        "c.close()\n"
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", XHIST_AMD_COMM_TIMEOUT_S="4", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240, env=env, cwd=root)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RAISED after" in r.stdout and "rendezvous of 2 ranks" in r.stdout, r.stdout + r.stderr
    assert "SUM 28" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_native_comm_wait_gives_up_on_a_stream_that_does_not_drain():
    """xhist_comm_wait's deadline on ONE GPU: a long spin kernel stands in for a collective whose peer died — the wait returns
    XHIST_ERR_COMM after the deadline (not after the kernel), the communicator is aborted, later calls fail at once"""
    import subprocess
    import sys

    code = (
        "import time, torch\n"
        "from xhistogram_amd import _native\n"
        "_native.load()\n"
        "c = _native.Comm(0, 0, 1, _native.comm_unique_id())\n"
        "s = torch.cuda.current_stream().cuda_stream\n"
        "t = torch.arange(8, dtype=torch.int64, device='cuda')\n"
        "c.allreduce(t.data_ptr(), 8, _native.I64, _native.REDUCE_SUM, s)\n"
        "c.wait(s)\n"
        "torch.cuda._sleep(int(2.0e9 * 8))\n"  # ~7-8 s of GPU time on the stream
        "t0 = time.time()\n"
        "try:\n"
        "    c.wait(s)\n"
        "    print('NO ERROR')\n"
        "except RuntimeError as e:\n"
        "    print('RAISED after %.1f s: %s' % (time.time() - t0, e))\n"
        "try:\n"
        "    c.allreduce(t.data_ptr(), 8, _native.I64, _native.REDUCE_SUM, s)\n"
        "    print('SECOND CALL WENT THROUGH')\n"
        "except RuntimeError as e:\n"
        "    print('AFTERWARDS: %s' % e)\n"
        "c.close()\n"
        "torch.cuda.synchronize()\n"
        "print('DONE', int(t.sum()))\n"
    )
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", XHIST_AMD_COMM_TIMEOUT_S="1.5", PYTHONPATH=root)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240, env=env, cwd=root)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "RAISED after" in r.stdout and "deadline passed with the collective still in flight" in r.stdout, r.stdout + r.stderr
    took = float(r.stdout.split("RAISED after ")[1].split(" s")[0])
    # (the deadline fires after 1.5 s; ncclCommAbort then waits for the device to drain — for RCCL's own kernels that is immediate,
    #  they poll the abort flag; the spin kernel of this test is not RCCL's and runs its 7-8 s out.  A wait that had simply
    #  returned when the stream drained would not have raised at all.)
    assert 1.0 < took < 20.0, r.stdout
    assert "AFTERWARDS" in r.stdout and "was aborted earlier" in r.stdout, r.stdout
    assert "DONE 28" in r.stdout, r.stdout + r.stderr
