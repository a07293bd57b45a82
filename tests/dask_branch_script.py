"""Run under an interpreter that HAS dask (the image's /opt/conda/bin/python3.9; the default
python lacks it).  `lazy`: graph construction only (no GPU needed) — restates the reference's
test_histogram_shape / test_histogram_dask laziness and TypeError rules (test_core.py:231-313,
fixtures.py:8-17).  `compute`: chunked results on the GPU vs numpy (test_chunking.py:8-146)."""
import os
import sys
from itertools import combinations

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dask  # noqa: E402
import dask.array as dsa  # noqa: E402

from xhistogram_amd.core import histogram  # noqa: E402


def forbidden(shape, chunks=None, dtype=float):
    def boom():
        raise ValueError("Triggered forbidden computation")

    a = dsa.from_delayed(dask.delayed(boom)(), shape, dtype)
    return a.rechunk(chunks) if chunks is not None else a


def lazy():
    shape = (10, 15, 12, 20)
    b = forbidden(shape, chunks=(1,) + shape[1:])
    bins = np.linspace(-4, 4, 27)
    for bs in (None, 5, "auto"):
        c, _ = histogram(b, bins=bins, block_size=bs)
        assert c.shape == (26,) and isinstance(c, dsa.Array)
        for axis in list(range(4)) + list(range(-1, -5, -1)):
            c, _ = histogram(b, bins=bins, axis=axis, block_size=bs)
            s = list(shape)
            del s[axis]
            assert c.shape == tuple(s) + (26,) and isinstance(c, dsa.Array)
        for i, j in combinations(range(4), 2):
            c, _ = histogram(b, bins=bins, axis=(i, j), block_size=bs)
            assert c.shape == tuple(shape[k] for k in range(4) if k not in (i, j)) + (26,)
    for axis in [(0, 1, 2, 3), (3, 2, 0, 1)]:
        assert histogram(b, bins=bins, axis=axis)[0].shape == (26,)
    c, _ = histogram(b, b, bins=[bins, bins], weights=forbidden(shape), density=True)
    assert c.shape == (26, 26) and isinstance(c, dsa.Array)
    for bad in (10, "auto"):
        for args, kw in (((b,), {}), ((np.zeros(shape),), {"weights": forbidden(shape)})):
            try:
                histogram(*args, bins=bad, **kw)
            except TypeError:
                pass
            else:
                raise AssertionError("dask + non-array bins must raise TypeError")
    layers = list(histogram(b, bins=bins, axis=(1, 2))[0].dask.layers)
    assert any(k.startswith("bincount") for k in layers) and any(k.startswith("sum") for k in layers), layers
    print("LAZY-OK")


def compute():
    rng = np.random.default_rng(0)
    bins_a, bins_b = np.linspace(-4, 4, 9), np.linspace(-4, 4, 10)
    a, b, w = rng.standard_normal((10, 12)), rng.standard_normal((10, 12)), rng.uniform(0, 1, (10, 12))
    for xc in (1, 3, 10):
        for yc in (2, 12):
            da = dsa.from_array(a, chunks=(xc, yc))
            h, _ = histogram(da, bins=bins_a)
            np.testing.assert_array_equal(h.compute(), np.histogram(a, bins=bins_a)[0])
            h, _ = histogram(da, bins=bins_a, axis=0)
            np.testing.assert_array_equal(h.compute(), np.stack([np.histogram(a[:, j], bins=bins_a)[0] for j in range(12)]))
            db = dsa.from_array(b, chunks=(xc + 1, yc + 1))  # unaligned with da
            h, _ = histogram(da, db, bins=[bins_a, bins_b])
            np.testing.assert_array_equal(h.compute(), np.histogram2d(a.ravel(), b.ravel(), bins=[bins_a, bins_b])[0])
            dw = dsa.from_array(w, chunks=(xc + 1, yc + 1))
            h, _ = histogram(da, bins=bins_a, weights=dw)
            np.testing.assert_allclose(h.compute(), np.histogram(a, bins=bins_a, weights=w)[0], rtol=1e-6)
            h, _ = histogram(da, bins=bins_a, weights=dw, density=True, axis=1)
            want = np.stack([np.histogram(a[i], bins=bins_a, weights=w[i], density=True)[0] for i in range(10)])
            np.testing.assert_allclose(h.compute(), want, rtol=1e-6)
    t = rng.standard_normal((8, 36, 72)).astype(np.float32)  # C4 miniature: chunks on time
    h, _ = histogram(dsa.from_array(t, chunks=(2, 36, 72)), bins=np.linspace(-4, 4, 51), axis=[1, 2])
    np.testing.assert_array_equal(h.compute(scheduler="threads"), np.stack([np.histogram(t[i], bins=np.linspace(-4, 4, 51))[0] for i in range(8)]))
    # the same through the device-resident reduction (partials stay on the GPU, one copy back per output chunk)
    from xhistogram_amd import multigpu
    multigpu.set_dask_exchange("rccl")
    da, db, dw = dsa.from_array(a, chunks=(3, 5)), dsa.from_array(b, chunks=(4, 6)), dsa.from_array(w, chunks=(2, 12))
    h, _ = histogram(da, db, bins=[bins_a, bins_b], weights=dw)
    assert any(k.startswith("reduce_partials") for k in h.dask.layers)
    np.testing.assert_allclose(h.compute(), np.histogram2d(a.ravel(), b.ravel(), bins=[bins_a, bins_b], weights=w.ravel())[0], rtol=1e-6)
    h, _ = histogram(dsa.from_array(t, chunks=(2, 18, 72)), bins=np.linspace(-4, 4, 51), axis=[1, 2])
    np.testing.assert_array_equal(h.compute(scheduler="threads"), np.stack([np.histogram(t[i], bins=np.linspace(-4, 4, 51))[0] for i in range(8)]))
    multigpu.set_dask_exchange(None)
    print("COMPUTE-OK")


def spread():
    """block -> GPU mapping of the dask branch on the CPU: the block adapter is swapped for the oracle (test
    double) and the GPUs are virtual device numbers; what runs for real is the reference-shaped graph, the
    least-busy-GPU assignment of every block task and dask's sum over the reduced chunk dims.  Chunkings as in
    the reference's test_chunking.py:8-146 (aligned / unaligned between the arguments), C2- and C4-shaped."""
    import threading

    from oracle import oracle_np as onp
    from xhistogram_amd import core, multigpu

    seen, lock = [], threading.Lock()

    def oracle_bincount(*all_arrays, weights=False, axis=None, bins=None, density=None, block_size=None):
        arrays = [np.asarray(a) for a in all_arrays]
        w = arrays.pop() if weights else None
        nd = arrays[0].ndim
        ax = tuple(range(nd)) if axis is None else tuple(int(a) for a in axis)
        with lock:
            seen.append(core._host_device())
        h, _ = onp.histogram(*arrays, bins=bins if len(arrays) > 1 else bins[0], weights=w, axis=ax)
        kept = tuple(1 if i in ax else arrays[0].shape[i] for i in range(nd))
        return np.asarray(h).reshape(kept + tuple(len(b) - 1 for b in bins))

    core._bincount = oracle_bincount
    multigpu.set_devices([0, 1, 2, 3, 4, 5, 6, 7])
    multigpu.set_dask_exchange("host")  # first the reference's own graph; the device-resident reduction below
    rng = np.random.default_rng(1)
    # C2-shaped: one long sample axis in 16 chunks, full reduction, weights chunked differently (unaligned)
    x, w = rng.standard_normal(40_000), rng.uniform(0, 1, 40_000)
    e = np.linspace(-4, 4, 101)
    for sched in ("threads", "synchronous"):
        seen.clear()
        h, _ = histogram(dsa.from_array(x, chunks=2500), bins=e, weights=dsa.from_array(w, chunks=3000))
        assert not seen, "graph construction must not compute"
        np.testing.assert_allclose(h.compute(scheduler=sched), np.histogram(x, bins=e, weights=w)[0], rtol=1e-10)
        assert sorted(set(seen)) == list(range(8)), (sched, sorted(set(seen)))  # every GPU got blocks
    # C4-shaped: (time, lat, lon) chunked on time, histogram over lat / lon: disjoint output rows
    t = rng.standard_normal((24, 18, 36)).astype(np.float32)
    e4 = np.linspace(-4, 4, 51)
    want = np.stack([np.histogram(t[i], bins=e4)[0] for i in range(24)])
    for chunks in ((3, 18, 36), (5, 18, 36), (4, 9, 36)):  # aligned, ragged last chunk, chunked along a reduced dim too
        seen.clear()
        h, _ = histogram(dsa.from_array(t, chunks=chunks), bins=e4, axis=[1, 2])
        got = h.compute(scheduler="threads")
        np.testing.assert_array_equal(got, want)
        assert got.dtype == np.int64 and len(set(seen)) >= min(8, len(seen)) - 1, (chunks, seen)
    # two arguments chunked differently (dask rechunks inside blockwise), density on top
    a, b = rng.standard_normal((30, 40)), rng.standard_normal((30, 40))
    ea, eb = np.linspace(-4, 4, 9), np.linspace(-4, 4, 7)
    h, _ = histogram(dsa.from_array(a, chunks=(7, 40)), dsa.from_array(b, chunks=(11, 13)), bins=[ea, eb], density=True)
    np.testing.assert_allclose(h.compute(), np.histogram2d(a.ravel(), b.ravel(), bins=[ea, eb], density=True)[0], rtol=1e-10)
    assert all(v == 0 for v in multigpu._inflight.values())
    multigpu.set_dask_exchange(None)
    # the device-resident reduction (default with more than one GPU): same results through the two-stage graph — here the
    # block adapter double returns host arrays, which reduce_partials adds like the empty blocks of a real run
    assert multigpu.dask_exchange() == "rccl"  # 8 GPUs in use
    seen.clear()
    h, _ = histogram(dsa.from_array(x, chunks=2500), bins=e, weights=dsa.from_array(w, chunks=3000))
    assert not seen and any(k.startswith("reduce_partials") for k in h.dask.layers), list(h.dask.layers)
    np.testing.assert_allclose(h.compute(), np.histogram(x, bins=e, weights=w)[0], rtol=1e-10)
    h, _ = histogram(dsa.from_array(t, chunks=(5, 9, 36)), bins=e4, axis=[1, 2])
    got = h.compute(scheduler="threads")
    np.testing.assert_array_equal(got, want)
    assert got.dtype == np.int64 and got.shape == (24, 50)
    h, _ = histogram(dsa.from_array(a, chunks=(7, 40)), dsa.from_array(b, chunks=(11, 13)), bins=[ea, eb], density=True)
    np.testing.assert_allclose(h.compute(), np.histogram2d(a.ravel(), b.ravel(), bins=[ea, eb], density=True)[0], rtol=1e-10)
    multigpu.set_dask_exchange("host")
    h, _ = histogram(dsa.from_array(x, chunks=2500), bins=e)
    assert any(k.startswith("sum") for k in h.dask.layers) and not any(k.startswith("reduce_partials") for k in h.dask.layers)
    # chunks that already live on the GPUs (DeviceArray): chunk k is placed on GPU k mod 8 and its block runs THERE, whatever
    # the other GPUs are doing.  The double keeps the "device" memory on the host; only placement is under test here.
    from xhistogram_amd import devicearray
    from xhistogram_amd.devicearray import DeviceArray, to_device_chunks

    def host_backed(cls, a, device=None):
        a = np.ascontiguousarray(a)
        return cls(a, a.ctypes.data, a.shape, a.strides, a.dtype, device)

    def read_back(self):
        flat = self.owner.reshape(-1).view(np.uint8)[self.ptr - self.owner.ctypes.data:]
        return np.array(np.lib.stride_tricks.as_strided(flat.view(self.dtype), self.shape, self.strides))

    real = DeviceArray.from_numpy, DeviceArray.to_numpy
    DeviceArray.from_numpy, DeviceArray.to_numpy = classmethod(host_backed), read_back
    try:
        xr = to_device_chunks(dsa.from_array(x, chunks=2500)).persist(scheduler="synchronous")
        assert [xr.blocks[k].compute().device for k in range(16)] == [k % 8 for k in range(16)]
        seen.clear()
        h, _ = histogram(xr, bins=e, weights=dsa.from_array(w, chunks=2500))  # host weight chunks follow the resident ones
        np.testing.assert_allclose(h.compute(scheduler="threads"), np.histogram(x, bins=e, weights=w)[0], rtol=1e-10)
        assert sorted(seen) == sorted(list(range(8)) * 2), seen
        tr = to_device_chunks(dsa.from_array(t, chunks=(3, 18, 36)), devices=[2, 5])
        seen.clear()
        h, _ = histogram(tr, bins=e4, axis=[1, 2])
        np.testing.assert_array_equal(h.compute(scheduler="threads"), want)
        assert sorted(seen) == [2, 2, 2, 2, 5, 5, 5, 5], seen
    finally:
        DeviceArray.from_numpy, DeviceArray.to_numpy = real
    multigpu.set_dask_exchange(None)
    print("SPREAD-OK")


def soak():
    """randomised dask cases on the GPU under the threaded scheduler (the reference's test_chunking_hypotheses.py idea:
    random shapes, chunkings — aligned or not between the arguments —, axes, weights, density) against numpy.  Fresh
    edges in every case, so the plan of each graph is created while its first blocks are already running."""
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    n_cases = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    for case in range(n_cases):
        ndim = int(rng.integers(1, 4))
        shape = tuple(int(rng.integers(1, 9)) for _ in range(ndim - 1)) + (int(rng.integers(1, 400)),)
        d = int(rng.integers(1, 3))
        args = [rng.standard_normal(shape) for _ in range(d)]
        if args[0].size > 5:
            args[0].reshape(-1)[rng.integers(0, args[0].size, 2)] = [np.nan, 7.0]
        edges = [np.sort(rng.uniform(-3, 3, int(rng.integers(2, 30)))) if rng.random() < 0.5 else np.linspace(-3, 3 + 1e-3 * case, int(rng.integers(2, 30)))
                 for _ in range(d)]
        chunk = lambda: tuple(int(rng.integers(1, n + 1)) for n in shape)
        dargs = [dsa.from_array(a, chunks=chunk()) for a in args]
        w = rng.uniform(0, 2, shape) if rng.random() < 0.5 else None
        dw = None if w is None else dsa.from_array(w, chunks=chunk())
        if os.environ.get("XHIST_SOAK_RESIDENT"):
            # chunks that already live on the GPU (DeviceArray), some of them next to host chunks; unaligned chunkings make
            # dask slice and concatenate them on the device
            from xhistogram_amd.devicearray import to_device_chunks
            dargs = [to_device_chunks(a) if rng.random() < 0.75 else a for a in dargs]
            dw = to_device_chunks(dw) if dw is not None and rng.random() < 0.75 else dw
        axes = [None] + [tuple(c) for r in range(1, ndim + 1) for c in combinations(range(ndim), r)]
        axis = axes[int(rng.integers(0, len(axes)))]
        density = bool(rng.random() < 0.3)
        bins = edges if d > 1 else edges[0]
        h, _ = histogram(*dargs, bins=bins, axis=axis, weights=dw, density=density)
        got = h.compute(scheduler="threads")
        ax = tuple(range(ndim)) if axis is None else axis
        moved = [np.moveaxis(a, ax, tuple(range(-len(ax), 0))).reshape(-1, int(np.prod([shape[i] for i in ax]))) for a in args]
        wm = None if w is None else np.moveaxis(w, ax, tuple(range(-len(ax), 0))).reshape(moved[0].shape)
        rows = []
        for r in range(moved[0].shape[0]):
            hh = np.histogramdd([m[r] for m in moved], bins=edges, weights=None if wm is None else wm[r], density=False)[0]
            if density:
                areas = np.ones(())
                for e in edges:
                    areas = np.multiply.outer(areas, np.diff(e))
                with np.errstate(divide="ignore", invalid="ignore"):
                    hh = hh / areas / hh.sum()
            rows.append(hh)
        want = np.stack(rows).reshape(got.shape)
        if w is None and not density:
            np.testing.assert_array_equal(got, want, err_msg="case %d" % case)
        else:
            np.testing.assert_allclose(got, want, rtol=1e-6, atol=0, equal_nan=True, err_msg="case %d" % case)
    print("SOAK-OK %d" % n_cases)


def resident():
    """dask arrays whose chunks already live on the GPU (DeviceArray chunks, no torch under dask): every block is binned where
    it lies — nothing but the partial histograms crosses PCIe — for aligned and unaligned chunkings, both exchange forms."""
    from xhistogram_amd import multigpu
    from xhistogram_amd.devicearray import DeviceArray, to_device_chunks

    rng = np.random.default_rng(0)
    bins_a, bins_b = np.linspace(-4, 4, 9), np.linspace(-4, 4, 10)
    a, b, w = rng.standard_normal((10, 12)), rng.standard_normal((10, 12)).astype(np.float32), rng.uniform(0, 1, (10, 12))
    t = rng.standard_normal((8, 36, 72)).astype(np.float32)
    e50 = np.linspace(-4, 4, 51)
    for exchange in ("host", "rccl"):
        multigpu.set_dask_exchange(exchange)
        da = to_device_chunks(dsa.from_array(a, chunks=(3, 5))).persist()
        assert isinstance(da.blocks[1, 1].compute(), DeviceArray)
        h, _ = histogram(da, bins=bins_a)
        np.testing.assert_array_equal(h.compute(), np.histogram(a, bins=bins_a)[0])
        h, _ = histogram(da, bins=bins_a, axis=0)  # (the persisted chunks are used again)
        np.testing.assert_array_equal(h.compute(), np.stack([np.histogram(a[:, j], bins=bins_a)[0] for j in range(12)]))
        db = to_device_chunks(dsa.from_array(b, chunks=(4, 6)))  # unaligned with da: dask rechunks DeviceArray chunks
        dw = dsa.from_array(w, chunks=(2, 12))  # host chunks next to resident ones
        h, _ = histogram(da, db, bins=[bins_a, bins_b], weights=dw)
        np.testing.assert_allclose(h.compute(scheduler="threads"),
                                   np.histogram2d(a.ravel(), b.astype(np.float64).ravel(), bins=[bins_a, bins_b], weights=w.ravel())[0], rtol=1e-6)
        h, _ = histogram(da, bins=bins_a, weights=to_device_chunks(dw), density=True, axis=1)
        want = np.stack([np.histogram(a[i], bins=bins_a, weights=w[i], density=True)[0] for i in range(10)])
        np.testing.assert_allclose(h.compute(), want, rtol=1e-6)
        dt = to_device_chunks(dsa.from_array(t, chunks=(2, 36, 72))).persist()  # C4 miniature: chunks on time, resident
        h, _ = histogram(dt, bins=e50, axis=[1, 2])
        np.testing.assert_array_equal(h.compute(scheduler="threads"), np.stack([np.histogram(t[i], bins=e50)[0] for i in range(8)]))
        h, _ = histogram(dt, bins=e50, axis=[0, 2])  # reduced axes on both sides of a kept one
        np.testing.assert_array_equal(h.compute(scheduler="threads"), np.stack([np.histogram(t[:, j], bins=e50)[0] for j in range(36)]))
    multigpu.set_dask_exchange(None)
    print("RESIDENT-OK")


if __name__ == "__main__":
    {"lazy": lazy, "compute": compute, "spread": spread, "soak": soak, "resident": resident}[sys.argv[1]]()
