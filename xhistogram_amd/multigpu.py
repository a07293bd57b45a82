"""One ``histogram`` call over the GPUs of a node, from ONE process.

The reference spreads a histogram over workers through its dask branch (core.py:403-439): one
``_bincount`` task per block, then ``bin_counts.sum(drop_axes)`` adds the partial histograms of the
blocks.  On an MI355X node the workers are the (up to 8) GPUs, and this module is the piece that maps
blocks / shards onto them inside a single call:

* **dask inputs** keep the reference's lazy graph; each block task runs on the GPU that has the
  fewest blocks in flight (:func:`block_device`), so the threaded scheduler's concurrent blocks are
  staged over eight PCIe links instead of one and binned on eight GPUs; the partial histograms
  (a few KB each) are summed by the graph's ``sum`` layer exactly as in the reference.
* **host (numpy) inputs** above a size threshold are cut into one shard per GPU
  (:func:`host_sharded_counts`) along a kept axis (disjoint output rows, concatenated — BASELINE C4:
  chunks on ``time``) or a reduced axis (partial histograms, added — C2 / C3 / C5).
* **device-resident inputs** are :class:`Sharded` arrays — one torch tensor per GPU, placed once with
  :func:`scatter` and histogrammed any number of times with :func:`histogram`; the partials never
  leave the GPUs: they are summed by ONE RCCL all-reduce over xGMI (shards cut along a reduced axis)
  or their rows are gathered on the first GPU (kept axis), and the density epilogue runs after it.

The mechanics are the in-process form of "one rank per GPU": a :class:`DeviceGroup` owns one host
thread per GPU; the thread binds its GPU, uses that GPU's plans (``core._get_plan`` is keyed by
device) and streams, and — when the group has more than one GPU and partials live on the GPUs — its
own RCCL communicator (``xhist_comm_*`` of the C ABI; RCCL is thread-safe with one communicator per
thread).  ctypes releases the GIL for the duration of every native call, so the threads overlap.

Which GPUs: ``set_devices`` / ``$XHIST_AMD_DEVICES`` ("all", or "0,2,5"); default = every visible
GPU, except under a one-rank-per-GPU launcher (``$LOCAL_RANK`` / ``$SLURM_LOCALID`` / ``$OMPI_COMM_WORLD_LOCAL_RANK`` / … or ``$XHIST_AMD_DEVICE`` set), where a
process must keep to its own GPU.  Nothing here computes on the CPU: a shard whose GPU path fails
raises.
"""

from __future__ import annotations

import atexit
import contextlib
import os
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import _native, core

__all__ = [
    "set_devices", "get_devices", "visible_devices", "block_device", "on_device", "DeviceGroup", "plan_shards",
    "host_sharded_counts", "Sharded", "scatter", "histogram", "dask_exchange", "set_dask_exchange", "reduce_partials",
]

_range = range
_lock = threading.Lock()
_configured = None  # explicit set_devices(); None = policy below
_inflight = {}  # device -> blocks currently running there (block_device)
_rr = [0]
_groups = {}  # tuple(devices) -> DeviceGroup

_dask_exchange = None  # explicit set_dask_exchange(); None = policy in dask_exchange()

# host inputs are sharded over GPUs only when every shard still moves this many bytes over PCIe: below it a
# call is dominated by per-device fixed costs (plan creation, staging allocation, one more thread hop)
MIN_SHARD_BYTES = 32 << 20


# ---------------------------------------------------------------------------------------------
# which GPUs
# ---------------------------------------------------------------------------------------------
def visible_devices():
    """indices of the MI355X devices HIP shows to this process"""
    return list(_range(_native.device_count()))


def set_devices(devices):
    """GPUs this process spreads dask blocks and host shards over: None (default policy), "all", or a
    sequence of device indices"""
    global _configured
    if devices is None or devices == "all":
        _configured = devices
    else:
        _configured = [int(d) for d in devices]
        if not _configured:
            raise ValueError("at least one device")


def get_devices():
    """the GPUs in use, by the policy in the module docstring"""
    spec = _configured
    if spec is None:
        env = os.environ.get("XHIST_AMD_DEVICES", "").strip()
        if env:
            spec = "all" if env.lower() == "all" else [int(t) for t in env.split(",") if t.strip() != ""]
    if spec is None and core.launcher_local_rank() is not None:
        # a rank of a one-process-per-GPU job (torchrun, srun, mpirun) keeps to its GPU — unless the launcher says this is
        # the ONLY task on the node (srun -n1 / mpirun -n 1 around a single-process multi-GPU job): that one owns them all
        if core.launcher_local_size() != 1 or os.environ.get("XHIST_AMD_DEVICE") not in (None, ""):  # (the library's own variable always pins)
            return [core.default_device()]
    if spec is None or spec == "all":
        return visible_devices() or [core.default_device()]
    return list(spec)


def set_dask_exchange(mode):
    """how the partial histograms of dask blocks are added up: "host" (the reference's graph: each block returns a host
    array, dask's ``sum`` layer adds them), "rccl" (the partials stay on the GPUs that computed them; one task per
    output chunk adds each GPU's partials there and the GPUs' sums with ONE RCCL all-reduce over xGMI), or None = default"""
    global _dask_exchange
    if mode not in (None, "host", "rccl"):
        raise ValueError("dask exchange must be 'host', 'rccl' or None, got %r" % (mode,))
    _dask_exchange = mode


def _scheduler_shares_memory():
    """True when dask's active scheduler runs every task in THIS process (threaded — the default for arrays — or
    synchronous): only then may a block hand a device-resident partial to the reduce task as it is.  Under the
    multiprocessing scheduler or dask.distributed a partial crosses a process boundary; it survives that (DeviceBuffer
    pickles through host memory) but then the reference's own host-side sum is the cheaper graph."""
    try:
        from dask.base import get_scheduler

        get = get_scheduler()
    except Exception:
        return True
    if get is None:
        return True
    mod, name = getattr(get, "__module__", "") or "", getattr(get, "__name__", "") or ""
    return mod.startswith("dask.threaded") or (mod.startswith("dask.local") and name == "get_sync")


def dask_exchange():
    """"rccl" when the blocks are spread over more than one GPU and the scheduler keeps every task in this process (or
    $XHIST_AMD_DASK_EXCHANGE / set_dask_exchange says so), else "host": with one GPU the partials of an output chunk meet
    on the host anyway and the reference's own graph is kept"""
    mode = _dask_exchange or os.environ.get("XHIST_AMD_DASK_EXCHANGE", "").strip().lower() or None
    if mode in ("host", "rccl"):
        return mode
    return "rccl" if len(get_devices()) > 1 and _scheduler_shares_memory() else "host"


@contextlib.contextmanager
def on_device(device):
    """host (numpy) inputs of the calling thread are staged to and binned on GPU ``device``"""
    prev = getattr(core._tls, "device", None)
    core._tls.device = int(device)
    try:
        yield int(device)
    finally:
        core._tls.device = prev


# blocks one GPU stages and bins at a time.  A block holds device memory of its own size while it is in flight and the PCIe link
# is the bottleneck before two of them overlap (tools/dask_c4.py: 26-35 GB/s with 1 to 8 in flight, 45 GB/s one at a time), so more concurrency buys nothing — but dask's threaded scheduler runs
# os.cpu_count() blocks at once (256 on the test box: 256 x a 1.9 GB C4 chunk would not fit one 288 GB GPU)
MAX_BLOCKS_IN_FLIGHT = max(1, int(os.environ.get("XHIST_AMD_BLOCKS_IN_FLIGHT", "2")))
_slot_free = threading.Condition(_lock)


def _pick_block_device(devices):
    """least blocks in flight; ties go round so that a serial scheduler still visits every GPU; waits while every GPU
    has MAX_BLOCKS_IN_FLIGHT blocks"""
    with _slot_free:
        while True:
            n = len(devices)
            start = _rr[0] % n
            order = devices[start:] + devices[:start]
            best = min(order, key=lambda d: _inflight.get(d, 0))
            if _inflight.get(best, 0) < MAX_BLOCKS_IN_FLIGHT:
                break
            _slot_free.wait()
        _inflight[best] = _inflight.get(best, 0) + 1
        _rr[0] = devices.index(best) + 1
        return best


def _release_block_device(dev):
    with _slot_free:
        _inflight[dev] -= 1
        _slot_free.notify()


@contextlib.contextmanager
def block_device():
    """Context of ONE dask block: picks the GPU with the fewest blocks in flight among get_devices(),
    binds the calling thread's host inputs to it, and gives it back afterwards.  A thread that already
    has a GPU (a shard worker) keeps it."""
    if getattr(core._tls, "device", None) is not None:
        yield core._tls.device
        return
    dev = _pick_block_device(get_devices())
    try:
        with on_device(dev) as d:
            yield d
    finally:
        _release_block_device(dev)


# ---------------------------------------------------------------------------------------------
# one host thread per GPU
# ---------------------------------------------------------------------------------------------
_collective_lock = threading.RLock()


class DeviceGroup:
    """One host thread per GPU of ``devices``.  ``run(fn, items)`` executes ``fn(rank, device, item)`` for
    the k-th item on the k-th GPU's thread, all at once, and returns the results in order (the first
    exception is re-raised after every thread has finished).  ``comms()`` creates — once, collectively
    from the threads — one RCCL communicator per GPU for partials that live on the GPUs."""

    def __init__(self, devices):
        self.devices = [int(d) for d in devices]
        if len(set(self.devices)) != len(self.devices):
            raise ValueError("a GPU may appear once in a device group: %s" % (self.devices,))
        self._pools = [ThreadPoolExecutor(max_workers=1, thread_name_prefix="xhist-gpu%d" % d) for d in self.devices]
        self._comms = None
        self._closed = False
        # collectives must be issued in the same order on every GPU: two callers (two dask reduction tasks, two user
        # threads) that interleaved their per-GPU submissions would deadlock the communicators — and so could two GROUPS
        # that share a GPU (reductions over different subsets of the node's GPUs), hence one lock for the process
        self.collective = _collective_lock

    def __len__(self):
        return len(self.devices)

    def run(self, fn, items):
        if len(items) > len(self.devices):
            raise ValueError("%d items for %d GPUs" % (len(items), len(self.devices)))

        def bound(rank, device, item):
            with on_device(device):
                return fn(rank, device, item)

        futures = [self._pools[k].submit(bound, k, self.devices[k], item) for k, item in enumerate(items)]
        results, first_error = [], None
        for f in futures:
            try:
                results.append(f.result())
            except BaseException as exc:  # noqa: BLE001 - collected, re-raised below
                results.append(None)
                first_error = first_error or exc
        if first_error is not None:
            raise first_error
        return results

    def comms(self):
        """one ``_native.Comm`` per GPU (rank k = k-th device); collective creation from the GPU threads.  When the rendezvous
        fails on any rank (its deadline, a GPU that is gone), the communicators the OTHER ranks' threads did get are closed
        here — nothing half-built is kept or leaked."""
        with self.collective:
            if self._comms is None:
                world = len(self.devices)
                uid = _native.comm_unique_id()
                made = [None] * world

                def create(rank, device, _):
                    made[rank] = _native.Comm(device, rank, world, uid)
                    return made[rank]

                try:
                    self._comms = self.run(create, [None] * world)
                except BaseException:
                    for k, c in enumerate(made):
                        if c is not None:
                            try:
                                self._pools[k].submit(c.close).result()
                            except Exception:  # noqa: BLE001 - best effort
                                pass
                    raise
            return self._comms

    def exchange(self, fn, items):
        """``fn(comm, rank, device, item)`` on every GPU's thread, under the collective lock.  ANY exception out of it — a
        deadline (`xhist_comm_wait` / the rendezvous), an asynchronous RCCL error, but also a MemoryError or NotImplementedError
        raised between two collectives on one rank while its peers are already inside the next one — may leave communicators
        aborted or mid-collective; a long-lived worker must not be poisoned by one failed call, so they are
        dropped here and the next exchange builds new ones.

        What runs beside a collective: every collective of this module ends in ``comm.wait`` on the thread that issued it, so a
        GPU's thread never queues its next histogram kernel under its own all-reduce (the rule ``bench.py`` applies to C5 with a
        stream-side wait).  Another thread's kernel on the same GPU can still meet an RCCL kernel; for the exchange mode's
        persistent kernel that costs its arrival handshake (200 us, then the classic passes take that call — DESIGN 4.2b),
        never its deadline."""
        with self.collective:
            try:
                comms = self.comms()
                return self.run(lambda rank, device, item: fn(comms[rank], rank, device, item), items)
            except BaseException:
                self._drop_comms()
                raise

    def _drop_comms(self):
        with self.collective:
            comms, self._comms = self._comms, None
        if comms:
            try:
                self.run(lambda rank, device, c: c.close(), comms)
            except Exception:  # noqa: BLE001 - best effort: an aborted communicator only needs its memory back
                pass

    def close(self):
        if self._closed:
            return
        self._closed = True
        self._drop_comms()
        for p in self._pools:
            p.shutdown(wait=True)


def group_for(devices):
    """the (cached) DeviceGroup of a device list"""
    key = tuple(int(d) for d in devices)
    with _lock:
        g = _groups.get(key)
        if g is None:
            g = _groups[key] = DeviceGroup(key)
        return g


@atexit.register
def _close_groups():
    with _lock:
        groups = list(_groups.values())
        _groups.clear()
    for g in groups:
        g.close()


# ---------------------------------------------------------------------------------------------
# shards
# ---------------------------------------------------------------------------------------------
def shard_bounds(n, parts, k):
    """[start, stop) of the k-th of `parts` contiguous shares of n items (the first n % parts get one more)"""
    base, extra = divmod(int(n), int(parts))
    start = k * base + min(k, extra)
    return start, start + base + (1 if k < extra else 0)


def plan_shards(shape, drop_axes, n_devices):
    """How an N-D block is cut over ``n_devices`` GPUs: ``(mode, axis, bounds)``.

    mode "rows": ``axis`` is a KEPT axis — the shards own disjoint output rows and the results are
    concatenated along it (the reference's chunks along a loop dim; BASELINE C4: ``time``).
    mode "sum": ``axis`` is a REDUCED axis — every shard yields a full-shape partial histogram and the
    partials are added (chunks along a histogrammed dim, core.py:439; C2 / C3 / C5).
    A kept axis is preferred (no exchange arithmetic at all), and among the candidates the outermost
    one that has at least one index per GPU (outer shards are contiguous in C order).  ``bounds`` has
    one (start, stop) per shard — fewer than ``n_devices`` only when no axis is long enough."""
    shape = tuple(int(s) for s in shape)
    drop = [int(a) for a in drop_axes]
    kept = [i for i in _range(len(shape)) if i not in drop]
    for mode, cand in (("rows", kept), ("sum", drop)):
        for ax in cand:
            if shape[ax] >= n_devices:
                return mode, ax, [shard_bounds(shape[ax], n_devices, k) for k in _range(n_devices)]
    best_mode, best_ax = None, None
    for mode, cand in (("rows", kept), ("sum", drop)):
        for ax in cand:
            if best_ax is None or shape[ax] > shape[best_ax]:
                best_mode, best_ax = mode, ax
    if best_ax is None or shape[best_ax] < 2:
        return "sum", (drop[0] if drop else 0), [(0, shape[drop[0]] if drop else 1)]
    n = shape[best_ax]
    return best_mode, best_ax, [shard_bounds(n, n, k) for k in _range(n)]


def _take(a, axis, lo, hi):
    return a[(slice(None),) * axis + (slice(lo, hi),)]


def _slice_raw_weights(w_raw, ndim, axis, lo, hi):
    """shard of the weights AS GIVEN (before broadcasting): cut only where they actually extend"""
    if w_raw is None:
        return None
    wax = axis - (ndim - w_raw.ndim)
    if wax < 0 or w_raw.shape[wax] == 1:
        return w_raw
    return _take(w_raw, wax, lo, hi)


def _input_bytes(arrays):
    """bytes a call moves over PCIe: broadcast (stride-0) dims are staged once"""
    total = 0
    for a in arrays:
        n = 1
        for extent, stride in zip(a.shape, a.strides):
            n *= 1 if stride == 0 else extent
        total += n * a.dtype.itemsize
    return total


def host_sharded_counts(all_arrays, w_raw, n_inputs, has_weights, two, drop_axes, bins, bincount_kwargs, devices=None,
                        exchange=None):
    """core.histogram's counts for HOST (numpy) inputs, computed over several GPUs: one shard per GPU, each
    staged and binned by that GPU's host thread; partials concatenated (kept-axis shards) or added
    (reduced-axis shards).  Returns None when one GPU is the better plan (one GPU visible, or shards
    below MIN_SHARD_BYTES) — the caller then takes the single-GPU route.

    exchange: "host" (default for host inputs: the partials come back over PCIe anyway, a few KB..MB each, and
    are added on the host) or "rccl" (each partial goes back to its GPU and ONE all-reduce over xGMI adds
    them — $XHIST_AMD_EXCHANGE=rccl; the route device-resident shards always take, see histogram())."""
    devs = list(devices) if devices is not None else get_devices()
    if len(devs) < 2:
        return None
    if devices is None:
        n_use = min(len(devs), _input_bytes(all_arrays) // MIN_SHARD_BYTES)
        if n_use < 2:
            return None
        devs = devs[:n_use]
    shape = all_arrays[0].shape
    if any(s == 0 for s in shape):
        return None
    mode, ax, bounds = plan_shards(shape, drop_axes, len(devs))
    if len(bounds) < 2:
        return None
    devs = devs[: len(bounds)]
    ndim = len(shape)
    exchange = exchange or os.environ.get("XHIST_AMD_EXCHANGE", "host")
    if exchange not in ("host", "rccl"):
        raise ValueError("exchange must be 'host' or 'rccl', got %r" % (exchange,))
    group = group_for(devs)

    def one(rank, device, span):
        lo, hi = span
        shard = [_take(a, ax, lo, hi) for a in all_arrays]
        w_shard = _slice_raw_weights(w_raw, ndim, ax, lo, hi)
        return core._counts_one_device(shard, w_shard, n_inputs, has_weights, two, drop_axes, bins, bincount_kwargs, "numpy")

    parts = group.run(one, bounds)
    out_axis = ax + (1 if two else 0)
    if mode == "rows":
        return np.concatenate(parts, axis=out_axis)
    if exchange == "rccl":
        return _allreduce_host_partials(group, parts)
    total = parts[0].copy()
    for p in parts[1:]:
        total += p
    return total


def _allreduce_host_partials(group, parts):
    """partials (numpy, one per GPU of the group) -> their sum, by ONE RCCL all-reduce: each GPU's thread
    uploads its partial into a device buffer of the library, joins the all-reduce, and the first GPU's
    thread brings the result back"""
    tag = _native.F64 if parts[0].dtype == np.float64 else _native.I64
    shape, dtype = parts[0].shape, parts[0].dtype
    count = int(parts[0].size)
    with group.collective:
        return _allreduce_host_partials_locked(group, parts, tag, shape, dtype, count)


def _allreduce_host_partials_locked(group, parts, tag, shape, dtype, count):
    def one(comm, rank, device, part):
        buf = _native.DeviceBuffer(device, count * 8)
        try:
            buf.upload(np.ascontiguousarray(part))
            comm.allreduce(buf.ptr, count, tag, _native.REDUCE_SUM, 0)
            comm.wait(0)  # (a deadline instead of a hang when a peer GPU never enters the collective)
            if rank != 0:
                return None
            out = np.empty(shape, dtype)
            buf.download(out)
            return out
        finally:
            buf.close()

    return group.exchange(one, list(parts))[0]


# ---------------------------------------------------------------------------------------------
# dask: partial histograms that stay on their GPUs
# ---------------------------------------------------------------------------------------------
def _flatten(nested):
    if isinstance(nested, (list, tuple)):
        for item in nested:
            yield from _flatten(item)
    else:
        yield nested


def reduce_partials(nested, drop_axes=(), out_dtype="<i8", _allreduce=None, _alloc=None):
    """Second stage of the dask graph under ``dask_exchange() == "rccl"`` — replaces ``bin_counts.sum(drop_axes)``
    (core.py:439) for ONE output chunk.  ``nested`` holds the partial histograms of every block that contributes to the
    chunk (``_native.DevicePartial`` on the GPU that computed each; host arrays for empty blocks), all of one shape with
    the reduced axes as single-element dims.  The partials of each GPU are added up on that GPU (``xhist_buffer_add``) in a
    buffer of this task's OWN — the upstream partials are task outputs and stay untouched, so a re-run of this task (a
    retry, a recompute) sees what the first run saw; they are freed when dask drops them — the GPUs' sums by ONE in-place
    RCCL all-reduce issued from the GPUs' host threads, and the first GPU's copy comes back to the host: one
    device-to-host copy per output chunk instead of one per block.  The all-reduce always spans the WHOLE configured
    device group (GPUs without a partial for this chunk contribute zeros), so one set of communicators serves every chunk
    instead of one per subset of GPUs.  ``_allreduce`` / ``_alloc`` are test seams."""
    parts = list(_flatten(nested))
    dtype = np.dtype(out_dtype)
    host = [np.asarray(p) for p in parts if not isinstance(p, _native.DevicePartial)]
    on_gpu = [p for p in parts if isinstance(p, _native.DevicePartial)]
    total = None
    if on_gpu:
        shape, pdtype, count = on_gpu[0].shape, on_gpu[0].dtype, on_gpu[0].size
        tag = _native.F64 if pdtype == np.float64 else _native.I64
        by_dev = {}
        for p in on_gpu:
            assert p.shape == shape and p.dtype == pdtype, (p.shape, shape)
            by_dev.setdefault(p.device, []).append(p)
        group_devices = sorted(set(get_devices()) | set(by_dev)) if len(by_dev) > 1 else sorted(by_dev)
        alloc = _alloc or (lambda device, nbytes: _native.DeviceBuffer(device, nbytes))
        zeros = np.zeros(count, pdtype)
        sums = []
        try:
            for d in group_devices:  # this GPU's partials -> a zeroed buffer of this task (same GPU, NULL stream: ordered)
                acc = _native.DevicePartial(alloc(d, count * pdtype.itemsize), shape, pdtype)
                sums.append(acc)
                acc.buf.upload(zeros)
                for other in by_dev.get(d, ()):
                    acc.buf.add(other.buf, count, tag)
            if len(group_devices) > 1:
                if _allreduce is not None:
                    _allreduce(sums, count, tag)
                else:
                    group = group_for(group_devices)

                    def one(comm, rank, device, acc):
                        comm.allreduce(acc.buf.ptr, count, tag, _native.REDUCE_SUM, 0)
                        comm.wait(0)

                    group.exchange(one, sums)
            total = sums[0].to_numpy()  # (download waits for the NULL stream of that GPU)
        finally:
            for acc in sums:
                acc.buf.close()
    for h in host:
        total = h.astype(dtype, copy=True) if total is None else total + h
    if total is None:
        raise ValueError("no partial histograms to reduce")
    return total.squeeze(tuple(drop_axes)).astype(dtype, copy=False)


# ---------------------------------------------------------------------------------------------
# device-resident shards
# ---------------------------------------------------------------------------------------------
class Sharded:
    """An array cut along ``axis`` into one torch tensor per GPU (``parts[k]`` lives on ``devices[k]``) — the
    in-process counterpart of a dask array chunked along one dim with its chunks pinned to GPUs."""

    def __init__(self, parts, axis, devices=None):
        self.parts = list(parts)
        self.axis = int(axis)
        self.devices = [int(d) for d in devices] if devices is not None else [p.device.index for p in self.parts]
        if len(self.parts) != len(self.devices) or not self.parts:
            raise ValueError("one part per device")
        nd = self.parts[0].ndim
        if not -nd <= self.axis < nd:
            raise ValueError("shard axis %d out of range for %d-D parts" % (self.axis, nd))
        self.axis %= nd
        for p in self.parts[1:]:
            a, b = list(p.shape), list(self.parts[0].shape)
            a[self.axis] = b[self.axis] = 0
            if a != b:
                raise ValueError("parts may differ only along the shard axis")

    @property
    def ndim(self):
        return self.parts[0].ndim

    @property
    def shape(self):
        s = list(self.parts[0].shape)
        s[self.axis] = sum(int(p.shape[self.axis]) for p in self.parts)
        return tuple(s)


def scatter(array, devices=None, axis=0):
    """Cut ``array`` (numpy or torch) along ``axis`` into contiguous shares and place one on each GPU
    (host to device over each GPU's own PCIe link, or peer copies over xGMI for a device tensor)."""
    torch = core._torch()
    devs = list(devices) if devices is not None else get_devices()
    n = int(array.shape[axis])
    devs = devs[: max(1, min(len(devs), n))]
    group = group_for(devs)

    def one(rank, device, _):
        lo, hi = shard_bounds(n, len(devs), rank)
        piece = _take(array, axis % array.ndim, lo, hi)
        if not core._is_torch(piece):
            piece = torch.from_numpy(np.ascontiguousarray(piece))
        return piece.to(torch.device("cuda", _native.physical_device(device))).contiguous()

    return Sharded(group.run(one, [None] * len(devs)), axis % array.ndim, devs)


def _global_edges(args, bins, ranges, group, _extrema, _moments=None):
    """np.histogram_bin_edges (core.py:383-388) of data spread over GPUs: explicit edges and ranges go through
    numpy; an integer ``bins`` with no range needs the GLOBAL min / max — reduced on every GPU by the library's
    kernel, combined on the host (two numbers per GPU)"""
    out = []
    for a, b, r in zip(args, bins, ranges):
        proto = core._np_dtype_of(a.parts[0])
        if isinstance(b, str):
            # "sqrt" / "sturges" / "rice" / "scott": every GPU reduces its shard to five numbers, combined on the host
            ok, lo_hi = core._estimator_cut(b, r, proto)
            if not ok:
                raise TypeError("When the data is sharded over GPUs, bins must be edges, an int or one of %s (the other "
                                "estimators need all the data in one place)" % (core.ESTIMATORS_FROM_MOMENTS,))
            get = _moments or core._device_moments
            parts = group.run(lambda rank, device, part: get(part, lo_hi, b == "scott"), a.parts)
            size = sum(int(np.prod(p.shape)) for p in a.parts)
            edges = core._edges_from_moments(b, r, proto, size, core.combine_moments(parts))
            if edges is None:  # (nearly constant data or a tie under "scott": numpy's own summation order decides)
                whole = np.concatenate([np.asarray(p.detach().cpu().numpy() if core._is_torch(p) else p).reshape(-1) for p in a.parts])
                edges = np.histogram_bin_edges(whole, bins=b, range=r)
            out.append(edges)
            continue
        if np.ndim(b) == 0 and r is None:
            ext = group.run(lambda rank, device, part: _extrema(part), a.parts)
            nan = any(e[2] for e in ext)
            lo, hi = min(e[0] for e in ext), max(e[1] for e in ext)
            if nan:
                lo = hi = np.nan
            if not nan and lo > hi:  # every shard empty: numpy's (0, 1) default
                out.append(np.histogram_bin_edges(np.zeros(0, proto), bins=b, range=None))
            else:
                out.append(np.histogram_bin_edges(np.array([lo, hi]).astype(proto), bins=b, range=None))
        else:
            out.append(np.histogram_bin_edges(np.zeros(0, proto), bins=b, range=r))
    return out


# Test hooks (module-private, not part of any signature): the CPU tests run the sharding and exchange logic with the oracle as
# the per-shard compute and a plain sum as the exchange.  Set only through `_hooks(...)` in tests/; production never does.
_test_hooks = {"local": None, "reduce": None, "extrema": None, "moments": None}


@contextlib.contextmanager
def _hooks(**kw):
    """tests only: `with multigpu._hooks(local=fn, reduce=fn): ...` for the duration of the block"""
    unknown = set(kw) - set(_test_hooks)
    if unknown:
        raise TypeError("unknown hook(s): %s" % sorted(unknown))
    saved = dict(_test_hooks)
    _test_hooks.update(kw)
    try:
        yield
    finally:
        _test_hooks.update(saved)


def histogram(*args, bins=None, range=None, axis=None, weights=None, density=False, block_size="auto", exchange="rccl"):
    """``core.histogram`` of :class:`Sharded` inputs (same shard axis and devices for all; ``weights`` a
    Sharded too, or an array every shard broadcasts against).  Each GPU's thread bins its shard with the
    fused kernel; then

    * shard axis histogrammed over  -> ONE all-reduce(sum) of the partial histograms over RCCL / xGMI
      (``exchange="rccl"``; ``"p2p"``: peer copies to the first GPU and a sum there), int64 exact;
    * shard axis kept               -> the shards' rows are copied to the first GPU and concatenated;

    density (core.py:444-462) is applied after the exchange.  Returns ``(hist, edges)`` with ``hist`` a
    torch tensor on ``devices[0]``."""
    _local, _reduce, _extrema, _moments = (_test_hooks[k] for k in ("local", "reduce", "extrema", "moments"))
    if not args or not all(isinstance(a, Sharded) for a in args):
        raise TypeError("multigpu.histogram takes Sharded inputs (see multigpu.scatter); plain arrays go to core.histogram")
    first = args[0]
    for a in args[1:]:
        if a.axis != first.axis or a.devices != first.devices or [tuple(p.shape) for p in a.parts] != [tuple(p.shape) for p in first.parts]:
            raise ValueError("every input must be sharded the same way (axis, devices, part shapes)")
    n_inputs, ndim, world = len(args), first.ndim, len(first.devices)
    group = group_for(first.devices)
    axis = core._normalise_axis(axis, ndim)
    drop_axes = tuple(axis) if axis is not None else tuple(_range(ndim))
    has_weights = weights is not None
    if has_weights and isinstance(weights, Sharded):
        if weights.devices != first.devices:
            raise ValueError("weights must be sharded over the same devices")
        if weights.axis != first.axis or [tuple(p.shape) for p in weights.parts] != [tuple(p.shape) for p in first.parts]:
            raise ValueError("sharded weights must be cut like the inputs (axis %d, part shapes %s); weights that broadcast "
                             "go in as ONE plain tensor" % (first.axis, [tuple(p.shape) for p in first.parts]))
        w_parts = weights.parts
    else:
        w_parts = [weights] * world
    if _extrema is None:
        from .distributed import _local_extrema as _extrema
    bins = core._ensure_correctly_formatted_bins(bins, n_inputs)
    range = core._ensure_correctly_formatted_range(range, n_inputs)
    edges = _global_edges(args, bins, range, group, _extrema, _moments)
    kwargs = dict(weights=has_weights, axis=axis, bins=edges, density=False, block_size=block_size)
    reduce_shards = first.axis in drop_axes

    def local(rank, device, _):
        torch = core._torch()
        arrays = [a.parts[rank] for a in args]
        w = w_parts[rank]
        if _local is not None:
            return _local(arrays + ([w] if has_weights else []), has_weights, axis, edges, block_size)
        torch.cuda.set_device(_native.physical_device(device))
        dev = arrays[0].device
        w_raw = None
        if has_weights:
            w_raw = w.to(dev) if core._is_torch(w) else torch.as_tensor(np.asarray(w)).to(dev)
        allb = list(torch.broadcast_tensors(*(arrays + ([w_raw] if has_weights else []))))
        return core._counts_one_device(allb, w_raw, n_inputs, has_weights, False, drop_axes, edges, kwargs, "torch")

    parts = group.run(local, [None] * world)

    if not reduce_shards:
        torch = core._torch()
        home = parts[0].device
        counts = torch.cat([p.to(home) for p in parts], dim=first.axis)
    elif _reduce is not None:
        counts = _reduce(parts)
    elif world == 1:
        counts = parts[0]
    elif exchange == "p2p":
        home = parts[0].device
        counts = parts[0].clone()
        for p in parts[1:]:
            counts += p.to(home)
    elif exchange == "rccl":
        def allreduce(comm, rank, device, t):
            torch = core._torch()
            torch.cuda.set_device(_native.physical_device(device))
            t = t.contiguous()
            tag = {torch.int64: _native.I64, torch.float64: _native.F64, torch.float32: _native.F32}[t.dtype]
            stream = torch.cuda.current_stream(t.device).cuda_stream
            comm.allreduce(t.data_ptr(), t.numel(), tag, _native.REDUCE_SUM, stream)
            comm.wait(stream)
            return t

        counts = group.exchange(allreduce, parts)[0]
    else:
        raise ValueError("exchange must be 'rccl' or 'p2p', got %r" % (exchange,))

    keep = [s for i, s in enumerate(counts.shape) if i not in drop_axes]
    counts = counts.reshape(keep)
    h = core._density(counts, edges, n_inputs) if density else counts
    return h, edges
