"""Numpy-style API of the MI355X-native xhistogram hot path.

Drop-in for ``xhistogram.core`` (reference: /root/reference/xhistogram/core.py): ``histogram``
keeps the reference's exact signature and semantics (core.py:250-466) and so do the two internal
layers the reference's dask branch and xarray wrapper call, ``_bincount`` (core.py:197-247) and
``_bincount_2d_vectorized`` (core.py:137-194).  What changes is what runs underneath:
``_bincount_2d_vectorized`` hands strided [rows, cols] views to ``libxhist_amd.so`` and ONE fused HIP
kernel does digitize -> joint index -> scatter-add on the GPU (see csrc/xhist_kernels.hip.h).

Inputs may be numpy arrays (staged to the GPU by the library, result returned as numpy, exactly
like the reference), ``torch`` tensors resident on an MI355X (no copies; result returned as a
torch tensor on the same device) or, when dask is importable, dask arrays (the reference's
blockwise + sum graph, each block computed on the GPU).  There is no CPU implementation in this
package: without the native library and a GPU the compute call raises.
"""

from __future__ import annotations

import os
import re
import threading
from collections import OrderedDict
from collections.abc import Iterable

import warnings

import numpy as np

from . import _native
from .devicearray import DeviceArray, _nocopy_reshape_strides

# range is a keyword of histogram(), like in the reference
_range = range

__all__ = ["histogram", "histogram_two_weights"]


# ---------------------------------------------------------------------------------------------
# backends: numpy (host memory) and torch (device memory)
# ---------------------------------------------------------------------------------------------
def _is_torch(a):
    return type(a).__module__.split(".")[0] == "torch" and hasattr(a, "data_ptr")


def _is_devarr(a):
    return type(a) is DeviceArray


def _backend_of(arrays):
    """"torch" (GPU tensors), "device" (DeviceArray: GPU memory without torch) or "numpy" (host memory)"""
    if any(_is_torch(a) for a in arrays):
        return "torch"
    return "device" if any(_is_devarr(a) for a in arrays) else "numpy"


def _is_dask(a):
    mod = type(a).__module__.split(".")[0]
    return mod == "dask" and hasattr(a, "chunks")


def _torch():
    import torch

    return torch


_TORCH_TAGS = None


def _torch_tag(dtype):
    global _TORCH_TAGS
    if _TORCH_TAGS is None:
        t = _torch()
        m = {
            t.float64: _native.F64, t.float32: _native.F32, t.float16: _native.F16, t.int64: _native.I64,
            t.int32: _native.I32, t.int16: _native.I16, t.int8: _native.I8, t.uint8: _native.U8, t.bool: _native.BOOL,
        }
        for name, tag in (("uint16", _native.U16), ("uint32", _native.U32), ("uint64", _native.U64)):
            if hasattr(t, name):
                m[getattr(t, name)] = tag
        _TORCH_TAGS = m
    try:
        return _TORCH_TAGS[dtype]
    except KeyError:
        raise TypeError("torch dtype %s is not supported by the MI355X histogram path" % dtype) from None


_TAG_NP = {
    _native.F64: np.float64, _native.F32: np.float32, _native.F16: np.float16, _native.I64: np.int64,
    _native.I32: np.int32, _native.I16: np.int16, _native.I8: np.int8, _native.U64: np.uint64,
    _native.U32: np.uint32, _native.U16: np.uint16, _native.U8: np.uint8, _native.BOOL: np.bool_,
}


# the local rank of a one-process-per-GPU job, as the common launchers export it (torch.distributed.run, srun, Open MPI,
# MVAPICH2 / MPICH-hydra): such a process keeps to ITS GPU instead of spreading over the node's
LOCAL_RANK_VARS = ("XHIST_AMD_DEVICE", "LOCAL_RANK", "SLURM_LOCALID", "OMPI_COMM_WORLD_LOCAL_RANK", "MV2_COMM_WORLD_LOCAL_RANK",
                   "MPI_LOCALRANKID")


def launcher_local_rank():
    """the first of LOCAL_RANK_VARS that is set, as an int; None outside a one-process-per-GPU launcher"""
    for key in LOCAL_RANK_VARS:
        v = os.environ.get(key)
        if v not in (None, ""):
            try:
                return int(v)
            except ValueError:
                continue
    return None


# how many tasks the launcher started on THIS node (torch.distributed.run, Slurm, Open MPI, MVAPICH2, MPICH-hydra)
LOCAL_SIZE_VARS = ("LOCAL_WORLD_SIZE", "SLURM_NTASKS_PER_NODE", "SLURM_STEP_TASKS_PER_NODE", "SLURM_TASKS_PER_NODE",
                   "OMPI_COMM_WORLD_LOCAL_SIZE", "MV2_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS")


def _slurm_tasks_of_node(spec, node):
    """this node's entry of Slurm's per-node task list — "4(x2),1" = nodes 0 and 1 run four tasks, node 2 one — or None
    when the list cannot be read or does not reach `node`"""
    counts = []
    for item in spec.strip().split(","):
        m = re.fullmatch(r"\s*(\d+)(?:\(x(\d+)\))?\s*", item)
        if not m:
            return None
        counts += [int(m.group(1))] * int(m.group(2) or 1)
    return counts[node] if 0 <= node < len(counts) else None


def launcher_local_size():
    """tasks of this job on THIS node as the launcher reports them, or None when it does not say.  Slurm's
    SLURM_(STEP_)TASKS_PER_NODE lists every node ("1,4", "1(x2),4"): the entry of $SLURM_NODEID is taken (the
    leading integer is the FIRST node's count); a list of several nodes without a readable node id says nothing."""
    for key in LOCAL_SIZE_VARS:
        v = os.environ.get(key)
        if v in (None, ""):
            continue
        if key in ("SLURM_STEP_TASKS_PER_NODE", "SLURM_TASKS_PER_NODE"):
            try:
                node = int(os.environ.get("SLURM_NODEID", ""))
            except ValueError:
                node = 0 if re.fullmatch(r"\s*\d+\s*", v) else -1  # (one node, one figure: unambiguous without an id)
            n = _slurm_tasks_of_node(v, node)
            if n is not None:
                return n
            continue
        try:
            return int(v.strip())
        except ValueError:
            continue
    return None


def default_device():
    """GPU used for host (numpy) inputs: $XHIST_AMD_DEVICE, else the launcher's local rank ($LOCAL_RANK, $SLURM_LOCALID,
    $OMPI_COMM_WORLD_LOCAL_RANK, …) — taken modulo the number of GPUs this process can see: a task the launcher bound to ONE
    GPU (srun --gpus-per-task=1 / --gpu-bind, ROCR_VISIBLE_DEVICES: every task sees its GPU as device 0) has local rank k
    and device 0 — else 0."""
    explicit = os.environ.get("XHIST_AMD_DEVICE")
    if explicit not in (None, ""):
        try:
            return int(explicit)  # (the user's own choice is not second-guessed: a wrong index fails loudly in require_device)
        except ValueError:
            pass
    r = launcher_local_rank()
    if r is None:
        return 0
    try:
        n = _native.device_count()
    except Exception:
        n = 0
    return r % n if n > 0 else r


_tls = threading.local()


def _torch_device_index(device):
    """(logical) device index of this library for a torch.device: the tensor's own index — except under
    $XHIST_AMD_DEVICE_ALIAS (tests: several logical devices on one GPU), where a thread that multigpu bound to logical
    device k keeps k for the tensors that live on k's physical GPU"""
    idx = device.index if device.index is not None else _torch().cuda.current_device()
    bound = getattr(_tls, "device", None)
    if bound is not None and bound != idx and os.environ.get("XHIST_AMD_DEVICE_ALIAS") and _native.physical_device(bound) == idx:
        return bound
    return idx


def _host_device():
    """GPU the calling thread stages host (numpy) inputs to: the one its multi-GPU driver bound it to
    (multigpu.on_device: a shard worker, or one dask block), else default_device()"""
    d = getattr(_tls, "device", None)
    return default_device() if d is None else d


def _np_dtype_of(a):
    """numpy dtype describing the element type of a numpy array or torch tensor."""
    if _is_torch(a):
        return np.dtype(_TAG_NP[_torch_tag(a.dtype)])
    return a.dtype


# ---------------------------------------------------------------------------------------------
# plan cache (edge tables live on the GPU; building one costs a launch + a sync)
# ---------------------------------------------------------------------------------------------
_plans = OrderedDict()
_areas = OrderedDict()  # device-resident bin areas of the density epilogue, per set of edges
_plans_lock = threading.Lock()
_plan_create_lock = threading.Lock()
_PLAN_CACHE = 32


def _get_plan(edges, cmp_domain, device):
    key = (device, cmp_domain) + tuple((e.dtype.str, e.tobytes()) for e in edges)
    with _plans_lock:
        plan = _plans.get(key)
        if plan is not None:
            _plans.move_to_end(key)
            return plan
    # one creation at a time (and no duplicates): building the tables costs a launch and a device synchronisation, and
    # dask's threaded scheduler sends every block of a fresh graph here at once
    with _plan_create_lock:
        with _plans_lock:
            plan = _plans.get(key)
        if plan is None:
            plan = _native.Plan(edges, cmp_domain, device)
            with _plans_lock:
                _plans[key] = plan
                while len(_plans) > _PLAN_CACHE:
                    _plans.popitem(last=False)
    return plan


def _compare_domain(sample_dtypes, edges):
    """Decide how samples are compared with edges, following numpy's promotion in searchsorted
    (core.py:170): float64 compares unless BOTH sides are integers / datetimes, then exact int64;
    decided per input (CMP_PER_DIM | mask when the inputs differ).
    Returns (cmp_domain, edges converted to the domain's dtype, per-input 'view as int64' flag)."""
    doms = []
    conv = []
    for sd, e in zip(sample_dtypes, edges):
        e = np.asarray(e)
        if e.ndim != 1:
            raise AssertionError("bin edges must be 1-D")  # core.py:148
        if sd.kind in "mM" or e.dtype.kind in "mM":
            common = np.result_type(sd, e.dtype)  # TypeError when only one side is a datetime
            conv.append(e.astype(common).view(np.int64))
            doms.append((_native.CMP_I64, common))
            continue
        if sd.kind == "c" or e.dtype.kind == "c":
            raise TypeError("complex samples / bin edges are not supported")
        if sd.kind not in "fiub" or e.dtype.kind not in "fiub":
            raise TypeError("cannot histogram dtype %s against bin edges of dtype %s" % (sd, e.dtype))
        if sd.itemsize > 8 or e.dtype.itemsize > 8:
            raise TypeError("extended-precision floats are not supported on the GPU path")
        common = np.result_type(sd, e.dtype)
        if common.kind == "f":
            conv.append(e.astype(np.float64))
            doms.append((_native.CMP_F64, None))
        elif common == np.dtype(np.uint64):
            # unsigned on both sides: the int64 domain with the sign bit flipped (CMP_UNSIGNED)
            conv.append(e.astype(np.uint64))
            doms.append((_native.CMP_I64, "unsigned"))
        else:
            conv.append(e.astype(np.int64))
            doms.append((_native.CMP_I64, None))
    # Integer samples of at most 32 bits against integer edges within +-2^53: both sides are exact
    # in float64, so the comparison may run there — which is where the vector kernels are
    # (`bins=np.arange(257)` on uint8 / int32 data: 3-4x the generic int64 family).
    unsigned = [c == "unsigned" for _, c in doms]
    doms = [(d, None if c == "unsigned" else c) for d, c in doms]
    if any(unsigned):
        # one flag per plan: every int64-domain input must then be unsigned
        if any(d == _native.CMP_I64 and not u for (d, _), u in zip(doms, unsigned)):
            raise NotImplementedError("uint64 inputs together with signed 64-bit integer / datetime inputs")
        mask = sum(1 << k for k, (d, _) in enumerate(doms) if d == _native.CMP_I64)
        full = mask == (1 << len(doms)) - 1
        return (_native.CMP_I64 if full else _native.CMP_PER_DIM | mask) | _native.CMP_UNSIGNED, conv, [None] * len(doms)

    def small_exact(k):
        d, common = doms[k]
        e = np.asarray(edges[k])
        return (d == _native.CMP_I64 and common is None and sample_dtypes[k].itemsize <= 4 and
                (e.size == 0 or (int(e.min()) >= -(1 << 53) and int(e.max()) <= (1 << 53))))
    if any(d == _native.CMP_I64 for d, _ in doms) and all(d == _native.CMP_F64 or small_exact(k) for k, (d, _) in enumerate(doms)):
        conv = [np.asarray(e).astype(np.float64) for e in edges]
        return _native.CMP_F64, conv, [None] * len(edges)
    kinds = {d for d, _ in doms}
    if len(kinds) > 1:
        # 64-bit integer / datetime inputs next to float ones (a time axis against a value axis):
        # every input keeps its own domain, as numpy digitizes every argument on its own
        mask = sum(1 << k for k, (d, _) in enumerate(doms) if d == _native.CMP_I64)
        return _native.CMP_PER_DIM | mask, conv, [c for _, c in doms]
    return doms[0][0], conv, [c for _, c in doms]


# ---------------------------------------------------------------------------------------------
# L1: the hot path                                                     (core.py:137-194)
# ---------------------------------------------------------------------------------------------
def _torch_contiguous(a):
    """C-contiguous copy of a GPU tensor: the library's strided-copy kernel (transposing layouts tiled through LDS, 3-5 TB/s
    moved) instead of torch's generic one (~0.5 TB/s for the same layouts); small tensors and dtypes without a tag go to torch"""
    if a.is_contiguous():
        return a
    if a.numel() < (1 << 16) or a.ndim > 8 or not a.is_cuda:
        return a.contiguous()
    try:
        tag = _torch_tag(a.dtype)
    except TypeError:
        return a.contiguous()
    torch = _torch()
    out = torch.empty(a.shape, dtype=a.dtype, device=a.device)
    item = a.element_size()
    dev = _torch_device_index(a.device)
    _native.copy_nd(dev, a.shape, a.data_ptr(), tag, [st * item for st in a.stride()], out.data_ptr(), tag,
                    [st * item for st in out.stride()], torch.cuda.current_stream(a.device).cuda_stream)
    return out


def _strided_view(a2d, backend):
    """(pointer, dtype tag, row stride, col stride, 0, 0, keepalive) of a 2-D array, in elements.
    Falls back to a contiguous copy only for layouts the C ABI does not take (negative strides;
    arrays strided in both directions)."""
    if backend == "torch":
        rs, cs = a2d.stride()
        # A large view strided in BOTH directions (e.g. a reduction over a middle axis) would make
        # every lane touch its own cache line; one copy on the device (the reference makes the
        # same copy on the host, core.py:219-226) and the coalesced kernels is far cheaper.
        # (row stride 1 = reductions over leading axes: the library's row-per-lane kernels take
        # those views as they are)
        if rs < 0 or cs < 0 or (cs > 1 and rs > 1 and a2d.shape[1] > 1 and a2d.numel() >= (1 << 16)):
            a2d = _torch_contiguous(a2d)
            rs, cs = a2d.stride()
        if a2d.shape[0] <= 1:
            rs = 0 if a2d.shape[0] == 0 else rs
        return a2d.data_ptr(), _torch_tag(a2d.dtype), rs, cs, 0, 0, a2d
    if backend == "device":
        item = a2d.itemsize
        if any(st < 0 or st % item for st in a2d.strides) or (
                a2d.strides[0] > item and a2d.strides[1] > item and a2d.shape[1] > 1 and a2d.size >= (1 << 16)):
            a2d = a2d.copy()  # (the same two cases as for torch tensors above)
        rs, cs = (st // item for st in a2d.strides)
        if a2d.shape[0] <= 1:
            rs = 0 if a2d.shape[0] == 0 else rs
        return a2d.ptr, _native.dtype_tag(np.dtype(np.int64) if a2d.dtype.kind in "mM" else a2d.dtype), rs, cs, 0, 0, a2d
    if a2d.dtype.kind in "mM":
        a2d = a2d.view(np.int64)
    item = a2d.dtype.itemsize
    rs, cs = (s // item for s in a2d.strides)
    ok = rs >= 0 and cs in (0, 1) and all(s % item == 0 for s in a2d.strides) and (rs == 0 or rs >= a2d.shape[1] * cs)
    if a2d.shape[1] <= 1 and rs >= 0 and a2d.strides[0] % item == 0:
        ok, cs = True, 1  # (a byte stride that is no multiple of the item size — a field of a packed record — is copied)
    if rs == 1 and cs >= a2d.shape[0] and all(s % item == 0 for s in a2d.strides):
        ok = True  # rows are the contiguous direction (leading-axis reduction): staged as it lies
    if not ok or not a2d.dtype.isnative:
        a2d = np.ascontiguousarray(a2d, dtype=a2d.dtype.newbyteorder("="))
        rs, cs = a2d.shape[1], 1
    return a2d.ctypes.data, _native.dtype_tag(a2d.dtype), rs, cs, 0, 0, a2d


def _execute_views(views, wview, nrows, ncols, sample_dtypes, bins, backend, like, block_size, wview2=None):
    """Hand [rows, cols] views (ptr, tag, row stride, col stride, rows per group, group stride,
    keepalive) to the native library and return the [rows, nb_0, ...] histogram.  With ``wview2``
    (a second weight view): a [2, rows, nb_0, ...] pair from one pass (histogram_two_weights)."""
    cmp_domain, edges, _ = _compare_domain(sample_dtypes, bins)
    weighted = wview is not None
    if backend == "numpy":
        device = _host_device()
        stream = 0
        mem = _native.MEM_HOST
    elif backend == "device":
        device = like.device
        stream = 0
        mem = _native.MEM_DEVICE
    else:
        torch = _torch()
        if like.device.type != "cuda":
            raise RuntimeError("torch inputs must live on an MI355X (device='cuda'); got %s" % like.device)
        device = _torch_device_index(like.device)
        stream = torch.cuda.current_stream(like.device).cuda_stream
        mem = _native.MEM_DEVICE
    _native.require_device(device)
    plan = _get_plan(edges, cmp_domain, device)

    # the library zero-initialises (overwrites) the output itself: no fill here, which for torch
    # would be one more kernel launch per call
    out_shape = ((2,) if wview2 is not None else ()) + (nrows,) + plan.bins_shape
    n_out = int(np.prod(out_shape, dtype=np.int64))
    device_partial = backend != "torch" and getattr(_tls, "device_out", False) and wview2 is None
    if n_out > 0 and (device_partial or backend == "device"):
        # a dask block under the device-resident reduction (multigpu.reduce_partials): only host inputs cross PCIe, the
        # partial histogram stays on its GPU.  (A DeviceArray block outside that reduction: downloaded below.)
        out_dtype = np.float64 if weighted else np.int64
        buf = _native.DeviceBuffer(device, n_out * 8)
        out = _native.DevicePartial(buf, out_shape, out_dtype)
        out_ptr = buf.ptr
        if backend == "numpy":
            mem = _native.MEM_HOST_TO_DEVICE
        empty = False
    elif backend != "torch":
        out = np.empty(out_shape, dtype=np.float64 if weighted else np.int64)
        out_ptr = out.ctypes.data
        empty = out.size == 0
    else:
        out = torch.empty(out_shape, dtype=torch.float64 if weighted else torch.int64, device=like.device)
        out_ptr = out.data_ptr()
        empty = out.numel() == 0
    if empty:
        return out

    grouped = any(v[4] for v in views) or (weighted and wview[4]) or (wview2 is not None and wview2[4])
    if block_size in (None, "auto") or grouped:
        row_blocks = [(0, nrows)]
    else:
        if not isinstance(block_size, (int, np.integer)) or isinstance(block_size, bool):
            raise AssertionError("block_size must be None, 'auto' or an int")  # core.py:116
        if block_size <= 0:
            raise ZeroDivisionError("block_size must be positive")  # core.py:117 raises the same
        row_blocks = [(r, min(r + int(block_size), nrows)) for r in _range(0, nrows, int(block_size))]

    def shifted(view, r0):
        ptr, tag, rs, cs, ir, os_, _keep = view
        return _native.make_view(ptr + r0 * rs * np.dtype(_TAG_NP[tag]).itemsize, tag, rs, cs, ir, os_)

    row_bytes = plan.n_bins * 8
    for r0, r1 in row_blocks:
        if wview2 is not None:
            plan.execute_two_weights(
                [shifted(v, r0) for v in views], shifted(wview, r0), shifted(wview2, r0), r1 - r0, ncols,
                out_ptr + r0 * row_bytes, out_ptr + (nrows + r0) * row_bytes, mem, accumulate=False, stream=stream,
            )
            continue
        plan.execute(
            [shifted(v, r0) for v in views],
            shifted(wview, r0) if weighted else None,
            r1 - r0,
            ncols,
            out_ptr + r0 * row_bytes,
            weighted,
            mem,
            accumulate=False,
            stream=stream,
        )
    if backend == "device" and not device_partial:
        return out.to_numpy()  # DeviceArray in, numpy histogram out (the result is small; the data never moved)
    return out


def _bincount_2d_vectorized(*args, bins=None, weights=None, density=False, right=False, block_size=None):
    """Histogram independently on each row of 2-D arrays — the fused GPU replacement of the
    reference function of the same name (core.py:137-194).

    ``args``: D arrays of equal shape [M, C] (numpy or torch-on-GPU, any strides); ``bins``: D 1-D
    edge arrays; ``weights``: None or [M, C].  Returns [M, nb_0, ..., nb_{D-1}]: int64 counts
    (unweighted) or float64 sums.  ``density`` and ``right`` are accepted and ignored, as in the
    reference (core.py:138).  ``block_size`` (None, "auto" or a positive int) only sets how many
    rows one kernel launch covers: the reference's row blocking (core.py:86-134) bounds numpy
    temporaries that do not exist here, and never changes the result.
    """
    a0 = args[0]
    backend = _backend_of(args)
    for a, b in zip(args, bins):  # core.py:146-151
        assert a.ndim == 2
        assert np.ndim(b) == 1
        assert tuple(a.shape) == tuple(a0.shape)
    if weights is not None:
        assert tuple(weights.shape) == tuple(a0.shape)
    if len(bins) != len(args):
        raise ValueError("one array of bin edges per input array")
    nrows, ncols = (int(s) for s in a0.shape)
    dtypes = [_np_dtype_of(a) for a in args]
    args, weights = _prepare_dtypes(list(args), weights, dtypes, bins, backend)
    views = [_strided_view(a, backend) for a in args]
    wview = _strided_view(weights, backend) if weights is not None else None
    return _execute_views(views, wview, nrows, ncols, dtypes, bins, backend, a0, block_size)


def _prepare_dtypes(args, weights, dtypes, bins, backend):
    """dtype-level preparation shared by every entry: datetime64 inputs are brought to the unit
    they share with their edges (and later viewed as int64), complex weights are rejected like
    numpy's bincount does"""
    if backend in ("numpy", "device"):
        _, _, common = _compare_domain(dtypes, bins)
        if backend == "device" and any(c is not None and a.dtype != c for a, c in zip(args, common)):
            raise TypeError("datetime64 DeviceArrays must already have the unit they share with their bin edges")
        args = [a.astype(c) if c is not None and a.dtype != c else a for a, c in zip(args, common)]
        if weights is not None and weights.dtype.kind == "c":
            raise TypeError("Cannot cast array data from complex to float64 (weights)")  # numpy bincount
    elif weights is not None and weights.dtype.is_complex:
        raise TypeError("complex weights are not supported")
    return args, weights


# ---------------------------------------------------------------------------------------------
# L2: block adapter                                                    (core.py:197-247)
# ---------------------------------------------------------------------------------------------
def _rows_cols(a, axis, do_full_array):
    """[M, C] arrangement of an N-D block: kept axes -> rows, reduced axes (in the order given)
    -> cols (core.py:211-227).  A view whenever the strides allow it — broadcast (stride-0)
    inputs and trailing reduced axes are NOT materialised, unlike the reference's reshape."""
    if _is_torch(a):
        if do_full_array:
            return a.reshape(1, -1)
        moved = a.movedim(tuple(axis), tuple(_range(-len(axis), 0)))
        keep = moved.shape[: moved.ndim - len(axis)]
        m = 1
        for k in keep:
            m *= int(k)
        c = (a.numel() // m) if m else 0
        item = a.element_size()
        if m and c and _nocopy_reshape_strides(tuple(moved.shape), tuple(st * item for st in moved.stride()), (m, c), item) is None:
            moved = _torch_contiguous(moved)  # the reshape needs a copy: ours
        return moved.reshape(m, -1)
    if _is_devarr(a):
        if do_full_array:
            return a.reshape(1, -1)
        moved = a.moveaxis(tuple(axis), tuple(_range(-len(axis), 0)))
        m = int(np.prod(moved.shape[: moved.ndim - len(axis)], dtype=np.int64))
        return moved.reshape(m, (a.size // m) if m else 0)  # a view where the strides allow it, else one device copy
    if do_full_array:
        moved, m = a, 1
    else:
        moved = np.moveaxis(a, axis, tuple(_range(-len(axis), 0)))
        m = int(np.prod(moved.shape[: moved.ndim - len(axis)], dtype=np.int64))
    c = (a.size // m) if m else 0
    # (numpy's in-place `view.shape = ...` copies the data BEFORE it refuses a shape that needs a copy: ask the rule itself)
    if _nocopy_reshape_strides(moved.shape, moved.strides, (m, c), moved.itemsize) is None:
        return moved.reshape(m, c)
    v = moved.view()
    v.shape = (m, c)
    return v


def _elem_strides(a):
    """(shape, strides in elements) of a numpy array or torch tensor; None if not element-aligned"""
    if _is_torch(a):
        return tuple(int(s) for s in a.shape), tuple(int(s) for s in a.stride())
    item = a.dtype.itemsize
    if any(st % item for st in a.strides):
        return None
    return tuple(a.shape), tuple(st // item for st in a.strides)


def _merge_dims(dims):
    """merge consecutive (size, stride) dims that walk memory like one dim; size-1 dims vanish"""
    out = []
    for size, stride in dims:
        if size == 1:
            continue
        if out and (out[-1][1] == stride * size or (out[-1][1] == 0 and stride == 0)):
            out[-1] = (out[-1][0] * size, stride)
        else:
            out.append((size, stride))
    return out


def _collapse(a, axis, do_full_array, reduced_order):
    """Describe the [kept, reduced] arrangement of an N-D block (core.py:211-227) by strides
    instead of making it: (rows, cols, row stride, col stride, rows per group, group stride) or None.

    The reduced axes must walk memory as ONE strided dimension (their order inside a row does not
    matter to a histogram, so ``reduced_order`` — chosen once for all inputs — may permute them);
    the kept axes as one or two (kept axes on both sides of the reduced ones: grouped rows)."""
    ss = _elem_strides(a)
    if ss is None:
        return None
    shape, strides = ss
    if any(st < 0 for st in strides):
        return None
    ndim = len(shape)
    red = list(_range(ndim)) if do_full_array else list(reduced_order)
    kept = [] if do_full_array else [i for i in _range(ndim) if i not in red]
    cols = _merge_dims([(shape[i], strides[i]) for i in red])
    rows = _merge_dims([(shape[i], strides[i]) for i in kept])
    if len(cols) > 1 or len(rows) > 2:
        return None
    m = 1
    for i in kept:
        m *= shape[i]
    c = 1
    for i in red:
        c *= shape[i]
    cs = cols[0][1] if cols else 0
    if len(rows) == 2:
        (_, os_), (inner, rs) = rows
        return m, c, rs, cs, inner, os_
    rs = rows[0][1] if rows else 0
    return m, c, rs, cs, 0, 0


def _reduced_order(a, axis):
    """order of the reduced axes that walks ``a`` with decreasing strides (C-like)"""
    ss = _elem_strides(a)
    if ss is None:
        return list(axis)
    return sorted(axis, key=lambda i: -ss[1][i])


def _view_of(a, desc, backend, both_strided_limit=1 << 16):
    """native view tuple of an array described by _collapse, or None when a copy is the better plan"""
    m, c, rs, cs, ir, os_ = desc
    if backend == "torch":
        if rs > 1 and cs > 1 and m * c >= both_strided_limit:
            return None  # strided both ways: every lane would touch its own cache line
        return a.data_ptr(), _torch_tag(a.dtype), rs, cs, ir, os_, a
    if backend == "device":
        if rs > 1 and cs > 1 and m * c >= both_strided_limit:
            return None
        return a.ptr, _native.dtype_tag(np.dtype(np.int64) if a.dtype.kind in "mM" else a.dtype), rs, cs, ir, os_, a
    if not a.dtype.isnative:
        return None
    if a.dtype.kind in "mM":
        a = a.view(np.int64)
    # rows that overlap in memory (sliding windows: row stride below the row's extent) cannot be staged
    # as a pitched 2-D copy; the reference's reshape copies them, and so does the copying route here
    rows_side_by_side = ir if ir else m  # rows that follow each other at stride rs
    if m > 1 and c > 1 and rs != 0 and cs != 0 and ((cs == 1 and rs < c) or (rs == 1 and cs < rows_side_by_side)):
        return None
    if ir == 0 and not (cs in (0, 1) or rs in (0, 1)):
        ir, os_ = m, 0  # one group: makes the library stage the bytes as they lie, strides intact
    return a.ctypes.data, _native.dtype_tag(a.dtype), rs, cs, ir, os_, a


def _beyond_lds(bins, weighted):
    """the histogram is bigger than the LDS of the kernels that take any dtype / stride (uint32 or float64 counters)"""
    n_bins = 1
    for b in bins:
        n_bins *= max(len(b) - 1, 1)
    return n_bins * (8 if weighted else 4) > 144 * 1024


def _promote_for_big_histograms(arrays, w_array, dtypes, bins, backend="torch"):
    """Inputs whose dtype mixture only the generic kernel family takes (float32 next to float64,
    integers in a joint histogram, integer weights) AND whose histogram is beyond its LDS: that family then has
    memory-side atomics only (2 x 10^8 samples, 256 x 256 bins, float32 x float64: 24 ms).  Converting to float64 on
    the device is exact wherever the comparison runs in float64 anyway (numpy promotes the same way inside
    searchsorted, and bincount casts weights to double) and opens the packed / sliced / partitioned modes of the vector
    kernels (1.1 ms).  Small histograms stay as they are: the conversion pass would cost more than it saves."""
    weighted = w_array is not None
    if not _beyond_lds(bins, weighted):
        return arrays, w_array, dtypes
    torch = _torch() if backend == "torch" else None
    fast_floats = (np.dtype(np.float32), np.dtype(np.float64))
    d = len(arrays)
    vector_ok = all(dt == dtypes[0] for dt in dtypes) and dtypes[0] in fast_floats and d <= 3
    w_ok = (not weighted) or _np_dtype_of(w_array) in fast_floats
    if vector_ok and w_ok:
        return arrays, w_array, dtypes
    try:
        cmp_domain, _, _ = _compare_domain(dtypes, bins)
    except (TypeError, NotImplementedError, AssertionError):
        return arrays, w_array, dtypes  # let the regular path raise what it raises
    if cmp_domain != _native.CMP_F64 or d > 3:
        return arrays, w_array, dtypes  # exact int64 / datetime comparisons stay exact
    if backend in ("numpy", "device"):
        # host inputs: the conversion is a host pass (numpy), still far cheaper than ~2.5 x 10^10 atomics per second;
        # DeviceArrays: one strided-copy kernel with the conversion in it (DeviceArray.astype)
        if not vector_ok:
            arrays = [a if a.dtype == np.float64 else a.astype(np.float64) for a in arrays]
            dtypes = [np.dtype(np.float64)] * d
        if weighted and not w_ok:
            w_array = w_array.astype(np.float64)
        return arrays, w_array, dtypes
    if not vector_ok:
        arrays = [a if a.dtype == torch.float64 else a.to(torch.float64) for a in arrays]
        dtypes = [np.dtype(np.float64)] * d
    if weighted and not w_ok:
        w_array = w_array.to(torch.float64)
    return arrays, w_array, dtypes


def _block_placement(all_arrays, multigpu):
    """where one dask block runs: on the GPU its chunks already live on (DeviceArray chunks), else on the least busy one"""
    resident = next((a for a in all_arrays if _is_devarr(a)), None)
    return multigpu.block_device() if resident is None else multigpu.on_device(resident.device)


def _bincount_spread(*all_arrays, **kwargs):
    """One dask block (core.py:429-437): the block adapter on whichever of the node's GPUs has the fewest
    blocks in flight — dask's threaded scheduler runs many blocks at once, and each one is staged over
    its own GPU's PCIe link and binned there (multigpu.block_device)."""
    from . import multigpu

    with _block_placement(all_arrays, multigpu):
        return _bincount(*all_arrays, **kwargs)


def _bincount_partial(*all_arrays, **kwargs):
    """One dask block under the device-resident reduction: as _bincount_spread, but the result is a
    _native.DevicePartial on the block's GPU (only the inputs cross PCIe)."""
    from . import multigpu

    with _block_placement(all_arrays, multigpu):
        prev = getattr(_tls, "device_out", False)
        _tls.device_out = True
        try:
            return _bincount(*all_arrays, **kwargs)
        finally:
            _tls.device_out = prev


def _bincount(*all_arrays, weights=False, axis=None, bins=None, density=None, block_size=None):
    """Block adapter with the reference's contract AND signature (core.py:197-247): N-D block(s) in,
    array of shape kept-axes (1 for each reduced axis) + bin dims out.  Called directly for
    numpy/torch inputs and once per block by the dask branch (core.py:429-437).

    Where the reference moves the reduced axes last and reshapes (a copy for anything but
    trailing axes), the block is DESCRIBED to the native library as a strided [rows, cols] view —
    broadcast inputs, leading-axis and middle-axis reductions included — and only layouts no
    three strides can express fall back to that copy."""
    return _block_adapter(all_arrays, weights, axis, bins, density, block_size, False)


def _bincount_two_weights(*all_arrays, weights=False, axis=None, bins=None, density=None, block_size=None):
    """`_bincount` for histogram_two_weights (an extension, private): the LAST TWO of `all_arrays` are weight
    arrays, the result carries a leading pair axis."""
    return _block_adapter(all_arrays, weights, axis, bins, density, block_size, True)


def _block_adapter(all_arrays, weights, axis, bins, density, block_size, second_weights):
    backend = _backend_of(all_arrays)
    if backend == "device":
        # a dask block: chunks that live on a GPU next to host chunks (weights from a numpy-backed dask array) — the
        # host ones follow the resident ones to their GPU
        dev = next(a.device for a in all_arrays if _is_devarr(a))
        all_arrays = [a.to(dev) if _is_devarr(a) else DeviceArray.from_numpy(a, dev) for a in all_arrays]
    a0 = all_arrays[0]
    ndim = a0.ndim
    do_full_array = (axis is None) or (set(axis) == set(_range(ndim)))
    if do_full_array:
        kept_axes_shape = (1,) * ndim
    else:
        kept_axes_shape = tuple(int(a0.shape[i]) if i not in axis else 1 for i in _range(ndim))

    arrays = list(all_arrays)
    w2_array = arrays.pop() if second_weights else None  # histogram_two_weights (extension)
    w_array = arrays.pop() if weights else None
    n_inputs = len(arrays)
    dtypes = [_np_dtype_of(a) for a in arrays]
    arrays, w_array = _prepare_dtypes(arrays, w_array, dtypes, bins, backend)
    if second_weights and (w2_array.dtype.is_complex if backend == "torch" else w2_array.dtype.kind == "c"):
        raise TypeError("complex weights are not supported")
    if all(dt.kind in "fiub" for dt in dtypes) and (w_array is None or _np_dtype_of(w_array).kind in "fiub"):
        arrays, w_array, dtypes = _promote_for_big_histograms(arrays, w_array, dtypes, bins, backend)
    w_list = ([w_array] if weights else []) + ([w2_array] if second_weights else [])

    counts = None
    order = _reduced_order(arrays[0], list(_range(ndim)) if do_full_array else axis)
    descs = [_collapse(a, axis, do_full_array, order) for a in arrays + w_list]
    if backend != "numpy" and _beyond_lds(bins, weights) and any(d is not None and d[1] > 1 and d[3] != 1 for d in descs):
        # columns that are not unit-stride (strided or broadcast views, leading-axis reductions) go to kernels that
        # keep their histogram in LDS or, beyond it, in memory-side atomics: for a big histogram the reference's
        # copy into [rows, cols] blocks (core.py:211-229), made on the device, is far cheaper than those atomics
        descs = [None]
    if all(d is not None for d in descs) and len({d[:2] for d in descs}) == 1:
        views = [_view_of(a, d, backend) for a, d in zip(arrays + w_list, descs)]
        if all(v is not None for v in views):
            m, c = descs[0][:2]
            try:
                counts = _execute_views(views[:n_inputs], views[n_inputs] if weights else None, int(m), int(c), dtypes, bins,
                                        backend, a0, block_size, wview2=views[n_inputs + 1] if second_weights else None)
            except NotImplementedError:
                counts = None  # e.g. a host view spanning > 2^32 elements: take the copying route
    if counts is None:
        blocks = [_rows_cols(a, axis, do_full_array) for a in arrays + w_list]
        if second_weights:  # the copying route: two passes
            pair = [_bincount_2d_vectorized(*blocks[:n_inputs], bins=bins, weights=wb, block_size=block_size) for wb in blocks[n_inputs:]]
            counts = _torch().stack(pair) if backend == "torch" else np.stack(pair)
        else:
            weights_block = blocks.pop() if weights else None
            counts = _bincount_2d_vectorized(*blocks, bins=bins, weights=weights_block, density=density, block_size=block_size)
    if second_weights:
        return counts.reshape((2,) + kept_axes_shape + tuple(counts.shape[2:]))
    return counts.reshape(kept_axes_shape + tuple(counts.shape[1:]))


# ---------------------------------------------------------------------------------------------
# L3: public API                                                       (core.py:250-466)
# ---------------------------------------------------------------------------------------------
def _ensure_correctly_formatted_bins(bins, N_expected):
    """One bin specification per input array (the reference helper of the same name, core.py:37-48):
    a single int / estimator name / edge array serves every input; a sequence must have one entry
    per input."""
    if bins is None:
        raise ValueError("bins must be provided")
    shared = isinstance(bins, (int, str, np.ndarray))
    per_input = [bins] * N_expected if shared else bins
    if len(per_input) != N_expected:
        raise ValueError("The number of bin definitions doesn't match the number of args")
    return per_input


def _ensure_correctly_formatted_range(range_, N_expected):
    """One (lower, upper) pair — or None — per input array (core.py:51-70): a single pair serves every
    input; a sequence of pairs must have one per input."""
    if range_ is None:
        return [None] * N_expected
    entries_are_sequences = [isinstance(entry, Iterable) for entry in range_]
    if len(range_) == 2 and not all(entries_are_sequences):
        return [range_] * N_expected  # one (lower, upper) pair for all
    if len(range_) != N_expected:
        raise ValueError("The number of ranges doesn't match the number of args")
    if any(len(pair) != 2 for pair in range_):
        raise ValueError(
            "range should be provided as (lower_range, upper_range). In the "
            "case of multiple args, range should be a list of such tuples"
        )
    return range_


def _devarr_as_torch(a):
    """zero-copy torch view of a DeviceArray (the array keeps owning the memory), or None: no torch in this interpreter, no
    GPU visible to it, or a dtype / stride pattern torch's `__cuda_array_interface__` import does not take"""
    try:
        torch = _torch()
        if not torch.cuda.is_available() or a.dtype.kind not in "fiu" or a.dtype == np.float16:
            return None
        return torch.as_tensor(a, device=torch.device("cuda", _native.physical_device(a.device)))
    except Exception:
        return None


def _device_bin_edges(a, b, r, has_weights):
    """np.histogram_bin_edges (core.py:383-388) for a GPU-resident array without moving it:
    explicit edges are validated by numpy; an integer ``bins`` needs only the data's min/max
    (reduced on the GPU, NaN-propagating like numpy) and its dtype; string estimators need the
    data itself and take the slow path through host memory."""
    proto_dtype = _np_dtype_of(a)
    resident = _is_devarr(a)
    if isinstance(b, str):
        if has_weights:
            raise TypeError("Automated estimation of the number of bins is not supported for weighted data")
        edges = _device_estimator_edges(a, b, r, proto_dtype, resident)
        if edges is None:
            edges = _device_quartile_edges(a, b, r, proto_dtype, resident)
        if edges is None:
            edges = _device_doane_stone_edges(a, b, r, proto_dtype, resident)
        if edges is None and resident:
            # a DeviceArray where torch is importable: the order-statistics search and the "doane" / "stone" selectors run on
            # a zero-copy torch view of the same memory (`__cuda_array_interface__`); without torch (the dask interpreter of
            # this image) the host copy below remains
            t = _devarr_as_torch(a)
            if t is not None:
                edges = _device_quartile_edges(t, b, r, proto_dtype, False)
                if edges is None:
                    edges = _device_doane_stone_edges(t, b, r, proto_dtype, False)
                del t
        if edges is not None:
            return edges
        # what is left — a DeviceArray without torch, float16 data, "stone" of more than 1.6 x 10^7 elements, a bin
        # count that hangs on numpy's own summation order ("scott", "doane") — takes numpy's implementation on a host copy
        return np.histogram_bin_edges(a.to_numpy() if resident else a.detach().cpu().numpy(), bins=b, range=r)
    if np.ndim(b) == 0 and r is None:
        if (a.size if resident else a.numel()) == 0:
            return np.histogram_bin_edges(np.zeros(0, proto_dtype), bins=b, range=None)
        flat = a.reshape(1, -1)
        ptr, tag, rs, cs, _ir, _os, keep = _strided_view(flat, "device" if resident else "torch")
        if resident:
            dev, stream = a.device, 0
        else:
            torch = _torch()
            dev = _torch_device_index(a.device)
            stream = torch.cuda.current_stream(a.device).cuda_stream
        lo, hi = _native.minmax(_native.make_view(ptr, tag, rs, cs), 1, flat.shape[1], _native.MEM_DEVICE, dev, stream)
        return np.histogram_bin_edges(np.array([lo, hi]).astype(proto_dtype), bins=b, range=None)
    return np.histogram_bin_edges(np.zeros(0, proto_dtype), bins=b, range=r)


ESTIMATORS_FROM_MOMENTS = ("sqrt", "sturges", "rice", "scott")
ESTIMATORS_FROM_QUARTILES = ("fd", "auto")  # float32 / float64 torch tensors: _device_quartile_edges


def _range_cut(r, proto_dtype):
    """the (lo, hi) the data is cut to before a bin-width selector sees it, as numpy's `keep` mask compares: an empty range
    is widened by half a unit each way first (_get_outer_edges), and float32 data meets the bounds ROUNDED TO float32 —
    numpy compares a float32 array with a Python float in float32 (NEP 50 weak scalars), so an element equal to
    float32(lo) < lo is kept (data clipped to 0.7 with range=(0.7, 1.0): 8 bins, not 1)"""
    lo, hi = (float(r[0]) - 0.5, float(r[1]) + 0.5) if r[0] == r[1] else (float(r[0]), float(r[1]))
    if proto_dtype == np.float32:
        # ... per bound, and only where numpy's promotion stays in float32: Python scalars (weak) and NumPy scalars no wider
        # than float32; an np.float64 / np.int64 bound makes numpy compare in float64, so it is kept as given (ADVICE r4:
        # range=(np.float64(0.7), 1.0) on data clipped to float32(0.7) keeps 322 of 1000 elements, not all of them)
        def narrow(bound, value):
            strong = isinstance(bound, np.generic)  # (np.float64 is a Python float as well: ask numpy first)
            if strong and np.result_type(np.float32, bound.dtype) != np.float32:
                return value
            with np.errstate(over="ignore"):
                return float(np.float32(value))

        lo, hi = narrow(r[0], lo), narrow(r[1], hi)
    return lo, hi


def _estimator_cut(name, r, proto_dtype):
    """(supported, lo_hi): whether `name` is one of the estimators that need only moments of data of this dtype, and the
    range the data is cut to before the selector sees it (None: all of it).  numpy validates the range (its own errors)."""
    if name not in ESTIMATORS_FROM_MOMENTS or proto_dtype.kind not in "fiu" or proto_dtype == np.float16:
        return False, None
    if r is None:
        return True, None
    if np.ndim(r[0]) or np.ndim(r[1]):
        return False, None
    np.histogram_bin_edges(np.zeros(0, proto_dtype), bins=1, range=r)
    return True, _range_cut(r, proto_dtype)


def _device_moments(a, lo_hi, want_m2):
    """(n, min, max, mean, M2) of the elements of a GPU-resident array inside lo_hi (all of them for None): xhist_moments"""
    resident = _is_devarr(a)
    if (a.size if resident else a.numel()) == 0:
        return 0, np.inf, -np.inf, np.nan, np.nan
    flat = a.reshape(1, -1)
    ptr, tag, rs, cs, _ir, _os, keep = _strided_view(flat, "device" if resident else "torch")
    if resident:
        dev, stream = a.device, 0
    else:
        dev = _torch_device_index(a.device)
        stream = _torch().cuda.current_stream(a.device).cuda_stream
    out = _native.moments(_native.make_view(ptr, tag, rs, cs), 1, flat.shape[1], lo_hi[0] if lo_hi else None,
                          lo_hi[1] if lo_hi else None, want_m2, dev, stream)
    del keep
    return out


def combine_moments(parts):
    """moments of the union of disjoint shards from the shards' own (n, min, max, mean, M2) — Chan et al.'s pairwise update"""
    n, mn, mx, mean, m2 = 0, np.inf, -np.inf, 0.0, 0.0
    for pn, pmn, pmx, pmean, pm2 in parts:
        if pn == 0:
            continue
        if pmn != pmn or pmx != pmx:  # a NaN among the kept elements: numpy rejects the range
            mn = mx = np.nan
        tot = n + pn
        d = pmean - mean
        m2 = m2 + (0.0 if pm2 != pm2 else pm2) + d * d * n * pn / tot
        mean = mean + d * pn / tot
        if mn == mn:
            mn, mx = min(mn, pmn), max(mx, pmx)
        n = tot
    return n, mn, mx, (mean if n else np.nan), (m2 if n else np.nan)


def _edges_from_moments(name, r, proto_dtype, size, moments, iqr=None, width_of=None):
    """np.histogram_bin_edges(a, bins=name, range=r) from the moments of the cut data — numpy's `_get_bin_edges` for a string
    `bins` restated (numpy/lib/_histograms_impl.py): outer edges from `range` or the data's min / max (NaN -> numpy's
    ValueError), width from the selector, n = ceil((last - first) / width), and numpy's own linspace for the edges —
    bit-identical to numpy whenever n is.  n depends on the data only through exact quantities, except for "scott", whose
    standard deviation is summed in another order than np.std does: when (last - first) / width lies within 1e-6 of an
    integer, or the data are (nearly) constant, the answer is None and numpy decides on a host copy."""
    if size == 0:
        return np.histogram_bin_edges(np.zeros(0, proto_dtype), bins=name, range=r)
    n, mn, mx, mean, m2 = moments
    as_scalar = proto_dtype.type
    if r is None:
        # (first, last) = (a.min(), a.max()) as numpy scalars of the data's dtype; non-finite -> numpy's ValueError
        np.histogram_bin_edges(np.array([mn, mx]).astype(proto_dtype), bins=1, range=None)
        outer = (as_scalar(mn), as_scalar(mx))
    else:
        outer = (r[0], r[1])  # as the caller gave them: numpy computes with their types (python floats are "weak")
    first, last = outer
    if first == last:  # numpy/lib/_histograms_impl.py::_get_outer_edges
        first, last = first - 0.5, last + 0.5
    if n == 0:
        n_bins = 1
    else:
        if proto_dtype.kind == "f":
            ptp = as_scalar(mx) - as_scalar(mn)  # numpy's _ptp: in the data's own precision
        else:
            ptp = int(mx) - int(mn)
        if name == "sqrt":
            width = ptp / np.sqrt(n)
        elif name == "sturges":
            width = ptp / (np.log2(n) + 1.0)
        elif name == "rice":
            width = ptp / (2.0 * n ** (1.0 / 3))
        elif width_of is not None:  # "doane" / "stone": the caller's selector, given numpy's _ptp of the data
            width = width_of(ptp)
            if width is None:
                return None
        elif name in ESTIMATORS_FROM_QUARTILES:
            width = 2.0 * iqr * n ** (-1.0 / 3.0)  # _hist_bin_fd
            if name == "auto":  # _hist_bin_auto: the smaller of "fd" and "sturges" — "sturges" alone where the quartiles coincide
                sturges = ptp / (np.log2(n) + 1.0)
                width = min(width, sturges) if width else sturges
        else:
            std = np.sqrt(m2 / n)
            if not std > 1e-5 * max(abs(mx), abs(mn)):
                # (nearly) constant data: np.std's own rounding — the mean of equal float32 values is not always that value —
                # decides between one bin and "too many bins"; only numpy's summation order reproduces numpy there
                return None
            if proto_dtype == np.float32:
                std = np.float32(std)  # np.std of float32 data is a float32
            width = (24.0 * np.pi ** 0.5 / n) ** (1.0 / 3.0) * std
        if width:
            clamped = proto_dtype.kind in "iub" and width < 1
            if clamped:
                width = 1
            if np.result_type(first, last).kind in "iub":
                span = int(last) - int(first)  # numpy's _unsigned_subtract: exact for integers
            else:
                span = last - first
            q = span / width
            if not np.isfinite(q):
                return None
            if name == "scott" and not clamped and abs(q - np.rint(q)) <= 1e-6 * max(1.0, abs(q)):
                return None  # a tie at the ceil: np.std's own summation order decides
            if name == "doane" and not clamped and abs(q - np.rint(q)) <= 1e-5 * max(1.0, abs(q)):
                return None  # (likewise: numpy's mean / std / third moment in the data's own precision and order)
            n_bins = int(np.ceil(q))
        else:
            n_bins = 1
    return np.histogram_bin_edges(np.zeros(0, proto_dtype), bins=n_bins, range=outer)


def _device_order_statistics(flat, ranks, mn, mx, n):
    """The elements of a 1-D float GPU tensor (no NaNs; min `mn`, max `mx`, `n` elements) at the given 0-based positions of
    its sorted order, exactly, without sorting or copying it: the interval that holds a rank is narrowed by histograms
    of 32768 bins — this library's own streaming pass, 1.2 ms per 10^9 float64 — until it holds at most 65536
    elements, which are then fetched (a boolean mask on the device, a few KB to the host) and sorted there.
    {rank: value} or None (more than 8 passes: leave it to numpy)."""
    torch = _torch()
    B, small = 1 << 15, 1 << 16
    is32 = flat.dtype == torch.float32
    is_int = not flat.dtype.is_floating_point

    def bounds32(lo, top):  # lo <= x <= top for float32 x, decided exactly by float32 scalars
        lo32, top32 = np.float32(lo), np.float32(top)
        if float(lo32) < lo:
            lo32 = np.nextafter(lo32, np.float32(np.inf))
        if float(top32) > top:
            top32 = np.nextafter(top32, np.float32(-np.inf))
        return float(lo32), float(top32)

    out = {}
    first_pass = None
    for rank in sorted(set(int(r) for r in ranks)):
        lo, top, k = float(mn), float(mx), rank  # the interval [lo, top] holds the element; k = its position among the interval's
        found = None
        for step in _range(8):
            if lo == top:
                found = lo
                break
            edges = np.unique(np.linspace(lo, top, B + 1))
            if len(edges) < 2:
                found = lo
                break
            if step == 0 and first_pass is not None:
                counts = first_pass
            else:
                counts = histogram(flat, bins=edges)[0].cpu().numpy()
                if step == 0:
                    first_pass = counts
            cum = np.cumsum(counts)
            if int(cum[-1]) <= k:
                return None  # (the interval does not hold what the bookkeeping says: never observed; numpy decides)
            j = int(np.searchsorted(cum, k, side="right"))
            k -= int(cum[j - 1]) if j else 0
            lo = float(edges[j])
            top = float(edges[j + 1]) if j + 1 == len(edges) - 1 else float(np.nextafter(edges[j + 1], -np.inf))  # bins are [e_j, e_j+1), the last one closed
            if int(counts[j]) <= small:
                l, t = bounds32(lo, top) if is32 else ((int(np.ceil(lo)), int(np.floor(top))) if is_int else (lo, top))
                vals = np.sort(flat[(flat >= l) & (flat <= t)].cpu().numpy())
                if len(vals) != int(counts[j]):
                    return None
                found = float(vals[k])
                break
        if found is None:
            return None
        out[rank] = found
    return out


def _device_quartile_edges(a, name, r, proto_dtype, resident):
    """np.histogram_bin_edges(a, bins="fd" | "auto", range=r) for a float or integer GPU tensor without a host copy:
    numpy's selectors (numpy/lib/_histograms_impl.py: _hist_bin_fd, _hist_bin_auto) need the data only through its size,
    min, max and the two quartiles, and np.percentile's default method needs four order statistics for those — found
    exactly by _device_order_statistics — and its own interpolation (numpy/lib/_function_base_impl.py: _lerp), restated here
    with numpy's dtypes.  With a range the selector sees the data cut to it.  None: not that case (other estimator, dtype,
    a DeviceArray)."""
    if name not in ESTIMATORS_FROM_QUARTILES or resident or proto_dtype.kind not in "fiu" or proto_dtype == np.float16:
        return None
    size = a.numel()
    if size == 0:
        return np.histogram_bin_edges(np.zeros(0, proto_dtype), bins=name, range=r)
    lo_hi = None
    if r is not None:  # the selector sees the data cut to the range (numpy validates it: its own errors)
        if np.ndim(r[0]) or np.ndim(r[1]):
            return None
        np.histogram_bin_edges(np.zeros(0, proto_dtype), bins=1, range=r)
        lo_hi = _range_cut(r, proto_dtype)
    n, mn, mx, _, _ = _device_moments(a, lo_hi, False)
    n = int(n)
    if n == 0:
        return _edges_from_moments(name, r, proto_dtype, size, (0, mn, mx, np.nan, np.nan), iqr=0.0)
    if not (np.isfinite(mn) and np.isfinite(mx)):
        if r is None:
            np.histogram_bin_edges(np.array([mn, mx]).astype(proto_dtype), bins=1, range=None)  # numpy's ValueError
        return None
    if proto_dtype.kind in "iu" and max(abs(mn), abs(mx)) >= 2.0 ** 53:
        return None  # (integers the float64 edges of the search cannot tell apart)
    q = np.true_divide([75, 25], 100)
    virtual = (n - 1) * q
    previous = np.floor(virtual).astype(np.intp)
    nxt = previous + 1
    above = virtual >= n - 1
    previous[above] = n - 1
    nxt[above] = n - 1
    stats = _device_order_statistics(a.reshape(-1), list(previous) + list(nxt), mn, mx, n)
    if stats is None:
        return None
    lower = np.array([stats[int(i)] for i in previous]).astype(proto_dtype)
    upper = np.array([stats[int(i)] for i in nxt]).astype(proto_dtype)
    gamma = np.asanyarray(virtual - previous, dtype=virtual.dtype)
    diff = np.subtract(upper, lower)
    lerp = np.asanyarray(np.add(lower, diff * gamma))
    np.subtract(upper, diff * (1 - gamma), out=lerp, where=gamma >= 0.5, casting="unsafe", dtype=type(lerp.dtype))
    iqr = np.subtract(*lerp)
    return _edges_from_moments(name, r, proto_dtype, size, (n, mn, mx, np.nan, np.nan), iqr=iqr)


def _device_doane_stone_edges(a, name, r, proto_dtype, resident):
    """np.histogram_bin_edges(a, bins="doane" | "stone", range=r) for a float or integer GPU tensor without a host copy
    (numpy/lib/_histograms_impl.py: _hist_bin_doane, _hist_bin_stone).
    "doane" needs the skewness: mean, standard deviation and third moment are reduced on the device in float64; numpy sums
    them in the data's own precision and order, so a bin count that hangs on the last digits (or nearly constant data) is
    left to numpy, as for "scott".
    "stone" minimises a loss over 1 ... max(100, sqrt(n)) bin counts, each needing the histogram of that many uniform bins:
    those are this library's own kernels — exact counts, hence numpy's very numbers — for up to 4000 candidates
    (n <= 1.6 x 10^7; numpy itself needs n / 4 seconds there)."""
    if name not in ("doane", "stone") or resident or proto_dtype.kind not in "fiu" or proto_dtype == np.float16:
        return None
    size = a.numel()
    if size == 0:
        return np.histogram_bin_edges(np.zeros(0, proto_dtype), bins=name, range=r)
    lo_hi = None
    if r is not None:  # the selector sees the data cut to the range (numpy validates it: its own errors)
        if np.ndim(r[0]) or np.ndim(r[1]):
            return None
        np.histogram_bin_edges(np.zeros(0, proto_dtype), bins=1, range=r)
        lo_hi = _range_cut(r, proto_dtype)
    n, mn, mx, _, _ = _device_moments(a, lo_hi, False)
    n = int(n)
    if n == 0:
        return _edges_from_moments(name, r, proto_dtype, size, (0, mn, mx, np.nan, np.nan), width_of=lambda ptp: 0.0)
    if not (np.isfinite(mn) and np.isfinite(mx)):
        if r is None:
            np.histogram_bin_edges(np.array([mn, mx]).astype(proto_dtype), bins=1, range=None)  # numpy's ValueError
        return None
    if proto_dtype.kind in "iu" and max(abs(mn), abs(mx)) >= 2.0 ** 53:
        return None
    torch = _torch()
    flat = a.reshape(-1)
    if name == "doane":
        def width_of(ptp):
            if n <= 2:
                return 0.0
            sg1 = np.sqrt(6.0 * (n - 2) / ((n + 1.0) * (n + 3)))
            # (float64 temporaries of the whole array live on the GPU here — several times a float32 input; a device that
            # cannot hold them hands the case to numpy on a host copy instead of failing: ADVICE r3)
            try:
                xd = flat.to(torch.float64)
                if lo_hi is not None:
                    xd = xd[(xd >= lo_hi[0]) & (xd <= lo_hi[1])]
                mean = xd.mean()
                xd = xd - mean  # (a tensor of our own from here on — `flat.to` may have returned the caller's — so the division below is in place)
                sigma = float(torch.sqrt((xd ** 2).mean()))
                if not sigma > 1e-5 * max(abs(mx), abs(mn)):
                    return None if sigma > 0.0 or mx != mn else 0.0  # exactly constant data: numpy's 0.0 as well
                xd /= sigma
                g1 = float((xd ** 3).mean())
            except torch.cuda.OutOfMemoryError:
                return None
            return ptp / (1.0 + np.log2(n) + np.log2(1.0 + np.absolute(g1) / sg1))
    else:
        upper = max(100, int(np.sqrt(n)))
        if upper > 4000:
            return None
        as_scalar = proto_dtype.type
        first, last = (as_scalar(mn), as_scalar(mx)) if r is None else (r[0], r[1])  # (numpy hands the selector the OUTER edges)
        if first == last:
            first, last = first - 0.5, last + 0.5

        def width_of(ptp):
            if n <= 1 or ptp == 0:
                return 0

            def jhat(nbins):
                hh = ptp / nbins
                edges = np.histogram_bin_edges(np.zeros(0, proto_dtype), bins=nbins, range=(first, last))
                p_k = histogram(flat, bins=edges)[0].cpu().numpy() / n
                return (2 - (n + 1) * p_k.dot(p_k)) / hh

            nbins = min(_range(1, upper + 1), key=jhat)
            if nbins == upper:
                warnings.warn("The number of bins estimated may be suboptimal.", RuntimeWarning, stacklevel=3)
            return ptp / nbins
    return _edges_from_moments(name, r, proto_dtype, size, (n, mn, mx, np.nan, np.nan), width_of=width_of)


def _device_estimator_edges(a, name, r, proto_dtype, resident):
    """np.histogram_bin_edges(a, bins=name, range=r) for the estimators that need only n, min, max and the standard deviation
    of the data — "sqrt", "sturges", "rice", "scott" — from ONE fused reduction on the GPU (xhist_moments; "scott": two
    passes) instead of a host copy of the array (core.py:383-388).  None: another estimator / dtype, or a case only numpy's
    own summation order decides (_edges_from_moments) — the caller then lets numpy work on a host copy."""
    ok, lo_hi = _estimator_cut(name, r, proto_dtype)
    if not ok:
        return None
    size = a.size if resident else a.numel()
    moments = _device_moments(a, lo_hi, name == "scott") if size else None
    if moments is not None and proto_dtype.kind in "iu" and proto_dtype.itemsize == 8 and int(moments[0]) > 0:
        # the reduction reads every element as float64: 64-bit integers of magnitude 2^53 and more would come back rounded,
        # where numpy's min / max / ptp are exact integer arithmetic — numpy on a host copy decides those
        if not (abs(moments[1]) < 2.0 ** 53 and abs(moments[2]) < 2.0 ** 53):
            return None
    return _edges_from_moments(name, r, proto_dtype, size, moments)


def _density(counts, bins, n_inputs):
    """core.py:444-462.  bin areas are the outer product of the bin widths; the reference's
    ``np.prod(np.ix_(...))`` for three or more inputs (core.py:454) fails on numpy >= 1.24, the
    outer product is what it was meant to compute (and what np.histogramdd does)."""
    widths = [np.diff(b) for b in bins]
    bin_axes = tuple(_range(-n_inputs, 0))
    if _is_torch(counts):
        # the outer product is formed on the device (same float64 products): only the widths travel,
        # and only once per set of edges (a host-to-device copy per call costs more than the epilogue)
        torch = _torch()
        key = (str(counts.device),) + tuple(np.asarray(w, dtype=np.float64).tobytes() for w in widths)
        with _plans_lock:
            areas_t = _areas.get(key)
            if areas_t is not None:
                _areas.move_to_end(key)
        if areas_t is None:
            for w in widths:
                wt = torch.as_tensor(np.asarray(w, dtype=np.float64), device=counts.device)
                areas_t = wt if areas_t is None else areas_t[..., None] * wt
            with _plans_lock:
                _areas[key] = areas_t
                while len(_areas) > 8:
                    _areas.popitem(last=False)
        sums = counts.sum(dim=bin_axes, keepdim=True)
        return counts / areas_t / sums
    areas = widths[0]
    for w in widths[1:]:
        areas = np.multiply.outer(areas, w)
    if _is_dask(counts):
        sums = counts.sum(axis=bin_axes)
        return counts / areas / sums.reshape(sums.shape + n_inputs * (1,))
    sums = counts.sum(axis=bin_axes)
    with np.errstate(divide="ignore", invalid="ignore"):
        return counts / areas / np.reshape(sums, sums.shape + n_inputs * (1,))


def _reduce_in_two_steps(all_arrays, has_weights, drop_axes, bins, block_size, backend):
    """Reduced axes that are NOT adjacent (``axis=(0, 2)`` of a 3-D array) cannot be walked as one strided
    dimension, and the reference's moveaxis + reshape copies everything (core.py:218-226).  Histograms
    add up, so: histogram over the last block of adjacent reduced axes (no copy), then sum the — much
    smaller — result over the remaining reduced axes.  int64 counts stay exact.  None when the reduced
    axes are one block anyway, or the intermediate would not be small next to the data."""
    axes = sorted(int(a) for a in drop_axes)
    if len(axes) < 2 or axes[-1] - axes[0] + 1 == len(axes):
        return None
    last = [axes[-1]]
    while last[0] - 1 in axes:
        last.insert(0, last[0] - 1)
    rest = [a for a in axes if a not in last]
    shape = tuple(int(n) for n in all_arrays[0].shape)
    n_bins = 1
    for b in bins:
        n_bins *= max(len(b) - 1, 0)
    cols = 1
    for ax in last:
        cols *= shape[ax]
    n_in = len(all_arrays) - (1 if has_weights else 0)
    itemsize = sum(_np_dtype_of(a).itemsize for a in all_arrays[:n_in])
    if n_bins * 8 * 4 > cols * itemsize:
        return None
    part = _bincount(*all_arrays, weights=has_weights, axis=last, bins=bins, density=False, block_size=block_size)
    if backend == "torch":
        return part.sum(dim=rest, keepdim=True)
    return part.sum(axis=tuple(rest), keepdims=True)


def _weights_slab(args_b, w_raw, drop_axes, bins, block_size, backend):
    """Device-resident data with weights that vary only along reduced axes and are small next to the data —
    ``cos(lat)`` of shape (1, lat, 1) or cell areas (1, lat, lon) against (time, lat, lon): the reference
    materialises them at full size (``broadcast_arrays`` + reshape, core.py:366 / 211-229), a second array
    as big as the data.  Here only the SLAB they really span — the run of adjacent reduced axes from the
    first axis they vary along to the innermost reduced axis next to it — is written out ((lat, lon): 4 MB
    against 1.5 GB) and handed to the weighted kernels with stride 0 along every other axis; reduced axes
    outside the slab are summed afterwards (histograms add up).  (365, 720, 1440) float32, cos(lat), over
    (lat, lon): 0.54 -> 0.30 ms; over everything: 0.61 -> 0.31.  None when it does not apply."""
    if backend != "torch" or w_raw is None:
        return None
    shape = tuple(int(n) for n in args_b[0].shape)
    nd = len(shape)
    if w_raw.ndim > nd or nd < 2:
        return None
    wstrides = tuple(w_raw.stride())
    for ax in _range(w_raw.ndim):
        if wstrides[ax] == 0 and w_raw.shape[ax] > 1:
            w_raw = w_raw[tuple(slice(0, 1) if k == ax else slice(None) for k in _range(w_raw.ndim))]
    wshape = (1,) * (nd - w_raw.ndim) + tuple(int(n) for n in w_raw.shape)
    if any(wshape[ax] not in (1, shape[ax]) for ax in _range(nd)):
        return None  # (not broadcastable: the regular path raises)
    drop = sorted(int(ax) for ax in drop_axes)
    vary = [ax for ax in _range(nd) if wshape[ax] > 1]
    if not vary or len(vary) == nd or any(ax not in drop for ax in vary):
        return None  # scalar weights, full-size weights, or weights that differ from row to row
    lo, hi = vary[0], vary[-1]
    if any(ax not in drop for ax in _range(lo, hi + 1)):
        return None
    while hi + 1 in drop:
        hi += 1
    lo_far = lo
    while lo_far - 1 in drop:
        lo_far -= 1
    total = 1
    for n in shape:
        total *= n
    itemsize = sum(_np_dtype_of(a).itemsize for a in args_b)
    n_bins = 1
    for b in bins:
        n_bins *= max(len(b) - 1, 0)
    blk = None
    for first in ([lo_far, lo] if lo_far != lo else [lo]):  # the whole run of adjacent reduced axes if its slab is still small
        slab_elems = 1
        for ax in _range(first, hi + 1):
            slab_elems *= shape[ax]
        left_over = any(ax < first or ax > hi for ax in drop)
        if slab_elems * 8 * 8 > total * itemsize or slab_elems == total:
            continue  # the slab is not small next to the data
        if left_over and n_bins * 8 * 4 > slab_elems * itemsize:
            continue  # (as in _reduce_in_two_steps: the intermediate must stay small)
        blk = list(_range(first, hi + 1))
        break
    if blk is None:
        return None
    lo = blk[0]
    rest = [ax for ax in drop if ax not in blk]
    slab = w_raw.reshape(wshape).expand(tuple(shape[ax] if lo <= ax <= hi else 1 for ax in _range(nd))).contiguous()
    part = _bincount(*args_b, slab.expand(shape), weights=True, axis=blk, bins=bins, density=False, block_size=block_size)
    return part.sum(dim=rest, keepdim=True) if rest else part


def _weights_constant_along_reduced(args_b, w_raw, drop_axes, bins, block_size, backend):
    """Weights that do not vary along some of the reduced axes (``cos(lat)`` of shape (1, lat, 1) under a
    reduction over lat and lon; one weight per time step; ...): the reference materialises them at
    full size (``broadcast_arrays`` + reshape, core.py:366 / 211-229).  Here the samples are COUNTED
    over those axes, unweighted, and the weights are applied to the counts:

        sum_{r0, r1} w[r1] [x in bin]  =  sum_{r1} w[r1] * count_{r0}[r1, bin]

    which reads the data once and moves no weight array at all.  Bins nobody fell into get 0 whatever
    their weight (a NaN weight poisons only bins that received a sample, as in np.bincount).
    Returns the ``_bincount``-shaped result, or None when the rewrite does not apply / does not pay."""
    shape = tuple(int(n) for n in args_b[0].shape)
    nd = len(shape)
    if w_raw is None or w_raw.ndim > nd:
        return None
    # weights that arrive already broadcast (stride-0 views: np.broadcast_to, torch.expand, what the
    # xarray wrapper hands over) are as good as size-1 axes
    wstrides = tuple(w_raw.stride()) if backend == "torch" else tuple(w_raw.strides)
    for ax in _range(w_raw.ndim):
        if wstrides[ax] == 0 and w_raw.shape[ax] > 1:
            w_raw = w_raw[tuple(slice(0, 1) if k == ax else slice(None) for k in _range(w_raw.ndim))]
    wshape = (1,) * (nd - w_raw.ndim) + tuple(int(n) for n in w_raw.shape)
    cand = sorted(int(ax) for ax in drop_axes if wshape[ax] == 1 and shape[ax] > 1)
    if not cand:
        return None
    # count over ONE block of adjacent axes (the one ending at the last candidate): adjacent axes walk
    # memory as a single strided dimension, which the counting kernels take without a copy
    r0 = [cand[-1]]
    while r0[0] - 1 in cand:
        r0.insert(0, r0[0] - 1)
    r1 = [int(ax) for ax in drop_axes if int(ax) not in r0]
    n_bins = 1
    for b in bins:
        n_bins *= max(len(b) - 1, 0)
    cols = 1
    for ax in r0:
        cols *= shape[ax]
    itemsize = sum(_np_dtype_of(a).itemsize for a in args_b)
    if n_bins * 8 * 4 > cols * itemsize:  # the intermediate counts must stay small next to the data they summarise
        return None
    counts = _bincount(*args_b, weights=False, axis=sorted(r0), bins=bins, density=False, block_size=block_size)
    tail = (1,) * len(bins)
    if backend == "torch":
        torch = _torch()
        w = w_raw.reshape(wshape + tail).to(torch.float64)
        contrib = torch.where(counts != 0, counts.to(torch.float64) * w, torch.zeros((), dtype=torch.float64, device=counts.device))
        return contrib.sum(dim=r1, keepdim=True) if r1 else contrib
    w = np.asarray(w_raw).reshape(wshape + tail)
    if w.dtype.kind == "c":
        return None  # let the regular path raise numpy's error
    with np.errstate(invalid="ignore", over="ignore"):
        contrib = np.where(counts != 0, counts.astype(np.float64) * w.astype(np.float64), 0.0)
    return contrib.sum(axis=tuple(r1), keepdims=True) if r1 else contrib


def histogram_two_weights(*args, bins=None, range=None, axis=None, weights=None, block_size="auto"):
    """Two weighted histograms of the same data from ONE pass over it (an extension; the reference
    leaves it as a TODO at xarray.py:106 and runs the path once per weight array).

    ``weights`` is a pair ``(wa, wb)``; everything else is as in :func:`histogram`.  Returns
    ``(ha, hb, bins)`` with ``ha == histogram(*args, weights=wa, ...)[0]`` and likewise ``hb``.  The
    idiom: the mean of ``A`` in the bins of ``x`` is ``ha / hb`` for ``wa = A * w``, ``wb = w``.
    Device-resident float32/float64 inputs with histograms that fit LDS share one read and one
    digitize of the samples; every other case degrades to two passes with identical results."""
    if weights is None or len(weights) != 2:
        raise ValueError("weights must be a pair of arrays")
    wa, wb = weights
    if any(_is_dask(a) for a in list(args) + [wa, wb]):
        ha, bins_out = histogram(*args, bins=bins, range=range, axis=axis, weights=wa, block_size=block_size)
        hb, _ = histogram(*args, bins=bins_out, axis=axis, weights=wb, block_size=block_size)
        return ha, hb, bins_out
    ha, bins_out = _histogram(args, bins, range, axis, wa, False, block_size, wb)
    return ha[0], ha[1], bins_out


# ---------------------------------------------------------------------------------------------
# short cut for the plain device-resident call
# ---------------------------------------------------------------------------------------------
# `histogram(x, bins=edges)` on a contiguous GPU tensor spends 9 us in its kernel (10^6 float64 samples) and
# used to spend 42 us in the layers above it: backend detection, broadcasting, numpy's validation of the edges,
# dtype promotion rules, stride analysis, plan lookup.  None of that can change between two calls with the same
# edges and the same kind of input, so the outcome is cached per signature and the call goes straight to the
# plan.  Everything this path does not recognise falls through to the general code below it.
_FAST = OrderedDict()  # (device, bins signature) -> (plan, validated edge arrays, the plan cache's key of that plan)
_FAST_TAGS = None


def _fast_signature(bins, n_inputs):
    """hashable identity of explicit float bin edges (their bytes: in-place edits are seen), or None"""
    if type(bins) is np.ndarray:
        per_input = (bins,) * n_inputs
    elif type(bins) in (list, tuple) and len(bins) == n_inputs:
        per_input = tuple(bins)
    else:
        return None, None
    sig = []
    for b in per_input:
        if type(b) is not np.ndarray or b.ndim != 1 or b.dtype.kind != "f" or b.dtype.itemsize > 8 or b.size < 2:
            return None, None
        sig.append((b.dtype.str, b.tobytes()))
    return tuple(sig), per_input


def _resident_fast_path(args, bins, range_, axis, weights, density, block_size):
    """(hist, edges) for contiguous float32 / float64 GPU tensors of one shape and dtype, explicit float edges,
    full or trailing-axes reduction, optional same-shape float weights — or None (take the general path)."""
    global _FAST_TAGS
    a0 = args[0]
    if type(a0).__module__ != "torch" or range_ is not None or block_size not in ("auto", None) or not 1 <= len(args) <= 3:
        return None
    torch = _torch()
    if _FAST_TAGS is None:
        _FAST_TAGS = {torch.float64: _native.F64, torch.float32: _native.F32}
    tag = _FAST_TAGS.get(a0.dtype)
    if tag is None or not a0.is_cuda or not a0.is_contiguous() or a0.numel() == 0:
        return None
    shape, device, dtype = a0.shape, a0.device, a0.dtype
    for a in args[1:]:
        if type(a) is not type(a0) or a.dtype != dtype or a.shape != shape or a.device != device or not a.is_contiguous():
            return None
    wtag = None
    if weights is not None:
        if type(weights) is not type(a0) or weights.shape != shape or weights.device != device or not weights.is_contiguous():
            return None
        wtag = _FAST_TAGS.get(weights.dtype)
        if wtag is None:
            return None
    ndim = a0.ndim
    leading = False
    if axis is None:
        kept = 0
    else:
        k = len(axis)
        if k == 0:
            return None
        order = sorted(axis)
        if order == list(_range(ndim - k, ndim)):
            kept = ndim - k
        elif order == list(_range(k)):  # the LEADING axes (dim="time" of (time, lat, lon)): rows are the contiguous direction
            leading = True
            kept = k
        else:
            return None  # (reduced axes in the middle or apart have their own view logic)
    sig, per_input = _fast_signature(bins, len(args))
    if sig is None:
        return None
    dev_index = _torch_device_index(device)
    key = (dev_index, sig)
    with _plans_lock:
        hit = _FAST.get(key)
        if hit is not None:
            _FAST.move_to_end(key)
            cur = _plans.get(hit[2])
            if cur is None:  # the plan cache dropped these edges since: back in, so that both caches hold ONE object
                _plans[hit[2]] = hit[0]
                while len(_plans) > _PLAN_CACHE:
                    _plans.popitem(last=False)
            elif cur is not hit[0]:  # ... and rebuilt them
                hit = (cur, hit[1], hit[2])
                _FAST[key] = hit
    if hit is None:
        # first call with these edges: numpy validates them exactly as in the general path
        edges = [np.histogram_bin_edges(np.zeros(0, np.float64), bins=b, range=None) for b in per_input]
        cmp_domain, conv, _ = _compare_domain([np.dtype(np.float64)] * len(args), edges)
        if cmp_domain != _native.CMP_F64:
            return None
        _native.require_device(dev_index)
        plan_key = (dev_index, cmp_domain) + tuple((e.dtype.str, e.tobytes()) for e in conv)  # (_get_plan's key)
        hit = (_get_plan(conv, cmp_domain, dev_index), edges, plan_key)
        with _plans_lock:
            _FAST[key] = hit
            while len(_FAST) > _PLAN_CACHE:
                _FAST.popitem(last=False)
    plan, edges = hit[0], hit[1]
    weighted = weights is not None
    if leading:
        cols = 1
        for n in shape[:kept]:
            cols *= int(n)
        rows = a0.numel() // cols
        rs, cs, kept_shape = 1, rows, tuple(shape[kept:])
    else:
        rows = 1
        for n in shape[:kept]:
            rows *= int(n)
        cols = a0.numel() // rows
        rs, cs, kept_shape = cols, 1, tuple(shape[:kept])
    out = torch.empty((rows,) + plan.bins_shape, dtype=torch.float64 if weighted else torch.int64, device=device)
    if out.numel():
        views = [_native.make_view(a.data_ptr(), tag, rs, cs) for a in args]
        wview = _native.make_view(weights.data_ptr(), wtag, rs, cs) if weighted else None
        plan.execute(views, wview, rows, cols, out.data_ptr(), weighted, _native.MEM_DEVICE, accumulate=False,
                     stream=torch.cuda.current_stream(device).cuda_stream)
    h = out.reshape(kept_shape + plan.bins_shape)
    if density:
        h = _density(h, edges, len(args))
    return h, list(edges)


def _normalise_axis(axis, ndim):
    """None, or the list of non-negative axis numbers to histogram over (core.py:346-355)"""
    if axis is None:
        return None
    requested = np.atleast_1d(axis)
    assert requested.ndim == 1
    out = []
    for ax in requested:
        ax = int(ax)
        if ax < 0:
            ax += ndim
        assert ax < ndim, "axis must be less than ndim"
        out.append(ax)
    return out


def _counts_one_device(all_arrays, w_raw, n_inputs, has_weights, two, drop_axes, bins, bincount_kwargs, backend):
    """Partial / full histogram of numpy or torch inputs on ONE GPU, reduced axes kept as size-1 dims
    (+ a leading pair axis for two weight arrays): the block adapter, or one of the rewrites around it."""
    block_size = bincount_kwargs["block_size"]
    counts = None
    if not two and has_weights:
        counts = _weights_slab(all_arrays[:n_inputs], w_raw, drop_axes, bins, block_size, backend)
        if counts is None:
            counts = _weights_constant_along_reduced(all_arrays[:n_inputs], w_raw, drop_axes, bins, block_size, backend)
    if counts is None and not two:
        counts = _reduce_in_two_steps(all_arrays, has_weights, drop_axes, bins, block_size, backend)
    if counts is None:
        counts = (_bincount_two_weights if two else _bincount)(*all_arrays, **bincount_kwargs)
    return counts


def _dask_graph(all_arrays, has_weights, drop_axes, bins, bincount_kwargs):
    """The lazy graph of the reference's dask branch (core.py:403-439): ONE block-adapter task per block of
    the (rechunk-aligned) inputs, whose output block keeps every input axis — reduced ones as single-element
    chunks — and appends one unchunked axis per bin dimension; then a sum over the reduced axes adds the
    partial histograms of the blocks that share output rows.  What differs from the reference is where a
    block runs: `_bincount_spread` puts it on the least busy of the node's GPUs."""
    import dask.array as dsa

    ndim = all_arrays[0].ndim
    data_index = tuple(_range(ndim))
    bin_index = tuple(_range(ndim, ndim + len(bins)))
    operands = [item for arr in all_arrays for item in (arr, data_index)]
    from . import multigpu

    on_gpus = multigpu.dask_exchange() == "rccl"
    out_dtype = np.dtype("i8" if not has_weights else all_arrays[-1].dtype)
    partials = dsa.blockwise(
        _bincount_partial if on_gpus else _bincount_spread,
        data_index + bin_index,
        *operands,
        new_axes={ax: len(b) - 1 for ax, b in zip(bin_index, bins)},
        adjust_chunks={ax: (lambda extent: 1) for ax in drop_axes},
        meta=np.array((), out_dtype),
        **bincount_kwargs,
    )
    if not on_gpus:
        return partials.sum(drop_axes)
    # the partial histograms stay on the GPUs that computed them; one task per output chunk adds up the partials of each
    # GPU there and the GPUs' sums with ONE RCCL all-reduce (multigpu.reduce_partials), instead of dask's tree of host sums
    kept_index = tuple(i for i in data_index if i not in drop_axes)
    return dsa.blockwise(
        multigpu.reduce_partials,
        kept_index + bin_index,
        partials,
        data_index + bin_index,
        concatenate=False,
        meta=np.array((), out_dtype),
        dtype=out_dtype,
        drop_axes=tuple(int(a) for a in drop_axes),
        out_dtype=out_dtype.str,
    )


def histogram(*args, bins=None, range=None, axis=None, weights=None, density=False, block_size="auto"):
    """Histogram applied along specified axis / axes, computed on an MI355X.

    Same signature, argument meaning, return value and error behaviour as
    ``xhistogram.core.histogram`` (core.py:250-466):

    args : array_like
        Input data; N arguments give an N-dimensional histogram.  numpy arrays, GPU-resident
        torch tensors or dask arrays; all broadcast against each other.
    bins : int, str, array, or a list with one of those per argument
        Number of bins, a ``numpy.histogram_bin_edges`` estimator name, or the bin edges.  All
        but the last bin are half-open ``[left, right)``; the last bin includes both edges.
        With dask inputs, bins must be arrays of edges (TypeError otherwise).
    range : (float, float) or a list of such pairs, optional
        Outer edges for integer / string ``bins``; default ``(arg.min(), arg.max())``.
    axis : None, int or tuple of ints
        Axes to histogram over; default all (flattened).
    weights : array_like, optional
        Weights broadcastable to the data; each sample contributes its weight.  A NaN weight
        turns its own bin into NaN.
    density : bool
        Normalise so that the integral over the range is 1 (per kept row).
    block_size : int or 'auto'
        Number of rows (non-histogram positions) handled per kernel launch; results never
        depend on it.  (The reference's 'auto' heuristic divides by zero above 10^7 samples per
        row, core.py:114-117; here 'auto' and None mean one launch.)

    Returns ``(hist, bin_edges)``: ``hist`` has the kept axes followed by one axis per argument;
    int64 counts, or float64 when weighted / density.  numpy in -> numpy out, torch in -> torch
    out (same device), dask in -> lazy dask array.
    """
    return _histogram(args, bins, range, axis, weights, density, block_size, None)


def _histogram(args, bins, range, axis, weights, density, block_size, _second_weights):
    """body of :func:`histogram`; `_second_weights` is histogram_two_weights' second weight array (private: the public
    functions keep the reference's exact signatures, core.py:250-258 and :197-199)"""
    n_inputs = len(args)
    axis = _normalise_axis(axis, args[0].ndim if hasattr(args[0], "ndim") else np.ndim(args[0]))  # (np.ndim would compute a dask array)
    if _second_weights is None:
        fast = _resident_fast_path(args, bins, range, axis, weights, density, block_size)
        if fast is not None:
            return fast
    has_weights = weights is not None
    two = _second_weights is not None  # histogram_two_weights: the second weight array rides along
    all_arrays = list(args) + ([weights] if has_weights else []) + ([_second_weights] if two else [])

    # ---- one backend for every input, then broadcast against each other (core.py:366) -----------
    w_raw = None
    if any(_is_dask(a) for a in all_arrays):
        import dask.array as dsa

        backend = "dask"
        all_arrays = list(dsa.broadcast_arrays(*[a if _is_dask(a) else dsa.asarray(np.asarray(a)) for a in all_arrays]))
    elif any(_is_torch(a) for a in all_arrays):
        torch = _torch()
        backend = "torch"
        dev = next((a.device for a in all_arrays if _is_torch(a) and a.device.type == "cuda"), None)
        if dev is None:
            raise RuntimeError("torch inputs must live on an MI355X (device='cuda'); every tensor given is on the CPU")
        all_arrays = [a.to(dev) if _is_torch(a) else torch.as_tensor(np.asarray(a)).to(dev) for a in all_arrays]
        w_raw = all_arrays[n_inputs] if has_weights else None  # as given, before broadcasting
        all_arrays = list(torch.broadcast_tensors(*all_arrays))
    elif any(_is_devarr(a) for a in all_arrays):
        backend = "device"
        dev = next(a.device for a in all_arrays if _is_devarr(a))
        all_arrays = [a.to(dev) if _is_devarr(a) else DeviceArray.from_numpy(np.asarray(a), dev) for a in all_arrays]
        w_raw = all_arrays[n_inputs] if has_weights else None
        shape = np.broadcast_shapes(*[a.shape for a in all_arrays])
        all_arrays = [a.broadcast_to(shape) for a in all_arrays]
    else:
        backend = "numpy"
        all_arrays = [np.asarray(a) for a in all_arrays]
        w_raw = all_arrays[n_inputs] if has_weights else None
        all_arrays = list(np.broadcast_arrays(*all_arrays))
    drop_axes = tuple(axis) if axis is not None else tuple(_range(all_arrays[0].ndim))

    # ---- bin edges (core.py:377-388) ---------------------------------------------------------
    bins = _ensure_correctly_formatted_bins(bins, n_inputs)
    range = _ensure_correctly_formatted_range(range, n_inputs)
    if backend == "dask":
        if not all(isinstance(b, np.ndarray) for b in bins):
            raise TypeError("When using dask arrays, bins must be provided as numpy array(s) of edges")
    elif backend in ("torch", "device"):
        bins = [_device_bin_edges(a, b, r, has_weights) for a, b, r in zip(all_arrays, bins, range)]
    else:
        w_for_edges = all_arrays[n_inputs] if has_weights else None
        bins = [np.histogram_bin_edges(a, bins=b, range=r, weights=w_for_edges) for a, b, r in zip(all_arrays, bins, range)]
    bincount_kwargs = dict(weights=has_weights, axis=axis, bins=bins, density=density, block_size=block_size)

    # ---- counts ------------------------------------------------------------------------------
    if backend == "dask":
        bin_counts = _dask_graph(all_arrays, has_weights, drop_axes, bins, bincount_kwargs)
    else:
        bin_counts = None
        if backend == "numpy":
            # host inputs big enough to be worth it are cut into shards, one per visible GPU: every shard is
            # staged over its own GPU's PCIe link by that GPU's host thread, the partials are added up
            from . import multigpu

            bin_counts = multigpu.host_sharded_counts(all_arrays, w_raw, n_inputs, has_weights, two, drop_axes, bins, bincount_kwargs)
        if bin_counts is None:
            bin_counts = _counts_one_device(all_arrays, w_raw, n_inputs, has_weights, two, drop_axes, bins, bincount_kwargs, backend)
        squeeze_axes = tuple(int(i) + (1 if two else 0) for i in drop_axes)  # (two: a leading pair axis)
        if backend == "torch":
            keep = [s for i, s in enumerate(bin_counts.shape) if i not in squeeze_axes]
            bin_counts = bin_counts.reshape(keep)
        else:
            bin_counts = bin_counts.squeeze(squeeze_axes)

    h = _density(bin_counts, bins, n_inputs) if density else bin_counts
    return h, bins
