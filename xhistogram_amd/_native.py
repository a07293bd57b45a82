"""ctypes shim over libxhist_amd.so (C ABI: include/xhist_amd.h).

This is the only place where Python meets the native library.  There is no fallback: if the
shared object is missing, ``load()`` raises; if no MI355X is visible, every compute entry point
returns XHIST_ERR_NO_DEVICE and the shim raises ``RuntimeError``.  Nothing here imports torch,
dask or the test oracle.
"""

from __future__ import annotations

import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# $XHIST_AMD_LIB points development builds (A/B kernel variants) at another shared object
LIB_PATH = os.environ.get("XHIST_AMD_LIB") or os.path.join(_HERE, "libxhist_amd.so")

ABI_VERSION = 9
MAX_DIMS = 8

# status codes (xhist_status)
OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_HIP, ERR_NOMEM, ERR_EDGES, ERR_COMM = 0, -1, -2, -3, -4, -5, -6, -7

# dtype tags (xhist_dtype)
F64, F32, F16, I64, I32, I16, I8, U64, U32, U16, U8, BOOL = range(12)
CMP_F64, CMP_I64 = 0, 1
CMP_PER_DIM = 0x100  # | mask: bit d set <=> input d compares in int64 (XHIST_CMP_PER_DIM)
CMP_UNSIGNED = 0x200  # | onto CMP_I64 / CMP_PER_DIM: the int64-domain inputs are unsigned 64-bit (edges passed as uint64)
MEM_HOST, MEM_DEVICE, MEM_HOST_TO_DEVICE = 0, 1, 2
COMM_ID_BYTES = 128
REDUCE_SUM, REDUCE_MIN, REDUCE_MAX = 0, 1, 2

_NP_TAG = {
    np.dtype(np.float64): F64, np.dtype(np.float32): F32, np.dtype(np.float16): F16,
    np.dtype(np.int64): I64, np.dtype(np.int32): I32, np.dtype(np.int16): I16, np.dtype(np.int8): I8,
    np.dtype(np.uint64): U64, np.dtype(np.uint32): U32, np.dtype(np.uint16): U16, np.dtype(np.uint8): U8,
    np.dtype(np.bool_): BOOL,
}


def dtype_tag(dt):
    """xhist_dtype tag of a numpy dtype; TypeError for what numpy's bincount/searchsorted path
    would reject too (complex) or what this build does not carry (longdouble, object)."""
    dt = np.dtype(dt)
    try:
        return _NP_TAG[dt]
    except KeyError:
        raise TypeError("dtype %s is not supported by the MI355X histogram path" % dt) from None


class XhistArray(C.Structure):
    _fields_ = [
        ("data", C.c_void_p),
        ("dtype", C.c_int32),
        ("reserved", C.c_int32),
        ("row_stride", C.c_int64),
        ("col_stride", C.c_int64),
        ("inner_rows", C.c_int64),
        ("outer_stride", C.c_int64),
    ]


_lib = None
_lock = threading.Lock()

EXPORTS = (
    "xhist_abi_version", "xhist_last_error", "xhist_device_count", "xhist_device_info",
    "xhist_plan_create", "xhist_plan_destroy", "xhist_plan_execute", "xhist_plan_execute_two_weights", "xhist_bincount_rows",
    "xhist_minmax", "xhist_moments", "xhist_plan_set_param", "xhist_plan_describe", "xhist_plan_profile_read",
    "xhist_comm_unique_id", "xhist_comm_create", "xhist_comm_info", "xhist_comm_allreduce", "xhist_comm_allgather",
    "xhist_comm_wait", "xhist_comm_destroy", "xhist_buffer_alloc", "xhist_buffer_free", "xhist_buffer_copy", "xhist_buffer_add", "xhist_buffer_copy_nd",
    "xhist_pointer_device", "xhist_scratch_stats", "xhist_debug_hold_cus", "xhist_shutdown",
)


def _preload_torch_hip_runtime():
    """PyTorch-ROCm wheels carry their own copy of the HIP runtime (torch/lib/libamdhip64.so).  A process that loads this
    library first and imports torch afterwards ends up with two runtimes, and the second one finds no GPU
    (`torch.cuda.is_available()` turns False; measured on the MI355X box).  With torch's copy loaded first — the order every
    torch-first program has anyway — both see the GPU and share streams.  So: when torch is installed but not imported yet, its
    runtime is loaded before ours.  No torch (the dask interpreter): nothing to do."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    path = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    """Load libxhist_amd.so once; raise (never fall back) if it is absent or of another ABI."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libxhist_amd.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or xhistogram_amd/csrc/build.sh. There is no CPU fallback." % LIB_PATH
            )
        _preload_torch_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        lib.xhist_abi_version.restype = C.c_int
        lib.xhist_last_error.restype = C.c_char_p
        lib.xhist_device_count.argtypes = [C.POINTER(C.c_int)]
        lib.xhist_device_info.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
        lib.xhist_plan_create.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_void_p)]
        lib.xhist_plan_destroy.argtypes = [C.c_void_p]
        lib.xhist_plan_execute.argtypes = [
            C.c_void_p, C.POINTER(XhistArray), C.POINTER(XhistArray), C.c_int64, C.c_int64, C.c_void_p, C.c_int, C.c_int,
            C.c_int, C.c_void_p,
        ]
        lib.xhist_plan_execute_two_weights.argtypes = [
            C.c_void_p, C.POINTER(XhistArray), C.POINTER(XhistArray), C.POINTER(XhistArray), C.c_int64, C.c_int64, C.c_void_p, C.c_void_p,
            C.c_int, C.c_int, C.c_void_p,
        ]
        lib.xhist_bincount_rows.argtypes = [
            C.c_int, C.c_int, C.POINTER(XhistArray), C.POINTER(XhistArray), C.c_int64, C.c_int64, C.POINTER(C.c_void_p),
            C.POINTER(C.c_int64), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p,
        ]
        lib.xhist_minmax.argtypes = [C.c_int, C.POINTER(XhistArray), C.c_int64, C.c_int64, C.POINTER(C.c_double), C.c_int, C.c_void_p]
        lib.xhist_moments.argtypes = [C.c_int, C.POINTER(XhistArray), C.c_int64, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_int,
                                      C.POINTER(C.c_double), C.c_int, C.c_void_p]
        lib.xhist_scratch_stats.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.c_int]
        lib.xhist_debug_hold_cus.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p]
        lib.xhist_plan_set_param.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
        lib.xhist_plan_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        lib.xhist_plan_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int)]
        lib.xhist_comm_unique_id.argtypes = [C.c_void_p, C.c_size_t]
        lib.xhist_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        lib.xhist_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        lib.xhist_comm_allreduce.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]
        lib.xhist_comm_allgather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        lib.xhist_comm_wait.argtypes = [C.c_void_p, C.c_void_p]
        lib.xhist_comm_destroy.argtypes = [C.c_void_p]
        lib.xhist_buffer_alloc.argtypes = [C.c_int, C.c_size_t, C.POINTER(C.c_void_p)]
        lib.xhist_buffer_free.argtypes = [C.c_int, C.c_void_p]
        lib.xhist_buffer_copy.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
        lib.xhist_buffer_add.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
        lib.xhist_buffer_copy_nd.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_void_p, C.c_int, C.POINTER(C.c_int64),
                                             C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_void_p]
        lib.xhist_pointer_device.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        for name in EXPORTS:
            getattr(lib, name)  # AttributeError here = header and library disagree
        if lib.xhist_abi_version() != ABI_VERSION:
            raise RuntimeError("libxhist_amd.so ABI %d != expected %d" % (lib.xhist_abi_version(), ABI_VERSION))
        _lib = lib
    return _lib


def _raise(rc):
    msg = (load().xhist_last_error() or b"").decode("utf-8", "replace")
    if rc == ERR_EDGES or rc == ERR_INVALID:
        raise ValueError(msg)
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ERR_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError("xhist_amd: %s (status %d)" % (msg, rc))


def check(rc):
    if rc != OK:
        _raise(rc)


def device_count():
    n = C.c_int(0)
    check(load().xhist_device_count(C.byref(n)))
    return n.value


def physical_device(device):
    """HIP device behind a LOGICAL device index of this library.  The identity, unless $XHIST_AMD_DEVICE_ALIAS ("0,0": two
    logical devices on GPU 0) is set — a test facility that lets the per-GPU threads, plan caches and buffers of the
    multi-GPU code run with real kernels on a one-GPU box (the native library parses the same variable)."""
    env = os.environ.get("XHIST_AMD_DEVICE_ALIAS", "").strip()
    if not env:
        return int(device)
    try:
        alias = [int(t) for t in env.split(",")]
    except ValueError:
        return int(device)
    return alias[device] if 0 <= device < len(alias) and len(alias) == device_count() else int(device)


def device_info(device=0):
    name = C.create_string_buffer(256)
    cus = C.c_int(0)
    mem = C.c_size_t(0)
    check(load().xhist_device_info(device, name, 256, C.byref(cus), C.byref(mem)))
    return {"name": name.value.decode(), "compute_units": cus.value, "total_mem": mem.value}


def require_device(device=0):
    n = device_count()
    if device >= n:
        raise RuntimeError(
            "xhistogram_amd needs an AMD MI355X (gfx950) visible to HIP: device %d requested, %d visible. "
            "There is no CPU fallback." % (device, n)
        )


def make_view(ptr, tag, row_stride, col_stride, inner_rows=0, outer_stride=0):
    """xhist_array: element (r, c) at ptr[row_offset(r) + c * col_stride], strides in elements;
    inner_rows > 0 groups the rows (see include/xhist_amd.h)"""
    return XhistArray(C.c_void_p(ptr), tag, 0, int(row_stride), int(col_stride), int(inner_rows), int(outer_stride))


class Plan:
    """Device-resident edge tables for one set of bin edges (xhist_plan)."""

    def __init__(self, edges, cmp_domain=CMP_F64, device=0):
        lib = load()
        self.device = int(device)
        self.cmp = int(cmp_domain)
        base = self.cmp & ~CMP_UNSIGNED
        int_t = np.uint64 if self.cmp & CMP_UNSIGNED else np.int64

        def want(d):
            if (base & ~0xff) == CMP_PER_DIM:
                return int_t if (base >> d) & 1 else np.float64
            return int_t if base == CMP_I64 else np.float64
        self._edges = [np.ascontiguousarray(e, dtype=want(d)) for d, e in enumerate(edges)]
        for e in self._edges:
            if e.ndim != 1:
                raise ValueError("bin edges must be 1-D")
        d = len(self._edges)
        if not 1 <= d <= MAX_DIMS:
            raise NotImplementedError("this build histograms 1..%d input arrays at once, got %d" % (MAX_DIMS, d))
        ptrs = (C.c_void_p * d)(*[e.ctypes.data for e in self._edges])
        lens = (C.c_int64 * d)(*[e.shape[0] for e in self._edges])
        handle = C.c_void_p(0)
        check(lib.xhist_plan_create(self.device, d, ptrs, lens, self.cmp, C.byref(handle)))
        self._h = handle
        self.n_dims = d
        self.bins_shape = tuple(max(e.shape[0] - 1, 0) for e in self._edges)
        self.n_bins = int(np.prod(self.bins_shape, dtype=np.int64))

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            _lib.xhist_plan_destroy(h)

    def __del__(self):
        try:  # module globals may already be gone at interpreter shutdown
            self.close()
        except Exception:
            pass

    def set_param(self, key, value):
        check(load().xhist_plan_set_param(self._h, key.encode(), int(value)))

    def describe(self):
        buf = C.create_string_buffer(512)
        check(load().xhist_plan_describe(self._h, buf, 512))
        return buf.value.decode()

    def profile_read(self, cap=4096):
        """durations (ms) of the histogram kernel of the executes recorded since profiling was
        switched on with set_param("profile", R) / since the last read"""
        buf = (C.c_float * cap)()
        n = C.c_int(0)
        check(load().xhist_plan_profile_read(self._h, buf, cap, C.byref(n)))
        return [buf[i] for i in range(n.value)]

    def execute(self, sample_views, weight_view, n_rows, n_cols, out_ptr, weighted, mem_kind, accumulate=False, stream=0):
        d = self.n_dims
        if len(sample_views) != d:
            raise ValueError("plan was built for %d inputs, got %d" % (d, len(sample_views)))
        arr = (XhistArray * d)(*sample_views)
        w = C.byref(weight_view) if weight_view is not None else None
        check(
            load().xhist_plan_execute(
                self._h, arr, w, int(n_rows), int(n_cols), C.c_void_p(out_ptr), F64 if weighted else I64, int(mem_kind),
                1 if accumulate else 0, C.c_void_p(stream or 0),
            )
        )


    def bind(self, sample_views, weight_view, n_rows, n_cols, out_ptr, weighted, mem_kind, accumulate=False, stream=0):
        """A prepared execute: the ctypes arguments are built once and the returned callable only makes the
        native call — for loops over the same buffers (one call costs ~3 us of Python instead of ~8)."""
        d = self.n_dims
        if len(sample_views) != d:
            raise ValueError("plan was built for %d inputs, got %d" % (d, len(sample_views)))
        fn = load().xhist_plan_execute
        arr = (XhistArray * d)(*sample_views)
        w = C.byref(weight_view) if weight_view is not None else None
        call_args = (self._h, arr, w, C.c_int64(int(n_rows)), C.c_int64(int(n_cols)), C.c_void_p(out_ptr), F64 if weighted else I64,
                     int(mem_kind), 1 if accumulate else 0, C.c_void_p(stream or 0))
        keep = (sample_views, weight_view)  # the structs the pointers refer to stay alive with the callable

        def run(_keep=keep):
            rc = fn(*call_args)
            if rc:
                _raise(rc)

        return run

    def execute_two_weights(self, sample_views, wa_view, wb_view, n_rows, n_cols, out_a_ptr, out_b_ptr, mem_kind,
                            accumulate=False, stream=0):
        """two float64 histograms [n_rows, bins] of the same samples, weighted by wa / wb, in one pass
        when the fused kernel applies (xhist_plan_execute_two_weights)"""
        d = self.n_dims
        if len(sample_views) != d:
            raise ValueError("plan was built for %d inputs, got %d" % (d, len(sample_views)))
        arr = (XhistArray * d)(*sample_views)
        check(
            load().xhist_plan_execute_two_weights(
                self._h, arr, C.byref(wa_view), C.byref(wb_view), int(n_rows), int(n_cols), C.c_void_p(out_a_ptr),
                C.c_void_p(out_b_ptr), int(mem_kind), 1 if accumulate else 0, C.c_void_p(stream or 0),
            )
        )


def minmax(view, n_rows, n_cols, mem_kind, device=0, stream=0):
    out = (C.c_double * 2)()
    check(load().xhist_minmax(int(device), C.byref(view), int(n_rows), int(n_cols), out, int(mem_kind), C.c_void_p(stream or 0)))
    return out[0], out[1]


def moments(view, n_rows, n_cols, lo=None, hi=None, want_m2=False, device=0, stream=0):
    """(count, min, max, mean, M2) of the elements of a device-resident array inside [lo, hi] (all of them when lo is None):
    xhist_moments — the data's contribution to numpy's bin-width estimators, without moving the data"""
    out = (C.c_double * 5)()
    use_range = lo is not None
    check(load().xhist_moments(int(device), C.byref(view), int(n_rows), int(n_cols), 1 if use_range else 0, float(lo) if use_range else 0.0,
                               float(hi) if use_range else 0.0, 1 if want_m2 else 0, out, MEM_DEVICE, C.c_void_p(stream or 0)))
    return int(out[0]), out[1], out[2], out[3], out[4]


def scratch_stats(device=0):
    """the library's scratch cache on a GPU: bytes cached, bytes in use, bytes it may keep, recent peak of bytes in use"""
    out = (C.c_uint64 * 4)()
    check(load().xhist_scratch_stats(int(device), out, 4))
    return {"cached": int(out[0]), "live": int(out[1]), "limit": int(out[2]), "recent_peak": int(out[3])}


def debug_hold_cus(workgroups, lds_bytes, microseconds, stream=None, device=0):
    """test support: `workgroups` idle workgroups with `lds_bytes` of LDS each occupy compute units for `microseconds` on `stream`"""
    check(load().xhist_debug_hold_cus(int(device), int(workgroups), int(lds_bytes), int(microseconds), C.c_void_p(stream or 0)))


def comm_unique_id():
    """128 opaque bytes from RCCL (rank 0 calls this and hands them to every rank out of band)"""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    check(load().xhist_comm_unique_id(buf, COMM_ID_BYTES))
    return buf.raw


class Comm:
    """One RCCL communicator bound to one GPU (xhist_comm): the exchange step of sharded inputs for
    hosts without torch.distributed.  ``Comm(device, rank, world_size, unique_id)`` is collective —
    it returns once every rank has joined, or raises after ``$XHIST_AMD_COMM_TIMEOUT_S`` seconds (default 300; ``$XHIST_AMD_COMM_CREATE_TIMEOUT_S`` for this rendezvous alone) when one never does.  Buffers are device pointers; calls are asynchronous on
    ``stream`` and must be issued in the same order on every rank."""

    def __init__(self, device, rank, world_size, unique_id):
        lib = load()
        unique_id = bytes(unique_id)
        h = C.c_void_p()
        check(lib.xhist_comm_create(int(device), int(rank), int(world_size), unique_id, len(unique_id), C.byref(h)))
        self._h = h
        self.device, self.rank, self.world_size = int(device), int(rank), int(world_size)

    def rccl_version(self):
        v = C.c_int(0)
        check(load().xhist_comm_info(self._h, None, None, None, C.byref(v)))
        return int(v.value)

    def allreduce(self, ptr, count, tag, op=REDUCE_SUM, stream=0):
        check(load().xhist_comm_allreduce(self._h, C.c_void_p(ptr), int(count), int(tag), int(op), C.c_void_p(stream)))

    def allgather(self, send_ptr, recv_ptr, count, tag, stream=0):
        check(load().xhist_comm_allgather(self._h, C.c_void_p(send_ptr), C.c_void_p(recv_ptr), int(count), int(tag), C.c_void_p(stream)))

    def wait(self, stream=0):
        """block until everything enqueued on ``stream`` has completed — the host-side end of an exchange.  Unlike a bare
        stream synchronisation it watches the communicator: a peer that never entered the collective, or died in it, ends
        in a RuntimeError after ``$XHIST_AMD_COMM_TIMEOUT_S`` seconds (default 300) and an aborted communicator, not in a hang.  An aborted communicator fails every later
        call at once: its owner drops it and creates a new one (multigpu.DeviceGroup.exchange does)"""
        check(load().xhist_comm_wait(self._h, C.c_void_p(stream)))

    def close(self):
        if getattr(self, "_h", None):
            h, self._h = self._h, None
            check(load().xhist_comm_destroy(h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBuffer:
    """Device memory owned by the library (xhist_buffer_*): a partial histogram that stays on its GPU between
    the kernel and the exchange, for callers without a device allocator of their own (numpy inputs)."""

    def __init__(self, device, nbytes):
        h = C.c_void_p()
        check(load().xhist_buffer_alloc(int(device), int(nbytes), C.byref(h)))
        self.device, self.nbytes, self.ptr = int(device), int(nbytes), h.value

    def upload(self, host):
        """contiguous numpy array -> the buffer (final on return)"""
        assert host.flags.c_contiguous and host.nbytes <= self.nbytes
        check(load().xhist_buffer_copy(self.device, C.c_void_p(self.ptr), C.c_void_p(host.ctypes.data), host.nbytes, 0, None))

    def download(self, host):
        """the buffer -> a contiguous numpy array (final on return; waits for work queued on the NULL stream)"""
        assert host.flags.c_contiguous and host.nbytes <= self.nbytes
        check(load().xhist_buffer_copy(self.device, C.c_void_p(host.ctypes.data), C.c_void_p(self.ptr), host.nbytes, 1, None))

    def add(self, other, count, tag, stream=0):
        """self[i] += other[i] for `count` int64 / float64 elements, on the device"""
        check(load().xhist_buffer_add(self.device, C.c_void_p(self.ptr), C.c_void_p(other.ptr), int(count), int(tag), C.c_void_p(stream or 0)))

    def synchronize(self):
        """wait for the NULL stream of the buffer's device (exchange calls queued there)"""
        check(load().xhist_buffer_copy(self.device, None, None, 0, 1, None))

    def close(self):
        p, self.ptr = getattr(self, "ptr", None), None
        if p and _lib is not None:
            _lib.xhist_buffer_free(self.device, C.c_void_p(p))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # A device pointer means nothing in another process and must never be owned twice in this one: pickling (dask's
    # multiprocessing / distributed schedulers, spilling) and copy / deepcopy go through HOST memory — the bytes are
    # downloaded here and uploaded into a fresh allocation where the copy is rebuilt.
    def __reduce__(self):
        if not self.ptr:
            raise ValueError("DeviceBuffer was closed: nothing to serialise")
        host = np.empty(self.nbytes, np.uint8)
        if self.nbytes:
            self.download(host)
        return (_rebuild_buffer, (self.device, self.nbytes, host))


def _rebuild_buffer(device, nbytes, host):
    """unpickle a DeviceBuffer: the same GPU index where the receiving process has one, its own GPU otherwise"""
    count = C.c_int(0)
    check(load().xhist_device_count(C.byref(count)))
    if count.value < 1:
        raise RuntimeError("xhist_amd: no MI355X visible to the process that unpickles a DeviceBuffer; this library has no CPU path")
    if device >= count.value:
        from . import core

        device = core.default_device() % count.value
    buf = DeviceBuffer(device, nbytes)
    if nbytes:
        buf.upload(np.ascontiguousarray(host))
    return buf


class DevicePartial:
    """A partial histogram that stays on the GPU that computed it (numpy inputs, XHIST_MEM_HOST_TO_DEVICE): what a dask block
    task hands to the reduction over blocks and GPUs instead of a host array.  Shape bookkeeping only; the data is a
    DeviceBuffer of int64 counts or float64 sums."""

    def __init__(self, buf, shape, dtype):
        self.buf, self.shape, self.dtype = buf, tuple(int(n) for n in shape), np.dtype(dtype)

    @property
    def device(self):
        return self.buf.device

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and not isinstance(shape[0], (int, np.integer)) else shape
        assert int(np.prod(shape, dtype=np.int64)) == self.size, (shape, self.shape)
        self.shape = tuple(int(n) for n in shape)
        return self

    def to_numpy(self):
        out = np.empty(self.shape, self.dtype)
        self.buf.download(out)
        return out


def copy_nd(device, shape, src_ptr, src_tag, src_strides, dst_ptr, dst_tag, dst_strides, stream=0):
    """strided N-D copy between device buffers (byte strides), optionally converting to float64 (xhist_buffer_copy_nd)"""
    nd = len(shape)
    arr = C.c_int64 * max(nd, 1)
    check(load().xhist_buffer_copy_nd(int(device), nd, arr(*[int(n) for n in shape]), C.c_void_p(src_ptr), int(src_tag),
                                      arr(*[int(n) for n in src_strides]), C.c_void_p(dst_ptr), int(dst_tag),
                                      arr(*[int(n) for n in dst_strides]), C.c_void_p(stream or 0)))


def pointer_device(ptr):
    """GPU a device pointer lives on"""
    d = C.c_int(-1)
    check(load().xhist_pointer_device(C.c_void_p(ptr), C.byref(d)))
    return d.value


def shutdown():
    if _lib is not None:
        _lib.xhist_shutdown()
