"""Device-resident N-D arrays without torch: what the blocks of a dask array look like when they already live on the GPUs.

The reference's per-block contract (core.py:429-437) hands `_bincount` whatever the dask array's chunks are.  Host (numpy)
chunks cross PCIe on every call (55 GB/s against 6.7 TB/s of HBM); a dask array that is histogrammed more than once — or that
was produced on the GPUs — should keep its chunks there.  Under dask there is no torch in this stack, so the chunks are
`DeviceArray`s: a pointer into memory of the native library (or of any owner that speaks `__cuda_array_interface__`), a
shape, byte strides and a numpy dtype.  Views (basic indexing, transposes, broadcasts, reshapes where strides allow) cost
nothing; copies — the reference's moveaxis + reshape copy (core.py:218-226), the concatenation of unaligned chunks
(test_chunking.py:104-146), the promotion to float64 — are one strided-copy kernel of the library (`xhist_buffer_copy_nd`).

No arithmetic lives here: a DeviceArray is something to histogram (`core.histogram`, `core._bincount`), move and slice.
"""
import numbers

import numpy as np

from . import _native

__all__ = ["DeviceArray", "to_device_chunks"]


def _fake(shape, strides, dtype):
    """a numpy view with this shape / strides over one dummy element: numpy's own stride arithmetic (indexing, broadcasting,
    no-copy reshapes) applied without touching memory.  Never read."""
    return np.lib.stride_tricks.as_strided(np.empty(1, dtype), shape=shape, strides=strides, writeable=False)


def _offset(view, base):
    return view.__array_interface__["data"][0] - base.__array_interface__["data"][0]


def _c_strides(shape, itemsize):
    strides, step = [], itemsize
    for n in reversed(shape):
        strides.append(step)
        step *= max(int(n), 1)
    return tuple(reversed(strides))


def _nocopy_reshape_strides(shape, strides, newshape, itemsize):
    """Byte strides that show the same elements, in C order, under ``newshape`` — or None when only a copy can (numpy's
    no-copy reshape rule, restated: numpy's own ``view.shape = ...`` cannot be asked, it COPIES first and compares pointers
    afterwards, which on the dummy-backed views of this module reads memory that is not there)."""
    if any(n == 0 for n in newshape) or any(n == 0 for n in shape):
        return _c_strides(newshape, itemsize)
    old = [(int(n), int(st)) for n, st in zip(shape, strides) if n != 1]
    newshape = [int(n) for n in newshape]
    newstrides = [0] * len(newshape)
    oi, oj, ni, nj = 0, 1, 0, 1
    while ni < len(newshape) and oi < len(old):
        np_, op = newshape[ni], old[oi][0]
        while np_ != op:
            if np_ < op:
                np_ *= newshape[nj]
                nj += 1
            else:
                op *= old[oj][0]
                oj += 1
        for k in range(oi, oj - 1):  # the old axes taken together must walk memory as one C-ordered axis
            if old[k][1] != old[k + 1][0] * old[k + 1][1]:
                return None
        newstrides[nj - 1] = old[oj - 1][1]
        for k in range(nj - 1, ni, -1):
            newstrides[k - 1] = newstrides[k] * newshape[k]
        ni, nj, oi, oj = nj, nj + 1, oj, oj + 1
    last = newstrides[ni - 1] if ni >= 1 else itemsize
    for k in range(ni, len(newshape)):  # trailing axes of extent 1
        newstrides[k] = last
    return tuple(newstrides)


def _storage_dtype(dtype):
    """datetime64 / timedelta64 are int64 to the kernels (core.py: the int64 compare domain)"""
    dtype = np.dtype(dtype)
    return np.dtype(np.int64) if dtype.kind in "mM" else dtype


class DeviceArray:
    """N-D array in the memory of one GPU.  ``owner`` keeps the allocation alive (a `_native.DeviceBuffer`, or the foreign
    object whose `__cuda_array_interface__` was wrapped); ``ptr`` is the address of element (0, …, 0); ``strides`` are in bytes,
    as numpy's."""

    __slots__ = ("owner", "ptr", "shape", "strides", "dtype", "device", "__weakref__")
    __array_priority__ = 100.0

    def __init__(self, owner, ptr, shape, strides, dtype, device):
        self.owner = owner
        self.ptr = int(ptr)
        self.shape = tuple(int(n) for n in shape)
        self.strides = tuple(int(s) for s in strides)
        self.dtype = np.dtype(dtype)
        self.device = int(device)

    # pickling / copy.copy / copy.deepcopy: through host memory, into a fresh allocation (a device pointer means nothing in
    # another process, and two owners of one allocation would free it twice).  A view pickles as a contiguous copy of what
    # it shows; foreign owners (torch tensors) are not dragged along.
    def __reduce__(self):
        return (_rebuild, (self.to_numpy(), self.device))

    # ---- construction ---------------------------------------------------------------------------------
    @classmethod
    def empty(cls, shape, dtype, device=None):
        from . import core

        device = core._host_device() if device is None else int(device)
        shape = tuple(int(n) for n in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        dtype = np.dtype(dtype)
        _native.dtype_tag(_storage_dtype(dtype))  # TypeError for dtypes the library does not take
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        buf = _native.DeviceBuffer(device, nbytes)
        return cls(buf, buf.ptr, shape, _c_strides(shape, dtype.itemsize), dtype, device)

    @classmethod
    def from_numpy(cls, a, device=None):
        """upload a host array (final when the call returns)"""
        a = np.asarray(a)
        if not a.dtype.isnative:
            a = a.astype(a.dtype.newbyteorder("="))
        a = np.ascontiguousarray(a)
        out = cls.empty(a.shape, a.dtype, device)
        if a.size:
            out.owner.upload(a.view(_storage_dtype(a.dtype)).reshape(-1))
        return out

    @classmethod
    def from_cuda_array_interface(cls, obj, device=None):
        """wrap (not copy) device memory of another library: anything with `__cuda_array_interface__` — torch tensors on
        ROCm, cupy-rocm arrays.  ``obj`` is kept alive by the result."""
        cai = obj.__cuda_array_interface__
        ptr = int(cai["data"][0])
        shape = tuple(int(n) for n in cai["shape"])
        dtype = np.dtype(cai["typestr"])
        strides = cai.get("strides") or _c_strides(shape, dtype.itemsize)
        if device is None:
            device = _native.pointer_device(ptr) if ptr and int(np.prod(shape, dtype=np.int64)) else 0
        return cls(obj, ptr, shape, strides, dtype, device)

    @property
    def __cuda_array_interface__(self):
        return {
            "shape": self.shape,
            "typestr": self.dtype.str,
            "data": (self.ptr, False),
            "version": 3,
            "strides": None if self.is_contiguous() else self.strides,
        }

    # ---- bookkeeping ------------------------------------------------------------------------------------
    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    @property
    def itemsize(self):
        return self.dtype.itemsize

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    def __len__(self):
        if not self.shape:
            raise TypeError("len() of a 0-d array")
        return self.shape[0]

    def __repr__(self):
        return "DeviceArray(shape=%r, dtype=%s, device=%d)" % (self.shape, self.dtype, self.device)

    def is_contiguous(self):
        return self.size == 0 or all(n == 1 or s == c for n, s, c in zip(self.shape, self.strides, _c_strides(self.shape, self.itemsize)))

    def _like(self, ptr, shape, strides, dtype=None):
        return DeviceArray(self.owner, ptr, shape, strides, self.dtype if dtype is None else dtype, self.device)

    def _fake(self):
        return _fake(self.shape, self.strides, self.dtype)

    # ---- views ------------------------------------------------------------------------------------------
    def __getitem__(self, key):
        """basic indexing only (ints, slices, None, Ellipsis): always a view"""
        items = key if isinstance(key, tuple) else (key,)
        for it in items:
            if not (it is None or it is Ellipsis or isinstance(it, (slice, numbers.Integral))):
                raise TypeError("DeviceArray takes basic indexing only (ints, slices, None, ...); got %r" % (it,))
        base = self._fake()
        # (with an Ellipsis the result is always a view — a full set of ints would otherwise READ the dummy's memory)
        sub = base[tuple(items) if any(it is Ellipsis for it in items) else tuple(items) + (Ellipsis,)]
        return self._like(self.ptr + _offset(sub, base), sub.shape, sub.strides)

    def transpose(self, *axes):
        axes = axes[0] if len(axes) == 1 and not isinstance(axes[0], numbers.Integral) else axes
        if not axes or axes[0] is None:
            axes = tuple(reversed(range(self.ndim)))
        axes = tuple(int(a) % self.ndim if self.ndim else 0 for a in axes)
        assert sorted(axes) == list(range(self.ndim)), "axes must be a permutation"
        return self._like(self.ptr, [self.shape[a] for a in axes], [self.strides[a] for a in axes])

    @property
    def T(self):
        return self.transpose()

    def moveaxis(self, source, destination):
        nd = self.ndim
        src = [int(a) % nd for a in np.atleast_1d(source)]
        dst = [int(a) % nd for a in np.atleast_1d(destination)]
        assert len(src) == len(dst) and len(set(src)) == len(src) and len(set(dst)) == len(dst)
        order = [a for a in range(nd) if a not in src]
        for d, s_ax in sorted(zip(dst, src)):
            order.insert(d, s_ax)
        return self.transpose(*order)

    def broadcast_to(self, shape):
        base = self._fake()
        sub = np.broadcast_to(base, shape)
        return self._like(self.ptr, sub.shape, sub.strides)

    def reshape(self, *shape):
        """a view where the strides allow it, else a contiguous copy (numpy's rule)"""
        shape = shape[0] if len(shape) == 1 and not isinstance(shape[0], numbers.Integral) else shape
        shape = tuple(int(n) for n in shape)
        if -1 in shape:
            known = int(np.prod([n for n in shape if n != -1], dtype=np.int64))
            shape = tuple((self.size // known if known else 0) if n == -1 else n for n in shape)
        if int(np.prod(shape, dtype=np.int64)) != self.size:
            raise ValueError("cannot reshape array of size %d into shape %r" % (self.size, shape))
        strides = _nocopy_reshape_strides(self.shape, self.strides, shape, self.itemsize)
        if strides is None:
            c = self.copy()
            return c._like(c.ptr, shape, _c_strides(shape, self.itemsize))
        return self._like(self.ptr, shape, strides)

    def view(self, dtype):
        dtype = np.dtype(dtype)
        assert dtype.itemsize == self.itemsize, "views change the meaning of the bytes, not their number"
        return self._like(self.ptr, self.shape, self.strides, dtype)

    # ---- copies (one strided-copy kernel each) ----------------------------------------------------------------
    def _copy_into(self, dst, convert=False):
        tag = _native.dtype_tag(_storage_dtype(self.dtype))
        dst_tag = _native.F64 if convert else tag
        _native.copy_nd(self.device, self.shape, self.ptr, tag, self.strides, dst.ptr, dst_tag, dst.strides)

    def copy(self):
        """C-contiguous copy on the same GPU"""
        out = DeviceArray.empty(self.shape, self.dtype, self.device)
        if self.size:
            self._copy_into(out)
        return out

    def contiguous(self):
        return self if self.is_contiguous() else self.copy()

    def astype(self, dtype):
        """float64 is the one conversion the path needs on the device (numpy promotes to it inside searchsorted,
        core.py:170, and bincount casts weights to double)"""
        dtype = np.dtype(dtype)
        if dtype == self.dtype:
            return self
        if dtype != np.dtype(np.float64) or self.dtype.kind not in "fiub":
            raise TypeError("DeviceArray converts real numbers to float64 only (asked: %s -> %s)" % (self.dtype, dtype))
        out = DeviceArray.empty(self.shape, np.float64, self.device)
        if self.size:
            self._copy_into(out, convert=True)
        return out

    def to(self, device):
        """the same array on another GPU (through the host: block placement, not a data path)"""
        return self if int(device) == self.device else DeviceArray.from_numpy(self.to_numpy(), int(device))

    def to_numpy(self):
        src = self.contiguous()
        out = np.empty(self.shape, _storage_dtype(self.dtype))
        if out.size:
            flat = out.reshape(-1)
            _native.check(_native.load().xhist_buffer_copy(src.device, flat.ctypes.data, src.ptr, flat.nbytes, 1, None))
        return out.view(self.dtype) if out.dtype != self.dtype else out

    def __array__(self, dtype=None, copy=None):
        a = self.to_numpy()
        return a if dtype is None else a.astype(dtype, copy=False)

    # ---- the numpy functions dask calls on chunks ------------------------------------------------------------
    def __array_function__(self, func, types, args, kwargs):
        impl = _ARRAY_FUNCTIONS.get(func)
        if impl is None:
            return NotImplemented
        return impl(*args, **kwargs)


def _rebuild(host, device):
    """unpickle: upload to the same GPU index where the receiving process has one, else to its own GPU"""
    try:
        n = _native.device_count()
    except Exception:
        n = 0
    return DeviceArray.from_numpy(host, device if device < n else None)


def _concatenate(arrays, axis=0, out=None, dtype=None, casting=None):
    if out is not None or dtype is not None:
        return NotImplemented
    arrays = list(arrays)
    first = next(a for a in arrays if isinstance(a, DeviceArray))
    arrays = [a if isinstance(a, DeviceArray) else DeviceArray.from_numpy(a, first.device) for a in arrays]
    if any(a.dtype != first.dtype for a in arrays):
        raise TypeError("concatenating DeviceArrays of different dtypes")
    nd = first.ndim
    axis = int(axis) % nd
    for a in arrays:
        if a.ndim != nd or any(a.shape[k] != first.shape[k] for k in range(nd) if k != axis):
            raise ValueError("all the input array dimensions except for the concatenation axis must match exactly")
    shape = list(first.shape)
    shape[axis] = sum(a.shape[axis] for a in arrays)
    out = DeviceArray.empty(shape, first.dtype, first.device)
    at = 0
    for a in arrays:
        a = a.to(first.device)
        n = a.shape[axis]
        if a.size:
            piece = out[tuple(slice(at, at + n) if k == axis else slice(None) for k in range(nd))]
            a._copy_into(piece)
        at += n
    return out


def _moveaxis(a, source, destination):
    return a.moveaxis(source, destination)


_ARRAY_FUNCTIONS = {
    np.concatenate: _concatenate,
    np.moveaxis: _moveaxis,
    np.transpose: lambda a, axes=None: a.transpose(axes),
    np.broadcast_to: lambda a, shape, subok=False: a.broadcast_to(shape),
    np.reshape: lambda a, *shape, **kw: a.reshape(*(shape or (kw.get("shape", kw.get("newshape")),))),
    np.shape: lambda a: a.shape,
    np.ndim: lambda a: a.ndim,
    np.size: lambda a, axis=None: a.size if axis is None else a.shape[axis],
    np.ascontiguousarray: lambda a, dtype=None: a.contiguous(),
}


def to_device_chunks(dask_array, devices=None):
    """The same dask array with every chunk resident on a GPU (chunk number modulo the visible GPUs): one upload per chunk,
    after which `core.histogram` bins each block where it lies.  Lazy like any dask operation — ``.persist()`` the result to
    upload once and histogram many times."""
    from . import multigpu

    devices = list(multigpu.get_devices() if devices is None else devices)
    grid = dask_array.numblocks

    def upload(block, block_info=None):
        loc = block_info[0]["chunk-location"] if block_info else (0,) * len(grid)
        number = int(np.ravel_multi_index(loc, grid)) if grid else 0
        return DeviceArray.from_numpy(block, devices[number % len(devices)])

    return dask_array.map_blocks(upload, dtype=dask_array.dtype, meta=np.empty((0,) * dask_array.ndim, dask_array.dtype))
