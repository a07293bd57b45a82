"""DeviceArray view arithmetic, checked on the CPU: a DeviceArray only carries (pointer, shape, byte strides), so pointing it at
the memory of a numpy array and re-reading that memory through numpy shows whether every view operation addresses the
elements numpy's own operation would."""
import numpy as np
import pytest

from xhistogram_amd.devicearray import DeviceArray


def wrap(a):
    return DeviceArray(a, a.ctypes.data, a.shape, a.strides, a.dtype, 0)


def read(d, base):
    """the elements a DeviceArray addresses, read from the numpy buffer it was pointed at"""
    off = d.ptr - base.ctypes.data
    flat = base.reshape(-1).view(np.uint8)
    assert 0 <= off
    v = np.lib.stride_tricks.as_strided(flat[off:].view(d.dtype) if d.size else np.empty(0, d.dtype), shape=d.shape, strides=d.strides)
    return np.array(v)


@pytest.mark.parametrize("key", [
    (slice(None), 1), (2,), (slice(1, None, 2), slice(None), slice(0, 3)), (Ellipsis, 0), (None, slice(None), None, 1),
    (1, 2, 3), (slice(None, None, -1),), (slice(5, 1, -2), Ellipsis, slice(None, None, 2)), (-1, -2), (slice(0, 0),),
])
def test_basic_indexing_addresses_what_numpy_addresses(key):
    a = np.arange(6 * 5 * 4, dtype=np.float32).reshape(6, 5, 4)
    d = wrap(a)[key]
    want = a[key]
    assert d.shape == np.shape(want)
    if d.size and all(s >= 0 for s in d.strides):
        np.testing.assert_array_equal(read(d, a), want)
    elif d.size:  # negative strides: compare element addresses
        ref = a[key]
        assert d.ptr == ref.__array_interface__["data"][0] and d.strides == ref.strides


def test_fancy_indexing_is_refused():
    d = wrap(np.zeros((3, 3)))
    for key in ([0, 1], np.array([True, False, True]), (slice(None), [1])):
        with pytest.raises(TypeError):
            d[key]


def test_transpose_moveaxis_broadcast_reshape():
    a = np.arange(2 * 3 * 4, dtype=np.int16).reshape(2, 3, 4)
    d = wrap(a)
    np.testing.assert_array_equal(read(d.T, a), a.T)
    np.testing.assert_array_equal(read(d.transpose(1, 0, 2), a), a.transpose(1, 0, 2))
    np.testing.assert_array_equal(read(d.moveaxis(0, -1), a), np.moveaxis(a, 0, -1))
    np.testing.assert_array_equal(read(d.moveaxis((0, 1), (2, 0)), a), np.moveaxis(a, (0, 1), (2, 0)))
    np.testing.assert_array_equal(read(np.moveaxis(d, 1, 0), a), np.moveaxis(a, 1, 0))  # through __array_function__
    b = d[:, :1, :].broadcast_to((2, 5, 4))
    assert b.strides[1] == 0
    np.testing.assert_array_equal(read(b, a), np.broadcast_to(a[:, :1, :], (2, 5, 4)))
    np.testing.assert_array_equal(read(d.reshape(6, 4), a), a.reshape(6, 4))
    np.testing.assert_array_equal(read(d.reshape(-1), a), a.reshape(-1))
    np.testing.assert_array_equal(read(d[:, :, ::2].reshape(6, 2), a), a[:, :, ::2].reshape(6, 2))  # still a view
    assert d.is_contiguous() and not d.T.is_contiguous()
    assert d.view(np.uint16).dtype == np.uint16


def test_cuda_array_interface_round_trip():
    a = np.arange(12, dtype=np.float64).reshape(3, 4)
    d = wrap(a)
    cai = d.__cuda_array_interface__
    assert cai["shape"] == (3, 4) and cai["typestr"] == "<f8" and cai["strides"] is None and cai["data"][0] == a.ctypes.data
    t = d.T
    assert t.__cuda_array_interface__["strides"] == (8, 32)
    back = DeviceArray.from_cuda_array_interface(t, device=0)
    assert (back.ptr, back.shape, back.strides, back.dtype) == (t.ptr, t.shape, t.strides, t.dtype)


def test_backend_detection_and_block_placement():
    from xhistogram_amd import core, multigpu

    d = wrap(np.zeros(4))
    d.device = 3
    assert core._backend_of([np.zeros(2), d]) == "device" and core._backend_of([np.zeros(2)]) == "numpy"
    with core._block_placement([np.zeros(2), d], multigpu):
        assert core._host_device() == 3  # a resident chunk pins its block to its own GPU


def test_nocopy_reshape_rule_is_numpys():
    """DeviceArray.reshape decides view-or-copy itself (numpy's in-place shape assignment copies before it refuses, which the
    dummy-backed views of the module cannot afford): same verdicts and strides as numpy on real arrays"""
    from xhistogram_amd.devicearray import _nocopy_reshape_strides

    rng = np.random.default_rng(0)
    checked = refused = 0
    for _ in range(3000):
        nd = int(rng.integers(1, 5))
        shape = tuple(int(n) for n in rng.integers(1, 7, size=nd))
        a = np.arange(int(np.prod(shape)), dtype=np.float32).reshape(shape)
        v = a
        for ax in range(nd):  # a random view: steps, a transposition, a broadcast axis
            if rng.random() < 0.4:
                v = v[(slice(None),) * ax + (slice(None, None, int(rng.integers(1, 3))),)]
        if rng.random() < 0.4:
            v = v.transpose(rng.permutation(nd))
        if rng.random() < 0.2:
            v = np.broadcast_to(v[None], (2,) + v.shape)
        size = v.size
        facs = [f for f in (1, 2, 3, 4, 5, 6, 8, 9, 10, 12) if size % f == 0]
        new = []
        rest = size
        while rest > 1 and len(new) < 3:
            f = int(rng.choice([f for f in facs if rest % f == 0]))
            new.append(f)
            rest //= f
            if f == 1 and rng.random() < 0.5:
                break
        new.append(rest)
        new = tuple(new)
        got = _nocopy_reshape_strides(v.shape, v.strides, new, v.itemsize)
        w = v.view()
        try:
            w.shape = new  # real memory: numpy may copy and refuse, harmlessly
        except AttributeError:
            assert got is None, (v.shape, v.strides, new, got)
            refused += 1
            continue
        assert got is not None, (v.shape, v.strides, new)
        np.testing.assert_array_equal(np.lib.stride_tricks.as_strided(v, new, got), w)
        checked += 1
    assert checked > 500 and refused > 200
