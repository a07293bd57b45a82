"""The C-ABI library loads without a GPU and exports every symbol include/xhist_amd.h declares;
argument validation and the no-device failure mode behave as documented (no compute here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "xhist_amd.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xhist_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("xhist_plan_create", "xhist_plan_execute", "xhist_bincount_rows", "xhist_last_error", "xhist_minmax"):
        assert must in names
    assert len(names) >= 12


@pytest.fixture(scope="module")
def lib():
    from xhistogram_amd import _native

    return _native.load()


def test_every_declared_symbol_is_exported(lib):
    from xhistogram_amd import _native

    names = declared_functions()
    for n in names:
        assert hasattr(lib, n), "libxhist_amd.so does not export %s" % n
    assert set(_native.EXPORTS) == set(names), "ctypes shim and header disagree"
    abi = int(re.search(r"#define XHIST_ABI_VERSION (\d+)", open(HEADER).read()).group(1))
    assert lib.xhist_abi_version() == abi == _native.ABI_VERSION


def test_dtype_tags_match_header():
    from xhistogram_amd import _native

    text = open(HEADER).read()
    for name, val in re.findall(r"XHIST_(F64|F32|F16|I64|I32|I16|I8|U64|U32|U16|U8|BOOL)\s*=\s*(\d+)", text):
        assert getattr(_native, name) == int(val)
    for name, val in re.findall(r"XHIST_(ERR_[A-Z_]+)\s*=\s*(-\d+)", text):
        assert getattr(_native, name) == int(val)


def test_argument_validation_before_any_device_work(lib):
    from xhistogram_amd import _native

    e = np.array([0.0, 2.0, 1.0])
    ptrs = (C.c_void_p * 1)(e.ctypes.data)
    lens = (C.c_int64 * 1)(3)
    h = C.c_void_p(0)
    assert lib.xhist_plan_create(0, 1, ptrs, lens, 0, C.byref(h)) == _native.ERR_EDGES
    assert b"monotonically" in lib.xhist_last_error()
    assert lib.xhist_plan_create(0, 0, ptrs, lens, 0, C.byref(h)) == _native.ERR_INVALID
    assert lib.xhist_plan_create(0, 9, ptrs, lens, 0, C.byref(h)) == _native.ERR_INVALID
    assert lib.xhist_plan_create(0, 1, ptrs, lens, 7, C.byref(h)) == _native.ERR_INVALID
    nan = np.array([0.0, np.nan, 1.0])
    ptrs = (C.c_void_p * 1)(nan.ctypes.data)
    assert lib.xhist_plan_create(0, 1, ptrs, lens, 0, C.byref(h)) == _native.ERR_EDGES
    big = (C.c_int64 * 1)((1 << 30) + 1)  # refused before any edge is read
    assert lib.xhist_plan_create(0, 1, ptrs, big, 0, C.byref(h)) == _native.ERR_UNSUPPORTED
    assert lib.xhist_plan_destroy(None) == 0
    with pytest.raises(ValueError):
        _native.Plan([np.array([3.0, 2.0, 1.0])])


def test_no_gpu_means_loud_failure_not_fallback(lib):
    """in the GPU-less build container every compute entry point must refuse"""
    from xhistogram_amd import _native, core

    if _native.device_count() > 0:
        pytest.skip("a GPU is visible here")
    e = np.linspace(0, 1, 5)
    ptrs = (C.c_void_p * 1)(e.ctypes.data)
    lens = (C.c_int64 * 1)(5)
    h = C.c_void_p(0)
    assert lib.xhist_plan_create(0, 1, ptrs, lens, 0, C.byref(h)) == _native.ERR_NO_DEVICE
    assert b"no CPU path" in lib.xhist_last_error()
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        core.histogram(np.zeros(10), bins=e)
    with pytest.raises(RuntimeError):
        _native.Plan([e])


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "xhistogram_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "oracle_np" not in src and "liboracle" not in src, f
