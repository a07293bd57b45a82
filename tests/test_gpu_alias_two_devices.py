"""VERDICT r2 "next" #2b: the N > 1 machinery of the in-call multi-GPU path (device threads, per-device plan cache, staging
streams, device-resident partials, reduce_partials) with REAL kernels on a one-GPU box, through a test-only logical ->
physical device alias of the native library (XHIST_AMD_DEVICE_ALIAS=0,0; parsed at library load, hence the subprocess)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_logical_devices_on_one_gpu():
    env = dict(os.environ, XHIST_AMD_DEVICE_ALIAS="0,0")
    for k in ("LOCAL_RANK", "XHIST_AMD_DEVICE", "XHIST_AMD_DEVICES"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "alias_two_devices_script.py")], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
    out = r.stdout
    for name in ("two plan-cache keys", "device-resident shards", "concurrent partitioned-mode calls", "host shards over two logical GPUs",
                 "dask-style blocks on two logical GPUs", "pickle / deepcopy"):
        assert "ok " + name in out, out
    assert out.strip().endswith("ALL OK")


def test_alias_is_ignored_when_malformed_or_out_of_range():
    """CPU: the Python half parses the variable like the native half — anything that does not name visible GPUs is ignored"""
    from xhistogram_amd import _native

    if _native.device_count() != 0:
        pytest.skip("needs a box without GPUs (the alias must match the device count)")
    os.environ["XHIST_AMD_DEVICE_ALIAS"] = "0,0"
    try:
        assert _native.physical_device(1) == 1  # no GPU: no alias
        os.environ["XHIST_AMD_DEVICE_ALIAS"] = "zero"
        assert _native.physical_device(0) == 0
    finally:
        del os.environ["XHIST_AMD_DEVICE_ALIAS"]
