"""dask branch of core.histogram (the reference's blockwise + sum graph, core.py:403-439).
The default interpreter has no dask; the image's conda python 3.9 does, and the package needs
only numpy + ctypes, so the checks run there in a subprocess (skipped if it is absent)."""
import os
import subprocess

import pytest

PY39 = "/opt/conda/bin/python3.9"
SCRIPT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dask_branch_script.py")


def _have_dask_python():
    if not os.path.exists(PY39):
        return False
    return subprocess.run([PY39, "-c", "import dask.array, numpy"], capture_output=True).returncode == 0


needs = pytest.mark.skipif(not _have_dask_python(), reason="no interpreter with dask in this image")


def _run(mode, *extra, exchange=None, resident=False):
    env = dict(os.environ)
    if exchange:
        env["XHIST_AMD_DASK_EXCHANGE"] = exchange
    if resident:
        env["XHIST_SOAK_RESIDENT"] = "1"
    # conda's python ships an older libstdc++ than libamdhip64 needs: let the system one win
    sys_cxx = "/usr/lib/x86_64-linux-gnu/libstdc++.so.6"
    if os.path.exists(sys_cxx):
        env["LD_PRELOAD"] = (sys_cxx + ":" + env["LD_PRELOAD"]) if env.get("LD_PRELOAD") else sys_cxx
    r = subprocess.run([PY39, "-W", "ignore", SCRIPT, mode] + [str(e) for e in extra], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    return r.stdout


@needs
def test_dask_graph_is_lazy_and_shaped_like_the_reference():
    assert "LAZY-OK" in _run("lazy")


@needs
def test_dask_blocks_are_spread_over_the_gpus_of_the_node():
    assert "SPREAD-OK" in _run("spread")


@needs
@pytest.mark.gpu
def test_dask_blocks_compute_on_gpu():
    assert "COMPUTE-OK" in _run("compute")


@needs
@pytest.mark.gpu
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_dask_graphs_on_gpu_match_numpy(seed):
    assert "SOAK-OK" in _run("soak", seed, 40)


@needs
@pytest.mark.gpu
def test_random_dask_graphs_with_partials_kept_on_the_gpu():
    """the device-resident reduction (default with more than one GPU, forced here): block results stay on the GPU as
    DevicePartial, are added there and come back once per output chunk"""
    assert "SOAK-OK" in _run("soak", 7, 60, exchange="rccl")


@needs
@pytest.mark.gpu
def test_dask_arrays_with_chunks_resident_on_the_gpu():
    """the per-block contract (core.py:429-437) on chunks that already live on the GPU (DeviceArray): persisted chunks are
    binned where they lie, unaligned chunkings are sliced / concatenated on the device"""
    assert "RESIDENT-OK" in _run("resident")


@needs
@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["host", "rccl"])
def test_random_dask_graphs_over_resident_chunks(exchange):
    assert "SOAK-OK" in _run("soak", 11, 40, exchange=exchange, resident=True)
