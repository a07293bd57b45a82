#!/usr/bin/env python
"""BASELINE C4 the way the reference runs it: a dask array (time, lat, lon) chunked on time, histogram over lat / lon,
50 bins — through xhistogram_amd.core.histogram under an interpreter that has dask (/opt/conda/bin/python3.9 here).  The
blocks are host arrays: what is measured is the PCIe-bound host route (one GPU on the test box), next to numpy on the
same blocks.  python3.9 tools/dask_c4.py [time steps] [chunks]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import dask.array as dsa
from xhistogram_amd import multigpu
from xhistogram_amd.core import histogram

T = int(sys.argv[1]) if len(sys.argv) > 1 else 365
chunks = int(sys.argv[2]) if len(sys.argv) > 2 else 8
x = np.random.default_rng(0).standard_normal((T, 720, 1440), dtype=np.float32)
edges = np.linspace(-4, 4, 51)
da = dsa.from_array(x, chunks=((T + chunks - 1) // chunks, 720, 1440))
h, _ = histogram(da, bins=edges, axis=[1, 2])
h.compute()  # warm: plan, allocator
for sched, workers in (("threads", 8), ("threads", 2), ("synchronous", 1)):
    t0 = time.perf_counter()
    got = h.compute(scheduler=sched, num_workers=workers) if sched == "threads" else h.compute(scheduler=sched)
    dt = time.perf_counter() - t0
    print(json.dumps({"case": "dask C4 on the GPU host route", "shape": list(x.shape), "chunks_on_time": chunks, "scheduler": sched, "workers": workers,
                      "gpus": len(multigpu.get_devices()), "s": round(dt, 4), "GBps": round(x.nbytes / dt / 1e9, 1), "Msamples_s": round(x.size / dt / 1e6, 0)}), flush=True)
# the same graph over chunks that already live on the GPU (DeviceArray chunks, persisted): nothing but the partials moves
from xhistogram_amd.devicearray import to_device_chunks
dres = to_device_chunks(da).persist()
hr, _ = histogram(dres, bins=edges, axis=[1, 2])
assert np.array_equal(hr.compute(), got)
for sched, workers in (("threads", 8), ("threads", 2), ("synchronous", 1)):
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        got_r = hr.compute(scheduler=sched, num_workers=workers) if sched == "threads" else hr.compute(scheduler=sched)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    assert np.array_equal(got_r, got)
    print(json.dumps({"case": "dask C4, chunks resident on the GPU", "shape": list(x.shape), "chunks_on_time": chunks, "scheduler": sched, "workers": workers,
                      "exchange": multigpu.dask_exchange(), "s": round(best, 5), "GBps": round(x.nbytes / best / 1e9, 1), "Msamples_s": round(x.size / best / 1e6, 0)}), flush=True)
t0 = time.perf_counter()
want = np.stack([np.histogram(x[i], bins=edges)[0] for i in range(min(T, 16))])
dt = (time.perf_counter() - t0) * T / min(T, 16)
print(json.dumps({"case": "numpy on one core (extrapolated from 16 time steps)", "s": round(dt, 2), "Msamples_s": round(x.size / dt / 1e6, 1)}))
assert np.array_equal(got[: min(T, 16)], want)
