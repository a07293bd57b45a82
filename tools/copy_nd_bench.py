#!/usr/bin/env python
"""xhist_buffer_copy_nd (DeviceArray.copy / astype): GB/s moved (read + written) for the layouts the block adapter meets."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from xhistogram_amd.devicearray import DeviceArray

def timed(view, convert=False, reps=5):
    out = view.astype(np.float64) if convert else view.copy()
    out.owner.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = view.astype(np.float64) if convert else view.copy()
    out.owner.synchronize()
    dt = (time.perf_counter() - t0) / reps
    moved = view.size * view.itemsize + out.size * out.itemsize
    return round(dt * 1e3, 3), round(moved / dt / 1e9, 1)

rng = np.random.default_rng(0)
a = DeviceArray.from_numpy(rng.standard_normal((8192, 8192)).astype(np.float32))
t = DeviceArray.from_numpy(rng.standard_normal((25_000_000, 4)))
v = DeviceArray.from_numpy(rng.standard_normal((64, 720, 1440)).astype(np.float32))
for name, view, conv in (("contiguous (8192, 8192) f32", a, False), ("transpose (8192, 8192) f32", a.T, False), ("every other column", a[:, ::2], False),
                         ("one column of a (25e6, 4) f64 table", t[:, 1], False), ("(64, 720, 1440) f32, lat moved last", v.transpose(0, 2, 1), False),
                         ("f32 -> f64, contiguous", a, True), ("reversed rows", a[::-1], False)):
    ms, gbps = timed(view, conv)
    print(json.dumps({"case": name, "ms": ms, "GBps_moved": gbps}), flush=True)
