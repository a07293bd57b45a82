#!/usr/bin/env python
"""C1 (10^6 f64 samples, 100 bins): kernel time against workgroup count / size (plan overrides)."""
import json, sys
import numpy as np, torch
sys.path.insert(0, ".")
from xhistogram_amd import _native, core
edges = np.linspace(-4, 4, 101)
plan = core._get_plan([edges], _native.CMP_F64, 0)
stream = torch.cuda.current_stream().cuda_stream
for n in (1_000_000, 4_000_000):
    x = torch.randn(n, dtype=torch.float64, device="cuda")
    out = torch.zeros(100, dtype=torch.int64, device="cuda")
    xv = [_native.make_view(x.data_ptr(), _native.F64, n, 1)]
    for block in (0, 256, 512, 1024):
        for grid in (0, 16, 32, 64, 128, 256, 512):
            plan.set_param("block_threads", block); plan.set_param("grid_blocks", grid)
            run = plan.bind(xv, None, 1, n, out.data_ptr(), False, _native.MEM_DEVICE, stream=stream)
            for _ in range(20): run()
            plan.set_param("profile", 64); plan.set_param("profile_stride", 4)
            for _ in range(256): run()
            torch.cuda.synchronize()
            ms = plan.profile_read(); plan.set_param("profile", 0); plan.set_param("profile_stride", 1)
            t0 = __import__("time").perf_counter()
            for _ in range(500): run()
            torch.cuda.synchronize()
            wall = (__import__("time").perf_counter() - t0) / 500
            print(json.dumps({"n": n, "block": block, "grid": grid, "kernel_us": round(float(np.median(ms)) * 1e3, 2), "step_us": round(wall * 1e6, 2), "desc": plan.describe()[18:70]}), flush=True)
plan.set_param("block_threads", 0); plan.set_param("grid_blocks", 0)
