#!/usr/bin/env python
"""Why does the C4 shard kernel (456 x 1 036 800 f32, ~0.3 ms) vary 281..350 us from launch to launch?
Prints the per-launch durations (HIP events around the kernel) back to back, with a host sync between launches,
and with a different (cold) buffer every launch."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from xhistogram_amd import _native, core
dev = torch.device("cuda", 0)
edges = [np.linspace(-4, 4, 51)]
plan = core._get_plan(edges, _native.CMP_F64, 0)
rows, cols = 456, 720 * 1440
bufs = [torch.empty((rows, cols), dtype=torch.float32, device=dev).normal_() for _ in range(4)]
out = torch.zeros((rows, 50), dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream
def run(mode, n=24):
    plan.set_param("profile", n)
    for i in range(n):
        x = bufs[i % 4] if mode == "rotate 4 buffers (7.6 GB: nothing survives in the 256 MB cache)" else bufs[0]
        plan.execute([_native.make_view(x.data_ptr(), _native.F32, cols, 1)], None, rows, cols, out.data_ptr(), False, _native.MEM_DEVICE, stream=stream)
        if mode.startswith("host sync"):
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    ms = plan.profile_read(); plan.set_param("profile", 0)
    print(json.dumps({"mode": mode, "us": [round(m * 1e3, 1) for m in ms], "mean": round(float(np.mean(ms)) * 1e3, 1), "min": round(min(ms) * 1e3, 1), "max": round(max(ms) * 1e3, 1)}), flush=True)
for _ in range(3):
    run("warm")
for mode in ("back to back, same buffer", "host sync between launches, same buffer", "rotate 4 buffers (7.6 GB: nothing survives in the 256 MB cache)", "back to back, same buffer"):
    run(mode)
