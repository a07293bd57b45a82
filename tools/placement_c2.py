#!/usr/bin/env python
"""Does the distance between the sample array and the weight array change the C2 kernel's rate?  Both are carved from ONE device
allocation at chosen distances; 10 launches each, HIP-event kernel times from the plan's profile ring."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xhistogram_amd import core, _native

n = 1_000_000_000
GiB = 1 << 30
arena = torch.empty(8 * n * 2 + 6 * GiB, dtype=torch.uint8, device="cuda")
base = arena.data_ptr()
edges = [np.linspace(-4, 4, 101)]
plan = core._get_plan(edges, _native.CMP_F64, 0)
out = torch.zeros(100, dtype=torch.float64, device="cuda")
g = torch.Generator(device="cuda").manual_seed(0)
x = arena[: 8 * n].view(torch.float64)
x.normal_(generator=g)
for gap in (0, 4096, 1 << 20, 256 << 20, 512 << 20, 768 << 20, GiB, GiB + (256 << 20), GiB + (512 << 20), 2 * GiB, 3 * GiB, 4 * GiB, 5 * GiB):
    w = arena[8 * n + gap: 8 * n + gap + 8 * n].view(torch.float64)
    w.uniform_(generator=g)
    xv = _native.make_view(x.data_ptr(), _native.F64, n, 1)
    wv = _native.make_view(w.data_ptr(), _native.F64, n, 1)
    plan.set_param("profile", 16)
    for _ in range(12):
        plan.execute([xv], wv, 1, n, out.data_ptr(), True, _native.MEM_DEVICE, accumulate=False, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ts = plan.profile_read()[-10:]
    plan.set_param("profile", 0)
    print(json.dumps({"distance_GiB": round((w.data_ptr() - x.data_ptr()) / GiB, 4), "kernel_ms_mean": round(float(np.mean(ts)), 4), "min": round(float(np.min(ts)), 4),
                      "TBps": round(16e9 / np.mean(ts) / 1e9, 3)}), flush=True)
