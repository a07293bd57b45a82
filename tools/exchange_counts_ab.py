"""Exchange mode against the classic passes for COUNTS (development tool): 5*10^8 float64 pairs into 1024 x 1024 bins (the whole
histogram is the window: 32 rows of uint32 counters per workgroup), N(0,1) and uniform samples; and 2000 x 2000 bins (a window
of 512 of 2000 rows, picked by the probe).    python tools/exchange_counts_ab.py [samples]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from xhistogram_amd import _native, core

n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
_native.require_device(0)
g = torch.Generator(device="cuda")
g.manual_seed(6)
x = torch.empty((1, n), dtype=torch.float64, device="cuda")
y = torch.empty((1, n), dtype=torch.float64, device="cuda")


def timed(plan, edges, mode, steps=6):
    plan.set_param("exchange", mode)
    for _ in range(3):
        out = core._bincount_2d_vectorized(x, y, bins=edges)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        out = core._bincount_2d_vectorized(x, y, bins=edges)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, out, plan.describe()


for nb in (1024, 2000):
    edges = [np.linspace(-4.0, 4.0, nb + 1)] * 2
    plan = core._get_plan(edges, _native.CMP_F64, 0)
    plan.set_param("partition", 1)
    for dist in ("normal", "uniform"):
        if dist == "normal":
            x.normal_(generator=g), y.normal_(generator=g)
        else:
            x.uniform_(-4, 4, generator=g), y.uniform_(-4, 4, generator=g)
        t_cl, a, _ = timed(plan, edges, -1)
        t_auto, b, desc = timed(plan, edges, 0)
        print(json.dumps({"bins": "%dx%d" % (nb, nb), "samples": dist, "classic_ms": round(t_cl, 4), "default_ms": round(t_auto, 4), "identical": bool(torch.equal(a, b)),
                          "frac_of_8TBs_at_16B_default": round(n * 16 / (t_auto * 1e-3) / 8e12, 4), "exchange": desc.split("exchange=")[1][:70]}), flush=True)
    plan.set_param("exchange", 0)
    plan.set_param("partition", 0)
