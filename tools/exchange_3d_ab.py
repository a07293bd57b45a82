"""development: three float64 inputs + weights, 64 x 64 x 256 bins, exchange mode (default rule) against the classic passes"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xhistogram_amd import _native, core

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000_000
_native.require_device(0)
g = torch.Generator(device="cuda"); g.manual_seed(7)
xs = [torch.empty((1, n), dtype=torch.float64, device="cuda").normal_(generator=g) for _ in range(3)]
w = torch.empty((1, n), dtype=torch.float64, device="cuda").uniform_(generator=g)
for nbs in ((64, 64, 256), (100, 100, 100), (32, 32, 1024)):
    edges = [np.linspace(-4.0, 4.0, nb + 1) for nb in nbs]
    plan = core._get_plan(edges, _native.CMP_F64, 0)
    plan.set_param("partition", 1)
    res = {}
    for mode in (-1, 0, 1):
        plan.set_param("exchange", mode)
        for _ in range(3):
            out = core._bincount_2d_vectorized(*xs, bins=edges, weights=w)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            out = core._bincount_2d_vectorized(*xs, bins=edges, weights=w)
        e1.record(); torch.cuda.synchronize()
        res[mode] = (e0.elapsed_time(e1) / 5, out, plan.describe())
    ppm = int(res[0][2].split("exchange_window_ppm_before=")[1].split()[0])
    print(json.dumps({"bins": nbs, "classic_ms": round(res[-1][0], 4), "default_ms": round(res[0][0], 4), "forced_ms": round(res[1][0], 4), "window_ppm": ppm,
                      "same": bool(torch.allclose(res[-1][1], res[0][1], rtol=1e-9, atol=0)) and bool(torch.allclose(res[-1][1], res[1][1], rtol=1e-9, atol=0)),
                      "exchange": res[0][2].split("exchange=")[1][:50]}), flush=True)
    plan.set_param("exchange", 0); plan.set_param("partition", 0)
