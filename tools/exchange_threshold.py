"""Where the exchange mode stops paying (development tool): C5's shape with N(0, sigma) samples, sigma from 1 to 2.5 — the window of
480 rows holds less and less of them, what is outside goes to memory-side atomics — forced exchange against the classic passes.
    python tools/exchange_threshold.py [samples]"""
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
edges = [np.linspace(-4.0, 4.0, 1025)] * 2
g = torch.Generator(device="cuda")
g.manual_seed(5)
x = torch.empty((1, n), dtype=torch.float64, device="cuda")
y = torch.empty((1, n), dtype=torch.float64, device="cuda").normal_(generator=g)
w = torch.empty((1, n), dtype=torch.float64, device="cuda").uniform_(generator=g)
plan = core._get_plan(edges, _native.CMP_F64, 0)
plan.set_param("partition", 1)


def timed(mode, steps=6):
    plan.set_param("exchange", mode)
    for _ in range(3):
        out = core._bincount_2d_vectorized(x, y, bins=edges, weights=w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        out = core._bincount_2d_vectorized(x, y, bins=edges, weights=w)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, out, plan.describe()


for sigma in (1.0, 1.15, 1.3, 1.45, 1.6, 1.8, 2.0, 2.5):
    x.normal_(0.0, sigma, generator=g)
    t_cl, a, _ = timed(-1)
    t_ex, b, desc = timed(1)
    t_auto, c, desc_auto = timed(0)
    ppm = int(desc_auto.split("exchange_window_ppm_before=")[1].split()[0])
    ok = bool(torch.allclose(a, b, rtol=1e-9, atol=0)) and bool(torch.allclose(a, c, rtol=1e-9, atol=0))
    print(json.dumps({"sigma": sigma, "window_ppm": ppm, "classic_ms": round(t_cl, 4), "exchange_forced_ms": round(t_ex, 4), "auto_ms": round(t_auto, 4), "same_result": ok}), flush=True)
plan.set_param("exchange", 0)
plan.set_param("partition", 0)
