"""Table-free (arithmetic-edge) digitize against the bucket tables, 1-D histograms of 10^9 samples (development tool).
python tools/arith_ab.py   — prints one JSON line per (bins, weighted, dtype, forced arith or automatic choice)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from xhistogram_amd import _native

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(3)
n = 1_000_000_000
x64 = torch.empty(n, dtype=torch.float64, device=dev).normal_(generator=g)
w64 = torch.empty(n, dtype=torch.float64, device=dev).uniform_(generator=g)
stream = torch.cuda.current_stream(dev).cuda_stream
for f32 in (False, True):
    x = x64[: n // 2].to(torch.float32) if f32 else x64
    m = x.numel()
    for nb in (100, 10_000, 30_000, 60_000):
        for weighted in (False, True):
            if f32 and weighted:
                continue
            edges = [np.linspace(-4.0, 4.0, nb + 1)]
            out = torch.zeros(nb, dtype=torch.float64 if weighted else torch.int64, device=dev)
            for params in ({}, {"arith": 1}):
                plan = _native.Plan(edges, _native.CMP_F64, 0)
                for k, v in params.items():
                    plan.set_param(k, v)
                xv = [_native.make_view(x.data_ptr(), _native.F32 if f32 else _native.F64, m, 1)]
                wv = _native.make_view(w64.data_ptr(), _native.F64, m, 1) if weighted else None
                run = plan.bind(xv, wv, 1, m, out.data_ptr(), weighted, _native.MEM_DEVICE, False, stream)
                for _ in range(2):
                    run()
                torch.cuda.synchronize()
                plan.set_param("profile", 8)
                for _ in range(8):
                    run()
                torch.cuda.synchronize()
                ms = float(np.median(plan.profile_read()))
                byts = m * ((4 if f32 else 8) + (8 if weighted else 0))
                print(json.dumps({"nb": nb, "weighted": int(weighted), "f32": f32, "params": params, "ms": round(ms, 4), "gbs": round(byts / ms / 1e6),
                                  "sum": float(out.sum()), "desc": plan.describe()[:150]}), flush=True)
                plan.close()
