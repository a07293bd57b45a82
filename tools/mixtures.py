#!/usr/bin/env python
"""Dtype mixtures that used to run in the generic family (scalar loads behind a dtype switch): throughput next to the
homogeneous vector kernel that moves the same bytes."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xhistogram_amd import core, _native

n = 200_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
x32 = torch.empty(n, dtype=torch.float32, device="cuda").normal_(generator=g)
y64 = torch.empty(n, dtype=torch.float64, device="cuda").normal_(generator=g)
y32 = y64.float()
x64 = x32.double()
xi = (x32 * 10).int(); yi = (y32 * 10).int()
wi = torch.randint(0, 10, (n,), dtype=torch.int32, device="cuda")
wf = wi.double()
e = np.linspace(-4, 4, 65); ei = e * 10
cases = [
    ("f32 x f64 joint 64x64", (x32, y64), dict(bins=[e, e]), 12),
    ("f64 x f64 joint 64x64 (vector kernel, 16 B)", (x64, y64), dict(bins=[e, e]), 16),
    ("f32 x f32 joint 64x64 (vector kernel, 8 B)", (x32, y32), dict(bins=[e, e]), 8),
    ("i32 x i32 joint 64x64", (xi, yi), dict(bins=[ei, ei]), 8),
    ("f64, int32 weights, 100 bins", (y64,), dict(bins=np.linspace(-4, 4, 101), weights=wi), 12),
    ("f64, f64 weights, 100 bins (vector kernel, 16 B)", (y64,), dict(bins=np.linspace(-4, 4, 101), weights=wf), 16),
    ("f32, int32 weights, 100 bins", (x32,), dict(bins=np.linspace(-4, 4, 101), weights=wi), 8),
    ("u8 x f32 joint 64x64", ((x32 * 30 + 128).clamp(0, 255).to(torch.uint8), y32), dict(bins=[np.linspace(0, 255, 65), e]), 5),
]
for name, args, kw, nbytes in cases:
    for forced in (0, 1):
        blist = kw["bins"] if isinstance(kw["bins"], list) else [kw["bins"]]
        doms = core._compare_domain([core._np_dtype_of(a) for a in args], blist)
        plan = core._get_plan(doms[1], doms[0], 0)
        plan.set_param("force_generic", forced)
        try:
            for _ in range(2):
                core.histogram(*args, **kw)
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); core.histogram(*args, **kw); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
            ms = float(np.median(ts))
            print(json.dumps({"case": name, "forced_generic": forced, "ms": round(ms, 3), "Gsamples_s": round(n / ms / 1e6, 1), "GBs": round(n * nbytes / ms / 1e6, 0),
                              "family": plan.describe()[:32]}), flush=True)
        finally:
            plan.set_param("force_generic", 0)
