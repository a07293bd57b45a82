#!/usr/bin/env python
"""(N, k) tables histogrammed per column (axis=0): which shapes still take a copying route?"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xhistogram_amd import core, _native
N = 25_000_000
for dt in (torch.float32, torch.float64):
    for k in (4, 16):
        x = torch.empty((N, k), dtype=dt, device="cuda").normal_()
        w = torch.empty((N, k), dtype=dt, device="cuda").uniform_()
        for nb in (50, 5000, 200_000):
            for weighted in (False, True):
                e = np.linspace(-4, 4, nb + 1)
                kw = dict(bins=e, axis=0, weights=w if weighted else None)
                for _ in range(2):
                    core.histogram(x, **kw)
                torch.cuda.synchronize()
                ts = []
                for _ in range(5):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); core.histogram(x, **kw); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
                ms = float(np.median(ts))
                nbytes = x.element_size() * x.numel() * (2 if weighted else 1)
                desc = core._get_plan([e], _native.CMP_F64, 0).describe()
                print(json.dumps({"dtype": str(dt)[6:], "k": k, "bins": nb, "weighted": weighted, "ms": round(ms, 3), "TBs": round(nbytes / ms / 1e9, 2), "desc": desc[:70]}), flush=True)
        del x, w
