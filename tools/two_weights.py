"""Development tool: one fused two-weight pass against two single-weight passes (10^9 samples)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from xhistogram_amd import core

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(3)
x = torch.empty(n, dtype=torch.float64, device=dev).normal_(generator=g)
wa = torch.empty(n, dtype=torch.float64, device=dev).uniform_(generator=g)
wb = torch.empty(n, dtype=torch.float64, device=dev).uniform_(generator=g)
e = np.linspace(-4, 4, 101)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


for name, args, bins in (("1d_100bins", [x], e), ("2d_64x64", [x, wb * 8 - 4], [np.linspace(-4, 4, 65)] * 2)):
    t2 = timed(lambda: (core.histogram(*args, bins=bins, weights=wa), core.histogram(*args, bins=bins, weights=wb)))
    t1 = timed(lambda: core.histogram_two_weights(*args, bins=bins, weights=(wa, wb)))
    ha, hb, _ = core.histogram_two_weights(*args, bins=bins, weights=(wa, wb))
    ra = core.histogram(*args, bins=bins, weights=wa)[0]
    print(json.dumps(dict(case=name, n=n, two_passes_ms=round(t2, 3), fused_ms=round(t1, 3), speedup=round(t2 / t1, 3),
                          max_rel_diff=float(((ha - ra).abs() / ra.abs().clamp_min(1e-300)).max()))), flush=True)
