"""Development tool: look for performance cliffs over realistic (rows, cols, bins) shapes; prints achieved GB/s."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from xhistogram_amd import _native, core

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(5)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.025:  # warm up past the clock excursion of the first ~13 ms of a burst
        fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


CASES = [
    # rows, cols, dtype, bins per dim, weighted
    (365, 1_000_000, torch.float32, (100, 100), True),
    (365, 1_000_000, torch.float32, (100, 100), False),
    (3650, 100_000, torch.float32, (50, 50), True),
    (12, 50_000_000, torch.float32, (200, 200), True),
    (12, 50_000_000, torch.float64, (60,), True),
    (100_000, 1000, torch.float32, (50,), False),
    (100_000, 1000, torch.float32, (30, 30), True),
    (1, 1_000_000_000, torch.float32, (1000,), False),
    (1, 500_000_000, torch.float32, (256, 256), True),
    (64, 8_000_000, torch.float64, (20, 20, 20), False),
    (2000, 200_000, torch.float32, (400,), True),
    (40, 10_000_000, torch.float32, (2000,), True),
]
def edges_of(kind, nb, seed):
    if kind == "random":  # sorted uniform draws, end points kept (BASELINE C3's kind)
        e = np.sort(np.random.default_rng(seed).uniform(-4, 4, nb + 1))
        e[0], e[-1] = -4.0, 4.0
        return e
    if kind == "geometric":
        return np.geomspace(1e-3, 4.0, nb + 1)
    return np.linspace(-4, 4, nb + 1)


# non-uniform edges (round 4: packed bucket entries on a linear or a float-bit-pattern grid) over the shape classes above
NONUNIFORM = [
    (456, 1_036_800, torch.float32, (50,), False, "random"),
    (456, 1_036_800, torch.float32, (50,), False, "geometric"),
    (1, 500_000_000, torch.float64, (256, 256), False, "random"),
    (1, 500_000_000, torch.float64, (200, 200), False, "geometric"),
    (1, 1_000_000_000, torch.float32, (2000,), False, "geometric"),
    (1, 500_000_000, torch.float64, (300,), True, "geometric"),
    (1_000_000, 365, torch.float32, (50,), False, "geometric"),
    (64, 8_000_000, torch.float32, (64, 64), True, "random"),
]
for case in CASES + NONUNIFORM:
    rows, cols, dt, nbs, weighted = case[:5]
    kind = case[5] if len(case) > 5 else "linspace"
    d = len(nbs)
    arrs = [torch.empty((rows, cols), dtype=dt, device=dev).normal_(generator=g) for _ in range(d)]
    w = torch.empty((rows, cols), dtype=dt, device=dev).uniform_(generator=g) if weighted else None
    bins = [edges_of(kind, nb, 7 + k) for k, nb in enumerate(nbs)]
    ms = timed(lambda: core.histogram(*arrs, bins=bins if d > 1 else bins[0], weights=w, axis=1))
    plan = core._get_plan([np.asarray(b, dtype=np.float64) for b in bins], _native.CMP_F64, 0)
    byts = rows * cols * (d + (1 if weighted else 0)) * arrs[0].element_size()
    print(json.dumps(dict(rows=rows, cols=cols, dtype=str(dt).split(".")[1], bins=list(nbs), weighted=weighted, edges=kind, ms=round(ms, 3),
                          gbs=round(byts / ms / 1e6), desc=plan.describe()[:120])), flush=True)
    del arrs, w
    torch.cuda.empty_cache()
