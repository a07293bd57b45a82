"""Development tool: a typical ocean-model workload - joint T/S histograms per time step (32 rows x 3*10^7 f32 samples),
unweighted and volume-weighted, for several bin counts; plus mean-T-in-S-bins with two weights in one pass."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
from xhistogram_amd import core, _native
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
rows, cols = 32, 30_000_000
T = torch.empty((rows, cols), dtype=torch.float32, device=dev).normal_(generator=g) * 5 + 10
S = torch.empty((rows, cols), dtype=torch.float32, device=dev).normal_(generator=g) + 35
V = torch.empty((rows, cols), dtype=torch.float32, device=dev).uniform_(generator=g)
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
for nb in (50, 100, 140, 200, 300):
    bins = [np.linspace(-5, 30, nb + 1), np.linspace(30, 40, nb + 1)]
    for wname, w in (("none", None), ("f32", V)):
        ms = timed(lambda: core.histogram(T, S, bins=bins, weights=w, axis=1))
        plan = core._get_plan([np.asarray(b, dtype=np.float64) for b in bins], _native.CMP_F64, 0)
        byts = rows * cols * (8 + (4 if w is not None else 0))
        print(json.dumps(dict(nb=nb, weights=wname, ms=round(ms, 3), gbs=round(byts / ms / 1e6), desc=plan.describe()[:175])), flush=True)
# density + two weights (mean T in S bins)
bins = [np.linspace(30, 40, 101)]
ms = timed(lambda: core.histogram_two_weights(S, bins=bins[0], weights=(V * T, V), axis=1))
print(json.dumps(dict(case="mean_T_in_S_bins_two_weights", ms=round(ms, 3))))
