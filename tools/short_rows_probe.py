"""Development probe: the kernels behind reductions over a leading axis and over many short contiguous rows
(run under rocprofv3 --kernel-trace --stats for the per-kernel split).  python tools/short_rows_probe.py [case ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch

from xhistogram_amd import _native, core

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(3)
edges = np.linspace(-4, 4, 51)
CASES = {
    "time_leading": ((1825, 360, 720), 0),
    "lon_last_720": ((1825, 360, 720), 2),
    "lat_middle": ((1825, 360, 720), 1),
    "time_and_lon": ((1825, 360, 720), (0, 2)),
    "rows_365": ((1_000_000, 365), 1),
    "rows_64": ((5_703_125, 64), 1),
    "rows_20": ((18_250_000, 20), 1),
    "rows_1024": ((356_445, 1024), 1),
    "rows_3650": ((100_000, 3650), 1),
}
CASES.update({"rows_%d" % c: ((365_000_000 // c, c), 1) for c in (100, 128, 200, 256, 300, 400, 512, 720, 2048)})
PARAMS = [kv.split("=") for kv in os.environ.get("XHIST_PROBE_PARAMS", "").split(",") if kv]  # e.g. flat_rows=-1
WEIGHTED = os.environ.get("XHIST_PROBE_W", "") == "1"   # float32 weights
NDIM = int(os.environ.get("XHIST_PROBE_D", "1"))       # inputs of the joint histogram (20 bins each when > 1)
for name in (sys.argv[1:] or list(CASES)):
    shape, axis = CASES[name]
    xs = [torch.empty(shape, dtype=torch.float32, device=dev).normal_(generator=g) for _ in range(NDIM)]
    w = torch.empty(shape, dtype=torch.float32, device=dev).uniform_(generator=g) if WEIGHTED else None
    bins = edges if NDIM == 1 else [np.linspace(-4, 4, 21)] * NDIM
    plan = core._get_plan([np.asarray(b, dtype=np.float64) for b in ([edges] if NDIM == 1 else bins)], _native.CMP_F64, 0)
    for k, v in PARAMS:
        plan.set_param(k, int(v))
    core.histogram(*xs, bins=bins, axis=axis, weights=w)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        h, _ = core.histogram(*xs, bins=bins, axis=axis, weights=w)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    out_b = h.numel() * 8
    in_b = xs[0].numel() * 4 * (NDIM + (1 if WEIGHTED else 0))
    print("%-14s %8.3f ms  in %.2f GB out %.2f GB  -> %.2f TB/s of in+out   %s" % (name, sorted(ts)[2], in_b / 1e9, out_b / 1e9,
          (in_b + out_b) / (sorted(ts)[2] * 1e-3) / 1e12, plan.describe()[:230]), flush=True)
    del xs, w, h
    torch.cuda.empty_cache()
