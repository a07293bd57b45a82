"""Where the DEFAULT rule of the exchange mode loses against the classic passes (development tool): a matrix of histogram shapes
(1-3 inputs) and sample distributions, 3*10^8 float64 samples + float64 weights each; prints default / classic per case and
flags what is more than 3 % slower.    python tools/exchange_cliffs.py [samples] [only this shape, e.g. 1000000 or 512x2048]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xhistogram_amd import _native, core

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000_000
only = sys.argv[2] if len(sys.argv) > 2 else ""  # e.g. "1000000" or "1024x1024": only that shape (a flagged cell looked at again)
_native.require_device(0)
g = torch.Generator(device="cuda"); g.manual_seed(8)
xs = [torch.empty((1, n), dtype=torch.float64, device="cuda") for _ in range(3)]
w = torch.empty((1, n), dtype=torch.float64, device="cuda").uniform_(generator=g)


def fill(kind):
    for i, x in enumerate(xs):
        if kind == "normal":
            x.normal_(0.0, 1.0, generator=g)
        elif kind == "normal_off":  # off-centre, narrower
            x.normal_(1.5 - i, 0.6, generator=g)
        elif kind == "uniform":
            x.uniform_(-4, 4, generator=g)
        elif kind == "bimodal":  # two clusters 4 sigma apart: no window of rows holds both
            x.normal_(0.0, 0.5, generator=g)
            x[0, ::2] += 2.0
            x[0, 1::2] -= 2.0
        elif kind == "exp":
            x.exponential_(1.0, generator=g)
            x -= 4.0
        elif kind == "narrow":  # everything in a handful of rows
            x.normal_(0.3, 0.01, generator=g)


def timed(plan, args, edges, mode, steps=5):
    plan.set_param("exchange", mode)
    for _ in range(2):
        out = core._bincount_2d_vectorized(*args, bins=edges, weights=w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        out = core._bincount_2d_vectorized(*args, bins=edges, weights=w)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, out


shapes = [(1_000_000,), (200_000,), (1024, 1024), (512, 2048), (2048, 512), (4096, 256), (256, 4096), (300, 3000), (1500, 1500), (700, 700),
          (64, 64, 256), (32, 32, 1024), (100, 100, 100), (16, 512, 128), (200, 8, 640)]
bad = 0
for kind in ("normal", "normal_off", "uniform", "bimodal", "exp", "narrow"):
    fill(kind)
    for nbs in shapes:
        if only and "x".join(str(b) for b in nbs) != only:
            continue
        edges = [np.linspace(-4.0, 4.0, nb + 1) for nb in nbs]
        plan = core._get_plan(edges, _native.CMP_F64, 0)
        plan.set_param("partition", 1)
        args = xs[: len(nbs)]
        try:
            t_cl, a = timed(plan, args, edges, -1)
            t_df, b = timed(plan, args, edges, 0)
            desc = plan.describe()
        finally:
            plan.set_param("exchange", 0); plan.set_param("partition", 0)
        ok = bool(torch.allclose(a, b, rtol=1e-9, atol=0, equal_nan=True))
        offered = desc.split("exchange=")[1].split(" exchange_window")[0] if "exchange=" in desc else "-"
        ppm = int(desc.split("exchange_window_ppm_before=")[1].split()[0]) if "exchange_window_ppm_before=" in desc else -1
        slow = t_df > 1.03 * t_cl
        bad += int(slow) + int(not ok)
        print(json.dumps({"samples": kind, "bins": nbs, "classic_ms": round(t_cl, 3), "default_ms": round(t_df, 3), "ratio": round(t_df / t_cl, 3), "offered": offered[:24], "window_ppm": ppm,
                          "same": ok, "flag": "SLOWER" if slow else ("MISMATCH" if not ok else "")}), flush=True)
print("exchange cliffs: %d flagged" % bad)
