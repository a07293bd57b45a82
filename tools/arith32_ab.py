"""A/B of the float32 arithmetic digitize (scan=9) against the threshold tables on launch-bound to mid-size one-row calls,
with N(0,1) samples and with samples on bin centres (never `near`: what the fast path costs without its redo).
python tools/arith32_ab.py -> one line per (case, n, data, variant)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch

from sweep import timed
from xhistogram_amd import _native, core

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
g = torch.Generator(device=dev)
g.manual_seed(11)
NMAX = 100_000_000
xn = torch.empty(NMAX, dtype=torch.float32, device=dev).normal_(generator=g)
for nb, weighted in ((100, False), (1000, True), (50, False), (1000, False)):
    e = np.linspace(-4, 4, nb + 1)
    centres = torch.as_tensor(((np.arange(NMAX // 100) % nb + 0.5) * (8.0 / nb) - 4.0).astype(np.float32), device=dev).repeat(100)
    plan = core._get_plan([e], _native.CMP_F64, 0)
    for data, x in (("normal", xn), ("centres", centres)):
        for n in (10**6, 3 * 10**6, 10**7, 10**8):
            for variant in ("default", "arith32=-1", "default"):
                if variant != "default":
                    k, v = variant.split("=")
                    plan.set_param(k, int(v))
                out = torch.zeros(plan.bins_shape, dtype=torch.float64 if weighted else torch.int64, device=dev)
                xv = [_native.make_view(x.data_ptr(), _native.F32, n, 1)]
                wv = _native.make_view(xn.data_ptr(), _native.F32, n, 1) if weighted else None
                timed(plan, xv, wv, 1, n, out, weighted, stream, 5, _native)
                med, mn = timed(plan, xv, wv, 1, n, out, weighted, stream, 15, _native)
                d = plan.describe()
                print("bins %4d w=%d %-8s n=%9d %-11s med %.2f us min %.2f us | %s" % (
                    nb, weighted, data, n, variant, med * 1e3, mn * 1e3,
                    " ".join(t for t in d.split() if t.split("=")[0] in ("block", "grid", "copies", "scan", "unroll"))), flush=True)
                plan.set_param("arith32", 0)
