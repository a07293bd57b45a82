"""Kernel time against input size (10^5 … 10^9 samples) for one-row shapes (C1/C2 and relatives): where the launch-bound
regime ends and how close mid-size inputs (typical dask chunks) get to the streaming rate.
python tools/size_ramp.py [case names]  -> one JSON line per (case, n)"""
import json
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
NMAX = 1_000_000_000
x64 = torch.empty(NMAX, dtype=torch.float64, device=dev).normal_(generator=g)
w64 = torch.empty(NMAX, dtype=torch.float64, device=dev).uniform_(generator=g)
x32 = torch.empty(NMAX, dtype=torch.float32, device=dev).normal_(generator=g)
y64 = w64  # second input of the 2-D case (uniform [0, 1))
i32 = (x32[: NMAX // 2] * 20).to(torch.int32)
e100, e1000 = np.linspace(-4, 4, 101), np.linspace(-4, 4, 1001)
CASES = (
    ("f64+w", [x64], w64, [e100]), ("f64", [x64], None, [e100]), ("f32", [x32], None, [e100]),
    ("f64 1000 bins", [x64], None, [e1000]), ("f32+w32 1000 bins", [x32], x32, [e1000]),
    ("2-D f64 32x32", [x64, y64], None, [np.linspace(-4, 4, 33), np.linspace(0, 1, 33)]),
    ("i32 161 bins", [i32], None, [np.arange(-80, 82) - 0.5]),
)
TAG = {torch.float64: _native.F64, torch.float32: _native.F32, torch.int32: _native.I32}
only = sys.argv[1:]
for name, xs, w, edges in CASES:
    if only and name not in only:
        continue
    plan = core._get_plan(edges, _native.CMP_F64, 0)
    for n in (10**5, 3 * 10**5, 10**6, 3 * 10**6, 10**7, 3 * 10**7, 10**8, 3 * 10**8, 10**9):
        if any(n > a.numel() for a in xs):
            continue
        out = torch.zeros(plan.bins_shape, dtype=torch.float64 if w is not None else torch.int64, device=dev)
        v = [_native.make_view(a.data_ptr(), TAG[a.dtype], n, 1) for a in xs]
        wv = _native.make_view(w.data_ptr(), TAG[w.dtype], n, 1) if w is not None else None
        timed(plan, v, wv, 1, n, out, w is not None, stream, 5, _native)
        med, _ = timed(plan, v, wv, 1, n, out, w is not None, stream, 11, _native)
        by = n * (sum(a.element_size() for a in xs) + (w.element_size() if w is not None else 0))
        d = plan.describe()
        print(json.dumps(dict(case=name, n=n, us=round(med * 1e3, 2), TBps=round(by / med / 1e9, 3),
                              geom=" ".join(t for t in d.split() if t.split("=")[0] in ("block", "grid", "copies", "direct_store")))), flush=True)
