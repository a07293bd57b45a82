"""Development tool: time one 1-D uniform-bin case on 10^9 samples.
    python tools/one_case.py <n_bins> <weighted 0|1> [f64|f32] [key=value plan parameters ...]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch

from sweep import timed
from xhistogram_amd import _native, core

nb = int(sys.argv[1])
weighted = int(sys.argv[2])
f32 = len(sys.argv) > 3 and sys.argv[3] == "f32"
params = dict(kv.split("=") for kv in sys.argv[3:] if "=" in kv)
n = 1_000_000_000
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
g = torch.Generator(device=dev)
g.manual_seed(7)
x = torch.empty(n, dtype=torch.float32 if f32 else torch.float64, device=dev).normal_(generator=g)
w = torch.empty(n, dtype=torch.float64, device=dev).uniform_(generator=g) if weighted else None
edges = [np.linspace(-4, 4, nb + 1)]
p = core._get_plan(edges, _native.CMP_F64, 0)
for k, v in params.items():
    p.set_param(k, int(v))
out = torch.zeros(nb, dtype=torch.float64 if weighted else torch.int64, device=dev)
v = [_native.make_view(x.data_ptr(), _native.F32 if f32 else _native.F64, n, 1)]
wv = _native.make_view(w.data_ptr(), _native.F64, n, 1) if weighted else None
med, mn = timed(p, v, wv, 1, n, out, bool(weighted), stream, 5, _native)
bps = (4 if f32 else 8) + (8 if weighted else 0)
print(json.dumps(dict(nb=nb, weighted=weighted, f32=f32, params=params, ms=round(med, 4), gbs=round(bps * n / med / 1e6), desc=p.describe())), flush=True)
