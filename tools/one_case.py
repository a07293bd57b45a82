"""Development tool: time one 1-D case:  python tools/one_case.py <n_bins> <weighted 0|1>"""
import os, sys, json, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from xhistogram_amd import _native, core
from sweep import timed
n = 1_000_000_000
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
g = torch.Generator(device=dev); g.manual_seed(7)
x = torch.empty(n, dtype=torch.float64, device=dev).normal_(generator=g)
w = torch.empty(n, dtype=torch.float64, device=dev).uniform_(generator=g)
nb = int(sys.argv[1]); weighted = int(sys.argv[2])
edges = [np.linspace(-4, 4, nb + 1)]
p = core._get_plan(edges, _native.CMP_F64, 0)
out = torch.zeros(nb, dtype=torch.float64 if weighted else torch.int64, device=dev)
v = [_native.make_view(x.data_ptr(), _native.F64, n, 1)]
wv = _native.make_view(w.data_ptr(), _native.F64, n, 1) if weighted else None
med, mn = timed(p, v, wv, 1, n, out, bool(weighted), stream, 3, _native)
print(json.dumps(dict(nb=nb, weighted=weighted, ms=round(med, 4), desc=p.describe())), flush=True)
