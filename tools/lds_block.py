"""Development tool: block-size sweep for histograms with a large LDS footprint."""
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
cases = [("1d_30000_u", [np.linspace(-4, 4, 30001)], False, 8), ("1d_12000_w", [np.linspace(-4, 4, 12001)], True, 16),
         ("1d_6000_w", [np.linspace(-4, 4, 6001)], True, 16), ("1d_3000_w", [np.linspace(-4, 4, 3001)], True, 16),
         ("2d_180x180_u", [np.linspace(-4, 4, 181)] * 2, False, 16), ("2d_120x120_u", [np.linspace(-4, 4, 121)] * 2, False, 16)]
for name, edges, weighted, bps in cases:
    p = core._get_plan(edges, _native.CMP_F64, 0)
    nb = int(np.prod([len(e) - 1 for e in edges]))
    out = torch.zeros(nb, dtype=torch.float64 if weighted else torch.int64, device=dev)
    v = [_native.make_view(x.data_ptr(), _native.F64, n, 1)]
    if len(edges) == 2:
        v.append(_native.make_view(w.data_ptr(), _native.F64, n, 1))
    wv = _native.make_view(w.data_ptr(), _native.F64, n, 1) if weighted else None
    for block in (0, 512, 768, 1024):
        p.set_param("block_threads", block)
        med, mn = timed(p, v, wv, 1, n, out, weighted, stream, 3, _native)
        print(json.dumps(dict(case=name, block=block, ms=round(med, 4), gbs=round(bps * n / med / 1e6), desc=p.describe()[:150])), flush=True)
