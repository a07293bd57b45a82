"""Development tool: cost of samples that fall outside the bin range (they all meet in the trash slot of a
single-copy LDS histogram).  10^9 N(0,1) f64 samples, edges covering +-half_width."""
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

n = 1_000_000_000
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
g = torch.Generator(device=dev)
g.manual_seed(7)
x = torch.empty(n, dtype=torch.float64, device=dev).normal_(generator=g)
w = torch.empty(n, dtype=torch.float64, device=dev).uniform_(generator=g)
for name, nb, weighted in (("1d_6000_w_single_copy", 6000, True), ("1d_100_w_16_copies", 100, True), ("1d_60000_u_packed", 60000, False)):
    for half in (4.0, 1.0, 0.3, 0.05):
        edges = [np.linspace(-half, half, nb + 1)]
        p = core._get_plan(edges, _native.CMP_F64, 0)
        out = torch.zeros(nb, dtype=torch.float64 if weighted else torch.int64, device=dev)
        v = [_native.make_view(x.data_ptr(), _native.F64, n, 1)]
        wv = _native.make_view(w.data_ptr(), _native.F64, n, 1) if weighted else None
        med, _ = timed(p, v, wv, 1, n, out, weighted, stream, 3, _native)
        print(json.dumps(dict(case=name, half_width=half, ms=round(med, 3), desc=p.describe()[:110])), flush=True)
