"""Development tool: joint 2-D histograms of 64^2 ... 1024^2 bins against 10^5 ... 10^8 samples (one row): which mode the
library picks (LDS, packed, bin slices, partitioned, memory-side atomics) and the kernel time."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from sweep import timed
from xhistogram_amd import _native, core
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream(dev).cuda_stream
g = torch.Generator(device=dev); g.manual_seed(11)
NMAX = 100_000_000
x = torch.empty(NMAX, dtype=torch.float64, device=dev).normal_(generator=g)
y = torch.empty(NMAX, dtype=torch.float64, device=dev).normal_(generator=g)
w = torch.empty(NMAX, dtype=torch.float64, device=dev).uniform_(generator=g)
for nb in (64, 128, 256, 512, 1024):
    e = np.linspace(-4, 4, nb + 1)
    plan = core._get_plan([e, e], _native.CMP_F64, 0)
    for weighted in (False, True):
        row = {}
        for n in (10**5, 10**6, 10**7, 10**8):
            out = torch.zeros((nb, nb), dtype=torch.float64 if weighted else torch.int64, device=dev)
            v = [_native.make_view(x.data_ptr(), _native.F64, n, 1), _native.make_view(y.data_ptr(), _native.F64, n, 1)]
            wv = _native.make_view(w.data_ptr(), _native.F64, n, 1) if weighted else None
            timed(plan, v, wv, 1, n, out, weighted, stream, 3, _native)
            med, _ = timed(plan, v, wv, 1, n, out, weighted, stream, 7, _native)
            d = plan.describe()
            kv = dict(t.split("=", 1) for t in d.split() if "=" in t)
            row[n] = (round(med * 1e3, 1), kv.get("hist"), kv.get("block"), kv.get("grid"), kv.get("slices"), kv.get("parts"))
        print(json.dumps(dict(nb=nb, weighted=weighted, us=row)), flush=True)
