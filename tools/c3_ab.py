"""C3 (two float64 inputs, 256 x 256 non-uniform bins) — packed bucket entries against the two-level tables (development tool).
python tools/c3_ab.py [n_samples]   — one JSON line per variant; counts of every variant must be identical."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from xhistogram_amd import _native

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000_000
only = sys.argv[2] if len(sys.argv) > 2 and "=" not in sys.argv[2] else None  # substring of a case name
f32 = os.environ.get("C3_AB_F32") == "1"  # float32 samples (float64 weights stay)
tune = dict(kv.split("=") for kv in sys.argv[2:] if "=" in kv)  # plan parameters applied to every variant, e.g. block_threads=1024
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1234)
sdt = torch.float32 if f32 else torch.float64
x = torch.empty(n, dtype=sdt, device=dev).normal_(generator=g)
y = torch.empty(n, dtype=sdt, device=dev).normal_(generator=g)
w = torch.empty(n, dtype=torch.float64, device=dev).uniform_(generator=g)
stream = torch.cuda.current_stream(dev).cuda_stream


def nonuniform(k, seed):
    e = np.sort(np.random.default_rng(seed).uniform(-4, 4, k))
    e[0], e[-1] = -4.0, 4.0
    return e


cases = [
    ("c3", [nonuniform(257, 1), nonuniform(257, 2)], False),
    ("2d 64x64 nonuniform, weighted", [nonuniform(65, 3), nonuniform(65, 4)], True),
    ("2d 64x64 nonuniform", [nonuniform(65, 3), nonuniform(65, 4)], False),
    ("1d 257 nonuniform", [nonuniform(257, 1)], False),
    ("1d 257 nonuniform, weighted", [nonuniform(257, 1)], True),
    ("1d 2001 geometric", [np.geomspace(1e-3, 4.0, 2001)], False),
    ("1d 300 geometric, weighted", [np.geomspace(1e-3, 4.0, 301)], True),
    ("1d 401 symlog", [np.concatenate([-np.geomspace(4.0, 1e-3, 200), [0.0], np.geomspace(1e-3, 4.0, 200)])], False),
    ("2d 121 symlog x 200 geometric, weighted", [np.concatenate([-np.geomspace(4.0, 1e-2, 60), [0.0], np.geomspace(1e-2, 4.0, 60)]), np.geomspace(1e-3, 4.0, 51)], True),
    ("2d 200 geometric x 200 geometric", [np.geomspace(1e-3, 4.0, 201), np.geomspace(1e-2, 5.0, 201)], False),
]
for name, edges, weighted in cases:
    if only and only not in name:
        continue
    ref = None
    for params in ({"pack": -1}, {"pack": 0}, {"pack": 1}):
        plan = _native.Plan(edges, _native.CMP_F64, 0)
        params = dict(params, **{k: int(v) for k, v in tune.items()})
        for k, v in params.items():
            plan.set_param(k, v)
        arrs = [x, y][: len(edges)]
        xv = [_native.make_view(a.data_ptr(), _native.F32 if f32 else _native.F64, n, 1) for a in arrs]
        wv = _native.make_view(w.data_ptr(), _native.F64, n, 1) if weighted else None
        out = torch.zeros(plan.bins_shape, dtype=torch.float64 if weighted else torch.int64, device=dev)
        run = plan.bind(xv, wv, 1, n, out.data_ptr(), weighted, _native.MEM_DEVICE, False, stream)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        plan.set_param("profile", 20)
        for _ in range(20):
            run()
        torch.cuda.synchronize()
        t = plan.profile_read()
        ms = float(np.mean(t))
        byts = n * ((4 if f32 else 8) * len(edges) + (8 if weighted else 0))
        res = out.cpu().numpy()
        if ref is None:
            ref = res
        same = bool(np.array_equal(ref, res)) if not weighted else bool(np.allclose(ref, res, rtol=1e-9, atol=0))
        print(json.dumps({"case": name, "params": params, "ms_mean": round(ms, 4), "ms_min": round(float(np.min(t)), 4), "ms_max": round(float(np.max(t)), 4),
                          "frac": round(byts / ms / 1e6 / 8000, 4), "same_as_tables": same, "desc": plan.describe()[:200]}), flush=True)
        plan.close()
