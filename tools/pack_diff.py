"""Differential soak of the packed bucket entries at scale (development tool): random edge sets (uniform draws, geometric, symmetric
log, clusters, mixtures per dimension), float64 / float32 samples, 1-3 inputs, 2*10^7 samples each drawn so that they cover the
edges' range AND land on float32 images of edges by the thousand — counts with pack = 1 must equal counts with pack = -1 exactly.
python tools/pack_diff.py <seconds> [seed0]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from xhistogram_amd import _native

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream(dev).cuda_stream
n = 20_000_000


def edges_of(rng, kind, nb):
    if kind == "random":
        lo, hi = sorted(rng.uniform(-10, 10, 2))
        e = np.sort(rng.uniform(lo, hi, nb + 1))
    elif kind == "geometric":
        e = np.geomspace(10.0 ** rng.uniform(-6, -1), 10.0 ** rng.uniform(0, 3), nb + 1)
        if rng.random() < 0.3:
            e = -e[::-1]
    elif kind == "symlog":
        pos = np.geomspace(10.0 ** rng.uniform(-5, -1), 10.0 ** rng.uniform(0, 2), nb // 2 + 1)
        e = np.concatenate([-pos[::-1], [0.0], pos]) if rng.random() < 0.5 else np.concatenate([-pos[::-1], pos * rng.uniform(0.5, 2)])
    else:  # clusters: pairs / triples of edges a few float32 ulps apart
        base = np.sort(rng.uniform(-5, 5, max(2, nb // 3)))
        e = np.sort(np.concatenate([base, base * (1 + 3e-7), base * (1 - 2e-7)]))
    return np.unique(e)


t_end = time.time() + budget
seed, n_cases, n_packed = seed0, 0, 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    d = int(rng.choice([1, 1, 2, 2, 3]))
    f32 = bool(rng.random() < 0.5)
    kinds = [str(rng.choice(["random", "geometric", "symlog", "clusters"])) for _ in range(d)]
    nbmax = {1: 3000, 2: 250, 3: 40}[d]
    edges = [edges_of(rng, k, int(rng.integers(2, nbmax))) for k in kinds]
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    xs = []
    for e in edges:
        span = float(e[-1] - e[0])
        x = torch.empty(n, dtype=torch.float64, device=dev).uniform_(float(e[0]) - 0.05 * span, float(e[-1]) + 0.05 * span, generator=g)
        # a slice of the samples ON float32 images of edges and on the edges themselves
        et = torch.as_tensor(e, device=dev)
        idx = torch.randint(0, len(e), (n // 50,), device=dev, generator=g)
        x[: n // 100] = et[idx[: n // 100]]
        x[n // 100: n // 50] = et[idx[n // 100:]].to(torch.float32).to(torch.float64)
        xs.append(x.to(torch.float32) if f32 else x)
    tag = _native.F32 if f32 else _native.F64
    views = [_native.make_view(x.data_ptr(), tag, n, 1) for x in xs]
    res, descs = [], []
    for pk in (-1, 1):
        plan = _native.Plan(edges, _native.CMP_F64, 0)
        plan.set_param("pack", pk)
        out = torch.zeros(plan.bins_shape, dtype=torch.int64, device=dev)
        plan.execute(views, None, 1, n, out.data_ptr(), False, _native.MEM_DEVICE, stream=stream)
        torch.cuda.synchronize()
        res.append(out)
        descs.append(plan.describe())
        plan.close()
    if not torch.equal(res[0], res[1]):
        print(json.dumps({"MISMATCH": seed, "kinds": kinds, "f32": f32, "nb": [len(e) - 1 for e in edges], "desc": descs}), flush=True)
        sys.exit(1)
    n_cases += 1
    n_packed += any("scan=%d" % k in descs[1] for k in (6, 7, 8))
    seed += 1
print("pack_diff ok: %d cases (%d of them on packed entries) in %.0f s, seeds %d..%d" % (n_cases, n_packed, budget, seed0, seed - 1))
