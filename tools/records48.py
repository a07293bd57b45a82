#!/usr/bin/env python
"""Packed 8-byte records of the routing pass (float64 weights, one sign) against full float64 records: time per C5 shard and the
difference between the two results; weights of both signs: the exact pass runs in the same call, the plan remembers.
python tools/records48.py [samples]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from xhistogram_amd import core, _native

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 500_000_000
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
y = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
w = torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
edges = [np.linspace(-4, 4, 1025)] * 2


def plan_of():
    cmp_domain, conv, _ = core._compare_domain([np.dtype("f8")] * 2, edges)
    return core._get_plan(conv, cmp_domain, 0)


def run(weights, reps=5):
    core.histogram(x, y, bins=edges, weights=weights)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        h, _ = core.histogram(x, y, bins=edges, weights=weights)
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return h, best * 1e3


plan = plan_of()
t_exact = t_pack = 1e9
alternating = []
for rnd in range(4):  # alternate, so that neither form gets the better half of the box's mood
    plan.set_param("records48", -1)
    h_exact, t = run(w, 3)
    d_exact = plan.describe()
    t_exact = min(t_exact, t)
    alternating.append(round(t, 3))
    plan.set_param("records48", 0)
    h_pack, t = run(w, 3)
    d_pack = plan.describe()
    t_pack = min(t_pack, t)
    alternating.append(round(t, 3))
print(json.dumps({"alternating exact / packed, ms": alternating}))
rel = ((h_pack - h_exact).abs() / h_exact.abs().clamp_min(1e-300)).max().item()
print(json.dumps({"case": "one sign", "samples": n, "exact_ms": round(t_exact, 3), "packed_ms": round(t_pack, 3), "max_rel_diff": rel,
                  "records_exact": d_exact.split("records=")[-1], "records_packed": d_pack.split("records=")[-1]}))
ws = w - 0.5  # both signs
plan.set_param("records48", -1)
h_exact, t_exact = run(ws)
plan.set_param("records48", 0)
t0 = time.perf_counter()
h_first, _ = core.histogram(x, y, bins=edges, weights=ws)  # packed attempt + exact pass in one call
torch.cuda.synchronize()
t_first = (time.perf_counter() - t0) * 1e3
h_later, t_later = run(ws)  # the plan remembers
print(json.dumps({"case": "both signs", "exact_ms": round(t_exact, 3), "first_call_ms": round(t_first, 3), "later_calls_ms": round(t_later, 3),
                  "first_max_abs_diff": (h_first - h_exact).abs().max().item(), "later_max_abs_diff": (h_later - h_exact).abs().max().item(),
                  "records_later": plan.describe().split("records=")[-1]}))
