#!/usr/bin/env python
"""Development probe: generic-family throughput (forced) for A/B builds."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xhistogram_amd import core, _native
n = 300_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.empty(n, dtype=torch.float64, device="cuda").normal_(generator=g)
y = torch.empty(n, dtype=torch.float32, device="cuda").normal_(generator=g)
w = torch.empty(n, dtype=torch.float64, device="cuda").uniform_(generator=g)
def run(name, args, edges, weights=None, force=True):
    dts = [core._np_dtype_of(a) for a in args]
    dom, conv, _ = core._compare_domain(dts, edges)
    plan = core._get_plan(conv, dom, 0)
    plan.set_param("force_generic", 1 if force else 0)
    kw = dict(bins=edges if len(args) > 1 else edges[0], weights=weights)
    for _ in range(2): core.histogram(*args, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); core.histogram(*args, **kw); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    plan.set_param("force_generic", 0)
    print(json.dumps({"case": name, "ms": float(np.median(ts)), "Gsamples_s": n / float(np.median(ts)) / 1e6, "desc": plan.describe()[:40]}), flush=True)
e1 = [np.linspace(-4, 4, 101)]
run("f64_1d_uniform", [x], e1)
run("f64_1d_weighted", [x], e1, w)
run("mixed_f64_f32_2d", [x, y], [np.linspace(-4, 4, 65), np.linspace(-4, 4, 33)], force=False)
rng = np.random.default_rng(1)
e = np.sort(rng.uniform(-4, 4, 257)); e[0], e[-1] = -4, 4
run("f64_1d_nonuniform257", [x], [e])
run("i64_int_edges", [(x * 10).long()], [np.arange(-50, 52)], force=False)
run("bool_weights", [x], e1, w > 0.5, force=False)
