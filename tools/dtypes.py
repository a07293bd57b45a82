#!/usr/bin/env python
"""Development probe: throughput by sample dtype (which kernel family serves it)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xhistogram_amd import core, _native
edges = np.linspace(-4, 4, 101)
iedges = np.arange(-50, 52, 1)
n = 500_000_000
g = torch.Generator(device="cuda"); g.manual_seed(1)
base = torch.empty(n, dtype=torch.float32, device="cuda").normal_(generator=g)
for name, x, e in (("f64", base.double(), edges), ("f32", base, edges), ("f16", base.half(), edges),
                   ("i32_float_edges", (base * 10).int(), edges * 10), ("i64_float_edges", (base * 10).long(), edges * 10),
                   ("i64_int_edges(datetime-like)", (base * 10).long(), iedges), ("u8", (base * 30 + 128).clamp(0, 255).to(torch.uint8), np.linspace(0, 255, 52))):
    for _ in range(2):
        core.histogram(x, bins=e)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); core.histogram(x, bins=e); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    dom, conv, _ = core._compare_domain([core._np_dtype_of(x)], [e])
    desc = core._get_plan(conv, dom, 0).describe()
    print(json.dumps({"dtype": name, "ms": ms, "Gsamples_s": n / ms / 1e6, "gbs": n * x.element_size() / ms / 1e6, "desc": desc[:60]}), flush=True)
    del x
