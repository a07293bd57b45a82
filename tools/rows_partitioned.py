#!/usr/bin/env python
"""A few rows of a joint histogram beyond LDS (time steps of a T/S census): rows through the routing pass several at a time
against one row per pass (XHIST_AMD_ROW_BATCH=0).  python tools/rows_partitioned.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xhistogram_amd import core

g = torch.Generator(device="cuda").manual_seed(0)
for rows, n, nb, dt in ((32, 30_000_000, 300, torch.float32), (12, 100_000_000, 300, torch.float32), (8, 60_000_000, 512, torch.float64), (2, 500_000_000, 1024, torch.float64)):
    x = torch.randn((rows, n), dtype=dt, device="cuda", generator=g)
    y = torch.randn((rows, n), dtype=dt, device="cuda", generator=g)
    w = torch.rand((1, n), dtype=dt, device="cuda", generator=g).expand(rows, n)  # cell volumes: one row, broadcast over time
    edges = [np.linspace(-4, 4, nb + 1)] * 2
    for _ in range(2):
        h, _e = core.histogram(x, y, bins=edges, weights=w, axis=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        h, _e = core.histogram(x, y, bins=edges, weights=w, axis=1)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    cmp_domain, conv, _ = core._compare_domain([np.dtype("f4" if dt == torch.float32 else "f8")] * 2, edges)
    d = core._get_plan(conv, cmp_domain, 0).describe()
    print(json.dumps({"rows": rows, "samples_per_row": n, "bins": [nb, nb], "dtype": str(dt), "ms": round(ms, 3), "batch": os.environ.get("XHIST_AMD_ROW_BATCH", "1"),
                      "rows_per_pass": d.split("rows_per_pass=")[-1].split(" ")[0] if "rows_per_pass=" in d else d[:60], "checksum": float(h.sum())}), flush=True)
    del x, y, w
