#!/usr/bin/env python
"""Views no three strides describe (strided in both directions): histogram time, which includes the copy into [rows, cols]
blocks — the library's strided-copy kernel against torch's (XHIST_AMD_TORCH_COPY=1).  python tools/strided_views.py"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from xhistogram_amd import core

if os.environ.get("XHIST_AMD_TORCH_COPY") == "1":
    core._torch_contiguous = lambda a: a.contiguous()
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn((4096, 8192), dtype=torch.float32, device="cuda", generator=g)
t = torch.randn((64, 720, 1440), dtype=torch.float32, device="cuda", generator=g)
edges = np.linspace(-4, 4, 65)
cases = (("x[:, ::2] over axis 0", x[:, ::2], 0), ("x.T[::2] over axis 1", x.T[::2], 1), ("t[:, ::2, ::3] over axes (0, 2)", t[:, ::2, ::3], (0, 2)),
         ("t.permute(2, 0, 1)[::2] over axis 2", t.permute(2, 0, 1)[::2], 2))
for name, v, axis in cases:
    want = np.histogram(v.cpu().numpy(), bins=edges)[0].sum()
    for _ in range(2):
        h, _e = core.histogram(v, bins=edges, axis=axis)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        h, _e = core.histogram(v, bins=edges, axis=axis)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    assert int(h.sum()) == int(want)
    print(json.dumps({"case": name, "elements": v.numel(), "ms": round(ms, 3), "copy": "torch" if os.environ.get("XHIST_AMD_TORCH_COPY") == "1" else "copy_nd"}), flush=True)
