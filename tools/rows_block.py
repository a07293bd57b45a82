#!/usr/bin/env python
"""Development probe: block size of the row-streaming kernels vs row length (many rows)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    import torch
    from xhistogram_amd import core, _native
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(3)
    edges = np.linspace(-4, 4, 51)
    plan = core._get_plan([edges], _native.CMP_F64, 0)
    plan.set_param("lanes", -1)
    for cols in (1024, 2048, 3650, 8192, 16384, 65536, 262144):
        rows = max(1, 365_000_000 // cols)
        a = torch.empty((rows, cols), dtype=torch.float32, device=dev).normal_(generator=g)
        for block in (0, 64, 128, 256, 512):
            for gridmul in (0, 1):
                plan.set_param("block_threads", block)
                plan.set_param("grid_blocks", 0 if gridmul == 0 else rows)
                core.histogram(a, bins=edges, axis=1); torch.cuda.synchronize()
                ts = []
                for _ in range(3):
                    t0 = time.perf_counter(); core.histogram(a, bins=edges, axis=1); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
                t = float(np.median(ts))
                print(json.dumps({"cols": cols, "rows": rows, "block": block, "one_wg_per_row": gridmul, "ms": t * 1e3, "gbs": a.numel() * 4 / t / 1e9, "desc": plan.describe()[:120]}), flush=True)
        del a
    plan.set_param("block_threads", 0); plan.set_param("grid_blocks", 0); plan.set_param("lanes", 0)

if __name__ == "__main__":
    main()
