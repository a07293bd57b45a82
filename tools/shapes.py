#!/usr/bin/env python
"""Development probe: throughput of awkward shapes (many short rows, leading-axis reductions)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch

    from xhistogram_amd import core

    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    edges = np.linspace(-4, 4, 51)

    def run(name, x, axis, reps=5, **kw):
        core.histogram(x, bins=edges, axis=axis, **kw)
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            h, _ = core.histogram(x, bins=edges, axis=axis, **kw)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        nbytes = x.numel() * x.element_size()
        print(json.dumps({"case": name, "shape": list(x.shape), "axis": axis, "ms": t * 1e3, "gbs": nbytes / t / 1e9,
                          "frac": nbytes / t / 8e12, "out": list(h.shape)}), flush=True)

    # time-major field, like (time, lat, lon): 1825 x 360 x 720 f32 = 1.9 GB
    t = torch.empty((1825, 360, 720), dtype=torch.float32, device=dev).normal_(generator=g)
    run("reduce_lat_lon_rows_are_long(C4)", t, (1, 2))
    run("reduce_time_leading_axis", t, 0)
    run("reduce_lon_only_short_rows_720", t, 2)
    run("reduce_lat_middle_axis", t, 1)
    run("reduce_time_and_lon", t, (0, 2))
    del t
    from xhistogram_amd import _native
    plan = core._get_plan([edges], _native.CMP_F64, 0)
    for cols in (20, 64, 128, 256, 365, 512, 720, 1024, 2048, 3650, 8192, 16384):
        rows = max(1, 365_000_000 // cols)
        a = torch.empty((rows, cols), dtype=torch.float32, device=dev).normal_(generator=g)
        for mode, val in (("lanes", 1), ("rows", -1)):
            plan.set_param("lanes", val)
            run("f32_%dx%d_%s" % (rows, cols, mode), a, 1, reps=3)
        plan.set_param("lanes", 0)
        del a
    s = torch.empty((1_000_000, 365), dtype=torch.float32, device=dev).normal_(generator=g)
    run("1M_rows_of_365", s, 1)
    s2 = torch.empty((100_000, 3650), dtype=torch.float32, device=dev).normal_(generator=g)
    run("100k_rows_of_3650", s2, 1)
    s3 = torch.empty((10_000_000, 20), dtype=torch.float64, device=dev).normal_(generator=g)
    run("10M_rows_of_20_f64", s3, 1)


if __name__ == "__main__":
    main()
