"""C4 after idle (VERDICT r3 "next" #6): what a burst of dask-chunk-sized calls costs on a GPU that was idle, and whether software
can move it.  Variants, each after 0.5 s of idle, 20 launches of the C4 shard kernel (456 x 1 036 800 float32, 50 bins) timed one
by one with the library's HIP events:
  none        the burst as it is (bench.py's cold_frac)
  ramp50us    a ~50 us streaming kernel first ("wake the clocks")
  ramp2ms     ~2 ms of streaming reads first (the dip of the clock trace sits 1.5-4.5 ms into a burst)
  ramp6ms     ~6 ms of streaming reads first (past the dip)
  one_launch  the same 20 x 456 rows as ONE launch over 9120 rows (what coalescing the chunks of a graph would do)
python tools/c4_cold.py  -> one JSON line per variant"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

from xhistogram_amd import _native, core

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(1234)
rows, cols, burst = 456, 720 * 1440, 20
big = torch.empty((rows * burst, cols), dtype=torch.float32, device=dev).normal_(generator=g)  # 37.8 GB
x = big[:rows]
edges = [np.linspace(-4.0, 4.0, 51)]
plan = core._get_plan(edges, _native.CMP_F64, 0)
stream = torch.cuda.current_stream(dev).cuda_stream
out = torch.zeros((rows, 50), dtype=torch.int64, device=dev)
out_big = torch.zeros((rows * burst, 50), dtype=torch.int64, device=dev)
run = plan.bind([_native.make_view(x.data_ptr(), _native.F32, cols, 1)], None, rows, cols, out.data_ptr(), False, _native.MEM_DEVICE, False, stream)
run_big = plan.bind([_native.make_view(big.data_ptr(), _native.F32, cols, 1)], None, rows * burst, cols, out_big.data_ptr(), False, _native.MEM_DEVICE, False, stream)
scratch = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)  # 256 MB: one .sum() of it is ~40 us of streaming reads


def ramp(us):
    n = max(1, int(round(us / 40.0)))
    for _ in range(n):
        scratch.sum()


for _ in range(60):  # module load, plan warm-up, steady clocks
    run()
run_big()
torch.cuda.synchronize()
plan.set_param("profile", 200)
for _ in range(200):
    run()
torch.cuda.synchronize()
warm = float(np.mean(plan.profile_read()))
byts = rows * cols * 4
print(json.dumps({"variant": "warm (200 launches in a tight loop)", "kernel_ms_mean": round(warm, 4), "frac": round(byts / warm / 1e6 / 8000, 4)}), flush=True)
for variant in ("none", "ramp50us", "ramp2ms", "ramp6ms", "one_launch", "none"):
    res = []
    for rep in range(3):
        torch.cuda.synchronize()
        time.sleep(0.5)
        if variant.startswith("ramp"):
            ramp({"ramp50us": 50, "ramp2ms": 2000, "ramp6ms": 6000}[variant])
        if variant == "one_launch":
            plan.set_param("profile", 1)
            run_big()
            torch.cuda.synchronize()
            ms = plan.profile_read()
            res.append(float(ms[0]) / burst)
        else:
            plan.set_param("profile", burst)
            for _ in range(burst):
                run()
            torch.cuda.synchronize()
            res.append(float(np.mean(plan.profile_read())))
    m = float(np.mean(res))
    print(json.dumps({"variant": variant, "kernel_ms_per_456_rows": [round(v, 4) for v in res], "mean": round(m, 4), "frac": round(byts / m / 1e6 / 8000, 4),
                      "vs_warm": round(m / warm, 3)}), flush=True)
