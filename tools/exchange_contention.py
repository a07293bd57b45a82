"""What a C5-shaped call costs while somebody else holds compute units (VERDICT r5 "next" #3): the classic passes, the exchange
mode on a free GPU, the exchange mode while xhist_debug_hold_cus keeps k compute units busy on another stream (k = 1, 8, 64), and
while a stream of torch matmuls runs beside it.  HIP-event time of the call on its own stream, the notes the GPU left
(arrival misses / aborts in flight), and whether the result equals the classic one.
    python tools/exchange_contention.py [samples]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from xhistogram_amd import _native, core

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 27
_native.require_device(0)
edges = [np.linspace(-4.0, 4.0, 1025)] * 2
g = torch.Generator(device="cuda"); g.manual_seed(5)
x = torch.empty((1, n), dtype=torch.float64, device="cuda").normal_(generator=g)
y = torch.empty((1, n), dtype=torch.float64, device="cuda").normal_(generator=g)
w = torch.empty((1, n), dtype=torch.float64, device="cuda").uniform_(generator=g)
plan = core._get_plan(edges, _native.CMP_F64, 0)
plan.set_param("partition", 1)
side = torch.cuda.Stream()


def note(desc, key):
    return int(desc.split(key + "=")[1].split()[0])


def call(reps=5, before=None):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        if before:
            before()
            time.sleep(0.01)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = core._bincount_2d_vectorized(x, y, bins=edges, weights=w)
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    torch.cuda.synchronize()
    core._bincount_2d_vectorized(x, y, bins=edges, weights=w)
    torch.cuda.synchronize()
    return ts, out, plan.describe()


plan.set_param("exchange", -1)
for _ in range(3):
    core._bincount_2d_vectorized(x, y, bins=edges, weights=w)
ts, classic, _ = call()
print(json.dumps({"leg": "classic passes, free GPU", "samples": n, "ms": [round(t, 3) for t in ts]}), flush=True)
plan.set_param("exchange", 0)
for _ in range(3):
    core._bincount_2d_vectorized(x, y, bins=edges, weights=w)
ts, out, d = call()
print(json.dumps({"leg": "exchange mode, free GPU", "ms": [round(t, 3) for t in ts], "same_as_classic": bool(torch.allclose(out, classic, rtol=2.0 ** -34, atol=0)),
                  "arrival_misses": note(d, "exchange_arrival_misses"), "aborts": note(d, "exchange_aborts")}), flush=True)
for k in (1, 8, 64):
    def hold(k=k):
        _native.debug_hold_cus(k, 96 * 1024, 50_000, stream=side.cuda_stream)  # 50 ms, far longer than the call
    # (the classic passes under the same contention: their grids are one workgroup per compute unit too, so with any of them
    #  held the last workgroups run in a second round — what a held compute unit costs ANY full-chip kernel)
    plan.set_param("exchange", -1)
    ts, _, _ = call(before=hold)
    print(json.dumps({"leg": "classic passes, %d compute unit(s) held for 50 ms on another stream" % k, "ms": [round(t, 3) for t in ts]}), flush=True)
    plan.set_param("exchange", 0)
    m0 = note(plan.describe(), "exchange_arrival_misses")
    ts, out, d = call(before=hold)
    print(json.dumps({"leg": "exchange mode, %d compute unit(s) held for 50 ms on another stream" % k, "ms": [round(t, 3) for t in ts],
                      "same_as_classic": bool(torch.allclose(out, classic, rtol=2.0 ** -34, atol=0)), "arrival_misses": note(d, "exchange_arrival_misses") - m0,
                      "aborts": note(d, "exchange_aborts"), "mode_still_offered": "exchange=if the probe" in d}), flush=True)
a = torch.randn(8192, 8192, device="cuda", dtype=torch.float32)
def mm():
    with torch.cuda.stream(side):
        for _ in range(20):
            a @ a
m0 = note(plan.describe(), "exchange_arrival_misses")
ts, out, d = call(before=mm)
print(json.dumps({"leg": "exchange mode beside a stream of 8192^3 float32 matmuls", "ms": [round(t, 3) for t in ts],
                  "same_as_classic": bool(torch.allclose(out, classic, rtol=2.0 ** -34, atol=0)), "arrival_misses": note(d, "exchange_arrival_misses") - m0,
                  "aborts": note(d, "exchange_aborts")}), flush=True)
plan.set_param("exchange", 0); plan.set_param("partition", 0)
