"""Development tool: API-level shapes that leave the main vector kernels (broadcast weights, strided inputs,
integer / datetime compare domains, 4 inputs, non-adjacent reduced axes); achieved GB/s of the arrays read."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from xhistogram_amd import core

dev = torch.device("cuda", 0)
g = torch.Generator(device=dev)
g.manual_seed(11)


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def report(name, ms, nbytes):
    print(json.dumps(dict(case=name, ms=round(ms, 3), gbs=round(nbytes / ms / 1e6))), flush=True)


T, Y, X = 365, 720, 1440
a = torch.empty((T, Y, X), dtype=torch.float32, device=dev).normal_(generator=g)
e = np.linspace(-4, 4, 51)
nb = a.numel() * 4
report("full_reduce", timed(lambda: core.histogram(a, bins=e)), nb)
report("axis_lat_lon", timed(lambda: core.histogram(a, bins=e, axis=(1, 2))), nb)
report("axis_time", timed(lambda: core.histogram(a, bins=e, axis=0)), nb)
report("axis_time_lon_nonadjacent", timed(lambda: core.histogram(a, bins=e, axis=(0, 2))), nb)
w_lat = torch.cos(torch.linspace(-1.5, 1.5, Y, device=dev)).reshape(1, Y, 1)
report("weights_broadcast_lat_full_reduce", timed(lambda: core.histogram(a, bins=e, weights=w_lat)), nb)
report("weights_broadcast_lat_axis_lat_lon", timed(lambda: core.histogram(a, bins=e, weights=w_lat, axis=(1, 2))), nb)
w_area = torch.rand((Y, X), device=dev).reshape(1, Y, X)
report("weights_broadcast_area_axis_lat_lon", timed(lambda: core.histogram(a, bins=e, weights=w_area, axis=(1, 2))), nb + Y * X * 4)
report("strided_every_2nd_lon", timed(lambda: core.histogram(a[:, :, ::2], bins=e, axis=(1, 2))), nb)
b = torch.empty((T, Y, X), dtype=torch.float32, device=dev).normal_(generator=g)
report("joint_2d_40x40_axis_lat_lon", timed(lambda: core.histogram(a, b, bins=[np.linspace(-4, 4, 41)] * 2, axis=(1, 2))), 2 * nb)
report("density_joint_2d", timed(lambda: core.histogram(a, b, bins=[np.linspace(-4, 4, 41)] * 2, axis=(1, 2), density=True)), 2 * nb)
del b
i64 = torch.randint(0, 10_000, (200_000_000,), device=dev, dtype=torch.int64)
report("int64_samples_int_edges", timed(lambda: core.histogram(i64, bins=np.arange(0, 10_001, 100))), i64.numel() * 8)
report("int64_samples_float_edges", timed(lambda: core.histogram(i64, bins=np.linspace(0, 10_000, 101))), i64.numel() * 8)
i32 = i64.to(torch.int32)
report("int32_samples_int_edges", timed(lambda: core.histogram(i32, bins=np.arange(0, 10_001, 100))), i32.numel() * 4)
u8 = (i64 % 256).to(torch.uint8)
report("uint8_samples_arange257", timed(lambda: core.histogram(u8, bins=np.arange(257))), u8.numel())
h16 = a.reshape(-1)[: 200_000_000].to(torch.float16)
report("float16_samples", timed(lambda: core.histogram(h16, bins=e)), h16.numel() * 2)
c4 = [torch.empty(50_000_000, dtype=torch.float32, device=dev).normal_(generator=g) for _ in range(4)]
report("four_inputs_8x8x8x8", timed(lambda: core.histogram(*c4, bins=[np.linspace(-4, 4, 9)] * 4)), 4 * 50_000_000 * 4)
