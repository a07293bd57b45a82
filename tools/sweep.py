#!/usr/bin/env python
"""Tuning sweep of the histogram kernels on one MI355X (development tool, not the bench).
Prints one JSON line per variant: kernel ms (HIP events from the library), algorithmic GB/s."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(plan, views, wview, n_rows, n_cols, out, weighted, stream, reps, _native):
    import torch

    # warm up past the clock excursion of the first ~13 ms of a burst: launches for >= 25 ms, at least two
    import time

    t0 = time.perf_counter()
    k = 0
    while k < 2 or (time.perf_counter() - t0 < 0.025 and k < 4000):
        plan.execute(views, wview, n_rows, n_cols, out.data_ptr(), weighted, _native.MEM_DEVICE, stream=stream)
        k += 1
        if k % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    plan.set_param("profile", reps)
    for _ in range(reps):
        plan.execute(views, wview, n_rows, n_cols, out.data_ptr(), weighted, _native.MEM_DEVICE, stream=stream)
    torch.cuda.synchronize()
    ms = plan.profile_read()
    plan.set_param("profile", 0)
    return float(np.median(ms)), float(np.min(ms))


def torch_ref(fn, reps=5):
    import torch

    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1_000_000_000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--what", default="c2w,c2u,ref,c3,c4,c5")
    args = ap.parse_args()
    import torch

    from xhistogram_amd import _native, core

    n = args.n
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    x = torch.empty(n, dtype=torch.float64, device=dev).normal_(generator=g)
    w = torch.empty(n, dtype=torch.float64, device=dev).uniform_(generator=g)
    what = args.what.split(",")

    def emit(**kw):
        print(json.dumps(kw), flush=True)

    if "ref" in what:
        ms = torch_ref(lambda: x.sum())
        emit(case="ref_torch_sum_f64", ms=ms, gbs=8 * n / ms / 1e6)
        y = torch.empty_like(x)
        ms = torch_ref(lambda: y.copy_(x))
        emit(case="ref_torch_copy_f64", ms=ms, gbs=16 * n / ms / 1e6)
        del y
        ms = torch_ref(lambda: torch.dot(x, w))
        emit(case="ref_torch_dot_f64", ms=ms, gbs=16 * n / ms / 1e6)

    edges = np.linspace(-4, 4, 101)
    plan = core._get_plan([edges], _native.CMP_F64, 0)
    xv = [_native.make_view(x.data_ptr(), _native.F64, n, 1)]
    wv = _native.make_view(w.data_ptr(), _native.F64, n, 1)
    for tag, weighted, bps in (("c2w", True, 16), ("c2u", False, 8)):
        if tag not in what:
            continue
        out = torch.zeros(100, dtype=torch.float64 if weighted else torch.int64, device=dev)
        for block in (256, 512, 1024):
            for grid in (0, 256, 512, 1024, 2048):
                for copies in (0,):
                    plan.set_param("block_threads", block)
                    plan.set_param("grid_blocks", grid)
                    plan.set_param("lds_copies", copies)
                    med, mn = timed(plan, xv, wv if weighted else None, 1, n, out, weighted, stream, args.reps, _native)
                    emit(case=tag, block=block, grid=grid, copies=copies, ms=med, ms_min=mn, gbs=bps * n / med / 1e6,
                         frac=bps * n / med / 1e6 / 8000, desc=plan.describe())
        for k in ("block_threads", "grid_blocks", "lds_copies"):
            plan.set_param(k, 0)
        # data-distribution sensitivity: uniform samples over the bins (low contention)
        if tag == "c2w":
            xu = torch.empty(n, dtype=torch.float64, device=dev).uniform_(-4, 4, generator=g)
            xuv = [_native.make_view(xu.data_ptr(), _native.F64, n, 1)]
            med, mn = timed(plan, xuv, wv, 1, n, out, True, stream, args.reps, _native)
            emit(case="c2w_uniform_samples", ms=med, gbs=16 * n / med / 1e6)
            del xu
            xc = torch.zeros(n, dtype=torch.float64, device=dev)
            xcv = [_native.make_view(xc.data_ptr(), _native.F64, n, 1)]
            med, mn = timed(plan, xcv, wv, 1, n, out, True, stream, args.reps, _native)
            emit(case="c2w_constant_samples_one_bin", ms=med, gbs=16 * n / med / 1e6)
            del xc

    if "host" in what:
        import time
        m = 200_000_000
        xh_, wh_ = x[:m].cpu().numpy(), w[:m].cpu().numpy()
        for weighted in (True, False):
            core.histogram(xh_[:1000], bins=edges)
            t0 = time.perf_counter()
            core.histogram(xh_, bins=edges, weights=wh_ if weighted else None)
            dt = time.perf_counter() - t0
            emit(case="host_numpy_route_pcie_inclusive", weighted=weighted, samples=m, s=dt, samples_per_s=m / dt,
                 gbs=(16 if weighted else 8) * m / dt / 1e9)
        del xh_, wh_

    if "c3" in what:
        rng = np.random.default_rng(1)
        def nu(k):
            e = np.sort(rng.uniform(-4, 4, k)); e[0], e[-1] = -4.0, 4.0; return e
        ea, eb = nu(257), nu(257)
        p3 = core._get_plan([ea, eb], _native.CMP_F64, 0)
        out = torch.zeros(256 * 256, dtype=torch.int64, device=dev)
        y = w * 8.0 - 4.0
        v3 = [_native.make_view(x.data_ptr(), _native.F64, n, 1), _native.make_view(y.data_ptr(), _native.F64, n, 1)]
        for block in (0, 512, 768, 1024):
            p3.set_param("block_threads", block)
            med, mn = timed(p3, v3, None, 1, n, out, False, stream, 3, _native)
            emit(case="c3_2d_256x256_nonuniform", block=block, ms=med, gbs=16 * n / med / 1e6, frac=16 * n / med / 1e6 / 8000, desc=p3.describe())
        p3.set_param("block_threads", 0)
        # uniform 64x64 (fits LDS)
        p3b = core._get_plan([np.linspace(-4, 4, 65)] * 2, _native.CMP_F64, 0)
        out = torch.zeros(64 * 64, dtype=torch.int64, device=dev)
        med, mn = timed(p3b, v3, None, 1, n, out, False, stream, 3, _native)
        emit(case="2d_64x64_uniform", ms=med, gbs=16 * n / med / 1e6, desc=p3b.describe())
        del y

    if "c5" in what:
        p5 = core._get_plan([np.linspace(-4, 4, 1025)] * 2, _native.CMP_F64, 0)
        m = min(n, 500_000_000)
        out = torch.zeros(1024 * 1024, dtype=torch.float64, device=dev)
        y = w[:m] * 8.0 - 4.0
        v5 = [_native.make_view(x.data_ptr(), _native.F64, m, 1), _native.make_view(y.data_ptr(), _native.F64, m, 1)]
        wv5 = _native.make_view(w.data_ptr(), _native.F64, m, 1)
        med, mn = timed(p5, v5, wv5, 1, m, out, True, stream, 3, _native)
        emit(case="c5_2d_1024x1024_weighted_5e8", ms=med, gbs=24 * m / med / 1e6, frac=24 * m / med / 1e6 / 8000, desc=p5.describe())
        del y

    if "c4" in what:
        rows, cols = 456, 720 * 1440
        xf = torch.empty(rows * cols, dtype=torch.float32, device=dev).normal_(generator=g)
        p4 = core._get_plan([np.linspace(-4, 4, 51)], _native.CMP_F64, 0)
        out = torch.zeros(rows * 50, dtype=torch.int64, device=dev)
        v4 = [_native.make_view(xf.data_ptr(), _native.F32, cols, 1)]
        for block in (128, 256, 512):
            for grid in (0, 4096, 8192, 16384, 32768, 65536):
                p4.set_param("block_threads", block)
                p4.set_param("grid_blocks", grid)
                med, mn = timed(p4, v4, None, rows, cols, out, False, stream, args.reps, _native)
                emit(case="c4_f32_rows456x1036800_50bins", block=block, grid=grid, ms=med, gbs=4 * rows * cols / med / 1e6,
                     frac=4 * rows * cols / med / 1e6 / 8000, desc=p4.describe())
        # same data as one flat row
        for k in ("block_threads", "grid_blocks"):
            p4.set_param(k, 0)
        out = torch.zeros(50, dtype=torch.int64, device=dev)
        v4 = [_native.make_view(xf.data_ptr(), _native.F32, rows * cols, 1)]
        med, mn = timed(p4, v4, None, 1, rows * cols, out, False, stream, args.reps, _native)
        emit(case="f32_flat_50bins", ms=med, gbs=4 * rows * cols / med / 1e6, frac=4 * rows * cols / med / 1e6 / 8000, desc=p4.describe())


if __name__ == "__main__":
    main()
