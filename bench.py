#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X-native xhistogram hot path.

Workload (BASELINE.json configs[1], "C2"): 1-D histogram of 10^9 float64 samples with float64
weights into 100 uniform bins, per GPU.  A *step* is one pass of the fused hot path over the
resident batch (output memset + histogram kernel, and for N > 1 the RCCL all-reduce of the
[100] float64 partial that replaces the reference's dask `.sum(drop_axes)`, core.py:439).
Inputs are generated on the device before the timed region (data = synthetic N(0,1) samples,
U[0,1) weights).  N GPUs = N processes (torch.distributed/RCCL), each with its own 10^9-sample
shard (weak scaling); `value` is samples/s of the whole job.

Also reported on the same JSON line:
  roofline     achieved algorithmic GB/s of the histogram kernel (16 B/sample x 10^9 samples /
               mean kernel duration, HIP events recorded by the library on the launch stream
               around exactly the kernel, every timed step) against the 8 TB/s HBM peak
  cpu_baseline the numpy restatement of the reference path (oracle/, verified against the
               reference's golden vectors) timed on this box's host cores on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--samples", type=int, default=1_000_000_000, help="samples per GPU")
    ap.add_argument("--bins", type=int, default=100)
    ap.add_argument("--unweighted", action="store_true", help="8 B/sample variant (not the headline)")
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c3", "c4", "c5"],
                    help="BASELINE.json config (default c2 = the headline the driver measures); the others print the "
                         "same JSON line for their shape so every row of DESIGN.md's table can be reproduced")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=400_000_000)
    return ap.parse_args()


def cpu_baseline(xs_host, w_host, edges):
    """oracle (numpy searchsorted + bincount, per-chunk + sum like the reference's dask-threaded
    path) on all host cores; bounded sample of the same workload.  xs_host: list of 1-D arrays"""
    from oracle import oracle_np as onp

    threads = min(os.cpu_count() or 1, 32)
    chunk = 2_500_000
    onp.chunked_threaded([x[:chunk] for x in xs_host], edges, None if w_host is None else w_host[:chunk], chunk, 1)  # warm
    t0 = time.perf_counter()
    onp.chunked_threaded(xs_host, edges, w_host, chunk, threads)
    dt = time.perf_counter() - t0
    n = xs_host[0].shape[0]
    return {
        "value": n / dt,
        "unit": "samples/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d of the same samples (%d input array%s%s), %d-sample chunks on %d threads, %.2f s"
        % (n, len(xs_host), "s" if len(xs_host) > 1 else "", "" if w_host is None else " + weights", chunk, threads, dt),
    }


def build_workload(cfg, args, torch, dev, rank):
    """synthetic inputs of one BASELINE.json config, resident on `dev`"""
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    f64, f32 = torch.float64, torch.float32

    def nonuniform(k, seed):
        e = np.sort(np.random.default_rng(seed).uniform(-4, 4, k))
        e[0], e[-1] = -4.0, 4.0
        return e

    if cfg in ("c1", "c2"):
        n = 1_000_000 if cfg == "c1" else args.samples
        weighted = cfg == "c2" and not args.unweighted
        x = torch.empty(n, dtype=f64, device=dev).normal_(generator=g)
        w = torch.empty(n, dtype=f64, device=dev).uniform_(generator=g) if weighted else None
        return dict(
            arrays=[x], weights=w, edges=[np.linspace(-4.0, 4.0, args.bins + 1)], rows=1, cols=n, reduce="allreduce",
            metric="samples/s binned (f64), 1D %d-bin %s elems per GPU" % (args.bins, "10^6" if cfg == "c1" else "10^9") + (" + f64 weights" if weighted else ""),
            workload="%s: 1-D histogram, %d f64 samples per GPU, %d uniform bins on [-4,4], %s" % (cfg.upper(), n, args.bins, "f64 weights" if weighted else "unweighted"),
            dtype="f64", data="synthetic (N(0,1) samples%s, generated on device)" % (", U[0,1) weights" if weighted else ""))
    if cfg == "c3":
        n = args.samples
        x = torch.empty(n, dtype=f64, device=dev).normal_(generator=g)
        y = torch.empty(n, dtype=f64, device=dev).normal_(generator=g)
        return dict(
            arrays=[x, y], weights=None, edges=[nonuniform(257, 1), nonuniform(257, 2)], rows=1, cols=n, reduce="allreduce",
            metric="samples/s binned (2 x f64), 2D 256x256 non-uniform bins, 10^9 samples per GPU",
            workload="C3: 2-D joint histogram, two %d-sample f64 arrays per GPU, 256x256 non-uniform edges, unweighted" % n,
            dtype="f64", data="synthetic (two independent N(0,1) arrays, generated on device; edges = sorted U(-4,4), ends forced to +-4)")
    if cfg == "c4":
        rows, cols = 456, 720 * 1440  # 3650 time steps over 8 GPUs
        x = torch.empty((rows, cols), dtype=f32, device=dev).normal_(generator=g)
        return dict(
            arrays=[x], weights=None, edges=[np.linspace(-4.0, 4.0, 51)], rows=rows, cols=cols, reduce="none",
            metric="samples/s binned (f32), (time,lat,lon) histogram over lat,lon, 50 bins, 456 time steps per GPU",
            workload="C4: (456, 720, 1440) f32 per GPU (= 3650 time steps over 8 GPUs), dim=[lat, lon], 50 uniform bins; ranks own disjoint time rows",
            dtype="f32", data="synthetic (N(0,1) f32, generated on device)")
    n = min(args.samples, 500_000_000)  # c5: 4e9 samples over 8 GPUs
    x = torch.empty(n, dtype=f64, device=dev).normal_(generator=g)
    y = torch.empty(n, dtype=f64, device=dev).normal_(generator=g)
    w = torch.empty(n, dtype=f64, device=dev).uniform_(generator=g)
    return dict(
        arrays=[x, y], weights=w, edges=[np.linspace(-4.0, 4.0, 1025)] * 2, rows=1, cols=n, reduce="allreduce",
        density=True,
        metric="samples/s binned (2 x f64 + f64 weights), 2D weighted density, 1024x1024 bins, 5*10^8 samples per GPU",
        workload="C5: 2-D weighted density histogram, %d samples per GPU (4*10^9 over 8), 1024x1024 uniform bins (beyond LDS: "
                 "partitioned multi-pass); density epilogue (core.py:444-462) on the reduced result of every step" % n,
        dtype="f64", data="synthetic (two N(0,1) arrays + U[0,1) weights, generated on device)")


def main():
    args = parse()
    # stdout carries exactly ONE line (the JSON result of rank 0): whatever libraries print there — RCCL
    # writes a five-line version banner to stdout when the communicator is created — goes to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    from xhistogram_amd import _native, core

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # under torch.distributed.run (any N)
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or launched
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    _native.require_device(local)

    wl = build_workload(args.config, args, torch, dev, rank)
    arrays, w, edges = wl["arrays"], wl["weights"], wl["edges"]
    weighted = w is not None
    n_rows, n_cols = wl["rows"], wl["cols"]
    n = n_rows * n_cols  # samples per GPU and step
    tag = {torch.float64: _native.F64, torch.float32: _native.F32}
    plan = core._get_plan(edges, _native.CMP_F64, local)
    # two result buffers: the RCCL all-reduce of step k runs while step k+1's kernel streams
    out_shape = (n_rows,) + plan.bins_shape
    outs = [torch.zeros(out_shape, dtype=torch.float64 if weighted else torch.int64, device=dev) for _ in range(2)]
    pending = [None, None]
    stream = torch.cuda.current_stream(dev).cuda_stream
    xv = [_native.make_view(a.data_ptr(), tag[a.dtype], n_cols, 1) for a in arrays]
    wv = _native.make_view(w.data_ptr(), tag[w.dtype], n_cols, 1) if weighted else None
    reduce_partials = use_dist and wl["reduce"] == "allreduce"
    density = bool(wl.get("density"))
    dens = [None]
    counter = [0]

    def finish(k):
        """what follows a step's kernel once its all-reduce is done: the density epilogue (C5)"""
        if pending[k] is not None:
            pending[k].wait()
            pending[k] = None
            if density:
                dens[0] = core._density(outs[k], edges, len(edges))

    def step():
        k = counter[0] & 1
        counter[0] += 1
        finish(k)  # the reduction that last used this buffer must be done
        out = outs[k]
        plan.execute(xv, wv, n_rows, n_cols, out.data_ptr(), weighted, _native.MEM_DEVICE, accumulate=False, stream=stream)
        if reduce_partials:
            pending[k] = dist.all_reduce(out, op=dist.ReduceOp.SUM, async_op=True)
        elif density:
            dens[0] = core._density(out, edges, len(edges))

    def fence():
        for k in (0, 1):
            finish(k)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    plan.set_param("profile", min(args.steps, 4096))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    out = outs[(counter[0] - 1) & 1]
    kernel_ms = plan.profile_read()
    plan.set_param("profile", 0)
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # sanity: the result of the last step is a real histogram of this rank's shard (all ranks summed)
    total = float(out.sum().item())
    assert total > 0

    if rank == 0:
        bytes_per_sample = sum(a.element_size() for a in arrays) + (w.element_size() if weighted else 0)
        k_ms = float(np.mean(kernel_ms))
        achieved = bytes_per_sample * n / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if args.config == "c2" and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch_weighted" if weighted else "hbm_bytes_per_launch_unweighted")
            except Exception:
                traffic = None
        line = {
            "metric": wl["metric"],
            "value": world * n * args.steps / dt,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": wl["dtype"],
            "data": wl["data"],
            "config": {
                "workload": wl["workload"],
                "samples_per_gpu": n,
                "bins": [int(b) for b in plan.bins_shape],
                "weighted": weighted,
                "kernel": plan.describe(),
                "parallelism": ("sample-axis shards, one per GPU" if wl["reduce"] == "allreduce" else "kept-axis (time) shards, one per GPU, disjoint output rows")
                + ("; all-reduce(sum) of the partial histogram over RCCL each step, overlapped with the next step's kernel" if reduce_partials else ""),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel_ms_mean": k_ms,
                "kernel_ms_min": float(np.min(kernel_ms)),
                "kernel_launches_timed": len(kernel_ms),
                "algorithmic_bytes_per_launch": bytes_per_sample * n,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            m = min(args.cpu_sample // max(1, len(arrays)), n)
            flat = [a.reshape(-1)[:m].double().cpu().numpy() if args.config != "c4" else a.reshape(-1)[:m].cpu().numpy() for a in arrays]
            line["cpu_baseline"] = cpu_baseline(flat, w.reshape(-1)[:m].cpu().numpy() if weighted else None, edges)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
