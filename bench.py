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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=400_000_000)
    return ap.parse_args()


def cpu_baseline(x_host, w_host, edges):
    """oracle (numpy searchsorted + bincount, per-chunk + sum like the reference's dask-threaded
    path) on all host cores; bounded sample of the same workload"""
    from oracle import oracle_np as onp

    threads = min(os.cpu_count() or 1, 32)
    chunk = 2_500_000
    onp.chunked_threaded([x_host[:chunk]], [edges], None if w_host is None else w_host[:chunk], chunk, 1)  # warm
    t0 = time.perf_counter()
    onp.chunked_threaded([x_host], [edges], w_host, chunk, threads)
    dt = time.perf_counter() - t0
    return {
        "value": x_host.shape[0] / dt,
        "unit": "samples/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d of the same N(0,1) f64 samples%s, %d-sample chunks on %d threads, %.2f s"
        % (x_host.shape[0], "" if w_host is None else " + f64 weights", chunk, threads, dt),
    }


def main():
    args = parse()
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

    n = args.samples
    weighted = not args.unweighted
    edges = np.linspace(-4.0, 4.0, args.bins + 1)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    x = torch.empty(n, dtype=torch.float64, device=dev).normal_(generator=g)
    w = torch.empty(n, dtype=torch.float64, device=dev).uniform_(generator=g) if weighted else None

    plan = core._get_plan([edges], _native.CMP_F64, local)
    # two result buffers: the RCCL all-reduce of step k runs while step k+1's kernel streams
    outs = [torch.zeros(args.bins, dtype=torch.float64 if weighted else torch.int64, device=dev) for _ in range(2)]
    pending = [None, None]
    stream = torch.cuda.current_stream(dev).cuda_stream
    xv = [_native.make_view(x.data_ptr(), _native.F64, n, 1)]
    wv = _native.make_view(w.data_ptr(), _native.F64, n, 1) if weighted else None
    counter = [0]

    def step():
        k = counter[0] & 1
        counter[0] += 1
        if pending[k] is not None:  # the reduction that last used this buffer must be done
            pending[k].wait()
            pending[k] = None
        out = outs[k]
        plan.execute(xv, wv, 1, n, out.data_ptr(), weighted, _native.MEM_DEVICE, accumulate=False, stream=stream)
        if use_dist:
            pending[k] = dist.all_reduce(out, op=dist.ReduceOp.SUM, async_op=True)

    def fence():
        for k in (0, 1):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
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
        bytes_per_sample = 16 if weighted else 8
        k_ms = float(np.mean(kernel_ms))
        achieved = bytes_per_sample * n / (k_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch_weighted" if weighted else "hbm_bytes_per_launch_unweighted")
            except Exception:
                traffic = None
        line = {
            "metric": "samples/s binned (f64), 1D 100-bin 10^9 elems per GPU" + (" + f64 weights" if weighted else ""),
            "value": world * n * args.steps / dt,
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic (N(0,1) samples, U[0,1) weights, generated on device)",
            "config": {
                "workload": "C2: 1-D histogram, %d f64 samples per GPU, %d uniform bins on [-4,4], %s" % (n, args.bins, "f64 weights" if weighted else "unweighted"),
                "samples_per_gpu": n,
                "bins": args.bins,
                "weighted": weighted,
                "kernel": plan.describe(),
                "parallelism": "sample-axis shards, one per GPU" + ("; all-reduce(sum) of the [bins] partial over RCCL each step, overlapped with the next step's kernel" if use_dist else ""),
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
            m = min(args.cpu_sample, n)
            line["cpu_baseline"] = cpu_baseline(x[:m].cpu().numpy(), w[:m].cpu().numpy() if weighted else None, edges)
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
