"""A/B runs of the partitioned mode (BASELINE C5 shard: 5*10^8 x (2 f64 + f64 weights), 1024 x 1024 bins) in ONE process:
every variant is a set of xhist_plan_set_param overrides; per variant the HIP-event time of the whole step (zeroing +
routing pass + adding-up pass) over `--steps` launches after `--warmup`, and a checksum of the result against the first variant.
Run under `rocprofv3 --kernel-trace --stats` for the per-kernel split (kernel names carry the workgroup size).

  python tools/c5_ab.py [--n 500000000] [--dist normal|uniform|const] [--variants "default;route_spl=4;records48=-1"]
                        [--dtype f64|f32] [--dims 1|2|3] [--rows R] [--bins B] [--unweighted]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

import numpy as np
import torch

from xhistogram_amd import _native, core


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=500_000_000)
    ap.add_argument("--bins", type=int, default=1024)
    ap.add_argument("--dist", default="normal", choices=["normal", "uniform", "const"])
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--unweighted", action="store_true")
    ap.add_argument("--signs", default="one", choices=["one", "both"])
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--wdtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--dims", type=int, default=2)
    ap.add_argument("--edges", default="linspace", choices=["linspace", "jitter", "random"],
                    help="jitter: uneven edges (table lookups); random: sorted uniform draws, end points kept (BASELINE C3)")
    ap.add_argument("--rows", type=int, default=1, help="n is split into this many rows (one histogram per row)")
    ap.add_argument("--variants", default="default;records48=-1;default")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev)
    g.manual_seed(1234)
    rows = args.rows
    cols = args.n // rows
    n = rows * cols
    tdt = torch.float64 if args.dtype == "f64" else torch.float32
    tag = _native.F64 if args.dtype == "f64" else _native.F32

    def sample():
        t = torch.empty(n, dtype=tdt, device=dev)
        if args.dist == "normal":
            t.normal_(generator=g)
        elif args.dist == "uniform":
            t.uniform_(-4.0, 4.0, generator=g)
        else:
            t.fill_(0.123)
        return t

    xs = [sample() for _ in range(args.dims)]
    w = None
    if not args.unweighted:
        w = torch.empty(n, dtype=torch.float64 if args.wdtype == "f64" else torch.float32, device=dev).uniform_(generator=g)
        if args.signs == "both":
            w -= 0.5
    e = np.linspace(-4.0, 4.0, args.bins + 1)
    if args.edges == "jitter":
        e = np.sort(e + np.random.default_rng(5).uniform(-0.45, 0.45, e.size) * (e[1] - e[0]))
    if args.edges == "random":
        e = np.sort(np.random.default_rng(1).uniform(-4.0, 4.0, args.bins + 1))
        e[0], e[-1] = -4.0, 4.0
    edges = [e] * args.dims
    out = torch.zeros((rows,) + (args.bins,) * args.dims, dtype=torch.float64 if w is not None else torch.int64, device=dev)
    xv = [_native.make_view(a.data_ptr(), tag, cols, 1) for a in xs]
    wv = _native.make_view(w.data_ptr(), _native.F64 if args.wdtype == "f64" else _native.F32, cols, 1) if w is not None else None
    stream = torch.cuda.current_stream(dev).cuda_stream
    ref = None
    for spec in args.variants.split(";"):
        plan = _native.Plan(edges, _native.CMP_F64, 0)
        kv = {}
        for item in spec.split(","):
            item = item.strip()
            if not item or item == "default":
                continue
            k, _, v = item.partition("=")
            plan.set_param(k, int(v))
            kv[k] = int(v)
        run = plan.bind(xv, wv, rows, cols, out.data_ptr(), w is not None, _native.MEM_DEVICE, False, stream)
        for _ in range(args.warmup):
            run()
        torch.cuda.synchronize(dev)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        for a, b in evs:
            a.record()
            run()
            b.record()
        torch.cuda.synchronize(dev)
        ms = sorted(a.elapsed_time(b) for a, b in evs)
        res = out.clone()
        if ref is None:
            ref = res
        if res.dtype == torch.int64:
            ok = bool(torch.equal(res, ref))
        else:
            ok = bool(torch.allclose(res, ref, rtol=1e-9, atol=0.0))
        bps = args.dims * (8 if args.dtype == "f64" else 4) + ((8 if args.wdtype == "f64" else 4) if w is not None else 0)
        print(json.dumps({"variant": spec, "dist": args.dist, "n": n, "ms_median": ms[len(ms) // 2], "ms_min": ms[0], "ms_max": ms[-1],
                          "frac_of_8TBs": n * bps / (ms[len(ms) // 2] * 1e-3) / 8e12, "matches_first": ok, "sum": float(res.sum()),
                          "desc": plan.describe()}), flush=True)
        plan.close()


if __name__ == "__main__":
    main()
