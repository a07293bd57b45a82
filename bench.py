#!/usr/bin/env python
"""bench.py — headline benchmark of the MI355X-native xhistogram hot path.

Workload (BASELINE.json configs[1], "C2"): 1-D histogram of 10^9 float64 samples with float64
weights into 100 uniform bins, per GPU.  A *step* is one pass of the fused hot path over the
resident batch (output memset + histogram kernel, and for N > 1 the RCCL all-reduce of the
[100] float64 partial that replaces the reference's dask `.sum(drop_axes)`, core.py:439).
Inputs are generated on the device before the timed region (data = synthetic N(0,1) samples,
U[0,1) weights).  N GPUs = N processes (torch.distributed/RCCL), each with its own 10^9-sample
shard (weak scaling: the line's `value`); with N > 1 the same run also times the STRONG leg
(10^9 samples in total, 10^9 / N per GPU: SURVEY.md 8e case 1) and reports it under `"strong"`
(`--scaling strong` makes it the line's `value` instead).  `value` is samples/s of the whole job.

Launch: `python bench.py --gpus N` spawns its own N ranks (re-executes itself under
torch.distributed.run on 127.0.0.1) when it is not already running under a launcher; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it is one of the ranks.

Also reported on the same JSON line:
  roofline     achieved algorithmic GB/s of the histogram kernel (16 B/sample x 10^9 samples /
               mean kernel duration, HIP events recorded by the library on the launch stream
               around exactly the kernel, every timed step) against the 8 TB/s HBM peak
  cpu_baseline the numpy restatement of the reference path (oracle/, verified against the
               reference's golden vectors) timed on this box's host cores on a bounded sample:
               one thread, and a thread pool over sample chunks on every host core
               (os.cpu_count() and the CPU model are reported)
  verified     every timed leg's LAST output is compared, in this process and at full size, with an independent
               restatement in torch ops only (bucketize(right=True) + the last-edge rule + joint index + bincount:
               core.py:170-173, 178-181, 81) — int64 counts identical, float64 sums within 1e-6 relative (north_star);
               a mismatch is reported in the line AND ends the run with exit status 1
  distributions  (default N = 1 run) the headline's two legs again on uniform[-4,4) samples, on samples that all fall
               into ONE bin and on samples of which 90 % are out of range (the LDS-atomic contention legs)
  first_call_ms  plan creation + module load + first launch of the headline kernel, cold
  host_inputs  (default N = 1 run) the reference's REAL call shape — numpy arrays in, numpy out (core.py:442 through
               XHIST_MEM_HOST staging) — on pageable and on pinned host memory, next to the host-to-device copy rate of the
               same bytes measured on this box, and the same samples as 16 host chunks summed (what the dask graph's tasks do)
  ranks        who took part: backend, world size as the process group reports it, (rank, local rank, device, PCI bus id,
               uuid) of every rank gathered over that backend, RCCL version
  summary      LAST key of the line: roofline fraction of every config leg + all_verified, so that a truncated tail still says it
"""
import argparse
import json
import os
import socket
import subprocess
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
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 20; 200 for the sub-millisecond steps of --config c4)")
    ap.add_argument("--warmup", type=int, default=None, help="untimed steps before them (default 10; 50 for --config c4)")
    ap.add_argument("--samples", type=int, default=1_000_000_000, help="samples per GPU")
    ap.add_argument("--bins", type=int, default=100)
    ap.add_argument("--unweighted", action="store_true", help="8 B/sample variant (not the headline)")
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c3", "c4", "c5"],
                    help="BASELINE.json config (default c2 = the headline the driver measures); the others print the "
                         "same JSON line for their shape so every row of DESIGN.md's table can be reproduced")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="which leg the line's `value` reports: weak = --samples per GPU (default), strong = --samples in total, "
                         "split over the GPUs; with N > 1 the other leg is measured too and reported under its own key")
    ap.add_argument("--full", action="store_true",
                    help="c4 / c5 at the FULL size of BASELINE.json on ONE GPU: (3650, 720, 1440) f32 = 15.1 GB; 4*10^9 samples x 24 B = 96 GB")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="xhist_plan_set_param override for A/B runs, e.g. --tune fused=-1 (not for the headline)")
    ap.add_argument("--selftest", action="store_true",
                    help="control-flow self-test of the multi-rank harness WITHOUT kernels or GPUs: gloo on CPU, the histogram launch "
                         "replaced by a no-op (nothing is measured; `metric` says so) — exercises rank spawn, both scaling legs, the "
                         "gathers and the JSON line, which no 1-GPU box can")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="the default N = 1 run also times BASELINE configs C3, C4 and C5 in short legs (`\"configs\"` of the line, <= 45 s); this turns that off")
    ap.add_argument("--profiler-pass", action="store_true",
                    help="only the main leg's launches: no 8 B/sample leg riding along (c2), no cold burst (c4) — for rocprofv3 passes "
                         "that attribute counters and launch counts to ONE kernel")
    ap.add_argument("--no-verify", action="store_true", help="skip the in-process torch cross-check of every leg's output (profiler passes)")
    ap.add_argument("--distributions", action="store_true",
                    help="c2: also time (and verify) the headline's legs on uniform, one-bin and 90 %%-out-of-range samples; on by default in the default N = 1 run")
    ap.add_argument("--dist", default="normal", choices=["normal", "uniform"],
                    help="c1 / c2 sample distribution for profiler passes and A/B runs: N(0,1) (the headline) or uniform[-4,4) (every bin equally likely)")
    ap.add_argument("--cpu-sample", type=int, default=400_000_000)
    args = ap.parse_args()
    # a C4 shard step is one 0.3 ms kernel: the first tens of launches after an idle GPU run 5-8 % slower (0.309 ms over steps
    # 6..25, 0.290 ms over steps 51..250 and beyond), so its default run is longer; every other config's kernel takes milliseconds
    # and does not care (profiles/r02_z_warmup.txt)
    short_steps = args.config == "c4" and not args.full
    if args.steps is None:
        args.steps = 200 if short_steps else 20
    if args.warmup is None:
        # (millisecond kernels: the first ~6 launches after the idle gap of data generation run up to 14 % slow — the clock
        # excursion over the first ~13 ms of a burst, profiles/r04_z_c3_launch_series.txt — so 10 untimed launches, not 3)
        args.warmup = 50 if short_steps else 10
    return args


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: run N ranks of this script under torch.distributed.run
    (rendezvous on 127.0.0.1, a free port); rank 0's JSON line is the only thing they write to stdout"""
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: spawning %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    return subprocess.run(cmd, env=env).returncode


def csrc_sha16():
    """hash of the native sources the library was built from: profiles/traffic.json carries the same figure for the code state its
    counters were taken on (tools/pmc_traffic.py), and `roofline.traffic` is refused when they differ (the GPU box has no .git)"""
    import hashlib

    h = hashlib.sha256()
    d = os.path.join(ROOT, "xhistogram_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h", ".sh")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def rank_identity(torch, dist, dev, use_dist, selftest=False):
    """the `"ranks"` block: gathered over the process group itself, so N entries with N distinct devices are the backend's own
    proof that N ranks on N GPUs took part (VERDICT r5 "next" #4)"""
    me = {"rank": int(os.environ.get("RANK", "0")), "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "host": socket.gethostname(), "pid": os.getpid()}
    if not selftest and dev.type == "cuda":
        pr = torch.cuda.get_device_properties(dev)
        me.update(device=pr.name, arch=getattr(pr, "gcnArchName", None), pci_bus_id="%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
                  uuid=str(getattr(pr, "uuid", "")), hbm_bytes=int(pr.total_memory), compute_units=int(pr.multi_processor_count))
    else:
        me.update(device="cpu (selftest: no GPU, no kernel)")
    block = {"backend": dist.get_backend() if use_dist else "none (single process, no process group)",
             "world_size_seen": dist.get_world_size() if use_dist else 1}
    if use_dist:
        got = [None] * dist.get_world_size()
        dist.all_gather_object(got, me)
        block["devices"] = got
    else:
        block["devices"] = [me]
    ids = [d.get("uuid") or d.get("pci_bus_id") or "%s/%s" % (d["host"], d["pid"]) for d in block["devices"]]
    block["distinct_devices"] = len(set(ids))
    try:
        v = torch.cuda.nccl.version()
        block["rccl_version"] = ".".join(str(i) for i in v) if isinstance(v, tuple) else str(v)
    except Exception:  # noqa: BLE001
        block["rccl_version"] = None
    return block


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(xs_host, w_host, edges, one_chunk=False):
    """oracle (numpy searchsorted + bincount) on a bounded sample of the same workload, two legs as BASELINE.md
    section 4 asks: (1) ONE thread over the sample prefix; (2) a thread pool over contiguous sample chunks, partials
    summed — what the reference's dask-threaded path does (core.py:429-439) — on every host core.  `value` is leg 2
    (leg 1 for C1, whose 10^6 samples are a single chunk).  xs_host: list of 1-D arrays"""
    from oracle import oracle_np as onp

    n = xs_host[0].shape[0]
    ncpu = os.cpu_count() or 1
    nin = "%d input array%s%s" % (len(xs_host), "s" if len(xs_host) > 1 else "", "" if w_host is None else " + weights")
    warm = min(n, 1_000_000)
    onp.chunked_threaded([x[:warm] for x in xs_host], edges, None if w_host is None else w_host[:warm], warm, 1)
    # leg 1: one thread, one block (block_size=None semantics), on a prefix that takes a few seconds
    n1 = min(n, 20_000_000)
    t0 = time.perf_counter()
    onp.chunked_threaded([x[:n1] for x in xs_host], edges, None if w_host is None else w_host[:n1], n1, 1)
    dt1 = time.perf_counter() - t0
    single = {"value": n1 / dt1, "unit": "samples/s", "cores": 1,
              "sample": "%d of the same samples (%s), one block on one thread, %.2f s" % (n1, nin, dt1)}
    base = {"kind": "port", "os_cpu_count": ncpu, "cpu_model": cpu_model(), "single_thread": single}
    if one_chunk or n <= n1:
        return dict(single, **base)
    # leg 2: every host core; chunks of <= 10^7 samples, at least four per thread
    threads = ncpu
    chunk = int(min(10_000_000, max(250_000, n // (4 * threads))))
    t0 = time.perf_counter()
    onp.chunked_threaded(xs_host, edges, w_host, chunk, threads)
    dt = time.perf_counter() - t0
    return dict({
        "value": n / dt,
        "unit": "samples/s",
        "cores": threads,
        "sample": "%d of the same samples (%s), %d-sample chunks on a pool of %d threads (= os.cpu_count()), partial histograms summed, %.2f s"
        % (n, nin, chunk, threads, dt),
    }, **base)


def torch_reference(torch, arrays, w, edges, n_rows, n_cols, weighted, chunk=1 << 27):
    """The histogram of the first n_cols samples of every row, restated with torch ops ONLY — nothing of libxhist_amd.so and
    nothing of oracle/ is involved: bucketize(right=True) is searchsorted(side="right") (core.py:170), samples equal to the
    last edge move into the last bin (core.py:171-173), the joint index is the C-order ravel of the per-input bins
    (core.py:178-181), bincount adds ones / float64 weights (core.py:81), under- and overflow are dropped (core.py:189-192).
    Compares in float64 like numpy's promotion.  Returns [n_rows, prod(bins)] int64 / float64 on the inputs' device."""
    dev = arrays[0].device
    nb = [len(e) - 1 for e in edges]
    n_bins = int(np.prod(nb))
    et = [torch.as_tensor(np.asarray(e, dtype=np.float64), device=dev) for e in edges]
    out = torch.zeros(n_rows * n_bins, dtype=torch.float64 if weighted else torch.int64, device=dev)
    a2 = [a.reshape(n_rows, -1) for a in arrays]
    w2 = w.reshape(n_rows, -1) if weighted else None
    rows_per = max(1, chunk // max(1, n_cols))
    cols_per = n_cols if rows_per > 1 else chunk
    for r0 in range(0, n_rows, rows_per):
        r1 = min(n_rows, r0 + rows_per)
        for c0 in range(0, n_cols, cols_per):
            c1 = min(n_cols, c0 + cols_per)
            flat, ok = None, None
            for d in range(len(a2)):
                x = a2[d][r0:r1, c0:c1].to(torch.float64)
                idx = torch.bucketize(x, et[d], right=True)
                idx = torch.where(x == et[d][-1], idx - 1, idx)
                ok_d = (idx >= 1) & (idx <= nb[d])
                ok = ok_d if ok is None else ok & ok_d
                flat = (idx - 1) if flat is None else flat * nb[d] + (idx - 1)
                del x, idx, ok_d
            if n_rows > 1:
                flat = flat + (torch.arange(r0, r1, device=dev, dtype=torch.int64) * n_bins)[:, None]
            sel = flat[ok]
            if weighted:
                out += torch.bincount(sel, weights=w2[r0:r1, c0:c1][ok].to(torch.float64), minlength=n_rows * n_bins)
            else:
                out += torch.bincount(sel, minlength=n_rows * n_bins)
            del flat, ok, sel
    return out.reshape(n_rows, n_bins)


def host_inputs_leg(torch, core, dev, arrays, w, edges, n_host=200_000_000, chunks=16):
    """The reference's own call shape (core.py:442: numpy arrays in, numpy histogram out) through the XHIST_MEM_HOST boundary:
    the first n_host samples (+ weights) of the headline workload copied to host memory, then core.histogram on them — once from
    PAGEABLE numpy arrays (what every existing xhistogram caller has), once from PINNED ones (torch pin_memory, handed over as
    numpy views) — each against the host-to-device copy rate of the same bytes from the same kind of memory, measured here with
    a plain tensor copy.  Then the same samples as `chunks` host chunks, one histogram call per chunk, partials summed: what the
    tasks of the reference's dask graph do (core.py:403-439; dask itself is not in this image).  Every result is compared with the
    device-resident result of the same samples.  PCIe-inclusive: never the line's `value`."""
    n = min(n_host, arrays[0].numel())
    xs_dev = [a.reshape(-1)[:n] for a in arrays]
    w_dev = None if w is None else w.reshape(-1)[:n]
    nbytes = sum(a.element_size() for a in xs_dev) * n + (0 if w is None else w_dev.element_size() * n)
    want, _ = core.histogram(*xs_dev, bins=edges, weights=w_dev)
    torch.cuda.synchronize(dev)
    want = want.cpu().numpy()
    out = {"samples": n, "bytes_in": nbytes, "unit": "GB/s of input bytes (PCIe-inclusive wall time of the call, numpy in -> numpy out)"}

    def best(fn, reps=3):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            r = fn()
            torch.cuda.synchronize(dev)
            ts.append(time.perf_counter() - t0)
        return min(ts), r

    scratch = [torch.empty_like(a) for a in xs_dev] + ([] if w is None else [torch.empty_like(w_dev)])
    for kind in ("pageable", "pinned"):
        hosts = [a.cpu() for a in xs_dev] + ([] if w is None else [w_dev.cpu()])
        if kind == "pinned":
            hosts = [h.pin_memory() for h in hosts]
        nps = [h.numpy() for h in hosts]

        def copy_only():
            for d, h in zip(scratch, hosts):
                d.copy_(h, non_blocking=True)

        def call():
            return core.histogram(*nps[: len(xs_dev)], bins=edges, weights=None if w is None else nps[-1])[0]

        t_copy, _ = best(copy_only)
        t_call, got = best(call)
        ok = bool(np.allclose(got, want, rtol=VERIFY_RTOL, atol=0)) if w is not None else bool(np.array_equal(got, want))
        out[kind] = {"call_ms": round(t_call * 1e3, 3), "GBps": round(nbytes / t_call / 1e9, 2), "h2d_copy_ms": round(t_copy * 1e3, 3),
                     "h2d_copy_GBps": round(nbytes / t_copy / 1e9, 2), "frac_of_copy_rate": round(t_copy / t_call, 3),
                     "samples_per_s": n / t_call, "verified": ok}
        if kind == "pageable":  # the dask-shaped leg: `chunks` host chunks, one call each (core.py:415-439), partials summed on the host
            cuts = np.linspace(0, n, chunks + 1).astype(np.int64)

            def chunked():
                acc = None
                for c0, c1 in zip(cuts[:-1], cuts[1:]):
                    h = core.histogram(*[a[c0:c1] for a in nps[: len(xs_dev)]], bins=edges, weights=None if w is None else nps[-1][c0:c1])[0]
                    acc = h if acc is None else acc + h
                return acc

            t_ch, got_ch = best(chunked, reps=2)
            okc = bool(np.allclose(got_ch, want, rtol=VERIFY_RTOL, atol=0)) if w is not None else bool(np.array_equal(got_ch, want))
            out["host_chunks"] = {"chunks": chunks, "call_ms": round(t_ch * 1e3, 3), "GBps": round(nbytes / t_ch / 1e9, 2),
                                  "frac_of_copy_rate": round(t_copy / t_ch, 3), "verified": okc,
                                  "what": "%d pageable host chunks, one core.histogram call per chunk on one thread, partial histograms summed on the host (the shape of the reference's dask graph)" % chunks}
        del hosts, nps
    out["verified"] = {"ok": bool(out["pageable"]["verified"] and out["pinned"]["verified"] and out["host_chunks"]["verified"]),
                       "kind": "numpy results of the host calls against the device-resident result of the same samples"}
    return out


VERIFY_RTOL = 1e-6  # north_star: float64 weighted sums / density within 1e-6 relative; int64 counts bit-exact


def compare_with_reference(torch, got, ref):
    """{"ok", "kind", "max_rel", ...}: int64 counts must be identical; float64 sums within VERIFY_RTOL of the reference bin
    by bin (bins the reference leaves empty must be exactly empty)"""
    got = got.reshape(ref.shape)
    if ref.dtype == torch.int64:
        bad = int((got != ref).sum().item())
        return {"ok": bad == 0, "kind": "int64 counts identical", "bins_different": bad, "total": int(ref.sum().item())}
    diff = (got - ref).abs()
    scale = ref.abs()
    rel = torch.where(scale > 0, diff / scale.clamp_min(1e-300), torch.where(diff > 0, torch.full_like(diff, float("inf")), torch.zeros_like(diff)))
    rel = torch.nan_to_num(rel, nan=float("inf"))
    max_rel = float(rel.max().item())
    return {"ok": bool(max_rel <= VERIFY_RTOL), "kind": "float64 sums within %g relative, bin by bin" % VERIFY_RTOL, "max_rel": max_rel,
            "total": float(ref.sum().item())}


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
        x = torch.empty(n, dtype=f64, device=dev)
        x.uniform_(-4.0, 4.0, generator=g) if getattr(args, "dist", "normal") == "uniform" else x.normal_(generator=g)
        w = torch.empty(n, dtype=f64, device=dev).uniform_(generator=g) if weighted else None
        return dict(
            arrays=[x], weights=w, edges=[np.linspace(-4.0, 4.0, args.bins + 1)], rows=1, cols=n, reduce="allreduce",
            metric="samples/s binned (f64), 1D %d-bin %s elems per GPU" % (args.bins, "10^6" if cfg == "c1" else "10^9") + (" + f64 weights" if weighted else ""),
            workload="%s: 1-D histogram, %d f64 samples per GPU, %d uniform bins on [-4,4], %s" % (cfg.upper(), n, args.bins, "f64 weights" if weighted else "unweighted"),
            dtype="f64", data="synthetic (%s samples%s, generated on device)" % ("uniform[-4,4)" if getattr(args, "dist", "normal") == "uniform" else "N(0,1)", ", U[0,1) weights" if weighted else ""))
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
        rows, cols = (3650 if args.full else 456), 720 * 1440  # 3650 time steps: all on one GPU (--full), or 456 = 1/8 per GPU
        x = torch.empty((rows, cols), dtype=f32, device=dev).normal_(generator=g)
        return dict(
            arrays=[x], weights=None, edges=[np.linspace(-4.0, 4.0, 51)], rows=rows, cols=cols, reduce="none",
            metric="samples/s binned (f32), (time,lat,lon) histogram over lat,lon, 50 bins, %d time steps per GPU" % rows,
            workload=("C4 at full size on one GPU: (3650, 720, 1440) f32 = 15.1 GB, dim=[lat, lon], 50 uniform bins" if args.full else
                      "C4: (456, 720, 1440) f32 per GPU (= 3650 time steps over 8 GPUs), dim=[lat, lon], 50 uniform bins; ranks own disjoint time rows"),
            dtype="f32", data="synthetic (N(0,1) f32, generated on device)")
    n = 4_000_000_000 if args.full else min(args.samples, 500_000_000)  # c5: 4e9 samples, all on one GPU (--full: 96 GB) or over 8
    x = torch.empty(n, dtype=f64, device=dev).normal_(generator=g)
    y = torch.empty(n, dtype=f64, device=dev).normal_(generator=g)
    w = torch.empty(n, dtype=f64, device=dev).uniform_(generator=g)
    return dict(
        arrays=[x, y], weights=w, edges=[np.linspace(-4.0, 4.0, 1025)] * 2, rows=1, cols=n, reduce="allreduce",
        density=True, wants_whole_gpu=True,
        metric="samples/s binned (2 x f64 + f64 weights), 2D weighted density, 1024x1024 bins, %s samples per GPU" % ("4*10^9" if args.full else "5*10^8"),
        workload="C5: 2-D weighted density histogram, %d samples per GPU (4*10^9 over 8), 1024x1024 uniform bins (beyond LDS: "
                 "the exchange mode where the samples are concentrated, else the partitioned passes, DESIGN 4.2 / 4.2b); density epilogue (core.py:444-462) on the reduced result of every step" % n,
        dtype="f64", data="synthetic (two N(0,1) arrays + U[0,1) weights, generated on device)")


def main():
    args = parse()
    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(self_spawn(args))
    # stdout carries exactly ONE line (the JSON result of rank 0): whatever libraries print there — RCCL
    # writes a five-line version banner to stdout when the communicator is created — goes to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist

    from xhistogram_amd import _native, core

    if args.selftest:
        return selftest(args, torch, dist, result_fd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    launched = "RANK" in os.environ and "MASTER_ADDR" in os.environ  # under torch.distributed.run (any N)
    if world != args.gpus:
        raise SystemExit("bench.py rank %d: --gpus %d does not match WORLD_SIZE %d" % (rank, args.gpus, world))
    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if local >= visible:
        raise SystemExit("bench.py rank %d of %d: needs GPU %d, but this host shows %d MI355X device(s) to the process "
                         "(torch.cuda.device_count()); one rank per GPU" % (rank, world, local, visible))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or launched
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # a rank that never shows up must end the run with a message, not with PyTorch's default ten-minute wait
        import datetime
        limit = float(os.environ.get("XHIST_AMD_COMM_TIMEOUT_S", "120"))
        dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=limit if limit > 0 else 1800))
    _native.require_device(local)

    wl = build_workload(args.config, args, torch, dev, rank)
    t0 = time.perf_counter()
    plan = core._get_plan(wl["edges"], _native.CMP_F64, local)
    plan.create_ms = round((time.perf_counter() - t0) * 1e3, 3)  # the first plan of the process: library load + context + edge tables
    for kv in args.tune:
        key, _, val = kv.partition("=")
        plan.set_param(key, int(val))
    sync = lambda: torch.cuda.synchronize(dev)  # noqa: E731
    stream = torch.cuda.current_stream(dev).cuda_stream

    def other_configs():
        """BASELINE.json configs[2..4] (C3, C4, C5) under the same clock as the headline: short legs in this process after the
        C2 legs, reported under `"configs"` of the one JSON line (metric / config / dtype stay C2's).  Hard wall budget:
        steps shrink, sizes never; what does not fit the budget is reported as skipped, never as a smaller problem."""
        import copy

        out = {}
        t_start = time.perf_counter()
        budget_s = 45.0
        for cfg, steps, warmup in (("c3", 20, 10), ("c4", 200, 50), ("c5", 20, 10)):  # (the steps / warm-up of a dedicated --config run, see parse())
            if time.perf_counter() - t_start > budget_s:
                out[cfg] = {"skipped": "wall budget of %.0f s used up by the configs before it" % budget_s}
                continue
            a = copy.copy(args)
            a.config, a.steps, a.warmup, a.full, a.unweighted, a.no_cpu_baseline, a.profiler_pass = cfg, steps, warmup, False, False, True, False
            a.samples = 1_000_000_000  # BASELINE.json's sizes, whatever --samples the headline was given
            wl2 = None
            try:  # (a failure here must not take the headline line with it)
                wl2 = build_workload(cfg, a, torch, dev, rank)
                plan2 = core._get_plan(wl2["edges"], _native.CMP_F64, local)
                out[cfg] = measure_and_report(a, torch, dist, _native, core, wl2, plan2, dev, world, rank, use_dist, result_fd, sync=sync, stream=stream, as_extra=True)
                if cfg == "c5":
                    # the leg above moves float64 weights between its two passes as 48-bit records (36 mantissa bits, 2^-37
                    # relative per weight; one sign only, decided on the GPU) — narrower than the reference's float64 adds
                    # (core.py:81) though inside the 1e-6 contract.  The same call with FULL float64 records beside it:
                    out[cfg]["records"] = "packed48 (float64 weights rounded to 36 mantissa bits on their way from the workgroup that read them to the one that adds them; exact_records = full float64)"
                    plan2.set_param("records48", -1)
                    try:
                        a.steps, a.warmup = 10, 5
                        ex = measure_and_report(a, torch, dist, _native, core, wl2, plan2, dev, world, rank, use_dist, result_fd, sync=sync, stream=stream, as_extra=True)
                        out[cfg]["exact_records"] = {k: ex[k] for k in ("steps", "warmup", "value", "ms_per_step", "kernel_ms_mean", "kernel_ms_min", "achieved_GBps", "frac", "kernel", "verified")}
                    finally:
                        plan2.set_param("records48", 0)
                    # The packed leg above ran in the exchange mode if its description says so: N(0,1) samples put
                    # 94 % of the records into the window the probe picks.  Beside it: the same call on the classic pair of
                    # passes (the mode switched off), and the same shape with UNIFORM samples, which the probe refuses
                    # (47 % in the window) — what a call that cannot take the mode costs, and that the default does not misfire.
                    keys = ("steps", "warmup", "value", "ms_per_step", "kernel_ms_mean", "kernel_ms_min", "achieved_GBps", "frac", "kernel", "verified")
                    plan2.set_param("exchange", -1)
                    try:
                        a.steps, a.warmup = 10, 5
                        cl = measure_and_report(a, torch, dist, _native, core, wl2, plan2, dev, world, rank, use_dist, result_fd, sync=sync, stream=stream, as_extra=True)
                        out[cfg]["classic_passes"] = {k: cl[k] for k in keys}
                    finally:
                        plan2.set_param("exchange", 0)
                    if time.perf_counter() - t_start < budget_s + 15.0:
                        for t in wl2["arrays"]:
                            t.uniform_(-4.0, 4.0)
                        a.steps, a.warmup = 5, 3
                        un = measure_and_report(a, torch, dist, _native, core, wl2, plan2, dev, world, rank, use_dist, result_fd, sync=sync, stream=stream, as_extra=True)
                        out[cfg]["uniform_samples"] = {k: un[k] for k in keys}
            except Exception as e:  # noqa: BLE001
                out[cfg] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
            del wl2
            torch.cuda.empty_cache()
        return out

    default_run = (args.config == "c2" and world == 1 and not args.unweighted and not args.profiler_pass and not args.tune and args.dist == "normal" and
                   args.samples == 1_000_000_000 and not args.no_other_configs)
    measure_and_report(args, torch, dist, _native, core, wl, plan, dev, world, rank, use_dist, result_fd, sync=sync, stream=stream,
                       extra=other_configs if default_run else None)


def measure_and_report(args, torch, dist, _native, core, wl, plan, dev, world, rank, use_dist, result_fd, sync, stream, extra=None, as_extra=False):
    """the timed legs and the JSON line (shared with --selftest, which passes a plan double and CPU tensors).
    extra: callable returning the `"configs"` object (the other BASELINE configs, measured before the line is printed);
    as_extra: this call IS one of those — return its summary instead of printing a line"""
    arrays, w, edges = wl["arrays"], wl["weights"], wl["edges"]
    weighted = w is not None
    failed = [False]
    n_rows = wl["rows"]
    tag = {torch.float64: _native.F64, torch.float32: _native.F32}
    # two result buffers: the RCCL all-reduce of step k runs while step k+1's kernel streams
    out_shape = (n_rows,) + plan.bins_shape
    outs = [torch.zeros(out_shape, dtype=torch.float64 if weighted else torch.int64, device=dev) for _ in range(2)]
    reduce_partials = use_dist and wl["reduce"] == "allreduce"
    density = bool(wl.get("density"))
    bytes_per_sample = sum(a.element_size() for a in arrays) + (w.element_size() if weighted else 0)

    def fence():
        if use_dist:
            dist.barrier()
        sync()

    verify = not args.no_verify and not args.profiler_pass and not args.selftest

    def check_leg(out, dens_out, n_cols, weighted):
        """the leg's last output against torch_reference at full size (all-reduced like the output when the partials are)"""
        t0 = time.perf_counter()
        ref = torch_reference(torch, arrays, w if weighted else None, edges, n_rows, n_cols, weighted)
        if reduce_partials:
            dist.all_reduce(ref, op=dist.ReduceOp.SUM)
        v = compare_with_reference(torch, out, ref)
        if density and dens_out is not None:  # core.py:444-462 restated: counts / bin areas / the row's in-range total
            area = torch.as_tensor(np.diff(edges[0]), device=dev)
            for e in edges[1:]:
                area = (area[:, None] * torch.as_tensor(np.diff(e), device=dev)[None, :]).reshape(-1)
            vd = compare_with_reference(torch, dens_out, ref.to(torch.float64) / area[None, :] / ref.sum(dim=1, keepdim=True))
            v["density_ok"], v["density_max_rel"] = vd["ok"], vd["max_rel"]
            v["ok"] = v["ok"] and vd["ok"]
        del ref
        sync()
        v["reference"] = "torch.bucketize(right=True) + last-edge rule + joint index + torch.bincount over all %d samples, in this process" % (n_rows * n_cols)
        v["seconds"] = round(time.perf_counter() - t0, 3)
        return v

    def run_leg(n_cols, steps, warmup, weighted=weighted, outs=outs, cold=0, time_first=False):
        """`warmup` untimed + `steps` timed passes over the first n_cols samples of every row of this rank's
        resident arrays; returns what the JSON line needs.  A step = output zeroing + histogram kernel(s)
        (+ all-reduce of the partial over RCCL, overlapped with the next step's kernel; + the density epilogue).
        weighted=False on a weighted workload: the same samples without their weights (the 8 B/sample variant).
        cold > 0: before anything else, that many launches straight after half a second of idle GPU, timed one by one."""
        xv = [_native.make_view(a.data_ptr(), tag[a.dtype], wl["cols"], 1) for a in arrays]
        wv = _native.make_view(w.data_ptr(), tag[w.dtype], wl["cols"], 1) if weighted else None
        bytes_per_sample = sum(a.element_size() for a in arrays) + (w.element_size() if weighted else 0)
        pending = [None, None]
        dens = [None]
        counter = [0]
        # prepared calls (ctypes arguments built once), one per result buffer
        launch = [plan.bind(xv, wv, n_rows, n_cols, o.data_ptr(), weighted, _native.MEM_DEVICE, accumulate=False, stream=stream) for o in outs]

        def finish(k):
            if pending[k] is not None:
                pending[k].wait()
                pending[k] = None
                if density:
                    dens[0] = core._density(outs[k], edges, len(edges))

        def step():
            k = counter[0] & 1
            counter[0] += 1
            finish(k)  # the reduction that last used this buffer must be done
            if reduce_partials and wl.get("wants_whole_gpu"):
                # C5's exchange-mode kernel is 256 persistent workgroups that wait for one another: an RCCL
                # kernel that holds a few compute units while ITS peers are busy would hold part of this kernel back for as
                # long.  So step k + 1 starts when the all-reduce of step k is done (a stream-side wait, 8 MiB over xGMI)
                finish(k ^ 1)
            out = outs[k]
            launch[k]()
            if reduce_partials:
                pending[k] = dist.all_reduce(out, op=dist.ReduceOp.SUM, async_op=True)
            elif density:
                dens[0] = core._density(out, edges, len(edges))

        def drain():
            for k in (0, 1):
                finish(k)
            fence()

        cold_ms = None
        first_ms = None
        if time_first:  # the very first launch of this plan in this process: module load + function attributes + the kernel
            t0 = time.perf_counter()
            step()
            drain()
            first_ms = (time.perf_counter() - t0) * 1e3
        if cold:
            step()  # (the very first launch of a plan pays module loading and function attributes: 8 ms, not a clock effect)
            drain()
            time.sleep(0.5)
            plan.set_param("profile", cold)
            for _ in range(cold):
                step()
            drain()
            cold_ms = plan.profile_read()
            plan.set_param("profile", 0)
        for _ in range(warmup):
            step()
        drain()
        plan.set_param("profile", min(steps, 4096))
        # C1's kernel takes 9 us and the two HIP events around it another 8: time every 4th launch there
        stride = 4 if (args.config == "c1" and steps >= 8 and not args.selftest) else 1
        plan.set_param("profile_stride", stride)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        drain()
        dt = time.perf_counter() - t0
        kernel_ms = plan.profile_read()
        plan.set_param("profile", 0)
        plan.set_param("profile_stride", 1)
        out = outs[(counter[0] - 1) & 1]
        k_mean = float(np.mean(kernel_ms))
        per_rank = [k_mean]
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            ks = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
            dist.all_gather(ks, torch.tensor([k_mean], dtype=torch.float64, device=dev))
            per_rank = [float(k.item()) for k in ks]
        # the result of the last step must be the histogram of these samples: checked against torch ops at full size
        total = float(out.sum().item())
        assert total > 0
        verdict = check_leg(out, dens[0], n_cols, weighted) if verify else None
        # the exchange on its own: `steps` all-reduces of the partial, nothing else on the GPU
        allreduce_ms = None
        if reduce_partials:
            fence()
            t0 = time.perf_counter()
            for _ in range(steps):
                dist.all_reduce(outs[0], op=dist.ReduceOp.SUM)
            sync()
            allreduce_ms = (time.perf_counter() - t0) / steps * 1e3
            fence()
        n = n_rows * n_cols
        desc = plan.describe()
        return dict(desc=desc, dt=dt, n=n, kernel_ms=kernel_ms, kernel_ms_per_rank=per_rank, allreduce_ms=allreduce_ms, cold_ms=cold_ms,
                    verified=verdict, first_ms=first_ms,
                    value=world * n * steps / dt, ms_per_step=dt / steps * 1e3, last=out, dens=dens[0], bytes_per_sample=bytes_per_sample)

    cols_weak = wl["cols"]
    cols_strong = max(1, wl["cols"] // world)
    legs = {}
    main_leg = args.scaling
    # sub-millisecond kernels (the C4 shard): the first launches after an idle GPU are reported next to the steady rate
    cold = 20 if (args.config == "c4" and not args.full and not args.selftest and not args.profiler_pass) else 0
    legs[main_leg] = run_leg(cols_weak if main_leg == "weak" else cols_strong, args.steps, args.warmup, cold=cold, time_first=not as_extra and not args.profiler_pass)
    # the headline's 8 B/sample variant — north_star's target sentence is the 1-D 10^9-sample f64 histogram WITHOUT weights
    # (BASELINE.md section 3 "headline unweighted variant") — rides along on the same samples: `"unweighted": {...}`
    unweighted_leg = None
    if args.config == "c2" and weighted and not args.profiler_pass:
        outs_u = [torch.zeros(out_shape, dtype=torch.int64, device=dev) for _ in range(2)]
        unweighted_leg = run_leg(cols_weak if main_leg == "weak" else cols_strong, args.steps, args.warmup, weighted=False, outs=outs_u)
    # the bounded host sample of the CPU baseline is taken NOW: the distribution legs below overwrite the samples in place
    host_sample = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.selftest and not as_extra:
        k = min(args.cpu_sample // max(1, len(arrays)), arrays[0].numel())
        host_sample = ([a.reshape(-1)[:k].cpu().numpy() for a in arrays], w.reshape(-1)[:k].cpu().numpy() if weighted else None)
    # the reference's real call shape (numpy in, numpy out) on a bounded prefix of the same samples: VERDICT r5 "next" #6
    host_leg = None
    if args.config == "c2" and world == 1 and rank == 0 and extra is not None and not args.selftest:
        try:
            host_leg = host_inputs_leg(torch, core, dev, arrays, w, edges)
        except Exception as e:  # noqa: BLE001  (a failure here must not take the headline line with it)
            host_leg = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    # SURVEY.md 8(d) / hard part #2: the headline again on three other sample distributions — uniform[-4,4) (every bin equally
    # likely: the low-contention bound), ALL samples in one bin (every lane of every wavefront on one counter: the contention
    # bound of core.py:81's sequential add turned into LDS atomics) and 90 % of the samples out of range (the drop path).
    # Same buffers, samples overwritten in place; a few steps each; every leg verified like the main ones.
    dist_legs = None
    if args.config == "c2" and world == 1 and unweighted_leg is not None and (args.distributions or extra is not None) and not args.selftest:
        dist_legs = {}
        x = arrays[0]
        fills = (("uniform[-4,4)", lambda: x.uniform_(-4.0, 4.0)), ("all_in_one_bin", lambda: x.fill_(0.5)),
                 ("90pct_out_of_range", lambda: x.uniform_(-40.0, 40.0)))
        torch.manual_seed(4321)
        for name, fill in fills:
            fill()
            sync()
            # (10 untimed launches, like the headline's legs: each of these starts after an in-place refill and the torch kernels
            #  of the last leg's verification — with 3, the 8 B/sample leg on uniform samples was timed inside the clock excursion
            #  of a just-woken GPU, DESIGN 4.3, and read 3-6 % low through rounds 4-5; a dedicated run reads it level with N(0,1):
            #  profiles/r06_b_c2u_uniform_dedicated.txt)
            dist_legs[name] = (run_leg(cols_weak, 10, 10), run_leg(cols_weak, 10, 10, weighted=False, outs=outs_u))
    if world > 1:  # the other leg rides along (N = 1: the two legs are the same run)
        other = "strong" if main_leg == "weak" else "weak"
        legs[other] = run_leg(cols_weak if other == "weak" else cols_strong, args.steps, args.warmup)
    m = legs[main_leg]
    ranks_block = None if as_extra else rank_identity(torch, dist, dev, use_dist, selftest=args.selftest)  # (a collective: every rank)

    if rank == 0:
        def roofline(leg):
            k_ms = float(np.mean(leg["kernel_ms"]))
            bytes_per_sample = leg["bytes_per_sample"]
            achieved = bytes_per_sample * leg["n"] / (k_ms * 1e-3) / 1e9
            return {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": None,
                "kernel_ms_mean": k_ms,
                "kernel_ms_min": float(np.min(leg["kernel_ms"])),
                "kernel_launches_timed": len(leg["kernel_ms"]),
                "algorithmic_bytes_per_launch": bytes_per_sample * leg["n"],
            }

        roof = roofline(m)
        # HBM traffic of one launch from the PMC counters: a rocprofv3 --pmc pass cannot run inside this process, so
        # the figure comes from profiles/traffic.json (written by tools/pmc_traffic.py from --pmc FETCH_SIZE / WRITE_SIZE
        # passes over THIS command) and is reported only when it was taken on the kernel and size this run used —
        # `traffic_source` says where and on which code state; otherwise null
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        try:
            entry = json.load(open(tpath)).get("configs", {}).get(args.config + ("u" if args.unweighted else "") + ("_full" if args.full else ""))
        except Exception:
            entry = None
        sha_now = csrc_sha16()
        roof["csrc_sha16"] = sha_now  # the native sources this library was built from (tools/pmc_traffic.py files it with the counters)
        if entry and entry.get("kernel") == m["desc"] and entry.get("samples_per_launch") == m["n"] and entry.get("csrc_sha16") not in (None, sha_now):
            roof["traffic_source"] = None
            roof["traffic_refused"] = ("profiles/traffic.json was measured on csrc %s (code state %s), this library is built from csrc %s: "
                                       "counters of another code state are not reported" % (entry.get("csrc_sha16"), entry.get("code_state", "?"), sha_now))
        elif entry and entry.get("kernel") == m["desc"] and entry.get("samples_per_launch") == m["n"] and entry.get("csrc_sha16") == sha_now:
            roof["traffic"] = entry["hbm_bytes_per_launch"]
            roof["traffic_source"] = "profiles/traffic.json: %s (code state %s, csrc %s = this build's sources)" % (entry.get("source", "?"), entry.get("code_state", "?"), sha_now)
            # the rocprofv3 --kernel-trace average of the same command (the first launches after the idle gap of data generation
            # left out) next to the HIP-event mean of THAT profiled process: the tracer slows these kernels by a few
            # percent, so profiles/ reproduces `frac` only through `profiled_slowdown`
            if entry.get("rocprof_avg_us") is not None:
                roof["rocprof_avg_us"] = entry["rocprof_avg_us"]
                roof["rocprof_launches_skipped"] = entry.get("rocprof_launches_skipped")
                roof["events_ms_under_rocprof"] = entry.get("events_ms_under_rocprof")
                if entry.get("events_ms_under_rocprof"):
                    roof["profiled_slowdown"] = entry["events_ms_under_rocprof"] / roof["kernel_ms_mean"]
        else:
            roof["traffic_source"] = None

        def leg_summary(leg):
            return {
                "value": leg["value"], "unit": "samples/s", "ms_per_step": leg["ms_per_step"], "samples_per_gpu": leg["n"],
                "samples_total": leg["n"] * world, "kernel_ms_per_rank": leg["kernel_ms_per_rank"],
                "allreduce_ms_alone": leg["allreduce_ms"], "roofline_frac_rank0": roofline(leg)["frac"],
                "overhead_us_per_step": (leg["ms_per_step"] - float(np.mean(leg["kernel_ms"]))) * 1e3,
            }

        line = {
            "metric": ("SELFTEST of the harness control flow, nothing measured: " if args.selftest else "") + wl["metric"],
            "value": m["value"],
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": m["ms_per_step"],
            "higher_is_better": True,
            "scaling": main_leg,
            "vs_baseline": None,
            "dtype": wl["dtype"] + (" (records: %s)" % ("packed48" if "packed48" in m["desc"] else "float64") if args.config == "c5" else ""),
            "data": wl["data"],
            "config": {
                "workload": wl["workload"],
                "samples_per_gpu": m["n"],
                "samples_total": m["n"] * world,
                "bins": [int(b) for b in plan.bins_shape],
                "weighted": weighted,
                "kernel": m["desc"],
                "parallelism": ("sample-axis shards, one per GPU" if wl["reduce"] == "allreduce" else "kept-axis (time) shards, one per GPU, disjoint output rows")
                + (("; all-reduce(sum) of the partial histogram over RCCL each step, " + ("finished before the next step's kernel starts (it wants every compute unit)" if wl.get("wants_whole_gpu") else "overlapped with the next step's kernel")) if reduce_partials else ""),
            },
            "roofline": roof,
            "ranks": ranks_block,
            "kernel_ms_per_rank": m["kernel_ms_per_rank"],
            "allreduce_ms_alone": m["allreduce_ms"],
            # everything a step costs beyond its histogram kernel(s): zeroing, launch, the exchange's share that does not
            # overlap, the wait for the buffer's previous reduction (slowest rank's step time - rank 0's kernel time)
            "overhead_us_per_step": (m["ms_per_step"] - roof["kernel_ms_mean"]) * 1e3,
        }
        if m["cold_ms"]:
            c_ms = float(np.mean(m["cold_ms"]))
            c_ach = m["bytes_per_sample"] * m["n"] / (c_ms * 1e-3) / 1e9
            roof["cold"] = {"launches": len(m["cold_ms"]), "after_idle_s": 0.5, "kernel_ms_mean": c_ms, "kernel_ms": [round(float(v), 4) for v in m["cold_ms"]],
                            "achieved": c_ach, "frac": c_ach / HBM_PEAK_GBS}
            roof["cold_frac"] = c_ach / HBM_PEAK_GBS
        verified = {}
        if m["verified"] is not None:
            verified[args.config] = m["verified"]
        if as_extra:
            summary = {"workload": wl["workload"], "metric": wl["metric"], "dtype": wl["dtype"], "steps": args.steps, "warmup": args.warmup,
                       "value": m["value"], "unit": "samples/s", "ms_per_step": m["ms_per_step"], "kernel_ms_mean": roof["kernel_ms_mean"],
                       "kernel_ms_min": roof["kernel_ms_min"], "achieved_GBps": roof["achieved"], "frac": roof["frac"],
                       "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"], "kernel": m["desc"], "verified": m["verified"]}
            if "cold_frac" in roof:
                summary["cold_frac"] = roof["cold_frac"]
                summary["cold_kernel_ms_mean"] = roof["cold"]["kernel_ms_mean"]
            return summary
        if unweighted_leg is not None:
            u = unweighted_leg
            ur = roofline(u)
            line["unweighted"] = {
                "metric": "samples/s binned (f64), 1D %d-bin 10^9 elems per GPU, no weights (8 B/sample)" % plan.bins_shape[0],
                "value": u["value"], "unit": "samples/s", "ms_per_step": u["ms_per_step"], "kernel_ms_mean": ur["kernel_ms_mean"],
                "kernel_ms_per_rank": u["kernel_ms_per_rank"], "overhead_us_per_step": (u["ms_per_step"] - ur["kernel_ms_mean"]) * 1e3,
                "roofline": {k: ur[k] for k in ("bound", "achieved", "peak", "unit", "frac", "kernel_ms_mean", "kernel_ms_min",
                                                "kernel_launches_timed", "algorithmic_bytes_per_launch")},
                "kernel": u.get("desc"),
            }
            if u["verified"] is not None:
                verified["c2_unweighted"] = u["verified"]
        if dist_legs:
            def short(leg):
                r = roofline(leg)
                return {"kernel_ms_mean": r["kernel_ms_mean"], "frac": r["frac"], "value": leg["value"], "kernel": leg["desc"],
                        "verified": None if leg["verified"] is None else leg["verified"]["ok"]}

            line["distributions"] = {"steps": 10, "warmup": 10,
                                     "N(0,1) (the headline)": {"weighted": {"kernel_ms_mean": roof["kernel_ms_mean"], "frac": roof["frac"]},
                                                              "unweighted": {"kernel_ms_mean": line["unweighted"]["kernel_ms_mean"],
                                                                             "frac": line["unweighted"]["roofline"]["frac"]}}}
            for name, (lw, lu) in dist_legs.items():
                line["distributions"][name] = {"weighted": short(lw), "unweighted": short(lu)}
                for tag_, leg in (("weighted", lw), ("unweighted", lu)):
                    if leg["verified"] is not None:
                        verified["c2 %s %s" % (name, tag_)] = leg["verified"]
        if m["first_ms"] is not None:
            # cold start of the headline call in this process: table building + plan (xhist_plan_create), then module load +
            # function attributes + the first launch; the library is %d MB of code objects for ~1 350 kernel instantiations
            line["first_call_ms"] = {"first_launch_ms": round(m["first_ms"], 3), "plan_create_ms": getattr(plan, "create_ms", None),
                                     "so_bytes": os.path.getsize(_native.LIB_PATH) if hasattr(_native, "LIB_PATH") and os.path.exists(_native.LIB_PATH) else None}
        for name, leg in legs.items():
            if name != main_leg:
                line[name] = leg_summary(leg)
        for name, leg in legs.items():
            if name != main_leg and leg["verified"] is not None:
                verified["%s %s leg" % (args.config, name)] = leg["verified"]
        if extra is not None:
            t0 = time.perf_counter()
            line["configs"] = extra()
            line["configs"]["wall_s"] = round(time.perf_counter() - t0, 2)
            for cfg, summary in line["configs"].items():
                if isinstance(summary, dict):
                    if summary.get("verified") is not None:
                        verified[cfg] = summary["verified"]
                    ex = summary.get("exact_records")
                    if isinstance(ex, dict) and ex.get("verified") is not None:
                        verified[cfg + " exact records"] = ex["verified"]
        if host_leg is not None:
            line["host_inputs"] = host_leg
            if isinstance(host_leg.get("verified"), dict):
                verified["c2 host inputs (numpy in, numpy out)"] = host_leg["verified"]
        if host_sample is not None:
            line["cpu_baseline"] = cpu_baseline(host_sample[0], host_sample[1], edges, one_chunk=args.config == "c1")
        if verified:
            line["verified"] = dict(all_ok=all(v["ok"] for v in verified.values()), rtol_float64=VERIFY_RTOL, legs=verified)
            failed[0] = not line["verified"]["all_ok"]
        # LAST key: what a reader of a truncated tail needs — the roofline fraction (of 8 TB/s) of every leg and whether every
        # verified leg matched (VERDICT r5 "weak" #8: the line is longer than the 8 KB the driver keeps)
        summ = {args.config + ("u" if args.unweighted else ""): round(roof["frac"], 4)}
        if unweighted_leg is not None:
            summ["c2u"] = round(line["unweighted"]["roofline"]["frac"], 4)
        if dist_legs:
            summ["c2_uniform"] = round(line["distributions"]["uniform[-4,4)"]["weighted"]["frac"], 4)
            summ["c2u_uniform"] = round(line["distributions"]["uniform[-4,4)"]["unweighted"]["frac"], 4)
        cfgs = line.get("configs") or {}
        for cfg in ("c3", "c4", "c5"):
            c = cfgs.get(cfg)
            if isinstance(c, dict) and "frac" in c:
                summ[cfg] = round(c["frac"], 4)
                if "cold_frac" in c:
                    summ[cfg + "_cold"] = round(c["cold_frac"], 4)
                for sub, key in (("exact_records", "c5_exact"), ("classic_passes", "c5_classic"), ("uniform_samples", "c5_uniform")):
                    if isinstance(c.get(sub), dict) and "frac" in c[sub]:
                        summ[key] = round(c[sub]["frac"], 4)
            elif isinstance(c, dict):
                summ[cfg] = c.get("error") or c.get("skipped")
        if host_leg is not None and "pinned" in host_leg:
            summ["host_pinned_GBps"] = host_leg["pinned"]["GBps"]
            summ["host_pinned_frac_of_h2d_copy"] = host_leg["pinned"]["frac_of_copy_rate"]
            summ["host_pageable_GBps"] = host_leg["pageable"]["GBps"]
        summ["n_gpus"] = world
        summ["distinct_devices"] = ranks_block["distinct_devices"] if ranks_block else None
        summ["all_verified"] = line["verified"]["all_ok"] if "verified" in line else None
        line["summary"] = summ
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(line) + "\n").encode())
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if failed[0]:
        print("bench.py: a leg's output did NOT match the torch reference — see \"verified\" in the line", file=sys.stderr)
        raise SystemExit(1)


def selftest(args, torch, dist, result_fd):
    """--selftest: the multi-rank control flow of this script on CPU (gloo), with a plan double whose launch only marks
    the output; no kernel runs and nothing is measured.  What it proves: ranks spawn and rendezvous, both scaling legs
    run, the per-rank gathers and the all-reduce timing work, rank 0 prints ONE well-formed JSON line."""
    from xhistogram_amd import _native, core

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py rank %d: --gpus %d does not match WORLD_SIZE %d" % (rank, args.gpus, world))
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        dist.init_process_group("gloo")
    dev = torch.device("cpu")
    args.samples = min(args.samples, 20_000)
    wl = build_workload("c2", args, torch, dev, rank)

    class PlanDouble:
        bins_shape = (args.bins,)

        def __init__(self):
            self.n = 0

        def bind(self, xv, wv, n_rows, n_cols, out_ptr, weighted, mem, accumulate=False, stream=0):
            out = next(o for o in self.outs if o.data_ptr() == out_ptr)

            def run():
                out.fill_(1.0)  # "a histogram was produced": the harness only checks that the total is positive
                self.n += 1
            return run

        def set_param(self, key, value):
            self.k = int(value) if key == "profile" else getattr(self, "k", 0)

        def profile_read(self):
            return [1e-3] * max(1, getattr(self, "k", 1))

        def describe(self):
            return "selftest: no kernel"

    plan = PlanDouble()
    orig_zeros = torch.zeros

    def zeros_spy(*a, **k):  # the plan double needs to find the output tensors the harness allocates
        t = orig_zeros(*a, **k)
        plan.outs = getattr(plan, "outs", []) + [t]
        return t

    torch.zeros = zeros_spy
    try:
        measure_and_report(args, torch, dist, _native, core, wl, plan, dev, world, rank, use_dist, result_fd, sync=lambda: None, stream=0)
    finally:
        torch.zeros = orig_zeros


if __name__ == "__main__":
    main()
