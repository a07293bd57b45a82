"""Randomised differential soak of the exchange mode (xhistogram_amd/csrc/xhist_exchange.hip.h) against the classic partitioned
passes of the same library, both on the GPU (development tool; the classic passes are held to the oracle by the test-suite and
by tools/soak.py).  Random shapes: 1-3 inputs, bins per input, np.linspace ranges, sample distributions (normal, uniform, a
constant, heavy NaN / infinity mixes, samples ON edges), weights of one sign (either), both signs now and then (the exact
fallback), every third case with exact float64 records through the rings, sizes around the 4096-sample tile and up to a few million.

    python tools/soak_exchange.py [seconds] [seed] [one big case every N]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from xhistogram_amd import _native, core

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
big_every = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # every how many cases one with 2*10^7 ... 7*10^7 samples (0: never)


def one(seed):
    rng = np.random.default_rng(seed)
    d = int(rng.choice([1, 2, 2, 2, 3]))
    if d == 1:
        nbs = [int(rng.integers(20_000, 1_500_000))]
    elif d == 2:
        nbs = [int(rng.integers(40, 3000)), int(rng.choice([int(rng.integers(8, 4000)), 1024, 512, 100]))]
    else:
        nbs = [int(rng.integers(8, 200)), int(rng.integers(8, 120)), int(rng.integers(8, 600))]
    if int(np.prod(nbs)) > (1 << 23) or int(np.prod(nbs)) < 30_000:
        return None
    n = int(rng.choice([int(rng.integers(4, 9000)), int(rng.integers(9000, 400_000)), int(rng.integers(400_000, 4_000_000))]))
    if big_every and seed % big_every == 0:  # (now and then a call of many tiles per workgroup: the rings go round hundreds of times)
        n = int(rng.integers(20_000_000, 70_000_000))
    edges, samples = [], []
    for nb in nbs:
        lo = float(rng.uniform(-10, 5))
        hi = lo + float(rng.choice([1.0, 8.0, 1e-3, 1e6])) * float(rng.uniform(0.5, 2.0))
        e = np.linspace(lo, hi, nb + 1)
        edges.append(e)
        kind = rng.choice(["normal", "normal", "uniform", "wide", "const", "edges"])
        mid, span = 0.5 * (lo + hi), hi - lo
        if kind == "normal":
            x = rng.normal(mid + span * rng.uniform(-0.3, 0.3), span * rng.uniform(0.02, 0.3), n)
        elif kind == "uniform":
            x = rng.uniform(lo, hi, n)
        elif kind == "wide":
            x = rng.uniform(lo - span, hi + span, n)
        elif kind == "const":
            x = np.full(n, e[int(rng.integers(0, nb))] + span / nb * 0.37)
        else:
            x = e[rng.integers(0, nb + 1, n)].copy()
            x[::3] = np.nextafter(x[::3], np.inf)
            x[1::3] = np.nextafter(x[1::3], -np.inf)
        if rng.random() < 0.3:
            m = rng.random(n) < rng.uniform(0.001, 0.3)
            x[m] = rng.choice([np.nan, np.inf, -np.inf], int(m.sum()))
        samples.append(x[None, :])
    sign = rng.choice([1.0, 1.0, -1.0, 0.0])
    w = rng.uniform(0, 1, n) * float(rng.choice([1.0, 1e-200, 1e200]))
    w = w * sign if sign else rng.standard_normal(n)
    if rng.random() < 0.1:
        w[rng.integers(0, n, 3)] = np.nan
    w = w[None, :]
    dev = [torch.as_tensor(np.ascontiguousarray(s)).cuda() for s in samples]
    wd = torch.as_tensor(np.ascontiguousarray(w)).cuda()
    plan = core._get_plan(edges, _native.CMP_F64, 0)
    out = {}
    exact = seed % 3 == 0  # every third case with exact float64 records (12 bytes through two rings) against the classic exact passes
    for mode in (-1, 1):
        plan.set_param("partition", 1)
        plan.set_param("records48", -1 if exact else 0)
        plan.set_param("exchange", mode)
        try:
            out[mode] = core._bincount_2d_vectorized(*dev, bins=edges, weights=wd).cpu().numpy()
            desc = plan.describe()
        finally:
            plan.set_param("exchange", 0)
            plan.set_param("records48", 0)
            plan.set_param("partition", 0)
        if mode == 1 and "hist=partitioned" in desc and "exchange=forced" not in desc:
            return ("skipped", desc[-80:])
    a, b = out[-1], out[1]
    scale = float(np.nanmax(np.abs(w))) if np.isfinite(np.nanmax(np.abs(w))) else 1.0
    ok = np.allclose(a, b, rtol=1e-9, atol=1e-9 * scale * max(1, n) ** 0.5, equal_nan=True)
    return ("ok" if ok else "MISMATCH", dict(seed=seed, d=d, nbs=nbs, n=n, sign=float(sign), worst=float(np.nanmax(np.abs(a - b))) if not ok else 0.0))


def main():
    _native.require_device(0)
    t0 = time.time()
    seed = seed0
    counts = {"ok": 0, "MISMATCH": 0, "skipped": 0, "none": 0}
    while time.time() - t0 < budget:
        r = one(seed)
        seed += 1
        if r is None:
            counts["none"] += 1
            continue
        counts[r[0]] += 1
        if r[0] == "MISMATCH":
            print("MISMATCH", r[1], flush=True)
    print("exchange soak: %d cases in %.0f s from seed %d: %s" % (sum(counts.values()) - counts["none"], time.time() - t0, seed0, counts))
    return 1 if counts["MISMATCH"] else 0


if __name__ == "__main__":
    sys.exit(main())
