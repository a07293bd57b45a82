#!/usr/bin/env python
"""Where does a big partitioned-mode call spend its wall time?  (C5 at 4e9 samples: 40 ms of kernels, 430 ms per
step in bench.py.)  Times calls back to back, with a sync between them, and with the stream-ordered pool's release
threshold raised, at several sizes."""
import ctypes
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from xhistogram_amd import _native, core  # noqa: E402

dev = torch.device("cuda", 0)
hip = ctypes.CDLL("libamdhip64.so")


def set_threshold(nbytes):
    pool = ctypes.c_void_p()
    assert hip.hipDeviceGetDefaultMemPool(ctypes.byref(pool), 0) == 0
    v = ctypes.c_uint64(nbytes)
    assert hip.hipMemPoolSetAttribute(pool, 4, ctypes.byref(v)) == 0  # hipMemPoolAttrReleaseThreshold = 4


e = np.linspace(-4, 4, 1025)
for n in (500_000_000, 2_000_000_000, 4_000_000_000):
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    x = torch.empty(n, dtype=torch.float64, device=dev).normal_(generator=g)
    y = torch.empty(n, dtype=torch.float64, device=dev).normal_(generator=g)
    w = torch.empty(n, dtype=torch.float64, device=dev).uniform_(generator=g)
    for label, thr in (("default", 0), ("raised", 2**62)):
        set_threshold(thr)
        for sync_between in (False, True):
            core.histogram(x, y, bins=[e, e], weights=w)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                core.histogram(x, y, bins=[e, e], weights=w)
                if sync_between:
                    torch.cuda.synchronize()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 4
            print(json.dumps({"n": n, "pool_threshold": label, "sync_between_calls": sync_between, "ms_per_call": dt * 1e3}), flush=True)
    del x, y, w
    torch.cuda.empty_cache()
