"""Development probe: host-route executes (numpy inputs staged by the library) on 12 threads while ONE other thread does
something else with the HIP runtime — nothing (quiet), creating plans (create+keep / create+destroy), uploads, downloads,
kernels, or plain hipMalloc + hipFree (malloc_free).  With hipMallocAsync'ed staging (round 1 .. mid round 2) the last two
modes put a sample in the wrong bin about once in 10^4 calls.  python tools/race_probe.py <mode> [iterations per thread]"""
import sys, threading, time, numpy as np
sys.path.insert(0, ".")
from xhistogram_amd import _native
mode = sys.argv[1]
rng = np.random.default_rng(0)
ea, eb = np.linspace(-4, 4, 9), np.linspace(-4, 4, 10)
plan = _native.Plan([ea, eb], 0, 0)
stop = False
def churn():
    if mode in ("upload", "download", "add", "malloc_free"):
        buf = _native.DeviceBuffer(0, 4096); buf2 = _native.DeviceBuffer(0, 4096)
        h = np.arange(512, dtype=np.int64)
        while not stop:
            if mode == "upload": buf.upload(h)
            elif mode == "download": buf.download(h)
            elif mode == "add": buf.add(buf2, 512, _native.I64); buf.synchronize()
            else:
                t = _native.DeviceBuffer(0, 1 << 20); t.close()
            keep.append(0) if len(keep) < 10**7 else None
        return
    while not stop:
        p = _native.Plan([ea, eb], 0, 0)
        if mode == "create+destroy":
            p.close()
        else:
            keep.append(p)
keep = []
def hist2d(x, y):
    out = np.empty(plan.bins_shape, dtype=np.int64)
    xv = [_native.make_view(x.ctypes.data, _native.F64, 0, 0), _native.make_view(y.ctypes.data, _native.F64, 0, 0)]
    plan.execute(xv, None, 1, 1, out.ctypes.data, False, _native.MEM_HOST)
    return out
bad = [0]
def worker(seed):
    r = np.random.default_rng(seed)
    for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 300):
        x, y = r.standard_normal(1), r.standard_normal(1)
        want = np.histogram2d(x, y, bins=[ea, eb])[0].astype(np.int64)
        if not np.array_equal(hist2d(x, y), want):
            bad[0] += 1
ts = [threading.Thread(target=worker, args=(k,)) for k in range(12)]
c = threading.Thread(target=churn) if mode != "quiet" else None
if c: c.start()
[t.start() for t in ts]; [t.join() for t in ts]
stop = True
if c: c.join()
print(mode, "bad:", bad[0], "of", 12 * (int(sys.argv[2]) if len(sys.argv) > 2 else 300), "plans churned:", len(keep))
