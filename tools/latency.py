#!/usr/bin/env python
"""Development probe: per-call latency of small histograms (BASELINE C1 = 10^6 samples)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    import torch
    from xhistogram_amd import core, _native
    edges = np.linspace(-4, 4, 101)
    plan = core._get_plan([edges], _native.CMP_F64, 0)
    stream = torch.cuda.current_stream().cuda_stream
    for n in (1_000, 100_000, 1_000_000, 10_000_000):
        x = torch.randn(n, dtype=torch.float64, device="cuda")
        xh = x.cpu().numpy()
        out = torch.zeros(100, dtype=torch.int64, device="cuda")
        xv = [_native.make_view(x.data_ptr(), _native.F64, n, 1)]
        def api():
            core.histogram(x, bins=edges)
        def raw():
            plan.execute(xv, None, 1, n, out.data_ptr(), False, _native.MEM_DEVICE, stream=stream)
        bound = plan.bind(xv, None, 1, n, out.data_ptr(), False, _native.MEM_DEVICE, stream=stream)
        def general():
            core.histogram(x, bins=edges, block_size=1 << 40)  # an explicit block size takes the general path (same result)
        def host():
            core.histogram(xh, bins=edges)
        def ref():
            np.histogram(xh, bins=edges)
        for name, fn, reps in (("core.histogram(torch)", api, 200), ("core.histogram(torch, general path)", general, 200), ("plan.execute", raw, 200), ("plan.bind()()", bound, 200), ("core.histogram(numpy)", host, 20), ("numpy.histogram_cpu", ref, 5)):
            fn(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps): fn()
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / reps
            print(json.dumps({"n": n, "path": name, "us_per_call": dt * 1e6, "Msamples_per_s": n / dt / 1e6}), flush=True)
        plan.set_param("profile", 50)
        for _ in range(50): raw()
        torch.cuda.synchronize()
        ms = plan.profile_read(); plan.set_param("profile", 0)
        print(json.dumps({"n": n, "path": "kernel_only(HIP events)", "us_per_call": float(np.median(ms)) * 1e3}), flush=True)

if __name__ == "__main__":
    main()
