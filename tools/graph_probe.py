import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from xhistogram_amd import core, _native
edges = np.linspace(-4, 4, 101)
n = 1_000_000
x = torch.randn(n, dtype=torch.float64, device="cuda")
plan = core._get_plan([edges], _native.CMP_F64, 0)
out = torch.zeros(100, dtype=torch.int64, device="cuda")
xv = [_native.make_view(x.data_ptr(), _native.F64, n, 1)]
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    for _ in range(3):
        plan.execute(xv, None, 1, n, out.data_ptr(), False, _native.MEM_DEVICE, stream=s.cuda_stream)
torch.cuda.synchronize()
ref = out.clone()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g, stream=s):
        plan.execute(xv, None, 1, n, out.data_ptr(), False, _native.MEM_DEVICE, stream=torch.cuda.current_stream().cuda_stream)
    out.zero_()
    g.replay(); torch.cuda.synchronize()
    print("graph capture ok, replay equal:", bool(torch.equal(out, ref)))
    t0 = time.perf_counter()
    for _ in range(1000): g.replay()
    torch.cuda.synchronize()
    print("graph replay us/call: %.2f" % ((time.perf_counter() - t0) / 1000 * 1e6))
    t0 = time.perf_counter()
    for _ in range(1000): plan.execute(xv, None, 1, n, out.data_ptr(), False, _native.MEM_DEVICE, stream=0)
    torch.cuda.synchronize()
    print("eager us/call: %.2f" % ((time.perf_counter() - t0) / 1000 * 1e6))
except Exception as e:
    print("graph capture failed:", repr(e)[:300])
