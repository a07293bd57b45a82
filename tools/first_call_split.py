import time, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
torch.cuda.init(); x = torch.randn(10**6, dtype=torch.float64, device="cuda"); torch.cuda.synchronize()
from xhistogram_amd import _native, core
t0=time.perf_counter(); _native.require_device(0); t1=time.perf_counter()
e1=np.linspace(-4,4,101); e2=np.linspace(-3,3,81); e3=np.sort(np.random.default_rng(0).uniform(-4,4,90))
p1=core._get_plan([e1], _native.CMP_F64, 0); t2=time.perf_counter()
p2=core._get_plan([e2], _native.CMP_F64, 0); t3=time.perf_counter()
h=core._bincount_2d_vectorized(x[None,:], bins=[e1]); torch.cuda.synchronize(); t4=time.perf_counter()
h=core._bincount_2d_vectorized(x[None,:], bins=[e2]); torch.cuda.synchronize(); t5=time.perf_counter()
p3=core._get_plan([e3], _native.CMP_F64, 0); t6=time.perf_counter()
h=core._bincount_2d_vectorized(x[None,:], bins=[e3]); torch.cuda.synchronize(); t7=time.perf_counter()
xf=x.float()
h=core._bincount_2d_vectorized(xf[None,:], bins=[e1]); torch.cuda.synchronize(); t8=time.perf_counter()
print("load+device %.2f ms | plan1 %.2f | plan2 (same TU loaded) %.2f | first exec %.2f | second exec (other plan, same kernel) %.2f | plan3 (random edges) %.2f | exec3 (other kernel, same TU) %.2f | f32 first exec (other TU) %.2f" % tuple(1e3*v for v in (t1-t0,t2-t1,t3-t2,t4-t3,t5-t4,t6-t5,t7-t6,t8-t7)))
