"""Time one variant at one batch size: scripts/time_variant.py <name> <B> [T]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from model_predictive_control_amd import engine
from tests.helpers import make_problem
name, B = sys.argv[1], int(sys.argv[2])
kw = {"T": int(sys.argv[3])} if len(sys.argv) > 3 else {}
v, ov, x0, xref = make_problem(name, B, seed=3, n_courses=64, **kw)
xd, xr = torch.from_numpy(x0).cuda(), torch.from_numpy(xref).cuda()
out = engine.solve_batch(v, xd, xr)
up = out["u"].clone()
torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n):
    engine.solve_batch(v, xd, xr, up, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
st = out["status"].cpu().numpy()
print(f"{name} T={v.T} B={B}: {dt*1e3:.3f} ms/batch  {B/dt:.3e} solves/s  status {np.bincount(st, minlength=5)} iters {out['iters'].float().mean().item():.2f}")
