"""Minimal driver for profiling: a few launches of one solve on resident inputs.
usage: prof_solve.py [variant] [batch] [launches] [horizon]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from model_predictive_control_amd import engine, variants, workload
name = sys.argv[1] if len(sys.argv) > 1 else "dsr_ackermann"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
n = int(sys.argv[3]) if len(sys.argv) > 3 else 5
T = int(sys.argv[4]) if len(sys.argv) > 4 else None
v = variants.get(name, T=T) if T else variants.get(name)
courses = workload.make_courses(64)
inst = workload.make_instances(courses, B, seed=2003)
xref = torch.from_numpy(workload.reference_window(courses, inst, v.T)).cuda()
x0 = torch.from_numpy(inst["state"][:, :3].copy()).cuda()
out = engine.solve_batch(v, x0, xref)
up = out["u"].clone()
for _ in range(n):
    engine.solve_batch(v, x0, xref, up, out=out)
torch.cuda.synchronize()
print("done", out["iters"].float().mean().item())
