"""Steady-state time of one solve kernel on the benchmark's own inputs (settled clock):
time_headline.py [variant] [batch] [launches]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from model_predictive_control_amd import engine
bench.VARIANT = sys.argv[1] if len(sys.argv) > 1 else "dsr_ackermann"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
n = int(sys.argv[3]) if len(sys.argv) > 3 else 500
v, x0, xref, up, hp = bench.build_inputs(0, B)
out = engine.solve_batch(v, x0, xref, up, hp)
for _ in range(n):
    engine.solve_batch(v, x0, xref, up, hp, out=out)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n):
    engine.solve_batch(v, x0, xref, up, hp, out=out)
b.record(); torch.cuda.synchronize()
st = out["status"].cpu().numpy()
print(f"{bench.VARIANT} B={B}: {a.elapsed_time(b) / n * 1e3:.2f} us per launch, status {np.bincount(st, minlength=5)}, iters mean {out['iters'].float().mean().item():.2f}, u checksum {float(out['u'].abs().sum().item()):.12g}")
