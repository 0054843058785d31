"""Time one variant's solve on bench-shaped inputs: python scripts/time_variant.py name B [T].
Prints cold and warm ms per batch, status counts, iteration statistics."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_predictive_control_amd import engine, variants, workload

name, B = sys.argv[1], int(sys.argv[2])
T = int(sys.argv[3]) if len(sys.argv) > 3 else None
v = variants.get(name, T=T) if T else variants.get(name)
courses = workload.make_courses(64, seed=1001)
inst = workload.make_instances(courses, B, seed=2003)
window = workload.reference_window_next_points if name == "mpc_py" else workload.reference_window
d_xref = torch.from_numpy(window(courses, inst, v.T)).cuda()
d_x0 = torch.from_numpy(inst["state"][:, :3].copy()).cuda()
out = engine.solve_batch(v, d_x0, d_xref)
up = out["u"].clone()
for start, u in (("cold", None), ("warm", up)):
    engine.solve_batch(v, d_x0, d_xref, u, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        engine.solve_batch(v, d_x0, d_xref, u, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    st = out["status"].cpu().numpy(); it = out["iters"].cpu().numpy()
    print(f"{name} T={v.T} B={B} {start}: {dt*1e3:.3f} ms  status {np.bincount(st, minlength=5)}  iters mean {it.mean():.2f} max {it.max()}",
          "u checksum %.12g" % float(out["u"].double().abs().sum().item()))
