"""Per-wavefront view of the stage-wise kernel's work on the benchmark-shaped inputs (diagnostic
build: MPCX_LIB=.../libmpcx_diag.so): interior-point iterations and rounds per lane, and what the
slowest lane of each 64-lane wavefront needs, which is what the wavefront pays.
    python scripts/wave_stats.py name B T"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_predictive_control_amd import engine, variants, workload
name, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
v = variants.get(name, T=T)
courses = workload.make_courses(64, seed=1001)
inst = workload.make_instances(courses, B, seed=2003)
window = workload.reference_window_next_points if name == "mpc_py" else workload.reference_window
d_xref = torch.from_numpy(window(courses, inst, v.T)).cuda()
d_x0 = torch.from_numpy(inst["state"][:, :3].copy()).cuda()
out = engine.solve_batch(v, d_x0, d_xref)
for label, up in (("cold", None), ("warm", out["u"].clone())):
    o = engine.solve_batch(v, d_x0, d_xref, up)
    it = o["iters"].cpu().numpy()
    r2, rem = it // 10000, it % 10000
    r1, i1 = rem // 100, rem % 100
    w_i = i1.reshape(-1, 64).max(axis=1); w_r = r1.reshape(-1, 64).max(axis=1); w_2 = r2.reshape(-1, 64).max(axis=1)
    print(label, "lanes: ipm iters hist", np.bincount(i1).tolist(), "rounds hist", np.bincount(r1).tolist(), "second attempt", int((r2 > 0).sum()))
    print(label, "waves: max ipm iters hist", np.bincount(w_i).tolist(), "mean %.2f" % w_i.mean(), "| max rounds hist", np.bincount(w_r).tolist(),
          "mean %.2f" % w_r.mean(), "| waves with a second attempt", int((w_2 > 0).sum()), "of", w_i.size)
