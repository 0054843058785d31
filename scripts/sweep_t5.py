"""Launch time of the T = 5 lane-pair kernel for every variant on two families of warm inputs:
the benchmark's (vehicles after closed-loop ticks on their courses, previous optimum fed back) and
random ones (vehicles scattered around their courses, warm start = their own cold optimum).  A launch
ends with its slowest wavefront, so one wavefront that takes the second attempt shows here.
python scripts/sweep_t5.py [B] [launches]        (MPCX_LIB selects the library)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from model_predictive_control_amd import engine, variants, workload
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300


def timed(v, x0, xref, up, hp):
    out = engine._outputs(B, v.nu, v.T, x0.device)
    for _ in range(300):
        engine.solve_batch(v, x0, xref, up, hp, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        engine.solve_batch(v, x0, xref, up, hp, out=out)
    e1.record(); torch.cuda.synchronize()
    st = np.bincount(out["status"].cpu().numpy(), minlength=5).tolist()
    return e0.elapsed_time(e1) / n * 1e3, st


row = []
for name in ("dsr_ackermann", "diff_drive", "dsr_crab", "dsr_diff", "sim_ackermann", "sim_diff", "sim_crab"):
    bench.VARIANT = name
    v, x0, xref, up, hp = bench.build_inputs(0, B)
    t_loop, st1 = timed(v, x0, xref, up, hp)
    courses = workload.make_courses(64)
    inst = workload.make_instances(courses, B, seed=2003)
    xr = torch.from_numpy(workload.reference_window(courses, inst, v.T)).cuda()
    xs = torch.from_numpy(inst["state"][:, :3].copy()).cuda()
    cold = engine.solve_batch(v, xs, xr)
    t_rand, st2 = timed(v, xs, xr, cold["u"].clone(), None)
    t_cold, st3 = timed(v, xs, xr, None, None)
    print(f"{name:14s} closed-loop warm {t_loop:7.1f} us   random warm {t_rand:7.1f} us   random cold {t_cold:7.1f} us   statuses {st1} {st2} {st3}", flush=True)
