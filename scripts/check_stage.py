"""Stage-wise kernel on the long-horizon configurations: status counts, time per batch and the
distance to the oracle, cold and warm.  python scripts/check_stage.py [B] [name:T ...]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_predictive_control_amd import engine
from oracle import orc
from tests.helpers import make_problem

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cfgs = sys.argv[2:] or ["dsr_diff:30", "dsr_ackermann:30"]
nref = min(B, 4096)
for cfg in cfgs:
    name, T = cfg.split(":"); T = int(T)
    v, ov, x0, xref = make_problem(name, B, T=T, n_courses=64)
    d_x0, d_xref = torch.from_numpy(x0).cuda(), torch.from_numpy(xref).cuda()
    out = engine.solve_batch(v, d_x0, d_xref)
    torch.cuda.synchronize()
    up = out["u"].clone()
    for label, u in (("cold", None), ("warm", up)):
        o = engine.solve_batch(v, d_x0, d_xref, u)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            o = engine.solve_batch(v, d_x0, d_xref, u)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        st = o["status"].cpu().numpy()
        it = o["iters"].cpu().numpy()
        ref = orc.iterative_batch(ov, x0[:nref], xref[:nref], None if u is None else u[:nref].cpu().numpy(), None, nthreads=16)
        g = (st[:nref] <= 1) & (ref["status"] == 0)
        du = np.abs(o["u"][:nref].cpu().numpy() - ref["u"])[g].reshape(int(g.sum()), -1).max(axis=1)
        rel = (np.abs(o["obj"][:nref].cpu().numpy() - ref["obj"])[g] / np.maximum(1.0, np.abs(ref["obj"][g])))
        print(f"{name} T={T} B={B} {label}: {dt*1e3:.2f} ms  status {np.bincount(st, minlength=5).tolist()}  ipm iters mean {it.mean():.1f} max {it.max()}  "
              f"oracle status {np.bincount(ref['status'], minlength=5).tolist()}  du max {du.max():.2e} med {np.median(du):.2e}  dobj max {rel.max():.2e}", flush=True)
