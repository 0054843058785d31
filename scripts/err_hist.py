"""Histogram of |u_gpu - u_oracle| per instance (which lanes the polish did not settle)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from model_predictive_control_amd import engine, variants, workload
from oracle import mpc_oracle as O, orc
B = 65536
for name in sys.argv[1:] or ["dsr_ackermann"]:
    v = variants.get(name); ov = O.variant(name)
    courses = workload.make_courses(32)
    inst = workload.make_instances(courses, B, seed=7)
    xref = workload.reference_window(courses, inst, v.T)
    x0 = inst["state"][:, :3].copy()
    ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=16)
    for tag, up in (("cold", None), ("warm", ref["u"])):
        r = ref if up is None else orc.iterative_batch(ov, x0, xref, up, None, nthreads=16)
        out = engine.solve_batch(v, x0, xref, up)
        e = np.abs(out["u"].cpu().numpy() - r["u"]).reshape(B, -1).max(1)
        it = out["iters"].cpu().numpy()
        print(name, tag, "err>1e-9:", int((e > 1e-9).sum()), ">1e-6:", int((e > 1e-6).sum()), ">1e-4:", int((e > 1e-4).sum()),
              "max", e.max(), "waves with a >1e-9 lane:", int((e.reshape(-1, 32) > 1e-9).any(1).sum()), "of", B // 32,
              "iters mean", it.mean(), "wave-max mean", it.reshape(-1, 32).max(1).mean())
