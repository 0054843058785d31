"""mpc_dsr.py's T = 30 variant (fixed steer, Q 2^t, cond(H) about 5e10) on whichever kernel the library
routes the batch to: distance of the engine's and of the oracle's controls from the extended-precision
minimiser (tests/xu_problem.arbiter) on the instances where the two are more than 1e-5 apart.
python scripts/t30_accuracy.py [B] [seed]     (MPCX_LIB / MPCX_STAGEWISE select library and kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from model_predictive_control_amd import engine
from oracle import orc
from tests.helpers import make_problem
from tests.test_gpu_full_size import _settle_with_the_arbiter
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 37
T = 30
v, ov, x0, xref = make_problem("dsr_diff", B, seed=seed, n_courses=64, T=T)
ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=16)
out = engine.solve_batch(v, x0, xref)
up = out["u"].cpu().numpy()
ref2 = orc.iterative_batch(ov, x0, xref, up, None, nthreads=16)
out2 = engine.solve_batch(v, x0, xref, up)
ucold = np.tile(np.tile(np.asarray(ov.cold_start, float)[:ov.nu, None], (1, T))[None], (B, 1, 1))
for tag, o, r, ur in (("cold", out, ref, ucold), ("warm", out2, ref2, up)):
    g = {k: o[k].cpu().numpy() for k in ("u", "x", "obj", "status")}
    both = (g["status"] <= 1) & (r["status"] <= 1)
    du = np.abs(g["u"] - r["u"]).reshape(B, -1).max(1)
    far = np.flatnonzero(both & (du > 1e-5))[:200]
    d_gpu, d_ref = _settle_with_the_arbiter("dsr_diff", T, v, x0, xref, ur, g, r, far)
    print(f"dsr_diff T=30 B={B} {tag}: statuses engine {np.bincount(g['status'], minlength=5).tolist()} oracle {np.bincount(r['status'], minlength=5).tolist()}; "
          f"{int((both & (du > 1e-5)).sum())} instances more than 1e-5 apart (max {du[both].max():.2e}); from the extended-precision minimiser: "
          f"engine max {d_gpu.max(initial=0.0):.2e} median {np.median(d_gpu) if d_gpu.size else 0:.2e}, oracle max {d_ref.max(initial=0.0):.2e}", flush=True)
