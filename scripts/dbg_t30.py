import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_predictive_control_amd import engine
from oracle import orc
from tests.helpers import make_problem
B = 65536
v, ov, x0, xref = make_problem("dsr_diff", B, seed=37, n_courses=64, T=30)
ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=16)
out = engine.solve_batch(v, x0, xref)
up = out["u"].cpu().numpy()
ref2 = orc.iterative_batch(ov, x0, xref, up, None, nthreads=16)
out2 = engine.solve_batch(v, x0, xref, up)
sav = {}
for label, o, r, ur in (("cold", out, ref, None), ("warm", out2, ref2, up)):
    u = o["u"].cpu().numpy()
    du = np.abs(u - r["u"]).reshape(B, -1).max(axis=1)
    far = np.nonzero((du > 1e-5) & (r["status"] == 0))[0]
    gap = (o["obj"].cpu().numpy() - r["obj"])[far] / np.maximum(1, np.abs(r["obj"][far]))
    print(label, "far", far.size, "du", du[far][:8], "gap", gap[:8], "obj", r["obj"][far][:8], "ref kkt", r["kkt"][far][:8], "iters gpu", o["iters"].cpu().numpy()[far][:8], "ref iters", r["iters"][far][:8])
    sav[label + "_idx"] = far[:16]; sav[label + "_ugpu"] = u[far[:16]]; sav[label + "_uref"] = r["u"][far[:16]]
    sav[label + "_x0"] = x0[far[:16]]; sav[label + "_xref"] = xref[far[:16]]
    sav[label + "_ur"] = (np.tile(np.asarray(ov.cold_start, float)[None, :, None], (min(16, far.size), 1, 30)) if ur is None else ur[far[:16]])
np.savez("gpurun_out/r2b/t30far.npz", **sav)
