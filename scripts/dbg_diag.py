"""Distribution of interior-point iterations and active-set rounds of the stage-wise kernel
(diagnostic build: MPCX_LIB=model_predictive_control_amd/lib/exp/libmpcx_diag.so)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_predictive_control_amd import engine, variants
from oracle import mpc_oracle as O, orc
from tests.helpers import make_problem
import dataclasses
name, T, B, chain = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
kw = dict(T=T, chain=chain)
for a in sys.argv[5:]:
    k, val = a.split("="); kw[k] = float(val)
v, ov, x0, xref = make_problem(name, B, seed=84, **kw)
out = engine.solve_batch(v, x0, xref)
for label, up in (("cold", None), ("warm", out["u"].clone())):
    o = engine.solve_batch(v, x0, xref, up)
    it = o["iters"].cpu().numpy(); st = o["status"].cpu().numpy()
    r2, rem = it // 10000, it % 10000
    r1, i1 = rem // 100, rem % 100
    print(label, "status", np.bincount(st, minlength=5).tolist(), "| phase-1 rounds hist", np.bincount(r1).tolist(), "| second attempt (rounds+1) hist", np.bincount(r2).tolist(), "| ipm iters mean %.1f max %d" % (i1.mean(), i1.max()))
