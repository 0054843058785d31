"""Debug aid for the four-lane kernel: instance 0's condensed QP as the test helpers build it, next
to what a traced build of the kernel prints.  usage: MPCX_LIB=.../libmpcx_qtrace.so quad_debug.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from model_predictive_control_amd import engine
from oracle import mpc_oracle as O, orc
from tests.helpers import make_problem, condense_batch
v, ov, x0, xref = make_problem("mpc_py", 16, seed=61)
T = ov.T
cold = np.tile(np.asarray(ov.cold_start)[:ov.nu, None], (1, T))
ur = np.tile(cold[None], (16, 1, 1))
xb = orc.rollout_batch(ov, x0, ur)
H, g, lo, hi, rr, feas = condense_batch(ov, xref, xb[:, 2, :T], x0, ur)
print("ref H00 %.12g H10 %.12g H20 %.12g H30 %.12g H11 %.12g H21 %.12g H22(=blk11,00) %.12g  H[18,18] %.12g H[19,19] %.12g" % (H[0,0,0], H[0,1,0], H[0,2,0], H[0,3,0], H[0,1,1], H[0,2,1], H[0,2,2], H[0,18,18], H[0,19,19]))
print("ref g0 %.12g g1 %.12g g2 %.12g g3 %.12g g18 %.12g g19 %.12g" % tuple(g[0, [0,1,2,3,18,19]]))
print("ref lo", lo[0,:4], "hi", hi[0,:4], "rr", rr)
out = engine.solve_batch(v, x0, xref)
torch.cuda.synchronize()
ref = orc.iterative_batch(ov, x0, xref)
print("status", out["status"].cpu().numpy(), "iters", out["iters"].cpu().numpy())
print("du", np.abs(out["u"].cpu().numpy() - ref["u"]).reshape(16, -1).max(axis=1))
