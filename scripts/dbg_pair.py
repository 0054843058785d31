import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_predictive_control_amd import engine
from oracle import orc
from tests.helpers import make_problem, condense_batch, free_controls, kkt_certificate
name = sys.argv[1]; B = int(sys.argv[2])
v, ov, x0, xref = make_problem(name, B, chain=1)
import dataclasses
ov = dataclasses.replace(ov, chain=1)
ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=8)
out = engine.solve_batch(v, x0, xref)
it = out["iters"].cpu().numpy(); st = out["status"].cpu().numpy(); u = out["u"].cpu().numpy()
print("status", np.bincount(st, minlength=5), "iters code hist", {int(k): int(c) for k, c in zip(*np.unique(it // 100 * 100, return_counts=True))})
T = ov.T
ur = np.tile(np.asarray(ov.cold_start, float)[None, :ov.nu, None], (B, 1, T))
xb = orc.rollout_batch(ov, x0, ur)
H, g, lo, hi, rr, feas = condense_batch(ov, xref, xb[:, 2, :T], x0, ur)
ok = st <= 1
stat, viol, neg = kkt_certificate(H[ok], g[ok], lo[ok], hi[ok], rr, free_controls(ov, u[ok]))
bad = np.nonzero(ok)[0][(stat > 1e-8) | (neg > 1e-8)]
print("cert bad", bad.size, "codes", it[bad][:10], "du", np.abs(u - ref["u"]).reshape(B, -1).max(1)[bad][:10])
np.set_printoptions(precision=5, suppress=True, linewidth=200)
for b in bad[:3]:
    wg = free_controls(ov, u[b:b+1])[0].reshape(T, 2); wr = free_controls(ov, ref["u"][b:b+1])[0].reshape(T, 2)
    print("inst", b, "code", it[b]); print(" gpu", wg.T); print(" ref", wr.T); print(" lo0 hi0", lo[b, :2], hi[b, :2], "rr", rr)
out2 = engine.solve_batch(v, x0, xref, out["u"].clone())
it2 = out2["iters"].cpu().numpy(); st2 = out2["status"].cpu().numpy()
print("warm status", np.bincount(st2, minlength=5), "code hist", {int(k): int(c) for k, c in zip(*np.unique(it2 // 100 * 100, return_counts=True))})
