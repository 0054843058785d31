"""Stress inputs (vehicles far off course, any heading, random previous controls) through the
stage-wise kernel: instances whose controls differ from the oracle's by more than 1e-5, and the
solver-independent KKT certificate of both answers.  python scripts/stress_certificate.py name T"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_predictive_control_amd import engine
from oracle import mpc_oracle as O, orc
from tests.helpers import make_problem, condense_batch, free_controls, kkt_certificate
name, T = sys.argv[1], int(sys.argv[2])
B = 4000
rng = np.random.default_rng(99)
v, ov, x0, xref = make_problem(name, B, seed=77, T=T)
x0[:, :2] += rng.normal(0, 3.0, (B, 2))
x0[:, 2] = rng.uniform(-np.pi, np.pi, B) + rng.integers(-2, 3, B) * 2 * np.pi
um = np.minimum(np.asarray(ov.umax)[:ov.nu], 2.0)
up = rng.uniform(-1, 1, (B, ov.nu, T)) * um[None, :, None]
if ov.eq_mode == O.EQ_ACKERMANN:
    up[:, 2] = up[:, 0]; up[:, 3] = -up[:, 1]
hp = (rng.random(B) < 0.7).astype(np.uint8)
ref = orc.iterative_batch(ov, x0, xref, up, hp, nthreads=8)
out = engine.solve_batch(v, x0, xref, up, hp)
u = out["u"].cpu().numpy(); st = out["status"].cpu().numpy(); it = out["iters"].cpu().numpy()
du = np.abs(u - ref["u"]).reshape(B, -1).max(axis=1)
bad = np.nonzero(du > 1e-5)[0]
print("bad", bad, du[bad], "status", st[bad], "iters", it[bad], "ref status", ref["status"][bad], "obj gpu/ref", out["obj"].cpu().numpy()[bad], ref["obj"][bad])
cold = np.tile(np.asarray(ov.cold_start, float)[:ov.nu, None], (1, T))
ur = up.copy(); ur[hp == 0] = cold
xb = orc.rollout_batch(ov, x0, ur)
H, g, lo, hi, rr, feas = condense_batch(ov, xref, xb[:, 2, :T], x0, ur)
for nm, uu in (("gpu", u), ("ref", ref["u"])):
    stat, viol, neg = kkt_certificate(H, g, lo, hi, rr, free_controls(ov, uu))
    print(nm, "cert worst stat", stat.max(), "viol", viol.max(), "neg", neg.max(), " at bad:", stat[bad], viol[bad], neg[bad])
np.set_printoptions(precision=5, suppress=True, linewidth=200)
for b in bad[:2]:
    wg = free_controls(ov, u[b:b+1])[0].reshape(T, 2); wr = free_controls(ov, ref["u"][b:b+1])[0].reshape(T, 2)
    print("inst", b, "hp", hp[b]); print("gpu ch0", wg[:, 0]); print("ref ch0", wr[:, 0]); print("gpu ch1", wg[:, 1]); print("ref ch1", wr[:, 1])
    print("lo0 hi0", lo[b, :2], hi[b, :2], "rr", rr)
os.makedirs("gpurun_out/r2b", exist_ok=True)
np.savez("gpurun_out/r2b/bad.npz", bad=bad, u=u[bad], uref=ref["u"][bad], x0=x0[bad], xref=xref[bad], ur=ur[bad])
