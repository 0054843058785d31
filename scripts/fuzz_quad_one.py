"""One seed of scripts/fuzz_quad.py in detail: statuses, the instances whose certificate fails, their
oracle status and distance.  python scripts/fuzz_quad_one.py seed"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_predictive_control_amd import engine
from oracle import orc
from tests.helpers import make_problem, condense_batch, free_controls, kkt_certificate
from tests.test_gpu_fuzz import _horizon_10_case
seed = int(sys.argv[1])
base, kw, B = _horizon_10_case(seed)
print(base, B, kw)
v, ov, x0, xref = make_problem(base, B, seed=90 + seed, **kw)
T = ov.T
ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=16)
out = engine.solve_batch(v, x0, xref)
hp = (np.arange(B) % 3 != 0).astype(np.uint8)
up = np.where(np.isnan(ref["u"]), 0.0, ref["u"])
ref2 = orc.iterative_batch(ov, x0, xref, up, hp, nthreads=16)
out2 = engine.solve_batch(v, x0, xref, up, hp)
cold = np.tile(np.asarray(ov.cold_start, float)[:ov.nu, None], (1, T))
for label, o, r, ur in (("cold", out, ref, np.tile(cold[None], (B, 1, 1))), ("warm", out2, ref2, np.where(hp[:, None, None] != 0, up, cold[None]))):
    st = o["status"].cpu().numpy(); u = o["u"].cpu().numpy(); it = o["iters"].cpu().numpy()
    print(label, "status gpu", np.bincount(st, minlength=5).tolist(), "oracle", np.bincount(r["status"], minlength=5).tolist())
    ok = st <= 1
    xb = orc.rollout_batch(ov, x0[ok], ur[ok])
    H, g, lo, hi, rr, feas = condense_batch(ov, xref[ok], xb[:, 2, :T], x0[ok], ur[ok])
    stat, viol, neg = kkt_certificate(H, g, lo, hi, rr, free_controls(ov, u[ok]))
    idx = np.nonzero(ok)[0]
    badk = np.nonzero((stat > 1e-8) | (neg > 1e-8) | (viol > 1e-9))[0]
    for k in badk[:10]:
        b = idx[k]
        print("  instance", b, "gpu status", st[b], "iters", it[b], "stat %.2e viol %.2e neg %.2e" % (stat[k], viol[k], neg[k]), "oracle status", r["status"][b],
              "du %.2e" % np.abs(u[b] - r["u"][b]).max(), "cond(H) %.2e" % np.linalg.cond(H[k]))
