"""linear_mpc_control at random linearisation points (controls anywhere inside their bounds, states
rolled from them or perturbed) on random controllers, against the oracle and the certificate.
    python scripts/fuzz_linear.py seed..."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from model_predictive_control_amd import engine
from oracle import mpc_oracle as O, orc
from tests.helpers import make_problem, condense_batch, free_controls, kkt_certificate
from tests.test_gpu_fuzz import BASES, _random_overrides
flagged = 0
for seed in map(int, sys.argv[1:]):
    rng = np.random.default_rng(3000 + seed)
    base = [b for b in BASES if not b.startswith("sim_")][seed % 5]
    kw = _random_overrides(rng, base)
    B = int(rng.choice([1, 64, 130, 1200, 3500]))
    v, ov, x0, xref = make_problem(base, B, seed=80 + seed, **kw)
    T, nu = ov.T, ov.nu
    um = np.minimum(np.asarray(ov.umax, float)[:nu], 2.0)
    uref = rng.uniform(-1, 1, (B, nu, T)) * um[None, :, None]
    if ov.eq_mode == O.EQ_ACKERMANN: uref[:, 2] = uref[:, 0]; uref[:, 3] = -uref[:, 1]
    elif ov.eq_mode == O.EQ_CRAB: uref[:, 2] = uref[:, 0]; uref[:, 3] = uref[:, 1]
    elif ov.eq_mode == O.EQ_FIXED_STEER: uref[:, 1] = ov.steer_fixed; uref[:, 3] = ov.steer_fixed
    xbar = orc.rollout_batch(ov, x0, uref)
    if seed % 2:
        xbar[:, 2] += rng.normal(0, 0.3, xbar[:, 2].shape)     # a yaw the controls would not give
    ref = orc.linear_mpc_batch(ov, xref, xbar, x0, uref, nthreads=8)
    out = engine.linear_mpc_batch(v, xref, xbar, x0, uref)
    st, u = out["status"].cpu().numpy(), out["u"].cpu().numpy()
    du = np.abs(u - ref["u"]).reshape(B, -1).max(axis=1)
    ok = (st <= 1) & (ref["status"] <= 1)
    H, g, lo, hi, rr, feas = condense_batch(ov, xref[ok], xbar[ok][:, 2, :T], x0[ok], uref[ok])
    stat, viol, neg = kkt_certificate(H, g, lo, hi, rr, free_controls(ov, u[ok]))
    line = f"seed {seed} {base} T {T} B {B} status gpu {np.bincount(st, minlength=5).tolist()} ref {np.bincount(ref['status'], minlength=5).tolist()} du {np.nanmax(du[ok]) if ok.any() else 0:.2e} cert {stat.max() if ok.any() else 0:.2e} {viol.max() if ok.any() else 0:.2e}"
    print(line)
    if ((st <= 1) != (ref["status"] <= 1)).any() or (ok.any() and (np.nanmax(du[ok]) > 1e-6 or stat.max() > 1e-8 or viol.max() > 1e-9 or neg.max() > 1e-8)):
        flagged += 1
        print("FLAG", line)
print("flagged", flagged)
