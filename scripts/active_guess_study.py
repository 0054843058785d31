"""The first guess of the active set that the rounds start from: constraints whose multiplier exceeds
GUESS times their slack at the interior-point iterate (the kernels use 1).  Same numpy model as
scripts/warm_ipm_study.py; rounds of the slowest instance per group of 32 for several GUESS values.
python scripts/active_guess_study.py [variant] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from model_predictive_control_amd import engine
from oracle import mpc_oracle as O, orc
from tests.helpers import condense_batch, free_controls

name = sys.argv[1] if len(sys.argv) > 1 else "dsr_ackermann"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
bench.VARIANT = name
v, x0, xref, up, hp = bench.build_inputs(0, B)
out = engine.solve_batch(v, x0, xref, up, hp)
ov = O.variant(name)
x0, xref, up, hp = (t.cpu().numpy() for t in (x0, xref, up, hp))
keep = hp != 0
x0, xref, up = x0[keep], xref[keep], up[keep]
wstar = free_controls(ov, out["u"].cpu().numpy()[keep])
B, T = x0.shape[0], ov.T
n = 2 * T
xb = orc.rollout_batch(ov, x0, up)
H, g, lo, hi, rr, feas = condense_batch(ov, xref, xb[:, 2, :T], x0, up)
wprev = free_controls(ov, up)
# constraint rows C w <= d: box upper, box lower, rate upper, rate lower
D = np.zeros((n - 2, n))
for i in range(n - 2):
    D[i, i + 2], D[i, i] = 1.0, -1.0
rrv = np.array([rr[i & 1] for i in range(n - 2)])
C = np.concatenate([np.eye(n), -np.eye(n), D, -D])
d = np.concatenate([hi, -lo, np.tile(rrv, (B, 1)), np.tile(rrv, (B, 1))], axis=1)
has = np.isfinite(d)
d = np.where(has, d, 1e300)
m = has.sum(1)
kMu0, kSlackMin, kEtaCap = 0.1, 1e-16, 0.999


def ipm(w, lam, iters):
    """`iters` predictor-corrector iterations from (w, lam); returns the iterate."""
    w, lam = w.copy(), lam.copy()
    for _ in range(iters):
        s = np.maximum(d - w @ C.T, kSlackMin)
        hw = np.einsum("bij,bj->bi", H, w) + g
        isl = np.where(has, 1.0 / s, 0.0)
        mu = (s * lam * has).sum(1) / m
        K = H + np.einsum("kj,bk,ki->bji", C, lam * isl, C)
        dwa = np.linalg.solve(K, -hw[:, :, None])[:, :, 0]
        x = (dwa @ C.T) * isl
        rp = np.maximum(np.where(has, x, 0).max(1), 0.0)
        rdm = np.maximum(1.0, np.where(has, 1.0 - x, 1.0).max(1))
        apa = np.where(rp > 1.0, 1.0 / np.maximum(rp, 1e-300), 1.0); ada = 1.0 / rdm
        back = 1.0 - np.minimum(apa, ada)
        sm = back ** 3 * mu
        q = np.where(has, sm[:, None] * isl + lam * x * (x - 1.0), 0.0)
        dw = dwa + np.linalg.solve(K, -(q @ C)[:, :, None])[:, :, 0]
        z = (dw @ C.T) * isl
        dlam = np.where(has, lam * z + q - lam, 0.0)
        rp2 = np.where(has, z, 0).max(1)
        rd2 = np.where(has & (lam > 0), -dlam / np.where(lam > 0, lam, 1.0), 0).max(1)
        eta = np.minimum(kEtaCap, np.maximum(0.99, 1.0 - mu))
        ap = np.where(rp2 > eta, eta / np.maximum(rp2, 1e-300), 1.0)
        ad = np.where(rd2 > eta, eta / np.maximum(rd2, 1e-300), 1.0)
        w = w + ap[:, None] * dw
        lam = lam + ad[:, None] * dlam
    return w, lam


GUESS = 1.0


def rounds(w, lam, R=6):
    """The active-set rounds from an interior-point iterate; returns rounds to acceptance (99: none)."""
    w = w.copy()
    s = d - w @ C.T
    act = has & (lam > GUESS * s)
    lam = np.where(act, lam, 0.0)
    hmax = np.einsum("bii->bi", H).max(1)
    rho = 1e5 * hmax
    done = np.zeros(B, bool); took = np.full(B, 99); changed = np.ones(B, bool)
    wmaxb = np.abs(np.where(np.isfinite(hi), hi, 0)).max(1)
    for rnd in range(R + 1):
        hw = np.einsum("bij,bj->bi", H, w) + g
        v_ = w @ C.T - d
        cact = np.abs(np.where(act, v_, 0)).max(1)
        viol = np.where(has, v_, -1).max(1)
        gl = hw + lam @ C
        ga = rho[:, None] * (np.where(act, v_, 0) @ C)
        stat = np.abs(gl).max(1)
        scale = np.maximum(1.0, np.maximum(np.abs(g).max(1), np.abs(hw - g).max(1))) + 1.0
        tol = 1e-10 * np.maximum(scale, 4e-5 * rho * wmaxb)
        ok = ~changed & (stat <= tol) & (viol <= 1e-12) & (cact <= 1e-12) & ((-lam).max(1) <= 1e-9 * scale)
        took[ok & ~done] = rnd
        done |= ok
        if rnd == R:
            break
        K = H + rho[:, None, None] * np.einsum("kj,bk,ki->bji", C, act.astype(float), C)
        step = np.linalg.solve(K, -(gl + ga)[:, :, None])[:, :, 0]
        step[done] = 0.0
        w = w + step
        v2 = w @ C.T - d
        ln = np.where(act, lam + rho[:, None] * v2, 0.0)
        act2 = has & ((act & (ln >= -1e-9 * scale[:, None])) | (v2 > 1e-12))
        ch = (act2 != act).any(1)
        changed = np.where(done, False, ch)
        upd = ~done[:, None]
        act = np.where(upd, act2, act)
        lam = np.where(upd, np.where(act, np.maximum(ln, 0), 0), lam)
    return took, w


def report(tag, w0, lam0):
    print(tag)
    for k in range(1, 8):
        w, lam = ipm(w0, lam0, k)
        took, wf = rounds(w, lam)
        ok = took < 99
        per = 32
        mm = (B // per) * per
        gmax = took[:mm].reshape(-1, per).max(1)
        err = np.abs(wf - wstar).max(1)
        print(f"  {k} iterations: settled {ok.mean() * 100:.2f} %, rounds mean {took[ok].mean():.2f}; groups of 32 all settled {100 * (gmax < 99).mean():.1f} %, "
              f"their slowest instance {gmax[gmax < 99].mean() if (gmax < 99).any() else float('nan'):.2f} rounds; max |w - w*| of the settled {err[ok].max():.1e}")





MU0 = 0.05
centre = np.where(np.isfinite(lo[:, :2] + hi[:, :2]), 0.5 * (lo[:, :2] + hi[:, :2]), 0.0)
wc = np.tile(centre, (1, T))
gabs = np.abs(g).max(1)
s0 = np.maximum(d - wc @ C.T, kSlackMin)
lam0 = np.where(has, (MU0 * np.maximum(1.0, gabs))[:, None] / s0, 0.0)
w5, lam5 = ipm(wc, lam0, 5)
per = 32
mm = (B // per) * per
for gv in (0.01, 0.1, 0.3, 1.0, 3.0, 10.0, 100.0):
    GUESS = gv
    took, _ = rounds(w5, lam5)
    gmax = took[:mm].reshape(-1, per).max(1)
    ok = gmax < 99
    print(f"{name}: guess = multiplier > {gv:g} x slack: instances settled {100 * (took < 99).mean():.3f} %, rounds mean {took[took < 99].mean():.2f}; "
          f"groups of 32 all settled {100 * ok.mean():.1f} %, slowest instance {gmax[ok].mean():.2f} rounds", flush=True)
