"""Would the active-set rounds alone, started from the previous tick's optimum and the constraints
tight at it, settle the next tick's QP?  Takes the benchmark's own warm inputs (closed-loop ticks on
the GPU), condenses the QPs on the host and runs a numpy model of the kernels' rounds from that
guess; prints the distribution of rounds to acceptance, per instance and per group of 32 (one
wavefront of the lane-pair kernel waits for its slowest instance).
python scripts/warm_active_set_study.py [variant] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from model_predictive_control_amd import engine
from oracle import mpc_oracle as O, orc
from tests.helpers import condense_batch, free_controls

name = sys.argv[1] if len(sys.argv) > 1 else "dsr_ackermann"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
bench.VARIANT = name
v, x0, xref, up, hp = bench.build_inputs(0, B)
out = engine.solve_batch(v, x0, xref, up, hp)
ov = O.variant(name)
x0, xref, up, hp = (t.cpu().numpy() for t in (x0, xref, up, hp))
keep = hp != 0
x0, xref, up = x0[keep], xref[keep], up[keep]
ustar = out["u"].cpu().numpy()[keep]
B, T = x0.shape[0], ov.T
n = 2 * T
xb = orc.rollout_batch(ov, x0, up)
H, g, lo, hi, rr, feas = condense_batch(ov, xref, xb[:, 2, :T], x0, up)
w = free_controls(ov, up).copy()
wstar = free_controls(ov, ustar)
rrv = np.array([rr[i & 1] for i in range(n - 2)])
D = np.zeros((n - 2, n))
for i in range(n - 2):
    D[i, i + 2], D[i, i] = 1.0, -1.0
hasr = np.isfinite(rrv)
hasb_hi, hasb_lo = np.isfinite(hi), np.isfinite(lo)
tight = float(os.environ.get("TIGHT", "1e-9"))
w = np.minimum(np.maximum(w, lo), hi)
dv = w @ D.T
au = hasb_hi & (hi - w <= tight); al = hasb_lo & (w - lo <= tight)
bu = hasr[None] & (rrv[None] - dv <= tight); bl = hasr[None] & (rrv[None] + dv <= tight)
# what the optimum's set is, for comparison
dvs = wstar @ D.T
sau = hasb_hi & (hi - wstar <= 1e-9); sal = hasb_lo & (wstar - lo <= 1e-9)
sbu = hasr[None] & (rrv[None] - dvs <= 1e-9); sbl = hasr[None] & (rrv[None] + dvs <= 1e-9)
ndiff = (au != sau).sum(1) + (al != sal).sum(1) + (bu != sbu).sum(1) + (bl != sbl).sum(1)
print(f"{name}: {B} warm instances; constraints active at the optimum: mean {(sau.sum(1) + sal.sum(1) + sbu.sum(1) + sbl.sum(1)).mean():.2f}; "
      f"guess differs from the optimum's set in {np.bincount(ndiff, minlength=6)[:8].tolist()} (0, 1, 2 ... constraints)")
lamU = np.zeros((B, n)); lamL = np.zeros((B, n)); nuU = np.zeros((B, n - 2)); nuL = np.zeros((B, n - 2))
hmax = np.einsum("bii->bi", H).max(1)
rho = 1e5 * hmax
done = np.zeros(B, bool); rounds = np.full(B, 99)
changed = np.ones(B, bool)
R = int(os.environ.get("ROUNDS", "12"))
for rnd in range(R + 1):
    hw = np.einsum("bij,bj->bi", H, w) + g
    dv = w @ D.T
    vu, vl, ru, rl = w - hi, lo - w, dv - rrv, -dv - rrv
    cu, cl, du, dl = np.where(au, vu, 0), np.where(al, vl, 0), np.where(bu, ru, 0), np.where(bl, rl, 0)
    gl = hw + (lamU - lamL) + (nuU - nuL) @ D
    ga = rho[:, None] * ((cu - cl) + (du - dl) @ D)
    viol = np.maximum(np.where(hasb_hi, vu, -1).max(1), np.where(hasb_lo, vl, -1).max(1))
    viol = np.maximum(viol, np.maximum(np.where(hasr[None], ru, -1).max(1), np.where(hasr[None], rl, -1).max(1)))
    cact = np.maximum(np.maximum(np.abs(cu).max(1), np.abs(cl).max(1)), np.maximum(np.abs(du).max(1), np.abs(dl).max(1)))
    neg = np.maximum(np.maximum((-lamU).max(1), (-lamL).max(1)), np.maximum((-nuU).max(1), (-nuL).max(1)))
    stat = np.abs(gl).max(1)
    scale = np.maximum(1.0, np.maximum(np.abs(g).max(1), np.abs(hw - g).max(1))) + 1.0
    ok = ~changed & (stat <= 1e-10 * scale) & (viol <= 1e-12) & (cact <= 1e-12) & (neg <= 1e-9 * scale)
    newly = ok & ~done
    rounds[newly] = rnd
    done |= ok
    if rnd == R:
        break
    K = H + rho[:, None, None] * (np.einsum("bi,ij->bij", (au.astype(float) + al), np.eye(n))
                                   + np.einsum("bk,ki,kj->bij", (bu.astype(float) + bl), D, D))
    step = np.linalg.solve(K, -(gl + ga)[:, :, None])[:, :, 0]
    step[done] = 0.0
    w = w + step
    dv = w @ D.T
    ntol = -1e-9 * scale[:, None]
    vu2, vl2, ru2, rl2 = w - hi, lo - w, dv - rrv, -dv - rrv
    lu = np.where(au, lamU + rho[:, None] * vu2, 0); ll = np.where(al, lamL + rho[:, None] * vl2, 0)
    nu = np.where(bu, nuU + rho[:, None] * ru2, 0); nl = np.where(bl, nuL + rho[:, None] * rl2, 0)
    au2 = hasb_hi & ((au & (lu >= ntol)) | (vu2 > 1e-12)); al2 = hasb_lo & ((al & (ll >= ntol)) | (vl2 > 1e-12))
    bu2 = hasr[None] & ((bu & (nu >= ntol)) | (ru2 > 1e-12)); bl2 = hasr[None] & ((bl & (nl >= ntol)) | (rl2 > 1e-12))
    ch = (au2 != au).any(1) | (al2 != al).any(1) | (bu2 != bu).any(1) | (bl2 != bl).any(1)
    changed = np.where(done, False, ch)
    upd = ~done[:, None]
    au, al, bu, bl = np.where(upd, au2, au), np.where(upd, al2, al), np.where(upd, bu2, bu), np.where(upd, bl2, bl)
    lamU = np.where(upd, np.where(au, np.maximum(lu, 0), 0), lamU); lamL = np.where(upd, np.where(al, np.maximum(ll, 0), 0), lamL)
    nuU = np.where(upd, np.where(bu, np.maximum(nu, 0), 0), nuU); nuL = np.where(upd, np.where(bl, np.maximum(nl, 0), 0), nuL)
print("rounds to acceptance (index = rounds):", np.bincount(np.minimum(rounds, R + 1), minlength=R + 2).tolist(), "(last = not settled)")
err = np.abs(w - wstar).max(1)
print("settled instances: max |w - w*|", err[done].max() if done.any() else None)
per = 32
m = (B // per) * per
gmax = rounds[:m].reshape(-1, per).max(1)
print("per group of 32, slowest instance:", np.bincount(np.minimum(gmax, R + 1), minlength=R + 2).tolist(), "mean", np.minimum(gmax, R + 1).mean())
