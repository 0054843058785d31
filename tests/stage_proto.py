"""Numpy restatement of the stage-wise active-set rounds with exact elimination (test infrastructure).

Per channel and stage a control is FREE, FIXED (box active, or tied forward to a fixed one) or
TIED (rate constraint to the previous stage active: it moves with the previous control).  The
equality-constrained QP on such a working set is solved exactly by a Riccati sweep whose
policy is affine in (dx, v = previous control step); multipliers are recovered from the true
gradient by chain sums.  Checked here against dense algebra before it goes into the kernel.
"""
import sys, os, math
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mpc_oracle as O

FREE, FIXED, TIED = 0, 1, 2


class Prob:
    def __init__(self, v, x0, xref, uref):
        self.v = v
        T = self.T = v.T
        xbar = O.rollout(v, x0, uref)
        E, f = O.control_map(v)
        self.A, self.Bt, self.ct = [], [], []
        for t in range(T):
            a, b, c = O.linearize(v, xbar[2, t], uref[:, t])
            self.A.append(a); self.Bt.append(b @ E); self.ct.append(b @ f + c)
        self.Qt = O.stage_weights(v)
        self.rw, self.rdw, _ = O.free_cost_weights(v)
        self.x0 = np.asarray(x0, float)[:3]
        self.xref = np.asarray(xref, float)
        self.qp = O.condense(v, xref, xbar, x0, uref)
        self.lo = self.qp.lo.reshape(T, 2); self.hi = self.qp.hi.reshape(T, 2); self.rr = self.qp.rr

    def gradient(self, w):
        """H w + g by rollout + adjoint; also per-stage magnitude of the summed terms."""
        T = self.T
        w = w.reshape(T, 2)
        x = np.zeros((T + 1, 3)); x[0] = self.x0
        for t in range(T):
            x[t + 1] = self.A[t] @ x[t] + self.Bt[t] @ w[t] + self.ct[t]
        lam = np.zeros(3); lam_abs = np.zeros(3)
        gr = np.zeros((T, 2)); mag = np.zeros((T, 2))
        for t in range(T - 1, -1, -1):
            qe = 2 * self.Qt[t + 1] * (x[t + 1] - self.xref[:, t + 1])
            # size of what is summed (the difference x - xref cancels: its rounding error is
            # relative to |x| + |xref|, not to the difference)
            qa = 2 * self.Qt[t + 1] * (np.abs(x[t + 1]) + np.abs(self.xref[:, t + 1]))
            if t + 1 < T:
                lam = qe + self.A[t + 1].T @ lam
                lam_abs = qa + np.abs(self.A[t + 1].T) @ lam_abs
            else:
                lam = qe; lam_abs = qa
            gt = 2 * self.rw * w[t] + self.Bt[t].T @ lam
            mt = np.abs(2 * self.rw * w[t]) + np.abs(self.Bt[t].T) @ lam_abs
            if t > 0:
                gt = gt + 2 * self.rdw * (w[t] - w[t - 1]); mt += np.abs(2 * self.rdw * (w[t] - w[t - 1]))
            if t + 1 < T:
                gt = gt - 2 * self.rdw * (w[t + 1] - w[t]); mt += np.abs(2 * self.rdw * (w[t + 1] - w[t]))
            gr[t] = gt; mag[t] = mt
        return gr, mag


def modes_of(T, box, tie):
    """box[t,c] in {0, +1 (upper), -1 (lower)}, tie[t,c] for constraint (t, t+1) in {0, +1, -1}.
    Returns mode[t,c], fixedR[t,c]."""
    fixedR = np.zeros((T, 2), bool)
    mode = np.zeros((T, 2), int)
    for c in range(2):
        for t in range(T - 1, -1, -1):
            fixedR[t, c] = box[t, c] != 0 or (t + 1 < T and tie[t, c] != 0 and fixedR[t + 1, c])
        for t in range(T):
            if fixedR[t, c]:
                mode[t, c] = FIXED
            elif t > 0 and tie[t - 1, c] != 0:
                mode[t, c] = TIED
    return mode, fixedR


def eqp_riccati(P, w, gr, box, tie):
    """One exact Newton step of the equality-constrained QP on the working set: returns new w."""
    T = P.T
    w = w.reshape(T, 2)
    mode, fixedR = modes_of(T, box, tie)
    Pxx = np.diag(2 * P.Qt[T]); Pxv = np.zeros((3, 2)); Pvv = np.zeros((2, 2)); px = np.zeros(3); pv = np.zeros(2)
    Lx, Lv, kk = [None] * T, [None] * T, [None] * T
    tau_next = np.zeros(2)
    for t in range(T - 1, -1, -1):
        B, Am = P.Bt[t], P.A[t]
        Wr = np.diag(2 * P.rw)
        We = np.diag(2 * P.rdw) if t > 0 else np.zeros((2, 2))
        # offsets and sigma
        sig = np.zeros(2); o = np.zeros(2); tau = np.zeros(2)
        for c in range(2):
            if mode[t, c] == FIXED:
                if box[t, c] != 0:
                    tau[c] = P.hi[t, c] if box[t, c] > 0 else P.lo[t, c]
                else:   # tied forward to a fixed stage: w_{t+1} - w_t = +-rr
                    tau[c] = tau_next[c] - tie[t, c] * P.rr[c]
                o[c] = tau[c] - w[t, c]
            elif mode[t, c] == TIED:
                sig[c] = 1.0
                o[c] = tie[t - 1, c] * P.rr[c] - (w[t, c] - w[t - 1, c])
        tau_next = tau
        BP = B.T @ Pxx + Pxv.T
        Huu = Wr + We + BP @ B + B.T @ Pxv + Pvv
        Hux = BP @ Am
        Huv = -We
        hu = gr[t] + B.T @ px + pv         # rhs r = -gr
        free = [c for c in range(2) if mode[t, c] == FREE]
        Hi = np.zeros((2, 2))
        if len(free) == 2:
            Hi = np.linalg.inv(Huu)
        elif len(free) == 1:
            c = free[0]; Hi[c, c] = 1.0 / Huu[c, c]
        S = np.diag(sig)
        lx = Hi @ Hux
        lv = Hi @ (Huv + Huu @ S) - S
        k = Hi @ (hu + Huu @ o) - o
        Lx[t], Lv[t], kk[t] = lx, lv, k
        Mx = Hux - Huu @ lx
        Mv = Huv - Huu @ lv
        Wq = np.diag(2 * P.Qt[t]) if t > 0 else np.zeros((3, 3))
        Hxx = Wq + Am.T @ Pxx @ Am
        nPxx = Hxx - lx.T @ Hux - Mx.T @ lx
        nPxv = -lx.T @ Huv - Mx.T @ lv          # Hxv = 0
        nPvv = We - lv.T @ Huv - Mv.T @ lv
        npx = Am.T @ px - lx.T @ hu - Mx.T @ k
        npv = -lv.T @ hu - Mv.T @ k
        Pxx, Pxv, Pvv, px, pv = nPxx, nPxv, nPvv, npx, npv
    dx = np.zeros(3); vp = np.zeros(2); wn = w.copy()
    for t in range(T):
        u = -(Lx[t] @ dx + Lv[t] @ vp + kk[t])
        wn[t] = w[t] + u
        dx = P.A[t] @ dx + P.Bt[t] @ u
        vp = u
    return wn.reshape(-1)


def eqp_dense(P, w, box, tie):
    """Reference: the same working-set QP by a dense KKT solve (min-norm on dependent rows)."""
    T = P.T; n = 2 * T
    rows, rhs = [], []
    for t in range(T):
        for c in range(2):
            if box[t, c] != 0:
                r = np.zeros(n); r[2 * t + c] = 1.0; rows.append(r); rhs.append(P.hi[t, c] if box[t, c] > 0 else P.lo[t, c])
            if t + 1 < T and tie[t, c] != 0:
                r = np.zeros(n); r[2 * t + 2 + c] = 1.0; r[2 * t + c] = -1.0; rows.append(r); rhs.append(tie[t, c] * P.rr[c])
    H, g = P.qp.H, P.qp.g
    if not rows:
        return np.linalg.solve(H, -g)
    A = np.array(rows); b = np.array(rhs); k = A.shape[0]
    KKT = np.block([[H, A.T], [A, np.zeros((k, k))]])
    sol = np.linalg.lstsq(KKT, np.concatenate([-g, b]), rcond=None)[0]
    return sol[:n]


def recover_multipliers(T, gr, box, tie):
    """mu[t,c] (net box multiplier, sign as the gradient of w_t), nu[t,c] (net multiplier of
    constraint (t,t+1): + on w_{t+1}, - on w_t), resid[t,c] (stationarity residual of the free
    degrees of freedom), from  gr_t + mu_t + nu_{t-1} - nu_t = 0."""
    mode, fixedR = modes_of(T, box, tie)
    mu = np.zeros((T, 2)); nu = np.zeros((T, 2)); resid = np.zeros((T, 2))
    for c in range(2):
        # backward: ties into TIED stages
        for t in range(T - 2, -1, -1):
            if tie[t, c] != 0 and not fixedR[t + 1, c]:
                nxt = nu[t + 1, c] if (t + 2 < T and tie[t + 1, c] != 0) else 0.0
                nu[t, c] = nxt - gr[t + 1, c]
        # forward: fixed chains, boxes, free residuals
        for t in range(T):
            prev = nu[t - 1, c] if (t > 0 and tie[t - 1, c] != 0) else 0.0
            if t + 1 < T and tie[t, c] != 0 and fixedR[t + 1, c]:
                nu[t, c] = gr[t, c] + prev           # mu_t = 0 even if the box is marked too
            elif box[t, c] != 0:
                cur = nu[t, c] if (t + 1 < T and tie[t, c] != 0) else 0.0
                mu[t, c] = -gr[t, c] - prev + cur
            elif mode[t, c] == FREE:
                cur = nu[t, c] if (t + 1 < T and tie[t, c] != 0) else 0.0
                resid[t, c] = gr[t, c] + prev - cur
            # TIED and not box: its equation defined nu[t-1] in the backward pass
    return mu, nu, resid


def rounds(P, w0, box, tie, max_rounds=8, tol=2e-13, verbose=False):
    """Primal-dual active-set rounds.  Returns (w, accepted, n_rounds, n_heavy)."""
    T = P.T
    w = w0.copy()
    changed = True
    viol_new = False
    for rnd in range(max_rounds + 1):
        gr, mag = P.gradient(w)
        mu, nu, resid = recover_multipliers(T, gr, box, tie)
        wt = w.reshape(T, 2)
        # feasibility of everything, on-constraint-ness of the set
        viol = 0.0
        for t in range(T):
            for c in range(2):
                viol = max(viol, wt[t, c] - P.hi[t, c], P.lo[t, c] - wt[t, c])
                if t + 1 < T and math.isfinite(P.rr[c]):
                    viol = max(viol, abs(wt[t + 1, c] - wt[t, c]) - P.rr[c])
        stat_ok = (np.abs(resid) <= tol * mag + 1e-14).all()
        # sign: box upper needs mu >= 0, lower mu <= 0 ; tie +1 (upper) needs nu >= 0, -1 needs nu <= 0
        drop_box = (box != 0) & (mu * box < -tol * mag)
        drop_tie = (tie != 0) & (nu * tie < -tol * np.maximum(mag, np.roll(mag, -1, axis=0)))
        if verbose:
            print(f"  round {rnd}: viol {viol:.2e} stat {np.abs(resid / (mag + 1e-300)).max():.2e} drops {drop_box.sum() + drop_tie.sum()}")
        if rnd > 0 and not changed and stat_ok and viol <= 1e-12 and not drop_box.any() and not drop_tie.any():
            return w, True, rnd, rnd
        if rnd == max_rounds:
            break
        nbox = box.copy(); ntie = tie.copy()
        nbox[drop_box] = 0; ntie[drop_tie] = 0
        # solve the working-set QP
        wn = eqp_riccati(P, w, gr, nbox, ntie)
        wnt = wn.reshape(T, 2)
        # add violated constraints
        for t in range(T):
            for c in range(2):
                if wnt[t, c] - P.hi[t, c] > 1e-12: nbox[t, c] = 1
                if P.lo[t, c] - wnt[t, c] > 1e-12: nbox[t, c] = -1
                if t + 1 < T and math.isfinite(P.rr[c]):
                    d = wnt[t + 1, c] - wnt[t, c]
                    if d - P.rr[c] > 1e-12: ntie[t, c] = 1
                    if -d - P.rr[c] > 1e-12: ntie[t, c] = -1
        changed = (nbox != box).any() or (ntie != tie).any()
        box, tie, w = nbox, ntie, wn
    return w, False, max_rounds, max_rounds


def ipm_dense(P, iters=6, mu0=0.1):
    """The kernel's interior point on the condensed QP (strictly feasible start, Mehrotra),
    `iters` fixed iterations; returns w and the working set lam > s."""
    T = P.T; n = 2 * T
    H, g = P.qp.H, P.qp.g
    Cm, d = O.constraint_rows(P.qp.lo, P.qp.hi, P.qp.rr)
    m = d.size
    wc = 0.5 * (P.lo[0] + P.hi[0])
    w = np.tile(wc, T)
    s = np.maximum(d - Cm @ w, 1e-16)
    lam = mu0 / s
    for it in range(iters):
        s = np.maximum(d - Cm @ w, 1e-16)
        K = H + Cm.T @ ((lam / s)[:, None] * Cm)
        hw = H @ w + g
        mu = float(s @ lam) / m
        dwa = np.linalg.solve(K, -(hw))          # predictor: rhs -(Hw+g) (the C'lam term cancels with mu=0 target)
        x = (Cm @ dwa) / s
        rp = max(x.max(), 0.0); rdm = max((1 - x).max(), 1.0)
        apa = 1 / rp if rp > 1 else 1.0; ada = 1 / rdm
        muaff = float(((s * (1 - apa * x)) * (lam * (1 + ada * (x - 1)))).sum()) / m
        sm = (muaff / max(mu, 1e-300)) ** 3 * mu
        q = sm / s + lam * x * (x - 1)
        dw = dwa + np.linalg.solve(K, -(Cm.T @ q))
        z = (Cm @ dw) / s
        dlam = lam * z + q - lam
        rp2 = max(z.max(), 0.0); rd2 = max((-dlam / lam).max(), 0.0)
        eta = min(0.999, max(0.99, 1 - mu))
        ap = eta / rp2 if rp2 > eta else 1.0
        ad = eta / rd2 if rd2 > eta else 1.0
        w = w + ap * dw
        lam = lam + ad * dlam
    s = d - Cm @ w
    act = lam > s
    return w, act, Cm, d


def sets_from_mask(P, act):
    """Constraint-row order of O.constraint_rows -> box/tie arrays."""
    T = P.T
    box = np.zeros((T, 2), int); tie = np.zeros((T, 2), int)
    j = 0
    lo, hi = P.qp.lo, P.qp.hi
    for i in range(2 * T):
        if math.isfinite(hi[i]):
            if act[j]: box[i // 2, i % 2] = 1
            j += 1
        if math.isfinite(lo[i]):
            if act[j]: box[i // 2, i % 2] = -1
            j += 1
    for t in range(T - 1):
        for c in range(2):
            if math.isfinite(P.rr[c]):
                if act[j]: tie[t, c] = 1
                j += 1
                if act[j]: tie[t, c] = -1
                j += 1
    return box, tie


def sets_from_point(P, w, tol=1e-9):
    T = P.T
    wt = w.reshape(T, 2)
    box = np.zeros((T, 2), int); tie = np.zeros((T, 2), int)
    for t in range(T):
        for c in range(2):
            if P.hi[t, c] - wt[t, c] <= tol: box[t, c] = 1
            elif wt[t, c] - P.lo[t, c] <= tol: box[t, c] = -1
            if t + 1 < T and math.isfinite(P.rr[c]):
                d = wt[t + 1, c] - wt[t, c]
                if P.rr[c] - d <= tol: tie[t, c] = 1
                elif P.rr[c] + d <= tol: tie[t, c] = -1
    return box, tie


def free_of(v, u):
    if v.eq_mode == O.EQ_NONE: return np.stack([u[0], u[1]], 1).reshape(-1)
    if v.eq_mode == O.EQ_FIXED_STEER: return np.stack([u[0], u[2]], 1).reshape(-1)
    return np.stack([u[0], u[1]], 1).reshape(-1)


if __name__ == "__main__":
    from tests.helpers import make_problem
    name = sys.argv[1] if len(sys.argv) > 1 else "dsr_ackermann"
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    mode = sys.argv[4] if len(sys.argv) > 4 else "cold"
    _, ov, x0, xref = make_problem(name, B, T=T)
    nr, acc, errs, eq_err = [], [], [], []
    for b in range(B):
        ucold = np.tile(np.asarray(ov.cold_start, float)[:ov.nu, None], (1, T))
        ref_cold = O.iterative_linear_mpc_control(ov, xref[b], x0[b], None)
        if mode == "cold":
            P = Prob(ov, x0[b], xref[b], ucold)
            ref = ref_cold
            w0, act, Cm, d = ipm_dense(P, 6)
            box, tie = sets_from_mask(P, act)
        elif mode.startswith("warmipm"):
            P = Prob(ov, x0[b], xref[b], ref_cold.u)
            ref = O.iterative_linear_mpc_control(ov, xref[b], x0[b], ref_cold.u)
            w0, act, Cm, d = ipm_dense(P, int(mode[7:] or 6))
            box, tie = sets_from_mask(P, act)
        else:
            P = Prob(ov, x0[b], xref[b], ref_cold.u)
            ref = O.iterative_linear_mpc_control(ov, xref[b], x0[b], ref_cold.u)
            w0 = free_of(ov, ref_cold.u)
            w0 = np.clip(w0, P.qp.lo, P.qp.hi)
            box, tie = sets_from_point(P, w0)
        # check the riccati step against dense on this set
        gr, _ = P.gradient(w0)
        wr = eqp_riccati(P, w0, gr, box, tie)
        wd = eqp_dense(P, w0, box, tie)
        eq_err.append(np.abs(wr - wd).max())
        w, ok, r, _ = rounds(P, w0, box, tie, verbose=(b < 3))
        nr.append(r); acc.append(ok); errs.append(np.abs(w - ref.w).max())
    print(name, T, mode, "accepted", np.mean(acc), "rounds mean/max", np.mean(nr), np.max(nr),
          "err vs oracle max/median", np.max(errs), np.median(errs), "riccati-vs-dense EQP", np.max(eq_err))
