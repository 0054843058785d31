"""The stage-wise (Riccati) linear algebra of the long-horizon kernel, restated in numpy and
checked against dense linear algebra on the oracle's condensed QP: K dw = r solved stage by
stage equals numpy.linalg.solve on K = H + box terms + rate terms, and the rollout/adjoint
gradient equals H w + g.  This pins the claim the kernel rests on (same Newton steps as the
dense solver, O(T) work); the kernel itself is checked against the oracle on the GPU."""
import numpy as np
import pytest

from oracle import mpc_oracle as O
from tests.helpers import make_problem


def stage_data(v, xbar, uref):
    E, f = O.control_map(v)
    A, Bt, ct = [], [], []
    for t in range(v.T):
        a, b, c = O.linearize(v, xbar[2, t], uref[:, t])
        A.append(a); Bt.append(b @ E); ct.append(b @ f + c)
    return A, Bt, ct


def riccati_solve(v, A, Bt, Db, Dr, rhs):
    """argmin 1/2 dw'K dw - rhs'dw, K = H + diag(Db) + sum_i Dr_i (e_{i+2} - e_i)(e_{i+2} - e_i)',
    by the recursion of csrc/mpcx_stage.h: value function over (dx_t, dw_{t-1})."""
    T = v.T
    Qt = O.stage_weights(v)
    rw, rdw, _ = O.free_cost_weights(v)
    Pxx = np.diag(2 * Qt[T]); Pxu = np.zeros((3, 2)); Puu = np.zeros((2, 2)); px = np.zeros(3); pu = np.zeros(2)
    Lx, Lv, k = [None] * T, [None] * T, [None] * T
    for t in range(T - 1, -1, -1):
        Wr = np.diag(2 * rw + Db[2 * t:2 * t + 2])
        We = np.diag(2 * rdw + Dr[2 * (t - 1):2 * (t - 1) + 2]) if t > 0 else np.zeros((2, 2))
        B, Am = Bt[t], A[t]
        BP = B.T @ Pxx + Pxu.T
        Huu = Wr + We + BP @ B + B.T @ Pxu + Puu
        Hux = BP @ Am
        Huv = -We
        hu = -rhs[2 * t:2 * t + 2] + B.T @ px + pu
        Hi = np.linalg.inv(Huu)
        Lx[t], Lv[t], k[t] = Hi @ Hux, Hi @ Huv, Hi @ hu
        Wq = np.diag(2 * Qt[t]) if t > 0 else np.zeros((3, 3))
        Pxx, Pxu, Puu, px, pu = (Wq + Am.T @ Pxx @ Am - Hux.T @ Lx[t], -Hux.T @ Lv[t], We - Huv.T @ Lv[t],
                                 Am.T @ px - Lx[t].T @ hu, -Lv[t].T @ hu)
    dx = np.zeros(3); vprev = np.zeros(2); out = np.zeros(2 * T)
    for t in range(T):
        u = -(Lx[t] @ dx + Lv[t] @ vprev + k[t])
        out[2 * t:2 * t + 2] = u
        dx = A[t] @ dx + Bt[t] @ u
        vprev = u
    return out


def gradient(v, A, Bt, ct, x0, xref, w):
    """H w + g by a rollout and an adjoint sweep."""
    T = v.T
    Qt = O.stage_weights(v)
    rw, rdw, _ = O.free_cost_weights(v)
    x = np.zeros((T + 1, 3)); x[0] = x0
    for t in range(T):
        x[t + 1] = A[t] @ x[t] + Bt[t] @ w[2 * t:2 * t + 2] + ct[t]
    lam = np.zeros(3); gr = np.zeros(2 * T)
    for t in range(T - 1, -1, -1):
        lam = 2 * Qt[t + 1] * (x[t + 1] - xref[:, t + 1]) + (A[t + 1].T @ lam if t + 1 < T else 0)
        wt = w[2 * t:2 * t + 2]
        gt = 2 * rw * wt + Bt[t].T @ lam
        if t > 0:
            gt += 2 * rdw * (wt - w[2 * t - 2:2 * t])
        if t + 1 < T:
            gt -= 2 * rdw * (w[2 * t + 2:2 * t + 4] - wt)
        gr[2 * t:2 * t + 2] = gt
    return gr


@pytest.mark.parametrize("name,T,tol", [("dsr_ackermann", 30, 1e-10), ("dsr_crab", 17, 1e-10), ("diff_drive", 12, 1e-10),
                                         ("mpc_py", 10, 1e-10), ("diff_drive", 1, 1e-12), ("dsr_diff", 30, 1e-6)])
def test_stagewise_solve_and_gradient_equal_dense(name, T, tol):
    v, ov, x0, xref = make_problem(name, 3, T=T)
    rng = np.random.default_rng(T)
    for b in range(3):
        uref = np.tile(np.asarray(ov.cold_start, dtype=float)[:ov.nu, None], (1, T))
        uref[0] += rng.uniform(-0.2, 0.2, T)
        xbar = O.rollout(ov, x0[b], uref)
        qp = O.condense(ov, xref[b], xbar, x0[b], uref)
        A, Bt, ct = stage_data(ov, xbar, uref)
        n = 2 * T
        Db = rng.uniform(0, 1, n) * 10.0 ** rng.integers(-3, 6, n)          # barrier / penalty weights of any size
        Dr = rng.uniform(0, 1, max(n - 2, 0)) * 10.0 ** rng.integers(-3, 6, max(n - 2, 0))
        K = qp.H + np.diag(Db)
        for i in range(n - 2):
            r = np.zeros(n); r[i + 2] = 1.0; r[i] = -1.0
            K += Dr[i] * np.outer(r, r)
        rhs = rng.normal(size=n)
        ref = np.linalg.solve(K, rhs)
        got = riccati_solve(ov, A, Bt, Db, Dr, rhs)
        assert np.abs(got - ref).max() <= tol * np.abs(ref).max()
        w = rng.normal(size=n) * 0.3
        g_dense = qp.H @ w + qp.g
        g_sweep = gradient(ov, A, Bt, ct, x0[b], xref[b], w)
        assert np.abs(g_dense - g_sweep).max() <= 1e-11 * max(1.0, np.abs(g_dense).max())
