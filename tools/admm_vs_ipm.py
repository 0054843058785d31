"""Experiment behind the choice of solver (test infrastructure, numpy only): OSQP-style ADMM
with a cached Cholesky factor, the solver the project brief names, against the interior point
the kernels run, on the benchmark's own condensed QPs.

ADMM: split z = A w with A = [I; D] (D = first differences per control channel), l <= z <= u;
K = H + sigma I + rho A'A factored once per instance, per iteration two triangular solves, a
projection and a dual update, over-relaxation 1.6.  Counted: iterations until the controls are
within 1e-4 / 1e-6 of the minimiser (the oracle's certified answer), for a few rho.
Interior point: the kernel's Mehrotra iteration; counted: iterations until the same distances
(the kernels stop at 5-6 and finish with the active-set rounds).

    python tools/admm_vs_ipm.py > profiles/r2_admm_vs_ipm.txt
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import mpc_oracle as O, orc  # noqa: E402
from tests.helpers import condense_batch, free_controls, make_problem  # noqa: E402


def constraint_matrix(n, lo, hi, rr):
    D = np.zeros((n - 2, n))
    for i in range(n - 2):
        D[i, i + 2], D[i, i] = 1.0, -1.0
    A = np.vstack([np.eye(n), D])
    rrv = np.array([rr[i & 1] for i in range(n - 2)])
    return A, np.concatenate([lo, -rrv]), np.concatenate([hi, rrv])


def admm_iterations(H, g, A, l, u, wstar, rho, tols=(1e-4, 1e-6), max_iter=2000, sigma=1e-6, alpha=1.6):
    n, m = H.shape[0], A.shape[0]
    K = H + sigma * np.eye(n) + rho * A.T @ A
    L = np.linalg.cholesky(K)
    w = np.zeros(n); z = np.clip(A @ w, l, u); y = np.zeros(m)
    hit = [None] * len(tols)
    for it in range(1, max_iter + 1):
        rhs = sigma * w - g + A.T @ (rho * z - y)
        wt = np.linalg.solve(L.T, np.linalg.solve(L, rhs))
        zt = A @ wt
        w = alpha * wt + (1 - alpha) * w
        zr = alpha * zt + (1 - alpha) * z
        zn = np.clip(zr + y / rho, l, u)
        y = y + rho * (zr - zn)
        z = zn
        err = np.abs(w - wstar).max()
        for k, t in enumerate(tols):
            if hit[k] is None and err <= t:
                hit[k] = it
        if hit[-1] is not None:
            break
    return [h if h is not None else max_iter + 1 for h in hit]


def ipm_iterations(H, g, Cm, d, lo, hi, wstar, tols=(1e-4, 1e-6), max_iter=60, mu0=0.1):
    n = g.size; T = n // 2; m = d.size
    w = np.tile(0.5 * (lo[:2] + hi[:2]), T)
    s = np.maximum(d - Cm @ w, 1e-16); lam = mu0 * max(1.0, np.abs(g).max()) / s
    hit = [None] * len(tols)
    for it in range(1, max_iter + 1):
        s = np.maximum(d - Cm @ w, 1e-16)
        K = H + Cm.T @ ((lam / s)[:, None] * Cm)
        mu = float(s @ lam) / m
        dwa = np.linalg.solve(K, -(H @ w + g))
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
        w = w + (eta / rp2 if rp2 > eta else 1.0) * dw
        lam = lam + (eta / rd2 if rd2 > eta else 1.0) * dlam
        err = np.abs(w - wstar).max()
        for k, t in enumerate(tols):
            if hit[k] is None and err <= t:
                hit[k] = it
        if hit[-1] is not None:
            break
    return [h if h is not None else max_iter + 1 for h in hit]


def main():
    B = 48
    print("iterations until max |w - w*| <= 1e-4 / 1e-6 over", B, "instances per variant (median / 90th percentile / max);")
    print("ADMM capped at 2000, interior point at 60; '>' = cap reached\n")
    for name, T in (("diff_drive", 5), ("dsr_ackermann", 5), ("dsr_crab", 5), ("dsr_diff", 5), ("mpc_py", 10), ("dsr_ackermann", 30)):
        v, ov, x0, xref = make_problem(name, B, T=T)
        ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=8)
        ur = np.tile(np.asarray(ov.cold_start, float)[:ov.nu, None], (B, 1, T))
        xb = orc.rollout_batch(ov, x0, ur)
        H, g, lo, hi, rr, feas = condense_batch(ov, xref, xb[:, 2, :T], x0, ur)
        wstar = free_controls(ov, ref["u"])
        n = 2 * T
        rows = {}
        for b in range(B):
            A, l, u = constraint_matrix(n, lo[b], hi[b], rr)
            fin = np.isfinite(l) & np.isfinite(u)
            hs = np.abs(np.diag(H[b])).mean()
            for rho_rel in (0.1, 1.0, 10.0):
                rows.setdefault(f"ADMM rho = {rho_rel:g} x mean diag(H)", []).append(
                    admm_iterations(H[b], g[b], A[fin], l[fin], u[fin], wstar[b], rho_rel * hs))
            Cm, d = O.constraint_rows(lo[b], hi[b], rr)
            rows.setdefault("interior point (kernel's iteration)", []).append(ipm_iterations(H[b], g[b], Cm, d, lo[b], hi[b], wstar[b]))
        print(f"{name} T={T} (n = {n})")
        for label, r in rows.items():
            r = np.array(r)
            cap = 2001 if label.startswith("ADMM") else 61
            fmt = lambda x: (">" if x >= cap else "") + str(int(min(x, cap - 1)))
            cells = []
            for k, t in enumerate(("1e-4", "1e-6")):
                col = r[:, k]
                cells.append(f"{t}: {fmt(np.median(col))} / {fmt(np.quantile(col, 0.9))} / {fmt(col.max())}")
            print(f"  {label:42s} " + "   ".join(cells))
        print()


if __name__ == "__main__":
    main()
