"""The reference's CPU path as a timed baseline: cvxpy + ECOS, the problem object rebuilt from
scratch on every call as the reference does (mpc_diff_drive.py:172-200 and siblings), written
from this repository's own variant table (oracle/mpc_oracle.py), not from the reference's files.

Runs only where `import cvxpy` succeeds (it does not in the build image: no network); otherwise
reports that it is unavailable.  Part of the benchmark's cpu_baseline leg and usable on its own:

    python tools/cvxpy_baseline.py [variant] [n_solves]
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

UNAVAILABLE = "cvxpy baseline unavailable offline"


def solve_one(cp, v, O, xref, x0, uref, solver):
    """One linear_mpc_control call in the reference's formulation: variables x (3, T+1) and
    u (NU, T), stage costs from t = 1, dynamics linearised at (xbar, uref), box / rate /
    first-step constraints and the mode's equalities between control rows."""
    T, NU = v.T, v.nu
    xbar = O.rollout(v, x0, uref)
    x = cp.Variable((3, T + 1))
    u = cp.Variable((NU, T))
    Qt = O.stage_weights(v)
    cost = 0.0
    cons = [x[:, 0] == np.asarray(x0, dtype=float)[:3]]
    for t in range(T):
        cost += cp.quad_form(u[:, t], np.diag(v.R))
        if t != 0:
            cost += cp.quad_form(xref[:, t] - x[:, t], np.diag(Qt[t]))
        A, B, C = O.linearize(v, xbar[2, t], uref[:, t])
        cons.append(x[:, t + 1] == A @ x[:, t] + B @ u[:, t] + C)
        if t < T - 1:
            if any(v.Rd):
                cost += cp.quad_form(u[:, t + 1] - u[:, t], np.diag(v.Rd))
            for r in range(NU):
                if np.isfinite(v.dumax[r]):
                    cons.append(cp.abs(u[r, t + 1] - u[r, t]) <= v.dumax[r])
    cost += cp.quad_form(xref[:, T] - x[:, T], np.diag(v.Qf))
    for r in range(NU):
        if np.isfinite(v.umax[r]):
            cons.append(cp.abs(u[r, :]) <= v.umax[r])
        if np.isfinite(v.first_step[r]):
            cons.append(cp.abs(uref[r, 0] - u[r, 0]) <= v.first_step[r])
    if v.eq_mode == O.EQ_ACKERMANN:
        cons += [u[1, :] == -u[3, :], u[0, :] == u[2, :]]
    elif v.eq_mode == O.EQ_CRAB:
        cons += [u[1, :] == u[3, :], u[0, :] == u[2, :]]
    elif v.eq_mode == O.EQ_FIXED_STEER:
        cons += [u[1, :] == v.steer_fixed, u[3, :] == v.steer_fixed]
    prob = cp.Problem(cp.Minimize(cost), cons)
    prob.solve(solver=solver, verbose=False)
    if prob.status not in (cp.OPTIMAL, cp.OPTIMAL_INACCURATE):
        return None
    return np.asarray(u.value)


def run(variant: str, x0, xref, uprev=None, has_prev=None, n_solves: int = 200):
    """Times n_solves iterative_linear_mpc_control calls (one QP each) on one core and compares
    the controls with the oracle's; a dict, or the UNAVAILABLE string when cvxpy is missing."""
    try:
        import cvxpy as cp
    except Exception:
        return UNAVAILABLE
    from oracle import mpc_oracle as O
    v = O.variant(variant)
    solver = cp.ECOS if "ECOS" in cp.installed_solvers() else None
    n = min(n_solves, x0.shape[0])
    cold = np.tile(np.asarray(v.cold_start, dtype=float)[:v.nu, None], (1, v.T))
    worst, solved, dt = 0.0, 0, 0.0
    for b in range(n):
        warm = uprev is not None and (has_prev is None or has_prev[b])
        uref = np.asarray(uprev[b], dtype=float) if warm else cold
        t0 = time.perf_counter()
        u = solve_one(cp, v, O, xref[b], x0[b], uref, solver)     # problem construction + solve, as the reference does per call
        dt += time.perf_counter() - t0
        if u is None:
            continue
        solved += 1
        if v.chain <= 1:
            ref = O.iterative_linear_mpc_control(v, xref[b], x0[b], uref if warm else None)
            if ref.u is not None:
                worst = max(worst, float(np.abs(ref.u - u).max()))
    return {"value": n / dt, "unit": "solves/s", "cores": 1, "solves": n, "solved": solved, "seconds": dt,
            "solver": "ECOS" if solver else "cvxpy default", "max_abs_du_vs_oracle": worst,
            "note": "problem rebuilt per call like the reference; only the cvxpy construction + solve is timed (the check against the oracle is not)"}


if __name__ == "__main__":
    from model_predictive_control_amd import workload
    from oracle import mpc_oracle as O
    name = sys.argv[1] if len(sys.argv) > 1 else "dsr_ackermann"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    v = O.variant(name)
    courses = workload.make_courses(16)
    inst = workload.make_instances(courses, n, seed=7)
    xref = workload.reference_window(courses, inst, v.T)
    print(run(name, inst["state"][:, :3].copy(), xref, n_solves=n))
