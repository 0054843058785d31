"""Runs a solver's answers against the reference's own (x, u) programme (tests/xu_problem.py).
`ops` adapts either the HIP engine or the oracle: rollout(x0, u) -> xbar, linearize(theta, u) ->
(A, B, C), solve(x0, xref, uprev) and linear(xref, xbar, x0, uref) -> dict of numpy arrays
u, x, obj, status.  Test infrastructure."""
import numpy as np

from tests import xu_problem as XU

U_BAR = 1e-3        # controls, absolute
OBJ_BAR = 1e-4      # objective, relative
KKT_BAR = 1e-8      # stationarity of the (x, u) programme, relative


def stage_matrices(ops, xbar, uref):
    """ABC[b][t] from the linearisation operator at theta = xbar[b, 2, t], u = uref[b, :, t]."""
    B, NU, T = uref.shape
    A, Bm, C = ops.linearize(np.ascontiguousarray(xbar[:, 2, :T]).reshape(-1),
                             np.ascontiguousarray(uref.transpose(0, 2, 1)).reshape(-1, NU))
    A, Bm, C = A.reshape(B, T, 3, 3), Bm.reshape(B, T, 3, NU), C.reshape(B, T, 3)
    return [[(A[b, t], Bm[b, t], C[b, t]) for t in range(T)] for b in range(B)]


def check_answers(name, T, ops, x0, xref, uref, out, n_scipy, controls=True, scipy_maxiter=200):
    """Every solved instance of `out` must pass the (x, u) KKT certificate and report the cost of
    the reference's expression; the first n_scipy are also compared with scipy's SLSQP minimiser
    of the same rows.  Returns the worst figures."""
    xbar = ops.rollout(x0, uref)
    abc = stage_matrices(ops, xbar, uref)
    worst = dict(stat=0.0, eq=0.0, viol=0.0, dobj_out=0.0, du=0.0, dobj=0.0, above=0.0, n_scipy=0)
    assert (out["status"] <= 1).all(), np.bincount(out["status"])
    for b in range(x0.shape[0]):
        p = XU.build(name, xref[b], x0[b], uref[b], abc[b], T=T)
        stat, eq, viol = XU.kkt(p, out["x"][b], out["u"][b])
        c_gpu = XU.cost(p, out["x"][b], out["u"][b])
        worst["stat"], worst["eq"], worst["viol"] = max(worst["stat"], stat), max(worst["eq"], eq), max(worst["viol"], viol)
        worst["dobj_out"] = max(worst["dobj_out"], abs(c_gpu - out["obj"][b]) / max(1.0, abs(c_gpu)))
        if b < n_scipy:
            r = XU.solve(p, maxiter=scipy_maxiter)
            if r is None:
                continue
            xs, us, cs = r
            worst["n_scipy"] += 1
            worst["du"] = max(worst["du"], float(np.abs(us - out["u"][b]).max()))
            worst["dobj"] = max(worst["dobj"], abs(c_gpu - cs) / max(1.0, abs(cs)))
            worst["above"] = max(worst["above"], (c_gpu - cs) / max(1.0, abs(cs)))
    assert worst["stat"] <= KKT_BAR and worst["eq"] <= 1e-9 and worst["viol"] <= 1e-9, worst
    assert worst["dobj_out"] <= 1e-9, worst
    if n_scipy:
        assert worst["n_scipy"] >= (3 * min(n_scipy, x0.shape[0])) // 4, worst
        # never above scipy's cost; within the objective bar of it wherever scipy is asked for the
        # controls too (where it is not, scipy stops short of the minimum and only bounds it)
        assert worst["above"] <= 1e-9, worst
        if controls:
            assert worst["dobj"] <= OBJ_BAR and worst["du"] <= U_BAR, worst
    return worst


def run_variant(name, T, ops, x0, xref, cold_start, chained, n_scipy, controls=True, scipy_maxiter=200):
    """Cold start, then a warm start from the cold optimum at the state one plant step on."""
    B = x0.shape[0]
    nu = len(cold_start)
    ucold = np.tile(np.asarray(cold_start, dtype=np.float64)[None, :, None], (B, 1, T))
    if chained:      # one link of mpc_sim.py:295-308's chain: the QP at a given linearisation point
        out = ops.linear(xref, ops.rollout(x0, ucold), x0, ucold)
    else:
        out = ops.solve(x0, xref, None)
    w_cold = check_answers(name, T, ops, x0, xref, ucold, out, n_scipy, controls, scipy_maxiter)
    uprev = out["u"].copy()
    x1 = np.ascontiguousarray(ops.rollout(x0, uprev)[:, :, 1])         # the plant, one step on
    if chained:
        out2 = ops.linear(xref, ops.rollout(x1, uprev), x1, uprev)
    else:
        out2 = ops.solve(x1, xref, uprev)
    w_warm = check_answers(name, T, ops, x1, xref, uprev, out2, n_scipy, controls, scipy_maxiter)
    return w_cold, w_warm
