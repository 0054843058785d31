"""The quadratic programme the reference hands to cvxpy, in its own variables x (3, T+1) and
u (NU, T), written here from the reference's text: cost terms and constraint rows appear in the
order the reference adds them.  Nothing in this module comes from the oracle or from the
engine: no condensing, no elimination of the tied control rows, no control map.  The matrices
A_t, B_t, C_t are an argument (the tests pass the operators that the reference's golden vectors
pin).  Test infrastructure.

    diff_drive     mpc_diff_drive.py:170-197  (constants :19-48)
    dsr_ackermann  mpc_dsr.py:201-243         (constants :20-48)
    dsr_crab       mpc_dsr.py:269-311
    dsr_diff       mpc_dsr.py:337-370
    sim_ackermann  mpc_sim.py:312-356         (constants :176-195)
    sim_diff       mpc_sim.py:382-415
    sim_crab       mpc_sim.py:442-484
    mpc_py         mpc.py:117-163             (constants :33-47)
"""
import numpy as np
from scipy.optimize import LinearConstraint, minimize

DEG45, DEG90 = np.deg2rad(45), np.deg2rad(90)


def _dsr_file(T=5):
    """mpc_dsr.py:20-48"""
    return dict(NU=4, T=T, DT=0.2, R=[0.01] * 4, Q=[1.0, 1.0, 0.5], Qf=[1.0, 1.0, 0.5], cold=[0.0] * 4,
                MAX_V_SPEED=0.47, DIFF_V_SPEED=0.5, MAX_STEER=DEG90, DIFF_STEER=DEG45)


def _sim_file(T=5):
    """mpc_sim.py:167-195; dt = 0.2 is the default argument of every doLMPC_* (:312, :382, :442)"""
    return dict(NU=4, T=T, DT=0.2, R=[0.01] * 4, Q=[1.0, 1.0, 0.5], Qf=[1.0, 1.0, 0.5], cold=[0.0] * 4)


class Rows:
    """Collects cost terms and constraint rows over z = [x row-major (3, T+1); u row-major (NU, T)]."""

    def __init__(self, NU, T):
        self.NU, self.T = NU, T
        self.n = 3 * (T + 1) + NU * T
        self.P = np.zeros((self.n, self.n))      # cost = z'Pz + q'z + c
        self.q = np.zeros(self.n)
        self.c = 0.0
        self.terms = []                           # (row, const, weight): cost = sum weight (row z + const)^2
        self.eq, self.eq_rhs = [], []
        self.ub, self.ub_rhs = [], []             # row z <= rhs

    def x(self, i, t):
        return i * (self.T + 1) + t

    def u(self, j, t):
        return 3 * (self.T + 1) + j * self.T + t

    def row(self, terms):
        r = np.zeros(self.n)
        for k, a in terms:
            r[k] += a
        return r

    def quad(self, row, const, weight):
        """weight * (row z + const)^2"""
        self.P += weight * np.outer(row, row)
        self.q += 2.0 * weight * const * row
        self.c += weight * const * const
        self.terms.append((row, const, weight))

    def equal(self, row, rhs):
        self.eq.append(row); self.eq_rhs.append(rhs)

    def abs_le(self, row, const, bound):
        """|row z + const| <= bound, as cvxpy.abs(...) <= bound"""
        self.ub.append(row); self.ub_rhs.append(bound - const)
        self.ub.append(-row); self.ub_rhs.append(bound + const)


def build(name, xref, x0, uref, ABC, T=None):
    """Rows of variant `name` at one instance.  xref (3, T+1), x0 (3,), uref (NU, T);
    ABC[t] = (A, B, C) of stage t (from get_linear_model_matrix at xbar[2, t], uref[:, t])."""
    if name == "diff_drive":
        # mpc_diff_drive.py:19-48
        NU, T_, DT = 2, 5 if T is None else T, 0.2
        R, Q, Qf = [0.01, 0.01], [1.0, 1.0, 0.5], [1.0, 1.0, 0.5]
        MAX_V_SPEED, MAX_W_SPEED, DIFF_V_SPEED, DIFF_W_SPEED = 0.47, 3.77, 0.2, 1.0
        p = Rows(NU, T_)
        for t in range(T_):
            for j in range(NU):
                p.quad(p.row([(p.u(j, t), 1.0)]), 0.0, R[j])                                   # :179
            if t != 0:
                for i in range(3):
                    p.quad(p.row([(p.x(i, t), -1.0)]), xref[i, t], Q[i])                       # :182
            _dynamics(p, t, ABC[t])                                                            # :187
            if t < T_ - 1:
                p.abs_le(p.row([(p.u(0, t + 1), 1.0), (p.u(0, t), -1.0)]), 0.0, DIFF_V_SPEED)  # :189
                p.abs_le(p.row([(p.u(1, t + 1), 1.0), (p.u(1, t), -1.0)]), 0.0, DIFF_W_SPEED)  # :190
        for i in range(3):
            p.quad(p.row([(p.x(i, T_), -1.0)]), xref[i, T_], Qf[i])                            # :193
            p.equal(p.row([(p.x(i, 0), 1.0)]), x0[i])                                          # :195
        for t in range(T_):
            p.abs_le(p.row([(p.u(0, t), 1.0)]), 0.0, MAX_V_SPEED)                              # :196
            p.abs_le(p.row([(p.u(1, t), 1.0)]), 0.0, MAX_W_SPEED)                              # :197
        return p

    if name in ("dsr_ackermann", "dsr_crab", "dsr_diff"):
        k = _dsr_file(5 if T is None else T)
        NU, T_, DT = k["NU"], k["T"], k["DT"]
        p = Rows(NU, T_)
        for t in range(T_):
            for j in range(NU):
                p.quad(p.row([(p.u(j, t), 1.0)]), 0.0, k["R"][j])                              # mpc_dsr.py:210
            if t != 0:
                grow = (2 ** t) if name == "dsr_diff" else 1                                   # :349  Q*(2**t)
                for i in range(3):
                    p.quad(p.row([(p.x(i, t), -1.0)]), xref[i, t], k["Q"][i] * grow)           # :213
            _dynamics(p, t, ABC[t])                                                            # :222
            if t < T_ - 1:
                p.abs_le(p.row([(p.u(0, t + 1), 1.0), (p.u(0, t), -1.0)]), 0.0, k["DIFF_V_SPEED"] * DT)   # :225
                p.abs_le(p.row([(p.u(2, t + 1), 1.0), (p.u(2, t), -1.0)]), 0.0, k["DIFF_V_SPEED"] * DT)   # :226
                if name != "dsr_diff":
                    p.abs_le(p.row([(p.u(1, t + 1), 1.0), (p.u(1, t), -1.0)]), 0.0, k["DIFF_STEER"] * DT)  # :227
                    p.abs_le(p.row([(p.u(3, t + 1), 1.0), (p.u(3, t), -1.0)]), 0.0, k["DIFF_STEER"] * DT)  # :228
            if t == 0 and name != "dsr_diff":
                p.abs_le(p.row([(p.u(1, t), -1.0)]), uref[1, t], k["DIFF_STEER"] * DT)         # :231
                p.abs_le(p.row([(p.u(3, t), -1.0)]), uref[3, t], k["DIFF_STEER"] * DT)         # :232
        for i in range(3):
            p.quad(p.row([(p.x(i, T_), -1.0)]), xref[i, T_], k["Qf"][i])                       # :235 / :364 (Qf unscaled)
            p.equal(p.row([(p.x(i, 0), 1.0)]), x0[i])                                          # :237
        for t in range(T_):
            p.abs_le(p.row([(p.u(0, t), 1.0)]), 0.0, k["MAX_V_SPEED"])                         # :238
            p.abs_le(p.row([(p.u(2, t), 1.0)]), 0.0, k["MAX_V_SPEED"])                         # :239
            if name == "dsr_diff":
                p.equal(p.row([(p.u(1, t), 1.0)]), k["MAX_STEER"])                             # :369
                p.equal(p.row([(p.u(3, t), 1.0)]), k["MAX_STEER"])                             # :370
            else:
                p.abs_le(p.row([(p.u(1, t), 1.0)]), 0.0, k["MAX_STEER"])                       # :240
                p.abs_le(p.row([(p.u(3, t), 1.0)]), 0.0, k["MAX_STEER"])                       # :241
                s = 1.0 if name == "dsr_ackermann" else -1.0
                p.equal(p.row([(p.u(1, t), 1.0), (p.u(3, t), s)]), 0.0)                        # :242 u1 == -u3 / :310 u1 == u3
                p.equal(p.row([(p.u(0, t), 1.0), (p.u(2, t), -1.0)]), 0.0)                     # :243
        return p

    if name in ("sim_ackermann", "sim_crab", "sim_diff"):
        k = _sim_file(5 if T is None else T)
        NU, T_, dt = k["NU"], k["T"], k["DT"]
        # mpc_sim.py:176-189
        speed_inc = 0.2
        steer_inc = DEG45
        max_speed = 0.5 if name == "sim_crab" else 0.47
        max_steer = DEG45
        fixed_steer = DEG90
        p = Rows(NU, T_)
        for t in range(T_):
            for j in range(NU):
                p.quad(p.row([(p.u(j, t), 1.0)]), 0.0, k["R"][j])                              # mpc_sim.py:321
            if t != 0:
                for i in range(3):
                    p.quad(p.row([(p.x(i, t), -1.0)]), xref[i, t], k["Q"][i])                  # :324
            _dynamics(p, t, ABC[t])                                                            # :335
            if t < T_ - 1:
                p.abs_le(p.row([(p.u(0, t + 1), 1.0), (p.u(0, t), -1.0)]), 0.0, speed_inc * dt)        # :338
                p.abs_le(p.row([(p.u(2, t + 1), 1.0), (p.u(2, t), -1.0)]), 0.0, speed_inc * dt)        # :339
                if name != "sim_diff":
                    p.abs_le(p.row([(p.u(1, t + 1), 1.0), (p.u(1, t), -1.0)]), 0.0, steer_inc * dt)    # :340
                    p.abs_le(p.row([(p.u(3, t + 1), 1.0), (p.u(3, t), -1.0)]), 0.0, steer_inc * dt)    # :341
            if t == 0 and name != "sim_diff":
                p.abs_le(p.row([(p.u(1, t), -1.0)]), uref[1, t], steer_inc * dt)               # :344
                p.abs_le(p.row([(p.u(3, t), -1.0)]), uref[3, t], steer_inc * dt)               # :345
        for i in range(3):
            p.quad(p.row([(p.x(i, T_), -1.0)]), xref[i, T_], k["Qf"][i])                       # :348
            p.equal(p.row([(p.x(i, 0), 1.0)]), x0[i])                                          # :350
        for t in range(T_):
            p.abs_le(p.row([(p.u(0, t), 1.0)]), 0.0, max_speed)                                # :351
            p.abs_le(p.row([(p.u(2, t), 1.0)]), 0.0, max_speed)                                # :352
            if name == "sim_diff":
                p.equal(p.row([(p.u(1, t), 1.0)]), fixed_steer)                                # :414
                p.equal(p.row([(p.u(3, t), 1.0)]), fixed_steer)                                # :415
            else:
                p.abs_le(p.row([(p.u(1, t), 1.0)]), 0.0, max_steer)                            # :353
                p.abs_le(p.row([(p.u(3, t), 1.0)]), 0.0, max_steer)                            # :354
                s = 1.0 if name == "sim_ackermann" else -1.0
                p.equal(p.row([(p.u(1, t), 1.0), (p.u(3, t), s)]), 0.0)                        # :355 / :483
                p.equal(p.row([(p.u(0, t), 1.0), (p.u(2, t), -1.0)]), 0.0)                     # :356
        return p

    if name == "mpc_py":
        # mpc.py:33-47
        NU, HL, DT = 4, 10 if T is None else T, 0.1
        R, R_diff = [0.001] * 4, [0.001] * 4
        Q, Q_final = [0.1, 0.1, 0.1], [0.1, 0.1, 0.0]
        MAX_TRAVEL_SPEED, MAX_STEER_SPEED, MAX_STEER = 0.5, 3, 3.14
        p = Rows(NU, HL)
        for t in range(HL):
            for j in range(NU):
                p.quad(p.row([(p.u(j, t), 1.0)]), 0.0, R[j])                                   # mpc.py:131
            if t != 0:
                for i in range(3):
                    p.quad(p.row([(p.x(i, t), -1.0)]), xref[i, t], Q[i])                       # :134
            _dynamics(p, t, ABC[t])                                                            # :142
            if t == 0:
                p.abs_le(p.row([(p.u(1, t), 1.0)]), -uref[1, t], MAX_STEER_SPEED * DT)         # :145
                p.abs_le(p.row([(p.u(3, t), 1.0)]), -uref[3, t], MAX_STEER_SPEED * DT)         # :146
            p.equal(p.row([(p.u(0, t), 1.0), (p.u(2, t), -1.0)]), 0.0)                         # :148
            p.equal(p.row([(p.u(1, t), 1.0), (p.u(3, t), 1.0)]), 0.0)                          # :149
            p.abs_le(p.row([(p.u(1, t), 1.0)]), 0.0, MAX_STEER)                                # :150
            p.abs_le(p.row([(p.u(3, t), 1.0)]), 0.0, MAX_STEER)                                # :151
            if t < HL - 1:
                for j in range(NU):
                    p.quad(p.row([(p.u(j, t + 1), 1.0), (p.u(j, t), -1.0)]), 0.0, R_diff[j])   # :153
                p.abs_le(p.row([(p.u(1, t + 1), 1.0), (p.u(1, t), -1.0)]), 0.0, MAX_STEER_SPEED * DT)   # :154
                p.abs_le(p.row([(p.u(3, t + 1), 1.0), (p.u(3, t), -1.0)]), 0.0, MAX_STEER_SPEED * DT)   # :155
        for i in range(3):
            p.quad(p.row([(p.x(i, HL), -1.0)]), xref[i, HL], Q_final[i])                       # :157
            p.equal(p.row([(p.x(i, 0), 1.0)]), x0[i])                                          # :161
        for t in range(HL):
            p.abs_le(p.row([(p.u(0, t), 1.0)]), 0.0, MAX_TRAVEL_SPEED)                         # :162
            p.abs_le(p.row([(p.u(2, t), 1.0)]), 0.0, MAX_TRAVEL_SPEED)                         # :163
        return p
    raise KeyError(name)


def _dynamics(p, t, abc):
    """x[:, t + 1] == A @ x[:, t] + B @ u[:, t] + C"""
    A, B, C = abc
    for i in range(3):
        terms = [(p.x(i, t + 1), 1.0)]
        terms += [(p.x(k, t), -A[i, k]) for k in range(3)]
        terms += [(p.u(j, t), -B[i, j]) for j in range(p.NU)]
        p.equal(p.row(terms), C[i])


def cost(p, x, u):
    """The reference's cost expression at a point, term by term (the expanded form z'Pz + q'z + c
    cancels badly when the positions are large against the tracking errors)."""
    z = np.concatenate([np.asarray(x).ravel(), np.asarray(u).ravel()])
    return float(sum(w * (row @ z + const) ** 2 for row, const, w in p.terms))


def residuals(p, x, u):
    """(largest equality residual, largest inequality violation) of a point"""
    z = np.concatenate([np.asarray(x).ravel(), np.asarray(u).ravel()])
    Aeq, beq = np.array(p.eq), np.array(p.eq_rhs)
    Aub, bub = np.array(p.ub), np.array(p.ub_rhs)
    return float(np.abs(Aeq @ z - beq).max()), float(np.maximum(Aub @ z - bub, 0.0).max())


def solve(p, method="SLSQP", z0=None, maxiter=200):
    """Minimiser of the rows by scipy (SLSQP or trust-constr): (x (3, T+1), u (NU, T), cost), or
    None when the point scipy ends at violates a constraint by more than 1e-9.  SLSQP's own
    success flag is not used: at this tolerance it often runs into its iteration limit while
    sitting on the minimiser, so callers compare costs instead (a point with a cost above the
    minimum cannot pass for it)."""
    Aeq, beq = np.array(p.eq), np.array(p.eq_rhs)
    Aub, bub = np.array(p.ub), np.array(p.ub_rhs)
    sc = 1.0 / max(1.0, np.abs(p.q).max())
    P2 = 2.0 * p.P
    fun = lambda z: sc * (z @ p.P @ z + p.q @ z)
    jac = lambda z: sc * (P2 @ z + p.q)
    if z0 is None:
        # a point that satisfies the equalities (least norm): SLSQP then stays on them
        z0 = np.linalg.lstsq(Aeq, beq, rcond=None)[0]
    if method == "SLSQP":
        cons = [{"type": "eq", "fun": lambda z: Aeq @ z - beq, "jac": lambda z: Aeq},
                {"type": "ineq", "fun": lambda z: bub - Aub @ z, "jac": lambda z: -Aub}]
        r = minimize(fun, z0, jac=jac, constraints=cons, method="SLSQP", options={"ftol": 1e-14, "maxiter": maxiter})
    else:
        cons = [LinearConstraint(Aeq, beq, beq), LinearConstraint(Aub, -np.inf, bub)]
        r = minimize(fun, z0, jac=jac, hess=lambda z: sc * P2, constraints=cons, method="trust-constr",
                     options={"gtol": 1e-12, "xtol": 1e-14, "barrier_tol": 1e-12, "maxiter": 3000})
    z = r.x
    if np.abs(Aeq @ z - beq).max() > 1e-9 or (Aub @ z - bub).max() > 1e-9:
        return None
    nx = 3 * (p.T + 1)
    xs, us = z[:nx].reshape(3, p.T + 1), z[nx:].reshape(p.NU, p.T)
    return xs, us, cost(p, xs, us)


def kkt(p, x, u, act_tol=1e-9):
    """Optimality certificate of a point in the reference's own variables: with the constraints
    the point touches (slack <= act_tol), multipliers nu (free sign, equalities) and lam >= 0
    are fitted by non-negative least squares to  2Pz + q + Aeq' nu + Aact' lam = 0.
    Returns (stationarity residual relative to 1 + max(|2Pz|, |q|), largest equality residual,
    largest inequality violation).  The problem is convex, so a point with all three at rounding
    level is a minimiser; u is unique because every u has a positive cost weight."""
    from scipy.optimize import nnls
    z = np.concatenate([np.asarray(x, dtype=np.float64).ravel(), np.asarray(u, dtype=np.float64).ravel()])
    Aeq, beq = np.array(p.eq), np.array(p.eq_rhs)
    Aub, bub = np.array(p.ub), np.array(p.ub_rhs)
    grad = 2.0 * (p.P @ z) + p.q
    slack = bub - Aub @ z
    act = Aub[slack <= act_tol]
    # column scaling keeps nnls well conditioned when cost weights span many decades
    M = np.concatenate([Aeq.T, -Aeq.T, act.T], axis=1)
    scale = 1.0 + max(np.abs(2.0 * (p.P @ z)).max(), np.abs(p.q).max())
    _, rnorm = nnls(M, -grad / scale, maxiter=30 * M.shape[1])
    return rnorm, float(np.abs(Aeq @ z - beq).max()), float(np.maximum(-slack, 0.0).max())


def arbiter(p, x, u, act_tol=1e-9):
    """Higher-precision minimiser of the programme on the active set a given point sits on: the
    equalities plus the inequality rows with slack <= act_tol are taken as equalities, their
    solution set is parametrised by Gauss-Jordan elimination and the reduced quadratic is
    minimised, everything in numpy.longdouble (64-bit mantissa on x86).  Returns
    (x*, u*, kkt(p, x*, u*)): when that certificate passes, (x*, u*) is the minimiser of the
    whole programme, computed about 2000 times finer than float64 can."""
    L = np.longdouble
    z = np.concatenate([np.asarray(x, dtype=np.float64).ravel(), np.asarray(u, dtype=np.float64).ravel()])
    Aeq, beq = np.array(p.eq), np.array(p.eq_rhs)
    Aub, bub = np.array(p.ub), np.array(p.ub_rhs)
    act = (bub - Aub @ z) <= act_tol
    M = np.concatenate([np.concatenate([Aeq, Aub[act]]), np.concatenate([beq, bub[act]])[:, None]], axis=1).astype(L)
    m, n = M.shape[0], M.shape[1] - 1
    pivots, r = [], 0
    for c in range(n):
        if r >= m:
            break
        k = r + int(np.argmax(np.abs(M[r:, c])))
        if abs(M[k, c]) <= 1e-12:
            continue
        M[[r, k]] = M[[k, r]]
        M[r] /= M[r, c]
        rows = np.arange(m) != r
        M[rows] -= M[rows, c:c + 1] * M[r:r + 1]
        pivots.append(c)
        r += 1
    assert np.abs(M[r:, n]).max(initial=0.0) <= 1e-9, "the active rows contradict each other"
    free = [c for c in range(n) if c not in set(pivots)]
    zp = np.zeros(n, dtype=L)
    zp[pivots] = M[:len(pivots), n]
    N = np.zeros((n, len(free)), dtype=L)
    for j, f in enumerate(free):
        N[f, j] = 1
        N[pivots, j] = -M[:len(pivots), f]
    P2 = 2 * p.P.astype(L)
    if free:
        Hr = N.T @ (P2 @ N)
        gr = N.T @ (P2 @ zp + p.q.astype(L))
        K = np.concatenate([Hr, -gr[:, None]], axis=1)
        k = len(free)
        for c in range(k):                      # Gaussian elimination with row pivoting
            j = c + int(np.argmax(np.abs(K[c:, c])))
            K[[c, j]] = K[[j, c]]
            K[c] /= K[c, c]
            rows = np.arange(k) != c
            K[rows] -= K[rows, c:c + 1] * K[c:c + 1]
        zs = zp + N @ K[:, k]
    else:
        zs = zp
    zs = zs.astype(np.float64)
    nx = 3 * (p.T + 1)
    xs, us = zs[:nx].reshape(3, p.T + 1), zs[nx:].reshape(p.NU, p.T)
    return xs, us, kkt(p, xs, us)
