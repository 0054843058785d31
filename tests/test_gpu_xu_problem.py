"""HIP engine answers (through the C ABI) against the quadratic programme in the reference's own
variables x (3, T+1), u (NU, T) (tests/xu_problem.py: rows written from the reference's text; no
condensing, no control map, nothing from the oracle).  A_t, B_t, C_t and xbar come from the
engine's linearisation and rollout operators, which the reference's golden vectors pin
(test_linearize_golden, test_rollout_golden).  Per variant, cold and warm: every instance must
pass the (x, u) KKT certificate and report the cost of the reference's expression, and is
compared with scipy's SLSQP minimiser of the same rows (controls 1e-3, objective 1e-4
relative).  GPU only."""
import numpy as np
import pytest

from tests import xu_checks
from tests import xu_problem as XU

pytestmark = pytest.mark.gpu

CASES = [("diff_drive", 5), ("dsr_ackermann", 5), ("dsr_crab", 5), ("dsr_diff", 5), ("sim_ackermann", 5),
         ("sim_diff", 5), ("sim_crab", 5), ("mpc_py", 10)]


class EngineOps:
    def __init__(self, v):
        from model_predictive_control_amd import engine
        self.v, self.e = v, engine

    @staticmethod
    def _np(out):
        return {k: t.cpu().numpy() for k, t in out.items()}

    def rollout(self, x0, u):
        return self.e.rollout_batch(self.v, x0, u).cpu().numpy()

    def linearize(self, theta, u):
        return tuple(t.cpu().numpy() for t in self.e.linearize_batch(self.v, theta, u))

    def solve(self, x0, xref, uprev):
        return self._np(self.e.solve_batch(self.v, x0, xref, uprev))

    def linear(self, xref, xbar, x0, uref):
        return self._np(self.e.linear_mpc_batch(self.v, xref, xbar, x0, uref))


def _inputs(name, T, B):
    from model_predictive_control_amd import variants, workload
    v = variants.get(name, T=T)
    courses = workload.make_courses(16)
    inst = workload.make_instances(courses, B, seed=11)
    xref = workload.reference_window_next_points(courses, inst, T) if name == "mpc_py" else workload.reference_window(courses, inst, T)
    return v, inst["state"][:, :3].copy(), xref


@pytest.mark.parametrize("name,T", CASES)
def test_gpu_minimises_the_reference_programme(name, T):
    v, x0, xref = _inputs(name, T, 64)
    w = xu_checks.run_variant(name, T, EngineOps(v), x0, xref, list(v.cold_start)[:v.nu], v.chain > 1, n_scipy=64)
    print(name, w)


def test_gpu_minimises_the_reference_programme_dsr_ackermann_horizon_30():
    v, x0, xref = _inputs("dsr_ackermann", 30, 64)
    xu_checks.run_variant("dsr_ackermann", 30, EngineOps(v), x0, xref, [0.0] * 4, False, n_scipy=4)


def test_gpu_minimises_the_reference_programme_dsr_diff_horizon_30():
    """mpc_dsr.py as it ships (Q * 2**t, :349) at T = 30: certificate on all 64; against scipy on
    the objective only (SLSQP stops above the minimum on this one: cond ~ 5e10)."""
    v, x0, xref = _inputs("dsr_diff", 30, 64)
    xu_checks.run_variant("dsr_diff", 30, EngineOps(v), x0, xref, [0.0] * 4, False, n_scipy=3, controls=False)


@pytest.mark.parametrize("name", ["sim_ackermann", "sim_diff", "sim_crab"])
@pytest.mark.parametrize("warm", [False, True])
def test_chain_link_by_link(name, warm):
    """MPC.iterativeLMPC's chain (mpc_sim.py:292-308) re-enacted on the host with one single-QP
    launch per link: xbar once from the incoming controls, uref from zeros, column i filled from
    the result of link i-1, link i solved at that uref.  Every link must pass the (x, u)
    certificate, and the one-launch chained solve must return what the last link returns."""
    T, B = 5, 64
    v, x0, xref = _inputs(name, T, B)
    ops = EngineOps(v)
    prev = np.zeros((B, 4, T))
    if warm:
        prev = ops.solve(x0, xref, None)["u"]
        x0 = np.ascontiguousarray(ops.rollout(x0, prev)[:, :, 1])
    chained = ops.solve(x0, xref, prev if warm else None)
    xbar = ops.rollout(x0, prev)                                       # :294
    uref = np.zeros((B, 4, T))                                         # :292
    for i in range(T):
        uref[:, :, i] = prev[:, :, i]                                  # :296-299
        link = ops.linear(xref, xbar, x0, uref)                        # :302-306
        assert (link["status"] <= 1).all()
        abc = xu_checks.stage_matrices(ops, xbar, uref)
        for b in range(B):
            p = XU.build(name, xref[b], x0[b], uref[b], abc[b], T=T)
            stat, eq, viol = XU.kkt(p, link["x"][b], link["u"][b])
            assert stat <= xu_checks.KKT_BAR and eq <= 1e-9 and viol <= 1e-9, (i, b, stat, eq, viol)
        prev = link["u"]
    assert (chained["status"] <= 1).all()
    du = np.abs(chained["u"] - link["u"]).max()
    dx = np.abs(chained["x"] - link["x"]).max()
    dobj = np.abs(chained["obj"] - link["obj"]).max()
    assert du <= 1e-10 and dx <= 1e-10 and dobj <= 1e-10, (du, dx, dobj)
