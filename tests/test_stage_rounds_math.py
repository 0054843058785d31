"""The math of the stage-wise kernel's active-set rounds, restated in numpy (tests/stage_proto.py):
the Riccati sweep with fixed / tied controls solves the working-set QP exactly (against a dense
KKT solve), the chain sums recover its multipliers, and interior point + rounds lands on the
oracle's minimiser."""
import numpy as np
import pytest

from oracle import mpc_oracle as O
from tests import stage_proto as SP
from tests.helpers import make_problem


@pytest.mark.parametrize("name,T", [("dsr_ackermann", 12), ("diff_drive", 9), ("mpc_py", 10), ("dsr_diff", 8)])
def test_rounds_reach_the_oracle_minimiser(name, T):
    _, ov, x0, xref = make_problem(name, 6, T=T)
    for b in range(6):
        ucold = np.tile(np.asarray(ov.cold_start, float)[:ov.nu, None], (1, T))
        ref = O.iterative_linear_mpc_control(ov, xref[b], x0[b], None)
        P = SP.Prob(ov, x0[b], xref[b], ucold)
        w0, act, Cm, d = SP.ipm_dense(P, 6)
        box, tie = SP.sets_from_mask(P, act)
        gr, _ = P.gradient(w0)
        wr = SP.eqp_riccati(P, w0, gr, box, tie)
        wd = SP.eqp_dense(P, w0, box, tie)
        assert np.abs(wr - wd).max() <= 1e-9 * max(1.0, np.abs(wd).max())
        w, ok, rounds, _ = SP.rounds(P, w0, box, tie)
        assert ok and rounds <= 4
        assert np.abs(w - ref.w).max() <= 1e-9
        # multipliers recovered by the chain sums satisfy stationarity with the dense gradient
        bx, ti = SP.sets_from_point(P, w, 1e-10)
        gr, mag = P.gradient(w)
        mu, nu, resid = SP.recover_multipliers(T, gr, bx, ti)
        assert np.abs(resid).max() <= 1e-10 * (1.0 + mag.max())
        full = gr + mu
        full[1:] += nu[:-1] * (ti[:-1] != 0)
        full[:-1] -= nu[:-1] * (ti[:-1] != 0)
        assert np.abs(full).max() <= 1e-10 * (1.0 + mag.max())
