"""The batched condensing and the KKT certificate of tests/helpers.py (what certifies GPU
outputs independently of any solver) against the oracle's per-instance functions."""
import numpy as np
import pytest

from oracle import mpc_oracle as O
from oracle import orc
from tests.helpers import condense_batch, free_controls, kkt_certificate, make_problem


@pytest.mark.parametrize("name,T", [("diff_drive", 5), ("dsr_ackermann", 5), ("dsr_crab", 7), ("dsr_diff", 30), ("mpc_py", 10)])
def test_batched_condense_equals_oracle_condense(name, T):
    v, ov, x0, xref = make_problem(name, 6, T=T)
    rng = np.random.default_rng(T)
    uref = np.tile(np.asarray(ov.cold_start, float)[:ov.nu, None], (6, 1, T)) + rng.uniform(-0.2, 0.2, (6, ov.nu, T))
    xbar = orc.rollout_batch(ov, x0, uref)
    H, g, lo, hi, rr, feas = condense_batch(ov, xref, xbar[:, 2, :T], x0, uref)
    for b in range(6):
        qp = O.condense(ov, xref[b], xbar[b], x0[b], uref[b])
        np.testing.assert_allclose(H[b], qp.H, rtol=1e-12, atol=1e-12 * np.abs(qp.H).max())
        np.testing.assert_allclose(g[b], qp.g, rtol=1e-11, atol=1e-12 * np.abs(qp.g).max())
        assert np.array_equal(lo[b], qp.lo) and np.array_equal(hi[b], qp.hi) and np.array_equal(rr, qp.rr)
        assert feas[b] == qp.feasible


@pytest.mark.parametrize("name,T", [("diff_drive", 5), ("dsr_ackermann", 12), ("dsr_diff", 5), ("mpc_py", 10)])
def test_certificate_accepts_the_minimiser_and_rejects_a_perturbed_point(name, T):
    v, ov, x0, xref = make_problem(name, 24, T=T)
    ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=4)
    ur = np.tile(np.asarray(ov.cold_start, float)[:ov.nu, None], (24, 1, T))
    xbar = orc.rollout_batch(ov, x0, ur)
    H, g, lo, hi, rr, feas = condense_batch(ov, xref, xbar[:, 2, :T], x0, ur)
    w = free_controls(ov, ref["u"])
    stat, viol, neg = kkt_certificate(H, g, lo, hi, rr, w)
    assert stat.max() <= 1e-9 and viol.max() <= 1e-10 and neg.max() <= 1e-9
    # a feasible point 1e-3 away from the minimiser is not stationary
    w2 = np.clip(w - 1e-3 * np.sign(w), lo, hi)
    w2 = np.where(np.abs(w2) < 2e-3, w2 + 1e-3, w2)
    stat2, _, neg2 = kkt_certificate(H, g, lo, hi, rr, w2, act_tol=1e-9)
    assert (np.maximum(stat2, neg2) > 1e-6).mean() > 0.9
