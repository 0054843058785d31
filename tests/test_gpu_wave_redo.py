"""Hard instances of the wavefront-per-instance kernel, found by scripts/find_wave_unsettled.py
(random controllers of the fuzz generator, 4.9 million solves with that kernel forced; all are warm
starts, of the steer-fixed dual-steer mode and of mpc.py's controller at other horizons).
Most were on their optimal active set with multipliers that jittered above the stationarity
tolerance (penalty times one ulp of the controls): the active-set rounds now ease the penalty there
and settle.  What the kernel still leaves INACCURATE is solved again by the stage-wise pass that
follows it."""
import numpy as np
import pytest

from oracle import orc
from tests.helpers import compare, make_problem
from tests.test_gpu_fuzz import BASES, _random_overrides

pytestmark = pytest.mark.gpu

# seed of the finder script -> instances (of its 8,192) to look at on the warm solve, and whether
# the kernel alone still leaves them unsettled
CASES = {28: [2019], 53: [4684], 78: [3662, 6193], 119: [150, 540, 560, 650, 717, 727, 871, 1891], 168: [4643], 194: [3124]}
UNSETTLED = {168}


def _case(seed):
    rng = np.random.default_rng(31000 + seed)
    bases = [b for b in BASES if b != "sim_ackermann"]
    base = bases[seed % len(bases)]
    kw = _random_overrides(rng, base)
    kw["T"] = int(rng.choice([3, 7, 8, 13, 18, 24, 30]))
    v, ov, x0, xref = make_problem(base, 8192, seed=500 + seed, **kw)
    sel = np.unique(np.concatenate([np.arange(k - 3, k + 4) for k in CASES[seed]]))
    return v, ov, x0[sel], xref[sel], np.searchsorted(sel, CASES[seed])


@pytest.mark.parametrize("seed", sorted(CASES))
def test_unsettled_instances_are_solved_again(seed):
    import torch
    from model_predictive_control_amd import engine
    v, ov, x0, xref, hard = _case(seed)
    cold = engine.solve_batch(v, x0, xref)
    assert (cold["status"].cpu().numpy() == 0).all()
    up = cold["u"].cpu().numpy()
    # the kernel alone: with the answer written over the incoming controls the second pass cannot
    # start from the same inputs and is left out
    alias = engine._outputs(x0.shape[0], v.nu, v.T, cold["u"].device)
    alias["u"].copy_(cold["u"])
    raw = engine.solve_batch(v, x0, xref, alias["u"], None, out=alias)
    st_raw = raw["status"].cpu().numpy()
    if seed in UNSETTLED:
        assert (st_raw[hard] == 1).all() and (st_raw <= 1).all(), st_raw
    else:
        assert (st_raw == 0).all(), st_raw
    hard = np.flatnonzero(st_raw == 1)
    # the shipped call
    warm = engine.solve_batch(v, x0, xref, up)
    assert (warm["status"].cpu().numpy() == 0).all(), warm["status"]
    ref = orc.iterative_batch(ov, x0, xref, up, None, nthreads=8)
    compare(warm, ref, u_tol=1e-6, x_tol=1e-6, obj_tol=1e-8, problem=dict(ov=ov, x0=x0, xref=xref, uref=up))
    # the instances the kernel had settled are bit for bit what it wrote
    easy = np.setdiff1d(np.arange(x0.shape[0]), hard)
    assert torch.equal(warm["u"][easy], raw["u"][easy]) and torch.equal(warm["obj"][easy], raw["obj"][easy])
    # and the entry point that allocates nothing takes the same route
    out = engine._outputs(x0.shape[0], v.nu, v.T, cold["u"].device)
    fixed = engine.solve_batch(v, engine._dev(x0), engine._dev(xref), engine._dev(up), None, out=out, noalloc=True)
    for k in ("u", "x", "obj", "status"):
        assert torch.equal(fixed[k], warm[k]), k
