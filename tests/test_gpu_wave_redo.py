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
CASES = {28: [2019], 53: [4684], 78: [3662, 6193], 119: [150, 540, 560, 650, 717, 727, 871, 1891], 168: [4643, 6295], 194: [3124]}
UNSETTLED = {168}


# the same generator with T = 30 for every seed (the fixed-horizon instantiation, 4.9 million solves):
# instances whose interior point breaks down on the condensed matrix (2^t stage weights, warm start)
# and which the kernel alone reports FAILED
FAILED_AT_30 = {18: [3871], 78: [744], 233: [866]}


def _case(seed, cases=CASES, T=None):
    rng = np.random.default_rng(31000 + seed)
    bases = [b for b in BASES if b != "sim_ackermann"]
    base = bases[seed % len(bases)]
    kw = _random_overrides(rng, base)
    kw["T"] = int(rng.choice([3, 7, 8, 13, 18, 24, 30]))
    if T is not None:
        kw["T"] = T
    v, ov, x0, xref = make_problem(base, 8192, seed=500 + seed, **kw)
    # the warm start of an instance is the cold answer of the same instance: a slice behaves like the batch
    sel = np.unique(np.concatenate([np.arange(k - 3, k + 4) for k in cases[seed]]))
    return v, ov, x0[sel], xref[sel], np.searchsorted(sel, cases[seed])


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


@pytest.mark.parametrize("seed", sorted(FAILED_AT_30))
def test_failed_instances_are_solved_again(seed):
    from model_predictive_control_amd import engine
    v, ov, x0, xref, hard = _case(seed, FAILED_AT_30, T=30)
    cold = engine.solve_batch(v, x0, xref)
    assert (cold["status"].cpu().numpy() == 0).all()
    up = cold["u"].cpu().numpy()
    alias = engine._outputs(x0.shape[0], v.nu, v.T, cold["u"].device)
    alias["u"].copy_(cold["u"])
    st_raw = engine.solve_batch(v, x0, xref, alias["u"], None, out=alias)["status"].cpu().numpy()
    assert (st_raw[hard] == 4).all(), st_raw            # the kernel alone (no second pass when u is aliased)
    warm = engine.solve_batch(v, x0, xref, up)
    assert (warm["status"].cpu().numpy() == 0).all(), warm["status"]
    ref = orc.iterative_batch(ov, x0, xref, up, None, nthreads=8)
    compare(warm, ref, u_tol=1e-6, x_tol=1e-6, obj_tol=1e-8, problem=dict(ov=ov, x0=x0, xref=xref, uref=up))


def test_mpc_dsr_horizon_30_small_batch_against_the_arbiter():
    """mpc_dsr.py's T = 30 variant (Q 2^t, cond(H) about 5e10) on the wavefront-per-instance kernel
    (1,024 instances: its fixed-horizon instantiation): where engine and oracle are more than 1e-5
    apart, the engine has to be within the 1e-3 bar of the extended-precision minimiser of the
    reference's (x, u) programme, and the objectives within 1e-4 relative everywhere."""
    from model_predictive_control_amd import engine
    from tests.helpers import OBJ_BAR, U_BAR
    from tests.test_gpu_full_size import _settle_with_the_arbiter
    B, T = 1024, 30
    v, ov, x0, xref = make_problem("dsr_diff", B, seed=37, n_courses=64, T=T)
    assert engine._lib.load().mpcx_workspace_bytes(engine.C.byref(v), B) == 55 * T * 8 * B   # the route with the second pass
    ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=8)
    out = engine.solve_batch(v, x0, xref)
    up = out["u"].cpu().numpy()
    ref2 = orc.iterative_batch(ov, x0, xref, up, None, nthreads=8)
    out2 = engine.solve_batch(v, x0, xref, up)
    ucold = np.tile(np.tile(np.asarray(ov.cold_start, float)[:ov.nu, None], (1, T))[None], (B, 1, 1))
    for o, r, ur in ((out, ref, ucold), (out2, ref2, up)):
        g = {k: o[k].cpu().numpy() for k in ("u", "x", "obj", "status")}
        assert (g["status"] == 0).all()
        both = r["status"] <= 1
        assert (np.abs(g["obj"][both] - r["obj"][both]) / np.maximum(1.0, np.abs(r["obj"][both]))).max() <= OBJ_BAR
        du = np.abs(g["u"] - r["u"]).reshape(B, -1).max(1)
        far = np.flatnonzero(both & (du > 1e-5))
        d_gpu, d_ref = _settle_with_the_arbiter("dsr_diff", T, v, x0, xref, ur, g, r, far)
        assert d_gpu.max(initial=0.0) <= U_BAR, (d_gpu.max(), d_ref.max())
