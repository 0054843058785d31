"""Randomised controllers against the oracle: every variant's structure with random horizon,
sample time, weights, bounds (some absent) and batch size, cold and warm, so that every solve
kernel and every routing decision sees parameters nobody tuned it on."""
import math

import numpy as np
import pytest

from oracle import orc
from tests.helpers import compare, make_problem

pytestmark = pytest.mark.gpu

@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from model_predictive_control_amd import engine
    return engine


BASES = ["diff_drive", "dsr_ackermann", "dsr_crab", "dsr_diff", "sim_ackermann", "mpc_py"]


def _random_overrides(rng, base):
    nu = 2 if base == "diff_drive" else 4
    T = int(rng.choice([1, 2, 3, 5, 5, 7, 10, 10, 13, 18]))
    ov = {"T": T, "dt": float(rng.choice([0.1, 0.2, 0.25]))}
    ov["R"] = tuple(float(10.0 ** rng.uniform(-2.5, -0.5)) for _ in range(nu))
    ov["Q"] = tuple(float(10.0 ** rng.uniform(-0.5, 0.7)) for _ in range(3))
    ov["Qf"] = tuple(float(q * 10.0 ** rng.uniform(0.0, 0.5)) for q in ov["Q"])
    if rng.random() < 0.4:
        ov["Rd"] = tuple(float(10.0 ** rng.uniform(-2.0, 0.0)) for _ in range(nu))
    if rng.random() < 0.3:
        ov["q_growth"] = float(rng.choice([1.1, 1.3]))
    speed = float(rng.uniform(0.3, 1.0))
    turn = float(rng.uniform(0.6, 3.0)) if base == "diff_drive" else float(rng.uniform(0.4, 1.5))
    um = [speed, turn] if nu == 2 else [speed, turn, speed, turn]
    dm = [float(rng.uniform(0.05, 0.4)), float(rng.uniform(0.1, 1.0))] * (nu // 2)
    if rng.random() < 0.25:                       # a channel without rate bound
        k = int(rng.integers(0, 2))
        for i in range(k, nu, 2):
            dm[i] = math.inf
    if rng.random() < 0.15:                       # no bounds on the speed at all
        for i in range(0, nu, 2):
            um[i] = math.inf
    if base == "dsr_diff":                        # steer fixed at 90 degrees: no bounds on the steer rows
        um[1] = um[3] = dm[1] = dm[3] = math.inf
    ov["umax"], ov["dumax"] = tuple(um), tuple(dm)
    return ov


@pytest.mark.parametrize("seed", range(60))
def test_random_controller(eng, seed):
    rng = np.random.default_rng(1000 + seed)
    base = BASES[seed % len(BASES)]
    ov_kw = _random_overrides(rng, base)
    B = int(rng.choice([1, 7, 64, 65, 130, 1200, 3500]))
    v, ov, x0, xref = make_problem(base, B, seed=50 + seed, **ov_kw)
    ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=8)
    out = eng.solve_batch(v, x0, xref)
    # the certificate as well: agreeing with the oracle is not enough if both stop short the same way
    compare(out, ref, u_tol=1e-6, x_tol=1e-6, obj_tol=1e-8, problem=dict(ov=ov, x0=x0, xref=xref))
    hp = (np.arange(B) % 3 != 0).astype(np.uint8)
    up = np.where(np.isnan(ref["u"]), 0.0, ref["u"])
    ref2 = orc.iterative_batch(ov, x0, xref, up, hp, nthreads=8)
    out2 = eng.solve_batch(v, x0, xref, up, hp)
    compare(out2, ref2, u_tol=1e-6, x_tol=1e-6, obj_tol=1e-8, problem=dict(ov=ov, x0=x0, xref=xref, uref=up, has_prev=hp))


def _horizon_10_case(seed):
    """A random controller that the four-lanes-per-instance kernel takes: T = 10, unchained, a finite
    box on both free channels (rate bounds may be absent), odd batch sizes on both sides of the
    16-instance wavefront and of the 16,384 instances the chip holds at once."""
    rng = np.random.default_rng(7000 + seed)
    base = ["diff_drive", "dsr_ackermann", "dsr_crab", "dsr_diff", "mpc_py"][seed % 5]
    kw = _random_overrides(rng, base)
    kw["T"] = 10
    nu = 2 if base == "diff_drive" else 4
    um = list(kw["umax"])
    for i in range(0, nu, 2):
        if not math.isfinite(um[i]):
            um[i] = float(rng.uniform(0.3, 1.0))
    kw["umax"] = tuple(um)
    B = int(rng.choice([1, 15, 16, 17, 250, 1023, 4099, 16385, 20011]))
    return base, kw, B


@pytest.mark.parametrize("seed", range(15))
def test_random_controller_at_horizon_10(eng, seed):
    base, kw, B = _horizon_10_case(seed)
    v, ov, x0, xref = make_problem(base, B, seed=90 + seed, **kw)
    ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=16)
    out = eng.solve_batch(v, x0, xref)
    compare(out, ref, u_tol=1e-6, x_tol=1e-6, obj_tol=1e-8, problem=dict(ov=ov, x0=x0, xref=xref))
    hp = (np.arange(B) % 3 != 0).astype(np.uint8)
    up = np.where(np.isnan(ref["u"]), 0.0, ref["u"])
    ref2 = orc.iterative_batch(ov, x0, xref, up, hp, nthreads=16)
    out2 = eng.solve_batch(v, x0, xref, up, hp)
    compare(out2, ref2, u_tol=1e-6, x_tol=1e-6, obj_tol=1e-8, problem=dict(ov=ov, x0=x0, xref=xref, uref=up, has_prev=hp))
