"""HIP engine (through the C ABI) against the oracle and the golden vectors.  GPU only."""
import math

import numpy as np
import pytest

from oracle import mpc_oracle as O
from oracle import orc
from tests.helpers import OBJ_BAR, U_BAR, certify, compare, make_problem, slsqp_check

pytestmark = pytest.mark.gpu

T5_VARIANTS = ["diff_drive", "dsr_ackermann", "dsr_crab", "dsr_diff", "sim_ackermann", "sim_diff", "sim_crab"]


@pytest.fixture(scope="module")
def eng():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from model_predictive_control_amd import engine
    return engine


# ------------------------------------------------------------------------------- golden vectors

@pytest.mark.parametrize("tag,vname,keys", [
    ("dd", "diff_drive", ("dd_lin_theta", None)),
    ("dsr", "dsr_ackermann", ("dsr_lin_theta", "dsr_lin_u")),
    ("sim", "sim_ackermann", ("dsr_lin_theta", "dsr_lin_u")),
    ("mpcpy", "mpc_py", ("dsr_lin_theta", "dsr_lin_u")),
])
def test_linearize_golden(eng, golden, tag, vname, keys):
    from model_predictive_control_amd import variants
    v = variants.get(vname)
    th = golden[keys[0]]
    u = golden[keys[1]] if keys[1] else np.stack([golden["dd_lin_v"], np.zeros_like(th)], axis=1)
    A, B, C = eng.linearize_batch(v, th, u)
    np.testing.assert_allclose(A.cpu().numpy(), golden[f"{tag}_lin_A"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(B.cpu().numpy(), golden[f"{tag}_lin_B"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(C.cpu().numpy(), golden[f"{tag}_lin_C"], rtol=0, atol=1e-13)


@pytest.mark.parametrize("vname,key,T", [("diff_drive", "dd_roll_T5", 5), ("diff_drive", "dd_roll_T30", 30),
                                         ("dsr_diff", "dsr_roll_T5", 5), ("dsr_diff", "dsr_roll_T30", 30),
                                         ("mpc_py", "mpcpy_roll", 10)])
def test_rollout_golden(eng, golden, vname, key, T):
    from model_predictive_control_amd import variants
    v = variants.get(vname, T=T)
    got = eng.rollout_batch(v, golden["roll_x0"], golden[f"{key}_u"]).cpu().numpy()
    np.testing.assert_allclose(got, golden[f"{key}_xbar"], rtol=0, atol=1e-12)


def _course_table(eng, g):
    cs = [(g["eight_x"], g["eight_y"], g["eight_yaw_smooth"])]
    for k in range(3):
        cs.append((g[f"spline{k}_rx"], g[f"spline{k}_ry"], g[f"spline{k}_ryaw_smooth"]))
    return eng.CourseTable(cs)


def _same_distance(got, want):
    """`mind` (unused by every caller in the reference): sqrt of the bit-exact minimum, so equal
    to an ulp; its sign comes from atan2, whose last bit may differ between libm and the
    device library, which can only matter when the angle is within rounding of 0."""
    bad = np.abs(got - want) > 4e-16 * np.maximum(1.0, np.abs(want))
    assert bad.sum() == 0, (int(bad.sum()), got[bad][:5], want[bad][:5])


def test_nearest_and_reference_golden(eng, golden):
    g = golden
    tab = _course_table(eng, g)
    rows = g["near_rows"]
    state = np.stack([rows[:, 1], rows[:, 2], rows[:, 3], rows[:, 4]], axis=1)
    cid, pind = rows[:, 0].astype(np.int32), rows[:, 5].astype(np.int32)
    ind, mind, xr = eng.nearest_ref_batch(1, 5, 0.2, 2, state, cid, pind, tab)
    assert np.array_equal(ind.cpu().numpy(), g["ref2_ind"])           # bit-exact index
    assert np.array_equal(xr.cpu().numpy(), g["ref2_xref"])
    _same_distance(mind.cpu().numpy(), g["near_mind"])
    ind, mind, xr = eng.nearest_ref_batch(0, 5, 0.2, 0.1, state, cid, pind, tab)
    assert np.array_equal(ind.cpu().numpy(), g["ref1_ind"]) and np.array_equal(xr.cpu().numpy(), g["ref1_xref"])
    ind, mind, xr = eng.nearest_ref_batch(0, 5, 0.2, 0.05, state, cid, pind, tab)
    assert np.array_equal(ind.cpu().numpy(), g["ref1b_ind"]) and np.array_equal(xr.cpu().numpy(), g["ref1b_xref"])
    ind, mind, xr = eng.nearest_ref_batch(1, 30, 0.2, 2, state, cid, pind, tab, sign=False)
    assert np.array_equal(xr.cpu().numpy(), g["ref2_T30_xref"])
    _same_distance(mind.cpu().numpy(), g["near_mind_sim"])


def test_nearest_empty_window_and_wide_search(eng, golden):
    tab = _course_table(eng, golden)
    L = int(tab.clen_host[0])
    state = np.array([[0.0, 0.0, 0.0, 0.1], [1.0, 2.0, 0.0, 0.1]])
    ind, mind, xr = eng.nearest_ref_batch(1, 5, 0.2, 2, state, [0, 0], [L, L - 1], tab)
    ind = ind.cpu().numpy()
    assert ind[0] == -1 and np.isnan(xr.cpu().numpy()[0]).all()     # min([]) raises in the reference
    assert ind[1] == L - 1
    # search windows wider than one 16-lane sweep
    rng = np.random.default_rng(3)
    B = 500
    cid = rng.integers(0, 4, B).astype(np.int32)
    pind = (rng.random(B) * (tab.clen_host[cid] - 1)).astype(np.int32)
    st = np.stack([rng.uniform(-5, 30, B), rng.uniform(-8, 8, B), np.zeros(B), rng.uniform(0, 0.5, B)], axis=1)
    cx, cy, cyaw = tab.cx.cpu().numpy(), tab.cy.cpu().numpy(), tab.cyaw.cpu().numpy()
    for ns in (1, 7, 16, 17, 40):
        ind, mind, xr = eng.nearest_ref_batch(1, 5, 0.2, 2, st, cid, pind, tab, n_search=ns)
        wi, wm, wx = orc.nearest_ref_batch(1, 5, 0.2, 2, st, cid, pind, cx, cy, cyaw, tab.clen_host, n_search=ns)
        assert np.array_equal(ind.cpu().numpy(), wi) and np.array_equal(xr.cpu().numpy(), wx)
        _same_distance(mind.cpu().numpy(), wm)


def test_nearest_with_a_non_finite_position(eng, golden):
    """x = NaN makes every distance NaN; Python's min()/index() then return the first element
    of the window, and so must the kernel (it must not index with an unset argmin)."""
    tab = _course_table(eng, golden)
    cx, cy, cyaw = tab.cx.cpu().numpy(), tab.cy.cpu().numpy(), tab.cyaw.cpu().numpy()
    state = np.array([[np.nan, 0.0, 0.0, 0.1], [1.0, np.inf, 0.0, 0.1], [1.0, 2.0, 0.0, 0.1]])
    cid, pind = np.zeros(3, np.int32), np.array([3, 0, 5], np.int32)
    ind, mind, xr = eng.nearest_ref_batch(1, 5, 0.2, 2, state, cid, pind, tab)
    wi, wm, wx = orc.nearest_ref_batch(1, 5, 0.2, 2, state, cid, pind, cx, cy, cyaw, tab.clen_host)
    assert np.array_equal(ind.cpu().numpy(), wi) and wi[0] == 3 and wi[1] == 0
    assert np.array_equal(xr.cpu().numpy(), wx)
    assert np.isnan(mind.cpu().numpy()[0]) and np.isnan(wm[0])


def test_degenerate_window_parameters_stay_on_the_course_row(eng, golden):
    """Inputs where the reference's int(round(travel / dl)) raises or runs off the list: a speed
    that is not a number gives a window of NaN (the solve then reports a non-finite input), an
    infinite speed or a vanishing spacing the last point of the row, a negative previous index
    an empty window; none of them reads outside the table."""
    tab = _course_table(eng, golden)
    cx = tab.cx.cpu().numpy()
    n = int(tab.clen_host[0])
    state = np.array([[1.0, 0.5, 0.0, np.nan], [1.0, 0.5, 0.0, np.inf], [1.0, 0.5, 0.0, 0.3], [1.0, 0.5, 0.0, 0.3]])
    cid = np.zeros(4, np.int32)
    pind = np.array([0, 0, 0, -1], np.int32)
    ind, _, xr = eng.nearest_ref_batch(0, 5, 0.2, 1.0, state, cid, pind, tab)
    ind, xr = ind.cpu().numpy(), xr.cpu().numpy()
    assert ind[0] >= 0 and np.isnan(xr[0]).all()
    assert (xr[1, 0] == cx[0, n - 1]).all()
    assert np.isfinite(xr[2]).all()
    assert ind[3] == -1 and np.isnan(xr[3]).all()
    _, _, xz = eng.nearest_ref_batch(0, 5, 0.2, 1e-300, state[2:3], cid[:1], pind[:1], tab)     # spacing ~ 0
    assert (xz.cpu().numpy()[0, 0] == cx[0, n - 1]).all()
    from model_predictive_control_amd._lib import MpcxError
    with pytest.raises(MpcxError):                                                               # "zero points ahead"
        eng.nearest_ref_batch(1, 5, 0.2, 0.0, state[2:3], cid[:1], pind[:1], tab)


def test_bad_course_ids_and_rejected_courses_are_empty_windows(eng, golden):
    """A course id outside the table, a course the builder rejected (clen = -1) and one claiming
    more points than a table row holds are searched as empty windows (ind = -1, NaN window),
    never read out of bounds; `ind_out` may be the buffer `pind` came in."""
    import torch
    tab = _course_table(eng, golden)
    clen = tab.clen_host.copy()
    clen[1], clen[2] = -1, tab.Lmax + 5
    tab.clen = torch.from_numpy(clen).cuda()
    state = np.tile(np.array([[1.0, 2.0, 0.0, 0.1]]), (6, 1))
    cid = np.array([0, 1, 2, 7, -3, 3], np.int32)
    pind = np.array([4, 4, 4, 4, 4, 4], np.int32)
    ind, mind, xr = eng.nearest_ref_batch(1, 5, 0.2, 2, state, cid, pind, tab)
    ind, xr = ind.cpu().numpy(), xr.cpu().numpy()
    assert ind[0] >= 4 and ind[5] >= 4 and np.isfinite(xr[[0, 5]]).all()
    assert (ind[1:5] == -1).all() and np.isnan(xr[1:5]).all() and np.isnan(mind.cpu().numpy()[1:5]).all()
    # the closed-loop step freezes such vehicles as "window lost"
    from model_predictive_control_amd import _lib, variants
    import ctypes as C
    v = variants.get("dsr_ackermann")
    lib = _lib.load()
    B = 6
    u = torch.zeros((B, 4, 5), dtype=torch.float64, device="cuda")
    status = torch.zeros(B, dtype=torch.int32, device="cuda")
    tind = torch.full((B,), 4, dtype=torch.int32, device="cuda")
    d_state = torch.from_numpy(state).cuda()
    has_prev = torch.zeros(B, dtype=torch.uint8, device="cuda"); done = torch.zeros(B, dtype=torch.uint8, device="cuda")
    ticks = torch.zeros(B, dtype=torch.int32, device="cuda")
    d_cid = torch.from_numpy(cid).cuda()
    p = lambda t: C.c_void_p(t.data_ptr())
    _lib.check(lib.mpcx_advance_batch(C.byref(v), B, 1.5, 0.1, p(u), p(status), p(tind), p(d_cid), p(tab.cx), p(tab.cy),
                                      p(tab.clen), tab.K, tab.Lmax, p(d_state), None, p(has_prev), p(done), p(ticks), None,
                                      None))
    torch.cuda.synchronize()
    assert done.cpu().numpy().tolist() == [0, 2, 2, 2, 2, 0]
    # in place: pind and ind_out the same buffer
    d_pind = torch.from_numpy(pind).cuda()
    d_state4 = torch.from_numpy(state).cuda()
    _lib.check(lib.mpcx_nearest_ref_batch(1, 5, 0.2, 2.0, 10, 1, B, p(d_state4), p(d_cid), p(d_pind), p(tab.cx), p(tab.cy),
                                          p(tab.cyaw), p(tab.clen), tab.K, tab.Lmax, p(d_pind), None, None, None, None))
    torch.cuda.synchronize()
    assert np.array_equal(d_pind.cpu().numpy(), ind)
    # x0_out: the pose columns of the state for the solve that follows, whatever the window
    d_x0 = torch.full((B, 3), -7.0, dtype=torch.float64, device="cuda")
    d_pind = torch.from_numpy(pind).cuda()
    _lib.check(lib.mpcx_nearest_ref_batch(1, 5, 0.2, 2.0, 10, 1, B, p(d_state4), p(d_cid), p(d_pind), p(tab.cx), p(tab.cy),
                                          p(tab.cyaw), p(tab.clen), tab.K, tab.Lmax, p(d_pind), None, None, p(d_x0), None))
    torch.cuda.synchronize()
    assert np.array_equal(d_x0.cpu().numpy(), state[:, :3], equal_nan=True)


# ------------------------------------------------------------------------------- QP solve

@pytest.mark.parametrize("name", T5_VARIANTS)
def test_solve_cold_and_warm(eng, name):
    B = 4096 + 37      # ragged tail: last wavefront partly idle
    v, ov, x0, xref = make_problem(name, B)
    ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=8)
    out = eng.solve_batch(v, x0, xref)
    du, dx, dobj = compare(out, ref, problem=dict(ov=ov, x0=x0, xref=xref))
    assert du <= U_BAR and dobj <= OBJ_BAR
    # warm start from the previous optimum (unshifted, as the reference's loop does), mixed with cold
    hp = (np.arange(B) % 3 != 0).astype(np.uint8)
    ref2 = orc.iterative_batch(ov, x0, xref, ref["u"], hp, nthreads=8)
    out2 = eng.solve_batch(v, x0, xref, ref["u"], hp)
    compare(out2, ref2, problem=dict(ov=ov, x0=x0, xref=xref, uref=ref["u"], has_prev=hp))


@pytest.mark.parametrize("name", ["diff_drive", "dsr_ackermann", "dsr_crab", "dsr_diff", "sim_crab"])
def test_linear_mpc_control(eng, name):
    B = 2048
    v, ov, x0, xref = make_problem(name, B, seed=11, chain=1)
    rng = np.random.default_rng(5)
    uref = np.zeros((B, v.nu, v.T))
    uref[:, 0] = rng.uniform(-0.4, 0.4, (B, v.T))
    if v.nu == 4:
        uref[:, 2] = uref[:, 0]
        uref[:, 1] = rng.uniform(-0.5, 0.5, (B, v.T))
        uref[:, 3] = -uref[:, 1] if name == "dsr_ackermann" else uref[:, 1]
    else:
        uref[:, 1] = rng.uniform(-1.0, 1.0, (B, v.T))
    xbar = orc.rollout_batch(ov, x0, uref)
    xbar[:, :2] += rng.normal(0, 0.05, (B, 2, v.T + 1))     # any linearisation point is legal input
    ref = orc.linear_mpc_batch(ov, xref, xbar, x0, uref, nthreads=8)
    out = eng.linear_mpc_batch(v, xref, xbar, x0, uref)
    compare(out, ref, problem=dict(ov=ov, x0=x0, xref=xref, uref=uref, xbar=xbar))


def test_infeasible_nonfinite_and_tiny_batches(eng):
    v, ov, x0, xref = make_problem("dsr_ackermann", 256, seed=3)
    up = np.zeros((256, 4, 5))
    up[:, 1] = 0.2
    up[:, 3] = -0.2
    up[5, 3] = 0.9        # |uref1 - w| <= r and |uref3 + w| <= r cannot both hold -> empty window
    up[9, 1, 0] = 1.9     # first-step window entirely outside the steer box
    up[9, 3, 0] = -1.9
    x0[17, 1] = np.nan
    xref[33, 2, 4] = np.inf
    up[41, 0, 2] = np.nan
    ref = orc.iterative_batch(ov, x0, xref, up, None, nthreads=2)
    assert ref["status"][5] == O.ST_INFEASIBLE and ref["status"][9] == O.ST_INFEASIBLE
    assert ref["status"][17] == O.ST_NONFINITE and ref["status"][33] == O.ST_NONFINITE and ref["status"][41] == O.ST_NONFINITE
    out = eng.solve_batch(v, x0, xref, up)
    x0c, xrc, upc = np.nan_to_num(x0), np.nan_to_num(xref, posinf=0.0), np.nan_to_num(up)   # unsolved rows are skipped
    compare(out, ref, problem=dict(ov=ov, x0=x0c, xref=xrc, uref=upc))
    for B in (1, 2, 63, 65):
        out = eng.solve_batch(v, x0[:B], xref[:B], up[:B])
        refB = {k: a[:B] for k, a in ref.items()}
        compare(out, refB)
    out = eng.solve_batch(v, x0[:0], xref[:0])
    assert out["u"].shape[0] == 0


@pytest.mark.parametrize("B", [2048 + 21, 4096 + 21, 32768 + 85])
def test_mpc_py_horizon_10(eng, B):
    """mpc.py's controller (T = 10, dt = 0.1, input-difference cost, no rate bound on the
    speeds) on each kernel its batch size routes it to (a wavefront per instance up to 3,072,
    the T = 10 instantiation of the lane-pair kernel up to 32,768, stage-wise beyond): ragged
    batch, warm/cold mix, non-finite and infeasible instances, tiny batches, and the one-QP
    entry point."""
    v, ov, x0, xref = make_problem("mpc_py", B, seed=13)
    ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=8)
    out = eng.solve_batch(v, x0, xref)
    compare(out, ref, problem=dict(ov=ov, x0=x0, xref=xref))
    hp = (np.arange(B) % 4 != 0).astype(np.uint8)
    up = ref["u"].copy()
    up[7, 1, 0] = 3.5           # first-step window outside the steer box -> infeasible
    up[7, 3, 0] = -3.5
    x0b, xrefb = x0.copy(), xref.copy()
    x0b[11, 0] = np.nan
    xrefb[19, 1, 10] = np.inf
    ref2 = orc.iterative_batch(ov, x0b, xrefb, up, hp, nthreads=8)
    assert ref2["status"][7] == O.ST_INFEASIBLE and ref2["status"][11] == O.ST_NONFINITE and ref2["status"][19] == O.ST_NONFINITE
    out2 = eng.solve_batch(v, x0b, xrefb, up, hp)
    compare(out2, ref2)
    for n in (1, 31, 33, 64):
        outn = eng.solve_batch(v, x0b[:n], xrefb[:n], up[:n], hp[:n])
        compare(outn, {k: a[:n] for k, a in ref2.items()})
    # linear_mpc_control at an arbitrary linearisation point (controlLaw, mpc.py:117-187)
    n = 512
    rng = np.random.default_rng(2)
    uref = np.zeros((n, 4, 10))
    uref[:, 0] = rng.uniform(-0.4, 0.4, (n, 10)); uref[:, 2] = uref[:, 0]
    uref[:, 1] = rng.uniform(-0.5, 0.5, (n, 10)); uref[:, 3] = -uref[:, 1]
    xbar = orc.rollout_batch(ov, x0[:n], uref)
    refl = orc.linear_mpc_batch(ov, xref[:n], xbar, x0[:n], uref, nthreads=8)
    outl = eng.linear_mpc_batch(v, xref[:n], xbar, x0[:n], uref)
    compare(outl, refl)


@pytest.mark.parametrize("name,T,B", [("dsr_diff", 1, 64), ("dsr_ackermann", 5, 300), ("diff_drive", 12, 300),
                                      ("dsr_crab", 12, 3072 + 64), ("dsr_ackermann", 40, 200)])
def test_no_constraint_at_all(eng, name, T, B):
    """Every bound infinite: the QP has no inequality, its minimiser is the Newton point, and the
    solvers' complementarity measure is zero by definition rather than by rounding (they used to
    stop there with the starting point).  Wavefront-per-instance and stage-wise kernels, cold and
    warm, against the oracle and the certificate."""
    inf = float("inf")
    nu = 2 if name == "diff_drive" else 4
    v, ov, x0, xref = make_problem(name, B, seed=29, T=T, umax=(inf,) * nu, dumax=(inf,) * nu, first_step=(inf,) * nu)
    ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=8)
    assert (ref["status"] == 0).all() and ref["kkt"].max() <= 1e-12
    out = eng.solve_batch(v, x0, xref)
    compare(out, ref, problem=dict(ov=ov, x0=x0, xref=xref))
    assert (out["status"].cpu().numpy() == 0).all()
    assert np.abs(out["u"].cpu().numpy()).max() > 1e-3          # not the starting point
    up = ref["u"].copy()
    ref2 = orc.iterative_batch(ov, x0, xref, up, None, nthreads=8)
    out2 = eng.solve_batch(v, x0, xref, up)
    compare(out2, ref2, problem=dict(ov=ov, x0=x0, xref=xref, uref=up))


@pytest.mark.parametrize("T,B", [(16, 300), (16, 3072 + 37), (40, 200), (64, 130)])
def test_long_horizon_kernels_edge_cases(eng, T, B):
    """Horizons other than 5 and 10: batches the chip holds at once (3,072 instances at T = 16) run
    one wavefront per instance up to T = 32, larger batches and longer horizons one lane per instance with stage-wise linear algebra.
    Warm/cold mix, non-finite and infeasible instances, tiny batches, absent bounds, and the
    one-QP entry point, for each."""
    v, ov, x0, xref = make_problem("dsr_ackermann", B, seed=21, T=T)
    ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=8)
    out = eng.solve_batch(v, x0, xref)
    compare(out, ref, problem=dict(ov=ov, x0=x0, xref=xref))
    hp = (np.arange(B) % 4 != 0).astype(np.uint8)
    up = ref["u"].copy()
    up[7, 1, 0] = 3.5           # first-step window outside the steer box -> infeasible
    up[7, 3, 0] = -3.5
    x0b, xrefb = x0.copy(), xref.copy()
    x0b[11, 0] = np.nan
    xrefb[19, 1, T] = np.inf
    up[23, 0, 3] = np.nan
    ref2 = orc.iterative_batch(ov, x0b, xrefb, up, hp, nthreads=8)
    assert ref2["status"][7] == O.ST_INFEASIBLE and ref2["status"][11] == O.ST_NONFINITE
    assert ref2["status"][19] == O.ST_NONFINITE and ref2["status"][23] == O.ST_NONFINITE
    out2 = eng.solve_batch(v, x0b, xrefb, up, hp)
    compare(out2, ref2)
    for n in (1, 2, 65):
        outn = eng.solve_batch(v, x0b[:n], xrefb[:n], up[:n], hp[:n])
        compare(outn, {k: a[:n] for k, a in ref2.items()})
    # a controller without rate or first-step bounds (only the box remains)
    inf4 = (math.inf,) * 4
    v3, ov3, _, _ = make_problem("dsr_ackermann", 1, T=T, dumax=inf4, first_step=inf4)
    ref3 = orc.iterative_batch(ov3, x0, xref, None, None, nthreads=8)
    compare(eng.solve_batch(v3, x0, xref), ref3)
    # one QP at an arbitrary linearisation point
    rng = np.random.default_rng(2)
    uref = np.zeros((B, 4, T))
    uref[:, 0] = rng.uniform(-0.4, 0.4, (B, T)); uref[:, 2] = uref[:, 0]
    uref[:, 1] = rng.uniform(-0.5, 0.5, (B, T)); uref[:, 3] = -uref[:, 1]
    xbar = orc.rollout_batch(ov, x0, uref)
    refl = orc.linear_mpc_batch(ov, xref, xbar, x0, uref, nthreads=8)
    compare(eng.linear_mpc_batch(v, xref, xbar, x0, uref), refl)


def test_long_horizons(eng):
    """Horizons beyond 5: the lane-pair kernel (10), one wavefront per instance (up to 32), and the
    global-workspace kernel (longer, or chained)."""
    for name, T, B in (("mpc_py", 10, 1024), ("dsr_ackermann", 30, 512), ("diff_drive", 12, 512),
                       ("dsr_crab", 32, 256), ("dsr_ackermann", 40, 128), ("sim_ackermann", 8, 256),
                       ("diff_drive", 1, 256), ("dsr_ackermann", 2, 256), ("diff_drive", 64, 96)):
        v, ov, x0, xref = make_problem(name, B, T=T)
        ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=8)
        out = eng.solve_batch(v, x0, xref)
        compare(out, ref, problem=dict(ov=ov, x0=x0, xref=xref))
        ref2 = orc.iterative_batch(ov, x0, xref, ref["u"], None, nthreads=8)
        out2 = eng.solve_batch(v, x0, xref, ref["u"])
        compare(out2, ref2, problem=dict(ov=ov, x0=x0, xref=xref, uref=ref["u"]))


def test_headline_batch_full_size(eng):
    """The headline 65,536-instance T=5 general-bicycle batch, checked instance by instance."""
    B = 65536
    v, ov, x0, xref = make_problem("dsr_ackermann", B, seed=2003, n_courses=64)
    ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=16)
    assert ref["kkt"].max() < 1e-8          # the oracle's own certificate
    out = eng.solve_batch(v, x0, xref)
    compare(out, ref, problem=dict(ov=ov, x0=x0, xref=xref))
    # size-independent properties of the returned controls: inside the box, rate limits hold
    u = out["u"].cpu().numpy()
    assert np.abs(u[:, 0]).max() <= 0.47 + 1e-9 and np.abs(u[:, 1]).max() <= np.pi / 2 + 1e-9
    assert np.abs(np.diff(u[:, 0], axis=1)).max() <= 0.1 + 1e-9
    assert np.abs(np.diff(u[:, 1], axis=1)).max() <= np.pi / 4 * 0.2 + 1e-9
    assert np.array_equal(u[:, 0], u[:, 2]) and np.array_equal(u[:, 1], -u[:, 3])
    # idempotence: the same batch again gives bit-identical output
    out2 = eng.solve_batch(v, x0, xref)
    assert np.array_equal(out2["u"].cpu().numpy(), u)


@pytest.mark.parametrize("B", [512, 2048])
def test_ill_conditioned_long_horizon(eng, B):
    """What mpc_dsr.iterative_linear_mpc_control runs at T = 30 (its hard-wired fixed-steer QP with
    stage weights Q 2^t, mpc_dsr.py:193,349; cond(H) about 5e10), cold and warm, through the
    wavefront-per-instance kernel (512) and the stage-wise kernel (2,048): every instance solved
    and certified; the optimum is flat, so the oracle's own dense solve is the softer side and
    the controls are held to the 1e-3 bar with a median below 1e-7."""
    v, ov, x0, xref = make_problem("dsr_diff", B, T=30)
    ref = orc.iterative_batch(ov, x0, xref, None, None, nthreads=8)
    out = eng.solve_batch(v, x0, xref)
    hp = (np.arange(B) % 3 != 0).astype(np.uint8)
    ref2 = orc.iterative_batch(ov, x0, xref, ref["u"], hp, nthreads=8)
    out2 = eng.solve_batch(v, x0, xref, ref["u"], hp)
    for o, r, ur, h in ((out, ref, None, None), (out2, ref2, ref["u"], hp)):
        st = o["status"].cpu().numpy()
        assert (st == 0).all(), np.bincount(st, minlength=5)
        certify(ov, x0, xref, o, uref=ur, has_prev=h)
        both = r["status"] <= 1
        du = np.abs(o["u"].cpu().numpy() - r["u"])[both].reshape(int(both.sum()), -1).max(axis=1)
        assert du.max() <= U_BAR and np.median(du) <= 1e-7
        rel = np.abs(o["obj"].cpu().numpy() - r["obj"])[both] / np.maximum(1.0, np.abs(r["obj"][both]))
        assert rel.max() <= OBJ_BAR and np.median(rel) <= 1e-9


@pytest.mark.parametrize("name,T,controls", [("diff_drive", 5, True), ("dsr_ackermann", 5, True), ("dsr_crab", 5, True),
                                              ("dsr_diff", 5, True), ("mpc_py", 10, True), ("dsr_ackermann", 30, True),
                                              ("dsr_diff", 30, False)])
def test_gpu_answers_against_scipy_slsqp(eng, name, T, controls):
    """The GPU's answers, not the oracle's, against a third solver on the condensed QP built on the
    host: 64 instances per variant, cold and warm-linearised."""
    B = 3000 if T == 30 else 64      # T = 30: large enough for the stage-wise kernel
    v, ov, x0, xref = make_problem(name, B, seed=17, T=T)
    out = eng.solve_batch(v, x0, xref)
    slsqp_check(ov, x0, xref, out, n_subset=64, controls=controls)
    up = out["u"].cpu().numpy()
    out2 = eng.solve_batch(v, x0, xref, up)
    slsqp_check(ov, x0, xref, out2, uref=up, n_subset=64, seed=1, controls=controls)
