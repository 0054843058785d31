"""The oracle (numpy twin and C restatement) against vectors captured from the
reference's own numpy-only functions (tests/golden/make_golden.py)."""
import math

import numpy as np
import pytest

from oracle import mpc_oracle as O
from oracle import orc

TOL = 1e-13   # closed-form float64 arithmetic; only summation order / libm ulps differ


def _courses(g):
    cs = [(g["eight_x"], g["eight_y"], g["eight_yaw_smooth"])]
    for k in range(3):
        cs.append((g[f"spline{k}_rx"], g[f"spline{k}_ry"], g[f"spline{k}_ryaw_smooth"]))
    return cs


def test_linearise_unicycle(golden):
    g = golden
    v = O.variant("diff_drive")
    for i in range(len(g["dd_lin_theta"])):
        A, B, C = O.linearize_unicycle(g["dd_lin_v"][i], g["dd_lin_theta"][i], v.dt)
        np.testing.assert_allclose(A, g["dd_lin_A"][i], rtol=0, atol=TOL)
        np.testing.assert_allclose(B, g["dd_lin_B"][i], rtol=0, atol=TOL)
        np.testing.assert_allclose(C, g["dd_lin_C"][i], rtol=0, atol=TOL)
    u = np.stack([g["dd_lin_v"], np.zeros_like(g["dd_lin_v"])], axis=1)
    A, B, C = orc.linearize_batch(v, g["dd_lin_theta"], u)
    np.testing.assert_allclose(A, g["dd_lin_A"], rtol=0, atol=TOL)
    np.testing.assert_allclose(B, g["dd_lin_B"], rtol=0, atol=TOL)
    np.testing.assert_allclose(C, g["dd_lin_C"], rtol=0, atol=TOL)


@pytest.mark.parametrize("tag,vname", [("dsr", "dsr_ackermann"), ("sim", "sim_ackermann"), ("mpcpy", "mpc_py")])
def test_linearise_dual_steer(golden, tag, vname):
    g = golden
    v = O.variant(vname)
    th, u = g["dsr_lin_theta"], g["dsr_lin_u"]
    for i in range(len(th)):
        A, B, C = O.linearize_dsr(th[i], u[i, 0], u[i, 1], u[i, 2], u[i, 3], v.dt, v.wb)
        np.testing.assert_allclose(A, g[f"{tag}_lin_A"][i], rtol=0, atol=TOL)
        np.testing.assert_allclose(B, g[f"{tag}_lin_B"][i], rtol=0, atol=TOL)
        np.testing.assert_allclose(C, g[f"{tag}_lin_C"][i], rtol=0, atol=TOL)
    A, B, C = orc.linearize_batch(v, th, u)
    np.testing.assert_allclose(A, g[f"{tag}_lin_A"], rtol=0, atol=TOL)
    np.testing.assert_allclose(B, g[f"{tag}_lin_B"], rtol=0, atol=TOL)
    np.testing.assert_allclose(C, g[f"{tag}_lin_C"], rtol=0, atol=TOL)


def test_hand_checked_known_answers(golden):
    # mpc_dsr.get_linear_model_matrix(0.4, 0.3, 0.2, 0.3, -0.2) and transfrom_gbm_command(0.3,0.3,0.2,-0.2)
    _, B, _ = O.linearize_dsr(0.4, 0.3, 0.2, 0.3, -0.2, 0.2)
    np.testing.assert_allclose(B[0], [0.08253356, -0.01693927, 0.09800666, -0.00596008], atol=1e-8)
    np.testing.assert_allclose(B[2], [-0.03973387, 0.05880399, 0.03973387, -0.05880399], atol=1e-8)
    vx, vy, w = O.wheel_to_twist(0.3, 0.2, 0.3, -0.2)
    assert abs(vx - 0.29401997335237245) < 1e-15 and abs(vy) < 1e-16 and abs(w - 0.11920159847703672) < 1e-15


def test_wheel_to_twist(golden):
    u = golden["dsr_lin_u"]
    for key in ("dsr_twist", "sim_twist", "mpcpy_twist"):
        for i in range(len(u)):
            np.testing.assert_allclose(O.wheel_to_twist(u[i, 0], u[i, 1], u[i, 2], u[i, 3]), golden[key][i],
                                       rtol=0, atol=1e-15)


@pytest.mark.parametrize("T", [5, 30])
def test_rollout(golden, T):
    g = golden
    x0 = g["roll_x0"]
    for vname, key in (("diff_drive", "dd"), ("dsr_diff", "dsr")):
        v = O.variant(vname, T=T)
        u, want = g[f"{key}_roll_T{T}_u"], g[f"{key}_roll_T{T}_xbar"]
        for i in range(len(x0)):
            np.testing.assert_allclose(O.rollout(v, x0[i], u[i]), want[i], rtol=0, atol=1e-12)
        np.testing.assert_allclose(orc.rollout_batch(v, x0, u), want, rtol=0, atol=1e-12)
    if T == 5:
        v = O.variant("sim_ackermann")
        np.testing.assert_allclose(orc.rollout_batch(v, x0, g["dsr_roll_T5_u"]), g["sim_roll_T5_xbar"],
                                   rtol=0, atol=1e-12)


def test_rollout_mpc_py(golden):
    v = O.variant("mpc_py")
    got = orc.rollout_batch(v, golden["roll_x0"], golden["mpcpy_roll_u"])
    np.testing.assert_allclose(got, golden["mpcpy_roll_xbar"], rtol=0, atol=1e-12)


def test_eight_shape_and_smooth_yaw(golden):
    x, y, yaw = O.make_eight_shape_trajectory(400, 10)
    assert np.array_equal(np.array(x), golden["eight_x"])
    assert np.array_equal(np.array(y), golden["eight_y"])
    assert np.array_equal(np.array(yaw), golden["eight_yaw"])
    assert np.array_equal(np.array(O.smooth_yaw(list(yaw))), golden["eight_yaw_smooth"])


def test_wrap_and_goal(golden):
    for a, b in zip(golden["wrap_in"], golden["wrap_out"]):
        assert O.pi_2_pi(a) == b
    for x, y, vx, tind, nind, r1, r2 in golden["check_goal"]:
        assert O.check_goal(x, y, vx, [0.5, -0.5], int(tind), int(nind)) == bool(r1)
        assert O.check_goal(x, y, vx, [0.5, -0.5], int(tind), int(nind), goal_dis=1.5, stop_speed=0.2) == bool(r2)


def test_nearest_index_bit_exact(golden):
    g = golden
    cs = _courses(g)
    for i, (c, x, y, yaw, vx, pind) in enumerate(g["near_rows"]):
        cx, cy, cyaw = cs[int(c)]
        ind, mind = O.calc_nearest_index(x, y, cx, cy, cyaw, int(pind))
        assert ind == g["near_ind"][i]
        assert mind == g["near_mind"][i]
        _, m2 = O.calc_nearest_index(x, y, cx, cy, cyaw, int(pind), signed=False)
        assert m2 == g["near_mind_sim"][i]


def test_reference_windows(golden):
    g = golden
    cs = _courses(g)
    for i, (c, x, y, yaw, vx, pind) in enumerate(g["near_rows"]):
        cx, cy, cyaw = cs[int(c)]
        xr, ind = O.calc_ref_trajectory(5, x, y, vx, cx, cy, cyaw, 0.1, int(pind), 0.2)
        assert ind == g["ref1_ind"][i] and np.array_equal(xr, g["ref1_xref"][i])
        xr, ind = O.calc_ref_trajectory(5, x, y, vx, cx, cy, cyaw, 0.05, int(pind), 0.2)
        assert ind == g["ref1b_ind"][i] and np.array_equal(xr, g["ref1b_xref"][i])
        xr, ind = O.calc_ref_trajectory2(5, x, y, cx, cy, cyaw, 2, int(pind))
        assert ind == g["ref2_ind"][i] and np.array_equal(xr, g["ref2_xref"][i])
        assert ind == g["refsim_ind"][i] and np.array_equal(xr, g["refsim_xref"][i])
        xr, _ = O.calc_ref_trajectory2(30, x, y, cx, cy, cyaw, 2, int(pind))
        assert np.array_equal(xr, g["ref2_T30_xref"][i])


def _tables(cs):
    L = max(len(c[0]) for c in cs)
    K = len(cs)
    tx = np.zeros((K, L)); ty = np.zeros((K, L)); tw = np.zeros((K, L))
    ln = np.zeros(K, dtype=np.int32)
    for k, (cx, cy, cyaw) in enumerate(cs):
        n = len(cx)
        tx[k, :n], ty[k, :n], tw[k, :n], ln[k] = cx, cy, cyaw, n
    return tx, ty, tw, ln


def test_reference_windows_c(golden):
    g = golden
    tx, ty, tw, ln = _tables(_courses(g))
    rows = g["near_rows"]
    state = np.stack([rows[:, 1], rows[:, 2], rows[:, 3], rows[:, 4]], axis=1)
    cid = rows[:, 0].astype(np.int32)
    pind = rows[:, 5].astype(np.int32)
    ind, mind, xr = orc.nearest_ref_batch(1, 5, 0.2, 2, state, cid, pind, tx, ty, tw, ln)
    assert np.array_equal(ind, g["ref2_ind"]) and np.array_equal(xr, g["ref2_xref"])
    assert np.array_equal(mind, g["near_mind"])
    ind, mind, xr = orc.nearest_ref_batch(0, 5, 0.2, 0.1, state, cid, pind, tx, ty, tw, ln)
    assert np.array_equal(ind, g["ref1_ind"]) and np.array_equal(xr, g["ref1_xref"])
    ind, mind, xr = orc.nearest_ref_batch(0, 5, 0.2, 0.05, state, cid, pind, tx, ty, tw, ln)
    assert np.array_equal(ind, g["ref1b_ind"]) and np.array_equal(xr, g["ref1b_xref"])
    ind, mind, xr = orc.nearest_ref_batch(1, 30, 0.2, 2, state, cid, pind, tx, ty, tw, ln, sign=False)
    assert np.array_equal(xr, g["ref2_T30_xref"]) and np.array_equal(mind, g["near_mind_sim"])


def test_empty_window_raises(golden):
    cx, cy, cyaw = _courses(golden)[0]
    with pytest.raises(ValueError):
        O.calc_nearest_index(0.0, 0.0, cx, cy, cyaw, len(cx))
    tx, ty, tw, ln = _tables(_courses(golden))
    ind, mind, xr = orc.nearest_ref_batch(1, 5, 0.2, 2, np.zeros((1, 4)), [0], [int(ln[0])], tx, ty, tw, ln)
    assert ind[0] == -1 and np.isnan(xr).all()


@pytest.mark.parametrize("k", [0, 1, 2])
def test_spline_course(golden, k):
    g = golden
    rx, ry, ryaw, rk, s = O.calc_spline_course(g[f"spline{k}_wx"], g[f"spline{k}_wy"], float(g[f"spline{k}_ds"]))
    assert len(rx) == len(g[f"spline{k}_rx"])
    np.testing.assert_allclose(s, g[f"spline{k}_s"], rtol=0, atol=0)
    np.testing.assert_allclose(rx, g[f"spline{k}_rx"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(ry, g[f"spline{k}_ry"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(ryaw, g[f"spline{k}_ryaw"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(rk, g[f"spline{k}_rk"], rtol=1e-10, atol=1e-10)
    if k == 0:   # hand-checked values for the docstring waypoints of the reference, ds = 0.1
        assert len(rx) == 432 and rx[1] == pytest.approx(-2.456214414403149, abs=1e-14)
        assert ryaw[0] == pytest.approx(-1.2619119773905885, abs=1e-14)


def test_mpc_py_reference_pruning(golden):
    g = golden
    pts = [list(g["eight_x"][:200]), list(g["eight_y"][:200]), list(g["eight_yaw_smooth"][:200])]
    for pose, want, ln in zip(g["mpcpy_prune_pose"], g["mpcpy_prune_xref"], g["mpcpy_prune_len"]):
        got = O.prune_reference(pts, pose)
        assert np.array_equal(got, want) and len(pts[0]) == ln
