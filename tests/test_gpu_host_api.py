"""The call-compatible Python modules (B = 1 through the HIP engine) against the oracle.
These read like the reference's own mains: figure-eight course, closed loop, per-mode QPs."""
import math

import numpy as np
import pytest

from oracle import mpc_oracle as O

pytestmark = pytest.mark.gpu


def _eight(pop=True):
    cx, cy, cyaw = O.make_eight_shape_trajectory(400, 10)
    start = (cx[0], cy[0], cyaw[0])
    if pop:
        cx.pop(0); cy.pop(0); cyaw.pop(0)
    return cx, cy, cyaw, start


def test_diff_drive_first_tick_known_answer(capsys):
    from model_predictive_control_amd import mpc_diff_drive as m
    cx, cy, cyaw, (sx, sy, syaw) = _eight()
    cyaw = m.smooth_yaw(cyaw)
    st = m.State(x=sx, y=sy, yaw=syaw)
    ind, mind = m.calc_nearest_index(st, cx, cy, cyaw, 0)
    assert ind == 0
    xref, ind = m.calc_ref_trajectory2(st, cx, cy, cyaw, 2, 0)
    assert xref.shape == (3, 6) and ind == 0
    np.testing.assert_allclose(xref[:, 0], [0.83622771, 0.52335956, 0.56045346], atol=1e-8)
    ov, ow, ox, oy, oyaw = m.iterative_linear_mpc_control(xref, [sx, sy, syaw], None, None)
    # cold-start optimum measured on the reference QP: v at the 0.47 box on every stage
    np.testing.assert_allclose(ov, 0.47, atol=1e-6)
    np.testing.assert_allclose(ow, [3.32808e-2, 7.04498e-3, 2.46811e-4, -1.29073e-3, -1.22696e-3], atol=1e-6)
    assert ox.shape == (6,) and oyaw.shape == (6,) and ov.shape == (5,)
    want = O.iterative_linear_mpc_control(O.variant("diff_drive"), xref, [sx, sy, syaw], None)
    np.testing.assert_allclose(np.stack([ov, ow]), want.u, atol=2e-5)
    np.testing.assert_allclose(np.stack([ox, oy, oyaw]), want.x, atol=2e-5)


def test_diff_drive_scalar_functions_match_golden(golden):
    from model_predictive_control_amd import mpc_diff_drive as m
    A, B, C = m.get_linear_model_matrix(0.3, 0.7)
    assert A.shape == (3, 3) and B.shape == (3, 2) and C.shape == (3,)
    np.testing.assert_allclose([A[0, 2], A[1, 2], B[0, 0], B[1, 0], B[2, 1]],
                               [-0.03865306, 0.04589053, 0.15296844, 0.12884354, 0.2], atol=1e-8)
    np.testing.assert_allclose(C, [0.02705714, -0.03212337, 0], atol=1e-8)
    x0, u, want = golden["roll_x0"][0], golden["dd_roll_T5_u"][0], golden["dd_roll_T5_xbar"][0]
    got = m.predict_motion(list(x0), list(u[0]), list(u[1]), np.zeros((3, 6)))
    np.testing.assert_allclose(got, want, atol=1e-12)


def test_diff_drive_closed_loop_follows_oracle():
    from model_predictive_control_amd import mpc_diff_drive as m
    ticks = 12
    t, x, y, yaw, vx, w = m.main(max_ticks=ticks)
    assert len(x) == ticks + 1
    # the same loop with the oracle in place of the engine
    v = O.variant("diff_drive")
    cx, cy, cyaw, (sx, sy, syaw) = _eight()
    O.smooth_yaw(cyaw)
    px, py, pyaw, pv = sx, sy, syaw, 0.0
    tind, _ = O.calc_nearest_index(px, py, cx, cy, cyaw, 0)
    u = None
    for k in range(ticks):
        xref, tind = O.calc_ref_trajectory2(5, px, py, cx, cy, cyaw, 2, tind)
        r = O.iterative_linear_mpc_control(v, xref, [px, py, pyaw], u)
        u = r.u
        px, py, pyaw = O.step_unicycle(px, py, pyaw, u[0, 0], u[1, 0], v.dt)
        assert abs(x[k + 1] - px) < 1e-4 and abs(y[k + 1] - py) < 1e-4 and abs(yaw[k + 1] - pyaw) < 1e-4


def test_dsr_module_modes_and_return_order():
    from model_predictive_control_amd import mpc_dsr as m
    cx, cy, cyaw, (sx, sy, syaw) = _eight()
    cyaw = m.smooth_yaw(cyaw)
    st = m.State(x=sx + 0.1, y=sy - 0.1, yaw=syaw)
    xref, ind = m.calc_ref_trajectory2(st, cx, cy, cyaw, 2, 0)
    x0 = [st.x, st.y, st.yaw]
    out = m.iterative_linear_mpc_control(xref, x0, None, None, None, None)   # hard-wired "diff" QP
    want = O.iterative_linear_mpc_control(O.variant("dsr_diff"), xref, x0, None)
    ovf, ovr, osf, osr, ox, oy, oyaw = out
    np.testing.assert_allclose(np.stack([ovf, osf, ovr, osr]), want.u, atol=2e-5)
    np.testing.assert_allclose(osf, math.pi / 2) and np.testing.assert_allclose(osr, math.pi / 2)
    # second tick, warm
    out2 = m.iterative_linear_mpc_control(xref, x0, ovf, ovr, osf, osr)
    want2 = O.iterative_linear_mpc_control(O.variant("dsr_diff"), xref, x0, want.u)
    np.testing.assert_allclose(np.stack([out2[0], out2[2], out2[1], out2[3]]), want2.u, atol=2e-5)
    # the three QPs at an explicit linearisation point
    uref = np.zeros((4, 5)); uref[0] = uref[2] = 0.2; uref[1] = 0.1; uref[3] = -0.1
    xbar = m.predict_motion(x0, uref[0], uref[2], uref[1], uref[3], xref)
    np.testing.assert_allclose(xbar, O.rollout(O.variant("dsr_ackermann"), x0, uref), atol=1e-12)
    for fn, name in ((m.linear_mpc_control_ackermann, "dsr_ackermann"), (m.linear_mpc_control_diff, "dsr_diff")):
        got = fn(xref, xbar, x0, uref)
        w = O.linear_mpc_control(O.variant(name), xref, xbar, x0, uref)
        np.testing.assert_allclose(np.stack([got[0], got[2], got[1], got[3]]), w.u, atol=2e-5)
        np.testing.assert_allclose(np.stack(got[4:]), w.x, atol=2e-5)
    urefc = uref.copy(); urefc[3] = 0.1
    got = m.linear_mpc_control_crab(xref, xbar, x0, urefc)
    w = O.linear_mpc_control(O.variant("dsr_crab"), xref, xbar, x0, urefc)
    np.testing.assert_allclose(np.stack([got[0], got[2], got[1], got[3]]), w.u, atol=2e-5)
    # keyword order used by the reference's callers (mpc_dsr.py:215-220)
    A, B, C = m.get_linear_model_matrix(theta=0.4, v_f=0.3, v_r=0.3, delta_f=0.2, delta_r=-0.2)
    np.testing.assert_allclose(B[2], [-0.03973387, 0.05880399, 0.03973387, -0.05880399], atol=1e-8)
    assert m.transfrom_gbm_command(0.3, 0.3, 0.2, -0.2) == pytest.approx((0.29401997335237245, 0.0, 0.11920159847703672))


def test_failed_solve_returns_nones_and_prints(capsys):
    from model_predictive_control_amd import mpc_dsr as m
    xref = np.zeros((3, 6))
    uref = np.zeros((4, 5)); uref[1, 0] = 0.2; uref[3, 0] = 0.9     # empty first-step window
    out = m.linear_mpc_control_ackermann(xref, np.zeros((3, 6)), [0.0, 0.0, 0.0], uref)
    assert out == (None,) * 7
    assert "Error: Cannot solve mpc.." in capsys.readouterr().out


def test_sim_chained_solves_and_modes():
    from model_predictive_control_amd import mpc_sim as m
    ctl = m.MPC(show_animation=False)
    cx, cy, cyaw = m.TrajectoryGenerator().makeEightShapeTrajectory()
    cyaw = m.smooth_yaw(cyaw)
    st = m.State(x=cx[3] + 0.05, y=cy[3] - 0.05, yaw=cyaw[3])
    ind, mind = ctl.getNearestIndex(st, cx, cy, cyaw, 0)
    assert mind >= 0
    xref, ind = ctl.getReferenceTrajectory(st, cx, cy, cyaw, 2, 0)
    x0 = [st.x, st.y, st.yaw]
    for mode, name in (("ackermann", "sim_ackermann"), ("diff", "sim_diff"), ("crab", "sim_crab")):
        got = ctl.iterativeLMPC(xref, x0, None, None, None, None, mode)
        want = O.iterative_linear_mpc_control(O.variant(name), xref, x0, None)     # chain = 5
        np.testing.assert_allclose(np.stack([got[0], got[2], got[1], got[3]]), want.u, atol=1e-7)
        got2 = ctl.iterativeLMPC(xref, x0, got[0], got[1], got[2], got[3], mode)
        want2 = O.iterative_linear_mpc_control(O.variant(name), xref, x0, want.u)
        np.testing.assert_allclose(np.stack([got2[0], got2[2], got2[1], got2[3]]), want2.u, atol=1e-7)
    t, x, y, yaw, vx, vy, w, state = ctl.doSimulation(cx, cy, cyaw, m.State(x=cx[0], y=cy[0], yaw=cyaw[0]),
                                                      mode="ackermann", max_ticks=5)
    assert len(x) == 6 and math.hypot(x[-1] - x[0], y[-1] - y[0]) > 0.05


def test_mpc_py_controller_tick():
    from model_predictive_control_amd.mpc import ModelPredictiveControl
    cx, cy, cyaw, _ = _eight(pop=False)
    O.smooth_yaw(cyaw)
    ctl = ModelPredictiveControl()
    ctl.setRobotModelParameters(wheel_base=1)
    ctl.isReferenceRetrived_ = True
    ctl.pts_x_, ctl.pts_y_, ctl.pts_yaw_ = list(cx[:150]), list(cy[:150]), list(cyaw[:150])
    pts = [list(cx[:150]), list(cy[:150]), list(cyaw[:150])]
    pose = (cx[2] + 0.02, cy[2] - 0.03, cyaw[2])
    v = O.variant("mpc_py")
    u = np.zeros((4, 10))
    for tick in range(2):
        out = ctl.start(*pose)
        xref = O.prune_reference(pts, pose)
        r = O.iterative_linear_mpc_control(v, xref, list(pose), u)
        u = r.u
        np.testing.assert_allclose(ctl.control_input_, u, atol=1e-6)
        vx, vy, w = O.wheel_to_twist(u[0, 0], u[1, 0], u[2, 0], u[3, 0])
        assert out.shape == (3, 1)
        np.testing.assert_allclose(out.reshape(3), [vx, vy, w], atol=1e-6)
    assert len(ctl.predict_state_) == 20


def test_host_pipeline_matches_resident_solve():
    """Batches handed over in host memory through the three-slot pipeline (pinned buffers, a
    stream per slot) give what the resident solve gives, batch by batch, cold and warm."""
    from model_predictive_control_amd import engine
    from tests.helpers import make_problem
    B = 700
    pipe = engine.HostPipeline(make_problem("dsr_ackermann", 1)[0], B, depth=3)
    problems, slots = [], []
    for k in range(5):                      # more batches than slots: slots are reused
        v, ov, x0, xref = make_problem("dsr_ackermann", B, seed=100 + k)
        warm = k % 2 == 1
        up = np.zeros((B, 4, 5)); up[:, 0] = up[:, 2] = 0.1 * k
        hp = (np.arange(B) % 2).astype(np.uint8)
        want = engine.solve_batch(v, x0, xref, up if warm else None, hp if warm else None)
        s = pipe.acquire()
        s.x0[:] = x0; s.xref[:] = xref; s.uprev[:] = up; s.has_prev[:] = hp
        pipe.submit(s, warm=warm)
        got = {k2: a.copy() for k2, a in pipe.result(s).items()}
        for key in ("u", "x", "obj", "status"):
            np.testing.assert_array_equal(got[key], want[key].cpu().numpy())


def test_long_horizon_solves_on_different_streams_do_not_share_the_workspace_at_once():
    """The stage-wise kernel's workspace has one user at a time: calls enqueued on three streams
    back to back must each give the answer they give alone."""
    import torch
    from model_predictive_control_amd import engine
    from tests.helpers import make_problem
    B, T = 4000, 14
    probs = [make_problem("dsr_ackermann", B, seed=200 + k, T=T) for k in range(3)]
    alone = [engine.solve_batch(p[0], p[2], p[3])["u"].cpu().numpy() for p in probs]
    streams = [torch.cuda.Stream() for _ in range(3)]
    dev_in = [(torch.from_numpy(p[2]).cuda(), torch.from_numpy(p[3]).cuda()) for p in probs]
    torch.cuda.synchronize()
    outs = []
    for rep in range(3):
        outs = []
        for s, p, (x0, xref) in zip(streams, probs, dev_in):
            with torch.cuda.stream(s):
                outs.append(engine.solve_batch(p[0], x0, xref))
        torch.cuda.synchronize()
        for o, want in zip(outs, alone):
            np.testing.assert_array_equal(o["u"].cpu().numpy(), want)


def test_scenario_drivers_run_a_few_ticks(tmp_path, golden):
    """main / main1..4 of the reference's modules (figure-eight and recorded-trajectory
    scenarios) drive the closed loop on the GPU: a few ticks each, the vehicle moves and every
    logged value is finite."""
    import csv
    from model_predictive_control_amd import mpc_dsr, mpc_sim, mpc_diff_drive
    header = ["x", "y", "yaw", "vx", "vy", "w", "front angle", "front speed", "rear angle", "rear speed", "time_stamp"]
    rec = tmp_path / "recorded.csv"
    with open(rec, "w", newline="") as f:
        w = csv.writer(f); w.writerow(header); w.writerows([list(r) for r in golden["nodes"]])
    runs = [mpc_diff_drive.main(max_ticks=4), mpc_dsr.main(max_ticks=4), mpc_dsr.main2(str(rec), max_ticks=4),
            mpc_dsr.main3(str(rec), max_ticks=4), mpc_sim.main1(max_ticks=3), mpc_sim.main2(str(rec), max_ticks=3),
            mpc_sim.main3(str(rec), max_ticks=3)]
    for res in runs:
        t, x, y = res[0], res[1], res[2]
        assert len(t) >= 3 and np.isfinite(np.array(x, dtype=float)).all() and np.isfinite(np.array(y, dtype=float)).all()
    x8, y8, yaw8 = mpc_diff_drive.main2()
    assert len(x8) == 400
