#!/usr/bin/env python3
"""Regenerates tests/golden/reference_vectors.npz by running the reference's own
numpy-only functions in the build container.

Run from the repository root:  python tests/golden/make_golden.py

How the reference is used: its modules are imported in place from
/root/reference (no bytecode is written, nothing is copied).  They all start
with `import cvxpy`, which is not installed in this image, so an EMPTY module
object is registered under that name first; none of the functions called below
touches it.  The functions that do need cvxpy (linear_mpc_control*, doLMPC_*,
controlLaw) are not called and have no golden vectors: the QP solve stays
"parity unpinned" (see oracle/mpc_oracle.py).

Only inputs and outputs (arrays of numbers) are stored.
"""
import contextlib
import io
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ.setdefault("MPLBACKEND", "Agg")

import numpy as np  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.npz")


def load_reference():
    sys.modules.setdefault("cvxpy", types.ModuleType("cvxpy"))
    sys.path.insert(0, REF)
    import cubic_spline_planner
    import mpc
    import mpc_diff_drive
    import mpc_dsr
    import mpc_sim
    return cubic_spline_planner, mpc, mpc_diff_drive, mpc_dsr, mpc_sim


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def main():
    csp, mpc, dd, dsr, sim = load_reference()
    rng = np.random.default_rng(20241111)
    g = {}

    # ---- linearisation -------------------------------------------------------------
    N = 64
    th = rng.uniform(-4.0, 4.0, N)
    vr = rng.uniform(-0.6, 0.6, N)
    A = np.zeros((N, 3, 3)); Bm = np.zeros((N, 3, 2)); Cc = np.zeros((N, 3))
    for i in range(N):
        A[i], Bm[i], Cc[i] = dd.get_linear_model_matrix(vr[i], th[i])
    g.update(dd_lin_theta=th, dd_lin_v=vr, dd_lin_A=A, dd_lin_B=Bm, dd_lin_C=Cc)

    u4 = np.stack([rng.uniform(-0.6, 0.6, N), rng.uniform(-1.7, 1.7, N),
                   rng.uniform(-0.6, 0.6, N), rng.uniform(-1.7, 1.7, N)], axis=1)
    u4[0] = 0.0                      # cold-start operating point
    u4[1] = (0.3, 0.2, 0.3, -0.2)    # a hand-checked known answer
    th4 = rng.uniform(-4.0, 4.0, N)
    th4[1] = 0.4
    for tag, fn in (("dsr", lambda t, u: dsr.get_linear_model_matrix(t, u[0], u[1], u[2], u[3])),
                    ("sim", lambda t, u: sim.GeneralBicycleModel().getRobotModelMatrice(t, u[0], u[1], u[2], u[3])),
                    ):
        A = np.zeros((N, 3, 3)); Bm = np.zeros((N, 3, 4)); Cc = np.zeros((N, 3))
        for i in range(N):
            A[i], Bm[i], Cc[i] = fn(th4[i], u4[i])
        g.update({f"{tag}_lin_A": A, f"{tag}_lin_B": Bm, f"{tag}_lin_C": Cc})
    g.update(dsr_lin_theta=th4, dsr_lin_u=u4)
    ctl = mpc.ModelPredictiveControl()
    ctl.setRobotModelParameters(wheel_base=1)
    A = np.zeros((N, 3, 3)); Bm = np.zeros((N, 3, 4)); Cc = np.zeros((N, 3))
    for i in range(N):
        A[i], Bm[i], Cc[i] = ctl.getRobotModelMatrice(th4[i], u4[i, 0], u4[i, 1], u4[i, 2], u4[i, 3])
    g.update(mpcpy_lin_A=A, mpcpy_lin_B=Bm, mpcpy_lin_C=Cc)

    # ---- wheel -> twist --------------------------------------------------------------
    tw = np.zeros((N, 3))
    tw2 = np.zeros((N, 3))
    tw3 = np.zeros((N, 3))
    for i in range(N):
        vf, sf, vr_, sr = u4[i]
        tw[i] = dsr.transfrom_gbm_command(vf, vr_, sf, sr)
        tw2[i] = sim.GeneralBicycleModel().transformWheelCommandToRobotCommand(vf, vr_, sf, sr)
        tw3[i] = ctl.getVelocitiesFromRobotModels(front_steer=sf, front_speed=vf, rear_steer=sr,
                                                  rear_speed=vr_).reshape(3)
    g.update(dsr_twist=tw, sim_twist=tw2, mpcpy_twist=tw3)

    # ---- rollouts ----------------------------------------------------------------------
    M = 24
    x0 = np.stack([rng.uniform(-10, 10, M), rng.uniform(-10, 10, M), rng.uniform(-3.5, 3.5, M)], axis=1)
    for T in (5, 30):
        dd.T = T
        dsr.T = T
        v = rng.uniform(-0.5, 0.5, (M, T)); w = rng.uniform(-3.0, 3.0, (M, T))
        xb = np.zeros((M, 3, T + 1))
        for i in range(M):
            xb[i] = quiet(dd.predict_motion, list(x0[i]), list(v[i]), list(w[i]), np.zeros((3, T + 1)))
        g.update({f"dd_roll_T{T}_u": np.stack([v, w], axis=1), f"dd_roll_T{T}_xbar": xb})
        u = np.stack([rng.uniform(-0.5, 0.5, (M, T)), rng.uniform(-1.6, 1.6, (M, T)),
                      rng.uniform(-0.5, 0.5, (M, T)), rng.uniform(-1.6, 1.6, (M, T))], axis=1)
        xb = np.zeros((M, 3, T + 1))
        for i in range(M):
            xb[i] = quiet(dsr.predict_motion, list(x0[i]), list(u[i, 0]), list(u[i, 2]), list(u[i, 1]),
                          list(u[i, 3]), np.zeros((3, T + 1)))
        g.update({f"dsr_roll_T{T}_u": u, f"dsr_roll_T{T}_xbar": xb})
        if T == 5:
            ctrl = sim.MPC(horizon=5, show_animation=False)
            xb2 = np.zeros((M, 3, T + 1))
            for i in range(M):
                xb2[i] = quiet(ctrl.predictMotion, list(x0[i]), list(u[i, 0]), list(u[i, 2]), list(u[i, 1]),
                               list(u[i, 3]), np.zeros((3, T + 1)))
            g["sim_roll_T5_xbar"] = xb2
    dd.T = 5
    dsr.T = 5
    g["roll_x0"] = x0
    T = 10
    u = np.stack([rng.uniform(-0.5, 0.5, (M, T)), rng.uniform(-1.6, 1.6, (M, T)),
                  rng.uniform(-0.5, 0.5, (M, T)), rng.uniform(-1.6, 1.6, (M, T))], axis=1)
    xb = np.zeros((M, 3, T + 1))
    for i in range(M):
        xb[i] = quiet(ctl.predictMotion, np.zeros((3, T + 1)), x0[i].reshape(3, 1), u[i])
    g.update(mpcpy_roll_u=u, mpcpy_roll_xbar=xb)

    # ---- courses -------------------------------------------------------------------------
    ex, ey, eyaw = dd.makeEightShapeTrajectory(400, 10)
    g.update(eight_x=np.array(ex), eight_y=np.array(ey), eight_yaw=np.array(eyaw))
    sm = dd.smooth_yaw(list(eyaw))
    g["eight_yaw_smooth"] = np.array(sm)
    wp = [([-2.5, 0.0, 2.5, 5.0, 7.5, 3.0, -1.0], [0.7, -6, 5, 6.5, 0.0, 5.0, -2.0], 0.1),
          ([0.0, 4.0, 8.5, 13.0, 17.5, 22.0, 27.0, 31.0], [0.0, 2.0, -3.0, 4.5, 1.0, -5.0, 0.5, 3.0], 0.1),
          ([0.0, 10.0, 20.5, 35.0, 70.5], [0.0, -6.0, 5.0, 6.5, 0.0], 0.5)]
    for k, (wx, wy, ds) in enumerate(wp):
        rx, ry, ryaw, rk, s = csp.calc_spline_course(wx, wy, ds=ds)
        g.update({f"spline{k}_wx": np.array(wx, dtype=float), f"spline{k}_wy": np.array(wy, dtype=float),
                  f"spline{k}_ds": np.array(ds), f"spline{k}_rx": np.array(rx), f"spline{k}_ry": np.array(ry),
                  f"spline{k}_ryaw": np.array(ryaw), f"spline{k}_rk": np.array(rk), f"spline{k}_s": np.array(s),
                  f"spline{k}_ryaw_smooth": np.array(dd.smooth_yaw(list(ryaw)))})

    # ---- nearest index and reference windows ------------------------------------------------
    courses = [(list(ex), list(ey), list(sm))]
    for k in range(len(wp)):
        courses.append((list(g[f"spline{k}_rx"]), list(g[f"spline{k}_ry"]), list(g[f"spline{k}_ryaw_smooth"])))
    P = 96
    rows = []   # course, x, y, yaw, vx, pind
    for c, (cx, cy, cyaw) in enumerate(courses):
        L = len(cx)
        for _ in range(P):
            kind = rng.integers(0, 6)
            if kind == 0:
                pind = int(rng.integers(max(L - 12, 0), L))       # window clipped by the end of the course
            else:
                pind = int(rng.integers(0, L))
            j = min(pind + int(rng.integers(0, 12)), L - 1)
            if kind == 1:
                x, y = cx[j], cy[j]                                  # exactly on a course point
            elif kind == 2 and j + 1 < L:
                x, y = 0.5 * (cx[j] + cx[j + 1]), 0.5 * (cy[j] + cy[j + 1])   # near-tie between neighbours
            else:
                x, y = cx[j] + rng.normal(0, 0.3), cy[j] + rng.normal(0, 0.3)
            rows.append((c, x, y, cyaw[j] + rng.normal(0, 0.2), rng.uniform(-0.5, 0.5), pind))
    rows.append((0, ex[0], ey[0], eyaw[0], 0.0, 0))                  # a hand-checked known answer
    rows = np.array(rows)
    n = rows.shape[0]
    ind1 = np.zeros(n, dtype=np.int64); mind1 = np.zeros(n); mind_sim = np.zeros(n)
    xrefA = np.zeros((n, 3, 6)); indA = np.zeros(n, dtype=np.int64)
    xrefA05 = np.zeros((n, 3, 6)); indA05 = np.zeros(n, dtype=np.int64)
    xrefB = np.zeros((n, 3, 6)); indB = np.zeros(n, dtype=np.int64)
    xrefB30 = np.zeros((n, 3, 31)); xrefS = np.zeros((n, 3, 6)); indS = np.zeros(n, dtype=np.int64)
    ctrl = sim.MPC(horizon=5, show_animation=False)
    for i, (c, x, y, yaw, vx, pind) in enumerate(rows):
        cx, cy, cyaw = courses[int(c)]
        pind = int(pind)
        st = dd.State(x=x, y=y, yaw=yaw, vx=vx)
        ind1[i], mind1[i] = dd.calc_nearest_index(st, cx, cy, cyaw, pind)
        _, mind_sim[i] = ctrl.getNearestIndex(st, cx, cy, cyaw, pind)
        xrefA[i], indA[i] = dd.calc_ref_trajectory(st, cx, cy, cyaw, 0.1, pind)
        xrefA05[i], indA05[i] = dsr.calc_ref_trajectory(dsr.State(x=x, y=y, yaw=yaw, vx=vx), cx, cy, cyaw, 0.05, pind)
        xrefB[i], indB[i] = dd.calc_ref_trajectory2(st, cx, cy, cyaw, 2, pind)
        xrefS[i], indS[i] = ctrl.getReferenceTrajectory(st, cx, cy, cyaw, 2, pind)
        dsr.T = 30
        xrefB30[i], _ = dsr.calc_ref_trajectory2(dsr.State(x=x, y=y, yaw=yaw, vx=vx), cx, cy, cyaw, 2, pind)
        dsr.T = 5
    g.update(near_rows=rows, near_ind=ind1, near_mind=mind1, near_mind_sim=mind_sim,
             ref1_xref=xrefA, ref1_ind=indA, ref1b_xref=xrefA05, ref1b_ind=indA05,
             ref2_xref=xrefB, ref2_ind=indB, ref2_T30_xref=xrefB30, refsim_xref=xrefS, refsim_ind=indS)

    # ---- small helpers ----------------------------------------------------------------------
    ang = rng.uniform(-12, 12, 64)
    g.update(wrap_in=ang, wrap_out=np.array([dd.pi_2_pi(a) for a in ang]))
    cg = []
    for _ in range(64):
        x, y, vx = rng.uniform(-3, 3), rng.uniform(-3, 3), rng.uniform(-0.4, 0.4)
        tind, nind = int(rng.integers(380, 400)), 400
        r1 = dd.check_goal(dd.State(x=x, y=y, vx=vx), [0.5, -0.5], tind, nind)
        r2 = ctrl.checkGoal(sim.State(x=x, y=y, vx=vx), [0.5, -0.5], tind, nind)
        cg.append((x, y, vx, tind, nind, float(r1), float(r2)))
    g["check_goal"] = np.array(cg)

    # ---- mpc.py stateful reference pruning ------------------------------------------------------
    ctl2 = mpc.ModelPredictiveControl()
    ctl2.isReferenceRetrived_ = True
    ctl2.pts_x_, ctl2.pts_y_, ctl2.pts_yaw_ = list(ex[:200]), list(ey[:200]), list(sm[:200])
    prn = []
    lens = []
    poses = []
    for j in (0, 3, 9, 9, 30, 31, 60):
        pose = np.array([[ex[j] + 0.05], [ey[j] - 0.05], [sm[j]]])
        prn.append(ctl2.getReferenceTrajectoryWithinHorizon(pose))
        lens.append(len(ctl2.pts_x_))
        poses.append(pose.reshape(3))
    g.update(mpcpy_prune_pose=np.array(poses), mpcpy_prune_xref=np.array(prn), mpcpy_prune_len=np.array(lens))

    # ---- mpc.py reference window by look-ahead distance (mpc.py:300-345), stateful like the above
    ctl3 = mpc.ModelPredictiveControl()
    ctl3.isReferenceRetrived_ = True
    ctl3.reference_ = [[float(ex[i]), float(ey[i]), float(sm[i])] for i in range(0, 240, 2)]
    la_pose, la_xref, la_len, la_dist = [], [], [], []
    for j, dist in ((0, 1.0), (4, 1.0), (4, 0.3), (20, 2.5), (21, 0.05), (50, 6.0), (110, 1.0)):
        ctl3.LOOKAHEAD_DIST_ = dist
        pose = np.array([[ex[j] + 0.03], [ey[j] + 0.04], [sm[j]]])
        la_xref.append(ctl3.getReferenceTrajectoryWithLookAheadDistance(pose))
        la_len.append(len(ctl3.reference_))
        la_pose.append(pose.reshape(3))
        la_dist.append(dist)
    g.update(mpcpy_la_pose=np.array(la_pose), mpcpy_la_xref=np.array(la_xref), mpcpy_la_len=np.array(la_len),
             mpcpy_la_dist=np.array(la_dist))

    # ---- node-list preprocessing: ICR-arc / crab interpolation, de-duplication, mode split ------
    # synthetic node lists in the CSV schema x | y | yaw | vx | vy | w | ... | time | mode
    n_nodes = 14
    nodes = np.zeros((n_nodes, 11))
    nodes[:, 0] = np.cumsum(rng.uniform(0.2, 1.0, n_nodes))
    nodes[:, 1] = rng.uniform(-1.0, 1.0, n_nodes)
    nodes[:, 2] = np.cumsum(rng.uniform(-0.3, 0.3, n_nodes))
    nodes[:, 3] = rng.uniform(0.1, 0.5, n_nodes)
    nodes[:, 4] = rng.uniform(-0.2, 0.2, n_nodes)
    nodes[:, 5] = rng.uniform(-0.6, 0.6, n_nodes)
    nodes[[3, 8], 5] = 0.0                       # w == 0 segments are skipped by the arc interpolator
    nodes[:, 6:10] = rng.uniform(-1, 1, (n_nodes, 4))
    nodes[:, 10] = np.arange(n_nodes) * 0.2
    modes = ["ackermann"] * 4 + ["crab"] * 3 + ["diff"] * 4 + ["crab"] * 3
    g["nodes"] = nodes
    g["nodes_mode"] = np.array([["ackermann", "crab", "diff"].index(m) for m in modes])
    node_list = [list(r) for r in nodes]
    node_list_m = [list(r) + [m] for r, m in zip(nodes, modes)]
    gen = sim.TrajectoryGenerator()
    for k_int in (5, 3):
        for md in ("ackermann", "crab"):
            px, py, pyaw = dsr.interpolateReference(node_list, interpolate_num=k_int, mode=md)
            g[f"interp_{md}_{k_int}"] = np.array([px, py, pyaw])
            sx_, sy_, syaw_ = gen.interpolateReference(node_list, interpolate_num=k_int, mode=md)
            assert np.array_equal(np.array([sx_, sy_, syaw_]), g[f"interp_{md}_{k_int}"])   # the two files agree
    px, py, pyaw, pm = gen.interpolateReference2(node_list_m, interpolate_num=5)
    g["interp2_xyyaw"] = np.array([px, py, pyaw])
    g["interp2_mode"] = np.array([["ackermann", "crab", "diff"].index(m) for m in pm])
    g["split_groups"] = np.array(gen.splitTrajectoryWithMotionModes(pm))
    g["split_groups_short"] = np.array(gen.splitTrajectoryWithMotionModes(["crab", "crab", "diff", "diff", "diff", "crab"]))
    rx_ = [0.0, 0.001, 0.002, 0.5, 0.5, 0.501, 1.0, 1.0]
    ry_ = [0.0, 0.001, 0.0, 0.2, 0.2, 0.2, 0.4, 0.4]
    ryaw_ = [0.0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7]
    g["dedup_in"] = np.array([rx_, ry_, ryaw_])
    g["dedup_out"] = np.array(dsr.removeRepeatedPoints(rx_, ry_, ryaw_))
    assert np.array_equal(np.array(gen.removeRepeatedPoints(rx_, ry_, ryaw_)), g["dedup_out"])

    np.savez_compressed(OUT, **g)
    print("wrote", OUT, "with", len(g), "arrays,", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
