"""Recorded-trajectory preprocessing (CSV node lists, ICR-arc / crab interpolation,
de-duplication, split by motion mode) against the reference's own outputs."""
import csv

import numpy as np
import pytest

from model_predictive_control_amd import trajectory_io as tio

MODES = ["ackermann", "crab", "diff"]


def _nodes(golden, with_mode=False):
    rows = [list(r) for r in golden["nodes"]]
    if with_mode:
        rows = [r + [MODES[m]] for r, m in zip(rows, golden["nodes_mode"])]
    return rows


@pytest.mark.parametrize("k", [5, 3])
@pytest.mark.parametrize("mode", ["ackermann", "crab"])
def test_interpolation_between_nodes(golden, mode, k):
    got = np.array(tio.interpolate_nodes(_nodes(golden), k, mode))
    want = golden[f"interp_{mode}_{k}"]
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-13)
    if mode == "ackermann":      # 'diff' takes the same branch in the reference
        np.testing.assert_array_equal(np.array(tio.interpolate_nodes(_nodes(golden), k, "diff")), got)
    assert tio.interpolate_nodes(_nodes(golden), k, "sideways") is None


def test_interpolation_by_mode_and_split(golden):
    px, py, pyaw, pm = tio.interpolate_nodes_by_mode(_nodes(golden, True), 5)
    np.testing.assert_allclose(np.array([px, py, pyaw]), golden["interp2_xyyaw"], rtol=0, atol=1e-13)
    assert [MODES.index(m) for m in pm] == list(golden["interp2_mode"])
    assert tio.split_by_motion_mode(pm) == golden["split_groups"].tolist()
    # the reference never closes the last run
    assert tio.split_by_motion_mode(["crab", "crab", "diff", "diff", "diff", "crab"]) == golden["split_groups_short"].tolist()
    assert tio.split_by_motion_mode([]) == [] and tio.split_by_motion_mode(["crab"]) == []


def test_remove_repeated_points(golden):
    got = np.array(tio.remove_repeated_points(*golden["dedup_in"]))
    np.testing.assert_array_equal(got, golden["dedup_out"])
    assert tio.remove_repeated_points([], [], []) == ([], [], [])


def test_csv_round_trip(tmp_path, golden):
    header = ["x", "y", "yaw", "vx", "vy", "w", "fa", "fs", "ra", "rs", "t"]
    plain = tmp_path / "nodes.csv"
    with open(plain, "w", newline="") as f:
        w = csv.writer(f); w.writerow(header); w.writerows(_nodes(golden))
    back = tio.read_nodes_csv(str(plain))
    np.testing.assert_array_equal(np.array(back), golden["nodes"])      # repr() round-trips float64
    moded = tmp_path / "nodes_mode.csv"
    with open(moded, "w", newline="") as f:
        w = csv.writer(f); w.writerow(header + ["mode"]); w.writerows(_nodes(golden, True))
    back = tio.read_nodes_csv(str(moded), with_mode=True)
    assert [r[11] for r in back] == [MODES[m] for m in golden["nodes_mode"]]
    np.testing.assert_array_equal(np.array([r[:11] for r in back]), golden["nodes"])


def test_reference_named_wrappers(golden):
    from model_predictive_control_amd import mpc_dsr, mpc_sim
    gen = mpc_sim.TrajectoryGenerator()
    a = mpc_dsr.interpolateReference(_nodes(golden), interpolate_num=5, mode="crab")
    b = gen.interpolateReference(_nodes(golden), interpolate_num=5, mode="crab")
    np.testing.assert_array_equal(np.array(a), np.array(b))
    np.testing.assert_allclose(np.array(a), golden["interp_crab_5"], atol=1e-13)
    assert gen.splitTrajectoryWithMotionModes(["a", "a", "b", "b"]) == [[0, 1]]
    assert mpc_dsr.removeRepeatedPoints([0, 0, 1], [0, 0, 1], [1, 2, 3]) == ([0, 1], [0, 1], [1, 3])


def test_mpc_py_reference_window_by_look_ahead_distance():
    """ModelPredictiveControl.getReferenceTrajectoryWithLookAheadDistance against the windows and
    the in-place pruning the reference's method produced on the same node list and poses."""
    import os
    from model_predictive_control_amd import mpc
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "reference_vectors.npz"))
    ctl = mpc.ModelPredictiveControl()
    ctl.isReferenceRetrived_ = True
    ex, ey, sm = g["eight_x"], g["eight_y"], g["eight_yaw_smooth"]
    ctl.reference_ = [[float(ex[i]), float(ey[i]), float(sm[i])] for i in range(0, 240, 2)]
    for pose, want, ln, dist in zip(g["mpcpy_la_pose"], g["mpcpy_la_xref"], g["mpcpy_la_len"], g["mpcpy_la_dist"]):
        ctl.LOOKAHEAD_DIST_ = float(dist)
        got = ctl.getReferenceTrajectoryWithLookAheadDistance(pose.reshape(3, 1))
        assert np.array_equal(got, want)
        assert len(ctl.reference_) == ln
    st = mpc.State(1, 2, 3, 4, 5, 6, 7, 8, 9, 10)
    assert (st.x_, st.w_, st.rear_speed_) == (1, 6, 10)
