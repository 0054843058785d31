"""Device-resident closed loop (reference window -> solve -> update_state -> check_goal per
tick, no host round trip) against the same loop built from the oracle's pieces."""
import numpy as np
import pytest
import torch

from oracle import mpc_oracle as O

pytestmark = pytest.mark.gpu


def _setup(name, B, seed, **overrides):
    from model_predictive_control_amd import engine, variants, workload
    v = variants.get(name, **overrides)
    ov = O.variant(name, **overrides)
    courses = workload.make_courses(6, seed=seed)
    inst = workload.make_instances(courses, B, seed=seed + 1)
    table = engine.CourseTable(courses)
    tabs = (table.cx.cpu().numpy(), table.cy.cpu().numpy(), table.cyaw.cpu().numpy(), table.clen_host)
    state = inst["state"][:, :4].copy()
    return engine, v, ov, table, tabs, state, inst["course_id"], inst["pind"]


@pytest.mark.parametrize("name", ["diff_drive", "dsr_ackermann", "dsr_diff", "sim_crab", "sim_ackermann", "sim_diff"])
def test_closed_loop_matches_oracle_loop(name):
    B, n = 96, 15
    engine, v, ov, table, tabs, state, cid, pind = _setup(name, B, 31)
    loop = engine.ClosedLoop(v, table, state, cid, log_ticks=n, pind0=pind).run(n)
    torch.cuda.synchronize()
    want = O.closed_loop_batch(ov, *tabs, state, cid, n, pind0=pind)
    np.testing.assert_array_equal(loop.done.cpu().numpy(), want["done"])
    np.testing.assert_array_equal(loop.ticks.cpu().numpy(), want["ticks"])
    log, tind = loop.log.cpu().numpy(), loop.target_ind.cpu().numpy()
    # each tick's QP answer (the last of five chained ones for the mpc_sim variants) agrees to
    # ~1e-12; 15 ticks of plant integrate the differences
    np.testing.assert_array_equal(tind, want["target_ind"])
    np.testing.assert_allclose(log, want["log"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(loop.state.cpu().numpy(), want["state"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(loop.twist.cpu().numpy(), want["twist"], atol=1e-6, rtol=0)
    # the vehicles moved along their courses
    assert (tind >= pind).all()
    assert np.median(np.hypot(*(log[-1][:, :2] - log[0][:, :2]).T)) > 0.05


@pytest.mark.parametrize("B", [96, 3200])
def test_closed_loop_at_a_long_horizon(B):
    """T = 12: the wavefront-per-instance kernel (small fleet) and the stage-wise kernel (large
    fleet, with its workspace) inside the tick."""
    n = 8
    engine, v, ov, table, tabs, state, cid, pind = _setup("dsr_ackermann", B, 37, T=12)
    loop = engine.ClosedLoop(v, table, state, cid, log_ticks=n, pind0=pind).run(n)
    torch.cuda.synchronize()
    want = O.closed_loop_batch(ov, *tabs, state, cid, n, pind0=pind)
    np.testing.assert_array_equal(loop.done.cpu().numpy(), want["done"])
    np.testing.assert_array_equal(loop.target_ind.cpu().numpy(), want["target_ind"])
    np.testing.assert_allclose(loop.log.cpu().numpy(), want["log"], atol=1e-6, rtol=0)
    np.testing.assert_allclose(loop.state.cpu().numpy(), want["state"], atol=1e-6, rtol=0)


def test_goal_freezes_a_vehicle_and_failure_means_cold_start():
    from model_predictive_control_amd import engine, variants
    v = variants.get("diff_drive")
    # a short straight course: the vehicle starts next to its end, at rest -> goal at tick 1
    cx = np.linspace(0.0, 2.0, 21); cy = np.zeros(21); cyaw = np.zeros(21)
    table = engine.CourseTable([(cx, cy, cyaw)])
    state = np.array([[1.9, 0.0, 0.0, 0.0], [0.2, 0.05, 0.0, 0.0], [np.nan, 0.0, 0.0, 0.0]])
    loop = engine.ClosedLoop(v, table, state, np.zeros(3, np.int32)).run(4)
    torch.cuda.synchronize()
    done, ticks = loop.done.cpu().numpy(), loop.ticks.cpu().numpy()
    assert done[0] == 1 and ticks[0] < 4          # reached the goal, then frozen
    assert done[1] == 0 and ticks[1] == 4
    # non-finite state: every solve fails, the state never moves, every tick is a cold start
    assert loop.has_prev.cpu().numpy()[2] == 0 and np.isnan(loop.state.cpu().numpy()[2, 0])
    frozen = loop.state.cpu().numpy()[0].copy()
    loop.run(2)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(loop.state.cpu().numpy()[0], frozen)


def test_course_ids_outside_the_table_lose_their_window():
    """A vehicle whose course id is not a row of the table has an empty search window on every
    kernel of the tick (where the reference raises): it is marked lost on the first tick and never
    moves; its neighbours are unaffected."""
    from model_predictive_control_amd import engine, variants
    v = variants.get("dsr_ackermann")
    cx = np.linspace(0.0, 20.0, 201); cy = np.zeros(201); cyaw = np.zeros(201)
    table = engine.CourseTable([(cx, cy, cyaw)])
    state = np.array([[0.1, 0.02, 0.0, 0.0]] * 3)
    loop = engine.ClosedLoop(v, table, state, np.array([0, 5, -2], np.int32)).run(3)
    torch.cuda.synchronize()
    done, got = loop.done.cpu().numpy(), loop.state.cpu().numpy()
    assert done.tolist() == [0, 2, 2]
    np.testing.assert_array_equal(got[1], state[1])
    np.testing.assert_array_equal(got[2], state[2])
    assert got[0, 0] > state[0, 0]


def test_graph_replay_equals_eager_ticks():
    """The tick (reference window -> solve -> advance) allocates nothing, so it can be
    recorded as a HIP graph; replaying it must give exactly the eager loop's states."""
    B, n = 512, 8
    engine, v, ov, table, tabs, state, cid, pind = _setup("dsr_ackermann", B, 41)
    eager = engine.ClosedLoop(v, table, state, cid, pind0=pind).run(n + 2)
    graph = engine.ClosedLoop(v, table, state, cid, pind0=pind).capture()   # capturing runs 2 warm-up ticks
    graph.run(n)
    torch.cuda.synchronize()
    assert graph.n_done_ticks == eager.n_done_ticks == n + 2
    for name in ("state", "twist", "target_ind", "done", "ticks", "has_prev"):
        np.testing.assert_array_equal(getattr(graph, name).cpu().numpy(), getattr(eager, name).cpu().numpy())
    np.testing.assert_array_equal(graph.last_u.cpu().numpy(), eager.last_u.cpu().numpy())


def test_capture_of_a_workspace_loop():
    """A loop whose solve streams through a workspace (T = 30 beyond the chip-resident batch):
    captured on the library's fixed workspace when it fits (2,000 vehicles: 27 MB of 64 MiB), and
    then equal to the eager loop bit for bit; refused up front when it does not (6,000: 81 MB; the
    growing workspace would be allocated and waited for inside the capture), after which the loop
    still runs eagerly."""
    from model_predictive_control_amd import _lib
    engine, v, ov, table, tabs, state, cid, pind = _setup("dsr_ackermann", 2000, 33, T=30)
    eager = engine.ClosedLoop(v, table, state, cid, pind0=pind).run(6)
    graph = engine.ClosedLoop(v, table, state, cid, pind0=pind).capture()
    graph.run(4)
    torch.cuda.synchronize()
    assert graph.n_done_ticks == eager.n_done_ticks == 6
    for name in ("state", "target_ind", "done", "ticks", "has_prev"):
        np.testing.assert_array_equal(getattr(graph, name).cpu().numpy(), getattr(eager, name).cpu().numpy())
    np.testing.assert_array_equal(graph.last_u.cpu().numpy(), eager.last_u.cpu().numpy())
    assert (graph.out["status"].cpu().numpy() <= 1).mean() > 0.99

    engine, v, ov, table, tabs, state, cid, pind = _setup("dsr_ackermann", 6000, 33, T=30)
    loop = engine.ClosedLoop(v, table, state, cid, pind0=pind)
    with pytest.raises(_lib.MpcxError, match="workspace"):
        loop.capture()
    loop.run(2)
    torch.cuda.synchronize()
    assert int((loop.ticks == 2).sum().item()) == 6000
    assert (loop.out["status"].cpu().numpy() <= 1).mean() > 0.99 and float(loop.uprev.abs().sum().item()) > 0


def test_capture_of_a_chip_resident_long_horizon_loop():
    """A small long-horizon fleet (T = 30, 300 vehicles) goes one wavefront per instance, followed by
    the stage-wise pass over what that leaves unsettled, which needs the workspace: captured, the pair
    runs on the library's fixed workspace and equals the eager loop bit for bit."""
    engine, v, ov, table, tabs, state, cid, pind = _setup("dsr_diff", 300, 35, T=30)
    assert engine._lib.load().mpcx_workspace_bytes(engine.C.byref(v), 300) == 55 * 30 * 8 * 320
    eager = engine.ClosedLoop(v, table, state, cid, pind0=pind).run(6)
    graph = engine.ClosedLoop(v, table, state, cid, pind0=pind).capture()
    graph.run(4)
    torch.cuda.synchronize()
    assert graph.n_done_ticks == eager.n_done_ticks == 6
    for name in ("state", "target_ind", "done", "ticks", "has_prev"):
        np.testing.assert_array_equal(getattr(graph, name).cpu().numpy(), getattr(eager, name).cpu().numpy())
    np.testing.assert_array_equal(graph.last_u.cpu().numpy(), eager.last_u.cpu().numpy())
    assert (graph.out["status"].cpu().numpy() == 0).all()
