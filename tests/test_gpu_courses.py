"""Device course builder (calc_spline_course + smooth_yaw, one block per course) against the
reference's own outputs (golden vectors) and against the host builder on random courses."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_golden_courses(golden):
    from model_predictive_control_amd import engine
    g = golden
    for ds_group in ({0, 1, 2},):
        for k in ds_group:
            ds = float(g[f"spline{k}_ds"])
            for smooth, key in ((False, "ryaw"), (True, "ryaw_smooth")):
                tab = engine.CourseTable.from_waypoints([(g[f"spline{k}_wx"], g[f"spline{k}_wy"])], ds=ds, smooth=smooth)
                n = int(tab.clen_host[0])
                assert n == len(g[f"spline{k}_rx"])        # the sample count is exact
                np.testing.assert_allclose(tab.cx.cpu().numpy()[0, :n], g[f"spline{k}_rx"], rtol=0, atol=1e-12)
                np.testing.assert_allclose(tab.cy.cpu().numpy()[0, :n], g[f"spline{k}_ry"], rtol=0, atol=1e-12)
                np.testing.assert_allclose(tab.cyaw.cpu().numpy()[0, :n], g[f"spline{k}_{key}"], rtol=0, atol=1e-12)
                np.testing.assert_allclose(tab.ck.cpu().numpy()[0, :n], g[f"spline{k}_rk"], rtol=1e-10, atol=1e-10)


def test_many_random_courses_match_the_host_builder():
    from model_predictive_control_amd import engine, cubic_spline_planner as csp
    rng = np.random.default_rng(17)
    ways = []
    for k in range(300):
        n = int(rng.integers(2, 20))
        x = np.cumsum(rng.uniform(0.5, 6.0, n)) * (1 if k % 7 else -1)
        y = rng.uniform(-6, 6, n)
        if k % 5 == 0:                      # loops: yaw winds past +-pi, smooth_yaw has work to do
            t = np.linspace(0, 2.2 * np.pi, n)
            x, y = 5 * np.cos(t) + 0.1 * rng.standard_normal(n), 5 * np.sin(t)
        ways.append((x, y))
    tab = engine.CourseTable.from_waypoints(ways, ds=0.1, smooth=True)
    cx, cy, cyaw, ck = (t.cpu().numpy() for t in (tab.cx, tab.cy, tab.cyaw, tab.ck))
    for k, (x, y) in enumerate(ways):
        rx, ry, ryaw, rk, s = csp.calc_spline_course(x, y, ds=0.1)
        ryaw = csp.smooth_yaw(list(ryaw))
        n = int(tab.clen_host[k])
        assert n == len(rx), k
        np.testing.assert_allclose(cx[k, :n], rx, rtol=0, atol=1e-10)
        np.testing.assert_allclose(cy[k, :n], ry, rtol=0, atol=1e-10)
        np.testing.assert_allclose(cyaw[k, :n], ryaw, rtol=0, atol=1e-9)
        np.testing.assert_allclose(ck[k, :n], rk, rtol=1e-7, atol=1e-8)


def test_bad_courses_are_reported():
    from model_predictive_control_amd import engine, _lib
    with pytest.raises(_lib.MpcxError):      # a repeated waypoint: the reference divides by zero
        engine.CourseTable.from_waypoints([([0.0, 1.0, 1.0, 2.0], [0.0, 0.0, 0.0, 1.0])])
    with pytest.raises(_lib.MpcxError):      # more samples than the table holds
        engine.CourseTable.from_waypoints([([0.0, 50.0], [0.0, 0.0])], ds=0.1, Lmax=100)
    tab = engine.CourseTable.from_waypoints([([0.0, 1.0], [0.0, 0.0])], ds=0.25)   # two points: a straight segment
    assert int(tab.clen_host[0]) == 4
    np.testing.assert_allclose(tab.cx.cpu().numpy()[0, :4], [0.0, 0.25, 0.5, 0.75], atol=1e-15)
