"""Single-instance (B = 1) marshalling shared by the call-compatible modules: Python
lists / numpy arrays in, one launch of the batched HIP operators, numpy arrays out."""
from __future__ import annotations

import numpy as np
import torch

from . import _lib, engine


def solve_one(v, xref, x0, ucols):
    """One iterative_linear_mpc_control call.  ucols: (NU, T) array-like or None (cold).
    Returns (ok, u (NU,T), x (3,T+1), status)."""
    xref = np.asarray(xref, dtype=np.float64).reshape(1, 3, v.T + 1)
    x0 = np.asarray([float(x0[0]), float(x0[1]), float(x0[2])]).reshape(1, 3)
    up = None if ucols is None else np.asarray(ucols, dtype=np.float64).reshape(1, v.nu, v.T)
    out = engine.solve_batch(v, x0, xref, up)
    return _unpack(out)


def linear_one(v, xref, xbar, x0, uref):
    """One linear_mpc_control call at the given linearisation point."""
    xref = np.asarray(xref, dtype=np.float64).reshape(1, 3, v.T + 1)
    xbar = np.asarray(xbar, dtype=np.float64).reshape(1, 3, v.T + 1)
    x0 = np.asarray([float(x0[0]), float(x0[1]), float(x0[2])]).reshape(1, 3)
    uref = np.asarray(uref, dtype=np.float64).reshape(1, v.nu, v.T)
    return _unpack(engine.linear_mpc_batch(v, xref, xbar, x0, uref))


def _unpack(out):
    status = int(out["status"].cpu()[0])
    ok = status in (_lib.ST_OK, _lib.ST_INACCURATE)
    if not ok:
        # what every solve function of the reference does on a failed solve
        # (mpc_diff_drive.py:209-215)
        print("Error: Cannot solve mpc..")
        return False, None, None, status
    return True, out["u"].cpu().numpy()[0], out["x"].cpu().numpy()[0], status


def rollout_one(v, x0, ucols, shape_like=None):
    """predict_motion: xbar (3, T+1)."""
    x0 = np.asarray([float(x0[0]), float(x0[1]), float(x0[2])]).reshape(1, 3)
    u = np.asarray(ucols, dtype=np.float64).reshape(1, v.nu, v.T)
    return engine.rollout_batch(v, x0, u).cpu().numpy()[0]


def linearize_one(v, theta, ucol):
    """get_linear_model_matrix: (A (3,3), B (3,NU), C (3,))."""
    A, B, C = engine.linearize_batch(v, np.asarray([float(theta)]), np.asarray(ucol, dtype=np.float64).reshape(1, v.nu))
    return A.cpu().numpy()[0], B.cpu().numpy()[0], C.cpu().numpy()[0]


_TABLES = {}      # content fingerprint of (cx, cy, cyaw) -> device CourseTable, most recent last
_TABLES_MAX = 8


def course_table(cx, cy, cyaw):
    """Device table of one course given as lists, cached by content: the reference's callers pass
    the same lists tick after tick (and smooth or pop them in place in between, which a cache
    keyed by identity would miss), so a tick costs a hash of the lists instead of an upload."""
    ax, ay, aw = (np.ascontiguousarray(a, dtype=np.float64) for a in (cx, cy, cyaw))
    key = (ax.size, hash(ax.tobytes()), hash(ay.tobytes()), hash(aw.tobytes()))
    entry = _TABLES.pop(key, None)
    # a hit counts only if the arrays themselves are equal: two courses may share a hash
    if entry is not None and not (np.array_equal(entry[1], ax) and np.array_equal(entry[2], ay)
                                  and np.array_equal(entry[3], aw)):
        entry = None
    if entry is None:
        entry = (engine.CourseTable([(ax, ay, aw)]), ax.copy(), ay.copy(), aw.copy())
        while len(_TABLES) >= _TABLES_MAX:
            _TABLES.pop(next(iter(_TABLES)))
    _TABLES[key] = entry
    return entry[0]


def nearest_ref_one(mode, T, dt, param, state, cx, cy, cyaw, pind, n_search, sign=True):
    """calc_nearest_index + reference window for one state on one course given as lists.
    Raises ValueError on the empty search window like min([]) does in the reference
    (mpc_diff_drive.py:110)."""
    table = course_table(cx, cy, cyaw)
    st = np.asarray([[float(state.x), float(state.y), float(getattr(state, "yaw", 0.0)),
                      float(getattr(state, "vx", 0.0))]])
    ind, mind, xref = engine.nearest_ref_batch(mode, T, dt, param, st, None, np.asarray([int(pind)], dtype=np.int32),
                                               table, n_search=n_search, sign=sign)
    ind = int(ind.cpu()[0])
    if ind < 0:
        raise ValueError("min() arg is an empty sequence")
    return ind, float(mind.cpu()[0]), xref.cpu().numpy()[0]


def closed_loop_one(v, cx, cy, cyaw, state, goal_dis, stop_speed, dt, max_time, max_ticks=None, n_search=10,
                    chunk=64):
    """do_simulation for one vehicle on the device-resident loop (engine.ClosedLoop with B = 1):
    initial yaw compensation, first nearest index from the course start, then per tick reference
    window -> solve -> update_state -> check_goal with no host round trip; the host looks at the
    `done` flag once per `chunk` ticks.  Returns per-tick lists t, x, y, yaw, vx, vy, w (first
    entry = the initial state) and updates `state` to the final one.  Raises ValueError where the
    reference does (empty search window)."""
    table = course_table(cx, cy, cyaw)
    # ticks the reference's `while max_time >= time` loop can take (time accumulates in floating point)
    limit, tm = 0, 0.0
    while max_time >= tm and (max_ticks is None or limit < max_ticks):
        tm += dt
        limit += 1
    st0 = np.asarray([[float(state.x), float(state.y), float(state.yaw), float(getattr(state, "vx", 0.0))]])
    loop = engine.ClosedLoop(v, table, st0, None, goal_dis=goal_dis, stop_speed=stop_speed, n_search=n_search,
                             log_ticks=limit, log_twist=True)
    if int(loop.target_ind.cpu()[0]) < 0:
        raise ValueError("min() arg is an empty sequence")
    done = 0
    while loop.n_done_ticks < limit and not done:
        loop.run(min(chunk, limit - loop.n_done_ticks))
        done = int(loop.done.cpu()[0])
    n = int(loop.ticks.cpu()[0])
    log = loop.log[:n + 1, 0].cpu().numpy()
    tw = loop.twist_log[:n + 1, 0].cpu().numpy()
    # the reference logs the initial state's own vy and w as the first entry (mpc_dsr.py:531-536)
    tw[0, 0], tw[0, 1] = float(getattr(state, "vy", 0.0)), float(getattr(state, "w", 0.0))
    t = [0.0]
    for _ in range(n):
        t.append(t[-1] + dt)
    # the reference has advanced `state` in place tick by tick, also when it then raises
    state.x, state.y, state.yaw, state.vx = (float(a) for a in log[n])
    if n > 0:
        if hasattr(state, "vy"):
            state.vy = float(tw[n, 0])
        if hasattr(state, "w"):
            state.w = float(tw[n, 1])
    if done == 2:
        raise ValueError("min() arg is an empty sequence")
    return dict(t=t, x=log[:, 0].tolist(), y=log[:, 1].tolist(), yaw=log[:, 2].tolist(), vx=log[:, 3].tolist(),
                vy=tw[:, 0].tolist(), w=tw[:, 1].tolist(), goal=done == 1, target_ind=int(loop.target_ind.cpu()[0]))
