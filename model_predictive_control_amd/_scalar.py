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


def nearest_ref_one(mode, T, dt, param, state, cx, cy, cyaw, pind, n_search, sign=True):
    """calc_nearest_index + reference window for one state on one course given as lists.
    Raises ValueError on the empty search window like min([]) does in the reference
    (mpc_diff_drive.py:110)."""
    table = engine.CourseTable([(np.asarray(cx, dtype=np.float64), np.asarray(cy, dtype=np.float64),
                                 np.asarray(cyaw, dtype=np.float64))])
    st = np.asarray([[float(state.x), float(state.y), float(getattr(state, "yaw", 0.0)),
                      float(getattr(state, "vx", 0.0))]])
    ind, mind, xref = engine.nearest_ref_batch(mode, T, dt, param, st, None, np.asarray([int(pind)], dtype=np.int32),
                                               table, n_search=n_search, sign=sign)
    ind = int(ind.cpu()[0])
    if ind < 0:
        raise ValueError("min() arg is an empty sequence")
    return ind, float(mind.cpu()[0]), xref.cpu().numpy()[0]
