"""ctypes loader for the C restatement (oracle/mpc_oracle.c).

Test infrastructure: imported only by tests/, the smoke check and the
CPU-baseline leg of the benchmark.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

from . import mpc_oracle as O

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmpc_oracle.so")


class OrcVariant(C.Structure):
    _fields_ = [
        ("model", C.c_int32), ("nu", C.c_int32), ("T", C.c_int32), ("eq_mode", C.c_int32),
        ("chain", C.c_int32), ("pad_", C.c_int32),
        ("dt", C.c_double), ("wb", C.c_double),
        ("R", C.c_double * 4), ("Rd", C.c_double * 4),
        ("Q", C.c_double * 3), ("q_growth", C.c_double), ("Qf", C.c_double * 3),
        ("umax", C.c_double * 4), ("dumax", C.c_double * 4), ("first_step", C.c_double * 4),
        ("steer_fixed", C.c_double),
        ("cold_start", C.c_double * 4),
    ]


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "mpc_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        assert _lib.orc_variant_size() == C.sizeof(OrcVariant)
    return _lib


def set_polish(on: bool):
    """Active-set polish after the interior-point solve (default on; the timed CPU baseline
    turns it off so the baseline is not charged for the checker's extra accuracy)."""
    lib().orc_set_polish(C.c_int(1 if on else 0))


def _pad4(t, fill=0.0):
    t = tuple(t)
    return t + (fill,) * (4 - len(t))


def to_c(v: O.Variant) -> OrcVariant:
    s = OrcVariant()
    s.model, s.nu, s.T, s.eq_mode, s.chain = v.model, v.nu, v.T, v.eq_mode, v.chain
    s.dt, s.wb = v.dt, v.wb
    s.R[:] = _pad4(v.R)
    s.Rd[:] = _pad4(v.Rd)
    s.Q[:] = v.Q
    s.q_growth = v.q_growth
    s.Qf[:] = v.Qf
    s.umax[:] = _pad4(v.umax, O.INF)
    s.dumax[:] = _pad4(v.dumax, O.INF)
    s.first_step[:] = _pad4(v.first_step, O.INF)
    s.steer_fixed = v.steer_fixed
    s.cold_start[:] = _pad4(v.cold_start)
    return s


def _p(a, ct=C.c_double):
    return None if a is None else a.ctypes.data_as(C.POINTER(ct))


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def iterative_batch(v: O.Variant, x0, xref, uprev=None, has_prev=None, nthreads: int = 1):
    """Returns dict(u, x, obj, status, iters, kkt)."""
    x0 = _f64(x0)
    xref = _f64(xref)
    B = x0.shape[0]
    T, NU = v.T, v.nu
    assert x0.shape == (B, 3) and xref.shape == (B, 3, T + 1)
    if uprev is not None:
        uprev = _f64(uprev)
        assert uprev.shape == (B, NU, T)
    if has_prev is not None:
        has_prev = np.ascontiguousarray(has_prev, dtype=np.uint8)
    out = _alloc(B, NU, T)
    cv = to_c(v)
    rc = lib().orc_iterative_batch(C.byref(cv), C.c_int64(B), _p(x0), _p(xref), _p(uprev),
                                   _p(has_prev, C.c_uint8), _p(out["u"]), _p(out["x"]), _p(out["obj"]),
                                   _p(out["status"], C.c_int32), _p(out["iters"], C.c_int32),
                                   _p(out["kkt"]), C.c_int(nthreads))
    assert rc == 0
    return out


def linear_mpc_batch(v: O.Variant, xref, xbar, x0, uref, nthreads: int = 1):
    x0 = _f64(x0)
    xref = _f64(xref)
    xbar = _f64(xbar)
    uref = _f64(uref)
    B = x0.shape[0]
    T, NU = v.T, v.nu
    assert xref.shape == (B, 3, T + 1) and xbar.shape == (B, 3, T + 1) and uref.shape == (B, NU, T)
    out = _alloc(B, NU, T)
    cv = to_c(v)
    rc = lib().orc_linear_mpc_batch(C.byref(cv), C.c_int64(B), _p(xref), _p(xbar), _p(x0), _p(uref),
                                    _p(out["u"]), _p(out["x"]), _p(out["obj"]),
                                    _p(out["status"], C.c_int32), _p(out["iters"], C.c_int32),
                                    _p(out["kkt"]), C.c_int(nthreads))
    assert rc == 0
    return out


def _alloc(B, NU, T):
    return dict(u=np.empty((B, NU, T)), x=np.empty((B, 3, T + 1)), obj=np.empty(B),
                status=np.empty(B, dtype=np.int32), iters=np.empty(B, dtype=np.int32), kkt=np.empty(B))


def rollout_batch(v: O.Variant, x0, u):
    x0 = _f64(x0)
    u = _f64(u)
    B = x0.shape[0]
    xbar = np.empty((B, 3, v.T + 1))
    cv = to_c(v)
    assert lib().orc_rollout_batch(C.byref(cv), C.c_int64(B), _p(x0), _p(u), _p(xbar)) == 0
    return xbar


def linearize_batch(v: O.Variant, theta, u):
    theta = _f64(theta)
    u = _f64(u)
    B = theta.shape[0]
    assert u.shape == (B, v.nu)
    A = np.empty((B, 3, 3))
    Bm = np.empty((B, 3, v.nu))
    Cc = np.empty((B, 3))
    cv = to_c(v)
    assert lib().orc_linearize_batch(C.byref(cv), C.c_int64(B), _p(theta), _p(u), _p(A), _p(Bm), _p(Cc)) == 0
    return A, Bm, Cc


def nearest_ref_batch(mode: int, T: int, dt: float, param: float, state, course_id, pind,
                      cx, cy, cyaw, clen, n_search: int = 10, sign: bool = True):
    """mode 0 = calc_ref_trajectory (param = dl), 1 = calc_ref_trajectory2 (param = n_search_ind).
    Returns (ind, mind, xref)."""
    state = _f64(state)
    B = state.shape[0]
    cx, cy, cyaw = _f64(cx), _f64(cy), _f64(cyaw)
    K, Lmax = cx.shape
    course_id = np.ascontiguousarray(course_id, dtype=np.int32)
    pind = np.ascontiguousarray(pind, dtype=np.int32)
    clen = np.ascontiguousarray(clen, dtype=np.int32)
    ind = np.empty(B, dtype=np.int32)
    mind = np.empty(B)
    xref = np.empty((B, 3, T + 1))
    rc = lib().orc_nearest_ref_batch(C.c_int(mode), C.c_int(T), C.c_double(dt), C.c_double(param),
                                     C.c_int(n_search), C.c_int(1 if sign else 0), C.c_int64(B),
                                     _p(state), _p(course_id, C.c_int32), _p(pind, C.c_int32),
                                     _p(cx), _p(cy), _p(cyaw), _p(clen, C.c_int32), C.c_int64(Lmax),
                                     _p(ind, C.c_int32), _p(mind), _p(xref))
    assert rc == 0
    return ind, mind, xref
