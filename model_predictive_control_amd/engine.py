"""Batched operators on device tensors.  torch is only used for device memory and the
current HIP stream; every computation happens in libmpcx.so."""
from __future__ import annotations

import math

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import CVariant


def _device() -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.MpcxError("no GPU visible: the mpcx engine runs on MI355X only (no CPU fallback)")
    return torch.device("cuda", torch.cuda.current_device())


def _dev(a, dtype=torch.float64) -> torch.Tensor:
    """numpy / list / tensor -> contiguous device tensor of `dtype`."""
    if isinstance(a, torch.Tensor):
        t = a
    else:
        t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(device=_device(), dtype=dtype, non_blocking=False).contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _outputs(B: int, NU: int, T: int, dev) -> Dict[str, torch.Tensor]:
    return dict(u=torch.empty((B, NU, T), dtype=torch.float64, device=dev),
                x=torch.empty((B, 3, T + 1), dtype=torch.float64, device=dev),
                obj=torch.empty((B,), dtype=torch.float64, device=dev),
                status=torch.empty((B,), dtype=torch.int32, device=dev),
                iters=torch.empty((B,), dtype=torch.int32, device=dev))


def solve_batch(v: CVariant, x0, xref, uprev=None, has_prev=None, out=None, noalloc: bool = False) -> Dict[str, torch.Tensor]:
    """iterative_linear_mpc_control for B instances: x0 [B,3], xref [B,3,T+1],
    uprev [B,NU,T] or None (cold start), has_prev [B] uint8 or None.  `noalloc` (with `out`
    given and device tensors in) goes through the entry point that allocates nothing."""
    lib = _lib.load()
    x0 = _dev(x0)
    xref = _dev(xref)
    B, T, NU = x0.shape[0], v.T, v.nu
    if tuple(x0.shape) != (B, 3) or tuple(xref.shape) != (B, 3, T + 1):
        raise ValueError(f"x0 must be [B,3] and xref [B,3,{T + 1}]")
    if uprev is not None:
        uprev = _dev(uprev)
        if tuple(uprev.shape) != (B, NU, T):
            raise ValueError(f"uprev must be [B,{NU},{T}]")
    if has_prev is not None:
        has_prev = _dev(has_prev, torch.uint8)
        if tuple(has_prev.shape) != (B,):
            raise ValueError("has_prev must be [B]")
    if out is None:
        out = _outputs(B, NU, T, x0.device)
    entry = lib.mpcx_solve_batch_noalloc if noalloc else lib.mpcx_solve_batch
    _lib.check(entry(C.byref(v), B, _ptr(x0), _ptr(xref), _ptr(uprev), _ptr(has_prev),
                     _ptr(out["u"]), _ptr(out["x"]), _ptr(out["obj"]), _ptr(out["status"]),
                     _ptr(out["iters"]), _stream()))
    return out


def solve_mixed(variant_list, variant_id, x0, xref, uprev=None, has_prev=None) -> Dict[str, torch.Tensor]:
    """A batch whose instances use different variants of one shape (same NU and T): instances are
    grouped by variant so that every wavefront is homogeneous, each group is one launch, and the
    results come back in the caller's order.  variant_id [B] indexes variant_list."""
    x0, xref = _dev(x0), _dev(xref)
    vid = _dev(variant_id, torch.int64)
    B = x0.shape[0]
    nu, T = variant_list[0].nu, variant_list[0].T
    if any(v.nu != nu or v.T != T for v in variant_list):
        raise ValueError("solve_mixed needs variants of one shape (same NU and T)")
    uprev = None if uprev is None else _dev(uprev)
    has_prev = None if has_prev is None else _dev(has_prev, torch.uint8)
    order = torch.argsort(vid, stable=True)
    counts = torch.bincount(vid, minlength=len(variant_list)).cpu().tolist()
    out = _outputs(B, nu, T, x0.device)
    start = 0
    for k, n in enumerate(counts):
        if n == 0:
            continue
        idx = order[start:start + n]
        part = solve_batch(variant_list[k], x0[idx], xref[idx], None if uprev is None else uprev[idx],
                           None if has_prev is None else has_prev[idx])
        for key in out:
            out[key][idx] = part[key]
        start += n
    return out


def linear_mpc_batch(v: CVariant, xref, xbar, x0, uref, out=None) -> Dict[str, torch.Tensor]:
    """linear_mpc_control* for B instances at the linearisation point (xbar, uref)."""
    lib = _lib.load()
    x0 = _dev(x0)
    xref = _dev(xref)
    xbar = _dev(xbar)
    uref = _dev(uref)
    B, T, NU = x0.shape[0], v.T, v.nu
    if tuple(x0.shape) != (B, 3) or tuple(xref.shape) != (B, 3, T + 1) or tuple(xbar.shape) != (B, 3, T + 1) \
            or tuple(uref.shape) != (B, NU, T):
        raise ValueError("shape mismatch")
    if out is None:
        out = _outputs(B, NU, T, x0.device)
    _lib.check(lib.mpcx_linear_mpc_batch(C.byref(v), B, _ptr(xref), _ptr(xbar), _ptr(x0), _ptr(uref),
                                         _ptr(out["u"]), _ptr(out["x"]), _ptr(out["obj"]), _ptr(out["status"]),
                                         _ptr(out["iters"]), _stream()))
    return out


def rollout_batch(v: CVariant, x0, u) -> torch.Tensor:
    """predict_motion for B instances: xbar [B,3,T+1]."""
    lib = _lib.load()
    x0 = _dev(x0)
    u = _dev(u)
    B = x0.shape[0]
    if tuple(u.shape) != (B, v.nu, v.T):
        raise ValueError(f"u must be [B,{v.nu},{v.T}]")
    xbar = torch.empty((B, 3, v.T + 1), dtype=torch.float64, device=x0.device)
    _lib.check(lib.mpcx_rollout_batch(C.byref(v), B, _ptr(x0), _ptr(u), _ptr(xbar), _stream()))
    return xbar


def linearize_batch(v: CVariant, theta, u):
    """get_linear_model_matrix for B operating points: theta [B], u [B,NU] -> A, B, C."""
    lib = _lib.load()
    theta = _dev(theta)
    u = _dev(u)
    B = theta.shape[0]
    if tuple(u.shape) != (B, v.nu):
        raise ValueError(f"u must be [B,{v.nu}]")
    A = torch.empty((B, 3, 3), dtype=torch.float64, device=theta.device)
    Bm = torch.empty((B, 3, v.nu), dtype=torch.float64, device=theta.device)
    Cc = torch.empty((B, 3), dtype=torch.float64, device=theta.device)
    _lib.check(lib.mpcx_linearize_batch(C.byref(v), B, _ptr(theta), _ptr(u), _ptr(A), _ptr(Bm), _ptr(Cc), _stream()))
    return A, Bm, Cc


class CourseTable:
    """Padded SoA course tables on the device: cx, cy, cyaw [K, Lmax], clen [K]."""

    def __init__(self, courses):
        """courses: list of (cx, cy, cyaw) sequences."""
        K = len(courses)
        L = max(len(c[0]) for c in courses)
        tx = np.zeros((K, L))
        ty = np.zeros((K, L))
        tw = np.zeros((K, L))
        ln = np.zeros(K, dtype=np.int32)
        for k, (cx, cy, cyaw) in enumerate(courses):
            n = len(cx)
            tx[k, :n], ty[k, :n], tw[k, :n], ln[k] = cx, cy, cyaw, n
        self.K, self.Lmax = K, L
        self.cx, self.cy, self.cyaw = _dev(tx), _dev(ty), _dev(tw)
        self.clen = _dev(ln, torch.int32)
        self.clen_host = ln
        self.ck = None

    @classmethod
    def from_waypoints(cls, waypoints, ds: float = 0.1, smooth: bool = True, Lmax: Optional[int] = None):
        """Builds the tables on the device from waypoint lists [(x_k, y_k), ...]:
        cubic_spline_planner.calc_spline_course per course, then smooth_yaw
        (mpc_diff_drive.py:404-417) when `smooth`.  A course whose samples do not fit Lmax, or
        with repeated waypoints, raises."""
        lib = _lib.load()
        K = len(waypoints)
        N = max(len(w[0]) for w in waypoints)
        wx = np.zeros((K, N)); wy = np.zeros((K, N)); nw = np.zeros(K, dtype=np.int32)
        longest = 0.0
        for k, (x, y) in enumerate(waypoints):
            n = len(x)
            wx[k, :n], wy[k, :n], nw[k] = x, y, n
            longest = max(longest, float(np.hypot(np.diff(x), np.diff(y)).sum()))
        if Lmax is None:
            Lmax = int(math.ceil(longest / ds)) + 2
        self = cls.__new__(cls)
        dev = _device()
        self.K, self.Lmax = K, int(Lmax)
        self.cx = torch.zeros((K, Lmax), dtype=torch.float64, device=dev)
        self.cy = torch.zeros_like(self.cx)
        self.cyaw = torch.zeros_like(self.cx)
        self.ck = torch.zeros_like(self.cx)
        self.clen = torch.zeros(K, dtype=torch.int32, device=dev)
        d_wx, d_wy, d_nw = _dev(wx), _dev(wy), _dev(nw, torch.int32)
        _lib.check(lib.mpcx_build_courses_batch(K, N, _ptr(d_wx), _ptr(d_wy), _ptr(d_nw), float(ds), int(Lmax),
                                                1 if smooth else 0, _ptr(self.cx), _ptr(self.cy), _ptr(self.cyaw),
                                                _ptr(self.ck), _ptr(self.clen), _stream()))
        self.clen_host = self.clen.cpu().numpy()
        if (self.clen_host < 0).any():
            bad = int(np.nonzero(self.clen_host < 0)[0][0])
            raise _lib.MpcxError(f"course {bad}: repeated or non-finite waypoints, or more than Lmax={Lmax} samples")
        return self


def nearest_ref_batch(mode: int, T: int, dt: float, param: float, state, course_id, pind, table: CourseTable,
                      n_search: int = 10, sign: bool = True, want_xref: bool = True):
    """calc_nearest_index + calc_ref_trajectory (mode 0, param = dl) or calc_ref_trajectory2
    (mode 1, param = n_search_ind).  Returns (ind [B] int32, mind [B], xref [B,3,T+1])."""
    lib = _lib.load()
    state = _dev(state)
    B = state.shape[0]
    if tuple(state.shape) != (B, 4):
        raise ValueError("state must be [B,4] = x, y, yaw, vx")
    course_id = None if course_id is None else _dev(course_id, torch.int32)
    pind = _dev(pind, torch.int32)
    ind = torch.empty((B,), dtype=torch.int32, device=state.device)
    mind = torch.empty((B,), dtype=torch.float64, device=state.device)
    xref = torch.empty((B, 3, T + 1), dtype=torch.float64, device=state.device) if want_xref else None
    _lib.check(lib.mpcx_nearest_ref_batch(mode, T, dt, param, n_search, 1 if sign else 0, B, _ptr(state),
                                          _ptr(course_id), _ptr(pind), _ptr(table.cx), _ptr(table.cy),
                                          _ptr(table.cyaw), _ptr(table.clen), table.K, table.Lmax,
                                          _ptr(ind), _ptr(mind), _ptr(xref), None, _stream()))
    return ind, mind, xref


class ClosedLoop:
    """Device-resident closed loop over a batch of vehicles: per tick, reference window
    (calc_ref_trajectory2) -> iterative_linear_mpc_control -> update_state, warm-start carry,
    check_goal, with no host round trip (do_simulation of mpc_diff_drive.py:319-383,
    mpc_dsr.py:504-571, MPC.doSimulation of mpc_sim.py:201-247, minus the plotting).

    state [B,4] = x, y, yaw, vx;  course_id [B] indexes `table`.  After `run(n)`:
    `state`, `twist` (vy, w), `done` (0 running / 1 goal / 2 window lost), `ticks`,
    `target_ind`, the last solve in `out`, and `log` [n_ticks+1, B, 4] if asked for."""

    def __init__(self, v: CVariant, table: CourseTable, state, course_id=None, goal_dis: float = 1.5,
                 stop_speed: float = 0.5 / 3.6, n_search_ind: int = 2, n_search: int = 10, log_ticks: int = 0,
                 pind0=None, log_twist: bool = False):
        self.lib = _lib.load()
        self.v, self.table = v, table
        self.state = _dev(state).clone()
        B = self.B = self.state.shape[0]
        if tuple(self.state.shape) != (B, 4):
            raise ValueError("state must be [B,4] = x, y, yaw, vx")
        dev = self.state.device
        self.course_id = None if course_id is None else _dev(course_id, torch.int32)
        self.goal_dis, self.stop_speed = float(goal_dis), float(stop_speed)
        self.n_search_ind, self.n_search = int(n_search_ind), int(n_search)
        self.twist = torch.zeros((B, 2), dtype=torch.float64, device=dev)
        self.has_prev = torch.zeros(B, dtype=torch.uint8, device=dev)
        self.done = torch.zeros(B, dtype=torch.uint8, device=dev)
        self.ticks = torch.zeros(B, dtype=torch.int32, device=dev)
        # the first nearest-index search starts at the course start, as the reference's does,
        # unless the caller knows where along its course each vehicle is
        self.target_ind = torch.zeros(B, dtype=torch.int32, device=dev) if pind0 is None else _dev(pind0, torch.int32).clone()
        self.xref = torch.empty((B, 3, v.T + 1), dtype=torch.float64, device=dev)
        self.x0 = torch.empty((B, 3), dtype=torch.float64, device=dev)
        # the solve reads the previous optimum while it writes the new one: two buffers
        self.out = _outputs(B, v.nu, v.T, dev)
        self.uprev = torch.zeros((B, v.nu, v.T), dtype=torch.float64, device=dev)
        self.log = torch.empty((log_ticks + 1, B, 4), dtype=torch.float64, device=dev) if log_ticks else None
        # (vy, w) applied per tick, kept beside the state log when asked for (row 0: at rest)
        self.twist_log = torch.zeros((log_ticks + 1, B, 2), dtype=torch.float64, device=dev) if log_ticks and log_twist else None
        self.n_done_ticks = 0
        self._solve = self.lib.mpcx_solve_batch
        self._initial_alignment()

    def _initial_alignment(self):
        """Initial yaw compensation against the course's first yaw and the first nearest index
        searched from the course start (mpc_diff_drive.py:336-350)."""
        cid = self.course_id.long() if self.course_id is not None else torch.zeros(self.B, dtype=torch.long, device=self.state.device)
        # a course id outside the table is an empty window to the kernels; here it must not index
        ok = (cid >= 0) & (cid < self.table.K)
        d = torch.where(ok, self.state[:, 2] - self.table.cyaw[cid.clamp(0, self.table.K - 1), 0], torch.zeros_like(self.state[:, 2]))
        two_pi = 2.0 * math.pi
        self.state[:, 2] = torch.where(d >= math.pi, self.state[:, 2] - two_pi,
                                       torch.where(d <= -math.pi, self.state[:, 2] + two_pi, self.state[:, 2]))
        t = self.table
        _lib.check(self.lib.mpcx_nearest_ref_batch(1, self.v.T, self.v.dt, float(self.n_search_ind), self.n_search, 1,
                                                   self.B, _ptr(self.state), _ptr(self.course_id), _ptr(self.target_ind),
                                                   _ptr(t.cx), _ptr(t.cy), _ptr(t.cyaw), _ptr(t.clen), t.K, t.Lmax,
                                                   _ptr(self.target_ind), None, None, None, _stream()))
        if self.log is not None:
            self.log[0].copy_(self.state)

    def prepare(self):
        """First third of a tick: the inputs of the next solve (`x0`, `xref`; `uprev` and
        `has_prev` are already in place)."""
        t, v = self.table, self.v
        _lib.check(self.lib.mpcx_nearest_ref_batch(1, v.T, v.dt, float(self.n_search_ind), self.n_search, 1, self.B,
                                                   _ptr(self.state), _ptr(self.course_id), _ptr(self.target_ind),
                                                   _ptr(t.cx), _ptr(t.cy), _ptr(t.cyaw), _ptr(t.clen), t.K, t.Lmax,
                                                   _ptr(self.target_ind), None, _ptr(self.xref), _ptr(self.x0), _stream()))

    def tick(self):
        t, v, lib, st = self.table, self.v, self.lib, _stream()
        self.prepare()
        o = self.out
        _lib.check(self._solve(C.byref(v), self.B, _ptr(self.x0), _ptr(self.xref), _ptr(self.uprev),
                               _ptr(self.has_prev), _ptr(o["u"]), _ptr(o["x"]), _ptr(o["obj"]),
                               _ptr(o["status"]), _ptr(o["iters"]), st))
        k = self.n_done_ticks + 1
        log = self.log[k] if self.log is not None and k < self.log.shape[0] else None
        _lib.check(lib.mpcx_advance_batch(C.byref(v), self.B, self.goal_dis, self.stop_speed, _ptr(o["u"]),
                                          _ptr(o["status"]), _ptr(self.target_ind), _ptr(self.course_id),
                                          _ptr(t.cx), _ptr(t.cy), _ptr(t.clen), t.K, t.Lmax, _ptr(self.state),
                                          _ptr(self.twist), _ptr(self.has_prev), _ptr(self.done), _ptr(self.ticks),
                                          _ptr(log), st))
        if self.twist_log is not None and k < self.twist_log.shape[0]:
            self.twist_log[k].copy_(self.twist)
        o["u"], self.uprev = self.uprev, o["u"]   # this optimum warm-starts the next tick
        self.n_done_ticks = k

    def capture(self):
        """Records two consecutive ticks (one per warm-start buffer) as a HIP graph; `run`
        then replays it: 8 launches become one graph launch.  The register/LDS-resident solve
        kernels allocate nothing; a loop whose solve streams through a workspace switches, from
        here on, to the entry point that uses the library's fixed workspace (64 MiB per GPU,
        shared by everything on that GPU that takes this route: one such loop at a time) and is
        refused if it does not fit.  Logging is off in a captured loop.  Capturing itself
        advances the loop by two (warm-up) ticks."""
        if self.log is not None:
            raise _lib.MpcxError("a logging loop cannot be captured: the log slot changes every tick")
        need = self.lib.mpcx_workspace_bytes(C.byref(self.v), self.B)
        if need != 0:
            have = self.lib.mpcx_static_workspace_bytes()      # also the per-device set-up of that entry point
            if need > have:
                raise _lib.MpcxError(f"this loop's solve needs a workspace of {need} bytes: the growing one is allocated and "
                                     f"waited for on an event, which a stream capture does not allow, and the fixed one holds {have}")
            self._solve = self.lib.mpcx_solve_batch_noalloc
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.tick(); self.tick()      # warm-up on the capture stream
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        done = self.n_done_ticks
        with torch.cuda.graph(g):
            self.tick(); self.tick()      # an even number of ticks leaves the two u buffers in place
        self.n_done_ticks = done          # recorded, not run
        self._graph = g
        return self

    def run(self, n_ticks: int):
        g = getattr(self, "_graph", None)
        while g is not None and n_ticks >= 2:
            g.replay()
            self.n_done_ticks += 2
            n_ticks -= 2
        for _ in range(n_ticks):
            self.tick()
        return self

    @property
    def last_u(self):
        return self.uprev


class SegmentedLoop:
    """Device-resident closed loop over courses that change motion mode along the way
    (MPC.doSimulationWithAllMotionModes, mpc_sim.py:254-265, over the segments
    splitTrajectoryWithMotionModes cuts, :674-690): every vehicle drives its list of segments one
    after the other, each segment a course of its own (a row of `table`) in its own mode, exactly
    as the reference runs one doSimulation per segment: cold start, initial yaw compensation and
    first nearest index from the segment's start, until the goal test of that segment passes or
    `max_ticks_per_segment` ticks are spent; then the next segment starts from the state reached.

    variants[m]: the solver variant of mode m (same model, horizon and dt for all);
    seg_mode [K]: mode index of every table row;  seg_start [B], seg_count [B]: each vehicle's rows
    (consecutive).

    A tick never waits for the device (no host round trip, every launch on the current stream):
    reference windows for the whole fleet; the vehicles sorted by the mode they are in (a stable
    device sort; vehicles at rest last) and their solve inputs gathered in that order; one solve
    launch for the whole sorted fleet, every mode's run with its own variant and the runs' bounds
    staying in device memory (mpcx_solve_modes_batch); controls and statuses scattered back; state
    advance and goal test;
    segment hand-over (mpcx_segment_step_batch) and the first nearest index of the vehicles that
    entered a segment.  That needs solves the T = 5 lane-pair kernel takes (every mpc_sim mode
    does); other variant sets fall back to one compacted launch per mode with the host reading
    the index lists (`self.device_resident` tells which)."""

    def __init__(self, variants, table: CourseTable, seg_mode, seg_start, seg_count, state, max_ticks_per_segment: int,
                 goal_dis: float = 1.5, stop_speed: float = 0.2, n_search_ind: int = 2, n_search: int = 10,
                 log_ticks: int = 0):
        self.lib = _lib.load()
        self.variants, self.table = list(variants), table
        v0 = self.variants[0]
        if any(v.T != v0.T or v.nu != v0.nu or v.dt != v0.dt or v.model != v0.model for v in self.variants):
            raise ValueError("the modes of a segmented loop must share model, horizon and dt")
        self.state = _dev(state).clone()
        B = self.B = self.state.shape[0]
        dev = self.state.device
        M = len(self.variants)
        self.seg_mode = _dev(seg_mode, torch.int64)
        self.seg_start = _dev(seg_start, torch.int32)
        self.seg_count = _dev(seg_count, torch.int32)
        self.cap = int(max_ticks_per_segment)
        self.goal_dis, self.stop_speed = float(goal_dis), float(stop_speed)
        self.n_search_ind, self.n_search = int(n_search_ind), int(n_search)
        self.seg = torch.zeros(B, dtype=torch.int32, device=dev)
        self.seg_ticks = torch.zeros(B, dtype=torch.int32, device=dev)
        self.finished = (self.seg_count <= 0).to(torch.uint8)
        self.course_id = (self.seg_start + self.seg).to(torch.int32)
        self.twist = torch.zeros((B, 2), dtype=torch.float64, device=dev)
        self.has_prev = torch.zeros(B, dtype=torch.uint8, device=dev)
        self.done = self.finished.clone()
        self.ticks = torch.zeros(B, dtype=torch.int32, device=dev)
        self.target_ind = torch.zeros(B, dtype=torch.int32, device=dev)
        self.xref = torch.empty((B, 3, v0.T + 1), dtype=torch.float64, device=dev)
        self.x0 = torch.empty((B, 3), dtype=torch.float64, device=dev)
        self.out = _outputs(B, v0.nu, v0.T, dev)
        self.out["status"].fill_(_lib.ST_FAILED)
        self.uprev = torch.zeros((B, v0.nu, v0.T), dtype=torch.float64, device=dev)
        self.log = torch.empty((log_ticks + 1, B, 4), dtype=torch.float64, device=dev) if log_ticks else None
        self.n_done_ticks = 0
        # buffers of the device-resident tick
        self._sorted = _outputs(B, v0.nu, v0.T, dev)
        self._pind_next = torch.zeros(B, dtype=torch.int32, device=dev)
        self._enter = torch.zeros(B, dtype=torch.uint8, device=dev)
        self._ind_tmp = torch.zeros(B, dtype=torch.int32, device=dev)
        self._was_active = torch.zeros(B, dtype=torch.uint8, device=dev)
        self._order = torch.zeros(B, dtype=torch.int32, device=dev)
        self._bounds = torch.zeros(M + 2, dtype=torch.int32, device=dev)
        self._seg_mode32 = self.seg_mode.to(torch.int32)
        self._gathered = dict(x0=torch.empty_like(self.x0), xref=torch.empty_like(self.xref), up=torch.empty_like(self.uprev),
                              hp=torch.zeros(B, dtype=torch.uint8, device=dev))
        n_scratch = self.lib.mpcx_mode_sort_scratch_ints(B, M)
        self._scratch = torch.zeros(max(1, n_scratch), dtype=torch.int32, device=dev) if n_scratch >= 0 else None
        self.device_resident = self._scratch is not None and self._range_solve_available()
        self._enter_segment(torch.nonzero(self.finished == 0).flatten())
        if self.log is not None:
            self.log[0].copy_(self.state)

    def _range_solve_available(self) -> bool:
        """Whether the modes' solves can go as one launch over the mode-sorted fleet (tried here with
        empty runs).  Sets up the variant table the kernel reads: a host array and its device copy."""
        M = len(self.variants)
        self._vtab_host = (_lib.CVariant * M)(*self.variants)
        raw = np.frombuffer(bytes(self._vtab_host), dtype=np.uint8).copy()
        self._vtab_dev = torch.from_numpy(raw).to(self.state.device)
        o = self._sorted
        empty = torch.zeros(M + 1, dtype=torch.int32, device=self.state.device)
        rc = self._solve_modes(empty, self.x0, self.xref, self.uprev, self.has_prev, o)
        if rc == _lib.E_UNSUPPORTED:
            return False
        _lib.check(rc)
        return True

    def _solve_modes(self, bounds, x0, xref, up, hp, o):
        return self.lib.mpcx_solve_modes_batch(C.cast(self._vtab_host, C.c_void_p), len(self.variants), _ptr(self._vtab_dev),
                                               _ptr(bounds), self.B, _ptr(x0), _ptr(xref), _ptr(up), _ptr(hp), _ptr(o["u"]),
                                               _ptr(o["x"]), _ptr(o["obj"]), _ptr(o["status"]), _ptr(o["iters"]), _stream())

    def _enter_segment(self, idx):
        """What doSimulation does before its loop (mpc_sim.py:204-222) for the vehicles `idx`
        entering their current segment: yaw compensation against the segment's first yaw, first
        nearest index searched from the segment's start, cold start.  (Set-up, and the ticks of the
        fallback path; the device-resident tick does this in mpcx_segment_step_batch.)"""
        if idx.numel() == 0:
            return
        t = self.table
        cid = self.course_id[idx].long()
        yaw = self.state[idx, 2]
        ok = (cid >= 0) & (cid < t.K)      # rows outside the table: an empty window, never an index
        d = torch.where(ok, yaw - t.cyaw[cid.clamp(0, t.K - 1), 0], torch.zeros_like(yaw))
        two_pi = 2.0 * math.pi
        self.state[idx, 2] = torch.where(d >= math.pi, yaw - two_pi, torch.where(d <= -math.pi, yaw + two_pi, yaw))
        st = self.state[idx].contiguous()
        ind, _, _ = nearest_ref_batch(1, self.variants[0].T, self.variants[0].dt, float(self.n_search_ind), st,
                                      self.course_id[idx].contiguous(), torch.zeros(idx.numel(), dtype=torch.int32, device=st.device),
                                      t, n_search=self.n_search, want_xref=False)
        self.target_ind[idx] = ind
        self.has_prev[idx] = 0
        self.done[idx] = 0
        self.seg_ticks[idx] = 0

    def _nearest(self, pind, ind_out, xref, x0=None):
        t, v0 = self.table, self.variants[0]
        _lib.check(self.lib.mpcx_nearest_ref_batch(1, v0.T, v0.dt, float(self.n_search_ind), self.n_search, 1, self.B,
                                                   _ptr(self.state), _ptr(self.course_id), _ptr(pind),
                                                   _ptr(t.cx), _ptr(t.cy), _ptr(t.cyaw), _ptr(t.clen), t.K, t.Lmax,
                                                   _ptr(ind_out), None, _ptr(xref), _ptr(x0), _stream()))

    def _advance(self, k):
        t, v0, o = self.table, self.variants[0], self.out
        log = self.log[k] if self.log is not None and k < self.log.shape[0] else None
        _lib.check(self.lib.mpcx_advance_batch(C.byref(v0), self.B, self.goal_dis, self.stop_speed, _ptr(o["u"]),
                                               _ptr(o["status"]), _ptr(self.target_ind), _ptr(self.course_id),
                                               _ptr(t.cx), _ptr(t.cy), _ptr(t.clen), t.K, t.Lmax, _ptr(self.state),
                                               _ptr(self.twist), _ptr(self.has_prev), _ptr(self.done), _ptr(self.ticks),
                                               _ptr(log), _stream()))

    def tick(self):
        if not self.device_resident:
            return self._tick_with_host_lists()
        t, lib, st, M = self.table, self.lib, _stream(), len(self.variants)
        self._nearest(self.target_ind, self.target_ind, self.xref, self.x0)   # x0 = [state.x, state.y, state.yaw]
        v0, so, o, g = self.variants[0], self._sorted, self.out, self._gathered
        was_active = self._was_active
        # the fleet sorted by mode (vehicles at rest last), the solve's inputs gathered in that order
        _lib.check(lib.mpcx_mode_sort_batch(self.B, M, v0.nu, v0.T, _ptr(self.finished), _ptr(self.done), _ptr(self.course_id),
                                            _ptr(self._seg_mode32), t.K, _ptr(self.x0), _ptr(self.xref), _ptr(self.uprev),
                                            _ptr(self.has_prev), _ptr(was_active), _ptr(self._order), _ptr(self._bounds),
                                            _ptr(g["x0"]), _ptr(g["xref"]), _ptr(g["up"]), _ptr(g["hp"]), _ptr(self._scratch), st))
        # one launch for all modes: every run of the sorted fleet gets wavefronts of its own and its
        # mode's variant (MPC.iterativeLMPC's dispatch on `mode`, mpc_sim.py:301-306)
        _lib.check(self._solve_modes(self._bounds, g["x0"], g["xref"], g["up"], g["hp"], so))
        # rows past the last run (vehicles at rest) carry stale values; the advance ignores them (done != 0)
        _lib.check(lib.mpcx_mode_unsort_batch(self.B, v0.nu, v0.T, _ptr(self._order), _ptr(so["u"]), _ptr(so["status"]),
                                              _ptr(o["u"]), _ptr(o["status"]), st))
        k = self.n_done_ticks + 1
        self._advance(k)
        o["u"], self.uprev = self.uprev, o["u"]
        self.n_done_ticks = k
        _lib.check(lib.mpcx_segment_step_batch(self.B, self.cap, _ptr(was_active), _ptr(self.seg_start), _ptr(self.seg_count),
                                               _ptr(self.seg), _ptr(self.seg_ticks), _ptr(self.finished), _ptr(self.done),
                                               _ptr(self.has_prev), _ptr(self.course_id), _ptr(self.state), _ptr(t.cyaw),
                                               t.K, t.Lmax, _ptr(self.target_ind), _ptr(self._pind_next), _ptr(self._enter), st))
        # first nearest index of the vehicles that entered a segment, searched from its start
        self._nearest(self._pind_next, self._ind_tmp, None)
        self.target_ind.copy_(torch.where(self._enter != 0, self._ind_tmp, self.target_ind))
        if self.log is not None and k < self.log.shape[0]:
            self.log[k].copy_(self.state)     # after a possible yaw compensation of the next segment

    def _tick_with_host_lists(self):
        t = self.table
        self._nearest(self.target_ind, self.target_ind, self.xref, self.x0)   # x0 = [state.x, state.y, state.yaw]
        o = self.out
        mode = self.seg_mode[self.course_id.long().clamp(0, t.K - 1)]
        active = (self.finished == 0) & (self.done == 0)
        for m, v in enumerate(self.variants):
            idx = torch.nonzero(active & (mode == m)).flatten()
            if idx.numel() == 0:
                continue
            part = solve_batch(v, self.x0[idx], self.xref[idx], self.uprev[idx], self.has_prev[idx])
            o["u"][idx] = part["u"]
            o["status"][idx] = part["status"]
        k = self.n_done_ticks + 1
        self._advance(k)
        o["u"], self.uprev = self.uprev, o["u"]
        self.n_done_ticks = k
        # segment bookkeeping: goal reached or the segment's ticks spent -> next segment; a lost
        # reference window (where the reference raises) ends the vehicle's run
        self.seg_ticks += active.to(torch.int32)
        lost = active & (self.done == 2)
        move = active & ~lost & ((self.done == 1) | (self.seg_ticks >= self.cap))
        self.finished |= lost.to(torch.uint8)
        if bool(move.any().item()):
            self.seg += move.to(torch.int32)
            ended = move & (self.seg >= self.seg_count)
            self.finished |= ended.to(torch.uint8)
            go = move & ~ended
            self.course_id = torch.where(go, (self.seg_start + self.seg).to(torch.int32), self.course_id)
            self.done = torch.where(ended, torch.ones_like(self.done), self.done)
            self._enter_segment(torch.nonzero(go).flatten())
        if self.log is not None and k < self.log.shape[0]:
            self.log[k].copy_(self.state)     # after a possible yaw compensation of the next segment

    def run(self, n_ticks: int, until_finished: bool = False):
        """n_ticks ticks; a tick is a no-op for vehicles that have finished.  `until_finished`
        looks at the fleet's flags before every tick (one host round trip each) and stops early."""
        for _ in range(n_ticks):
            if until_finished and bool((self.finished != 0).all().item()):
                break
            self.tick()
        return self


class HostPipeline:
    """Solves for callers whose data lives in host memory: batches go host -> device -> solve ->
    host through `depth` slots, each with pinned host buffers, device buffers and a HIP stream
    of its own, so the upload of one batch, the solve of another and the download of a third
    overlap (the link carries about 650 bytes per instance each way combined; the solve itself
    is an order of magnitude shorter than either copy).

        pipe = HostPipeline(v, B)
        slot = pipe.acquire()                 # waits until the slot's previous batch is done
        slot.x0[:] = ...; slot.xref[:] = ...  # numpy views of the pinned input buffers
        slot.uprev[:] = ...; slot.has_prev[:] = ...
        pipe.submit(slot, warm=True)          # enqueues copy-in, solve, copy-out; returns at once
        ...
        res = pipe.result(slot)               # waits for that batch; numpy views of pinned outputs

    Results stay valid until the slot is acquired again."""

    class Slot:
        pass

    def __init__(self, v: CVariant, B: int, depth: int = 3):
        dev = _device()
        self.v, self.B, self.depth = v, int(B), int(depth)
        T, NU = v.T, v.nu
        shapes = dict(x0=((B, 3), torch.float64), xref=((B, 3, T + 1), torch.float64),
                      uprev=((B, NU, T), torch.float64), has_prev=((B,), torch.uint8))
        self._slots = []
        for _ in range(self.depth):
            s = HostPipeline.Slot()
            s.stream = torch.cuda.Stream(device=dev)
            s.done = torch.cuda.Event()
            s.busy = False
            s.h_in = {k: torch.empty(shp, dtype=dt).pin_memory() for k, (shp, dt) in shapes.items()}
            s.d_in = {k: torch.empty(shp, dtype=dt, device=dev) for k, (shp, dt) in shapes.items()}
            s.d_out = _outputs(B, NU, T, dev)
            s.h_out = {k: torch.empty_like(t, device="cpu").pin_memory() for k, t in s.d_out.items()}
            for k, t in s.h_in.items():
                setattr(s, k, t.numpy())
            s.out = {k: t.numpy() for k, t in s.h_out.items()}
            self._slots.append(s)
        self._next = 0

    def acquire(self):
        s = self._slots[self._next]
        self._next = (self._next + 1) % self.depth
        if s.busy:
            s.done.synchronize()
            s.busy = False
        return s

    def submit(self, s, warm: bool = True):
        with torch.cuda.stream(s.stream):
            names = ("x0", "xref", "uprev", "has_prev") if warm else ("x0", "xref")
            for k in names:
                s.d_in[k].copy_(s.h_in[k], non_blocking=True)
            solve_batch(self.v, s.d_in["x0"], s.d_in["xref"], s.d_in["uprev"] if warm else None,
                        s.d_in["has_prev"] if warm else None, out=s.d_out)
            for k in s.d_out:
                s.h_out[k].copy_(s.d_out[k], non_blocking=True)
            s.done.record(s.stream)
        s.busy = True
        return s

    def result(self, s):
        if s.busy:
            s.done.synchronize()
            s.busy = False
        return s.out
