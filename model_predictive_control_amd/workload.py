"""Deterministic synthetic workload for the benchmark and the parity tests: spline
courses plus instance states scattered around them (numpy, float64, seeded)."""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from . import cubic_spline_planner as csp


def make_courses(n_courses: int = 1024, seed: int = 1001, ds: float = 0.1) -> List[Tuple[np.ndarray, ...]]:
    """Course k: 8 waypoints, x_j = cumulative sum of U(3,6), y_j = U(-6,6), through
    calc_spline_course(ds) and smooth_yaw."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n_courses):
        wx = np.cumsum(rng.uniform(3.0, 6.0, 8))
        wy = rng.uniform(-6.0, 6.0, 8)
        rx, ry, ryaw, _, _ = csp.calc_spline_course(wx, wy, ds)
        ryaw = csp.smooth_yaw(list(ryaw))
        out.append((np.asarray(rx), np.asarray(ry), np.asarray(ryaw)))
    return out


def make_instances(courses, B: int, seed: int = 2002, vmax: float = 0.47) -> Dict[str, np.ndarray]:
    """course_id ~ U{0..K-1}, pind ~ U{0..len-12}, position = course point pind + U{0..3}
    plus N(0, 0.15^2), yaw = course yaw + N(0, 0.1^2), vx ~ U(0, vmax)."""
    rng = np.random.default_rng(seed)
    K = len(courses)
    lens = np.array([len(c[0]) for c in courses])
    cid = rng.integers(0, K, B)
    pind = (rng.random(B) * np.maximum(lens[cid] - 12, 1)).astype(np.int64)
    j = np.minimum(pind + rng.integers(0, 4, B), lens[cid] - 1)
    L = lens.max()
    tx = np.zeros((K, L)); ty = np.zeros((K, L)); tw = np.zeros((K, L))
    for k, (cx, cy, cyaw) in enumerate(courses):
        tx[k, :len(cx)], ty[k, :len(cx)], tw[k, :len(cx)] = cx, cy, cyaw
    state = np.empty((B, 4))
    state[:, 0] = tx[cid, j] + rng.normal(0.0, 0.15, B)
    state[:, 1] = ty[cid, j] + rng.normal(0.0, 0.15, B)
    state[:, 2] = tw[cid, j] + rng.normal(0.0, 0.1, B)
    state[:, 3] = rng.uniform(0.0, vmax, B)
    return dict(state=state, course_id=cid.astype(np.int32), pind=pind.astype(np.int32))


def reference_window(courses, inst, T: int, n_ahead: int = 2) -> np.ndarray:
    """Host-side xref [B,3,T+1] of calc_ref_trajectory2 semantics (nearest of the next 10
    course points, never behind pind; that point, then the following one repeated) for
    building solver inputs without a device round trip."""
    B = inst["state"].shape[0]
    xref = np.empty((B, 3, T + 1))
    for b in range(B):
        cx, cy, cyaw = courses[inst["course_id"][b]]
        p = int(inst["pind"][b])
        hi = min(p + 10, len(cx))
        d = (inst["state"][b, 0] - cx[p:hi]) ** 2 + (inst["state"][b, 1] - cy[p:hi]) ** 2
        ind = max(p + int(np.argmin(d)), p)
        nxt = min(ind + n_ahead - 1, len(cx) - 1)
        xref[b, :, 0] = (cx[ind], cy[ind], cyaw[ind])
        xref[b, :, 1:] = np.array([cx[nxt], cy[nxt], cyaw[nxt]])[:, None]
    return xref


def reference_window_next_points(courses, inst, T: int, stride: int = 3) -> np.ndarray:
    """Host-side xref [B,3,T+1] the way mpc.py builds it: the next T+1 points of the (pruned)
    recorded trajectory ahead of the robot (mpc.py:292-296), here every `stride`-th point of
    the synthetic course from the instance's index on, the last point repeated at the end."""
    B = inst["state"].shape[0]
    xref = np.empty((B, 3, T + 1))
    steps = stride * np.arange(T + 1)
    for b in range(B):
        cx, cy, cyaw = courses[inst["course_id"][b]]
        idx = np.minimum(int(inst["pind"][b]) + steps, len(cx) - 1)
        xref[b] = np.stack([cx[idx], cy[idx], cyaw[idx]])
    return xref
