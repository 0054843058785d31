"""Host-side preprocessing of recorded trajectories: CSV node lists, interpolation between
nodes, de-duplication, and splitting a course by motion mode.  These produce the course
(cx, cy, cyaw[, cmode]) the controller then follows; they run once per course, on the host,
vectorised over the nodes.

Node row (mpc.py:234-235, mpc_dsr.py:711-712):
    x | y | yaw | vx | vy | w | front angle | front speed | rear angle | rear speed | time_stamp
plus, in mpc_sim's files, a 12th string column with the motion mode (mpc_sim.py:570-574).
"""
from __future__ import annotations

import csv
from typing import List, Sequence, Tuple

import numpy as np

ARC_MODES = ("ackermann", "diff")


def read_nodes_csv(file_name: str, with_mode: bool = False) -> List[list]:
    """retriveReferenceFromCSV (mpc_dsr.py:696-713, mpc.py:221-237) / retriveTrajectoryFromCSV
    (mpc_sim.py:559-575): the header row is skipped; every column is a float, except column 11
    (the mode) when `with_mode`."""
    nodes = []
    with open(file_name, newline="") as f:
        rows = csv.reader(f)
        next(rows, None)
        for row in rows:
            if with_mode:
                nodes.append([e if i == 11 else float(e) for i, e in enumerate(row)])
            else:
                nodes.append([float(e) for e in row])
    return nodes


def _numeric(nodes: Sequence[Sequence]) -> np.ndarray:
    return np.array([[float(v) for v in row[:6]] for row in nodes], dtype=np.float64).reshape(-1, 6)


def _arc_points(frm: np.ndarray, to: np.ndarray, k: int):
    """`k` points per segment on the circular arc about the instantaneous centre of rotation
    of the segment's end node, the yaw sweeping linearly from the start node's to the end
    node's (mpc_dsr.py:731-745).  frm, to: [S, 6] rows x, y, yaw, vx, vy, w."""
    i = np.arange(k)[None, :]
    icr0 = (-to[:, 4] / to[:, 5])[:, None]
    icr1 = (to[:, 3] / to[:, 5])[:, None]
    y0 = frm[:, 2][:, None]
    yaw = y0 + (to[:, 2][:, None] - y0) / k * i
    dc = np.cos(y0) - np.cos(yaw)
    dsn = np.sin(y0) - np.sin(yaw)
    x = dc * icr0 - dsn * icr1 + frm[:, 0][:, None]
    y = dsn * icr0 + dc * icr1 + frm[:, 1][:, None]
    return x, y, yaw


def _line_points(frm: np.ndarray, to: np.ndarray, k: int):
    """`k` points per segment on the straight line between the nodes, yaw held at the start
    node's (crab motion, mpc_dsr.py:756-764)."""
    i = np.arange(k)[None, :]
    x = frm[:, 0][:, None] + (to[:, 0] - frm[:, 0])[:, None] / k * i
    y = frm[:, 1][:, None] + (to[:, 1] - frm[:, 1])[:, None] / k * i
    yaw = np.repeat(frm[:, 2][:, None], k, axis=1)
    return x, y, yaw


def interpolate_nodes(nodes, interpolate_num: int = 5, mode: str = "ackermann"):
    """interpolateReference (mpc_dsr.py:715-766, mpc_sim.py:577-611): lists pts_x, pts_y, pts_yaw.
    Arc modes skip the segments whose end node has w == 0; an unknown mode returns None, as
    the reference's function falls off its end."""
    a = _numeric(nodes)
    frm, to = a[:-1], a[1:]
    if mode in ARC_MODES:
        keep = to[:, 5] != 0
        x, y, yaw = _arc_points(frm[keep], to[keep], interpolate_num)
    elif mode == "crab":
        x, y, yaw = _line_points(frm, to, interpolate_num)
    else:
        return None
    return list(x.reshape(-1)), list(y.reshape(-1)), list(yaw.reshape(-1))


def interpolate_nodes_by_mode(nodes, interpolate_num: int = 5):
    """interpolateReference2 (mpc_sim.py:613-653): every segment interpolated by the mode in
    column 11 of its end node; returns pts_x, pts_y, pts_yaw, pts_mode.  Segments of an arc
    mode with w == 0, and segments of an unknown mode, contribute nothing."""
    a = _numeric(nodes)
    modes = [row[11] for row in nodes[1:]]
    frm, to = a[:-1], a[1:]
    S, k = len(modes), interpolate_num
    arc = np.array([m in ARC_MODES for m in modes], dtype=bool) & (to[:, 5] != 0)
    crab = np.array([m == "crab" for m in modes], dtype=bool)
    x = np.empty((S, k)); y = np.empty((S, k)); yaw = np.empty((S, k))
    if arc.any():
        x[arc], y[arc], yaw[arc] = _arc_points(frm[arc], to[arc], k)
    if crab.any():
        x[crab], y[crab], yaw[crab] = _line_points(frm[crab], to[crab], k)
    keep = arc | crab
    pm = [m for m, kp in zip(modes, keep) for _ in range(k) if kp]
    return list(x[keep].reshape(-1)), list(y[keep].reshape(-1)), list(yaw[keep].reshape(-1)), pm


def remove_repeated_points(cx, cy, cyaw, epsilon: float = 0.00001):
    """removeRepeatedPoints (mpc_dsr.py:768-786, mpc_sim.py:655-671): a point is dropped when
    its squared distance to the last point KEPT is below epsilon (so the scan is sequential)."""
    nx: list = []
    ny: list = []
    nyaw: list = []
    for x, y, yaw in zip(cx, cy, cyaw):
        if nx and (x - nx[-1]) ** 2 + (y - ny[-1]) ** 2 < epsilon:
            continue
        nx.append(x); ny.append(y); nyaw.append(yaw)
    return nx, ny, nyaw


def split_by_motion_mode(cmode) -> List[List[int]]:
    """splitTrajectoryWithMotionModes (mpc_sim.py:673-690): [from, to] index pairs of the runs
    of equal mode.  As in the reference a run is only closed by a change of mode, so the last
    run of the course is not reported."""
    m = list(cmode)
    groups: List[List[int]] = []
    start = None
    for i in range(len(m) - 1):
        if start is None:
            start = i
        if m[start] != m[i + 1]:
            groups.append([start, i])
            start = None
    return groups
