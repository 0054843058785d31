"""Call-compatible stand-in for the reference's mpc_dsr.py (dual steer/drive "general
bicycle", NX=3, NU=4, control rows v_f, delta_f, v_r, delta_r).

Same constants, names, argument order and return conventions; hot-path functions run
on the GPU through libmpcx.so.  Module constants are read at call time.
"""
from __future__ import annotations

import math

import numpy as np

from . import _scalar, trajectory_io, variants

NX = 3
NU = 4
T = 5

R = np.diag([0.01, 0.01, 0.01, 0.01])
Rd = np.diag([0.01, 0.01, 0.01, 0.01])   # unused by the reference as well
Q = np.diag([1.0, 1.0, 0.5])
Qf = Q
GOAL_DIS = 1.5
STOP_SPEED = 0.5 / 3.6
MAX_TIME = 500.0

MAX_ITER = 1
DU_TH = 0.1

N_IND_SEARCH = 10

DT = 0.2

LENGTH = 4.5
WIDTH = 2.0

MAX_V_SPEED = 0.47
DIFF_V_SPEED = 0.5
MAX_STEER = np.deg2rad(90)
DIFF_STEER = np.deg2rad(45)

show_animation = False

# Which QP iterative_linear_mpc_control solves.  The reference hard-wires the "diff" one
# (mpc_dsr.py:193; the ackermann and crab calls are commented out at :192,:194).
LMPC_MODE = "diff"


def _variant(mode):
    common = dict(T=int(T), dt=float(DT), R=np.diag(R), Q=np.diag(Q), Qf=np.diag(Qf))
    dv, ds = DIFF_V_SPEED * DT, DIFF_STEER * DT
    inf = float("inf")
    if mode == "diff":       # mpc_dsr.py:337-395
        return variants.get("dsr_diff", umax=[MAX_V_SPEED, inf, MAX_V_SPEED, inf], dumax=[dv, inf, dv, inf],
                            steer_fixed=float(MAX_STEER), **common)
    name = {"ackermann": "dsr_ackermann", "crab": "dsr_crab"}[mode]   # :201-267, :269-335
    return variants.get(name, umax=[MAX_V_SPEED, MAX_STEER, MAX_V_SPEED, MAX_STEER], dumax=[dv, ds, dv, ds],
                        first_step=[inf, ds, inf, ds], **common)


class State:
    """vehicle state (mpc_dsr.py:54-66)"""

    def __init__(self, x=0.0, y=0.0, yaw=0.0, vx=0.0, vy=0.0, w=0.0):
        self.x, self.y, self.yaw, self.vx, self.vy, self.w = x, y, yaw, vx, vy, w
        self.predelta = None


def pi_2_pi(angle):
    while angle > math.pi:
        angle = angle - 2.0 * math.pi
    while angle < -math.pi:
        angle = angle + 2.0 * math.pi
    return angle


def calc_nearest_index(state, cx, cy, cyaw, pind):
    ind, mind, _ = _scalar.nearest_ref_one(1, int(T), float(DT), 2, state, cx, cy, cyaw, pind, N_IND_SEARCH)
    return ind, mind


def update_state(state, vx, vy, w):
    """Plant step of the simulator (host-side scalar helper, mpc_dsr.py:114-123)."""
    c, s = math.cos(state.yaw), math.sin(state.yaw)
    state.x = state.x + vx * c * DT - vy * s * DT
    state.y = state.y + vx * s * DT + vy * c * DT
    state.yaw = state.yaw + w * DT
    state.vx, state.vy, state.w = vx, vy, w
    return state


def predict_motion(x0, vf, vr, sf, sr, xref):
    u = [list(vf)[:T], list(sf)[:T], list(vr)[:T], list(sr)[:T]]
    return _scalar.rollout_one(_variant(LMPC_MODE), x0, u)


def transfrom_gbm_command(vf, vr, sf, sr, wheel_base=1):
    """Wheel commands -> body twist (vx, vy, w): (H^T H)^-1 H^T in closed form
    (host-side scalar helper, mpc_dsr.py:141-167)."""
    v1x, v1y = math.cos(sf) * vf, math.sin(sf) * vf
    v2x, v2y = math.cos(sr) * vr, math.sin(sr) * vr
    return 0.5 * (v1x + v2x), 0.5 * (v1y + v2y), (v1y - v2y) / wheel_base


def _ret7(ok, u, x):
    if not ok:
        return None, None, None, None, None, None, None
    # (ovf, ovr, osf, osr, ox, oy, oyaw) = rows (0, 2, 1, 3) of u (mpc_dsr.py:252-255)
    return u[0].copy(), u[2].copy(), u[1].copy(), u[3].copy(), x[0].copy(), x[1].copy(), x[2].copy()


def iterative_linear_mpc_control(xref, x0, ovf, ovr, osf, osr):
    ucols = None
    if not (ovf is None or ovr is None or osf is None or osr is None):
        ucols = [list(ovf), list(osf), list(ovr), list(osr)]
    ok, u, x, _ = _scalar.solve_one(_variant(LMPC_MODE), xref, x0, ucols)
    return _ret7(ok, u, x)


def linear_mpc_control_ackermann(xref, xbar, x0, uref):
    ok, u, x, _ = _scalar.linear_one(_variant("ackermann"), xref, xbar, x0, uref)
    return _ret7(ok, u, x)


def linear_mpc_control_crab(xref, xbar, x0, uref):
    ok, u, x, _ = _scalar.linear_one(_variant("crab"), xref, xbar, x0, uref)
    return _ret7(ok, u, x)


def linear_mpc_control_diff(xref, xbar, x0, uref):
    ok, u, x, _ = _scalar.linear_one(_variant("diff"), xref, xbar, x0, uref)
    return _ret7(ok, u, x)


def get_linear_model_matrix(theta, v_f, delta_f, v_r, delta_r, wheel_base=1):
    v = variants.copy(_variant("ackermann"), wb=float(wheel_base))
    return _scalar.linearize_one(v, theta, [v_f, delta_f, v_r, delta_r])


def calc_ref_trajectory(state, cx, cy, cyaw, dl, pind):
    ind, _, xref = _scalar.nearest_ref_one(0, int(T), float(DT), float(dl), state, cx, cy, cyaw, pind, N_IND_SEARCH)
    return xref, ind


def calc_ref_trajectory2(state, cx, cy, cyaw, n_search_ind, pind):
    ind, _, xref = _scalar.nearest_ref_one(1, int(T), float(DT), int(n_search_ind), state, cx, cy, cyaw, pind,
                                           N_IND_SEARCH)
    return xref, ind


def check_goal(state, goal, tind, nind):
    d = math.hypot(state.x - goal[0], state.y - goal[1])
    isgoal = d <= GOAL_DIS
    if abs(tind - nind) >= 5:
        isgoal = False
    return bool(isgoal and abs(state.vx) <= STOP_SPEED)


def do_simulation(cx, cy, cyaw, dl, initial_state, max_ticks=None):
    """Closed loop of mpc_dsr.py:504-589 without the plotting, run on the device-resident loop
    (the fixed-steer QP iterative_linear_mpc_control calls, mpc_dsr.py:193); `cyaw` is smoothed in
    place and `initial_state` ends as the final state, as in the reference."""
    state = initial_state
    cyaw = smooth_yaw(cyaw)
    r = _scalar.closed_loop_one(_variant(LMPC_MODE), cx, cy, cyaw, state, GOAL_DIS, STOP_SPEED, DT, MAX_TIME,
                                max_ticks=max_ticks, n_search=N_IND_SEARCH)
    if r["goal"]:
        print("Goal")
    return r["t"], r["x"], r["y"], r["yaw"], r["vx"], r["w"]


def smooth_yaw(yaw):
    for i in range(len(yaw) - 1):
        dyaw = yaw[i + 1] - yaw[i]
        while dyaw >= math.pi / 2.0:
            yaw[i + 1] -= math.pi * 2.0
            dyaw = yaw[i + 1] - yaw[i]
        while dyaw <= -math.pi / 2.0:
            yaw[i + 1] += math.pi * 2.0
            dyaw = yaw[i + 1] - yaw[i]
    return yaw


def makeEightShapeTrajectory(n, size):
    x, y, yaw = [], [], []
    for i in range(n):
        x.append(0.8 * math.sin(2 * math.pi / 60 * i) * size)
        y.append(math.sin(1 * math.pi / 60 * i) * size)
        dx = 0.8 * math.cos(2 * math.pi / 60 * i) * 2 * math.pi / 60
        dy = math.cos(1 * math.pi / 60 * i) * 1 * math.pi / 60
        yaw.append(math.atan2(dy, dx))
    return x, y, yaw


# --- recorded-trajectory preprocessing (mpc_dsr.py:696-786) ---------------------------------

def retriveReferenceFromCSV(file_name):
    return trajectory_io.read_nodes_csv(file_name)


def interpolateReference(node_lists, interpolate_num=5, mode='ackermann'):
    return trajectory_io.interpolate_nodes(node_lists, interpolate_num, mode)


def removeRepeatedPoints(cx, cy, cyaw, epsilon=0.00001):
    return trajectory_io.remove_repeated_points(cx, cy, cyaw, epsilon)


# --- scenario drivers (mpc_dsr.py:623-695; no plotting, the recorded file is an argument) --------

def _run_course(cx, cy, cyaw, max_ticks=None):
    cx, cy, cyaw = list(cx), list(cy), list(cyaw)
    initial_state = State(x=cx[0], y=cy[0], yaw=cyaw[0])
    cx.pop(0); cy.pop(0); cyaw.pop(0)
    return do_simulation(cx, cy, cyaw, 0.5, initial_state, max_ticks=max_ticks)


def _run_recorded(file_name, interpolate_num, mode, max_ticks):
    cx, cy, cyaw = interpolateReference(retriveReferenceFromCSV(file_name), interpolate_num, mode)
    return _run_course(*removeRepeatedPoints(cx, cy, cyaw), max_ticks=max_ticks)


def main(max_ticks=None):
    """Figure-eight course."""
    return _run_course(*makeEightShapeTrajectory(400, 10), max_ticks=max_ticks)


def main2(file_name='reference.csv', max_ticks=None):
    """Recorded trajectory, ICR-arc interpolation."""
    return _run_recorded(file_name, 3, 'ackermann', max_ticks)


def main3(file_name='reference_crab.csv', max_ticks=None):
    """Recorded crab trajectory, straight-line interpolation."""
    return _run_recorded(file_name, 3, 'crab', max_ticks)


def main4(file_name='reference_diff.csv', max_ticks=None):
    """Recorded trajectory for the fixed-steer mode (interpolated as the reference does: 'crab')."""
    return _run_recorded(file_name, 2, 'crab', max_ticks)
