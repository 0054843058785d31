"""Call-compatible stand-in for the reference's mpc_sim.py: GeneralBicycleModel and the
class-based MPC with a motion-mode argument ('ackermann' | 'diff' | 'crab').

iterativeLMPC keeps the reference's behaviour of running `horizon_` chained QPs per
call: its mode dispatch sits inside the loop that copies the warm start into uref
(mpc_sim.py:295-308), so solve i sees uref[:, :i+1] only, column i taken from solve
i-1.  The chain runs inside one kernel launch.
"""
from __future__ import annotations

import math

import numpy as np

from . import _scalar, trajectory_io, variants


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


class State:

    def __init__(self, x=0.0, y=0.0, yaw=0.0, vx=0.0, vy=0.0, w=0.0):
        self.x, self.y, self.yaw, self.vx, self.vy, self.w = x, y, yaw, vx, vy, w


class GeneralBicycleModel:
    """mpc_sim.py:87-163"""

    def __init__(self, wheel_base=1, nx=3, nu=4) -> None:
        self.wheel_base_ = wheel_base
        self.nx_ = nx
        self.nu_ = nu

    def updateState(self, state, vx, vy, w, dt=0.2):
        """Plant step (host-side scalar helper)."""
        c, s = math.cos(state.yaw), math.sin(state.yaw)
        state.x = state.x + vx * c * dt - vy * s * dt
        state.y = state.y + vx * s * dt + vy * c * dt
        state.yaw = state.yaw + w * dt
        state.vx, state.vy, state.w = vx, vy, w
        return state

    def transformWheelCommandToRobotCommand(self, vf, vr, sf, sr):
        """Wheel speed & steer -> (vx, vy, w) (host-side scalar helper)."""
        v1x, v1y = math.cos(sf) * vf, math.sin(sf) * vf
        v2x, v2y = math.cos(sr) * vr, math.sin(sr) * vr
        return 0.5 * (v1x + v2x), 0.5 * (v1y + v2y), (v1y - v2y) / self.wheel_base_

    def getRobotModelMatrice(self, theta, v_f, delta_f, v_r, delta_r, dt=0.2):
        """x(k+1) = A x(k) + B u(k) + C at the operating point."""
        v = variants.get("sim_ackermann", dt=float(dt), wb=float(self.wheel_base_))
        return _scalar.linearize_one(v, theta, [v_f, delta_f, v_r, delta_r])


MODES = ("ackermann", "diff", "crab")      # mode index of engine.SegmentedLoop


class MPC:
    """mpc_sim.py:165-553"""

    def __init__(self, horizon=5, nx=3, nu=4, xy_tolerance=1.5, stop_speed=0.2, show_animation=False) -> None:
        self.horizon_ = horizon
        self.nx_ = nx
        self.nu_ = nu
        self.gbm_ = GeneralBicycleModel(wheel_base=1, nx=nx, nu=nu)
        self.xy_tolerance_ = xy_tolerance
        self.stop_speed_ = stop_speed
        self.ackermann_steer_inc_rate_ = np.deg2rad(45)
        self.ackermann_speed_inc_rate_ = 0.2
        self.ackermann_max_speed_ = 0.47
        self.ackermann_max_steer_ = np.deg2rad(45)
        self.differential_speed_inc_rate_ = 0.2
        self.differential_fixed_steer_ = np.deg2rad(90)
        self.differential_max_speed_ = 0.47
        self.crab_steer_inc_rate_ = np.deg2rad(45)
        self.crab_speed_inc_rate_ = 0.2
        self.crab_max_speed_ = 0.5
        self.crab_max_steer_ = np.deg2rad(45)
        self.R_ = np.diag([0.01, 0.01, 0.01, 0.01])
        self.Rd_ = np.diag([0.01, 0.01, 0.01, 0.01])  # unused by the reference as well
        self.Q_ = np.diag([1.0, 1.0, 0.5])
        self.Qf_ = np.diag([1.0, 1.0, 0.5])
        self.show_animation_ = show_animation

    def _variant(self, mode, dt=0.2, chain=1):
        common = dict(T=int(self.horizon_), dt=float(dt), wb=float(self.gbm_.wheel_base_), R=np.diag(self.R_),
                      Q=np.diag(self.Q_), Qf=np.diag(self.Qf_), chain=int(chain))
        inf = float("inf")
        if mode == "ackermann":   # mpc_sim.py:312-380
            dv, ds = self.ackermann_speed_inc_rate_ * dt, self.ackermann_steer_inc_rate_ * dt
            vm, sm = self.ackermann_max_speed_, self.ackermann_max_steer_
            return variants.get("sim_ackermann", umax=[vm, sm, vm, sm], dumax=[dv, ds, dv, ds],
                                first_step=[inf, ds, inf, ds], **common)
        if mode == "crab":        # mpc_sim.py:442-508
            dv, ds = self.crab_speed_inc_rate_ * dt, self.crab_steer_inc_rate_ * dt
            vm, sm = self.crab_max_speed_, self.crab_max_steer_
            return variants.get("sim_crab", umax=[vm, sm, vm, sm], dumax=[dv, ds, dv, ds],
                                first_step=[inf, ds, inf, ds], **common)
        if mode == "diff":        # mpc_sim.py:382-440
            dv, vm = self.differential_speed_inc_rate_ * dt, self.differential_max_speed_
            return variants.get("sim_diff", umax=[vm, inf, vm, inf], dumax=[dv, inf, dv, inf],
                                steer_fixed=float(self.differential_fixed_steer_), **common)
        return None

    def doSimulation(self, cx, cy, cyaw, initial_state, mode='ackermann', max_time=500, dt=0.2, max_ticks=None):
        """Closed loop of mpc_sim.py:201-252 without the plotting, run on the device-resident loop
        (per tick: reference window, the chained QPs of iterativeLMPC in one launch, updateState,
        checkGoal; no host round trip).  `cyaw` is smoothed in place and `initial_state` ends as
        the final state, as in the reference."""
        state = initial_state
        cyaw = smooth_yaw(cyaw)
        v = self._variant(mode, chain=self.horizon_)     # `dt` only advances the clock (mpc_sim.py:235); model and QP keep 0.2
        if v is None:                       # unknown mode: the reference's iterativeLMPC returns None and raises
            raise TypeError("cannot unpack non-iterable NoneType object")
        r = _scalar.closed_loop_one(v, cx, cy, cyaw, state, self.xy_tolerance_, self.stop_speed_, dt, max_time,
                                    max_ticks=max_ticks, n_search=10)
        if r["goal"]:
            print("Goal Reached.")
        return r["t"], r["x"], r["y"], r["yaw"], r["vx"], r["vy"], r["w"], state

    def doSimulationWithAllMotionModes(self, idx_group, cx, cy, cyaw, cmode, initial_state, max_time=500, dt=0.2,
                                       max_ticks=None):
        """mpc_sim.py:254-265: one closed loop per course segment, each in its own mode and each
        on the device-resident loop.  (The reference returns nothing; the per-segment results are
        returned here.  Like the reference, the segments run with doSimulation's default max_time
        and dt; `max_ticks` bounds every segment.)  A fleet of vehicles on such courses runs as
        one batch through engine.SegmentedLoop."""
        current_state = initial_state
        results = []
        for idx in idx_group:
            lx = list(cx[idx[0]: idx[1] + 1])
            ly = list(cy[idx[0]: idx[1] + 1])
            lyaw = list(cyaw[idx[0]: idx[1] + 1])
            res = self.doSimulation(lx, ly, lyaw, current_state, cmode[idx[0]], max_ticks=max_ticks)
            results.append(res)
            current_state = res[-1]
        return results

    def segment_variants(self):
        """The three modes' solver variants in the order engine.SegmentedLoop indexes them."""
        return [self._variant(m, chain=self.horizon_) for m in MODES]

    def predictMotion(self, x0, vf, vr, sf, sr, xref):
        T = self.horizon_
        u = [list(vf)[:T], list(sf)[:T], list(vr)[:T], list(sr)[:T]]
        return _scalar.rollout_one(self._variant("ackermann"), x0, u)

    @staticmethod
    def _ret7(ok, u, x):
        if not ok:
            return None, None, None, None, None, None, None
        return u[0].copy(), u[2].copy(), u[1].copy(), u[3].copy(), x[0].copy(), x[1].copy(), x[2].copy()

    def iterativeLMPC(self, xref, x0, ovf, ovr, osf, osr, mode='ackermann', max_iter=1):
        v = self._variant(mode, chain=self.horizon_)
        if v is None:
            print("Mode not defined. ")
            return ovf, ovr, osf, osr, None, None, None
        ucols = None
        if not (ovf is None or ovr is None or osf is None or osr is None):
            ucols = [list(ovf), list(osf), list(ovr), list(osr)]
        ok, u, x, _ = _scalar.solve_one(v, xref, x0, ucols)
        return self._ret7(ok, u, x)

    def doLMPC_Ackermann(self, xref, xbar, x0, uref, dt=0.2):
        return self._ret7(*_scalar.linear_one(self._variant("ackermann", dt), xref, xbar, x0, uref)[:3])

    def doLMPC_Differential(self, xref, xbar, x0, uref, dt=0.2):
        return self._ret7(*_scalar.linear_one(self._variant("diff", dt), xref, xbar, x0, uref)[:3])

    def doLMPC_Crab(self, xref, xbar, x0, uref, dt=0.2):
        return self._ret7(*_scalar.linear_one(self._variant("crab", dt), xref, xbar, x0, uref)[:3])

    def getNearestIndex(self, state, cx, cy, cyaw, pind, n_ind_search=10):
        ind, mind, _ = _scalar.nearest_ref_one(1, int(self.horizon_), 0.2, 2, state, cx, cy, cyaw, pind,
                                               int(n_ind_search), sign=False)
        return ind, mind

    def getReferenceTrajectory(self, state, cx, cy, cyaw, n_search_ind, pind):
        ind, _, xref = _scalar.nearest_ref_one(1, int(self.horizon_), 0.2, int(n_search_ind), state, cx, cy, cyaw,
                                               pind, 10, sign=False)
        return xref, ind

    def checkGoal(self, state, goal, tind, nind):
        d = math.hypot(state.x - goal[0], state.y - goal[1])
        reached = d <= self.xy_tolerance_
        if abs(tind - nind) >= 5:
            reached = False
        return bool(reached and abs(state.vx) <= self.stop_speed_)


class TrajectoryGenerator:
    """mpc_sim.py:554-703: CSV ingest, interpolation between recorded nodes, de-duplication,
    splitting by motion mode (all in trajectory_io.py) and the synthetic figure-eight."""

    def retriveTrajectoryFromCSV(self, file_name):
        return trajectory_io.read_nodes_csv(file_name, with_mode=True)

    def interpolateReference(self, node_lists, interpolate_num=5, mode='ackermann'):
        return trajectory_io.interpolate_nodes(node_lists, interpolate_num, mode)

    def interpolateReference2(self, node_lists, interpolate_num=5):
        return trajectory_io.interpolate_nodes_by_mode(node_lists, interpolate_num)

    def removeRepeatedPoints(self, cx, cy, cyaw, epsilon=0.00001):
        return trajectory_io.remove_repeated_points(cx, cy, cyaw, epsilon)

    def splitTrajectoryWithMotionModes(self, cmode):
        return trajectory_io.split_by_motion_mode(cmode)

    def makeEightShapeTrajectory(self, size=10, n=121):
        x, y, yaw = [], [], []
        for i in range(n):
            x.append(0.8 * math.sin(2 * math.pi / 60 * i) * size)
            y.append(math.sin(1 * math.pi / 60 * i) * size)
            dx = 0.8 * math.cos(2 * math.pi / 60 * i) * 2 * math.pi / 60
            dy = math.cos(1 * math.pi / 60 * i) * 1 * math.pi / 60
            yaw.append(math.atan2(dy, dx))
        return x, y, yaw


# --- scenario drivers (mpc_sim.py:706-784; no plotting, the recorded file is an argument) --------

def _run_course(cx, cy, cyaw, mode, max_ticks):
    cx, cy, cyaw = list(cx), list(cy), list(cyaw)
    initial_state = State(x=cx[0], y=cy[0], yaw=cyaw[0])
    cx.pop(0); cy.pop(0); cyaw.pop(0)
    return MPC().doSimulation(cx, cy, cyaw, initial_state, mode, max_ticks=max_ticks)


def _run_recorded(file_name, interpolate_num, mode, max_ticks):
    tg = TrajectoryGenerator()
    cx, cy, cyaw = tg.interpolateReference(tg.retriveTrajectoryFromCSV(file_name), interpolate_num, mode)
    return _run_course(*tg.removeRepeatedPoints(cx, cy, cyaw), mode=mode, max_ticks=max_ticks)


def main1(max_ticks=None):
    """Figure-eight course, ackermann mode."""
    return _run_course(*TrajectoryGenerator().makeEightShapeTrajectory(), mode='ackermann', max_ticks=max_ticks)


def main2(file_name='reference.csv', max_ticks=None):
    return _run_recorded(file_name, 3, 'ackermann', max_ticks)


def main3(file_name='reference_crab.csv', max_ticks=None):
    return _run_recorded(file_name, 3, 'crab', max_ticks)


def main4(file_name='reference_diff.csv', max_ticks=None):
    return _run_recorded(file_name, 2, 'diff', max_ticks)


def main5(file_name='output.csv', max_ticks=None):
    """Recorded trajectory with a motion mode per node: one simulation per mode segment."""
    tg = TrajectoryGenerator()
    cx, cy, cyaw, cmode = tg.interpolateReference2(tg.retriveTrajectoryFromCSV(file_name), 3)
    idx_group = tg.splitTrajectoryWithMotionModes(cmode)
    initial_state = State(x=cx[0], y=cy[0], yaw=cyaw[0])
    return MPC().doSimulationWithAllMotionModes(idx_group, cx, cy, cyaw, cmode, initial_state, max_ticks=max_ticks)
