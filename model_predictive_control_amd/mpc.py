"""Call-compatible stand-in for the reference's mpc.py: ModelPredictiveControl, the
ROS-side controller (HL=10, DT=0.1, dual steer, ackermann equalities, steer-rate
bounds and an input-difference cost).  controlLaw / predictMotion /
getRobotModelMatrice run on the GPU through libmpcx.so.
"""
from __future__ import annotations

import csv
import math

import numpy as np

from . import _scalar, variants


class State:
    """Pose, body twist and wheel commands of the dual-steer platform (mpc.py:13-27)."""

    def __init__(self, x, y, yaw, vx, vy, w, front_steer, front_speed, rear_steer, rear_speed) -> None:
        self.x_, self.y_, self.yaw_ = x, y, yaw
        self.vx_, self.vy_, self.w_ = vx, vy, w
        self.front_steer_, self.front_speed_ = front_steer, front_speed
        self.rear_steer_, self.rear_speed_ = rear_steer, rear_speed


class ModelPredictiveControl:

    def __init__(self) -> None:
        self.ITERATION_TIMES_ = 1
        self.HL_ = 10
        self.NX_ = 3
        self.NU_ = 4
        self.R_ = np.diag([0.001, 0.001, 0.001, 0.001])
        self.R_diff_ = np.diag([0.001, 0.001, 0.001, 0.001])
        self.Q_ = np.diag([0.1, 0.1, 0.1])
        self.Q_final_ = np.diag([0.1, 0.1, 0])
        self.LOOKAHEAD_DIST_ = 1
        self.MAX_TRAVEL_SPEED_ = 0.5
        self.MAX_STEER_SPEED_ = 3
        self.MAX_STEER_ = 3.14
        self.DT_ = 0.1
        self.isModelParameterRetrived_ = False
        self.wheel_base_ = 0.0
        self.kinematical_matrix_ = None
        self.isReferenceRetrived_ = False
        self.reference_ = None
        self.trim_dist_ = 1
        self.pts_x_ = []
        self.pts_y_ = []
        self.pts_yaw_ = []
        self.control_input_ = None
        self.predict_state_ = []

    def _variant(self):
        """mpc.py:117-187 with the current attributes."""
        inf = float("inf")
        ds = self.MAX_STEER_SPEED_ * self.DT_
        vm, sm = self.MAX_TRAVEL_SPEED_, self.MAX_STEER_
        return variants.get("mpc_py", T=int(self.HL_), dt=float(self.DT_), wb=float(self.wheel_base_),
                            R=np.diag(self.R_), Rd=np.diag(self.R_diff_), Q=np.diag(self.Q_),
                            Qf=np.diag(self.Q_final_), umax=[vm, sm, vm, sm], dumax=[inf, ds, inf, ds],
                            first_step=[inf, ds, inf, ds])

    def initialization(self, wheel_base, file_name):
        self.setRobotModelParameters(wheel_base=wheel_base)
        self.retriveReferenceFromCSV(file_name=file_name)
        self.interpolateReference()

    def start(self, current_pos_x, current_pos_y, current_pos_yaw):
        """One control tick (mpc.py:69-115): returns the (3,1) body twist [vx, vy, w] of the
        first optimal control, or False when not initialised."""
        if not self.isModelParameterRetrived_:
            print("Error: Model Parameters haven't been retrived.")
            return False
        if not self.isReferenceRetrived_:
            print("Error: Reference hasn't been retrived.")
            return False
        if self.control_input_ is None:
            self.control_input_ = np.zeros((self.NU_, self.HL_))
        output_v = None
        for _ in range(self.ITERATION_TIMES_):
            x_current = self.updateCurrentState(x=current_pos_x, y=current_pos_y, yaw=current_pos_yaw)
            x_ref = self.getReferenceTrajectoryWithinHorizon(x_current=x_current)
            x_predicted = self.predictMotion(x_ref=x_ref, x_current=x_current, control_input=self.control_input_)
            ox, oy, oyaw, fsp, fst, rsp, rst = self.controlLaw(x_ref=x_ref, x_current=x_current,
                                                               x_predicted=x_predicted,
                                                               u_predicted=self.control_input_)
            for i in range(self.HL_):   # raises TypeError on a failed solve, like the reference (:97-102)
                self.control_input_[0, i] = fsp[i]
                self.control_input_[1, i] = fst[i]
                self.control_input_[2, i] = rsp[i]
                self.control_input_[3, i] = rst[i]
            output_v = self.getVelocitiesFromRobotModels(front_speed=fsp[0], front_steer=fst[0],
                                                         rear_speed=rsp[0], rear_steer=rst[0])
        return output_v

    def controlLaw(self, x_ref, x_current, x_predicted, u_predicted):
        """x_ref (NX, HL+1), x_current (NX, 1), x_predicted (NX, HL+1), u_predicted (NU, HL) ->
        (x, y, yaw, front_speed, front_steer, rear_speed, rear_steer) or seven Nones."""
        x0 = [x_current[0, 0], x_current[1, 0], x_current[2, 0]]
        ok, u, x, _ = _scalar.linear_one(self._variant(), x_ref, x_predicted, x0, u_predicted)
        if not ok:
            return None, None, None, None, None, None, None
        return x[0].copy(), x[1].copy(), x[2].copy(), u[0].copy(), u[1].copy(), u[2].copy(), u[3].copy()

    def getVelocitiesFromRobotModels(self, front_steer, front_speed, rear_steer, rear_speed):
        if not self.isModelParameterRetrived_:
            print("Error: Model Parameters haven't been retrived.")
            return None
        vo = np.array([[math.cos(front_steer) * front_speed], [math.sin(front_steer) * front_speed],
                       [math.cos(rear_steer) * rear_speed], [math.sin(rear_steer) * rear_speed]])
        return self.kinematical_matrix_ @ vo

    def setRobotModelParameters(self, wheel_base):
        self.isModelParameterRetrived_ = True
        self.wheel_base_ = wheel_base
        # (H^T H)^-1 H^T for H = [[1,0,0],[0,1,wb/2],[1,0,0],[0,1,-wb/2]] in closed form
        self.kinematical_matrix_ = np.array([[0.5, 0.0, 0.5, 0.0],
                                             [0.0, 0.5, 0.0, 0.5],
                                             [0.0, 1.0 / wheel_base, 0.0, -1.0 / wheel_base]])

    def retriveReferenceFromCSV(self, file_name):
        """Rows: x | y | yaw | vx | vy | w | front angle | front speed | rear angle | rear speed |
        time_stamp; the header row is skipped."""
        with open(file_name, newline='') as f:
            rows = list(csv.reader(f))[1:]
        self.reference_ = [[float(e) for e in row] for row in rows]
        self.isReferenceRetrived_ = True

    def interpolateReference(self):
        """Ten points on the ICR arc between consecutive nodes with w != 0 (mpc.py:240-265)."""
        ref = self.reference_
        for i in range(len(ref) - 1):
            a, b = ref[i], ref[i + 1]
            vx, vy, w = b[3], b[4], b[5]
            if w == 0:
                continue
            icr = [-vy / w, vx / w]
            for k in range(10):
                yaw = a[2] + (b[2] - a[2]) / 10 * k
                self.pts_x_.append((math.cos(a[2]) - math.cos(yaw)) * icr[0]
                                   - (math.sin(a[2]) - math.sin(yaw)) * icr[1] + a[0])
                self.pts_y_.append((math.sin(a[2]) - math.sin(yaw)) * icr[0]
                                   + (math.cos(a[2]) - math.cos(yaw)) * icr[1] + a[1])
                self.pts_yaw_.append(yaw)
            self.pts_x_.append(b[0])
            self.pts_y_.append(b[1])
            self.pts_yaw_.append(b[2])

    def getReferenceTrajectoryWithinHorizon(self, x_current):
        """Drops everything up to the LAST stored point closer than trim_dist_ (in place),
        then returns the first HL+1 points (mpc.py:267-298)."""
        if not self.isReferenceRetrived_:
            return None
        last = -1
        px, py = float(np.ravel(x_current[0])[0]), float(np.ravel(x_current[1])[0])
        for i in range(len(self.pts_x_)):
            if math.sqrt((px - self.pts_x_[i]) ** 2 + (py - self.pts_y_[i]) ** 2) < self.trim_dist_:
                last = i
        if last >= 0:
            del self.pts_x_[:last + 1]
            del self.pts_y_[:last + 1]
            del self.pts_yaw_[:last + 1]
        x_ref = np.zeros((self.NX_, self.HL_ + 1))
        for i in range(self.HL_ + 1):
            x_ref[0, i] = self.pts_x_[i]
            x_ref[1, i] = self.pts_y_[i]
            x_ref[2, i] = self.pts_yaw_[i]
        return x_ref

    def getReferenceTrajectoryWithLookAheadDistance(self, x_current):
        """Reference window from the raw node list (mpc.py:300-345): drops every node up to the
        LAST one closer than trim_dist_ (in place), walks on from the first remaining node while
        the travelled distance stays within LOOKAHEAD_DIST_, and fills the HL+1 columns from the
        back with the nodes walked over, the first of them repeated in front."""
        if not self.isReferenceRetrived_:
            return None
        ref = self.reference_
        px, py = float(np.ravel(x_current[0])[0]), float(np.ravel(x_current[1])[0])
        near = [i for i, n in enumerate(ref) if math.sqrt((px - n[0]) ** 2 + (py - n[1]) ** 2) < self.trim_dist_]
        if near:
            del ref[:near[-1] + 1]
        m, travel = min(1, len(ref)), 0
        while m < len(ref):
            travel = travel + math.sqrt((ref[m][0] - ref[m - 1][0]) ** 2 + (ref[m][1] - ref[m - 1][1]) ** 2)
            if travel > self.LOOKAHEAD_DIST_:
                break
            m += 1
        x_ref = np.zeros((self.NX_, self.HL_ + 1))
        for j in range(self.HL_ + 1):
            node = ref[max(0, m - 1 - (self.HL_ - j))]
            x_ref[:, j] = node[:3]
        return x_ref

    def updateCurrentState(self, x, y, yaw):
        return np.array([[float(x)], [float(y)], [float(yaw)]])

    def getRobotModelMatrice(self, theta, v_f, delta_f, v_r, delta_r):
        return _scalar.linearize_one(self._variant(), theta, [v_f, delta_f, v_r, delta_r])

    def predictMotion(self, x_ref, x_current, control_input):
        """x_predicted (NX, HL+1); also appends every predicted pose to predict_state_ as the
        reference does (mpc.py:418)."""
        x0 = [x_current[0, 0], x_current[1, 0], x_current[2, 0]]
        xp = _scalar.rollout_one(self._variant(), x0, np.asarray(control_input)[:, :self.HL_])
        for i in range(1, self.HL_ + 1):
            self.predict_state_.append([xp[0, i], xp[1, i], xp[2, i]])
        return xp
