"""Course builder with the call signatures of the reference's cubic_spline_planner.py
(CubicSpline1D :12-171, CubicSpline2D :174-308, calc_spline_course :311-323).

It produces the course tables (cx, cy, cyaw) the per-tick hot path reads, once per
course, on the host.  Unlike the reference it evaluates all samples at once with
numpy and solves the tridiagonal system for the second-derivative coefficients
directly (Thomas algorithm) instead of a dense np.linalg.solve.
"""
from __future__ import annotations

import math

import numpy as np


class CubicSpline1D:
    """Natural cubic spline through (x_i, y_i), x ascending."""

    def __init__(self, x, y):
        xs = np.asarray(x, dtype=np.float64)
        ys = np.asarray(y, dtype=np.float64)
        h = np.diff(xs)
        if np.any(h < 0):
            raise ValueError("x coordinates must be sorted in ascending order")
        n = xs.size
        self.x, self.y, self.nx = xs, ys, n
        self.a = ys.copy()
        # rows 1..n-2:  h[i-1] c[i-1] + 2 (h[i-1]+h[i]) c[i] + h[i] c[i+1] = rhs[i];  c[0] = c[n-1] = 0
        c = np.zeros(n)
        if n > 2:
            slope = np.diff(ys) / h
            rhs = 3.0 * np.diff(slope)
            diag = 2.0 * (h[:-1] + h[1:])
            low = h[1:-1].copy()     # sub-diagonal of the (n-2)x(n-2) interior system
            up = h[1:-1].copy()
            m = n - 2
            dp = diag.copy()
            rp = rhs.copy()
            for i in range(1, m):    # forward elimination
                f = low[i - 1] / dp[i - 1]
                dp[i] -= f * up[i - 1]
                rp[i] -= f * rp[i - 1]
            sol = np.zeros(m)
            sol[-1] = rp[-1] / dp[-1]
            for i in range(m - 2, -1, -1):
                sol[i] = (rp[i] - up[i] * sol[i + 1]) / dp[i]
            c[1:-1] = sol
        self.c = c
        self.d = np.diff(c) / (3.0 * h)
        self.b = np.diff(ys) / h - h / 3.0 * (2.0 * c[:-1] + c[1:])

    def _segment(self, x):
        """bisect(self.x, x) - 1, vectorised (cubic_spline_planner.py:140-144)."""
        return np.searchsorted(self.x, x, side="right") - 1

    def _eval(self, x, order):
        x = np.asarray(x, dtype=np.float64)
        scalar = x.ndim == 0
        xv = np.atleast_1d(x)
        inside = (xv >= self.x[0]) & (xv <= self.x[-1])
        i = np.clip(self._segment(xv), 0, self.nx - 2)
        dx = xv - self.x[i]
        if order == 0:
            val = self.a[i] + self.b[i] * dx + self.c[i] * dx ** 2.0 + self.d[i] * dx ** 3.0
        elif order == 1:
            val = self.b[i] + 2.0 * self.c[i] * dx + 3.0 * self.d[i] * dx ** 2.0
        else:
            val = 2.0 * self.c[i] + 6.0 * self.d[i] * dx
        if scalar:
            return float(val[0]) if inside[0] else None
        return np.where(inside, val, np.nan)

    def calc_position(self, x):
        """y(x); None outside the data range (scalar input), NaN for array input."""
        return self._eval(x, 0)

    def calc_first_derivative(self, x):
        return self._eval(x, 1)

    def calc_second_derivative(self, x):
        return self._eval(x, 2)


class CubicSpline2D:
    """Cubic spline through 2-D waypoints, parameterised by cumulative chord length."""

    def __init__(self, x, y):
        x = np.asarray(x, dtype=np.float64)
        y = np.asarray(y, dtype=np.float64)
        self.ds = np.hypot(np.diff(x), np.diff(y))
        self.s = np.concatenate([[0.0], np.cumsum(self.ds)])
        self.sx = CubicSpline1D(self.s, x)
        self.sy = CubicSpline1D(self.s, y)

    def calc_position(self, s):
        return self.sx.calc_position(s), self.sy.calc_position(s)

    def calc_curvature(self, s):
        dx = self.sx.calc_first_derivative(s)
        ddx = self.sx.calc_second_derivative(s)
        dy = self.sy.calc_first_derivative(s)
        ddy = self.sy.calc_second_derivative(s)
        return (ddy * dx - ddx * dy) / ((dx ** 2 + dy ** 2) ** (3 / 2))

    def calc_yaw(self, s):
        dx = self.sx.calc_first_derivative(s)
        dy = self.sy.calc_first_derivative(s)
        if np.ndim(dx) == 0:
            return math.atan2(dy, dx)
        return np.arctan2(dy, dx)


def calc_spline_course(x, y, ds=0.1):
    """(rx, ry, ryaw, rk, s) as lists, sampled every `ds` of chord length."""
    sp = CubicSpline2D(x, y)
    s = np.arange(0, sp.s[-1], ds)
    rx, ry = sp.calc_position(s)
    ryaw = sp.calc_yaw(s)
    rk = sp.calc_curvature(s)
    return list(rx), list(ry), list(ryaw), list(rk), list(s)


def smooth_yaw(yaw):
    """In-place sequential unwrap of a course's yaw list (mpc_diff_drive.py:404-417)."""
    for i in range(len(yaw) - 1):
        d = yaw[i + 1] - yaw[i]
        while d >= math.pi / 2.0:
            yaw[i + 1] -= math.pi * 2.0
            d = yaw[i + 1] - yaw[i]
        while d <= -math.pi / 2.0:
            yaw[i + 1] += math.pi * 2.0
            d = yaw[i + 1] - yaw[i]
    return yaw
