"""TEST INFRASTRUCTURE ONLY -- NumPy/SymPy restatement of the reference plant models.

Nothing in the shipped package may import this module; it exists so that
  (a) the reference solver (imported from /root/reference when it is present)
      can be driven with plant models whose own source files cannot be imported
      in this image (they need pydrake / CasADi, which are not installed), and
  (b) the golden vectors produced that way can be checked by a second, independent statement of the same plants:
      the C checker's own (oracle/oracle_models.h for the car, oracle/plants_ad.cpp for the other three, written by
      hand from the reference and differentiated by forward-mode AD), which has to reproduce them.

Every function cites the reference lines it restates.  The arithmetic is
written once, over a tiny "math namespace" `m` that is either NumPy or SymPy,
which is the reference's own idiom (control/models/car.py:67).

Parity status
-------------
* CarModel: dynamics/cost arithmetic restated from control/models/car.py:64-94.
  The reference differentiates symbolically with pydrake.symbolic
  (car.py:5-29); here the same expressions are differentiated symbolically
  with SymPy.  Exact derivatives either way, so differences are rounding-level,
  but the derivative *engine* is not the reference's: model-level parity for
  the car is pinned by the imported reference solver driving this restated
  model and by finite-difference checks, not by pydrake itself.
* Acrobot / Euler-angle quadrotor / quaternion quadrotor: the reference builds
  CasADi graphs (acrobot.py:35-99, Quadrotor_EA.py:57-155,
  Quadrotor_Quat.py:74-175).  CasADi is not installed, and no output of those
  graphs is stored anywhere in the reference: MODEL-LEVEL PARITY UNPINNED.
  Known answers that are available (hover equilibrium, rotation-matrix
  convention) are checked in tests/.
"""
from __future__ import annotations

import numpy as np
import sympy as sp


# --------------------------------------------------------------------------
# math namespaces
# --------------------------------------------------------------------------
class _NpMath:
    sin = staticmethod(np.sin)
    cos = staticmethod(np.cos)
    tan = staticmethod(np.tan)
    sqrt = staticmethod(np.sqrt)


class _SpMath:
    sin = staticmethod(sp.sin)
    cos = staticmethod(sp.cos)
    tan = staticmethod(sp.tan)
    sqrt = staticmethod(sp.sqrt)


NP = _NpMath
SP = _SpMath


def _ns(x):
    """Pick the math namespace from the element type (car.py:67 idiom)."""
    x0 = x[0]
    return SP if isinstance(x0, sp.Basic) else NP


# --------------------------------------------------------------------------
# Car  (control/models/car.py)
# --------------------------------------------------------------------------
CAR_DT = 0.1          # car.py:77
CAR_EPS = 1e-6        # car.py:82, :90
CAR_W_CONTROL = 0.1   # car.py:86


def car_xdot(x, u):
    """car.py:64-74: [v cos(th), v sin(th), v tan(steer), u0, u1]."""
    m = _ns(x)
    heading, v, steer = x[2], x[3], x[4]
    return [v * m.cos(heading), v * m.sin(heading), v * m.tan(steer), u[0], u[1]]


def car_step(x, u):
    """car.py:76-79: x_next = xdot * dt + x (that operand order)."""
    xd = car_xdot(x, u)
    return [xd[i] * CAR_DT + x[i] for i in range(5)]


def car_cost_stage(x, u, r, v_target):
    """car.py:81-87."""
    m = _ns(x)
    c_circle = (m.sqrt(x[0] ** 2 + x[1] ** 2 + CAR_EPS) - r) ** 2
    c_speed = (x[3] - v_target) ** 2
    c_control = (u[0] ** 2 + u[1] ** 2) * CAR_W_CONTROL
    return c_circle + c_speed + c_control


def car_cost_final(x, r, v_target):
    """car.py:89-94."""
    m = _ns(x)
    c_circle = (m.sqrt(x[0] ** 2 + x[1] ** 2 + CAR_EPS) - r) ** 2
    c_speed = (x[3] - v_target) ** 2
    return c_circle + c_speed


# --------------------------------------------------------------------------
# Acrobot  (control/models/acrobot.py)
# --------------------------------------------------------------------------
ACRO = dict(m1=1.0, m2=1.0, l1=1.0, l2=2.0, lc1=0.5, lc2=1.0,
            Ic1=0.083, Ic2=0.33, b1=0.1, b2=0.1, g=9.81)  # acrobot.py:23-33


def acrobot_xdot(x, u):
    """acrobot.py:38-72.  The 2x2 solve M qdd = B u - Bias (acrobot.py:71,
    `ca.solve`) is written out by Cramer's rule; CasADi's linear-solver
    plugin may round differently in the last bits."""
    m = _ns(x)
    P = ACRO
    th1, th2, th1d, th2d = x[0], x[1], x[2], x[3]
    s1, s2, s12 = m.sin(th1), m.sin(th2), m.sin(th1 + th2)
    c2 = m.cos(th2)
    m2l1lc2 = P["m2"] * P["l1"] * P["lc2"]
    I1 = P["Ic1"] + P["m1"] * P["lc1"] ** 2
    I2 = P["Ic2"] + P["m2"] * P["lc2"] ** 2
    M11 = I1 + I2 + P["m2"] * P["l1"] ** 2 + 2 * m2l1lc2 * c2
    M12 = I2 + m2l1lc2 * c2
    M22 = I2
    cor1 = -2 * m2l1lc2 * s2 * th2d * th1d - m2l1lc2 * s2 * th2d ** 2
    cor2 = m2l1lc2 * s2 * th1d ** 2
    g = P["g"]
    grav1 = g * P["m1"] * P["lc1"] * s1 + g * P["m2"] * (P["l1"] * s1 + P["lc2"] * s12)
    grav2 = g * P["m2"] * P["lc2"] * s12
    bias1 = cor1 + grav1 + P["b1"] * th1d
    bias2 = cor2 + grav2 + P["b2"] * th2d
    r1 = 0.0 * u[0] - bias1      # B = [[0],[1]]  (acrobot.py:65-66)
    r2 = u[0] - bias2
    det = M11 * M22 - M12 * M12
    qdd1 = (M22 * r1 - M12 * r2) / det
    qdd2 = (M11 * r2 - M12 * r1) / det
    return [th1d, th2d, qdd1, qdd2]


# --------------------------------------------------------------------------
# Euler-angle quadrotor  (control/models/Quadrotor_EA.py)
# --------------------------------------------------------------------------
QUAD = dict(L=0.15, kF=1.0, kM=0.0245, m=0.775, g=9.81,
            I=(0.0015, 0.0025, 0.0035))  # Quadrotor_EA.py:47-54, Quadrotor_Quat.py:63-70


def _matvec(A, v):
    return [sum(A[i][j] * v[j] for j in range(len(v))) for i in range(len(A))]


def _cross(a, b):
    return [a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]]


def _rotor_wrench(u):
    """Quadrotor_EA.py:77-94 (identical in Quadrotor_Quat.py:88-106)."""
    P = QUAD
    uF = [P["kF"] * ui for ui in u]
    thrust = uF[0] + uF[1] + uF[2] + uF[3]
    Mx = P["L"] * (uF[1] - uF[3])
    My = P["L"] * (uF[2] - uF[0])
    uT = [P["kM"] * ui for ui in u]
    Mz = uT[0] - uT[1] + uT[2] - uT[3]
    return thrust, [Mx, My, Mz]


def quad_ea_xdot(x, u):
    """Quadrotor_EA.py:60-128.  State [p(3), yaw, pitch, roll, v(3),
    yaw_rate, pitch_rate, roll_rate]."""
    m = _ns(x)
    P = QUAD
    y, p, r = x[3], x[4], x[5]
    lin_vel = [x[6], x[7], x[8]]
    ang_vel = [x[9], x[10], x[11]]
    rpy_vel = [ang_vel[2], ang_vel[1], ang_vel[0]]               # :64
    cr, cp, cy = m.cos(r), m.cos(p), m.cos(y)
    sr, sp_, sy = m.sin(r), m.sin(p), m.sin(y)
    R = [[cp * cy, cy * sp_ * sr - sy * cr, cy * sp_ * cr + sy * sr],
         [sy * cp, sy * sp_ * sr + cy * cr, sy * sp_ * cr - cy * sr],
         [-sp_, cp * sr, cr * cp]]                                # :71-75
    thrust, tau = _rotor_wrench(u)
    # :83-87  Ftot_N = Fgrav_N + R_NB @ [0,0,thrust];  a = Ftot/m
    lin_acc = [(0.0 + R[0][2] * thrust) / P["m"],
               (0.0 + R[1][2] * thrust) / P["m"],
               (-P["m"] * P["g"] + R[2][2] * thrust) / P["m"]]
    M = [[1.0, 0.0, -sp_], [0.0, cr, sr * cp], [0.0, -sr, cr * cp]]  # :97-101
    w = _matvec(M, rpy_vel)                                       # :102
    Iw = [P["I"][i] * w[i] for i in range(3)]
    wxIw = _cross(w, Iw)
    body = [(tau[i] - wxIw[i]) * (1.0 / P["I"][i]) for i in range(3)]
    alpha = _matvec(R, body)                                      # :105
    Minv = [[cy / cp, sy / cp, 0.0], [-sy, cy, 0.0],
            [cy / cp * sp_, sy / cp * sp_, 1.0]]                  # :108-112
    yaw_rate, pitch_rate = ang_vel[0], ang_vel[1]                 # :116
    Mdt = [[-cy * sp_ * pitch_rate - sy * cp * yaw_rate, -cy * yaw_rate, 0.0],
           [-sy * sp_ * pitch_rate + cy * cp * yaw_rate, -sy * yaw_rate, 0.0],
           [-cp * pitch_rate, 0.0, 0.0]]                          # :118-122
    Mdt_v = _matvec(Mdt, rpy_vel)
    rpy_acc = _matvec(Minv, [alpha[i] - Mdt_v[i] for i in range(3)])  # :124
    ang_acc = [rpy_acc[2], rpy_acc[1], rpy_acc[0]]                # :125
    return lin_vel + ang_vel + lin_acc + ang_acc                  # :128


# --------------------------------------------------------------------------
# Quaternion quadrotor  (control/models/Quadrotor_Quat.py, maths/casadi_quaternions.py)
# --------------------------------------------------------------------------
def quat_normalized(q):
    """casadi_quaternions.py:8 -- every helper normalises first."""
    m = _ns(q)
    n = m.sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3])
    return [q[0] / n, q[1] / n, q[2] / n, q[3] / n]


def quat_left(qh):
    """casadi_quaternions.py:15-20, on the already-normalised quaternion."""
    w, x, y, z = qh
    return [[w, -x, -y, -z], [x, w, -z, y], [y, z, w, -x], [z, -y, x, w]]


def quat_attitude_jacobian(q):
    """casadi_quaternions.py:22: G = L(q^) H, the last three columns of L."""
    Lm = quat_left(quat_normalized(q))
    return [[Lm[i][j + 1] for j in range(3)] for i in range(4)]


def quat_rotation(q):
    """casadi_quaternions.py:26: R = H^T T L T L H."""
    Lm = quat_left(quat_normalized(q))
    T = [1.0, -1.0, -1.0, -1.0]
    # (T L T L)[i][j] = sum_k T_i L[i][k] T_k L[k][j]
    return [[sum(T[i + 1] * Lm[i + 1][k] * T[k] * Lm[k][j + 1] for k in range(4))
             for j in range(3)] for i in range(3)]


def quad_quat_xdot(x, u):
    """Quadrotor_Quat.py:78-111.  State [q(4), p(3), w(3), v(3)]."""
    P = QUAD
    quat = [x[0], x[1], x[2], x[3]]
    ang_vel = [x[7], x[8], x[9]]
    lin_vel = [x[10], x[11], x[12]]
    R = quat_rotation(quat)                                       # :83
    Rt = [[R[j][i] for j in range(3)] for i in range(3)]
    w = _matvec(Rt, ang_vel)                                      # :84
    G = quat_attitude_jacobian(quat)
    qdot = [0.5 * s for s in _matvec(G, w)]                       # :86
    thrust, tau = _rotor_wrench(u)
    lin_acc = [(0.0 + R[0][2] * thrust) / P["m"],
               (0.0 + R[1][2] * thrust) / P["m"],
               (-P["m"] * P["g"] + R[2][2] * thrust) / P["m"]]   # :95-99
    Iw = [P["I"][i] * w[i] for i in range(3)]
    wxIw = _cross(w, Iw)
    body = [(tau[i] - wxIw[i]) * (1.0 / P["I"][i]) for i in range(3)]
    ang_acc = _matvec(R, body)                                    # :109
    return qdot + lin_vel + ang_acc + lin_acc                     # :111


def quat_state_diff_jacobian(x):
    """Quadrotor_Quat.py:142-144: blkdiag(G(q) [4x3], I9) -> 13x12."""
    G = quat_attitude_jacobian([x[0], x[1], x[2], x[3]])
    J = [[0.0] * 12 for _ in range(13)]
    for i in range(4):
        for j in range(3):
            J[i][j] = G[i][j]
    for i in range(9):
        J[4 + i][3 + i] = 1.0
    return J


def quat_state_error(x, xref):
    """Quadrotor_Quat.py:150: state_err = Gx(x)^T (x - xref), 12 entries (the first three through G(q))."""
    Gx = quat_state_diff_jacobian(x)
    dx = [x[i] - xref[i] for i in range(13)]
    return [sum(Gx[i][j] * dx[i] for i in range(13)) for j in range(12)]


def quat_cost(x, u, xref, uref, Q, R, with_action=True):
    """Quadrotor_Quat.py:150-153: 0.5 e^T Q e + 0.5 du^T R du with e = quat_state_error, Q [12][12] and R [4][4]
    any matrices (rows of numbers or symbols)."""
    e = quat_state_error(x, xref)
    c = 0.5 * sum(e[i] * Q[i][j] * e[j] for i in range(12) for j in range(12))
    if with_action:
        du = [u[i] - uref[i] for i in range(4)]
        c = c + 0.5 * sum(du[i] * R[i][j] * du[j] for i in range(4) for j in range(4))
    return c


# --------------------------------------------------------------------------
# RK4 + chain-rule Jacobians (acrobot.py:76-97, Quadrotor_EA.py:132-149,
# Quadrotor_Quat.py:116-133).  CasADi differentiates the RK4 graph by AD; the
# chain rule below is the same derivative evaluated in a fixed order.
# --------------------------------------------------------------------------
def rk4_step(xdot, x, u, dt):
    n = len(x)
    f1 = xdot(x, u)
    f2 = xdot([x[i] + 0.5 * f1[i] * dt for i in range(n)], u)
    f3 = xdot([x[i] + 0.5 * f2[i] * dt for i in range(n)], u)
    f4 = xdot([x[i] + 1.0 * f3[i] * dt for i in range(n)], u)
    return [x[i] + (f1[i] + 2 * f2[i] + 2 * f3[i] + f4[i]) * dt / 6.0 for i in range(n)]


class _Lambdified:
    """SymPy-differentiated, lambdified continuous dynamics f, A=df/dx, B=df/du."""

    def __init__(self, xdot, nx, nu):
        xs = sp.symbols(f"x0:{nx}", real=True)
        us = sp.symbols(f"u0:{nu}", real=True)
        f = sp.Matrix(xdot(list(xs), list(us)))
        self.sym = dict(x=xs, u=us, f=f, A=f.jacobian(xs), B=f.jacobian(us))
        args = list(xs) + list(us)
        self._f = sp.lambdify(args, f, "numpy", cse=True)
        self._A = sp.lambdify(args, self.sym["A"], "numpy", cse=True)
        self._B = sp.lambdify(args, self.sym["B"], "numpy", cse=True)
        self.nx, self.nu = nx, nu

    def f(self, x, u):
        return np.asarray(self._f(*x, *u), dtype=np.float64).reshape(self.nx)

    def A(self, x, u):
        return np.asarray(self._A(*x, *u), dtype=np.float64).reshape(self.nx, self.nx)

    def B(self, x, u):
        return np.asarray(self._B(*x, *u), dtype=np.float64).reshape(self.nx, self.nu)


def rk4_with_jacobians(lam: _Lambdified, x, u, dt):
    """x_next, d x_next/dx, d x_next/du by the chain rule through the four stages."""
    n = lam.nx
    I = np.eye(n)
    x = np.asarray(x, dtype=np.float64)
    u = np.asarray(u, dtype=np.float64)
    k1 = lam.f(x, u); A1 = lam.A(x, u); B1 = lam.B(x, u)
    x2 = x + 0.5 * k1 * dt
    k2 = lam.f(x2, u); A2 = lam.A(x2, u); B2 = lam.B(x2, u)
    x3 = x + 0.5 * k2 * dt
    k3 = lam.f(x3, u); A3 = lam.A(x3, u); B3 = lam.B(x3, u)
    x4 = x + 1.0 * k3 * dt
    k4 = lam.f(x4, u); A4 = lam.A(x4, u); B4 = lam.B(x4, u)
    K1x = A1;                       K1u = B1
    K2x = A2 @ (I + 0.5 * dt * K1x); K2u = A2 @ (0.5 * dt * K1u) + B2
    K3x = A3 @ (I + 0.5 * dt * K2x); K3u = A3 @ (0.5 * dt * K2u) + B3
    K4x = A4 @ (I + dt * K3x);       K4u = A4 @ (dt * K3u) + B4
    xn = x + (k1 + 2 * k2 + 2 * k3 + k4) * dt / 6.0
    fx = I + (K1x + 2 * K2x + 2 * K3x + K4x) * dt / 6.0
    fu = (K1u + 2 * K2u + 2 * K3u + K4u) * dt / 6.0
    return xn, fx, fu


# --------------------------------------------------------------------------
# Model classes.  `make_models(Base)` builds them on top of a given `Model`
# base class so that the reference's own base-class methods (rollout,
# cost_trj, Q_terms, CalcDifferentialState: control/models/model.py:15-53) are
# the ones that run when Base is the imported reference class.
# --------------------------------------------------------------------------
def step_with_jacobians(lam, x, u, dt, euler):
    """(x_next, d x_next/dx, d x_next/du) of the RK4 step, or of the Euler step x + xdot dt."""
    if not euler:
        return rk4_with_jacobians(lam, x, u, dt)
    x = np.asarray(x, dtype=np.float64); u = np.asarray(u, dtype=np.float64)
    return x + lam.f(x, u) * dt, np.eye(len(x)) + lam.A(x, u) * dt, lam.B(x, u) * dt


def make_models(Base):
    class CarModel(Base):
        """control/models/car.py:54-102."""

        def __init__(self, r, v_target):
            self.nx, self.nu, self.ndx = 5, 2, 5
            self.r, self.v_target = r, v_target
            xs = sp.symbols("x0:5", real=True)
            us = sp.symbols("u0:2", real=True)
            l = car_cost_stage(list(xs), list(us), r, v_target)
            lf = car_cost_final(list(xs), r, v_target)
            f = sp.Matrix(car_step(list(xs), list(us)))
            lx = sp.Matrix([l]).jacobian(xs)
            lu = sp.Matrix([l]).jacobian(us)
            lfx = sp.Matrix([lf]).jacobian(xs)
            exprs = dict(l_x=lx, l_u=lu, l_xx=lx.jacobian(xs), l_ux=lu.jacobian(xs),
                         l_uu=lu.jacobian(us), f_x=f.jacobian(xs), f_u=f.jacobian(us),
                         lf_x=lfx, lf_xx=lfx.jacobian(xs))
            self.sym = dict(x=xs, u=us, **exprs)
            args = list(xs) + list(us)
            self._fn = {k: sp.lambdify(args if not k.startswith("lf") else list(xs), v, "numpy")
                        for k, v in exprs.items()}

        def discrete_dynamics(self, x, u):
            return np.array(car_step(x, u))

        def _cost_stage(self, x, u):
            return car_cost_stage(x, u, self.r, self.v_target)

        def _cost_final(self, x):
            return car_cost_final(x, self.r, self.v_target)

        def stage(self, x, u):
            a = (*x, *u)
            g = lambda k, shp: np.asarray(self._fn[k](*a), dtype=np.float64).reshape(shp)
            return (g("l_x", 5), g("l_u", 2), g("l_xx", (5, 5)), g("l_ux", (2, 5)),
                    g("l_uu", (2, 2)), g("f_x", (5, 5)), g("f_u", (5, 2)))

        def final(self, x):
            return (np.asarray(self._fn["lf_x"](*x), dtype=np.float64).reshape(5),
                    np.asarray(self._fn["lf_xx"](*x), dtype=np.float64).reshape(5, 5))

    class _QuadraticRK4Model(Base):
        """Shared shape of acrobot.py:101-166 and Quadrotor_EA.py:157-224."""
        _xdot = None
        _dims = (0, 0)
        _lam_cache = {}

        def __init__(self, Q, Qf, R, dt, N, integration="RK4", **kwargs):
            self.Q, self.Qf, self.R, self.dt, self.N = Q, Qf, R, dt, N
            self.nx, self.nu = self._dims
            self.ndx = self.nx
            self.xref = np.zeros(self.nx)
            self.uref = np.zeros(self.nu)
            self.euler = integration != "RK4"          # acrobot.py:76-86
            key = type(self).__name__
            if key not in _QuadraticRK4Model._lam_cache:
                _QuadraticRK4Model._lam_cache[key] = _Lambdified(type(self)._xdot, self.nx, self.nu)
            self.lam = _QuadraticRK4Model._lam_cache[key]

        def set_references(self, xref, uref):
            self.xref, self.uref = xref, uref

        def discrete_dynamics(self, x, u):
            x = np.asarray(x, dtype=np.float64); u = np.asarray(u, dtype=np.float64)
            f = self.lam.f
            dt = self.dt
            f1 = f(x, u)
            if self.euler:
                return x + f1 * dt
            f2 = f(x + 0.5 * f1 * dt, u)
            f3 = f(x + 0.5 * f2 * dt, u)
            f4 = f(x + 1.0 * f3 * dt, u)
            return x + (f1 + 2 * f2 + 2 * f3 + f4) * dt / 6.0

        def _cost_stage(self, x, u, xref, uref):
            xerr = (x - xref).reshape((self.nx, 1))
            uerr = u - uref
            return 0.5 * xerr.T @ self.Q @ xerr + 0.5 * uerr.T @ self.R @ uerr

        def _cost_final(self, xf, xref):
            return 0.5 * (xf - xref).T @ self.Qf @ (xf - xref)

        def cost_trj(self, xtraj, utraj):
            J = 0
            xf = xtraj[-1, :]
            for n in range(self.N - 1):
                J += self._cost_stage(xtraj[n, :], utraj[n, :], self.xref, self.uref)
            J += self._cost_final(xf, self.xref)
            return J

        def stage(self, x, u):
            _, f_x, f_u = step_with_jacobians(self.lam, x, u, self.dt, self.euler)
            l_x = self.Q @ (x - self.xref)
            l_u = self.R @ (u - self.uref)
            return l_x, l_u, self.Q, np.zeros((self.nu, self.nx)), self.R, f_x, f_u

        def final(self, x):
            return self.Qf @ (x - self.xref), self.Qf

    class Acrobot2DModel(_QuadraticRK4Model):
        _xdot = staticmethod(acrobot_xdot)
        _dims = (4, 1)

    class QuadrotorEAModel(_QuadraticRK4Model):
        _xdot = staticmethod(quad_ea_xdot)
        _dims = (12, 4)

    class QuadrotorQuatModel(Base):
        """control/models/Quadrotor_Quat.py:31-275; Q, Qf [12 x 12] and R [4 x 4] any matrices."""
        _cache = {}

        def __init__(self, Q, Qf, R, dt=0.01, N=20, integration="RK4", **kwargs):
            self.nx, self.nu, self.ndx = 13, 4, 12
            self.Q, self.Qf, self.R = (np.asarray(W, dtype=np.float64) for W in (Q, Qf, R))
            self.dt, self.N = dt, N
            self.xref = np.zeros(13)
            self.uref = np.zeros(4)
            self.euler = integration != "RK4"          # Quadrotor_Quat.py:116-126
            if "lam" not in self._cache:
                self._cache["lam"] = _Lambdified(quad_quat_xdot, 13, 4)
                xs = sp.symbols("x0:13", real=True)
                self._cache["G"] = sp.lambdify(list(xs), sp.Matrix(quat_state_diff_jacobian(list(xs))), "numpy", cse=True)
            self.lam = self._cache["lam"]
            # the cost and its derivatives are differentiated with this plant's weights written in as numbers
            # (entries that are zero drop out of the expressions); one set of functions per pair of weights
            self._fn = {False: self._cost_functions(self.Q, self.R), True: self._cost_functions(self.Qf, self.R)}

        def _cost_functions(self, W, R):
            key = (W.tobytes(), R.tobytes())
            if key not in self._cache:
                xs = sp.symbols("x0:13", real=True)
                us = sp.symbols("u0:4", real=True)
                xr = sp.symbols("xr0:13", real=True)
                ur = sp.symbols("ur0:4", real=True)
                num = lambda A: [[sp.Float(float(v)) if v != 0 else 0 for v in row] for row in A]
                c = quat_cost(list(xs), list(us), list(xr), list(ur), num(W), num(R))
                cx = sp.Matrix([c]).jacobian(xs)
                cu = sp.Matrix([c]).jacobian(us)
                args = list(xs) + list(us) + list(xr) + list(ur)
                ex = dict(c=sp.Matrix([c]), l_x=cx, l_u=cu, l_xx=cx.jacobian(xs), l_ux=cu.jacobian(xs),
                          l_uu=cu.jacobian(us))
                self._cache[key] = {k: sp.lambdify(args, v, "numpy", cse=True) for k, v in ex.items()}
            return self._cache[key]

        def set_references(self, xref, uref):
            self.xref, self.uref = xref, uref

        def _ev(self, key, x, u, final, shp):
            u = np.zeros(4) if u is None else u
            a = (*np.asarray(x).ravel(), *np.asarray(u).ravel(), *self.xref, *self.uref)
            return np.asarray(self._fn[final][key](*a), dtype=np.float64).reshape(shp)

        def G(self, x):
            return np.asarray(self._cache["G"](*np.asarray(x).ravel()), dtype=np.float64).reshape(13, 12)

        def discrete_dynamics(self, x, u):
            x = np.asarray(x, dtype=np.float64).ravel(); u = np.asarray(u, dtype=np.float64).ravel()
            f = self.lam.f
            dt = self.dt
            f1 = f(x, u)
            if self.euler:
                return x + f1 * dt
            f2 = f(x + 0.5 * f1 * dt, u)
            f3 = f(x + 0.5 * f2 * dt, u)
            f4 = f(x + 1.0 * f3 * dt, u)
            return x + (f1 + 2 * f2 + 2 * f3 + f4) * dt / 6.0

        def _cost_stage(self, x, u, xref, uref):
            return float(self._ev("c", x, u, False, ()))

        def _cost_final(self, xf, xref):
            # terminal cost = state part only, weights Qf (Quadrotor_Quat.py:153);
            # evaluate the expression at u = uref so the action term is 0.
            return float(self._ev("c", xf, self.uref, True, ()))

        def cost_trj(self, xtraj, utraj):
            J = 0
            xf = xtraj[-1, :]
            for n in range(self.N - 1):
                J += self._cost_stage(xtraj[n, :], utraj[n, :], self.xref, self.uref)
            J += self._cost_final(xf, self.xref)
            return J

        def CalcDifferentialState(self, state, ref_state):
            return self.G(state).T @ (np.asarray(state) - np.asarray(ref_state))

        def stage(self, x, u):
            _, A, B = step_with_jacobians(self.lam, x, u, self.dt, self.euler)
            G = self.G(x)
            f_x = G.T @ A @ G
            f_u = G.T @ B
            l_x = self._ev("l_x", x, u, False, (1, 13))
            l_u = self._ev("l_u", x, u, False, (1, 4))
            l_xx = self._ev("l_xx", x, u, False, (13, 13))
            l_uu = self._ev("l_uu", x, u, False, (4, 4))
            l_ux = self._ev("l_ux", x, u, False, (4, 13))
            return l_x, l_u, l_xx, l_ux, l_uu, f_x, f_u

        def final(self, x):
            G = self.G(x)
            Vxn = self._ev("l_x", x, self.uref, True, (1, 13))
            Vxxn = self._ev("l_xx", x, self.uref, True, (13, 13))
            V_xx = G.T @ Vxxn @ G
            V_x = Vxn @ G
            return V_x.T.ravel(), V_xx

        def Q_terms(self, state, l_x, l_u, l_xx, l_ux, l_uu, f_x, f_u, V_x, V_xx):
            G = self.G(state)
            Q_x = (l_x @ G).T.ravel() + f_x.T @ V_x
            Q_u = l_u.T.ravel() + f_u.T @ V_x
            Q_xx = G.T @ l_xx @ G + f_x.T @ V_xx @ f_x
            Q_ux = l_ux @ G + f_u.T @ V_xx @ f_x
            Q_uu = l_uu + f_u.T @ V_xx @ f_u
            return Q_x, Q_u, Q_xx, Q_ux, Q_uu

    return dict(CarModel=CarModel, Acrobot2DModel=Acrobot2DModel,
                QuadrotorEAModel=QuadrotorEAModel, QuadrotorQuatModel=QuadrotorQuatModel)
