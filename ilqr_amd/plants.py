"""The four plants of the reference, each stated once as plain arithmetic.

The reference takes plant derivatives from symbolic / AD engines at run time
(pydrake.symbolic: control/models/car.py:5-52; CasADi: control/models/acrobot.py:35-99,
Quadrotor_EA.py:57-155, Quadrotor_Quat.py:74-175).  Here the plant equations are
written over a small "math namespace" that is either NumPy or SymPy (the
reference's own idiom, control/models/car.py:67): tools/gen_models.py runs them on
SymPy symbols, differentiates, and prints the straight-line C in csrc/gen/ that the
kernels inline.  Every function cites the reference lines it follows.
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
