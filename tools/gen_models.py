#!/usr/bin/env python3
"""Model code generator: SymPy expressions -> straight-line C (host + gfx950 device).

The reference obtains plant derivatives from symbolic / AD back-ends at run
time (pydrake.symbolic in control/models/car.py:5-52; CasADi graphs in
control/models/acrobot.py:35-99, Quadrotor_EA.py:57-155,
Quadrotor_Quat.py:74-175).  Here the same expressions are differentiated once,
offline, common sub-expressions are eliminated, and the result is printed as
plain C functions.  Each function is declared `ILQR_HD static inline`, where
`ILQR_HD` expands to `__host__ __device__` under hipcc and to nothing under
gcc, so one text serves the HIP kernels and the CPU checker.

The symbolic expressions are obtained by running the plant arithmetic of
ilqr_amd/plants.py on SymPy symbols.

Outputs:
    ilqr_amd/csrc/gen/<model>_gen.h
    ilqr_amd/csrc/gen/opcounts.json  (operation counts of the emitted code; used for the
                                      algorithmic-flop figure of the roofline report)

The CPU checker has its own statement of the plants (written for driving the reference
solver) and its own copies of the headers, printed by the same functions from that
statement (`main(..., plants=..., out_dir=...)`); a test compares the two sets of headers.

Usage:  python tools/gen_models.py [car acrobot quad_ea quad_quat]
"""
from __future__ import annotations

import json
import os
import sys
import time

import sympy as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ilqr_amd import plants as mn  # noqa: E402  (module the gen_* functions take the plant arithmetic from)

OUT_DIR = os.path.join(ROOT, "ilqr_amd", "csrc", "gen")


import ilqr_amd.codegen as _cg  # noqa: E402
from ilqr_amd.codegen import HEADER, emit_function  # noqa: E402

_cg.VERBOSE = True


def gen_car():
    xs = sp.symbols("x0:5", real=True)
    us = sp.symbols("u0:2", real=True)
    ps = sp.symbols("p0:2", real=True)   # p = [r, v_target]
    x, u = list(xs), list(us)
    l = mn.car_cost_stage(x, u, ps[0], ps[1])
    lf = mn.car_cost_final(x, ps[0], ps[1])
    f = sp.Matrix(mn.car_step(x, u))
    lx = sp.Matrix([l]).jacobian(xs)
    lu = sp.Matrix([l]).jacobian(us)
    lfx = sp.Matrix([lf]).jacobian(xs)
    ins = [("p", ps), ("x", xs), ("u", us)]
    out = HEADER.format(what="Kinematic car: reference control/models/car.py:64-94 and its "
                             "symbolic derivatives (car.py:5-29).  p = [r, v_target].")
    ops = {}
    s, ops["dyn"] = emit_function("car_gen_dynamics", ins, [("xn", f)], "x_next = xdot*dt + x, dt=0.1 (car.py:76-79)")
    out += s
    s, ops["cost_stage"] = emit_function("car_gen_cost_stage", ins, [("c", [l])], "stage cost (car.py:81-87)")
    out += s
    s, ops["cost_final"] = emit_function("car_gen_cost_final", ins[:2], [("c", [lf])], "final cost (car.py:89-94)")
    out += s
    s, ops["stage"] = emit_function(
        "car_gen_stage", ins,
        [("l_x", lx), ("l_u", lu), ("l_xx", lx.jacobian(xs)), ("l_ux", lu.jacobian(xs)),
         ("l_uu", lu.jacobian(us)), ("f_x", f.jacobian(xs)), ("f_u", f.jacobian(us))],
        "stage derivatives (car.py:31-44): l_x[5] l_u[2] l_xx[5x5] l_ux[2x5] l_uu[2x2] f_x[5x5] f_u[5x2]")
    out += s
    s, ops["final"] = emit_function("car_gen_final", ins[:2], [("V_x", lfx), ("V_xx", lfx.jacobian(xs))],
                                    "final-cost derivatives (car.py:46-52)")
    out += s
    # ---- backward-stage terms, sparsity of f_x, f_u, l_xx folded in by SymPy.  V_xx comes in as the full
    # (not exactly symmetric) matrix the reference carries, and both triangles of Q_xx and Q_uu go out:
    # the reference never symmetrises, and the rounding-level asymmetry decides how ill-conditioned solves
    # regularise.
    n, m = 5, 2
    w = n + m
    Vx = sp.symbols("Vx0:5", real=True)
    Vs = sp.symbols("V0:25", real=True)
    V = sp.Matrix(n, n, lambda i, j: Vs[i * n + j])
    fx = f.jacobian(xs)
    fu = f.jacobian(us)
    # name the six state-dependent entries of f_x so that products are formed from them, not re-derived
    fa_pos = [(0, 2), (0, 3), (1, 2), (1, 3), (2, 3), (2, 4)]
    fa = sp.symbols("fa0:6", real=True)
    fxs = sp.Matrix(fx)
    for k, (i, j) in enumerate(fa_pos):
        fxs[i, j] = fa[k]
    assert all(fxs[i, j].is_number or fxs[i, j] in fa for i in range(n) for j in range(n))
    Sm = fxs.T * V                                # f_x' V_xx first, as model.py:50 evaluates left to right
    Tm = fu.T * V                                 # f_u' V_xx
    Ss = sp.Matrix(n, n, lambda i, j: sp.Symbol(f"S{i}{j}", real=True))
    Ts = sp.Matrix(m, n, lambda i, j: sp.Symbol(f"T{i}{j}", real=True))
    lxx = lx.jacobian(xs)
    Qxx = lxx + Ss * fxs
    Qux = lu.jacobian(xs) + Ts * fxs
    Quu = lu.jacobian(us) + Ts * fu
    Qx_e = lx.T + fxs.T * sp.Matrix(Vx)
    Qu_e = lu.T + fu.T * sp.Matrix(Vx)
    full = sp.zeros(w, w)
    full[:n, :n] = Qxx
    full[n:, :n] = Qux
    full[n:, n:] = Quu
    QL = [full[i, j] for i in range(w) for j in range(i + 1)]
    QU = [Qxx[i, j] for i in range(n) for j in range(i + 1, n)] + [Quu[i, j] for i in range(m) for j in range(i + 1, m)]
    pre = [(fa[k], fx[i, j]) for k, (i, j) in enumerate(fa_pos)] + \
          [(Ss[i, j], Sm[i, j]) for i in range(n) for j in range(n)] + \
          [(Ts[i, j], Tm[i, j]) for i in range(m) for j in range(n)]
    s, ops["backward_terms"] = emit_function(
        "car_gen_backward_terms", [("p", ps), ("x", xs), ("u", us), ("Vx", Vx), ("V", Vs)],
        [("Qx", Qx_e), ("Qu", Qu_e), ("QL", QL), ("QU", QU), ("fa", list(fa))],
        "Q-function terms of one backward stage (control/models/model.py:45-53) from the full V_xx [5x5]: "
        "Qx[5], Qu[2], QL[28] = lower triangle (packed by rows) of [[Qxx, Qux'], [Qux, Quu]], QU[11] = the entries "
        "above the diagonal of Qxx (row by row) and of Quu, fa[6] = the state-dependent "
        "entries (0,2) (0,3) (1,2) (1,3) (2,3) (2,4) of f_x", pre=pre)
    out += s
    Fm = sp.Matrix(n, w, lambda i, j: fxs[i, j] if j < n else fu[i, j - n])
    P = sp.Matrix(w, w, lambda i, j: sum((sp.Float(0.1) * Fm[t, i]) * Fm[t, j] for t in range(n)))
    s, ops["regu_p"] = emit_function(
        "car_gen_regu_p", [("fa", fa)], [("PL", [P[i, j] for i in range(w) for j in range(i + 1)])],
        "packed lower triangle of (0.1 [fx fu])'[fx fu], the regularisation update of iLQRController.py:111-113 at regu = 0.1")
    out += s
    return out, ops


def _gen_continuous(prefix, xdot, nx, nu, what):
    xs = sp.symbols(f"x0:{nx}", real=True)
    us = sp.symbols(f"u0:{nu}", real=True)
    f = sp.Matrix(xdot(list(xs), list(us)))
    A = f.jacobian(xs)
    B = f.jacobian(us)
    ins = [("x", xs), ("u", us)]
    out = HEADER.format(what=what)
    ops = {}
    s, ops["xdot"] = emit_function(f"{prefix}_gen_xdot", ins, [("f", f)], "continuous dynamics xdot = f(x,u)")
    out += s
    s, ops["xdot_jac"] = emit_function(f"{prefix}_gen_xdot_jac", ins, [("f", f), ("A", A), ("B", B)],
                                       f"f[{nx}], A = df/dx [{nx}x{nx}], B = df/du [{nx}x{nu}]")
    out += s
    # directional derivative: one column of [A B] (or any combination of columns) without forming the matrices
    vx = sp.symbols(f"vx0:{nx}", real=True)
    vu = sp.symbols(f"vu0:{nu}", real=True)
    jv = A * sp.Matrix(vx) + B * sp.Matrix(vu)
    s, ops["xdot_jvp"] = emit_function(f"{prefix}_gen_xdot_jvp", ins + [("vx", vx), ("vu", vu)], [("f", f), ("jv", jv)],
                                       f"f[{nx}] and the directional derivative jv = (df/dx) vx + (df/du) vu")
    out += s
    return out, ops, (xs, us)


def gen_acrobot():
    out, ops, _ = _gen_continuous(
        "acrobot", mn.acrobot_xdot, 4, 1,
        "Acrobot: reference control/models/acrobot.py:23-72 (manipulator equation, 2x2 solve "
        "written by Cramer's rule) and its continuous-time Jacobians (acrobot.py:90-91).")
    return out, ops


def gen_quad_ea():
    out, ops, _ = _gen_continuous(
        "quad_ea", mn.quad_ea_xdot, 12, 4,
        "Euler-angle quadrotor: reference control/models/Quadrotor_EA.py:47-128 and its "
        "continuous-time Jacobians (Quadrotor_EA.py:146-147).")
    return out, ops


def gen_quad_quat():
    out, ops, (xs, us) = _gen_continuous(
        "quad_quat", mn.quad_quat_xdot, 13, 4,
        "Quaternion quadrotor: reference control/models/Quadrotor_Quat.py:63-111 with "
        "maths/casadi_quaternions.py:5-34, continuous-time Jacobians (Quadrotor_Quat.py:130-131), "
        "attitude Jacobian and the attitude part of the error state with its derivatives "
        "(Quadrotor_Quat.py:142-150).")
    G = sp.Matrix(mn.quat_attitude_jacobian(list(xs[:4])))
    s, ops["G"] = emit_function("quad_quat_gen_G", [("x", xs)], [("G", G)],
                                "attitude Jacobian G(q) [4x3] of the normalised quaternion "
                                "(casadi_quaternions.py:22); the 13x12 state Jacobian is blkdiag(G, I9)")
    out += s
    # The error state e = Gx(x)'(x - xref) is nonlinear in x only through its first three entries,
    # a = G(q)'(q - qref); the other nine are differences.  The cost 0.5 e'Qe + 0.5 du'R du and its derivatives for
    # ANY Q, R are assembled from a, da/dq and the three Hessians of a by the caller (models.hpp), so that the weights
    # stay run-time parameters.
    qs = list(xs[:4])
    qr = sp.symbols("qr0:4", real=True)
    a = sp.Matrix(mn.quat_state_error(list(xs), list(qr) + [0] * 9)[:3])
    s, ops["err"] = emit_function("quad_quat_gen_err", [("qr", qr), ("x", xs)], [("a", a)],
                                  "a[3] = G(q)'(q - qref): the attitude entries of state_err (Quadrotor_Quat.py:150)")
    out += s
    Ja = a.jacobian(qs)
    Ha = [sp.hessian(a[i], qs) for i in range(3)]
    s, ops["err_derivs"] = emit_function(
        "quad_quat_gen_err_derivs", [("qr", qr), ("x", xs)],
        [("a", a), ("Ja", Ja), ("Ha", [Ha[i][r, c] for i in range(3) for r in range(4) for c in range(4)])],
        "a[3], Ja = da/dq [3x4], Ha[i] = d2 a_i / dq2 [3][4x4]")
    out += s
    return out, ops


GENERATORS = dict(car=gen_car, acrobot=gen_acrobot, quad_ea=gen_quad_ea, quad_quat=gen_quad_quat)


def main(argv, plants=None, out_dir=None):
    """Headers for the named plants (default: all).  `plants`: another module stating the same plant functions;
    `out_dir`: where its headers go (no operation counts are written then)."""
    global mn
    if plants is not None:
        mn = plants
    which = argv[1:] or list(GENERATORS)
    d = out_dir or OUT_DIR
    os.makedirs(d, exist_ok=True)
    ops_path = os.path.join(OUT_DIR, "opcounts.json")
    allops = json.load(open(ops_path)) if os.path.exists(ops_path) else {}
    for name in which:
        sys.stderr.write(f"[gen] {name}\n")
        text, ops = GENERATORS[name]()
        with open(os.path.join(d, f"{name}_gen.h"), "w") as fh:
            fh.write(text)
        allops[name] = ops
    if out_dir is None:
        with open(ops_path, "w") as fh:
            json.dump(allops, fh, indent=1, sort_keys=True)
            fh.write("\n")


if __name__ == "__main__":
    main(sys.argv)
