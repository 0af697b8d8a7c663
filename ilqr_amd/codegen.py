"""SymPy expressions -> straight-line C for host and gfx950 device, and plant models defined by the user.

The reference obtains plant derivatives from symbolic back-ends at run time
(pydrake.symbolic in control/models/car.py:5-52, CasADi elsewhere) and lets a
user define a plant by writing `discrete_dynamics`, `_cost_stage`, `_cost_final`
in Python (README.md:25-31).  Here the same expressions are differentiated once,
common sub-expressions are eliminated and the result is printed as plain C
functions (`ILQR_HD static inline`, where ILQR_HD is `__host__ __device__` under
hipcc and nothing under gcc).  `tools/gen_models.py` uses this for the four
shipped plants; `generate_user_header` / `build_user_library` use it for a plant
the user writes with the math namespace `ilqr_amd.sym`: the functions are run once
on SymPy symbols, the solver is compiled for that plant with hipcc, and the
resulting shared library is loaded like the built-in one.
"""
from __future__ import annotations

import hashlib
import os
import sys
import time

import sympy as sp
from sympy.printing.c import C99CodePrinter

_HERE = os.path.dirname(os.path.abspath(__file__))
VERBOSE = False
MAX_USER_W = 8        # nx + nu the lane-per-trajectory backward stage is built for (everything in registers)
WIDE_NX, WIDE_NU = 12, 4   # a plant beyond that is solved as a 12-state, 4-control plant by the 16-lane kernels


class _Printer(C99CodePrinter):
    """C printer that never calls pow() for the small rational powers that occur."""

    def _print_Pow(self, expr):
        b, e = expr.base, expr.exp
        if e.is_Integer:
            n = int(e)
            bs = self.parenthesize(b, 1000)
            if n == 0:
                return "1.0"
            mul = "*".join([bs] * abs(n))
            return f"({mul})" if n > 0 else f"(1.0/({mul}))"
        if e.is_Rational and e.q == 2:
            n = int(e.p)
            bs = self._print(b)
            root = f"sqrt({bs})"
            k = abs(n)
            # b^(k/2) = b^(k//2) * sqrt(b) for odd k
            pw = "*".join([f"({bs})"] * (k // 2) + [root])
            return f"({pw})" if n > 0 else f"(1.0/({pw}))"
        return super()._print_Pow(expr)

    def _print_Rational(self, expr):
        return f"({float(expr.p)!r}/{float(expr.q)!r})"

    def _print_Integer(self, expr):
        return f"{float(int(expr))!r}"

    def _print_Float(self, expr):
        return repr(float(expr))


_P = _Printer()


def _count_ops(exprs):
    """Count arithmetic in a list of (already CSE'd) expressions.
    FMA-able mul+add pairs are counted as separate mul and add (2 flops)."""
    c = dict(add=0, mul=0, div=0, sqrt=0, sin=0, cos=0, tan=0)
    for e in exprs:
        for node in sp.preorder_traversal(e):
            if isinstance(node, sp.Add):
                c["add"] += len(node.args) - 1
            elif isinstance(node, sp.Mul):
                nargs = len(node.args)
                # a*b**-1 is a division
                ndiv = sum(1 for a in node.args if isinstance(a, sp.Pow) and a.exp.is_negative)
                c["mul"] += max(nargs - 1 - ndiv, 0)
                c["div"] += ndiv
            elif isinstance(node, sp.Pow):
                ex = node.exp
                if ex.is_Integer:
                    c["mul"] += abs(int(ex)) - 1
                    # a lone reciprocal (not inside a Mul) is handled by the Mul case mostly
                elif ex.is_Rational and ex.q == 2:
                    c["sqrt"] += 1
                    c["mul"] += abs(int(ex.p)) // 2
            elif isinstance(node, sp.sin):
                c["sin"] += 1
            elif isinstance(node, sp.cos):
                c["cos"] += 1
            elif isinstance(node, sp.tan):
                c["tan"] += 1
    c["flops"] = sum(c.values())
    return c


def emit_function(name, inputs, outputs, doc, pre=None):
    """inputs: list of (cname, [symbols]); outputs: list of (cname, sympy Matrix or list).
    Dense row-major outputs; structural zeros are written as 0.0.
    pre: optional list of (symbol, expression) evaluated first, in order, as named temporaries."""
    flat = []
    for _, M in outputs:
        M = sp.Matrix(M)
        flat.extend(list(M))          # row-major
    pre = pre or []
    npre = len(pre)
    flat = [e for _, e in pre] + flat
    t0 = time.time()
    # sin(a) and cos(a) of the same argument share one range reduction: both come from one ILQR_SINCOS
    kinds = {}
    for e in flat:
        for node in sp.preorder_traversal(e):
            if isinstance(node, (sp.sin, sp.cos)):
                kinds.setdefault(node.args[0], set()).add(type(node))
    pairs = [a for a, k in kinds.items() if k == {sp.sin, sp.cos} and not a.has(sp.sin, sp.cos, sp.tan)]
    pairs.sort(key=str)
    sc = {a: (sp.Symbol(f"sn{i}"), sp.Symbol(f"cs{i}")) for i, a in enumerate(pairs)}
    if sc:
        rep = {}
        for a, (S, Cc) in sc.items():
            rep[sp.sin(a)] = S
            rep[sp.cos(a)] = Cc
        flat = [e.xreplace(rep) for e in flat]
    repl, red = sp.cse(flat, symbols=sp.numbered_symbols("t"), optimizations="basic", order="none")
    lines = []
    args = ", ".join([f"const double* __restrict__ {n}" for n, _ in inputs] +
                     [f"double* __restrict__ {n}" for n, _ in outputs])
    lines.append(f"/* {doc} */")
    lines.append(f"ILQR_HD static inline void {name}({args})")
    lines.append("{")
    sub = {}
    for n, syms in inputs:
        for i, s in enumerate(syms):
            sub[s] = sp.Symbol(f"{n}[{i}]")
    used = set()
    for e in [r for _, r in repl] + red:
        used |= e.free_symbols
    for n, syms in inputs:
        if not any(s in used for s in syms):
            lines.append(f"    (void){n};")
    for a in pairs:
        S, Cc = sc[a]
        lines.append(f"    double {S}, {Cc};")
        lines.append(f"    ILQR_SINCOS({_P.doprint(a.xreplace(sub))}, {S}, {Cc});")
    for t, e in repl:
        lines.append(f"    const double {t} = {_P.doprint(e.xreplace(sub))};")
    for k in range(npre):
        lines.append(f"    const double {pre[k][0]} = {_P.doprint(red[k].xreplace(sub))};")
    red = red[npre:]
    k = 0
    for n, M in outputs:
        M = sp.Matrix(M)
        for i in range(len(M)):
            lines.append(f"    {n}[{i}] = {_P.doprint(red[k].xreplace(sub))};")
            k += 1
    lines.append("}")
    ops = _count_ops([e for _, e in repl] + red)
    ops["sin"] += len(pairs); ops["cos"] += len(pairs); ops["flops"] += 2 * len(pairs)
    if VERBOSE:
        sys.stderr.write(f"  {name}: {len(repl)} temporaries, {ops['flops']} flops, cse {time.time() - t0:.1f}s\n")
    return "\n".join(lines) + "\n", ops


HEADER = """/* GENERATED by tools/gen_models.py -- do not edit.
 * {what}
 * Dense row-major outputs.  ILQR_HD = __host__ __device__ under hipcc.
 */
#pragma once
#include <math.h>
#ifndef ILQR_HD
#  if defined(__HIPCC__)
#    define ILQR_HD __host__ __device__
#  else
#    define ILQR_HD
#  endif
#endif
/* sine and cosine of one argument: one shared range reduction on the GPU */
#ifndef ILQR_SINCOS
#  if defined(__HIP_DEVICE_COMPILE__)
#    define ILQR_SINCOS(a, s, c) sincos((a), &(s), &(c))
#  else
#    define ILQR_SINCOS(a, s, c) ((s) = sin(a), (c) = cos(a))
#  endif
#endif

"""




# ---------------------------------------------------------------------------
# plants defined by the user
# ---------------------------------------------------------------------------
def user_expressions(nx, nu, n_par, dynamics, cost_stage, cost_final, diff_state=None, ndx=None, overrides=None):
    """Run the user's functions once on SymPy symbols: dynamics(x, u, p), cost_stage(x, u, p), cost_final(x, p) and,
    if the plant overrides CalcDifferentialState (control/models/model.py:22-23), diff_state(state, ref_state, p).
    Returns (xs, us, ps, f, l, lf), with a seventh entry (ss, rs, dx) when diff_state is given.

    A plant whose error state has another dimension than its state (ndx != nx: control/models/Quadrotor_Quat.py) also
    overrides stage, final and Q_terms (:248-275); `overrides` = dict(stage=, final=, q_terms=) holds them as functions
    of symbols plus p, and an eighth entry carries what they returned: dict(stage=[l_x, l_u, l_xx, l_ux, l_uu, f_x, f_u],
    final=[V_x, V_xx], q=[Q_x, Q_u, Q_xx, Q_ux, Q_uu], q_in=the symbols Q_terms was called with)."""
    if nx + nu > MAX_USER_W and (nx > WIDE_NX or nu > WIDE_NU):
        raise NotImplementedError(
            f"nx = {nx}, nu = {nu}: plants with nx + nu > {MAX_USER_W} are solved by the kernels built for "
            f"{WIDE_NX} states and {WIDE_NU} controls, so nx <= {WIDE_NX} and nu <= {WIDE_NU}")
    import numpy as np
    xs = sp.symbols(f"x0:{nx}", real=True)
    us = sp.symbols(f"u0:{nu}", real=True)
    ps = sp.symbols(f"p0:{max(1, n_par)}", real=True)
    x = np.array(xs, dtype=object)
    u = np.array(us, dtype=object)
    p = np.array(ps[:n_par], dtype=object)
    f = sp.Matrix([sp.sympify(e) for e in np.asarray(dynamics(x, u, p), dtype=object).ravel()])
    if f.shape[0] != nx:
        raise ValueError(f"discrete_dynamics returned {f.shape[0]} components, expected nx = {nx}")
    l = sp.sympify(cost_stage(x, u, p))
    lf = sp.sympify(cost_final(x, p))
    if diff_state is None:
        return xs, us, ps, f, l, lf
    ss = sp.symbols(f"s0:{nx}", real=True)
    rs = sp.symbols(f"r0:{nx}", real=True)
    dx = sp.Matrix([sp.sympify(e) for e in np.asarray(diff_state(np.array(ss, dtype=object), np.array(rs, dtype=object), p),
                                                       dtype=object).ravel()])
    ndx = nx if ndx is None else int(ndx)
    if dx.shape[0] != ndx:
        raise ValueError(f"CalcDifferentialState returned {dx.shape[0]} components, the plant declares ndx = {ndx}")
    if ndx == nx and not overrides:
        return xs, us, ps, f, l, lf, (ss, rs, dx)
    if not overrides or any(k not in overrides for k in ("stage", "final", "q_terms")):
        raise NotImplementedError(
            f"an error state of another dimension than the state (ndx = {ndx}, nx = {nx}) needs the plant's own stage, final "
            "and Q_terms as well (Quadrotor_Quat.py:248-275)")
    if nx + nu > MAX_USER_W:
        raise NotImplementedError("plants with their own Q_terms are solved one trajectory per lane: nx + nu <= "
                                  f"{MAX_USER_W}")

    def mat(v, rows, cols, what):
        a = np.asarray(v, dtype=object)
        if a.size != rows * cols:
            raise ValueError(f"{what}: {a.shape} where {rows} x {cols} is expected")
        return sp.Matrix(rows, cols, [sp.sympify(e) for e in a.ravel()])

    st = overrides["stage"](x, u, p)
    shapes = [(nx, 1), (nu, 1), (nx, nx), (nu, nx), (nu, nu), (ndx, ndx), (ndx, nu)]
    names = ["l_x", "l_u", "l_xx", "l_ux", "l_uu", "f_x", "f_u"]
    stage = [mat(v, r, c, "stage() " + k) for v, (r, c), k in zip(st, shapes, names)]
    fin = overrides["final"](x, p)
    final = [mat(fin[0], ndx, 1, "final() V_x"), mat(fin[1], ndx, ndx, "final() V_xx")]
    # Q_terms on symbols for every argument: what it does with them is the plant's own
    q_in = dict(l_x=sp.symbols(f"lx0:{nx}", real=True), l_u=sp.symbols(f"lu0:{nu}", real=True),
                l_xx=sp.symbols(f"lxx0:{nx * nx}", real=True), l_ux=sp.symbols(f"lux0:{nu * nx}", real=True),
                l_uu=sp.symbols(f"luu0:{nu * nu}", real=True), f_x=sp.symbols(f"fx0:{ndx * ndx}", real=True),
                f_u=sp.symbols(f"fu0:{ndx * nu}", real=True), V_x=sp.symbols(f"Vx0:{ndx}", real=True),
                V_xx=sp.symbols(f"Vxx0:{ndx * ndx}", real=True))
    arr = lambda k, shp: np.array(q_in[k], dtype=object).reshape(shp)
    q = overrides["q_terms"](x, arr("l_x", (nx,)), arr("l_u", (nu,)), arr("l_xx", (nx, nx)), arr("l_ux", (nu, nx)),
                             arr("l_uu", (nu, nu)), arr("f_x", (ndx, ndx)), arr("f_u", (ndx, nu)), arr("V_x", (ndx,)),
                             arr("V_xx", (ndx, ndx)), p)
    qs = [mat(q[0], ndx, 1, "Q_terms() Q_x"), mat(q[1], nu, 1, "Q_terms() Q_u"), mat(q[2], ndx, ndx, "Q_terms() Q_xx"),
          mat(q[3], nu, ndx, "Q_terms() Q_ux"), mat(q[4], nu, nu, "Q_terms() Q_uu")]
    return xs, us, ps, f, l, lf, (ss, rs, dx), dict(ndx=ndx, stage=stage, final=final, q=qs, q_in=q_in)


def is_wide(nx, nu):
    return nx + nu > MAX_USER_W


def pad_wide(nx, nu, exprs):
    """The same plant stated with WIDE_NX states and WIDE_NU controls, which is what the 16-lane backward kernel is
    written for.  The added states are constant zero (f = 0, no cost) and the added controls cost u^2/2 and move
    nothing: every term of the solver that involves them vanishes identically (their rows and columns of f_x, f_u, l_xx
    are zero, their Q_u is zero, so d = 0 and K = 0 for them, and the blocks 0 and I they add to the Q-function
    Hessian do not change which of its eigenvalues are negative), so cost, decisions and the real part of the
    trajectory are those of the plant as written."""
    xs, us, ps, f, l, lf = exprs[:6]
    xp = tuple(xs) + sp.symbols(f"x{nx}:{WIDE_NX}", real=True)
    up = tuple(us) + sp.symbols(f"u{nu}:{WIDE_NU}", real=True)
    fp = sp.Matrix(list(f) + [sp.Integer(0)] * (WIDE_NX - nx))
    lp = l + sp.Rational(1, 2) * sum(v * v for v in up[nu:])
    if len(exprs) == 6:
        return xp, up, ps, fp, lp, lf
    ss, rs, dx = exprs[6]
    sp_ = tuple(ss) + sp.symbols(f"s{nx}:{WIDE_NX}", real=True)
    rp = tuple(rs) + sp.symbols(f"r{nx}:{WIDE_NX}", real=True)
    dxp = sp.Matrix(list(dx) + [a - b for a, b in zip(sp_[nx:], rp[nx:])])
    return xp, up, ps, fp, lp, lf, (sp_, rp, dxp)


CODEGEN_VERSION = "3"   # part of the library key: bump when the printed C or the solver sources change meaning


def user_key(nx, nu, n_par, exprs):
    """Identity of a traced plant: dimensions and the canonical text of its three expressions."""
    xs, us, ps, f, l, lf = exprs[:6]
    text = "|".join([CODEGEN_VERSION, str(nx), str(nu), str(n_par), sp.srepr(f), sp.srepr(l), sp.srepr(lf)]
                    + ([sp.srepr(exprs[6][2])] if len(exprs) > 6 else [])
                    + ([sp.srepr(m) for k in ("stage", "final", "q") for m in exprs[7][k]] if len(exprs) > 7 else []))
    return hashlib.sha256(text.encode()).hexdigest()[:12]


def generate_user_header(nx, nu, n_par, exprs, name="user"):
    """C text of user_gen.h: USER_NX / USER_NU / USER_NP and user_gen_{dynamics,cost_stage,cost_final,stage,final}
    with the signatures of the car's generated functions."""
    wide = is_wide(nx, nu)
    real_nx, real_nu = nx, nu
    if wide:
        exprs = pad_wide(nx, nu, exprs)
        nx, nu = WIDE_NX, WIDE_NU
    xs, us, ps, f, l, lf = exprs[:6]
    diff = exprs[6] if len(exprs) > 6 else None
    own = exprs[7] if len(exprs) > 7 else None        # the plant's own stage / final / Q_terms (error-state plants)
    lx = sp.Matrix([l]).jacobian(xs)
    lu = sp.Matrix([l]).jacobian(us)
    lfx = sp.Matrix([lf]).jacobian(xs)
    ins = [("p", ps), ("x", xs), ("u", us)]
    out = HEADER.format(what=f"User-defined plant '{name}' (the Model interface of control/models/model.py:9-43).")
    out = out.replace("GENERATED by tools/gen_models.py", "GENERATED by ilqr_amd/codegen.py")
    out += f"#define USER_NX {nx}\n#define USER_NU {nu}\n#define USER_NP {max(1, n_par)}\n"
    out += f"#define USER_REAL_NX {real_nx}\n#define USER_REAL_NU {real_nu}\n"
    out += f"#define USER_NDX {own['ndx'] if own else nx}\n"
    if own:
        out += ("#define USER_HAS_QTERMS 1   /* the plant's own stage, final and Q_terms on an error state "
                "(Quadrotor_Quat.py:248-275) */\n")
    if wide:
        out += "#define USER_WIDE 1   /* stated with 12 states and 4 controls for the 16-lane backward kernel */\n"
    if diff is not None:
        out += "#define USER_HAS_DIFF 1   /* the plant's own CalcDifferentialState (model.py:22-23 overridden) */\n"
    out += "\n"
    parts = [
        ("user_gen_dynamics", ins, [("xn", f)], "x_next = discrete_dynamics(x, u)"),
        ("user_gen_cost_stage", ins, [("c", [l])], "stage cost"),
        ("user_gen_cost_final", ins[:2], [("c", [lf])], "final cost"),
        ("user_gen_stage", ins, [("l_x", lx), ("l_u", lu), ("l_xx", lx.jacobian(xs)), ("l_ux", lu.jacobian(xs)),
                                 ("l_uu", lu.jacobian(us)), ("f_x", f.jacobian(xs)), ("f_u", f.jacobian(us))],
         "stage derivatives: l_x l_u l_xx l_ux l_uu f_x f_u (row-major)"),
        ("user_gen_final", ins[:2], [("V_x", lfx), ("V_xx", lfx.jacobian(xs))], "final-cost derivatives"),
    ]
    if own:
        names = ["l_x", "l_u", "l_xx", "l_ux", "l_uu", "f_x", "f_u"]
        parts[3] = ("user_gen_stage", ins, list(zip(names, own["stage"])),
                    "the plant's own stage(): l_x l_u l_xx l_ux l_uu on the state, f_x f_u on the error state (row-major)")
        parts[4] = ("user_gen_final", ins[:2], [("V_x", own["final"][0]), ("V_xx", own["final"][1])], "the plant's own final()")
        qi = own["q_in"]
        parts.append(("user_gen_qterms", [("p", ps), ("x", xs)] + [(k, qi[k]) for k in
                                                                   ("l_x", "l_u", "l_xx", "l_ux", "l_uu", "f_x", "f_u", "V_x", "V_xx")],
                      list(zip(["Q_x", "Q_u", "Q_xx", "Q_ux", "Q_uu"], own["q"])), "the plant's own Q_terms()"))
    if diff is not None:
        parts.append(("user_gen_diff_state", [("p", ps), ("s", diff[0]), ("r", diff[1])], [("dx", diff[2])],
                      "CalcDifferentialState(state, ref_state)"))
    if wide:
        # what lane j of a team asks for: column j of [f_x f_u] as a directional derivative, and the cost terms of the
        # joint variable z = (x, u) in one piece
        vx = sp.symbols(f"vx0:{nx}", real=True)
        vu = sp.symbols(f"vu0:{nu}", real=True)
        jv = f.jacobian(xs) * sp.Matrix(vx) + f.jacobian(us) * sp.Matrix(vu)
        parts.append(("user_gen_dynamics_jvp", ins + [("vx", vx), ("vu", vu)], [("jv", jv)],
                      "(d x_next/dx) vx + (d x_next/du) vu"))
        zs = list(xs) + list(us)
        lz = sp.Matrix([l]).jacobian(zs)
        parts.append(("user_gen_cost_terms", ins, [("l_z", lz), ("l_zz", lz.jacobian(zs))],
                      "l_z [16] = [l_x l_u], l_zz [16x16] = [[l_xx, l_ux'], [l_ux, l_uu]]"))
    for fname, fins, fouts, doc in parts:
        text, _ = emit_function(fname, fins, fouts, doc)
        out += text
    return out


def user_library_dir(key, name, out_root=None):
    return os.path.join(out_root or os.path.join(_HERE, "lib", "user"), f"{name}_{key}")


def build_user_library(nx, nu, n_par, exprs, name, out_root=None, hipcc=None, force=False):
    """Compile the solver for a traced plant (hipcc, gfx950).  The library lands in
    <out_root>/<name>_<key>/libilqr_user.so (default out_root: ilqr_amd/lib/user), next to the generated
    user_gen.h and a stamp of the solver sources it was compiled from; it is reused only while that stamp is the
    current one (a library built from older kernels or an older ilqr_problem / ilqr_outputs layout is rebuilt)."""
    d = user_library_dir(user_key(nx, nu, n_par, exprs), name, out_root)
    so = os.path.join(d, "libilqr_user.so")
    stamp_file = os.path.join(d, "sources.sha256")
    stamp = solver_sources_stamp()
    if os.path.exists(so) and not force and os.path.exists(stamp_file) and open(stamp_file).read().strip() == stamp:
        return so
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "user_gen.h"), "w") as fh:
        fh.write(generate_user_header(nx, nu, n_par, exprs, name))
    from .build import build_library
    # (a plant stated with 12 + 4 takes the 16-lane kernels of the main unit: the lane-per-trajectory unit is empty)
    build_library(so, defines=("ILQR_USER_MODEL", "ILQR_ONLY_USER"), include_dirs=(d,), hipcc=hipcc,
                  no_kernels_in=("lane_plants.hip",) if is_wide(nx, nu) else ())
    with open(stamp_file, "w") as fh:
        fh.write(stamp + "\n")
    return so


def solver_sources_stamp():
    """Hash of everything a solver library is compiled from besides the plant's own header."""
    from .build import sources
    h = hashlib.sha256()
    for path in sorted(sources()):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()
