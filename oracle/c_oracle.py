"""TEST INFRASTRUCTURE ONLY -- ctypes binding of the CPU checker (oracle/ilqr_oracle.c).

Only tests/, the smoke check and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libilqr_oracle.so")

TRACE_MAX = 64
MODEL_IDS = dict(car=0, acrobot=1, quad_ea=2, quad_quat=3, user=4)
EXIT_NAMES = {0: "d_tol", 1: "deltaJ<1", 2: "max_iter", 3: "linalg_error"}

INFO_DTYPE = np.dtype([
    ("J0", "<f8"), ("J_last", "<f8"), ("dmax", "<f8"), ("max_abs_q", "<f8"),
    ("J", "<f8", (TRACE_MAX,)), ("deltaJ", "<f8", (TRACE_MAX,)),
    ("ls_evals", "<i4", (TRACE_MAX,)), ("accepted", "u1", (TRACE_MAX,)),
    ("iters", "<i4"), ("status", "<i4"), ("regu_rounds", "<i4"), ("ls_total", "<i4"),
    ("regu_stages", "<i4"), ("pad_", "<i4"),
], align=True)


def build(force: bool = False) -> str:
    if force or not os.path.exists(_LIB_PATH) or not os.path.exists(os.path.join(_HERE, "_build", "libeigvals_lapack.so")):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


_libs = {}


def build_user(header_dir: str, out_dir: str = None) -> str:
    """The checker compiled with the user-defined plant whose user_gen.h lies in header_dir (model id 4)."""
    out_dir = out_dir or os.path.join(_HERE, "_build", "user_" + os.path.basename(os.path.normpath(header_dir)))
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, "libilqr_oracle_user.so")
    src = os.path.join(_HERE, "ilqr_oracle.c")
    hdr = os.path.join(header_dir, "user_gen.h")
    plants = os.path.join(_HERE, "plants_ad.cpp")
    deps = [src, hdr, plants, os.path.join(_HERE, "eigvals_lapack.h"), os.path.join(_HERE, "ilqr_oracle.h"),
            os.path.join(_HERE, "oracle_models.h")]
    if not os.path.exists(out) or os.path.getmtime(out) < max(os.path.getmtime(d) for d in deps):
        obj = os.path.join(out_dir, "plants_ad.o")
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fno-fast-math", "-fPIC", "-fno-exceptions", "-fno-rtti",
                               "-c", "-o", obj, plants])
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-mfma", "-fPIC", "-fopenmp",
                               "-Wno-unused-function", "-DORACLE_USER_MODEL", f"-I{header_dir}", f"-I{_HERE}",
                               "-shared", "-o", out, src, obj, "-lm"])
    return out


def lib(path=None):
    if path is None:
        path = _LIB_PATH
        if not os.path.exists(path):
            build()
    if path not in _libs:
        L = C.CDLL(path)
        dp = C.POINTER(C.c_double)
        L.ilqr_oracle_solve_batch.restype = C.c_int
        L.ilqr_oracle_solve_batch.argtypes = [C.c_int, dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_long,
                                              dp, dp, dp, C.c_void_p, C.c_int]
        L.ilqr_oracle_model_dims.restype = None
        L.ilqr_oracle_model_dims.argtypes = [C.c_int] + [C.POINTER(C.c_int)] * 4
        L.ilqr_oracle_eval.restype = C.c_int
        L.ilqr_oracle_eval.argtypes = [C.c_int] + [dp] * 15
        L.ilqr_oracle_has_negative_eig.restype = C.c_int
        L.ilqr_oracle_has_negative_eig.argtypes = [dp, C.c_int, dp]
        L.ilqr_oracle_has_negative_eig_lapack.restype = C.c_int
        L.ilqr_oracle_has_negative_eig_lapack.argtypes = [dp, C.c_int, dp]
        L.ilqr_oracle_set_psd_rule.restype = None
        L.ilqr_oracle_set_psd_rule.argtypes = [C.c_int]
        L.ilqr_oracle_set_cost_N.restype = None
        L.ilqr_oracle_set_cost_N.argtypes = [C.c_int]
        L.ilqr_oracle_has_negative_eig_ldl.restype = C.c_int
        L.ilqr_oracle_has_negative_eig_ldl.argtypes = [dp, C.c_int, dp]
        L.ilqr_oracle_unit_qterms.restype = None
        L.ilqr_oracle_unit_qterms.argtypes = [C.c_int, C.c_int] + [dp] * 14
        L.ilqr_oracle_unit_gains_vterms.restype = C.c_int
        L.ilqr_oracle_unit_gains_vterms.argtypes = [C.c_int, C.c_int] + [dp] * 10
        _libs[path] = L
    return _libs[path]


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def dims(model, library=None):
    mid = MODEL_IDS[model] if isinstance(model, str) else model
    v = [C.c_int() for _ in range(4)]
    lib(library).ilqr_oracle_model_dims(mid, *[C.byref(x) for x in v])
    return tuple(x.value for x in v)


def info_size():
    return INFO_DTYPE.itemsize


def solve_batch(model, par, N, max_iter, max_ls, d_tol, x0, u, nthreads=0, library=None, rule="lapack", cost_N=0):
    """x0[B,nx], u[B,N-1,nu] (not modified) -> dict(x[B,N,nx], u[B,N-1,nu], info structured array).
    rule: how `np.any(np.linalg.eigvals(Q) < 0)` is decided -- "lapack" (DGEEV restated: the reference's own verdicts)
    or "ldl" (the LDL' rule of the HIP kernels: the kernels' algorithm on the CPU).
    cost_N: knot points the trajectory cost sums over when the plant was built with another horizon than N (0 = N)."""
    mid = MODEL_IDS[model] if isinstance(model, str) else model
    nx, nu, ndx, npar = dims(mid, library)
    par = np.ascontiguousarray(par, dtype=np.float64)
    assert par.size == npar, (par.size, npar)
    x0 = np.ascontiguousarray(x0, dtype=np.float64).reshape(-1, nx)
    B = x0.shape[0]
    u = np.array(u, dtype=np.float64, order="C").reshape(B, N - 1, nu)
    x = np.zeros((B, N, nx))
    info = np.zeros(B, dtype=INFO_DTYPE)
    if nthreads <= 0:
        nthreads = len(os.sched_getaffinity(0))
    lib(library).ilqr_oracle_set_psd_rule(dict(lapack=0, ldl=1)[rule])
    lib(library).ilqr_oracle_set_cost_N(0 if int(cost_N) == int(N) else int(cost_N))
    rc = lib(library).ilqr_oracle_solve_batch(mid, _p(par), N, max_iter, max_ls, float(d_tol), B, _p(x0), _p(u), _p(x),
                                       info.ctypes.data_as(C.c_void_p), nthreads)
    lib(library).ilqr_oracle_set_psd_rule(0)
    lib(library).ilqr_oracle_set_cost_N(0)
    assert rc == 0
    return dict(x=x, u=u, info=info)


def eval_point(model, par, x, u, library=None):
    mid = MODEL_IDS[model] if isinstance(model, str) else model
    nx, nu, ndx, npar = dims(mid, library)
    par = np.ascontiguousarray(par, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    out = dict(xn=np.zeros(nx), cs=np.zeros(1), cf=np.zeros(1), l_x=np.zeros(nx), l_u=np.zeros(nu),
               l_xx=np.zeros((nx, nx)), l_ux=np.zeros((nu, nx)), l_uu=np.zeros((nu, nu)),
               f_x=np.zeros((ndx, ndx)), f_u=np.zeros((ndx, nu)), V_x=np.zeros(ndx), V_xx=np.zeros((ndx, ndx)))
    rc = lib(library).ilqr_oracle_eval(mid, _p(par), _p(x), _p(u), *[_p(out[k]) for k in
                                ("xn", "cs", "cf", "l_x", "l_u", "l_xx", "l_ux", "l_uu", "f_x", "f_u", "V_x", "V_xx")])
    assert rc == 0
    out["cs"] = float(out["cs"][0]); out["cf"] = float(out["cf"][0])
    return out


def has_negative_eig(Q, rule="lapack"):
    """np.any(np.linalg.eigvals(Q) < 0) by the checker: LAPACK restated (default) or the older LDL^T rule."""
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    W = np.zeros_like(Q)
    f = lib().ilqr_oracle_has_negative_eig_lapack if rule == "lapack" else lib().ilqr_oracle_has_negative_eig_ldl
    return f(_p(Q), Q.shape[0], _p(W))


# parameter-block packers (layouts documented in oracle/ilqr_oracle.h)
def pack_car(r, v_target):
    return np.array([r, v_target], dtype=np.float64)


def pack_quadratic(dt, Q, Qf, R, xref, uref, euler=False):
    """[dt, Q, Qf, R, xref, uref, euler]: the block of the acrobot and of both quadrotors (full matrices, row-major;
    euler: `integration` other than "RK4")."""
    return np.concatenate([[dt], np.asarray(Q, float).ravel(), np.asarray(Qf, float).ravel(),
                           np.asarray(R, float).ravel(), np.asarray(xref, float).ravel(),
                           np.asarray(uref, float).ravel(), [1.0 if euler else 0.0]])


pack_quat = pack_quadratic
