"""ctypes binding of libilqr_hip.so (C ABI declared in include/ilqr_hip.h).

The shared library is the only compute path: if it is missing or cannot be
loaded this module raises, there is no CPU fallback.  PyTorch is used only to
own device memory and streams.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ILQR_HIP_LIB") or os.path.join(_HERE, "lib", "libilqr_hip.so")   # override: A/B builds

MODEL_CAR, MODEL_ACROBOT, MODEL_QUAD_EA, MODEL_QUAD_QUAT, MODEL_USER = 0, 1, 2, 3, 4
EXIT_DTOL, EXIT_SMALL_DELTAJ, EXIT_MAXITER, EXIT_LINALG = 0, 1, 2, 3
EXIT_NAMES = {0: "d_tol", 1: "deltaJ<1", 2: "max_iter", 3: "linalg_error"}
FLAG_PARALLEL_LS = 1
FLAG_LS_SEQUENTIAL = 2
FLAG_TIMING = 4
FLAG_SINGLE_STREAM = 8
FLAG_PARAMS_PER_TRAJECTORY = 16
FLAG_LAPACK_EIGVALS = 32
FLAG_NO_REPLAY = 64
PHASES = ("init", "backward", "forward1", "backtrack", "commit", "export")

EXPORTS = ("ilqr_model_info", "ilqr_workspace_bytes", "ilqr_solve_batch", "ilqr_rollout_cost",
           "ilqr_model_eval", "ilqr_mpc_advance", "ilqr_last_timing", "ilqr_last_error", "ilqr_version",
           "ilqr_solve_batch_static", "ilqr_eigvals_negative", "ilqr_linearize_continuous", "ilqr_abi_version")
STATIC_WORKSPACE_BYTES = 32 << 20
ABI_VERSION = 2          # ILQR_ABI_VERSION of include/ilqr_hip.h: the struct layouts declared below

_dp = C.c_void_p  # device pointers travel as plain addresses


class Problem(C.Structure):
    _fields_ = [("model_id", C.c_int32), ("N", C.c_int32), ("max_iter", C.c_int32), ("max_ls", C.c_int32),
                ("d_tol", C.c_double), ("B", C.c_int64), ("stride", C.c_int64), ("params", _dp),
                ("trace_len", C.c_int32), ("flags", C.c_int32), ("cost_N", C.c_int32), ("reserved_", C.c_int32)]


class Outputs(C.Structure):
    _fields_ = [(k, _dp) for k in ("J0", "J", "dmax", "max_abs_q", "iters", "status", "regu_rounds", "ls_total",
                                   "regu_stages", "ls_evals", "accepted", "J_trace", "deltaJ_trace", "last_search",
                                   "replayed")]


class Timing(C.Structure):
    _fields_ = [("ms", C.c_double * 6), ("launches", C.c_int32 * 6), ("iterations", C.c_int32), ("pad_", C.c_int32)]


class LibraryMissing(RuntimeError):
    pass


_libs = {}


def lib(path=None):
    """Load a solver library (once per path; default libilqr_hip.so).  Raises LibraryMissing when it is not built."""
    path = os.path.abspath(path or LIB_PATH)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise LibraryMissing(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    L = C.CDLL(path)
    L.ilqr_model_info.restype = C.c_int
    L.ilqr_model_info.argtypes = [C.c_int32] + [C.POINTER(C.c_int32)] * 4
    L.ilqr_workspace_bytes.restype = C.c_size_t
    L.ilqr_workspace_bytes.argtypes = [C.POINTER(Problem)]
    L.ilqr_solve_batch.restype = C.c_int
    L.ilqr_solve_batch.argtypes = [C.POINTER(Problem), _dp, _dp, _dp, C.POINTER(Outputs), _dp, C.c_size_t, _dp]
    L.ilqr_solve_batch_static.restype = C.c_int
    L.ilqr_solve_batch_static.argtypes = [C.POINTER(Problem), _dp, _dp, _dp, C.POINTER(Outputs), _dp]
    L.ilqr_eigvals_negative.restype = C.c_int
    L.ilqr_eigvals_negative.argtypes = [C.c_int32, _dp, C.c_int64, C.c_int64, _dp, _dp]
    L.ilqr_rollout_cost.restype = C.c_int
    L.ilqr_rollout_cost.argtypes = [C.POINTER(Problem), _dp, _dp, _dp, _dp, _dp]
    L.ilqr_model_eval.restype = C.c_int
    L.ilqr_model_eval.argtypes = [C.POINTER(Problem)] + [_dp] * 15
    L.ilqr_linearize_continuous.restype = C.c_int
    L.ilqr_linearize_continuous.argtypes = [C.POINTER(Problem), _dp, _dp, _dp, _dp, _dp]
    L.ilqr_mpc_advance.restype = C.c_int
    L.ilqr_mpc_advance.argtypes = [C.POINTER(Problem), _dp, _dp, _dp]
    L.ilqr_last_timing.restype = C.c_int
    L.ilqr_last_timing.argtypes = [C.POINTER(Timing)]
    L.ilqr_last_error.restype = C.c_char_p
    L.ilqr_version.restype = C.c_char_p
    L.ilqr_abi_version.restype = C.c_int32
    if L.ilqr_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path}: struct layout version {L.ilqr_abi_version()}, this binding is written for {ABI_VERSION}")
    _libs[path] = L
    return L


def check(rc: int, L=None):
    if rc != 0:
        raise RuntimeError(f"libilqr_hip error {rc}: {(L or lib()).ilqr_last_error().decode()}")


def model_info(model_id: int, L=None):
    L = L or lib()
    v = [C.c_int32() for _ in range(4)]
    check(L.ilqr_model_info(model_id, *[C.byref(x) for x in v]), L)
    return tuple(x.value for x in v)
