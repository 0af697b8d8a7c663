"""Batched host API over libilqr_hip: device buffers in, device buffers out.

`BatchSolver` is the batched form of the reference's
`iLQR(model, N, max_iter, max_linesearch_iters, d_tol).run_ilqr(x0, u_trj)`
(control/iLQRController.py:8-84): B independent trajectories per call.

Two entry levels:
  * `solve_soa` works on strided structure-of-arrays device tensors
    (x0 [nx, stride], u [N-1, nu, stride], x [N, nx, stride]) with no copies or
    layout changes -- this is what the C ABI takes;
  * `solve` accepts trajectory-major arrays (x0 [B, nx], u [B, N-1, nu]; NumPy
    or torch, host or device), transposes on the device and returns
    trajectory-major results.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Optional

import numpy as np
import torch

from . import _lib


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


@dataclass
class BatchResult:
    """Per-trajectory results of one batched solve (tensors on the solver's device)."""
    x: torch.Tensor
    u: torch.Tensor
    J0: torch.Tensor
    J: torch.Tensor
    dmax: torch.Tensor
    max_abs_q: torch.Tensor
    iters: torch.Tensor
    status: torch.Tensor
    regu_rounds: torch.Tensor
    ls_total: torch.Tensor
    regu_stages: torch.Tensor = None
    last_search: torch.Tensor = None      # 0 no line search in this call, 1 the last one accepted, 2 it was abandoned
    replayed: torch.Tensor = None         # iterations not run because they repeat the one before them exactly
    traces: dict = field(default_factory=dict)


class BatchSolver:
    def __init__(self, model_id: int, params, N: int, max_iter: int, max_linesearch_iters: int = 20,
                 d_tol: float = 1e-3, device="cuda:0", trace_len: int = 0, flags: int = 0, library=None,
                 cost_N: int = 0):
        # library: path of a solver library built for a user-defined plant (ilqr_amd.models.user); default built-ins
        self.lib = _lib.lib(library)
        self.model_id = int(model_id)
        self.nx, self.nu, self.ndx, self.npar = _lib.model_info(self.model_id, self.lib)
        self.N, self.max_iter, self.max_ls, self.d_tol = int(N), int(max_iter), int(max_linesearch_iters), float(d_tol)
        if self.N < 2:
            raise ValueError("N must be >= 2")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("BatchSolver needs a GPU device (there is no CPU path)")
        self.trace_len = int(trace_len)
        self.flags = int(flags)
        # knot points the trajectory cost sums over (0 = N): ilqr_problem::cost_N
        self.cost_N = int(cost_N)
        self.set_params(params)
        self._ws = None

    def _check(self, rc):
        _lib.check(rc, self.lib)

    # ------------------------------------------------------------ parameters
    def set_params(self, params):
        """One parameter block for the whole batch ([npar]) or one per trajectory ([B, npar]): batched goals, the
        batched form of set_references (control/models/acrobot.py:101-103)."""
        a = np.asarray(params.cpu() if torch.is_tensor(params) else params, dtype=np.float64)
        if a.ndim == 2:
            if a.shape[1] != self.npar:
                raise ValueError(f"model {self.model_id} takes {self.npar} parameters, got {a.shape[1]} per trajectory")
            self.params_np = np.ascontiguousarray(a)
            self.params_per_trajectory = True
            self.params = None
            self._params_key = None
            return
        p = torch.as_tensor(a.ravel())
        if p.numel() != self.npar:
            raise ValueError(f"model {self.model_id} takes {self.npar} parameters, got {p.numel()}")
        self.params = p.to(self.device)
        self.params_per_trajectory = False

    def _problem(self, B: int, stride: int) -> _lib.Problem:
        flags = self.flags
        par = self.params
        if self.params_per_trajectory:
            have = self.params_np.shape[0]
            if have < B:
                raise ValueError(f"{have} parameter blocks for {B} trajectories")
            key = (B, stride)
            if getattr(self, "_params_key", None) != key:
                if self.npar <= 8:      # strided like the other arrays: [npar, stride]
                    host = np.zeros((self.npar, stride))
                    host[:, :B] = self.params_np[:B].T
                else:                   # blocks one after the other: [B, npar]
                    host = self.params_np[:B]
                self._params_dev = torch.as_tensor(np.ascontiguousarray(host)).to(self.device)
                self._params_key = key
            par = self._params_dev
            flags |= _lib.FLAG_PARAMS_PER_TRAJECTORY
        return _lib.Problem(self.model_id, self.N, self.max_iter, self.max_ls, self.d_tol, B, stride,
                            C.c_void_p(par.data_ptr()), self.trace_len, flags, self.cost_N, 0)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def workspace_bytes(self, B: int, stride: Optional[int] = None) -> int:
        pr = self._problem(B, stride or B)
        return int(self.lib.ilqr_workspace_bytes(C.byref(pr)))

    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        return self._ws

    def alloc_outputs(self, stride: int) -> dict:
        dev = self.device
        o = dict(J0=torch.empty(stride, dtype=torch.float64, device=dev),
                 J=torch.empty(stride, dtype=torch.float64, device=dev),
                 dmax=torch.empty(stride, dtype=torch.float64, device=dev),
                 max_abs_q=torch.empty(stride, dtype=torch.float64, device=dev),
                 iters=torch.empty(stride, dtype=torch.int32, device=dev),
                 status=torch.empty(stride, dtype=torch.int32, device=dev),
                 regu_rounds=torch.empty(stride, dtype=torch.int32, device=dev),
                 ls_total=torch.empty(stride, dtype=torch.int32, device=dev),
                 regu_stages=torch.empty(stride, dtype=torch.int32, device=dev),
                 last_search=torch.empty(stride, dtype=torch.int32, device=dev),
                 replayed=torch.empty(stride, dtype=torch.int32, device=dev))
        if self.trace_len:
            T = self.trace_len
            o.update(ls_evals=torch.zeros((T, stride), dtype=torch.int32, device=dev),
                     accepted=torch.zeros((T, stride), dtype=torch.uint8, device=dev),
                     J_trace=torch.zeros((T, stride), dtype=torch.float64, device=dev),
                     deltaJ_trace=torch.zeros((T, stride), dtype=torch.float64, device=dev))
        return o

    # ------------------------------------------------------------ SoA level
    def solve_soa(self, x0: torch.Tensor, u: torch.Tensor, x: Optional[torch.Tensor] = None,
                  out: Optional[dict] = None, B: Optional[int] = None):
        """x0 [nx, stride], u [N-1, nu, stride] (overwritten), x [N, nx, stride] (written).
        Only enqueues work on the current stream.  Returns (x, u, out)."""
        stride = x0.shape[-1]
        B = stride if B is None else int(B)
        S = self.N - 1
        assert x0.shape == (self.nx, stride) and u.shape == (S, self.nu, stride), (x0.shape, u.shape)
        for t in (x0, u):
            assert t.is_contiguous() and t.dtype == torch.float64 and t.device == self.device
        if x is None:
            x = torch.empty((self.N, self.nx, stride), dtype=torch.float64, device=self.device)
        if out is None:
            out = self.alloc_outputs(stride)
        pr = self._problem(B, stride)
        nbytes = int(self.lib.ilqr_workspace_bytes(C.byref(pr)))
        ws = self._workspace(nbytes)
        oo = _lib.Outputs(*[_ptr(out.get(k)) for k, _ in _lib.Outputs._fields_])
        self._check(self.lib.ilqr_solve_batch(C.byref(pr), _ptr(x0), _ptr(u), _ptr(x), C.byref(oo), _ptr(ws),
                                             ws.numel(), self._stream()))
        return x, u, out

    def last_timing(self) -> dict:
        """Device time per kernel phase of this thread's last solve made with FLAG_TIMING:
        {phase: (milliseconds, launches)} plus "iterations"."""
        t = _lib.Timing()
        self._check(self.lib.ilqr_last_timing(C.byref(t)))
        d = {name: (float(t.ms[i]), int(t.launches[i])) for i, name in enumerate(_lib.PHASES)}
        d["iterations"] = int(t.iterations)
        return d

    # ------------------------------------------------------ trajectory-major
    def _to_dev(self, a):
        t = torch.as_tensor(np.asarray(a, dtype=np.float64)) if not torch.is_tensor(a) else a
        return t.to(device=self.device, dtype=torch.float64)

    def solve(self, x0, u) -> BatchResult:
        """x0 [B, nx], u [B, N-1, nu] -> BatchResult with x [B, N, nx], u [B, N-1, nu].  Inputs are not modified."""
        x0d = self._to_dev(x0).reshape(-1, self.nx)
        B = x0d.shape[0]
        ud = self._to_dev(u).reshape(B, self.N - 1, self.nu)
        x0s = x0d.t().contiguous()
        us = ud.permute(1, 2, 0).contiguous()
        xs, us, out = self.solve_soa(x0s, us)
        tr = {k: out[k].t() for k in ("ls_evals", "accepted", "J_trace", "deltaJ_trace") if k in out}
        return BatchResult(x=xs.permute(2, 0, 1).contiguous(), u=us.permute(2, 0, 1).contiguous(),
                           J0=out["J0"], J=out["J"], dmax=out["dmax"], max_abs_q=out["max_abs_q"],
                           iters=out["iters"], status=out["status"], regu_rounds=out["regu_rounds"],
                           ls_total=out["ls_total"], regu_stages=out["regu_stages"], last_search=out["last_search"],
                           replayed=out["replayed"], traces=tr)

    def rollout_cost(self, x0, u):
        """Model.rollout + cost_trj for a batch: returns (x [B, N, nx], J [B])."""
        x0d = self._to_dev(x0).reshape(-1, self.nx)
        B = x0d.shape[0]
        us = self._to_dev(u).reshape(B, self.N - 1, self.nu).permute(1, 2, 0).contiguous()
        x0s = x0d.t().contiguous()
        xs = torch.empty((self.N, self.nx, B), dtype=torch.float64, device=self.device)
        J = torch.empty(B, dtype=torch.float64, device=self.device)
        pr = self._problem(B, B)
        self._check(self.lib.ilqr_rollout_cost(C.byref(pr), _ptr(x0s), _ptr(us), _ptr(xs), _ptr(J), self._stream()))
        return xs.permute(2, 0, 1).contiguous(), J

    def model_eval(self, xs, us, what=("xn", "cs", "cf", "l_x", "l_u", "l_xx", "l_ux", "l_uu", "f_x", "f_u",
                                       "V_x", "V_xx")):
        """Evaluate the plant interface at B points: xs [B, nx], us [B, nu] -> dict of trajectory-major tensors."""
        xd = self._to_dev(xs).reshape(-1, self.nx)
        B = xd.shape[0]
        ud = self._to_dev(us).reshape(B, self.nu)
        n, m, nx = self.ndx, self.nu, self.nx
        shapes = dict(xn=(nx,), cs=(), cf=(), l_x=(n,), l_u=(m,), l_xx=(n, n), l_ux=(m, n), l_uu=(m, m),
                      f_x=(n, n), f_u=(n, m), V_x=(n,), V_xx=(n, n))
        bufs = {k: torch.empty((int(np.prod(shapes[k])) if shapes[k] else 1, B), dtype=torch.float64,
                               device=self.device) for k in what}
        order = ("xn", "cs", "cf", "l_x", "l_u", "l_xx", "l_ux", "l_uu", "f_x", "f_u", "V_x", "V_xx")
        pr = self._problem(B, B)
        xs_soa, us_soa = xd.t().contiguous(), ud.t().contiguous()   # named: must outlive the launch
        self._check(self.lib.ilqr_model_eval(C.byref(pr), _ptr(xs_soa), _ptr(us_soa),
                                            *[_ptr(bufs.get(k)) for k in order], self._stream()))
        return {k: bufs[k].t().reshape((B,) + shapes[k]) for k in what}

    def linearize_continuous(self, xs, us):
        """A = d xdot/dx [B, ndx, ndx] and B = d xdot/du [B, ndx, nu] of the continuous-time dynamics at B points
        (on the error state for the quaternion plant): the plants' _linearize_continuous."""
        xd = self._to_dev(xs).reshape(-1, self.nx)
        B = xd.shape[0]
        ud = self._to_dev(us).reshape(B, self.nu)
        n, m = self.ndx, self.nu
        Ab = torch.empty((n * n, B), dtype=torch.float64, device=self.device)
        Bb = torch.empty((n * m, B), dtype=torch.float64, device=self.device)
        pr = self._problem(B, B)
        xs_soa, us_soa = xd.t().contiguous(), ud.t().contiguous()
        self._check(self.lib.ilqr_linearize_continuous(C.byref(pr), _ptr(xs_soa), _ptr(us_soa), _ptr(Ab), _ptr(Bb),
                                                      self._stream()))
        return Ab.t().reshape(B, n, n), Bb.t().reshape(B, n, m)

    def mpc_advance(self, x0_soa: torch.Tensor, u_soa: torch.Tensor, B: Optional[int] = None):
        """In place on SoA buffers: x0 <- f(x0, u[0]); u[:-1] <- u[1:]."""
        stride = x0_soa.shape[-1]
        pr = self._problem(stride if B is None else B, stride)
        self._check(self.lib.ilqr_mpc_advance(C.byref(pr), _ptr(x0_soa), _ptr(u_soa), self._stream()))
