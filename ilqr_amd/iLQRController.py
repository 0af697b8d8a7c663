"""Drop-in for the reference solver class (control/iLQRController.py:7-147).

    from ilqr_amd.iLQRController import iLQR
    ctl = iLQR(plant, N=40, max_iter=50, max_linesearch_iters=20, d_tol=1e-3)
    x_trj, u_trj = ctl.run_ilqr(x0, u_trj)

`run_ilqr` keeps the reference's contract for a single trajectory (x0 [nx],
u_trj [N-1, nu]): the returned `u_trj` is the caller's array, optimised in
place (iLQRController.py:72, :84), `x_trj` is a new [N, nx] array, and
numpy.linalg.LinAlgError is raised where NumPy would have raised it (non-finite
Q handed to eigvals, :110; singular Q_uu handed to inv, :126).

It additionally accepts a batch: x0 [B, nx] with u_trj [B, N-1, nu] solves B
independent problems in one launch and returns x [B, N, nx] and the (in place
updated) u.  NumPy in, NumPy out; torch CUDA tensors in, torch tensors out.
Per-trajectory diagnostics of the last call are in `self.info`
(BatchResult: J0, J, iters, status, ls_total, regu_rounds, dmax, max_abs_q).

`eigvals="lapack"` (keyword, default "ldl") makes every `np.linalg.eigvals(Q) < 0` test of the regularisation loop
(iLQRController.py:110) LAPACK's own, operation for operation (ILQR_FLAG_LAPACK_EIGVALS): ill-conditioned solves
then regularise, iterate and exit like the reference's; solves the reference reproduces itself are the same under
either rule, and the default is far faster.

All arithmetic runs in the HIP kernels behind libilqr_hip.so; there is no host
implementation of the solver in this package.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib
from .batch import BatchResult, BatchSolver
from .models.model import Model


class iLQR:
    def __init__(self, model: Model, N: int, max_iter: int, max_linesearch_iters: int = 20, d_tol: float = 1e-3,
                 **kwargs):
        # unknown keyword arguments are accepted and ignored, as in iLQRController.py:14
        self.model = model
        self.nx, self.nu, self.ndx = self.model.get_dims()
        self.N = N
        self.max_iter = max_iter
        self.max_linesearch_iters = max_linesearch_iters
        self.d_tol = d_tol

        self.num_ls_success = 0
        self.num_ls_fails = 0
        self.num_iters = 0

        self.info: BatchResult | None = None
        self.device = kwargs.get("device", "cuda:0")
        self.trace_len = int(kwargs.get("trace_len", 0))
        self.eigvals = str(kwargs.get("eigvals", "ldl"))
        if self.eigvals not in ("ldl", "lapack"):
            raise ValueError('eigvals must be "ldl" or "lapack"')
        self._solver = None
        self._solver_key = None
        if getattr(model, "_model_id", None) is None:
            raise TypeError(
                f"{type(model).__name__} has no generated device code: use CarModel, Acrobot2DModel, "
                "QuadrotorEAModel or QuadrotorQuatModel from ilqr_amd.models, or define the plant through "
                "ilqr_amd.models.user.UserModel")

    def _cost_N(self) -> int:
        """Knot points cost_trj sums over.  The acrobot's and the quadrotors' cost_trj sum the stage costs of
        the MODEL's horizon and the final cost of the last state (acrobot.py:139, Quadrotor_EA.py:199,
        Quadrotor_Quat.py:212), whatever horizon the solver runs; Model.cost_trj (model.py:31-37: car, plants
        written by a user) sums over the trajectory it is given."""
        if not getattr(self.model, "_cost_sums_model_horizon", False):
            return 0
        mN = int(self.model.N)
        if mN > self.N:
            raise IndexError(f"index {self.N - 1} is out of bounds: the model's cost_trj sums {mN - 1} stages of a "
                             f"trajectory of {self.N - 1} (model.N = {mN}, solver N = {self.N})")
        if mN < 1:
            raise ValueError("model.N must be >= 1")
        return 0 if mN == self.N else mN

    # ------------------------------------------------------------------
    def _get_solver(self) -> BatchSolver:
        par = np.ascontiguousarray(self.model._param_block(), dtype=np.float64)
        cost_N = self._cost_N()
        flags = _lib.FLAG_LAPACK_EIGVALS if self.eigvals == "lapack" else 0
        key = (par.tobytes(), self.N, self.max_iter, self.max_linesearch_iters, float(self.d_tol), self.trace_len, cost_N,
               flags)
        if self._solver is None or key != self._solver_key:
            self._solver = BatchSolver(self.model._model_id, par, self.N, self.max_iter, self.max_linesearch_iters,
                                       self.d_tol, device=self.device, trace_len=self.trace_len, flags=flags,
                                       library=getattr(self.model, "_library", None), cost_N=cost_N)
            self._solver_key = key
        return self._solver

    def run_ilqr(self, x0, u_trj):
        solver = self._get_solver()
        self.num_iters += 1                                     # counts calls (iLQRController.py:40)
        is_torch = torch.is_tensor(u_trj)
        x0a = x0 if torch.is_tensor(x0) else np.asarray(x0, dtype=np.float64)
        single = x0a.ndim == 1
        S = self.N - 1
        if single:
            if tuple(u_trj.shape) != (S, self.nu):
                raise ValueError(f"u_trj must have shape ({S}, {self.nu}), got {tuple(u_trj.shape)}")
            res = solver.solve(self.model._lib_x(x0a[None]), self.model._lib_u(u_trj[None]))
        else:
            if tuple(u_trj.shape) != (x0a.shape[0], S, self.nu):
                raise ValueError(f"u_trj must have shape (B, {S}, {self.nu}), got {tuple(u_trj.shape)}")
            res = solver.solve(self.model._lib_x(x0a), self.model._lib_u(u_trj))
        res.x = self.model._own("x", res.x)       # (the library may state the plant wider than it was written)
        res.u = self.model._own("u", res.u)
        self.info = res
        status = res.status.cpu().numpy()
        # reference counters: reset at every iteration that reaches the line search (:57-58), so after the call
        # they describe the last such iteration; a call none of whose iterations got there leaves them alone
        last = res.last_search.cpu().numpy()
        if single:
            if last[0] != 0:
                self.num_ls_success, self.num_ls_fails = int(last[0] == 1), int(last[0] == 2)
        else:
            old_s = np.broadcast_to(np.asarray(self.num_ls_success), last.shape)
            old_f = np.broadcast_to(np.asarray(self.num_ls_fails), last.shape)
            self.num_ls_success = np.where(last == 0, old_s, last == 1).astype(int)
            self.num_ls_fails = np.where(last == 0, old_f, last == 2).astype(int)
        if single and status[0] == _lib.EXIT_LINALG:
            raise np.linalg.LinAlgError("Array must not contain infs or NaNs, or Q_uu is singular "
                                        "(iLQRController.py:110, :126)")
        if is_torch:
            u_out = res.u[0] if single else res.u
            u_trj.copy_(u_out.to(u_trj.device, u_trj.dtype))
            return (res.x[0] if single else res.x), u_trj
        u_np = res.u.cpu().numpy()
        u_trj[...] = u_np[0] if single else u_np                # in place, same object returned (:72, :84)
        x_np = res.x.cpu().numpy()
        return (x_np[0] if single else x_np), u_trj
