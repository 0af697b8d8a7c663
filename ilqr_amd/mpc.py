"""Batched receding-horizon driver: B closed loops advanced in lock step on the GPU.

What the reference's controller wrappers do for one vehicle around `run_ilqr`
(control/QuadrotorControllers.py:60-78 and :173-191, control/AcrobotControllers.py:48-66:
shift the previous controls one step forward, re-solve from the current state,
apply the first control) and what Driving.ipynb cell-2 does for the car (re-solve
from x_trj[1] with the controls carried over unshifted), for a whole batch, with
the state and the controls staying in device memory between solves.

    mpc = BatchMPC(BatchSolver(model_id, params, N, max_iter, max_ls, d_tol))
    log = mpc.run(x0, u_init, steps=50)          # x0 [B, nx], u_init [B, N-1, nu]

Per step:  solve (u optimised in place)  ->  apply u[0]  ->  next state.
  shift=True   next state = plant(x0, u[0]) and u[:-1] <- u[1:]      (ilqr_mpc_advance)
  shift=False  next state = x_trj[1] of the solve, u kept as it is   (Driving.ipynb)
With `reseed_std` a loop whose plan is all zeros gets a random one before its solve, per trajectory
(QuadrotorControllers.py:67-70: `if not utraj.any(): utraj[:] = randn(N-1, nu) * 2.0`).
"""
from __future__ import annotations

import torch

from .batch import BatchSolver


class BatchMPC:
    def __init__(self, solver: BatchSolver, shift: bool = True, reseed_std: float | None = None,
                 generator: torch.Generator | None = None):
        """reseed_std: standard deviation of the plan an all-zero plan is replaced with (the reference's wrappers
        use 2.0; None: plans are used as they are); generator: a generator on the solver's device for those draws."""
        self.solver = solver
        self.shift = bool(shift)
        self.reseed_std = reseed_std
        self.generator = generator

    def _reseed(self, us):
        """us [S, nu, B]: trajectories whose plan is zero everywhere get N(0, reseed_std^2) controls, the others keep
        theirs.  No host round trip: the draw is made for the whole batch and kept where the plan was zero."""
        dead = ~(us != 0).any(dim=0).any(dim=0)                                   # [B]
        noise = torch.randn(us.shape, dtype=us.dtype, device=us.device, generator=self.generator) * self.reseed_std
        us.copy_(torch.where(dead[None, None, :], noise, us))

    def run(self, x0, u_init, steps: int, keep_plans: bool = False) -> dict:
        """Returns trajectory-major logs on the solver's device:
        x [steps+1, B, nx] states visited, u [steps, B, nu] controls applied, J / iters / status [steps, B];
        with keep_plans also the full plans x_plan [steps, B, N, nx], u_plan [steps, B, N-1, nu]."""
        s = self.solver
        x0d = s._to_dev(x0).reshape(-1, s.nx)
        B = x0d.shape[0]
        x0s = x0d.t().contiguous()                                           # [nx, B]
        us = s._to_dev(u_init).reshape(B, s.N - 1, s.nu).permute(1, 2, 0).contiguous()   # [S, nu, B]
        xs = torch.empty((s.N, s.nx, B), dtype=torch.float64, device=s.device)
        out = s.alloc_outputs(B)
        dev = s.device
        log = dict(x=torch.empty((steps + 1, B, s.nx), dtype=torch.float64, device=dev),
                   u=torch.empty((steps, B, s.nu), dtype=torch.float64, device=dev),
                   J=torch.empty((steps, B), dtype=torch.float64, device=dev),
                   iters=torch.empty((steps, B), dtype=torch.int32, device=dev),
                   status=torch.empty((steps, B), dtype=torch.int32, device=dev))
        if keep_plans:
            log["x_plan"] = torch.empty((steps, B, s.N, s.nx), dtype=torch.float64, device=dev)
            log["u_plan"] = torch.empty((steps, B, s.N - 1, s.nu), dtype=torch.float64, device=dev)
        for t in range(steps):
            log["x"][t] = x0s.t()
            if self.reseed_std is not None:
                self._reseed(us)
            s.solve_soa(x0s, us, xs, out)
            log["u"][t] = us[0].t()
            log["J"][t] = out["J"]
            log["iters"][t] = out["iters"]
            log["status"][t] = out["status"]
            if keep_plans:
                log["x_plan"][t] = xs.permute(2, 0, 1)
                log["u_plan"][t] = us.permute(2, 0, 1)
            if self.shift:
                s.mpc_advance(x0s, us)
            else:
                x0s.copy_(xs[1])
        log["x"][steps] = x0s.t()
        return log
