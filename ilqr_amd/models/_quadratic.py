"""Shared constructor/state of the plants with quadratic cost and an RK4 or Euler integrator
(control/models/acrobot.py:6-19,76-86,101-103; control/models/Quadrotor_EA.py:23-43,132-142,157-159)."""
from __future__ import annotations

import numpy as np

from .model import Model


class QuadraticCostModel(Model):
    _dims = (0, 0, 0)
    # cost_trj sums the stage costs of the MODEL's horizon, then the final cost of the trajectory's last state
    # (acrobot.py:129-142, Quadrotor_EA.py:189-204, Quadrotor_Quat.py:201-216)
    _cost_sums_model_horizon = True

    def __init__(self, Q, Qf, R, dt=0.01, N=20, integration="RK4", **kwargs):
        # acrobot.py:76-86: "RK4" is RK4, anything else one explicit Euler step
        self.integration = integration
        self.Q = np.asarray(Q, dtype=np.float64)
        self.Qf = np.asarray(Qf, dtype=np.float64)
        self.R = np.asarray(R, dtype=np.float64)
        self.dt = dt
        self.N = N
        self.nx, self.nu, self.ndx = self._dims
        self.xref = np.zeros(self.nx, dtype=np.float64)
        self.uref = np.zeros(self.nu, dtype=np.float64)

    def set_references(self, xref, uref):
        """acrobot.py:101-103.  xref [nx] and uref [nu] for every trajectory, or xref [B, nx] / uref [B, nu]:
        references per trajectory of the batches solved from now on (batched goals)."""
        self.xref = xref
        self.uref = uref

    def _references(self):
        """(xref, uref) as arrays; both [B, .] if either was given per trajectory."""
        xr, ur = np.asarray(self.xref, dtype=np.float64), np.asarray(self.uref, dtype=np.float64)
        if xr.ndim == 2 or ur.ndim == 2:
            B = xr.shape[0] if xr.ndim == 2 else ur.shape[0]
            xr = np.broadcast_to(xr, (B, self.nx)) if xr.ndim < 2 else xr
            ur = np.broadcast_to(ur.reshape(-1, self.nu) if ur.ndim == 2 else ur, (B, self.nu))
            return xr, ur
        return xr.ravel(), ur.ravel()

    def cost_trj(self, x_trj, u_trj):
        x_trj = np.asarray(x_trj, dtype=np.float64)
        u_trj = np.asarray(u_trj, dtype=np.float64)
        S = int(self.N) - 1
        if S > u_trj.shape[0]:
            raise IndexError(f"index {u_trj.shape[0]} is out of bounds for axis 0 with size {u_trj.shape[0]}")
        total = 0.0
        if S > 0:
            for c in self._evaluator().model_eval(x_trj[:S], u_trj[:S], ("cs",))["cs"].cpu().numpy():
                total += float(c)
        total += self._cost_final(x_trj[-1])
        return total

    def _weights_block(self):
        return np.concatenate([[self.dt], self.Q.ravel(), self.Qf.ravel(), self.R.ravel()])

    def _param_block(self):
        head = self._weights_block()
        tail = np.array([0.0 if self.integration == "RK4" else 1.0])
        xr, ur = self._references()
        if xr.ndim == 2:
            return np.hstack([np.broadcast_to(head, (xr.shape[0], head.size)), xr, ur,
                              np.full((xr.shape[0], 1), tail[0])])
        return np.concatenate([head, xr, ur, tail])
