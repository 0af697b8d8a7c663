"""Shared constructor/state of the RK4 plants with quadratic cost
(control/models/acrobot.py:6-19,101-103; control/models/Quadrotor_EA.py:23-43,157-159)."""
from __future__ import annotations

import numpy as np

from .model import Model


class QuadraticCostModel(Model):
    _dims = (0, 0, 0)

    def __init__(self, Q, Qf, R, dt=0.01, N=20, integration="RK4", **kwargs):
        if integration != "RK4":
            raise NotImplementedError("only RK4 integration is generated for the GPU")
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

    def _weights_block(self):
        return np.concatenate([[self.dt], self.Q.ravel(), self.Qf.ravel(), self.R.ravel()])

    def _param_block(self):
        head = self._weights_block()
        xr, ur = self._references()
        if xr.ndim == 2:
            return np.hstack([np.broadcast_to(head, (xr.shape[0], head.size)), xr, ur])
        return np.concatenate([head, xr, ur])
