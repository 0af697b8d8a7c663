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
        self.xref = xref
        self.uref = uref

    def _param_block(self):
        return np.concatenate([[self.dt], self.Q.ravel(), self.Qf.ravel(), self.R.ravel(),
                               np.asarray(self.xref, dtype=np.float64).ravel(),
                               np.asarray(self.uref, dtype=np.float64).ravel()])
