"""Quaternion quadrotor with a 12-dim error state, drop-in for
control/models/Quadrotor_Quat.py:31-275 (Q, Qf [12 x 12] and R [4 x 4] any matrices, as there)."""
from __future__ import annotations

import numpy as np

from .. import _lib
from ._quadratic import QuadraticCostModel


class QuadrotorQuatModel(QuadraticCostModel):
    """x = [qw, qx, qy, qz, p(3), w(3), v(3)], u = 4 rotor commands; cost on e = G(x)'(x - xref)."""
    _model_id = _lib.MODEL_QUAD_QUAT
    _dims = (13, 4, 12)

    def __init__(self, Q, Qf, R, dt=0.01, N=20, integration="RK4", **kwargs):
        super().__init__(Q, Qf, R, dt, N, integration, **kwargs)

    def CalcDifferentialState(self, state, ref_state):
        """Quadrotor_Quat.py:218-228: G(state)' (state - ref_state), 12-vector."""
        state = np.asarray(state, dtype=np.float64).ravel()
        q = state[:4] / np.linalg.norm(state[:4])
        w, x, y, z = q
        G = np.array([[-x, -y, -z], [w, -z, y], [z, w, -x], [-y, x, w]])
        e = state - np.asarray(ref_state, dtype=np.float64).ravel()
        return np.concatenate([G.T @ e[:4], e[4:]])
