"""Kinematic car, drop-in for control/models/car.py:54-102."""
from __future__ import annotations

import numpy as np

from .. import _lib
from .model import Model


class CarModel(Model):
    """x = [x, y, heading, speed, steering angle], u = [acceleration, steering velocity];
    Euler step dt = 0.1; cost (sqrt(x^2+y^2+1e-6) - r)^2 + (v - v_target)^2 + 0.1 |u|^2."""
    _model_id = _lib.MODEL_CAR

    def __init__(self, r, v_target):
        self.nx = 5
        self.nu = 2
        self.ndx = 5
        self.r = r
        self.v_target = v_target

    def _param_block(self):
        """[r, v_target], or [B, 2] when either was given per trajectory (arrays of length B)."""
        r, v = np.asarray(self.r, dtype=np.float64), np.asarray(self.v_target, dtype=np.float64)
        if r.ndim == 0 and v.ndim == 0:
            return np.array([r, v], dtype=np.float64)
        r, v = np.broadcast_arrays(r.ravel() if r.ndim else r, v.ravel() if v.ndim else v)
        return np.column_stack([r, v])
