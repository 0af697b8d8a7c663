"""Euler-angle quadrotor, drop-in for control/models/Quadrotor_EA.py:22-224."""
from .. import _lib
from ._quadratic import QuadraticCostModel


class QuadrotorEAModel(QuadraticCostModel):
    """x = [p(3), yaw, pitch, roll, v(3), yaw_rate, pitch_rate, roll_rate], u = 4 rotor commands."""
    _model_id = _lib.MODEL_QUAD_EA
    _dims = (12, 4, 12)
