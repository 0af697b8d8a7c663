"""Acrobot, drop-in for control/models/acrobot.py:5-166."""
from .. import _lib
from ._quadratic import QuadraticCostModel


class Acrobot2DModel(QuadraticCostModel):
    """x = [th1, th2, th1_dot, th2_dot], u = [elbow torque]; RK4 step `dt`;
    cost 0.5 dx'Q dx + 0.5 du'R du, final 0.5 dx'Qf dx about (xref, uref)."""
    _model_id = _lib.MODEL_ACROBOT
    _dims = (4, 1, 4)

    def __init__(self, Q, Qf, R, dt, N, integration="RK4", **kwargs):
        super().__init__(Q, Qf, R, dt, N, integration, **kwargs)
