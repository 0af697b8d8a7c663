"""Linear-quadratic regulators around a goal state, the Drake-free form of the reference's terminal controllers.

The reference wraps Drake's DiscreteTimeLinearQuadraticRegulator / LinearQuadraticRegulator in LeafSystems
(control/QuadrotorControllers.py:84-128, :197-233, control/AcrobotControllers.py:68-115): linearise the plant at
the goal, solve the Riccati equation, output u = uref - K * CalcDifferentialState(x, goal).  Here the
linearisation comes from the plant's device code (`Model._linearize_discrete`: f_x, f_u of the integrator
step; `Model._linearize_continuous`: the Jacobians of xdot; both already on the 12-dimensional error state for
the quaternion plant, Quadrotor_Quat.py:231-246) and the Riccati equations are SciPy's.  These are a few 12x12 solves on the host; nothing here is a hot path.
"""
from __future__ import annotations

import numpy as np
from scipy.linalg import solve_continuous_are, solve_discrete_are


def dlqr(A, B, Q, R):
    """Discrete-time LQR: K, S with u = -K x minimising sum x'Qx + u'Ru (Drake's DiscreteTimeLinearQuadraticRegulator)."""
    A, B, Q, R = (np.atleast_2d(np.asarray(M, dtype=np.float64)) for M in (A, B, Q, R))
    S = solve_discrete_are(A, B, Q, R)
    K = np.linalg.solve(R + B.T @ S @ B, B.T @ S @ A)
    return K, S


def lqr(A, B, Q, R):
    """Continuous-time LQR: K, S with u = -K x (Drake's LinearQuadraticRegulator)."""
    A, B, Q, R = (np.atleast_2d(np.asarray(M, dtype=np.float64)) for M in (A, B, Q, R))
    S = solve_continuous_are(A, B, Q, R)
    return np.linalg.solve(R, B.T @ S), S


class LQRController:
    """u = uref - K * CalcDifferentialState(current_state, goal_state), K from the plant linearised at the goal.

    One class for QuadrotorEALQRController, QuadrotorQuatLQRController and AcrobotLQRController of the reference
    (their DoCalcVectorOutput bodies); the gain is cached per goal state.  mode "discrete": the integrator step's
    Jacobians and the discrete Riccati equation; any other mode: the Jacobians of xdot and the continuous one
    (QuadrotorControllers.py:118-124, :223-229).
    """

    def __init__(self, model, Q=None, R=None, uref=None, mode="discrete"):
        self.mode = mode
        self.model = model
        self.nx, self.nu, self.ndx = model.get_dims()
        self.Q = np.asarray(model.Q if Q is None else Q, dtype=np.float64)
        self.R = np.asarray(model.R if R is None else R, dtype=np.float64)
        self.uref = np.zeros(self.nu) if uref is None else np.asarray(uref, dtype=np.float64)
        self._goal = None
        self.K = None

    def gain(self, goal_state):
        goal_state = np.asarray(goal_state, dtype=np.float64)
        if self._goal is None or not np.array_equal(goal_state, self._goal):
            if self.mode == "discrete":
                A, B = self.model._linearize_discrete(goal_state, self.uref)
                self.K, self.S = dlqr(A, B, self.Q, self.R)
            else:
                A, B = self.model._linearize_continuous(goal_state, self.uref)
                self.K, self.S = lqr(A, B, self.Q, self.R)
            self._goal = goal_state.copy()
        return self.K

    def control(self, current_state, goal_state):
        K = self.gain(goal_state)
        dx = self.model.CalcDifferentialState(np.asarray(current_state, dtype=np.float64), goal_state)
        return np.asarray(self.uref - K @ np.asarray(dx, dtype=np.float64).ravel()).ravel()
