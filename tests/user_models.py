"""A plant written the way a user of the solver would write one (used by the user-model tests and goldens)."""
import numpy as np

from ilqr_amd import sym as m
from ilqr_amd.models.user import UserModel


class Rover(UserModel):
    """Unicycle with a speed state driving to a goal past a soft circular obstacle.

    x = [px, py, heading, speed], u = [acceleration, turn rate], p = [goal_x, goal_y, obstacle_x, obstacle_y].
    The obstacle bump has negative curvature, so the backward pass has to regularise (iLQRController.py:110-113).
    """
    DT = 0.1

    def __init__(self, goal=(3.0, 1.0), obstacle=(1.5, 0.4)):
        super().__init__(nx=4, nu=2, params=[*goal, *obstacle], name="rover")

    def discrete_dynamics(self, x, u):
        h = self.DT
        return m.array([x[0] + h * x[3] * m.cos(x[2]),
                        x[1] + h * x[3] * m.sin(x[2]),
                        x[2] + h * u[1],
                        x[3] + h * u[0]])

    def _bump(self, x):
        d2 = (x[0] - self.p[2]) ** 2 + (x[1] - self.p[3]) ** 2
        return 4.0 * m.exp(-d2 / 0.18)

    def _cost_stage(self, x, u):
        goal = (x[0] - self.p[0]) ** 2 + (x[1] - self.p[1]) ** 2
        return 0.05 * goal + self._bump(x) + 0.1 * u[0] ** 2 + 0.05 * u[1] ** 2 + 0.01 * x[3] ** 2

    def _cost_final(self, x):
        goal = (x[0] - self.p[0]) ** 2 + (x[1] - self.p[1]) ** 2
        return 50.0 * goal + 5.0 * x[3] ** 2


ROVER_SP = dict(N=60, max_iter=60, max_ls=20, d_tol=1e-3)


def rover_inputs(B, seed):
    rng = np.random.default_rng(seed)
    x0 = np.column_stack([rng.uniform(-1, 1, B), rng.uniform(-1, 1, B), rng.uniform(-np.pi, np.pi, B),
                          rng.uniform(0, 1.5, B)])
    u0 = 0.3 * rng.standard_normal((B, ROVER_SP["N"] - 1, 2))
    return x0, u0


class Tug(UserModel):
    """Planar rigid body with two body-frame thrusters and a reaction wheel, pushing to a goal pose past a soft keep-out
    disc: nx + nu = 9, beyond the one-trajectory-per-lane kernels.

    x = [px, py, heading, vx, vy, turn rate], u = [surge force, sway force, torque], p = [goal_x, goal_y, disc_x, disc_y].
    """
    DT = 0.1

    def __init__(self, goal=(2.0, 1.0), disc=(1.0, 0.3)):
        super().__init__(nx=6, nu=3, params=[*goal, *disc], name="tug")

    def discrete_dynamics(self, x, u):
        h = self.DT
        c, s = m.cos(x[2]), m.sin(x[2])
        return m.array([x[0] + h * x[3], x[1] + h * x[4], x[2] + h * x[5],
                        x[3] + h * (c * u[0] - s * u[1] - 0.2 * x[3]),
                        x[4] + h * (s * u[0] + c * u[1] - 0.2 * x[4]),
                        x[5] + h * (2.0 * u[2] - 0.3 * x[5])])

    def _cost_stage(self, x, u):
        goal = (x[0] - self.p[0]) ** 2 + (x[1] - self.p[1]) ** 2
        d2 = (x[0] - self.p[2]) ** 2 + (x[1] - self.p[3]) ** 2
        return (0.05 * goal + 3.0 * m.exp(-d2 / 0.2) + 0.02 * (x[3] ** 2 + x[4] ** 2 + x[5] ** 2) + 0.02 * x[2] ** 2
                + 0.1 * u[0] ** 2 + 0.1 * u[1] ** 2 + 0.2 * u[2] ** 2)

    def _cost_final(self, x):
        goal = (x[0] - self.p[0]) ** 2 + (x[1] - self.p[1]) ** 2
        return 40.0 * goal + 10.0 * x[2] ** 2 + 5.0 * (x[3] ** 2 + x[4] ** 2 + x[5] ** 2)


TUG_SP = dict(N=50, max_iter=40, max_ls=20, d_tol=1e-3)


def tug_inputs(B, seed):
    rng = np.random.default_rng(seed)
    x0 = np.column_stack([rng.uniform(-1, 0.5, B), rng.uniform(-1, 1, B), rng.uniform(-1.5, 1.5, B),
                          0.3 * rng.standard_normal((B, 3))])
    u0 = 0.2 * rng.standard_normal((B, TUG_SP["N"] - 1, 3))
    return x0, u0


class Compass(UserModel):
    """A unicycle whose heading state is kept in (-pi, pi] by the dynamics, with its own CalcDifferentialState
    (control/models/model.py:22-23 overridden): the heading difference is taken the short way round, so a trajectory
    that crosses the cut does not see a 2 pi jump in the feedback term K (x_new - x) of the forward pass.

    x = [px, py, heading], u = [speed, turn rate], p = [goal_x, goal_y].
    """
    DT = 0.1

    def __init__(self, goal=(-2.0, 0.5)):
        super().__init__(nx=3, nu=2, params=list(goal), name="compass")

    @staticmethod
    def _wrap(a):
        return m.atan2(m.sin(a), m.cos(a))

    def discrete_dynamics(self, x, u):
        h = self.DT
        return m.array([x[0] + h * u[0] * m.cos(x[2]), x[1] + h * u[0] * m.sin(x[2]), self._wrap(x[2] + h * u[1])])

    def CalcDifferentialState(self, state, ref_state):
        return m.array([state[0] - ref_state[0], state[1] - ref_state[1], self._wrap(state[2] - ref_state[2])])

    def _cost_stage(self, x, u):
        goal = (x[0] - self.p[0]) ** 2 + (x[1] - self.p[1]) ** 2
        return 0.1 * goal + 0.05 * u[0] ** 2 + 0.05 * u[1] ** 2

    def _cost_final(self, x):
        goal = (x[0] - self.p[0]) ** 2 + (x[1] - self.p[1]) ** 2
        return 30.0 * goal


COMPASS_SP = dict(N=40, max_iter=40, max_ls=20, d_tol=1e-3)


def compass_inputs(B, seed):
    """Starts heading roughly away from the goal (which lies towards -x): the turn takes the heading across +-pi."""
    rng = np.random.default_rng(seed)
    x0 = np.column_stack([rng.uniform(0, 1, B), rng.uniform(-1, 1, B), rng.uniform(2.2, 3.1, B) * rng.choice([-1, 1], B)])
    u0 = np.column_stack([np.full(B, 1.0), rng.choice([-1.0, 1.0], B) * 1.5])[:, None, :] * np.ones((1, COMPASS_SP["N"] - 1, 1)) \
        + 0.1 * rng.standard_normal((B, COMPASS_SP["N"] - 1, 2))
    return x0, u0
