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


class Skiff(UserModel):
    """A planar vehicle whose heading is a unit complex number (c, s) instead of an angle -- the 2-D relative of the
    reference's quaternion quadrotor (control/models/Quadrotor_Quat.py): the state has five components, the error
    state four, and the plant states its own CalcDifferentialState, stage, final and Q_terms through the map G(x)
    from error state to state, as Quadrotor_Quat.py:218-275 does.

    x = [px, py, c, s, speed], u = [acceleration, turn rate], error state [dpx, dpy, dheading, dspeed],
    p = [goal_x, goal_y, goal_c, goal_s, obstacle_x, obstacle_y].  (c, s) is normalised wherever it is used, like the
    quaternion in maths/casadi_quaternions.py:8.
    """
    DT = 0.1
    Q = np.diag([1.0, 1.0, 0.5, 0.1])
    QF = np.diag([60.0, 60.0, 20.0, 5.0])

    def __init__(self, goal=(2.5, 1.0, 1.0, 0.0), obstacle=(1.2, 0.3)):
        super().__init__(nx=5, nu=2, ndx=4, params=[*goal, *obstacle], name="skiff")

    @staticmethod
    def _heading(x):
        n = m.sqrt(x[2] * x[2] + x[3] * x[3])
        return x[2] / n, x[3] / n

    def G(self, x):
        """d state / d error state, 5 x 4: the heading moves along the tangent (-s, c) of the unit circle."""
        c, s = self._heading(x)
        return m.array([1.0, 0.0, 0.0, 0.0,
                        0.0, 1.0, 0.0, 0.0,
                        0.0, 0.0, -s, 0.0,
                        0.0, 0.0, c, 0.0,
                        0.0, 0.0, 0.0, 1.0]).reshape(5, 4)

    def discrete_dynamics(self, x, u):
        h = self.DT
        c, s = self._heading(x)
        cw, sw = m.cos(h * u[1]), m.sin(h * u[1])
        return m.array([x[0] + h * x[4] * c, x[1] + h * x[4] * s, c * cw - s * sw, s * cw + c * sw, x[4] + h * u[0]])

    def _error(self, x):
        ref = m.array([self.p[0], self.p[1], self.p[2], self.p[3], 0.0 * self.p[0]])
        return self.G(x).T @ (x - ref)

    def _bump(self, x):
        d2 = (x[0] - self.p[4]) ** 2 + (x[1] - self.p[5]) ** 2
        return 3.0 * m.exp(-d2 / 0.2)

    def _cost_stage(self, x, u):
        e = self._error(x)
        return 0.5 * (e @ (self.Q @ e)) + self._bump(x) + 0.5 * (0.2 * u[0] ** 2 + 0.1 * u[1] ** 2)

    def _cost_final(self, x):
        e = self._error(x)
        return 0.5 * (e @ (self.QF @ e))

    # ---- the overrides of Quadrotor_Quat.py:218-275
    def CalcDifferentialState(self, state, ref_state):
        return self.G(state).T @ (state - ref_state)

    def stage(self, x, u):
        l_x, l_u, l_xx, l_ux, l_uu, A, B = self.derivatives(x, u)
        G = self.G(x)
        return l_x, l_u, l_xx, l_ux, l_uu, G.T @ A @ G, G.T @ B

    def final(self, x):
        lf_x, lf_xx = self.final_derivatives(x)
        G = self.G(x)
        return G.T @ lf_x, G.T @ lf_xx @ G

    def Q_terms(self, x, l_x, l_u, l_xx, l_ux, l_uu, f_x, f_u, V_x, V_xx):
        G = self.G(x)
        return (G.T @ l_x + f_x.T @ V_x, l_u + f_u.T @ V_x, G.T @ l_xx @ G + f_x.T @ V_xx @ f_x,
                l_ux @ G + f_u.T @ V_xx @ f_x, l_uu + f_u.T @ V_xx @ f_u)


SKIFF_SP = dict(N=50, max_iter=50, max_ls=20, d_tol=1e-3)


def skiff_inputs(B, seed):
    rng = np.random.default_rng(seed)
    th = rng.uniform(-np.pi, np.pi, B)
    x0 = np.column_stack([rng.uniform(-1, 1, B), rng.uniform(-1, 1, B), np.cos(th), np.sin(th), rng.uniform(0, 1.0, B)])
    u0 = 0.3 * rng.standard_normal((B, SKIFF_SP["N"] - 1, 2))
    return x0, u0
