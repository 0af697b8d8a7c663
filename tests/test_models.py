"""Plant-model checks on the CPU: the checker's dynamics/cost/derivatives against
(a) the NumPy restatement of the reference model files, (b) central finite differences,
(c) the known answers the reference's notebooks hold."""
import numpy as np
import pytest

from oracle import models_numpy as mn


class _Base:
    def get_dims(self):
        return self.nx, self.nu, self.ndx

    def CalcDifferentialState(self, s, r):
        return s - r


MODELS = mn.make_models(_Base)


def _quad_weights():
    Q = np.diag([10.0] * 6 + [1.0] * 6)
    Qf = np.diag([500.0] * 3 + [100.0] * 3 + [20.0] * 6)
    R = np.diag([0.02] * 4)
    return Q, Qf, R


def _setup(kind, oracle, rng):
    if kind == "car":
        plant = MODELS["CarModel"](2.0, 2.0)
        par = oracle.pack_car(2.0, 2.0)
        x = np.array([rng.uniform(-4, 4), rng.uniform(-4, 4), rng.uniform(-3, 3), rng.uniform(0, 3), rng.uniform(-.3, .3)])
        u = rng.standard_normal(2)
    elif kind == "acrobot":
        Q, Qf, R = np.diag([1, 1, .1, .1]), 100.0 * np.eye(4), np.array([[0.01]])
        xref, uref = np.array([np.pi, 0, 0, 0.0]), np.zeros(1)
        plant = MODELS["Acrobot2DModel"](Q, Qf, R, 0.05, 10)
        plant.set_references(xref, uref)
        par = oracle.pack_quadratic(0.05, Q, Qf, R, xref, uref)
        x = rng.uniform(-2, 2, 4); u = rng.standard_normal(1)
    elif kind == "quad_ea":
        Q, Qf, R = _quad_weights()
        xref = np.zeros(12); xref[2] = 1.0; uref = np.zeros(4)
        plant = MODELS["QuadrotorEAModel"](Q, Qf, R, 0.01, 10)
        plant.set_references(xref, uref)
        par = oracle.pack_quadratic(0.01, Q, Qf, R, xref, uref)
        x = 0.5 * rng.standard_normal(12); u = 1.9 + rng.standard_normal(4)
    else:
        Q, Qf, R = _quad_weights()
        xref = np.zeros(13); xref[0] = 1.0; xref[6] = 1.0; uref = np.zeros(4)
        plant = MODELS["QuadrotorQuatModel"](Q, Qf, R, 0.01, 10)
        plant.set_references(xref, uref)
        par = oracle.pack_quat(0.01, Q, Qf, R, xref, uref)
        x = 0.5 * rng.standard_normal(13); x[:4] = rng.standard_normal(4); x[:4] /= np.linalg.norm(x[:4]) * rng.uniform(0.8, 1.2)
        u = 1.9 + rng.standard_normal(4)
    return plant, par, x, u


def _fd_jac(f, z, h=1e-6):
    z = np.asarray(z, dtype=np.float64)
    f0 = np.atleast_1d(f(z))
    J = np.zeros((f0.size, z.size))
    for i in range(z.size):
        e = np.zeros_like(z); e[i] = h
        J[:, i] = (np.atleast_1d(f(z + e)) - np.atleast_1d(f(z - e))) / (2 * h)
    return J


@pytest.mark.parametrize("kind", ["car", "acrobot", "quad_ea", "quad_quat"])
def test_checker_models_vs_numpy_restatement_and_fd(oracle, kind):
    rng = np.random.default_rng(5)
    for _ in range(4):
        plant, par, x, u = _setup(kind, oracle, rng)
        e = oracle.eval_point(kind, par, x, u)
        # dynamics and costs vs the NumPy restatement
        assert np.allclose(e["xn"], np.asarray(plant.discrete_dynamics(x, u)).ravel(), rtol=1e-12, atol=1e-13)
        if kind == "car":
            cs, cf = plant._cost_stage(x, u), plant._cost_final(x)
        else:
            cs, cf = plant._cost_stage(x, u, plant.xref, plant.uref), plant._cost_final(x, plant.xref)
        assert np.isclose(e["cs"], float(np.squeeze(cs)), rtol=1e-12)
        assert np.isclose(e["cf"], float(np.squeeze(cf)), rtol=1e-12)
        # derivatives vs the NumPy restatement (SymPy-lambdified)
        l_x, l_u, l_xx, l_ux, l_uu, f_x, f_u = plant.stage(x, u)
        V_x, V_xx = plant.final(x)
        for key, ref in (("l_x", l_x), ("l_u", l_u), ("l_xx", l_xx), ("l_ux", l_ux), ("l_uu", l_uu), ("f_x", f_x),
                         ("f_u", f_u), ("V_x", V_x), ("V_xx", V_xx)):
            ref = np.asarray(ref, dtype=np.float64).reshape(e[key].shape)
            sc = max(1.0, np.abs(ref).max())
            assert np.allclose(e[key], ref, rtol=1e-10, atol=1e-11 * sc), (kind, key, np.abs(e[key] - ref).max())
        # dynamics Jacobians vs central differences of the restated dynamics
        if kind != "quad_quat":
            Afd = _fd_jac(lambda z: np.asarray(plant.discrete_dynamics(z, u)).ravel(), x)
            Bfd = _fd_jac(lambda z: np.asarray(plant.discrete_dynamics(x, z)).ravel(), u)
            assert np.allclose(e["f_x"], Afd, rtol=1e-5, atol=2e-6 * max(1, np.abs(Afd).max()))
            assert np.allclose(e["f_u"], Bfd, rtol=1e-5, atol=2e-6 * max(1, np.abs(Bfd).max()))
        else:
            G = plant.G(x)
            Afd = _fd_jac(lambda z: np.asarray(plant.discrete_dynamics(z, u)).ravel(), x)
            Bfd = _fd_jac(lambda z: np.asarray(plant.discrete_dynamics(x, z)).ravel(), u)
            assert np.allclose(e["f_x"], G.T @ Afd @ G, rtol=1e-5, atol=2e-6 * max(1, np.abs(Afd).max()))
            assert np.allclose(e["f_u"], G.T @ Bfd, rtol=1e-5, atol=2e-6 * max(1, np.abs(Bfd).max()))
        # cost gradient / Hessian vs central differences
        if kind == "car":
            c = lambda z: plant._cost_stage(z, u)
        else:
            c = lambda z: float(np.squeeze(plant._cost_stage(z, u, plant.xref, plant.uref)))
        gfd = _fd_jac(c, x).ravel()
        assert np.allclose(np.asarray(l_x, dtype=float).ravel(), gfd, rtol=1e-5, atol=1e-5 * max(1, np.abs(gfd).max()))


def test_hover_equilibrium_known_answer(oracle):
    """EA_Drone.ipynb cell-2 / QuadrotorControllers.py:210: u = m g / 4 = 1.9006875 keeps both quadrotors at rest."""
    uh = np.full(4, 0.775 * 9.81 / 4)
    assert abs(uh[0] - 1.9006875) < 1e-12
    x = np.zeros(12); x[2] = 1.0
    assert np.allclose(mn.quad_ea_xdot(list(x), list(uh)), 0.0, atol=1e-12)
    xq = np.zeros(13); xq[0] = 1.0; xq[6] = 1.0
    assert np.allclose(mn.quad_quat_xdot(list(xq), list(uh)), 0.0, atol=1e-12)
    Q, Qf, R = _quad_weights()
    par = oracle.pack_quadratic(0.01, Q, Qf, R, np.zeros(12), np.zeros(4))
    e = oracle.eval_point("quad_ea", par, x, uh)
    assert np.allclose(e["xn"], x, atol=1e-12)
    parq = oracle.pack_quat(0.01, Q, Qf, R, np.zeros(13), np.zeros(4))
    e = oracle.eval_point("quad_quat", parq, xq, uh)
    assert np.allclose(e["xn"], xq, atol=1e-12)


def test_rotation_conventions_known_answer():
    """Quaternions.ipynb cell-0: QuaternionToRotation([w,x,y,z]) equals SciPy's rotation to ~1e-16;
    Drone.ipynb cell-11: the ZYX Euler matrix equals SciPy's from_euler('xyz', [roll, pitch, yaw])."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(0)
    for _ in range(10):
        q = rng.standard_normal(4)
        Rm = np.array(mn.quat_rotation(list(q)), dtype=float)
        qn = q / np.linalg.norm(q)
        Rs = Rotation.from_quat([qn[1], qn[2], qn[3], qn[0]]).as_matrix()
        assert np.abs(Rm - Rs).max() < 1e-14
        y, p, r = rng.uniform(-1, 1, 3)
        x = np.zeros(12); x[3:6] = [y, p, r]
        # thrust direction = third column of R_NB: acceleration of a unit-thrust hover input
        f = np.array(mn.quad_ea_xdot(list(x), [0.775, 0, 0, 0.0]), dtype=float)   # total thrust 0.775 -> a = R e3 - g e3
        Rs = Rotation.from_euler("xyz", [r, p, y]).as_matrix()
        assert np.allclose(f[6:9] + np.array([0, 0, 9.81]), Rs[:, 2], atol=1e-13)


def test_quaternion_error_state_dimensions(oracle):
    nx, nu, ndx, npar = oracle.dims("quad_quat")
    assert (nx, nu, ndx) == (13, 4, 12)
    assert oracle.dims("car")[:3] == (5, 2, 5) and oracle.dims("acrobot")[:3] == (4, 1, 4)
    assert oracle.dims("quad_ea")[:3] == (12, 4, 12)
