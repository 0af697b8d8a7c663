"""Quaternion helpers against the reference's own outputs (tests/golden/quat_utils.npz, produced by importing
maths/quaternions.py in tests/golden/make_golden.py), the Riccati helpers, and the LQR controller on the GPU plants."""
import numpy as np
import pytest

from conftest import load_golden


def test_quaternion_helpers_match_reference_outputs():
    from ilqr_amd.maths import quaternions as mq
    g = load_golden("quat_utils")
    for i, q in enumerate(g["q"]):
        assert np.allclose(mq.hat(q), g["hat"][i], rtol=0, atol=0)
        assert np.array_equal(mq.GetConjugate(q), g["conj"][i])
        assert np.array_equal(mq.GetLeftMatrix(q), g["left"][i])
        assert np.array_equal(mq.GetAttitudeJacobian(q), g["G"][i])
        assert np.allclose(mq.QuaternionToParam(q.copy()), g["q2p"][i], rtol=1e-15, atol=0)
        assert np.allclose(mq.QuaternionToRotation(q), g["rot"][i], rtol=0, atol=1e-15)
    for i, phi in enumerate(g["phi"]):
        got = mq.ParamToQuaternion(phi)
        assert got.shape == g["p2q"][i].shape and np.allclose(got, g["p2q"][i], rtol=1e-15, atol=0)
    np.random.seed(0)
    for near in (False, True):
        q = mq.SampleQuaternion(near)
        assert q.shape == (4,) and abs(np.linalg.norm(q) - 1.0) < 1e-12
        R = mq.QuaternionToRotation(q)
        assert np.allclose(R @ R.T, np.eye(3), atol=1e-12) and abs(np.linalg.det(R) - 1.0) < 1e-12


def test_riccati_helpers():
    from ilqr_amd.lqr import dlqr, lqr
    rng = np.random.default_rng(0)
    A = np.eye(4) + 0.1 * rng.standard_normal((4, 4)); B = rng.standard_normal((4, 2))
    Q = np.diag([1.0, 2.0, 3.0, 4.0]); R = np.diag([0.5, 0.1])
    K, S = dlqr(A, B, Q, R)
    res = A.T @ S @ A - S - A.T @ S @ B @ np.linalg.solve(R + B.T @ S @ B, B.T @ S @ A) + Q
    assert np.abs(res).max() < 1e-8 * np.abs(S).max()
    assert np.abs(np.linalg.eigvals(A - B @ K)).max() < 1.0
    Kc, Sc = lqr(A - np.eye(4), B, Q, R)
    res = (A - np.eye(4)).T @ Sc + Sc @ (A - np.eye(4)) - Sc @ B @ np.linalg.solve(R, B.T @ Sc) + Q
    assert np.abs(res).max() < 1e-8 * np.abs(Sc).max()
    assert np.linalg.eigvals(A - np.eye(4) - B @ Kc).real.max() < 0.0


@pytest.mark.gpu
def test_lqr_controller_stabilises_the_quadrotors_about_hover(oracle):
    from ilqr_amd import QuadrotorEAModel, QuadrotorQuatModel
    from ilqr_amd.lqr import LQRController
    Q = np.diag([10.0] * 6 + [1.0] * 6); R = np.diag([0.1] * 4)
    hover = np.full(4, 1.9006875)                       # QuadrotorControllers.py:210
    for cls, kind in ((QuadrotorEAModel, "quad_ea"), (QuadrotorQuatModel, "quad_quat")):
        plant = cls(Q, Q, R, dt=0.01, N=20)
        nx = plant.nx
        goal = np.zeros(nx); goal[2 if nx == 12 else 6] = 1.0
        if nx == 13:
            goal[0] = 1.0
        ctl = LQRController(plant, uref=hover)
        A, B = plant._linearize_discrete(goal, hover)
        par = plant._param_block()
        e = oracle.eval_point(kind, par, goal, hover)
        # (for the quaternion plant both sides are on the error state: G'AG, G'B)
        assert np.allclose(A, e["f_x"], rtol=1e-10, atol=1e-12) and np.allclose(B, e["f_u"], rtol=1e-10, atol=1e-12)
        K = ctl.gain(goal)
        assert K.shape == (4, 12) and np.abs(np.linalg.eigvals(A - B @ K)).max() < 1.0
        # closed loop on the nonlinear plant: a displaced start converges to the goal
        x = goal.copy(); x[(0 if nx == 12 else 4)] += 0.3; x[(1 if nx == 12 else 5)] -= 0.2
        d0 = np.linalg.norm(plant.CalcDifferentialState(x, goal))
        for _ in range(400):
            x = plant.discrete_dynamics(x, ctl.control(x, goal))
        assert np.linalg.norm(plant.CalcDifferentialState(x, goal)) < 0.05 * d0
        # mode "continuous" (QuadrotorControllers.py:123-124, :228-229): Jacobians of xdot against the checker's own
        # (hand-written dynamics, forward-mode AD), the continuous Riccati equation, and a stable closed loop
        import ctypes as C
        L = oracle.lib()
        fn = getattr(L, "oracle_quad_ea_xdot_jac" if nx == 12 else "oracle_quad_quat_xdot_jac")
        fn.restype = None
        dp = C.POINTER(C.c_double)
        f, Ao, Bo = np.zeros(nx), np.zeros((nx, nx)), np.zeros((nx, 4))
        xq = goal + 0.1 * np.random.default_rng(4).standard_normal(nx)
        fn(*(v.ctypes.data_as(dp) for v in (xq, hover, f, Ao, Bo)))
        if nx == 13:
            qn = xq[:4] / np.linalg.norm(xq[:4]); w_, x_, y_, z_ = qn
            Gx = np.zeros((13, 12)); Gx[:4, :3] = [[-x_, -y_, -z_], [w_, -z_, y_], [z_, w_, -x_], [-y_, x_, w_]]
            Gx[4:, 3:] = np.eye(9)
            Ao, Bo = Gx.T @ Ao @ Gx, Gx.T @ Bo
        Ac, Bc = plant._linearize_continuous(xq, hover)
        assert Ac.shape == (12, 12) and Bc.shape == (12, 4)
        assert np.allclose(Ac, Ao, rtol=1e-10, atol=1e-11) and np.allclose(Bc, Bo, rtol=1e-10, atol=1e-11)
        cc = LQRController(plant, uref=hover, mode="continuous")
        Kc = cc.gain(goal)
        Ag, Bg = plant._linearize_continuous(goal, hover)
        assert Kc.shape == (4, 12) and np.linalg.eigvals(Ag - Bg @ Kc).real.max() < 0.0
        # the continuous-time gain is stiff (closed-loop poles to -450/s: outside RK4's stability region at dt = 0.01),
        # so the closed loop is simulated with a finer step of the same plant
        fine = cls(Q, Q, R, dt=0.002, N=20)
        x = goal.copy(); x[(0 if nx == 12 else 4)] += 0.3
        for _ in range(1200):
            x = fine.discrete_dynamics(x, cc.control(x, goal))
        assert np.linalg.norm(plant.CalcDifferentialState(x, goal)) < 0.05 * 0.3
    from ilqr_amd import CarModel
    with pytest.raises(RuntimeError):          # car.py states its step only
        CarModel(2, 2)._linearize_continuous(np.zeros(5), np.zeros(2))


@pytest.mark.gpu
def test_lqr_controller_balances_the_acrobot_upright(oracle):
    """AcrobotLQRController (control/AcrobotControllers.py:68-115): the acrobot linearised at the upright goal with
    u = 0, gain from the discrete or the continuous Riccati equation with the plant's own Q and R, u = -K (x - goal).
    The Jacobians come from the GPU library and are checked against the checker's (hand-written dynamics, forward-mode
    AD); the closed loop on the nonlinear plant returns to the goal from a displaced start in both modes."""
    import ctypes as C
    from ilqr_amd import Acrobot2DModel
    from ilqr_amd.lqr import LQRController
    Q = np.diag([10.0, 10.0, 1.0, 1.0]); R = np.array([[1.0]])
    plant = Acrobot2DModel(Q, 100.0 * np.eye(4), R, 0.01, 50)
    goal = np.array([np.pi, 0.0, 0.0, 0.0])
    u0 = np.zeros(1)
    A, B = plant._linearize_discrete(goal, u0)
    e = oracle.eval_point("acrobot", plant._param_block(), goal, u0)
    assert A.shape == (4, 4) and B.shape == (4, 1)
    assert np.allclose(A, e["f_x"], rtol=1e-10, atol=1e-12) and np.allclose(B, e["f_u"], rtol=1e-10, atol=1e-12)
    assert np.abs(np.linalg.eigvals(A)).max() > 1.0            # upright is unstable without control
    L = oracle.lib()
    L.oracle_acrobot_xdot_jac.restype = None
    dp = C.POINTER(C.c_double)
    xq = goal + 0.2 * np.random.default_rng(8).standard_normal(4)
    uq = np.array([0.3])
    f, Ao, Bo = np.zeros(4), np.zeros((4, 4)), np.zeros((4, 1))
    L.oracle_acrobot_xdot_jac(*(v.ctypes.data_as(dp) for v in (xq, uq, f, Ao, Bo)))
    Ac, Bc = plant._linearize_continuous(xq, uq)
    assert np.allclose(Ac, Ao, rtol=1e-10, atol=1e-11) and np.allclose(Bc, Bo, rtol=1e-10, atol=1e-11)
    for mode in ("discrete", "continuous"):
        ctl = LQRController(plant, mode=mode)          # the plant's own Q, R; uref = 0 as in the reference
        K = ctl.gain(goal)
        assert K.shape == (1, 4)
        if mode == "discrete":
            assert np.abs(np.linalg.eigvals(A - B @ K)).max() < 1.0
        else:
            Ag, Bg = plant._linearize_continuous(goal, u0)
            assert np.linalg.eigvals(Ag - Bg @ K).real.max() < 0.0
        x = goal + np.array([0.05, -0.04, 0.0, 0.0])
        d0 = np.linalg.norm(x - goal)
        for _ in range(1500):
            x = plant.discrete_dynamics(x, ctl.control(x, goal))
        assert np.linalg.norm(x - goal) < 0.02 * d0, (mode, x - goal)
