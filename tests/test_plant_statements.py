"""The plants exist in three statements: ilqr_amd/plants.py (what the HIP library's code is generated from),
oracle/models_numpy.py (what the golden vectors were produced with, driving the imported reference solver) and the
C checker's own, hand-written one (oracle/oracle_models.h, oracle/plants_ad.cpp: forward-mode AD of hand-written
dynamics).  The first two are the same arithmetic and must give the same numbers; the third shares no text with them and
must agree with them to rounding, derivatives included.  CPU only."""
import numpy as np


def test_plant_statements_agree_numerically():
    from ilqr_amd import plants as p
    from oracle import models_numpy as o
    rng = np.random.default_rng(5)
    for _ in range(20):
        x5, u2 = list(rng.standard_normal(5)), list(rng.standard_normal(2))
        assert p.car_step(x5, u2) == o.car_step(x5, u2)
        assert p.car_cost_stage(x5, u2, 2.0, 1.5) == o.car_cost_stage(x5, u2, 2.0, 1.5)
        assert p.car_cost_final(x5, 2.0, 1.5) == o.car_cost_final(x5, 2.0, 1.5)
        x4, u1 = list(rng.standard_normal(4)), list(rng.standard_normal(1))
        assert p.acrobot_xdot(x4, u1) == o.acrobot_xdot(x4, u1)
        x12, u4 = list(rng.standard_normal(12)), list(rng.standard_normal(4))
        assert p.quad_ea_xdot(x12, u4) == o.quad_ea_xdot(x12, u4)
        x13 = list(rng.standard_normal(13))
        assert p.quad_quat_xdot(x13, u4) == o.quad_quat_xdot(x13, u4)
        xr, Qw, Rw = list(rng.standard_normal(13)), rng.standard_normal((12, 12)).tolist(), rng.standard_normal((4, 4)).tolist()
        ur = list(rng.standard_normal(4))
        assert p.quat_cost(x13, u4, xr, ur, Qw, Rw) == o.quat_cost(x13, u4, xr, ur, Qw, Rw)
        assert p.quat_state_error(x13, xr) == o.quat_state_error(x13, xr)
        assert p.rk4_step(p.acrobot_xdot, x4, u1, 0.05) == o.rk4_step(o.acrobot_xdot, x4, u1, 0.05)


def test_hand_written_checker_plants_agree_with_the_numpy_statement(oracle):
    """oracle/plants_ad.cpp against oracle/models_numpy.py on random points: continuous dynamics through one RK4 step,
    and the derivatives the solver uses (f_x, f_u by AD there; by central differences of the NumPy step here)."""
    from oracle import models_numpy as o
    rng = np.random.default_rng(6)
    Q12 = np.diag(rng.uniform(0.5, 2, 12)); R4 = np.diag(rng.uniform(0.1, 1, 4))
    cases = [("acrobot", o.acrobot_xdot, 4, 1, 0.05, oracle.pack_quadratic(0.05, np.eye(4), np.eye(4), [[0.1]], np.zeros(4), [0.0])),
             ("quad_ea", o.quad_ea_xdot, 12, 4, 0.01, oracle.pack_quadratic(0.01, Q12, Q12, R4, np.zeros(12), np.zeros(4)))]
    xr = np.zeros(13); xr[0] = 1.0
    cases.append(("quad_quat", o.quad_quat_xdot, 13, 4, 0.01, oracle.pack_quat(0.01, Q12, Q12, R4, xr, np.zeros(4))))
    for kind, xdot, nx, nu, dt, par in cases:
        for _ in range(10):
            x = 0.5 * rng.standard_normal(nx); u = 1.9 + 0.5 * rng.standard_normal(nu)
            if kind == "quad_quat":
                x[:4] = [1, 0, 0, 0] + 0.3 * rng.standard_normal(4)      # not normalised on purpose
            step = lambda xv, uv: np.array(o.rk4_step(xdot, list(xv), list(uv), dt), dtype=float)
            e = oracle.eval_point(kind, par, x, u)
            assert np.allclose(e["xn"], step(x, u), rtol=1e-12, atol=1e-13), kind
            if kind == "quad_quat":
                continue      # f_x, f_u are in the 12-dim error state there: covered by tests/test_models.py
            h = 1e-6
            fx = np.column_stack([(step(x + h * np.eye(nx)[j], u) - step(x - h * np.eye(nx)[j], u)) / (2 * h) for j in range(nx)])
            fu = np.column_stack([(step(x, u + h * np.eye(nu)[j]) - step(x, u - h * np.eye(nu)[j])) / (2 * h) for j in range(nu)])
            assert np.allclose(e["f_x"], fx, rtol=1e-6, atol=1e-7), kind
            assert np.allclose(e["f_u"], fu, rtol=1e-6, atol=1e-7), kind


def test_quaternion_cost_with_full_weights(oracle):
    """Quadrotor_Quat.py:150-175 takes any Q, Qf [12 x 12] and R [4 x 4].  The checker's cost (hand-written, AD) against
    the NumPy statement of 0.5 e'Qe + 0.5 du'R du for weights that are neither diagonal nor symmetric: values, and the
    13-dim gradient and Hessian against central differences of the NumPy value."""
    from oracle import models_numpy as o
    rng = np.random.default_rng(8)
    Q = rng.standard_normal((12, 12)); Qf = rng.standard_normal((12, 12)); R = rng.standard_normal((4, 4))
    xr = rng.standard_normal(13); ur = rng.standard_normal(4)
    par = oracle.pack_quadratic(0.01, Q, Qf, R, xr, ur)
    assert par.size == 323
    cost = lambda xv, uv: o.quat_cost(list(xv), list(uv), list(xr), list(ur), Q.tolist(), R.tolist())
    for _ in range(5):
        x = 0.5 * rng.standard_normal(13); x[:4] = [1, 0, 0, 0] + 0.3 * rng.standard_normal(4)
        u = rng.standard_normal(4)
        e = oracle.eval_point("quad_quat", par, x, u)
        assert np.isclose(e["cs"], cost(x, u), rtol=1e-12, atol=1e-13)
        assert np.isclose(e["cf"], o.quat_cost(list(x), list(u), list(xr), list(ur), Qf.tolist(), R.tolist(), with_action=False),
                          rtol=1e-12, atol=1e-13)
        h = 1e-5
        I = np.eye(13)
        g = np.array([(cost(x + h * I[j], u) - cost(x - h * I[j], u)) / (2 * h) for j in range(13)])
        gu = np.array([(cost(x, u + h * np.eye(4)[j]) - cost(x, u - h * np.eye(4)[j])) / (2 * h) for j in range(4)])
        # (eval_point returns the cost derivatives in the 13-dim state; the solver projects them with G afterwards)
        assert np.allclose(e["l_x"], g, rtol=1e-6, atol=1e-7)
        assert np.allclose(e["l_u"], gu, rtol=1e-6, atol=1e-7)
        assert np.allclose(e["l_uu"], 0.5 * (R + R.T), rtol=1e-12)
        h2 = 1e-4
        H = np.array([[(cost(x + h2 * I[i] + h2 * I[j], u) - cost(x + h2 * I[i] - h2 * I[j], u)
                        - cost(x - h2 * I[i] + h2 * I[j], u) + cost(x - h2 * I[i] - h2 * I[j], u)) / (4 * h2 * h2)
                       for j in range(13)] for i in range(13)])
        assert np.allclose(e["l_xx"], H, rtol=1e-4, atol=1e-5)
