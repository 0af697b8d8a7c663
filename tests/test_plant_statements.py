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
        xr, qd, rd = list(rng.standard_normal(13)), list(rng.uniform(0.1, 2, 12)), list(rng.uniform(0.1, 2, 4))
        assert p.quat_cost(x13, u4, xr, u4, qd, rd) == o.quat_cost(x13, u4, xr, u4, qd, rd)
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
