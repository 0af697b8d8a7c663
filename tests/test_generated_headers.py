"""The shipped library and the CPU checker each have their own statement of the four plants
(ilqr_amd/plants.py; oracle/models_numpy.py, the one the golden vectors were produced with) and their own
generated headers.  The two statements must say the same thing: same numbers on random points, and the
committed headers, printed from one and from the other, identical.  CPU only."""
import os

import numpy as np
import pytest

from conftest import ROOT

MODELS = ["car", "acrobot", "quad_ea", "quad_quat"]


@pytest.mark.parametrize("name", MODELS)
def test_headers_of_library_and_checker_are_the_same_text(name):
    a = open(os.path.join(ROOT, "ilqr_amd", "csrc", "gen", name + "_gen.h")).read()
    b = open(os.path.join(ROOT, "oracle", "gen", name + "_gen.h")).read()
    assert a == b and "static inline" in a


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
