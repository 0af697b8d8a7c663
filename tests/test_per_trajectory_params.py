"""One parameter block per trajectory (batched goals): the batched form of set_references
(control/models/acrobot.py:101-103) / of the car's (r, v_target) (control/models/car.py:55-59).

The reference solves one trajectory with one parameter set, so the expected result of a batch with mixed
parameters is, per trajectory, the result of the same solve in a batch that shares that trajectory's set."""
import os

import numpy as np
import pytest

from test_gpu_parity import _car_box, torch_gpu   # noqa: F401
from user_models import ROVER_SP, Rover, rover_inputs

pytestmark = pytest.mark.gpu


def _fields(r):
    return dict(x=r.x.cpu().numpy(), u=r.u.cpu().numpy(), J=r.J.cpu().numpy(), iters=r.iters.cpu().numpy(),
                status=r.status.cpu().numpy(), regu=r.regu_rounds.cpu().numpy(), ls=r.ls_total.cpu().numpy())


def _same(a, b, idx):
    for k in a:
        assert np.array_equal(a[k][idx], b[k], equal_nan=True), k


def test_car_mixed_targets_equal_the_shared_solves_bit_for_bit(torch_gpu, oracle):
    from ilqr_amd.batch import BatchSolver
    rng = np.random.default_rng(21)
    B = 768
    sets = np.array([[2.0, 2.0], [1.0, 3.0], [3.5, 1.0], [2.5, 0.5]])
    which = rng.integers(0, len(sets), B)
    x0, u0 = _car_box(rng, B), rng.standard_normal((B, 39, 2)) * 1e-4
    mixed = _fields(BatchSolver(0, sets[which], 40, 50, 20, 1e-3).solve(x0, u0))
    for g, par in enumerate(sets):
        idx = np.where(which == g)[0]
        shared = _fields(BatchSolver(0, par, 40, 50, 20, 1e-3).solve(x0[idx], u0[idx]))
        _same(mixed, shared, idx)
        # and the shared solves are the ones the checker agrees with
        ref = oracle.solve_batch("car", par, 40, 50, 20, 1e-3, x0[idx], u0[idx])["info"]
        calm = ref["max_abs_q"] < 1e5
        assert calm.sum() > 20 and np.array_equal(shared["iters"][calm], ref["iters"][calm])
    assert len(np.unique(mixed["J"].round(6))) > B // 2


def test_car_model_with_arrays_through_the_drop_in_api(torch_gpu):
    from ilqr_amd import CarModel, iLQR
    rng = np.random.default_rng(5)
    B = 64
    r, v = rng.uniform(1.0, 3.0, B), rng.uniform(0.5, 3.0, B)
    x0, u0 = _car_box(rng, B), rng.standard_normal((B, 39, 2)) * 1e-4
    x, u = iLQR(CarModel(r, v), 40, 50, 20, 1e-3).run_ilqr(x0, u0.copy())
    for b in (0, 17, 63):
        xs, us = iLQR(CarModel(r[b], v[b]), 40, 50, 20, 1e-3).run_ilqr(x0[b], u0[b].copy())
        assert np.array_equal(xs, x[b]) and np.array_equal(us, u[b])
    with pytest.raises(ValueError, match="one parameter set per trajectory"):
        CarModel(r, v).discrete_dynamics(x0[0], u0[0, 0])


def test_user_plant_with_goals_per_trajectory(torch_gpu):
    from ilqr_amd.iLQRController import iLQR
    plant = Rover()
    plant.use_library(os.path.join(plant.library_dir(), "libilqr_user.so"))
    B = 96
    x0, u0 = rover_inputs(B, 7)
    rng = np.random.default_rng(8)
    P = np.column_stack([rng.uniform(2, 4, B), rng.uniform(-1, 2, B), rng.uniform(1, 2, B), rng.uniform(0, 1, B)])
    plant.set_params(P)
    ctl = iLQR(plant, ROVER_SP["N"], ROVER_SP["max_iter"], ROVER_SP["max_ls"], ROVER_SP["d_tol"])
    x, u = ctl.run_ilqr(x0, u0.copy())
    it = ctl.info.iters.cpu().numpy()
    for b in (0, 31, 95):
        plant.set_params(P[b])
        c1 = iLQR(plant, ROVER_SP["N"], ROVER_SP["max_iter"], ROVER_SP["max_ls"], ROVER_SP["d_tol"])
        xs, us = c1.run_ilqr(x0[b], u0[b].copy())
        assert np.array_equal(xs, x[b]) and np.array_equal(us, u[b]) and int(c1.info.iters[0]) == it[b]


@pytest.mark.parametrize("kind", ["acrobot", "quad_ea", "quad_quat"])
def test_references_per_trajectory(torch_gpu, oracle, kind):
    """set_references with one (xref, uref) per trajectory (plants whose parameter block holds the weight
    matrices too): every trajectory of the mixed batch equals, bit for bit, its solve in a batch that shares
    its references, and each reference group agrees with the CPU checker solving it with that group's references
    (decisions identical and x, u within the parity bar on the solves the checker reproduces under a 1e-14
    perturbation of its inputs)."""
    from ilqr_amd import Acrobot2DModel, QuadrotorEAModel, QuadrotorQuatModel, iLQR
    from helpers import rel_traj
    from test_gpu_parity import _inputs
    rng = np.random.default_rng(12)
    if kind == "acrobot":
        mk = lambda: Acrobot2DModel(np.diag([1, 1, .1, .1]), 100.0 * np.eye(4), np.array([[0.01]]), 0.05, 40)
        N, B, sp = 40, 48, dict(max_iter=12, max_linesearch_iters=10, d_tol=1e-3)
        refs = [(np.array([np.pi, 0, 0, 0]), np.zeros(1)), (np.array([np.pi, 0.3, 0, 0]), np.array([0.2])),
                (np.array([3.0, -0.2, 0, 0]), np.array([-0.1]))]
    else:
        Q = np.diag([10.0] * 6 + [1.0] * 6); Qf = np.diag([500.0] * 3 + [100.0] * 3 + [20.0] * 6); R = np.diag([0.02] * 4)
        cls = QuadrotorEAModel if kind == "quad_ea" else QuadrotorQuatModel
        mk = lambda: cls(Q, Qf, R, dt=0.01, N=30)
        N, B, sp = 30, 24, dict(max_iter=8, max_linesearch_iters=10, d_tol=5e-4)
        nx = 12 if kind == "quad_ea" else 13
        def ref(z, dx):
            xr = np.zeros(nx)
            if nx == 13:
                xr[0] = 1.0; xr[6] = z; xr[4] = dx
            else:
                xr[2] = z; xr[0] = dx
            return xr
        refs = [(ref(1.0, 0.0), np.full(4, 1.9)), (ref(1.3, 0.2), np.full(4, 1.9)), (ref(0.8, -0.3), np.full(4, 2.0))]
    x0, u0 = _inputs(kind, rng, B, N)
    which = rng.integers(0, len(refs), B)
    plant = mk()
    plant.set_references(np.stack([refs[g][0] for g in which]), np.stack([refs[g][1] for g in which]))
    ctl = iLQR(plant, N, **sp)
    x, u = ctl.run_ilqr(x0, u0.copy())
    it_mixed = ctl.info.iters.cpu().numpy()
    compared = 0
    for g, (xr, ur) in enumerate(refs):
        idx = np.where(which == g)[0]
        shared = mk(); shared.set_references(xr, ur)
        xs, us = iLQR(shared, N, **sp).run_ilqr(x0[idx], u0[idx].copy())
        assert np.array_equal(xs, x[idx], equal_nan=True) and np.array_equal(us, u[idx], equal_nan=True), (kind, g)
        # the checker with this group's references
        par = shared._param_block()
        args = (kind, par, N, sp["max_iter"], sp["max_linesearch_iters"], sp["d_tol"])
        ref = oracle.solve_batch(*args, x0[idx], u0[idx])
        sg = np.random.default_rng(5).choice([-1.0, 1.0], size=u0[idx].shape)
        per = oracle.solve_batch(*args, x0[idx], u0[idx] * (1 + 1e-14 * sg))
        i, j = ref["info"], per["info"]
        sens = np.maximum(rel_traj(per["x"], ref["x"]), rel_traj(per["u"], ref["u"]))
        stable = ((i["iters"] == j["iters"]) & (i["status"] == j["status"]) & (i["regu_rounds"] == j["regu_rounds"])
                  & (i["status"] != 3) & (sens <= 1e-6))
        assert np.array_equal(it_mixed[idx][stable], i["iters"][stable]), (kind, g)
        tol = np.maximum(1e-8, 100 * sens)
        assert (rel_traj(xs, ref["x"])[stable] <= tol[stable]).all(), (kind, g)
        assert (rel_traj(us, ref["u"])[stable] <= tol[stable]).all(), (kind, g)
        compared += int(stable.sum())
    assert compared >= B // 2, (kind, compared)
    assert not np.array_equal(x[which == 0][0], x[which == 1][0])


def test_closed_loops_with_a_goal_per_loop(torch_gpu, oracle):
    """QuadrotorControllers.py:64-65 sets the goal of its one vehicle before every solve; in a batch of closed loops
    every loop has its own.  Loops with goal A inside a mixed batch follow, bit for bit, the same loops run in a
    batch that shares goal A."""
    from ilqr_amd import QuadrotorQuatModel
    from ilqr_amd.batch import BatchSolver
    from ilqr_amd.mpc import BatchMPC
    from test_gpu_parity import _inputs
    rng = np.random.default_rng(31)
    B, N, steps = 12, 20, 3
    Q = np.diag([10.0] * 6 + [1.0] * 6); Qf = np.diag([500.0] * 3 + [100.0] * 3 + [20.0] * 6); R = np.diag([0.02] * 4)
    goals = []
    for z, dx in ((1.0, 0.0), (1.2, 0.3)):
        xr = np.zeros(13); xr[0] = 1.0; xr[6] = z; xr[4] = dx
        goals.append(xr)
    which = np.arange(B) % 2
    x0, u0 = _inputs("quad_quat", rng, B, N)
    plant = QuadrotorQuatModel(Q, Qf, R, dt=0.01, N=N)
    plant.set_references(np.stack([goals[g] for g in which]), np.full(4, 1.9))
    s = BatchSolver(3, plant._param_block(), N, 5, 10, 5e-4)
    log = BatchMPC(s, shift=True).run(x0, u0, steps=steps)
    for g, xr in enumerate(goals):
        idx = np.where(which == g)[0]
        shared = QuadrotorQuatModel(Q, Qf, R, dt=0.01, N=N)
        shared.set_references(xr, np.full(4, 1.9))
        ref = BatchMPC(BatchSolver(3, shared._param_block(), N, 5, 10, 5e-4), shift=True).run(x0[idx], u0[idx], steps=steps)
        for k in ("x", "u", "J", "iters", "status"):
            a, b = log[k].cpu().numpy()[:, idx], ref[k].cpu().numpy()
            assert np.array_equal(a, b, equal_nan=True) if a.dtype.kind == "f" else np.array_equal(a, b), (g, k)
    xs = log["x"].cpu().numpy()
    assert not np.array_equal(xs[-1, 0], xs[-1, 1])
