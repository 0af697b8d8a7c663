"""GPU parity tests: the HIP path (through the C ABI) against the golden vectors of the reference
solver and against the CPU checker on seeded batches.  Run with `-m gpu` on an MI355X."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from helpers import compare_to_golden, cost_horizon, min_compared, min_within_1e8, model_key, rel_traj, tolerance

pytestmark = pytest.mark.gpu
MID = dict(car=0, acrobot=1, quad_ea=2, quad_quat=3)


@pytest.fixture(scope="module")
def torch_gpu():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    from ilqr_amd import _lib
    _lib.lib()          # fails loudly if libilqr_hip.so is missing
    return torch


def _solve(g_or_cfg, x0, u0, trace=64, flags=0):
    from ilqr_amd.batch import BatchSolver
    c = g_or_cfg
    s = BatchSolver(MID[str(c["model"])], c["par"], int(c["N"]), int(c["max_iter"]), int(c["max_ls"]),
                    float(c["d_tol"]), trace_len=trace, flags=flags, cost_N=cost_horizon(c))
    r = s.solve(x0, u0)
    cp = lambda t: t.cpu().numpy()
    return dict(x=cp(r.x), u=cp(r.u), J=cp(r.J), J0=cp(r.J0), iters=cp(r.iters), status=cp(r.status),
                regu=cp(r.regu_rounds), ls_total=cp(r.ls_total), ls=cp(r.traces["ls_evals"]),
                acc=cp(r.traces["accepted"]), maxq=cp(r.max_abs_q), dmax=cp(r.dmax))


@pytest.mark.parametrize("name", golden_names())
def test_hip_matches_reference_golden(torch_gpu, name):
    g = load_golden(name)
    r = _solve(g, g["x0"], g["u_init"])
    n = compare_to_golden(g, r["x"], r["u"], r["J"], r["iters"], r["status"], r["ls"], r["acc"], r["regu"], J0=r["J0"],
                          label=name)
    assert n >= min_compared(name, g), (name, n)
    # the 1e-8 bar as a count: where the reference reproduces itself to 1e-8, so do the kernels
    assert n.within_1e8 >= min_within_1e8(name, g), (name, n.within_1e8, n.stable)


def torch_equal(a, b):
    """Bit for bit, NaNs in the same places."""
    a, b = a.cpu().numpy(), b.cpu().numpy()
    return np.array_equal(a, b, equal_nan=a.dtype.kind == "f")


def _car_box(rng, B):
    return np.column_stack([rng.uniform(-4, 4, B), rng.uniform(-4, 4, B), rng.uniform(-np.pi, np.pi, B),
                            rng.uniform(0, 3, B), rng.uniform(-0.3, 0.3, B)])


def _cfg(oracle, kind):
    Q = np.diag([10.0] * 6 + [1.0] * 6); Qf = np.diag([500.0] * 3 + [100.0] * 3 + [20.0] * 6); R = np.diag([0.02] * 4)
    if kind == "car":
        return dict(model="car", par=oracle.pack_car(2.0, 2.0), N=40, max_iter=50, max_ls=20, d_tol=1e-3)
    if kind == "acrobot":
        par = oracle.pack_quadratic(0.05, np.diag([1, 1, .1, .1]), 100.0 * np.eye(4), [[0.01]], [np.pi, 0, 0, 0], [0.0])
        return dict(model="acrobot", par=par, N=100, max_iter=20, max_ls=10, d_tol=1e-3)
    if kind == "quad_ea":
        xref = np.zeros(12); xref[2] = 1.0
        return dict(model="quad_ea", par=oracle.pack_quadratic(0.01, Q, Qf, R, xref, np.zeros(4)), N=30, max_iter=20,
                    max_ls=10, d_tol=5e-4)
    xref = np.zeros(13); xref[0] = 1.0; xref[6] = 1.0
    return dict(model="quad_quat", par=oracle.pack_quat(0.01, Q, Qf, R, xref, np.zeros(4)), N=30, max_iter=20,
                max_ls=10, d_tol=5e-4)


def _inputs(kind, rng, B, N):
    if kind == "car":
        return _car_box(rng, B), rng.standard_normal((B, N - 1, 2)) * 1e-4
    if kind == "acrobot":
        x0 = np.column_stack([rng.uniform(-np.pi, np.pi, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B),
                              rng.uniform(-1, 1, B)])
        return x0, rng.standard_normal((B, N - 1, 1))
    if kind == "quad_ea":
        x0 = np.zeros((B, 12)); x0[:, 2] = 0.8; x0[:, 3:6] = 0.5 * rng.standard_normal((B, 3))
        return x0, 1.9 + 0.5 * rng.standard_normal((B, N - 1, 4))
    x0 = np.zeros((B, 13)); x0[:, 6] = 0.8
    q = np.column_stack([np.ones(B), 0.25 * rng.standard_normal((B, 3))])
    x0[:, :4] = q / np.linalg.norm(q, axis=1, keepdims=True)
    return x0, 1.9 + 0.5 * rng.standard_normal((B, N - 1, 4))


@pytest.mark.parametrize("kind,B", [("car", 4096), ("acrobot", 512), ("quad_ea", 256), ("quad_quat", 256)])
def test_hip_matches_checker_on_seeded_batch(torch_gpu, oracle, kind, B):
    """Same seeded inputs through the HIP path and the CPU checker.  A trajectory is compared tightly when the
    checker reproduces itself under 1e-14 input perturbations (the fixtures' `stable`/`sens` rule)."""
    rng = np.random.default_rng(dict(car=1234, acrobot=2, quad_ea=3, quad_quat=4)[kind])
    c = _cfg(oracle, kind)
    x0, u0 = _inputs(kind, rng, B, c["N"])
    ref = oracle.solve_batch(kind, c["par"], c["N"], c["max_iter"], c["max_ls"], c["d_tol"], x0, u0)
    i = ref["info"]
    stable = i["status"] != 3
    sens = np.zeros(B)
    for seed in (1, 2, 3):          # three independent 1e-14 perturbations of the inputs
        sgn = np.random.default_rng(seed).choice([-1.0, 1.0], size=u0.shape)
        sgx = np.random.default_rng(100 + seed).choice([-1.0, 1.0], size=x0.shape)
        per = oracle.solve_batch(kind, c["par"], c["N"], c["max_iter"], c["max_ls"], c["d_tol"],
                                 x0 * (1 + 1e-14 * sgx), u0 * (1 + 1e-14 * sgn))
        j = per["info"]
        stable &= ((i["iters"] == j["iters"]) & (i["status"] == j["status"]) & (i["regu_rounds"] == j["regu_rounds"])
                   & (i["ls_evals"] == j["ls_evals"]).all(1))
        sk = np.maximum(rel_traj(per["x"], ref["x"]), rel_traj(per["u"], ref["u"]))
        sens = np.maximum(sens, np.where(np.isfinite(sk), sk, np.inf))
    stable &= sens <= 1e-8
    r = _solve(c, x0, u0)
    lin = (i["status"] == 3) & (i["regu_rounds"] == 0)
    assert np.array_equal(r["status"][lin], i["status"][lin])
    same = ((r["iters"] == i["iters"]) & (r["status"] == i["status"]) & (r["regu"] == i["regu_rounds"])
            & (r["ls"] == i["ls_evals"]).all(1) & (r["acc"] == i["accepted"]).all(1))
    # the stability classification is itself a sample (three perturbations): allow a sliver of misclassified solves
    n_bad = int((~same & stable).sum())
    assert n_bad <= max(1, stable.sum() // 200), f"{kind}: {n_bad} of {int(stable.sum())} stable trajectories differ in decisions"
    stable &= same
    rx, ru = rel_traj(r["x"], ref["x"]), rel_traj(r["u"], ref["u"])
    tol = tolerance(sens, i["status"])
    n_far = int(((rx > tol) | (ru > tol))[stable].sum())
    assert n_far <= max(1, stable.sum() // 200), (kind, n_far, rx[stable].max(), ru[stable].max())
    Jr = i["J_last"]
    fin = stable & np.isfinite(Jr) & (rx <= tol) & (ru <= tol)
    assert (np.abs(r["J"][fin] - Jr[fin]) <= tol[fin] * np.abs(Jr[fin])).all()
    assert stable.mean() > (0.5 if kind == "car" else 0.15), f"too few comparable trajectories: {stable.mean():.2f}"
    # the statistics of the whole batch (ill-conditioned solves included) stay close
    assert abs(r["iters"].mean() - i["iters"].mean()) <= 0.15 * i["iters"].mean() + 0.5


def _variant(oracle, kind, variant):
    """The plant configurations of _cfg, or the same plants with what their constructors also accept: Euler integration
    (acrobot.py:83-86) and, for the quaternion plant, weights that are neither diagonal nor symmetric."""
    c = _cfg(oracle, kind)
    if variant == "notebook":
        return c
    rng = np.random.default_rng(17)
    par = np.array(c["par"], copy=True)
    par[-1] = 1.0                                     # euler
    if kind == "quad_quat":
        par[1:145] += 0.3 * rng.standard_normal(144)      # Q
        par[145:289] += 3.0 * rng.standard_normal(144)    # Qf
        par[289:305] += 0.004 * rng.standard_normal(16)   # R
    return dict(c, par=par)


@pytest.mark.parametrize("kind,variant", [("car", "notebook"), ("acrobot", "notebook"), ("quad_ea", "notebook"),
                                          ("quad_quat", "notebook"), ("acrobot", "euler"), ("quad_ea", "euler"),
                                          ("quad_quat", "euler_full_weights")])
def test_model_eval_matches_checker(torch_gpu, oracle, kind, variant):
    """discrete_dynamics, _cost_stage, _cost_final, stage, final at random points (fp64, rtol 1e-10)."""
    from ilqr_amd.batch import BatchSolver
    rng = np.random.default_rng(11)
    c = _variant(oracle, kind, variant)
    B = 128
    x0, u0 = _inputs(kind, rng, B, 3)
    xs = x0 + 0.1 * rng.standard_normal(x0.shape)
    us = u0[:, 0, :]
    s = BatchSolver(MID[kind], c["par"], 3, 1)
    out = {k: v.cpu().numpy() for k, v in s.model_eval(xs, us).items()}
    for b in range(B):
        e = oracle.eval_point(kind, c["par"], xs[b], us[b])
        if kind == "quad_quat":
            # the HIP path returns the cost derivatives already projected on the error state
            Gx = np.zeros((13, 12)); q = xs[b, :4] / np.linalg.norm(xs[b, :4]); w, x, y, z = q
            Gx[:4, :3] = [[-x, -y, -z], [w, -z, y], [z, w, -x], [-y, x, w]]; Gx[4:, 3:] = np.eye(9)
            e["l_x"] = e["l_x"] @ Gx; e["l_xx"] = Gx.T @ e["l_xx"] @ Gx; e["l_ux"] = e["l_ux"] @ Gx
        for k in ("xn", "cs", "cf", "l_x", "l_u", "l_xx", "l_ux", "l_uu", "f_x", "f_u", "V_x", "V_xx"):
            ref = np.asarray(e[k], dtype=np.float64)
            got = out[k][b].reshape(ref.shape)
            sc = max(1.0, np.abs(ref).max())
            assert np.allclose(got, ref, rtol=1e-10, atol=1e-11 * sc), (kind, k, b, np.abs(got - ref).max())


def test_rollout_cost_and_final_state_is_a_rollout(torch_gpu, oracle):
    """Property at the headline batch size: the returned x is bit-for-bit the rollout of the returned u
    (iLQRController.py:63, :71-72), x[0] == x0, iters <= max_iter, J == cost of (x, u) after an accepted search."""
    from ilqr_amd.batch import BatchSolver
    rng = np.random.default_rng(1234)
    B = 1 << 16
    c = _cfg(oracle, "car")
    x0, u0 = _car_box(rng, B), rng.standard_normal((B, 39, 2)) * 1e-4
    s = BatchSolver(0, c["par"], 40, 50, 20, 1e-3, trace_len=50)
    r = s.solve(x0, u0)
    x, u = r.x.cpu().numpy(), r.u.cpu().numpy()
    st, it = r.status.cpu().numpy(), r.iters.cpu().numpy()
    assert set(np.unique(st)) <= {0, 1, 2, 3}
    assert it.max() <= 50 and it.min() >= 1
    assert np.array_equal(x[:, 0, :], x0)
    xr, Jr = s.rollout_cost(x0, u)
    ok = st != 3
    assert np.array_equal(xr.cpu().numpy()[ok], x[ok])
    # after an accepted last search (or none abandoned), J is the cost of the returned trajectory
    acc, ls = r.traces["accepted"].cpu().numpy(), r.traces["ls_evals"].cpu().numpy()
    searched = ls > 0
    clean = ok & ((acc == 1) | ~searched).all(1)
    J = r.J.cpu().numpy()
    assert np.array_equal(Jr.cpu().numpy()[clean], J[clean])
    assert (J[clean] <= r.J0.cpu().numpy()[clean]).all()
    assert clean.mean() > 0.5


def test_same_batch_is_bitwise_reproducible_and_shards_agree(torch_gpu, oracle):
    """Trajectories are independent: solving a batch twice, or in two halves, gives bit-identical results."""
    rng = np.random.default_rng(3)
    B = 2048
    c = _cfg(oracle, "car")
    x0, u0 = _car_box(rng, B), rng.standard_normal((B, 39, 2)) * 1e-4
    a = _solve(c, x0, u0)
    b = _solve(c, x0, u0)
    for k in ("x", "u", "J", "iters", "status", "regu", "ls"):
        assert np.array_equal(a[k], b[k], equal_nan=True) if a[k].dtype.kind == "f" else np.array_equal(a[k], b[k]), k
    h1, h2 = _solve(c, x0[:1000], u0[:1000]), _solve(c, x0[1000:], u0[1000:])
    for k in ("x", "u", "J", "iters", "status"):
        whole = a[k]
        parts = np.concatenate([h1[k], h2[k]])
        assert np.array_equal(whole, parts, equal_nan=True) if whole.dtype.kind == "f" else np.array_equal(whole, parts), k


@pytest.mark.parametrize("name", ["car_bigu", "car_maxls2", "car_box256", "acrobot_N40", "quad_quat_N30"])
def test_line_search_schedule_does_not_change_results(torch_gpu, name):
    """Candidates evaluated one launch per alpha, side by side, or mixed: bit-identical results and counts."""
    from ilqr_amd import _lib
    g = load_golden(name)
    base = _solve(g, g["x0"], g["u_init"], flags=_lib.FLAG_LS_SEQUENTIAL)
    assert (base["ls"] > 1).any()
    for flags in (0, _lib.FLAG_PARALLEL_LS):
        r = _solve(g, g["x0"], g["u_init"], flags=flags)
        for k in ("x", "u", "J", "iters", "status", "regu", "ls", "acc", "ls_total"):
            eq = np.array_equal(base[k], r[k], equal_nan=True) if base[k].dtype.kind == "f" else np.array_equal(base[k], r[k])
            assert eq, (name, flags, k)


def test_drop_in_single_trajectory_api(torch_gpu):
    """control/iLQRController.py:8-24, :26, :72, :84: constructor, attributes, in-place u, fresh x, ignored kwargs."""
    from ilqr_amd import CarModel, iLQR
    plant = CarModel(2, 2)
    ctl = iLQR(plant, N=40, max_iter=50, max_linesearch_iters=20, regu_init=100, min_regu=0.001, max_regu=10000)
    assert (ctl.nx, ctl.nu, ctl.ndx, ctl.N, ctl.max_iter, ctl.max_linesearch_iters, ctl.d_tol) == (5, 2, 5, 40, 50, 20, 1e-3)
    assert (ctl.num_ls_success, ctl.num_ls_fails, ctl.num_iters) == (0, 0, 0)
    g = load_golden("car_readme")
    u = g["u_init"][0].copy()
    x, u_ret = ctl.run_ilqr(g["x0"][0], u)
    assert u_ret is u and x.shape == (40, 5) and ctl.num_iters == 1
    assert np.allclose(x, g["x"][0], rtol=1e-9, atol=1e-10) and np.allclose(u, g["u"][0], rtol=1e-9, atol=1e-10)
    assert int(ctl.info.iters[0]) == 5
    # receding-horizon protocol of Driving.ipynb cell-2: re-solve from x[1] with the carried-over u
    m = load_golden("car_mpc50")
    x0 = m["x0"][0].copy(); u = m["u_init"][0].copy()
    for k in range(5):
        assert np.allclose(x0, m["x0"][k], rtol=1e-9, atol=1e-10)
        xk, u = ctl.run_ilqr(x0, u)
        x0 = xk[1]
    # LinAlgError where NumPy raises it
    with pytest.raises(np.linalg.LinAlgError):
        ctl.run_ilqr(g["x0"][0], np.full((39, 2), 1e155))
    with pytest.raises(ValueError):
        ctl.run_ilqr(g["x0"][0], np.zeros((38, 2)))


def test_model_interface_single_point(torch_gpu, oracle):
    """Model.discrete_dynamics/_cost_stage/_cost_final/stage/final/rollout/cost_trj/get_dims for one state."""
    from ilqr_amd import Acrobot2DModel, CarModel, QuadrotorQuatModel
    plant = CarModel(2.0, 2.0)
    assert plant.get_dims() == (5, 2, 5)
    x, u = np.array([1.0, -2.0, 0.3, 1.5, 0.1]), np.array([0.2, -0.1])
    e = oracle.eval_point("car", oracle.pack_car(2.0, 2.0), x, u)
    assert np.allclose(plant.discrete_dynamics(x, u), e["xn"], rtol=1e-12)
    assert np.isclose(plant._cost_stage(x, u), e["cs"], rtol=1e-12) and np.isclose(plant._cost_final(x), e["cf"], rtol=1e-12)
    for got, key in zip(plant.stage(x, u), ("l_x", "l_u", "l_xx", "l_ux", "l_uu", "f_x", "f_u")):
        assert np.allclose(got, e[key], rtol=1e-10, atol=1e-12), key
    Vx, Vxx = plant.final(x)
    assert np.allclose(Vx, e["V_x"], rtol=1e-10) and np.allclose(Vxx, e["V_xx"], rtol=1e-10)
    utr = np.random.default_rng(0).standard_normal((7, 2)) * 0.1
    xtr = plant.rollout(x, utr)
    assert xtr.shape == (8, 5) and np.allclose(xtr[1], plant.discrete_dynamics(x, utr[0]), rtol=1e-13)
    J = plant.cost_trj(xtr, utr)
    assert np.isclose(J, sum(plant._cost_stage(xtr[k], utr[k]) for k in range(7)) + plant._cost_final(xtr[-1]), rtol=1e-12)
    ac = Acrobot2DModel(np.diag([1, 1, .1, .1]), 100 * np.eye(4), np.array([[0.01]]), 0.05, 10)
    ac.set_references(np.array([np.pi, 0, 0, 0.0]), np.zeros(1))
    par = oracle.pack_quadratic(0.05, ac.Q, ac.Qf, ac.R, ac.xref, ac.uref)
    xa, ua = np.array([0.3, -0.2, 0.5, 0.1]), np.array([0.7])
    ea = oracle.eval_point("acrobot", par, xa, ua)
    assert np.allclose(ac.discrete_dynamics(xa, ua), ea["xn"], rtol=1e-12)
    assert np.isclose(ac._cost_stage(xa, ua, ac.xref, ac.uref), ea["cs"], rtol=1e-12)
    Q = np.diag([10.0] * 6 + [1.0] * 6)
    qq = QuadrotorQuatModel(Q, Q, np.eye(4) * 0.02, 0.01, 10)
    # any Q, Qf, R (Quadrotor_Quat.py:152), and `integration` other than "RK4" = Euler (:123-126)
    rng = np.random.default_rng(3)
    Qn, Qfn, Rn = Q + 0.3 * rng.standard_normal((12, 12)), Q + rng.standard_normal((12, 12)), 0.02 * np.eye(4) + 0.004 * rng.standard_normal((4, 4))
    qe = QuadrotorQuatModel(Qn, Qfn, Rn, 0.01, 10, integration="Euler")
    xr = np.zeros(13); xr[0] = 1.0; xr[6] = 1.0
    qe.set_references(xr, np.zeros(4))
    parq = oracle.pack_quadratic(0.01, Qn, Qfn, Rn, xr, np.zeros(4), euler=True)
    xq = 0.3 * rng.standard_normal(13); xq[:4] = [0.9, 0.1, -0.2, 0.3]
    uq = 1.9 + 0.3 * rng.standard_normal(4)
    eq = oracle.eval_point("quad_quat", parq, xq, uq)
    assert np.allclose(qe.discrete_dynamics(xq, uq), eq["xn"], rtol=1e-12)
    assert np.isclose(qe._cost_stage(xq, uq, qe.xref, qe.uref), eq["cs"], rtol=1e-11)
    assert np.isclose(qe._cost_final(xq, qe.xref), eq["cf"], rtol=1e-11)
    s = np.zeros(13); s[0] = 1; r = np.zeros(13); r[0] = 1; r[4] = 0.5
    assert qq.CalcDifferentialState(s, r).shape == (12,)


def test_mpc_advance(torch_gpu, oracle):
    """Receding-horizon step on device buffers: x0 <- f(x0, u[0]); u[:-1] <- u[1:] (QuadrotorControllers.py:72-76)."""
    import torch
    from ilqr_amd.batch import BatchSolver
    rng = np.random.default_rng(5)
    B, N = 300, 12
    c = _cfg(oracle, "quad_quat")
    x0, u0 = _inputs("quad_quat", rng, B, N)
    s = BatchSolver(3, c["par"], N, 1)
    xs = torch.as_tensor(x0).cuda().t().contiguous()
    us = torch.as_tensor(u0).cuda().permute(1, 2, 0).contiguous()
    s.mpc_advance(xs, us)
    un = us.permute(2, 0, 1).cpu().numpy(); xn = xs.t().cpu().numpy()
    assert np.array_equal(un[:, :-1], u0[:, 1:]) and np.array_equal(un[:, -1], u0[:, -1])
    for b in range(0, B, 37):
        e = oracle.eval_point("quad_quat", c["par"], x0[b], u0[b, 0])
        assert np.allclose(xn[b], e["xn"], rtol=1e-12, atol=1e-13)


def test_batch_mpc_follows_reference_closed_loops(torch_gpu):
    """Receding-horizon driver against the reference's own closed-loop sequences: Driving.ipynb cell-2 (re-solve from
    x_trj[1], u carried over) and the quadrotor protocol (plant step with u[0], shift u: QuadrotorControllers.py:185-188)."""
    from ilqr_amd.batch import BatchSolver
    from ilqr_amd.mpc import BatchMPC
    m = load_golden("car_mpc50")
    s = BatchSolver(0, m["par"], int(m["N"]), int(m["max_iter"]), int(m["max_ls"]), float(m["d_tol"]))
    log = BatchMPC(s, shift=False).run(m["x0"][:1], m["u_init"][:1], steps=50, keep_plans=True)
    xs = log["x"].cpu().numpy()[:, 0]
    assert rel_traj(xs[:50], m["x0"]).max() < 1e-7
    assert np.array_equal(log["iters"].cpu().numpy()[:, 0], m["iters"])
    assert rel_traj(log["u_plan"].cpu().numpy()[:, 0], m["u"]).max() < 1e-6

    q = load_golden("quad_quat_mpc")
    steps, ntraj = int(q["mpc_steps"]), int(q["mpc_traj"])
    s = BatchSolver(3, q["par"], int(q["N"]), int(q["max_iter"]), int(q["max_ls"]), float(q["d_tol"]))
    first = np.arange(ntraj) * steps
    log = BatchMPC(s, shift=True).run(q["x0"][first], q["u_init"][first], steps=steps)
    xs = log["x"].cpu().numpy()
    its = log["iters"].cpu().numpy()
    for b in range(ntraj):
        ref = q["x0"][b * steps:(b + 1) * steps]
        assert rel_traj(xs[:steps, b], ref).max() < 1e-5, b
        st = q["dstable"][b * steps:(b + 1) * steps] == 1
        assert np.array_equal(its[:, b][st], q["iters"][b * steps:(b + 1) * steps][st]), b


def test_batch_mpc_reseeds_all_zero_plans_per_trajectory(torch_gpu, oracle):
    """QuadrotorControllers.py:67-70 for a batch: a loop whose plan is all zeros starts from a random plan, the others
    keep theirs; the loops then run as if they had been given those plans."""
    import torch
    from ilqr_amd.batch import BatchSolver
    from ilqr_amd.mpc import BatchMPC
    rng = np.random.default_rng(9)
    B, N, steps = 12, 20, 3
    c = _cfg(oracle, "quad_quat")
    x0, u0 = _inputs("quad_quat", rng, B, N)
    zero = np.arange(B) % 3 == 0
    u0[zero] = 0.0
    s = BatchSolver(3, c["par"], N, 5, 10, 5e-4)
    gen = torch.Generator(device="cuda"); gen.manual_seed(77)
    log = BatchMPC(s, shift=True, reseed_std=2.0, generator=gen).run(x0, u0, steps=steps, keep_plans=True)
    gen2 = torch.Generator(device="cuda"); gen2.manual_seed(77)
    noise = (torch.randn((N - 1, 4, B), dtype=torch.float64, device="cuda", generator=gen2) * 2.0).permute(2, 0, 1).cpu().numpy()
    assert np.abs(noise[zero]).min() > 0 and abs(noise[zero].std() - 2.0) < 0.3
    u1 = np.where(zero[:, None, None], noise, u0)
    ref = BatchMPC(s, shift=True).run(x0, u1, steps=steps, keep_plans=True)
    for k in ("x", "u", "J", "iters", "status", "u_plan"):
        a, b = log[k].cpu().numpy(), ref[k].cpu().numpy()
        assert np.array_equal(a, b, equal_nan=True) if a.dtype.kind == "f" else np.array_equal(a, b), k


def test_parts_on_internal_streams_do_not_change_results(torch_gpu, oracle):
    """A batch large enough to be cut into concurrently running parts (odd size, stride > B, traces on) gives
    bit-identical results to the same batch on the caller's stream alone, and leaves the padding untouched."""
    import torch
    from ilqr_amd import _lib
    from ilqr_amd.batch import BatchSolver
    rng = np.random.default_rng(11)
    B, stride, N = 70001, 70080, 40
    c = _cfg(oracle, "car")
    x0, u0 = _car_box(rng, B), rng.standard_normal((B, N - 1, 2)) * 1e-4
    res = []
    for flags in (0, _lib.FLAG_SINGLE_STREAM):
        s = BatchSolver(0, c["par"], N, 50, 20, 1e-3, trace_len=8, flags=flags)
        x0s = torch.zeros((5, stride), dtype=torch.float64, device="cuda"); x0s[:, :B] = torch.as_tensor(x0).cuda().t()
        us = torch.full((N - 1, 2, stride), 7.0, dtype=torch.float64, device="cuda")
        us[:, :, :B] = torch.as_tensor(u0).cuda().permute(1, 2, 0)
        xs = torch.full((N, 5, stride), -3.0, dtype=torch.float64, device="cuda")
        out = s.alloc_outputs(stride)
        for v in out.values():
            v.fill_(0)
        s.solve_soa(x0s, us, xs, out, B=B)
        torch.cuda.synchronize()
        assert (us[:, :, B:] == 7.0).all() and (xs[:, :, B:] == -3.0).all()
        assert (out["iters"][B:] == 0).all() and (out["ls_evals"][:, B:] == 0).all()
        res.append(dict(x=xs[:, :, :B].cpu().numpy(), u=us[:, :, :B].cpu().numpy(),
                        **{k: v[..., :B].cpu().numpy() for k, v in out.items()}))
    a, b = res
    for k in a:
        same = np.array_equal(a[k], b[k], equal_nan=True) if a[k].dtype.kind == "f" else np.array_equal(a[k], b[k])
        assert same, k
    assert a["iters"].min() >= 1 and set(np.unique(a["status"])) <= {0, 1, 2, 3}
    # and the per-phase timing facility reports every phase of the solve
    s = BatchSolver(0, c["par"], N, 50, 20, 1e-3, flags=_lib.FLAG_TIMING)
    s.solve(x0[:4096], u0[:4096])
    t = s.last_timing()
    assert t["iterations"] >= 5 and t["backward"][1] == t["iterations"] and t["backward"][0] > 0
    # (4096 trajectories are in the latency-bound regime from the start: no separate first-candidate launch)
    assert t["init"][1] == 1 and t["forward1"][1] == 0 and t["backtrack"][1] >= 1 and t["commit"][1] >= 1


@pytest.mark.parametrize("kind", ["quad_ea", "quad_quat"])
def test_ill_conditioned_regime_matches_checker_statistically(torch_gpu, oracle, kind):
    """Far from hover most 12-state solves are ill conditioned: every stage takes dozens of regularisation
    doublings and individual trajectories are not reproducible (not even by the reference under a 1e-14
    perturbation).  The population must still behave like the checker's when the checker applies the kernels' own
    PSD rule (LDL', `rule="ldl"`): same exit-reason histogram within a few counts, same amount of regularisation,
    and mostly identical iteration counts.  (With LAPACK's rule the checker regularises longer in this regime:
    tests/test_population_reference.py holds the numbers for the car.)"""
    rng = np.random.default_rng(21)
    B, N = 128, 100
    c = dict(_cfg(oracle, kind), N=N)
    if kind == "quad_ea":
        x0 = np.zeros((B, 12)); x0[:, 2] = 0.8; x0[:, 3:6] = 0.5 * rng.standard_normal((B, 3))
    else:
        x0 = np.zeros((B, 13)); x0[:, 6] = 0.8
        q = np.column_stack([np.ones(B), 0.25 * rng.standard_normal((B, 3))])
        x0[:, :4] = q / np.linalg.norm(q, axis=1, keepdims=True)
    u0 = 2.0 * rng.standard_normal((B, N - 1, 4))
    got = _solve(c, x0, u0)
    ref = oracle.solve_batch(kind, c["par"], N, c["max_iter"], c["max_ls"], c["d_tol"], x0, u0, rule="ldl")["info"]
    assert ref["regu_rounds"].mean() > 1000, "the case is meant to be ill conditioned"
    hg, hr = np.bincount(got["status"], minlength=4), np.bincount(ref["status"], minlength=4)
    assert np.abs(hg - hr).max() <= max(4, B // 20), (hg, hr)
    ratio = got["regu"].mean() / ref["regu_rounds"].mean()
    assert 0.8 < ratio < 1.25, ratio
    assert (got["iters"] == ref["iters"]).mean() > 0.85


def test_car_population_statistics_match_checker(torch_gpu, oracle):
    """Headline batch, all trajectories (a quarter of them are ill conditioned and not reproducible one by one):
    well-conditioned solves agree decision for decision with the checker under either PSD rule, and the population
    as a whole exits for the same reasons, regularises as much and iterates as long as the checker's under the
    kernels' rule."""
    rng = np.random.default_rng(1234)
    B = 4096
    c = _cfg(oracle, "car")
    x0, u0 = _car_box(rng, B), rng.standard_normal((B, 39, 2)) * 1e-4
    got = _solve(c, x0, u0)
    for rule in ("lapack", "ldl"):
        ref = oracle.solve_batch("car", c["par"], 40, 50, 20, 1e-3, x0, u0, rule=rule)["info"]
        calm = ref["max_abs_q"] < 1e5
        assert calm.mean() > 0.6
        same = (got["iters"] == ref["iters"]) & (got["regu"] == ref["regu_rounds"]) & (got["status"] == ref["status"])
        # (a decision that sits on its threshold to the last bit may fall the other way: the checker's matrix products
        # round like NumPy's, the kernels' matrix-vector products do not; one or two solves in thousands)
        assert (~same[calm]).sum() <= max(2, calm.sum() // 1000), (rule, (~same[calm]).sum())
        assert same.mean() > 0.88, (rule, same.mean())
    hg, hr = np.bincount(got["status"], minlength=4), np.bincount(ref["status"], minlength=4)
    assert np.abs(hg - hr).max() <= B // 64, (hg, hr)
    assert 0.9 < got["regu"].mean() / ref["regu_rounds"].mean() < 1.1
    assert abs(got["iters"].mean() - ref["iters"].mean()) < 0.15
    assert (got["iters"] == ref["iters"]).mean() > 0.95


def test_full_size_batch_properties(torch_gpu, oracle):
    """The bench workload at its full size (2^20 car solves, three concurrently running parts), checked on the
    device through properties that need no second solver: x is bit for bit the rollout of u from x0; a sub-batch
    solved alone gives the same bits (trajectories never interact, whatever part or slot they ran in); exit
    statistics match the CPU checker's on a sample of the same batch."""
    import torch
    from bench import car_inputs
    from ilqr_amd.batch import BatchSolver
    B = 1 << 20
    c = _cfg(oracle, "car")
    x0, u0 = car_inputs(B, 0)
    s = BatchSolver(0, c["par"], 40, 50, 20, 1e-3)
    x0s = torch.as_tensor(np.ascontiguousarray(x0.T), device="cuda")
    u0s = torch.as_tensor(np.ascontiguousarray(u0.transpose(1, 2, 0)), device="cuda")
    us = u0s.clone()
    xs, us, out = s.solve_soa(x0s, us)
    torch.cuda.synchronize()
    st, it = out["status"], out["iters"]
    assert int(it.min()) >= 1 and int(it.max()) <= 50 and int(st.min()) >= 0 and int(st.max()) <= 3
    ok = st != 3
    assert torch.equal(xs[0], x0s)
    # rollout property, on the device, in the strided layout
    xr = torch.empty_like(xs); Jr = torch.empty(B, dtype=torch.float64, device="cuda")
    import ctypes as C
    from ilqr_amd.batch import _ptr
    pr = s._problem(B, B)
    s._check(s.lib.ilqr_rollout_cost(C.byref(pr), _ptr(x0s), _ptr(us), _ptr(xr), _ptr(Jr), s._stream()))
    torch.cuda.synchronize()
    assert torch.equal(xr[:, :, ok], xs[:, :, ok])
    # a sub-batch alone: same bits
    idx = torch.arange(0, B, 257, device="cuda")[:4096]
    u2 = u0s[:, :, idx].contiguous()
    x2, u2, o2 = s.solve_soa(x0s[:, idx].contiguous(), u2)
    torch.cuda.synchronize()
    assert torch.equal(x2, xs[:, :, idx]) and torch.equal(u2, us[:, :, idx])
    assert torch.equal(o2["iters"], it[idx]) and torch.equal(o2["J"].nan_to_num(), out["J"][idx].nan_to_num())
    # exit statistics against the checker on the first 8192 trajectories
    ref = oracle.solve_batch("car", c["par"], 40, 50, 20, 1e-3, x0[:8192], u0[:8192], rule="ldl")["info"]
    g = np.bincount(st[:8192].cpu().numpy(), minlength=4) / 8192.0
    r = np.bincount(ref["status"], minlength=4) / 8192.0
    assert np.abs(g - r).max() < 0.01, (g, r)
    assert abs(float(it[:8192].double().mean()) - ref["iters"].mean()) < 0.3


def test_car_horizon_200_stress_batch(torch_gpu, oracle):
    """The contract's stress configuration: CarModel N = 200, max_linesearch_iters = 20 (candidates evaluated side by
    side), 65 536 trajectories.  At this horizon no solve is reproducible one by one, not by the reference itself
    (fixtures car_N200, car_circle_N200: 0 of 24 decision-stable), so the batch is held to what does not need a second
    solver -- x is bit for bit the rollout of the returned u, a sub-batch solved alone gives the same bits -- and to
    the checker's statistics under the kernels' PSD rule on a sample."""
    import torch
    from bench import car_inputs
    from ilqr_amd.batch import BatchSolver
    B, N = 1 << 16, 200
    c = _cfg(oracle, "car")
    x0, u0 = car_inputs(B, 5, N)
    s = BatchSolver(0, c["par"], N, 50, 20, 1e-3)
    r = s.solve(x0, u0)
    st, it = r.status.cpu().numpy(), r.iters.cpu().numpy()
    assert it.min() >= 1 and it.max() <= 50 and st.min() >= 0 and st.max() <= 3
    ok = torch.as_tensor(st != 3, device=r.x.device)
    xr, Jr = s.rollout_cost(x0, r.u)
    assert torch.equal(xr[ok], r.x[ok])
    idx = np.arange(0, B, 131)[:384]
    r2 = s.solve(x0[idx], u0[idx])
    assert torch.equal(r2.x, r.x[idx]) and torch.equal(r2.u, r.u[idx]) and torch.equal(r2.iters, r.iters[idx])
    n = 512
    ref = oracle.solve_batch("car", c["par"], N, 50, 20, 1e-3, x0[:n], u0[:n], rule="ldl")["info"]
    g = np.bincount(st[:n], minlength=4) / n
    h = np.bincount(ref["status"], minlength=4) / n
    assert np.abs(g - h).max() < 0.05, (g, h)
    assert abs(it[:n].mean() - ref["iters"].mean()) < 0.05 * ref["iters"].mean() + 0.5
    ratio = r.regu_rounds.cpu().numpy()[:n].mean() / ref["regu_rounds"].mean()
    assert 0.8 < ratio < 1.25, ratio
    # The reference's own solves at this horizon (status-class fixtures) run on to max_iter: LAPACK's eigenvalues keep
    # calling the regularised Q indefinite where the kernels' LDL' rule, which reads pivots below 1e-13 of their terms
    # as zero, lets the sweep through, after which the expected reduction falls below 1 and the solve stops.  The
    # checker shows both behaviours (rule="lapack": 7 of 8 and 16 of 16 at max_iter; rule="ldl": none and 2 of 16);
    # the kernels must show the second.
    for name in ("car_N200", "car_circle_N200"):
        q = load_golden(name)
        rq = _solve(q, q["x0"], q["u_init"])
        ck = oracle.solve_batch("car", q["par"], N, 50, 20, 1e-3, q["x0"], q["u_init"], rule="ldl")["info"]
        assert (q["status"] == 2).mean() >= 0.8
        assert abs((rq["status"] == 2).mean() - (ck["status"] == 2).mean()) <= 0.25, name


def test_quaternion_mpc_loop_at_contract_length(torch_gpu, oracle):
    """The contract's receding-horizon configuration at full length: quaternion quadrotor, N = 80, 50 steps (solve,
    plant step with u[0], shift u: QuadrotorControllers.py:173-191), 192 loops near hover.  The reference does not
    reproduce its own decisions at this horizon (fixture quad_quat_mpc80: 0 of 50 steps decision-stable), so the loops are held to
    properties: every step's plan is bit for bit the rollout of its controls from the state the loop was in, the
    state the loop moves to is the plant's step from there, loops do not depend on their batch-mates (a subset run
    alone gives the same bits), and the exit classes of the first solves match the checker's under the kernels'
    PSD rule."""
    import torch
    from ilqr_amd.batch import BatchSolver
    from ilqr_amd.mpc import BatchMPC
    rng = np.random.default_rng(44)
    B, N, steps = 192, 80, 50
    c = dict(_cfg(oracle, "quad_quat"), N=N)
    x0, u0 = _inputs("quad_quat", rng, B, N)
    x0[:, 4:6] = 0.2 * rng.standard_normal((B, 2))
    s = BatchSolver(3, c["par"], N, c["max_iter"], c["max_ls"], c["d_tol"])
    log = BatchMPC(s, shift=True).run(x0, u0, steps=steps, keep_plans=True)
    st = log["status"].cpu().numpy()
    assert st.shape == (steps, B) and st.min() >= 0 and st.max() <= 3
    good = torch.as_tensor((st != 3).all(axis=0), device="cuda")        # loops no solve of which ended in a LinAlgError
    assert int(good.sum()) > B // 4
    for t in (0, 1, 17, steps - 1):
        xr, _ = s.rollout_cost(log["x"][t], log["u_plan"][t])
        ok = torch.as_tensor(st[t] != 3, device="cuda")
        assert torch.equal(xr[ok], log["x_plan"][t][ok]), t
        # the next state is the plant's own step with the first control of the plan (= x_plan[1] of a rollout)
        assert torch.equal(log["x"][t + 1][ok], xr[ok][:, 1]), t
        assert torch.equal(log["u"][t], log["u_plan"][t][:, 0]), t
    sub = np.arange(0, B, 7)
    log2 = BatchMPC(s, shift=True).run(x0[sub], u0[sub], steps=steps)
    for k in ("x", "u", "iters", "status"):
        a, b = log[k][:, sub], log2[k]
        assert torch.equal(a.nan_to_num(), b.nan_to_num()) if a.dtype.is_floating_point else torch.equal(a, b), k
    ref = oracle.solve_batch("quad_quat", c["par"], N, c["max_iter"], c["max_ls"], c["d_tol"], x0, u0, rule="ldl")["info"]
    g, h = np.bincount(st[0], minlength=4), np.bincount(ref["status"], minlength=4)
    assert np.abs(g - h).max() <= max(6, B // 12), (g, h)


def test_lapack_eigvals_flag_reproduces_the_reference_at_long_horizons(torch_gpu, oracle):
    """ILQR_FLAG_LAPACK_EIGVALS at the contract's stress horizon.  The reference's Car N = 200 solves run on to
    max_iter because LAPACK keeps finding a negative eigenvalue in the regularised Q (fixtures car_N200,
    car_circle_N200: 22 of 24 end at max_iter); with the flag the kernels do the same, and match the checker (which
    restates LAPACK bit for bit) in exit classes, iteration counts and amount of regularisation."""
    from ilqr_amd import _lib
    for name in ("car_N200", "car_circle_N200"):
        q = load_golden(name)
        r = _solve(q, q["x0"], q["u_init"], flags=_lib.FLAG_LAPACK_EIGVALS)
        ck = oracle.solve_batch("car", q["par"], 200, 50, 20, 1e-3, q["x0"], q["u_init"])["info"]
        assert (q["status"] == 2).mean() >= 0.8 and (r["status"] == 2).mean() >= 0.75, (name, r["status"])
        assert abs((r["status"] == 2).mean() - (ck["status"] == 2).mean()) <= 0.15, name
        assert abs(r["iters"].mean() - q["iters"].mean()) <= 0.15 * q["iters"].mean(), (name, r["iters"].mean())
        # (8 and 16 solves of a regime in which every solve is chaotic and the rounds per solve range from 24 000 to
        # 1 500 000 in the reference itself: medians, to a factor of three)
        med = np.median(r["regu"])
        ratio, ratio_ck = med / np.median(q["regu_rounds"]), med / np.median(ck["regu_rounds"])
        print(name, "median regularisation rounds, GPU / reference %.2f, GPU / checker %.2f" % (ratio, ratio_ck))
        assert 1 / 3 < ratio < 3 and 1 / 3 < ratio_ck < 3, (name, ratio, ratio_ck)
    # decisions on reproducible solves are those of the default rule and of the reference
    for name in ("car_readme", "car_box256", "car_circle_N60", "acrobot_N40"):
        g = load_golden(name)
        r = _solve(g, g["x0"], g["u_init"], flags=_lib.FLAG_LAPACK_EIGVALS)
        n = compare_to_golden(g, r["x"], r["u"], r["J"], r["iters"], r["status"], r["ls"], r["acc"], r["regu"], J0=r["J0"],
                              label=name + " (LAPACK's rule)")
        assert n >= min_compared(name, g)
        assert n.within_1e8 >= min_within_1e8(name, g), (name, n.within_1e8, n.stable)


def test_lapack_eigvals_flag_on_the_quadrotor_fixtures(torch_gpu, oracle):
    """ILQR_FLAG_LAPACK_EIGVALS in the 16-lanes-per-trajectory kernel.  On the reproducible fixtures the decisions are
    the reference's one by one, as with the default rule; on the contract's configurations 3 and 4 far from hover
    (status-class fixtures, 16 solves each) the exits are compared with the reference's as classes, and a per-trajectory
    parameter block with the flag gives the bits of the shared block."""
    from ilqr_amd import _lib
    from ilqr_amd.batch import BatchSolver
    for name in ("quad_ea_N30", "quad_quat_N30", "quad_ea_hover_N40", "quad_quat_hover_N40", "quad_quat_fullw_N30",
                 "quad_ea_euler_N30", "quad_ea_near_N16", "quad_quat_near_N16"):
        g = load_golden(name)
        r = _solve(g, g["x0"], g["u_init"], flags=_lib.FLAG_LAPACK_EIGVALS)
        n = compare_to_golden(g, r["x"], r["u"], r["J"], r["iters"], r["status"], r["ls"], r["acc"], r["regu"], J0=r["J0"],
                              label=name + " (LAPACK's rule)")
        assert n >= min_compared(name, g), (name, n)
        assert n.within_1e8 >= min_within_1e8(name, g), (name, n.within_1e8, n.stable)
    for name in ("quad_ea_N100", "quad_quat_N80"):
        g = load_golden(name)
        r = _solve(g, g["x0"], g["u_init"], flags=_lib.FLAG_LAPACK_EIGVALS)
        same = (r["status"] == g["status"]).mean()
        print(name, "with LAPACK's rule: exits", np.bincount(r["status"], minlength=4), "reference",
              np.bincount(g["status"], minlength=4), "same exit on %.2f" % same, "mean iterations %.1f / %.1f"
              % (r["iters"].mean(), g["iters"].mean()))
        assert same >= 0.6, (name, r["status"], g["status"])
        assert abs(r["iters"].mean() - g["iters"].mean()) <= 0.3 * g["iters"].mean(), name
    # a parameter block per trajectory with the flag: the same bits as the shared block (both quadrotors; fixtures that
    # regularise, so that the scans of the LAPACK path run)
    for name, mid in (("quad_ea_N30", 2), ("quad_quat_fullw_N30", 3), ("quad_quat_hover_N40", 3)):
        g = load_golden(name)
        B = len(g["x0"])
        sp = (int(g["N"]), int(g["max_iter"]), int(g["max_ls"]), float(g["d_tol"]))
        par = np.asarray(g["par"], dtype=np.float64)
        a = BatchSolver(mid, par, *sp, flags=_lib.FLAG_LAPACK_EIGVALS, cost_N=cost_horizon(g)).solve(g["x0"], g["u_init"])
        b = BatchSolver(mid, np.tile(par, (B, 1)), *sp, flags=_lib.FLAG_LAPACK_EIGVALS,
                        cost_N=cost_horizon(g)).solve(g["x0"], g["u_init"])
        for k in ("x", "u", "J", "iters", "status", "regu_rounds", "ls_total", "max_abs_q"):
            assert torch_equal(getattr(a, k), getattr(b, k)), (name, k)


def test_line_search_counters_and_plant_horizon(torch_gpu):
    """iLQRController.py:57-58, :69-74: after run_ilqr, num_ls_success / num_ls_fails describe the last iteration that
    reached the line search (a call in which none did leaves them alone).  acrobot.py:139: the plant's cost_trj sums over
    the horizon the PLANT was built with; fixture acrobot_modelN25_N40 is the reference run that way."""
    from ilqr_amd import Acrobot2DModel, CarModel, iLQR
    g = load_golden("car_box256")
    ctl = iLQR(CarModel(2, 2), N=40, max_iter=50, max_linesearch_iters=20)
    n = 64
    u = g["u_init"][:n].copy()
    ctl.run_ilqr(g["x0"][:n], u)
    it = g["iters"][:n]
    searched = g["ls_evals"][:n] > 0
    for b in range(n):
        if not g["dstable"][b]:
            continue
        ks = np.nonzero(searched[b, :it[b]])[0]
        want = (0, 0) if len(ks) == 0 else (int(g["accepted"][b, ks[-1]]), int(not g["accepted"][b, ks[-1]]))
        assert (int(ctl.num_ls_success[b]), int(ctl.num_ls_fails[b])) == want, b
    # single trajectory, no trace requested
    one = iLQR(CarModel(2, 2), N=40, max_iter=50, max_linesearch_iters=20)
    r = load_golden("car_readme")
    one.run_ilqr(r["x0"][0], r["u_init"][0].copy())
    assert (one.num_ls_success, one.num_ls_fails) == (1, 0)
    m1 = load_golden("car_maxls1")           # evaluation number max_ls is never accepted (:66-67)
    one1 = iLQR(CarModel(2, 2), N=40, max_iter=int(m1["max_iter"]), max_linesearch_iters=1)
    one1.run_ilqr(m1["x0"][0], m1["u_init"][0].copy())
    assert (one1.num_ls_success, one1.num_ls_fails) == (0, 1)
    # plant horizon 25, solver horizon 40
    h = load_golden("acrobot_modelN25_N40")
    ac = Acrobot2DModel(np.diag([1, 1, .1, .1]), 100 * np.eye(4), np.array([[0.01]]), 0.05, 25)
    ac.set_references(np.array([np.pi, 0, 0, 0.0]), np.zeros(1))
    c = iLQR(ac, N=40, max_iter=int(h["max_iter"]), max_linesearch_iters=int(h["max_ls"]), d_tol=float(h["d_tol"]))
    uu = h["u_init"].copy()
    xx, _ = c.run_ilqr(h["x0"], uu)
    ok = h["dstable"] == 1
    assert ok.sum() >= 8
    assert np.array_equal(c.info.iters.cpu().numpy()[ok], h["iters"][ok])
    st = h["stable"] == 1      # (also reproduces its cost and trajectory to 1e-8 under a 1e-14 perturbation)
    assert st.sum() >= 8
    assert np.allclose(c.info.J.cpu().numpy()[st], h["J_last"][st], rtol=1e-6, equal_nan=True)
    assert np.allclose(xx[st], h["x"][st], rtol=1e-6, atol=1e-7, equal_nan=True)
    # cost_trj of the plant itself: 24 stage costs and the final cost of the last state
    J = ac.cost_trj(h["x"][0], h["u"][0])
    want = sum(ac._cost_stage(h["x"][0][k], h["u"][0][k]) for k in range(24)) + ac._cost_final(h["x"][0][-1])
    assert np.isclose(J, want, rtol=1e-12)
    with pytest.raises(IndexError):      # the reference indexes past the trajectory
        iLQR(Acrobot2DModel(ac.Q, ac.Qf, ac.R, 0.05, 50), N=40, max_iter=2).run_ilqr(h["x0"][0], h["u_init"][0].copy())


def test_drop_in_class_with_lapacks_eigenvalue_rule(torch_gpu):
    """iLQR(..., eigvals="lapack"): the regularisation loop's test (iLQRController.py:110) decided the way NumPy
    decides it.  On the 256 headline solves the reference recorded (car_box256.npz) the exits then are the reference's
    to one solve (15 run to max_iter, none ends in a LinAlgError; the default rule lets four of those stop early and
    raises two LinAlgErrors), and under either rule nine solves in ten take the reference's iteration count, exit and
    number of regularisation rounds -- the rest are solves the reference does not reproduce under a 1e-14 perturbation
    of its own inputs either."""
    from ilqr_amd import CarModel, iLQR
    g = load_golden("car_box256")
    ref_exits = np.bincount(g["status"], minlength=4)
    for mode, tol in (("lapack", 1), ("ldl", 4)):
        ctl = iLQR(CarModel(2, 2), N=40, max_iter=50, max_linesearch_iters=20, eigvals=mode)
        ctl.run_ilqr(g["x0"], g["u_init"].copy())
        it, st = ctl.info.iters.cpu().numpy(), ctl.info.status.cpu().numpy()
        same = (it == g["iters"]) & (st == g["status"]) & (ctl.info.regu_rounds.cpu().numpy() == g["regu_rounds"])
        exits = np.bincount(st, minlength=4)
        assert np.abs(exits - ref_exits).max() <= tol, (mode, exits, ref_exits)
        assert same.sum() >= 225 and same[g["dstable"].astype(bool)].all(), (mode, int(same.sum()))
    with pytest.raises(ValueError, match="eigvals"):
        iLQR(CarModel(2, 2), N=40, max_iter=50, eigvals="exact")
