"""The contract's second and third configurations at the size one GPU holds of them: Acrobot N = 100 with 65 536 random
starts, and the Euler-angle quadrotor N = 100 with 131 072 trajectories (one GPU's eighth of the 1M batch) from the
contract's starts far from hover.  Checked through properties that need no second solver at that size -- the returned
x is bit for bit the rollout of the returned u from x0 (control/iLQRController.py:63, :71-72), a sub-batch solved alone
gives the same bits (trajectories never interact, whatever part of the batch or slot they ran in) -- and against the
CPU checker's exit statistics on a sample of the same batch."""
import ctypes as C

import numpy as np
import pytest

from test_gpu_parity import _cfg, _inputs, torch_gpu   # noqa: F401

pytestmark = pytest.mark.gpu
MID = dict(acrobot=1, quad_ea=2)


@pytest.mark.parametrize("kind,B,N,sample", [("acrobot", 65536, 100, 1024), ("quad_ea", 131072, 100, 192)])
def test_configuration_at_full_size(torch_gpu, oracle, kind, B, N, sample):
    import torch
    from ilqr_amd.batch import BatchSolver, _ptr
    c = dict(_cfg(oracle, kind), N=N)
    rng = np.random.default_rng(dict(acrobot=2, quad_ea=3)[kind])
    x0, u0 = _inputs(kind, rng, B, N)
    if kind == "quad_ea":
        u0 = 2.0 * rng.standard_normal(u0.shape)        # the contract's initial controls
    s = BatchSolver(MID[kind], c["par"], N, c["max_iter"], c["max_ls"], c["d_tol"])
    x0s = torch.as_tensor(np.ascontiguousarray(x0.T), device="cuda")
    u0s = torch.as_tensor(np.ascontiguousarray(u0.transpose(1, 2, 0)), device="cuda")
    us = u0s.clone()
    xs, us, out = s.solve_soa(x0s, us)
    torch.cuda.synchronize()
    st, it = out["status"], out["iters"]
    assert int(it.min()) >= 1 and int(it.max()) <= c["max_iter"] and int(st.min()) >= 0 and int(st.max()) <= 3
    assert torch.equal(xs[0], x0s)
    # x is the rollout of u (a solve that ended in a LinAlgError keeps the trajectory it came in with: also a rollout)
    xr = torch.empty_like(xs)
    Jr = torch.empty(B, dtype=torch.float64, device="cuda")
    pr = s._problem(B, B)
    s._check(s.lib.ilqr_rollout_cost(C.byref(pr), _ptr(x0s), _ptr(us), _ptr(xr), _ptr(Jr), s._stream()))
    torch.cuda.synchronize()
    assert torch.equal(xr.nan_to_num(), xs.nan_to_num())
    # a sub-batch alone: same bits, same counts
    idx = torch.arange(0, B, 61, device="cuda")[:1024]
    u2 = u0s[:, :, idx].contiguous()
    x2, u2, o2 = s.solve_soa(x0s[:, idx].contiguous(), u2)
    torch.cuda.synchronize()
    assert torch.equal(x2.nan_to_num(), xs[:, :, idx].nan_to_num()) and torch.equal(u2.nan_to_num(), us[:, :, idx].nan_to_num())
    for k in ("iters", "status", "regu_rounds", "ls_total", "replayed"):
        assert torch.equal(o2[k], out[k][idx]), k
    assert torch.equal(o2["J"].nan_to_num(), out["J"][idx].nan_to_num())
    # exit statistics against the checker (the kernels' PSD rule) on the first `sample` trajectories
    ref = oracle.solve_batch(kind, c["par"], N, c["max_iter"], c["max_ls"], c["d_tol"], x0[:sample], u0[:sample],
                             rule="ldl")["info"]
    g = np.bincount(st[:sample].cpu().numpy(), minlength=4) / float(sample)
    r = np.bincount(ref["status"], minlength=4) / float(sample)
    print(kind, "exits GPU", g, "checker", r, "mean iterations", float(it[:sample].double().mean()), ref["iters"].mean(),
          "replayed iterations", int(out["replayed"].sum()), "of", int(it.sum()))
    assert np.abs(g - r).max() < (0.02 if kind == "acrobot" else 0.08), (g, r)
    assert abs(float(it[:sample].double().mean()) - ref["iters"].mean()) < (0.3 if kind == "acrobot" else 1.5)


def _zyx_quaternions(rng, B, std):
    """Unit quaternions [w, x, y, z] from ZYX Euler angles std * N(0, 1): the contract's starts of configuration 4."""
    from scipy.spatial.transform import Rotation
    q = Rotation.from_euler("ZYX", std * rng.standard_normal((B, 3))).as_quat()      # [x, y, z, w]
    return np.column_stack([q[:, 3], q[:, 0], q[:, 1], q[:, 2]])


def test_configuration_4_at_contract_size(torch_gpu, oracle):
    """Quaternion quadrotor, N = 80, the 50-step receding-horizon loop (solve, plant step with u[0], u[:-1] = u[1:];
    control/QuadrotorControllers.py:173-191) for 65 536 loops from the contract's starts: attitude = ZYX Euler angles
    0.5 N(0, 1), initial plan 2 N(0, 1), seed 4.  The reference does not converge from there (exits of its own 1 024
    solves: 16 / 850 / 158), so the loop is held to properties -- every state is the plant's own step from the state
    and control logged before it, a sub-batch alone gives the same bits at every step -- and its first solve to the
    CPU checker's exit classes on a 256 sample, under the kernels' default rule and under LAPACK's."""
    import torch
    from ilqr_amd import _lib
    from ilqr_amd.batch import BatchSolver
    from ilqr_amd.mpc import BatchMPC
    rng = np.random.default_rng(4)
    B, N, steps = 65536, 80, 50
    c = dict(_cfg(oracle, "quad_quat"), N=N)
    x0 = np.zeros((B, 13))
    x0[:, 6] = 0.8
    x0[:, :4] = _zyx_quaternions(rng, B, 0.5)
    u0 = 2.0 * rng.standard_normal((B, N - 1, 4))
    s = BatchSolver(3, c["par"], N, c["max_iter"], c["max_ls"], c["d_tol"])
    log = BatchMPC(s, shift=True).run(x0, u0, steps=steps)
    st, it = log["status"], log["iters"]
    assert st.shape == (steps, B) and int(st.min()) >= 0 and int(st.max()) <= 3 and int(it.max()) <= c["max_iter"]
    # the loop's states: x[t + 1] = discrete_dynamics(x[t], u[t]) for every loop and step, bit for bit
    for t in (0, 1, 24, steps - 1):
        xn = s.model_eval(log["x"][t], log["u"][t], what=("xn",))["xn"]
        assert torch.equal(xn.nan_to_num(), log["x"][t + 1].nan_to_num()), t
    # loops do not depend on their batch-mates
    sub = np.arange(0, B, 257)
    log2 = BatchMPC(s, shift=True).run(x0[sub], u0[sub], steps=steps)
    subd = torch.as_tensor(sub, device="cuda")
    for k in ("x", "u", "J", "iters", "status"):
        a, b = log[k][:, subd], log2[k]
        assert torch.equal(a.nan_to_num(), b.nan_to_num()) if a.dtype.is_floating_point else torch.equal(a, b), k
    # the first solve of the loop against the checker, both rules, 256 loops
    n = 256
    for rule, flags, slack in (("ldl", 0, 0.08), ("lapack", _lib.FLAG_LAPACK_EIGVALS, 0.1)):
        ref = oracle.solve_batch("quad_quat", c["par"], N, c["max_iter"], c["max_ls"], c["d_tol"], x0[:n], u0[:n],
                                 rule=rule)["info"]
        if flags:
            r = BatchSolver(3, c["par"], N, c["max_iter"], c["max_ls"], c["d_tol"], flags=flags).solve(x0[:n], u0[:n])
            g_st, g_it = r.status.cpu().numpy(), r.iters.cpu().numpy()
        else:
            g_st, g_it = st[0, :n].cpu().numpy(), it[0, :n].cpu().numpy()
        g = np.bincount(g_st, minlength=4) / float(n)
        h = np.bincount(ref["status"], minlength=4) / float(n)
        print("configuration 4, rule", rule, "exits GPU", g, "checker", h, "mean iterations", g_it.mean(), ref["iters"].mean())
        assert np.abs(g - h).max() <= slack, (rule, g, h)
        assert abs(g_it.mean() - ref["iters"].mean()) <= 0.1 * ref["iters"].mean() + 0.5, rule
    print("configuration 4: exits over all 50 x 65 536 solves", torch.bincount(st.flatten(), minlength=4).cpu().tolist())


def test_configuration_5_at_one_gpus_share(torch_gpu, oracle):
    """Car N = 200, max_linesearch_iters = 20, 2^19 trajectories (one GPU's eighth of the contract's 2^22) from the
    headline box of starts.  Properties at full size under the default rule -- x is bit for bit the rollout of u, a
    sub-batch alone gives the same bits -- exit fractions against the checker with the same rule on a sample, and the
    reference's own population of this configuration (tests/golden/car_n200_population_ref.npz) under
    ILQR_FLAG_LAPACK_EIGVALS, the rule that decides the exit class at this horizon."""
    import torch
    from bench import car_inputs
    from conftest import load_golden
    from ilqr_amd import _lib
    from ilqr_amd.batch import BatchSolver, _ptr
    B, N = 1 << 19, 200
    par = oracle.pack_car(2.0, 2.0)
    x0, u0 = car_inputs(B, 5, N)
    s = BatchSolver(0, par, N, 50, 20, 1e-3)
    x0s = torch.as_tensor(np.ascontiguousarray(x0.T), device="cuda")
    u0s = torch.as_tensor(np.ascontiguousarray(u0.transpose(1, 2, 0)), device="cuda")
    us = u0s.clone()
    xs, us, out = s.solve_soa(x0s, us)
    torch.cuda.synchronize()
    st, it = out["status"], out["iters"]
    assert int(it.min()) >= 1 and int(it.max()) <= 50 and int(st.min()) >= 0 and int(st.max()) <= 3
    xr = torch.empty_like(xs)
    Jr = torch.empty(B, dtype=torch.float64, device="cuda")
    pr = s._problem(B, B)
    s._check(s.lib.ilqr_rollout_cost(C.byref(pr), _ptr(x0s), _ptr(us), _ptr(xr), _ptr(Jr), s._stream()))
    torch.cuda.synchronize()
    assert torch.equal(xr.nan_to_num(), xs.nan_to_num())
    del xr
    idx = torch.arange(0, B, 509, device="cuda")[:1024]
    x2, u2, o2 = s.solve_soa(x0s[:, idx].contiguous(), u0s[:, :, idx].contiguous())
    torch.cuda.synchronize()
    assert torch.equal(x2.nan_to_num(), xs[:, :, idx].nan_to_num()) and torch.equal(u2.nan_to_num(), us[:, :, idx].nan_to_num())
    for k in ("iters", "status", "regu_rounds", "ls_total"):
        assert torch.equal(o2[k], out[k][idx]), k
    n = 512
    ref = oracle.solve_batch("car", par, N, 50, 20, 1e-3, x0[:n], u0[:n], rule="ldl")["info"]
    g = np.bincount(st[:n].cpu().numpy(), minlength=4) / float(n)
    h = np.bincount(ref["status"], minlength=4) / float(n)
    print("configuration 5, default rule: exits GPU", g, "checker", h, "over the batch",
          (torch.bincount(st, minlength=4).double() / B).cpu().tolist())
    assert np.abs(g - h).max() < 0.05, (g, h)
    assert abs(float(it[:n].double().mean()) - ref["iters"].mean()) < 0.05 * ref["iters"].mean() + 0.5
    del xs, us, x2, u2
    # the flag against the reference's own 256 solves of this configuration
    q = load_golden("car_n200_population_ref")
    px0, pu0 = car_inputs(int(q["n"]), int(q["seed"]), N)
    r = BatchSolver(0, par, N, 50, 20, 1e-3, flags=_lib.FLAG_LAPACK_EIGVALS).solve(px0, pu0)
    gs, gi = r.status.cpu().numpy(), r.iters.cpu().numpy()
    print("configuration 5, LAPACK's rule: exits GPU", np.bincount(gs, minlength=4), "reference", np.bincount(q["status"], minlength=4),
          "mean iterations %.2f / %.2f" % (gi.mean(), q["iters"].mean()))
    assert np.abs(np.bincount(gs, minlength=4) - np.bincount(q["status"], minlength=4)).max() <= 0.05 * len(gs)
    assert abs(gi.mean() - q["iters"].mean()) < 0.05 * q["iters"].mean()
