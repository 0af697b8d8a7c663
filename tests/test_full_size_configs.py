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
