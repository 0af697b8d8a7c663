import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ilqr_amd import _lib
from ilqr_amd.batch import BatchSolver
from oracle import c_oracle as co
Q = np.diag([10.0] * 6 + [1.0] * 6); Qf = np.diag([500.0] * 3 + [100.0] * 3 + [20.0] * 6); R = np.diag([0.02] * 4)
rng = np.random.default_rng(2); B = 256
for kind in ("quad_quat", "quad_ea", "acrobot"):
    if kind == "quad_quat":
        mid, N, mi, ml, dt = 3, 80, 20, 10, 5e-4
        xref = np.zeros(13); xref[0] = 1.0; xref[6] = 1.0
        par = co.pack_quat(0.01, Q, Qf, R, xref, np.zeros(4))
        x0 = np.zeros((B, 13)); x0[:, 6] = 0.8
        q = np.column_stack([np.ones(B), 0.25 * rng.standard_normal((B, 3))]); x0[:, :4] = q / np.linalg.norm(q, axis=1, keepdims=True)
        u0 = 2.0 * rng.standard_normal((B, N - 1, 4))
    elif kind == "quad_ea":
        mid, N, mi, ml, dt = 2, 100, 20, 10, 5e-4
        xref = np.zeros(12); xref[2] = 1.0
        par = co.pack_quadratic(0.01, Q, Qf, R, xref, np.zeros(4))
        x0 = np.zeros((B, 12)); x0[:, 2] = 0.8; x0[:, 3:6] = 0.5 * rng.standard_normal((B, 3))
        u0 = 2.0 * rng.standard_normal((B, N - 1, 4))
    else:
        mid, N, mi, ml, dt = 1, 100, 20, 10, 1e-3
        par = co.pack_quadratic(0.05, np.diag([1, 1, .1, .1]), 100.0 * np.eye(4), [[0.01]], [np.pi, 0, 0, 0], [0.0])
        x0 = np.column_stack([rng.uniform(-np.pi, np.pi, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-1, 1, B)])
        u0 = rng.standard_normal((B, N - 1, 1))
    ref = co.solve_batch(kind, par, N, mi, ml, dt, x0, u0)["info"]
    for flags in (0,):
        r = BatchSolver(mid, par, N, mi, ml, dt, flags=flags).solve(x0, u0)
        print(kind, "flags", flags, "gpu mean regu %.1f" % r.regu_rounds.double().mean().item(), "oracle %.1f" % ref["regu_rounds"].mean(),
              "status gpu", torch.bincount(r.status, minlength=4).cpu().tolist(), "oracle", np.bincount(ref["status"], minlength=4).tolist(),
              "iters identical %.3f" % (r.iters.cpu().numpy() == ref["iters"]).mean(),
              "regu identical %.3f" % (r.regu_rounds.cpu().numpy() == ref["regu_rounds"]).mean())
