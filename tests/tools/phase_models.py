#!/usr/bin/env python3
"""Per-phase device time of one batched solve for a plant at its contract configuration (uses ILQR_FLAG_TIMING)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ilqr_amd import _lib
from ilqr_amd.batch import BatchSolver
from oracle import c_oracle as co
kind, B = sys.argv[1], int(sys.argv[2])
mild = len(sys.argv) > 3 and sys.argv[3] == 'mild'     # controls near hover instead of the contract's 2 N(0,1)
extra = int(sys.argv[4]) if len(sys.argv) > 4 else 0
Nover = int(sys.argv[5]) if len(sys.argv) > 5 else 0
Q = np.diag([10.0] * 6 + [1.0] * 6); Qf = np.diag([500.0] * 3 + [100.0] * 3 + [20.0] * 6); R = np.diag([0.02] * 4)
rng = np.random.default_rng(2)
if kind == "acrobot":
    mid, N, mi, ml, dt = 1, 100, 20, 10, 1e-3
    par = co.pack_quadratic(0.05, np.diag([1, 1, .1, .1]), 100.0 * np.eye(4), [[0.01]], [np.pi, 0, 0, 0], [0.0])
    x0 = np.column_stack([rng.uniform(-np.pi, np.pi, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B), rng.uniform(-1, 1, B)])
    u0 = rng.standard_normal((B, N - 1, 1))
elif kind == "quad_ea":
    mid, N, mi, ml, dt = 2, (Nover or 100), 20, 10, 5e-4
    xref = np.zeros(12); xref[2] = 1.0
    par = co.pack_quadratic(0.01, Q, Qf, R, xref, np.zeros(4))
    x0 = np.zeros((B, 12)); x0[:, 2] = 0.8; x0[:, 3:6] = 0.5 * rng.standard_normal((B, 3))
    u0 = (1.9 + 0.5 * rng.standard_normal((B, N - 1, 4))) if mild else 2.0 * rng.standard_normal((B, N - 1, 4))
else:
    mid, N, mi, ml, dt = 3, 80, 20, 10, 5e-4
    xref = np.zeros(13); xref[0] = 1.0; xref[6] = 1.0
    par = co.pack_quat(0.01, Q, Qf, R, xref, np.zeros(4))
    x0 = np.zeros((B, 13)); x0[:, 6] = 0.8
    q = np.column_stack([np.ones(B), 0.25 * rng.standard_normal((B, 3))]); x0[:, :4] = q / np.linalg.norm(q, axis=1, keepdims=True)
    u0 = 2.0 * rng.standard_normal((B, N - 1, 4))
s = BatchSolver(mid, par, N, mi, ml, dt, flags=_lib.FLAG_TIMING | _lib.FLAG_SINGLE_STREAM | extra)
r = s.solve(x0, u0)
torch.cuda.synchronize()
t = s.last_timing()
print(json.dumps(dict(model=kind, B=B, mild=mild, status=torch.bincount(r.status, minlength=4).cpu().tolist(), phases=t, mean_iters=float(r.iters.double().mean()), mean_ls=float(r.ls_total.double().mean()),
                      mean_regu=float(r.regu_rounds.double().mean()))))
