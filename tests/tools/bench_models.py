#!/usr/bin/env python3
"""Throughput of the other plants at the configurations of the coverage contract (not the headline bench):
acrobot N=100 B=65536, Euler quadrotor N=100, quaternion quadrotor N=80 (starts: unit quaternions from ZYX Euler
angles 0.5 N(0,1), as the contract states them; --near-hover for the milder starts earlier lines were measured
with), car N=200 (`car200`, starts from the headline box).  One JSON line per plant, appended to --out if given."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from ilqr_amd.batch import BatchSolver
from oracle import c_oracle as co


def cfg(kind):
    Q = np.diag([10.0] * 6 + [1.0] * 6); Qf = np.diag([500.0] * 3 + [100.0] * 3 + [20.0] * 6); R = np.diag([0.02] * 4)
    if kind == "acrobot":
        par = co.pack_quadratic(0.05, np.diag([1, 1, .1, .1]), 100.0 * np.eye(4), [[0.01]], [np.pi, 0, 0, 0], [0.0])
        return 1, par, 100, 20, 10, 1e-3
    if kind == "car200":
        return 0, co.pack_car(2.0, 2.0), 200, 50, 20, 1e-3
    if kind == "quad_ea":
        xref = np.zeros(12); xref[2] = 1.0
        return 2, co.pack_quadratic(0.01, Q, Qf, R, xref, np.zeros(4)), 100, 20, 10, 5e-4
    xref = np.zeros(13); xref[0] = 1.0; xref[6] = 1.0
    return 3, co.pack_quat(0.01, Q, Qf, R, xref, np.zeros(4)), 80, 20, 10, 5e-4


def inputs(kind, rng, B, N):
    if kind == "acrobot":
        x0 = np.column_stack([rng.uniform(-np.pi, np.pi, B), rng.uniform(-np.pi, np.pi, B), rng.uniform(-1, 1, B),
                              rng.uniform(-1, 1, B)])
        return x0, rng.standard_normal((B, N - 1, 1))
    if kind == "car200":
        import bench
        return bench.car_inputs(B, 5, N)
    if kind == "quad_ea":
        x0 = np.zeros((B, 12)); x0[:, 2] = 0.8; x0[:, 3:6] = 0.5 * rng.standard_normal((B, 3))
        return x0, 2.0 * rng.standard_normal((B, N - 1, 4))
    x0 = np.zeros((B, 13)); x0[:, 6] = 0.8
    if NEAR_HOVER:
        q = np.column_stack([np.ones(B), 0.25 * rng.standard_normal((B, 3))])
        x0[:, :4] = q / np.linalg.norm(q, axis=1, keepdims=True)
    else:
        from scipy.spatial.transform import Rotation
        q = Rotation.from_euler("ZYX", 0.5 * rng.standard_normal((B, 3))).as_quat()      # [x, y, z, w]
        x0[:, :4] = np.column_stack([q[:, 3], q[:, 0], q[:, 1], q[:, 2]])
    return x0, 2.0 * rng.standard_normal((B, N - 1, 4))


ap = argparse.ArgumentParser()
ap.add_argument("--kinds", default="acrobot,quad_ea,quad_quat")
ap.add_argument("--batch", type=int, default=65536)
ap.add_argument("--cpu", type=int, default=512)
ap.add_argument("--lapack-eigvals", action="store_true", help="ILQR_FLAG_LAPACK_EIGVALS")
ap.add_argument("--near-hover", action="store_true")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--out", default="")
a = ap.parse_args()
NEAR_HOVER = a.near_hover
for kind in a.kinds.split(","):
    mid, par, N, mi, ml, dt = cfg(kind)
    B = a.batch
    x0, u0 = inputs(kind, np.random.default_rng(2), B, N)
    s = BatchSolver(mid, par, N, mi, ml, dt, flags=32 if a.lapack_eigvals else 0)
    x0d = torch.as_tensor(x0).cuda().t().contiguous(); u_init = torch.as_tensor(u0).cuda().permute(1, 2, 0).contiguous()
    u = u_init.clone(); x = torch.empty((N, x0.shape[1], B), dtype=torch.float64, device="cuda"); out = s.alloc_outputs(B)
    for rep in range(a.reps):
        u.copy_(u_init); torch.cuda.synchronize(); t0 = time.perf_counter()
        s.solve_soa(x0d, u, x, out); torch.cuda.synchronize(); dtm = time.perf_counter() - t0
    st = torch.bincount(out["status"], minlength=4).cpu().tolist()
    c = min(a.cpu, B)
    rule = "lapack" if a.lapack_eigvals else "ldl"
    if c > 0:
        t1 = time.perf_counter(); r = co.solve_batch("car" if kind == "car200" else kind, par, N, mi, ml, dt, x0[:c], u0[:c], rule=rule); tc = time.perf_counter() - t1
        same = float((out["iters"][:c].cpu().numpy() == r["info"]["iters"]).mean())
        cst = np.bincount(r["info"]["status"], minlength=4).tolist()
    else:
        tc, same, cst = float("nan"), float("nan"), None
    line = json.dumps(dict(model=kind, N=N, B=B, rule=rule, ms=1e3 * dtm, solves_per_s=B / dtm,
                           mean_iters=float(out["iters"].double().mean()), mean_regu_rounds=float(out["regu_rounds"].double().mean()),
                           replayed_iterations=int(out["replayed"].sum()), exit=st, cpu_sample=c, cpu_rule=rule,
                           cpu_solves_per_s=(c / tc if c else None), cpu_exit=cst, iters_identical_frac=same))
    print(line, flush=True)
    if a.out:
        with open(a.out, "a") as f:
            f.write(line + "\n")
