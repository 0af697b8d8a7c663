#!/usr/bin/env python3
"""Closed-loop throughput of the batched receding-horizon driver (ilqr_amd/mpc.py): B loops advanced in lock step,
state and controls resident on the GPU between solves.  Prints one JSON line per case."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import car_inputs
from ilqr_amd.batch import BatchSolver
from ilqr_amd.mpc import BatchMPC
from oracle import c_oracle as co


def run(name, solver, x0, u0, steps, shift):
    mpc = BatchMPC(solver, shift=shift)
    mpc.run(x0[:4096], u0[:4096], steps=2)            # warm up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    log = mpc.run(x0, u0, steps=steps)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    B = x0.shape[0]
    print(json.dumps(dict(case=name, B=B, steps=steps, seconds=dt, mpc_steps_per_s=B * steps / dt,
                          mean_iters_first=float(log["iters"][0].double().mean()),
                          mean_iters_last=float(log["iters"][-1].double().mean()))), flush=True)


B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
x0, u0 = car_inputs(B, 7)
run("car N=40, re-solve from x[1] (Driving.ipynb)", BatchSolver(0, co.pack_car(2.0, 2.0), 40, 50, 20, 1e-3), x0, u0, 20, False)
Q = np.diag([10.0] * 6 + [1.0] * 6); Qf = np.diag([500.0] * 3 + [100.0] * 3 + [20.0] * 6); R = np.diag([0.02] * 4)
xref = np.zeros(13); xref[0] = 1.0; xref[6] = 1.0
rng = np.random.default_rng(4); Bq = min(B, 16384); N = 80
xq = np.zeros((Bq, 13)); xq[:, 6] = 0.8
q = np.column_stack([np.ones(Bq), 0.1 * rng.standard_normal((Bq, 3))]); xq[:, :4] = q / np.linalg.norm(q, axis=1, keepdims=True)
uq = 1.9 + 0.2 * rng.standard_normal((Bq, N - 1, 4))
run("quaternion quadrotor N=80 near hover, plant step + shift (QuadrotorControllers.py)",
    BatchSolver(3, co.pack_quat(0.01, Q, Qf, R, xref, np.zeros(4)), N, 20, 10, 5e-4), xq, uq, 10, True)
