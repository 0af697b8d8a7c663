#!/usr/bin/env python3
"""How the time of one car solve (B trajectories) grows with max_iter: separates the throughput-bound first
iterations from the latency-bound tail of the few trajectories that never converge."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from bench import car_inputs
from ilqr_amd import _lib
from ilqr_amd.batch import BatchSolver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
x0, u0 = car_inputs(B, 0)
x0 = torch.as_tensor(x0.T.copy(), device="cuda")
u0 = torch.as_tensor(np.ascontiguousarray(u0.transpose(1, 2, 0)), device="cuda")
for mi in (1, 2, 3, 4, 6, 8, 12, 20, 30, 50):
    s = BatchSolver(_lib.MODEL_CAR, [2.0, 2.0], 40, mi, 20, 1e-3)
    ts = []
    for rep in range(3):
        u = u0.clone(); torch.cuda.synchronize(); t = time.perf_counter()
        out = s.solve_soa(x0, u); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    print(f"max_iter {mi:2d}: {min(ts) * 1e3:7.2f} ms", flush=True)
