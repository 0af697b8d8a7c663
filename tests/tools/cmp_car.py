#!/usr/bin/env python3
"""Population statistics of the headline car batch: HIP path vs the CPU checker (exit reasons, regularisation, iterations)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bench import car_inputs
from ilqr_amd.batch import BatchSolver
from oracle import c_oracle as co
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
x0, u0 = car_inputs(B, 1234, N)
par = co.pack_car(2.0, 2.0)
ref = co.solve_batch("car", par, N, 50, 20, 1e-3, x0, u0)["info"]
r = BatchSolver(0, par, N, 50, 20, 1e-3).solve(x0, u0)
it, st, rg = r.iters.cpu().numpy(), r.status.cpu().numpy(), r.regu_rounds.cpu().numpy()
print("status gpu", np.bincount(st, minlength=4).tolist(), "checker", np.bincount(ref["status"], minlength=4).tolist())
print("mean regu gpu %.1f checker %.1f" % (rg.mean(), ref["regu_rounds"].mean()), " mean iters gpu %.3f checker %.3f" % (it.mean(), ref["iters"].mean()))
print("iters identical %.4f  regu identical %.4f  status identical %.4f" % ((it == ref["iters"]).mean(), (rg == ref["regu_rounds"]).mean(), (st == ref["status"]).mean()))
calm = ref["max_abs_q"] < 1e5
if calm.any(): print("calm fraction %.4f: iters identical among calm %.5f, regu identical among calm %.5f" % (calm.mean(), (it == ref["iters"])[calm].mean(), (rg == ref["regu_rounds"])[calm].mean()))
