#!/usr/bin/env python3
"""Car N = 200 fixtures: regularisation rounds, iterations and exits per solve of the reference (fixture), the checker
with LAPACK's rule and the kernels with ILQR_FLAG_LAPACK_EIGVALS, side by side."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import load_golden
from oracle import c_oracle as co
from ilqr_amd.batch import BatchSolver
np.set_printoptions(linewidth=200)
for name in ("car_N200", "car_circle_N200"):
    q = load_golden(name)
    ck = co.solve_batch("car", q["par"], 200, 50, 20, 1e-3, q["x0"], q["u_init"])["info"]
    r = BatchSolver(0, q["par"], 200, 50, 20, 1e-3, flags=32).solve(q["x0"], q["u_init"])
    g = dict(regu=r.regu_rounds.cpu().numpy(), iters=r.iters.cpu().numpy(), status=r.status.cpu().numpy(), stages=r.regu_stages.cpu().numpy(),
             maxq=r.max_abs_q.cpu().numpy())
    print(name)
    print(" rounds  ref", q["regu_rounds"]); print("     checker", ck["regu_rounds"]); print("         gpu", g["regu"])
    print(" iters   ref", q["iters"]); print("     checker", ck["iters"]); print("         gpu", g["iters"])
    print(" status  ref", q["status"]); print("     checker", ck["status"]); print("         gpu", g["status"])
    print(" stages checker", ck["regu_stages"]); print("         gpu", g["stages"])
    print(" maxq checker", ck["max_abs_q"]); print("         gpu", g["maxq"])
