#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- the reference solver itself on the first trajectories of the headline batch.

Runs the imported reference `iLQR` (control/iLQRController.py, driven as in make_golden.py) on `n`
trajectories of bench.py's CarModel N=40 workload (`car_inputs(n, 1234)`) and stores what each solve ended with: exit
reason, iteration count, regularisation rounds, line-search evaluations, final cost.  A quarter of these
solves are ill conditioned and not reproducible one by one (not even by the reference under a 1e-14
perturbation); as a population they pin the statistics the GPU path and the CPU checker must show.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_population.py [n=8192] [workers=8]
writes tests/golden/car_population_ref.npz (needs /root/reference; takes minutes).
"""
import os
import sys
import time
from concurrent.futures import ProcessPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
import make_golden as mg  # noqa: E402

SPEC = dict(model="car", r=2.0, v_target=2.0)
SP = dict(N=40, max_iter=50, max_ls=20, d_tol=1e-3)


def one(args):
    x0, u0 = args
    r = mg._one((SPEC, SP, x0, u0, False))
    return (r["status"], r["iters"], r["regu_rounds"], int(r["ls_evals"].sum()), r["J_last"], r["max_abs_q"])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    from bench import car_inputs
    x0, u0 = car_inputs(n, 1234, SP["N"])
    t0 = time.time()
    out = []
    with ProcessPoolExecutor(workers) as pool:
        for i, r in enumerate(pool.map(one, [(x0[b], u0[b]) for b in range(n)], chunksize=16)):
            out.append(r)
            if (i + 1) % 512 == 0:
                sys.stderr.write(f"{i + 1}/{n} after {time.time() - t0:.0f} s\n")
    a = np.array(out, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "car_population_ref.npz"), n=n, seed=1234,
                        status=a[:, 0].astype(np.int8), iters=a[:, 1].astype(np.int16),
                        regu_rounds=a[:, 2].astype(np.int32), ls_total=a[:, 3].astype(np.int32),
                        J=a[:, 4], max_abs_q=a[:, 5])
    print("status", np.bincount(a[:, 0].astype(int), minlength=4).tolist(), "mean iters %.3f" % a[:, 1].mean(),
          "mean regu %.1f" % a[:, 2].mean(), "in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
