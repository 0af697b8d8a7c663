#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- the reference solver itself on the first trajectories of the headline batch.

Runs the imported reference `iLQR` (control/iLQRController.py, driven as in make_golden.py) on `n`
trajectories of bench.py's CarModel N=40 workload (`car_inputs(n, 1234)`) and stores what each solve ended with: exit
reason, iteration count, regularisation rounds, line-search evaluations, final cost.  A quarter of these
solves are ill conditioned and not reproducible one by one (not even by the reference under a 1e-14
perturbation); as a population they pin the statistics the GPU path and the CPU checker must show.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_population.py [n=8192] [workers=8] [plant=car]
writes tests/golden/car_population_ref.npz (needs /root/reference; takes minutes).

With plant=acrobot: the contract's second configuration (Acrobot2DModel N=100, max_iter 20, max_linesearch_iters 10,
x0 ~ U(-pi,pi)^2 x U(-1,1)^2, u_init ~ N(0,1), `acrobot_inputs(n, 2)`) -> tests/golden/acrobot_population_ref.npz.
With plant=quad_ea / quad_quat: configurations 3 and 4 (far from hover, u_init = 2 N(0,1); N = 100 / 80,
`quad_inputs`) -> tests/golden/quad_ea_population_ref.npz / quad_quat_population_ref.npz.
With plant=car_n200: configuration 5, the stress horizon (CarModel N = 200 from the headline workload's box of starts,
`car_inputs(n, 5, 200)`; every such solve is ill conditioned) -> tests/golden/car_n200_population_ref.npz.
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
SP200 = dict(N=200, max_iter=50, max_ls=20, d_tol=1e-3)


def acrobot_inputs(n, seed, N=100):
    rng = np.random.default_rng(seed)
    x0 = np.column_stack([rng.uniform(-np.pi, np.pi, n), rng.uniform(-np.pi, np.pi, n), rng.uniform(-1, 1, n),
                          rng.uniform(-1, 1, n)])
    return x0, rng.standard_normal((n, N - 1, 1))


ACROBOT_SP = dict(N=100, max_iter=20, max_ls=10, d_tol=1e-3)


def quad_inputs(kind, n, seed, N):
    """The contract's configurations 3 and 4: hovering height 0.8, attitude 0.5 N(0,1) (Euler angles; for the
    quaternion plant the unit quaternion of ZYX Euler angles 0.5 N(0,1)), u_init = 2 N(0,1)."""
    rng = np.random.default_rng(seed)
    if kind == "quad_ea":
        x0 = np.zeros((n, 12)); x0[:, 2] = 0.8; x0[:, 3:6] = 0.5 * rng.standard_normal((n, 3))
    else:
        x0 = np.zeros((n, 13)); x0[:, 6] = 0.8
        for b in range(n):
            x0[b, :4] = mg.euler_zyx_to_quat(0.5 * rng.standard_normal(3))
    return x0, 2.0 * rng.standard_normal((n, N - 1, 4))


QUAD_SP = dict(quad_ea=dict(N=100, max_iter=20, max_ls=10, d_tol=5e-4), quad_quat=dict(N=80, max_iter=20, max_ls=10, d_tol=5e-4))


def one(args):
    x0, u0 = args[:2]
    spec, sp = (args[2], args[3]) if len(args) > 2 else (SPEC, SP)
    r = mg._one((spec, sp, x0, u0, False))
    return (r["status"], r["iters"], r["regu_rounds"], int(r["ls_evals"].sum()), r["J_last"], r["max_abs_q"])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    plant = sys.argv[3] if len(sys.argv) > 3 else "car"
    if plant in ("quad_ea", "quad_quat"):
        sp = QUAD_SP[plant]
        seed = 3 if plant == "quad_ea" else 4
        x0, u0 = quad_inputs(plant, n, seed, sp["N"])
        spec = mg.ea_spec(sp["N"]) if plant == "quad_ea" else mg.quat_spec(sp["N"])
        jobs = [(x0[b], u0[b], spec, sp) for b in range(n)]
        name = plant + "_population_ref.npz"
    elif plant == "acrobot":
        x0, u0 = acrobot_inputs(n, 2)
        jobs = [(x0[b], u0[b], mg.acrobot_spec(100), ACROBOT_SP) for b in range(n)]
        name, seed = "acrobot_population_ref.npz", 2
    elif plant == "car_n200":
        from bench import car_inputs
        x0, u0 = car_inputs(n, 5, SP200["N"])
        jobs = [(x0[b], u0[b], SPEC, SP200) for b in range(n)]
        name, seed = "car_n200_population_ref.npz", 5
    else:
        from bench import car_inputs
        x0, u0 = car_inputs(n, 1234, SP["N"])
        jobs = [(x0[b], u0[b]) for b in range(n)]
        name, seed = "car_population_ref.npz", 1234
    t0 = time.time()
    out = []
    with ProcessPoolExecutor(workers) as pool:
        for i, r in enumerate(pool.map(one, jobs, chunksize=4)):
            out.append(r)
            if (i + 1) % 32 == 0:
                sys.stderr.write(f"{i + 1}/{n} after {time.time() - t0:.0f} s\n")
    a = np.array(out, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, name), n=n, seed=seed,
                        status=a[:, 0].astype(np.int8), iters=a[:, 1].astype(np.int16),
                        regu_rounds=a[:, 2].astype(np.int32), ls_total=a[:, 3].astype(np.int32),
                        J=a[:, 4], max_abs_q=a[:, 5])
    print("status", np.bincount(a[:, 0].astype(int), minlength=4).tolist(), "mean iters %.3f" % a[:, 1].mean(),
          "mean regu %.1f" % a[:, 2].mean(), "in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
