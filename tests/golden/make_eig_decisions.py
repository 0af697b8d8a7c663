#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- per-decision fixture for the regularisation loop of the reference.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_eig_decisions.py [workers=8]

Runs the imported reference `iLQR` (control/iLQRController.py, driven as in make_golden.py) and records, stage by
stage of its backward passes, what the loop at iLQRController.py:106-115 did:

    regu = 0.1; while np.any(np.linalg.eigvals(Q) < 0): Q += (regu F)'F; regu *= 2

For a recorded stage the fixture holds Q as it entered the loop (`Q0`), the first update P = (0.1 F)'F assembled
block by block with the reference's own expressions (`P`), the number of rounds the loop ran (`rounds`: LAPACK said
"negative" for Q(0) .. Q(rounds-1) and "none" for Q(rounds)), and `end` (0: the loop ended, 1: eigvals raised
LinAlgError on a non-finite Q).  The matrices the loop tested in between are Q(r+1) = Q(r) + 2^r P, summed one round
after the other (2^r P is exact); the generator checks that this reproduces, bit for bit, every matrix
the reference handed to eigvals, so a fixture stage stands for rounds+1 (matrix, LAPACK verdict) pairs.

Recorded: every stage that took more than 8 rounds, a tenth of those with 1-8 rounds, one in 200 of those with none,
and a tenth of the stages whose smallest |eigenvalue| is below 1e-12 of the largest |entry| (the structural zeros), from
the first solves of the headline car workload (bench.car_inputs(n, 1234)) and from the quadrotor fixtures' inputs.
Writes tests/golden/eig_decisions.npz.
"""
import os
import sys
import warnings
from concurrent.futures import ProcessPoolExecutor

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
import make_golden as mg  # noqa: E402


def record_solve(args):
    spec, sp, x0, u0, seed, max_stuck = args
    from control.iLQRController import iLQR
    plant = mg.make_plant(spec)
    ctl = iLQR(plant, sp["N"], sp["max_iter"], sp["max_ls"], sp["d_tol"])
    rng = np.random.default_rng(seed)
    out = []            # (Q0, P, rounds, end)
    stats = dict(stages=0, decisions=0, rebuilt_ok=0, rebuilt_bad=0)
    cur = dict(F=None, Q0=None, P=None, seq=[], minrel=np.inf)
    orig_eig = np.linalg.eigvals
    stage_fn = plant.stage

    def close(end):
        if cur["Q0"] is None:
            return
        rounds = len(cur["seq"]) - 1 if end == 0 else len(cur["seq"])
        stats["stages"] += 1
        stats["decisions"] += len(cur["seq"])
        # the running sum reproduces what the reference tested
        Q = cur["Q0"].copy()
        ok = True
        for r, Qr in enumerate(cur["seq"]):
            ok = ok and np.array_equal(Q, Qr, equal_nan=True)
            Q = Q + np.ldexp(cur["P"], r)
        stats["rebuilt_ok" if ok else "rebuilt_bad"] += 1
        keep = (rounds > 8 and sum(1 for o in out if o[2] > 8) < max_stuck) or (1 <= rounds <= 8 and rng.random() < 0.1) \
            or (rounds == 0 and (rng.random() < 0.005 or (cur["minrel"] < 1e-12 and rng.random() < 0.1)))
        if keep and ok:
            out.append((cur["Q0"], cur["P"], rounds, end))
        cur["Q0"] = None
        cur["seq"] = []
        cur["minrel"] = np.inf

    def stage(x, u):
        r = stage_fn(x, u)
        fx, fu = np.asarray(r[5]), np.asarray(r[6])
        regu = 0.1
        # the three updates of iLQRController.py:111-113 at regu = 0.1, assembled like _full_hessian (:139-147)
        Pxx = regu * fx.T @ fx
        Puu = regu * fu.T @ fu
        Pux = regu * fu.T @ fx
        cur["P"] = np.block([[Pxx, Pux.T], [Pux, Puu]])
        return r

    def eig(Q):
        if cur["Q0"] is None:
            cur["Q0"] = np.array(Q, copy=True)
        cur["seq"].append(np.array(Q, copy=True))
        try:
            w = orig_eig(Q)
        except np.linalg.LinAlgError:
            close(1)
            raise
        a = float(np.abs(Q).max())
        if a > 0 and len(cur["seq"]) == 1:
            cur["minrel"] = float(np.abs(w).min()) / a
        if not np.any(w < 0):
            close(0)
        return w

    plant.stage = stage
    np.linalg.eigvals = eig
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            import contextlib
            import io
            with contextlib.redirect_stdout(io.StringIO()):
                ctl.run_ilqr(np.array(x0, dtype=np.float64), np.array(u0, dtype=np.float64, copy=True))
    except np.linalg.LinAlgError:
        pass
    finally:
        np.linalg.eigvals = orig_eig
    return out, stats


def collect(pool, spec, sp, X0, U0, max_stuck):
    jobs = [(spec, sp, X0[b], U0[b], 1000 + b, max_stuck) for b in range(len(X0))]
    recs, tot = [], dict(stages=0, decisions=0, rebuilt_ok=0, rebuilt_bad=0)
    for out, st in pool.map(record_solve, jobs, chunksize=4):
        recs += out
        for k in tot:
            tot[k] += st[k]
    return recs, tot


def pack(recs):
    w = recs[0][0].shape[0]
    return dict(Q0=np.stack([r[0] for r in recs]).reshape(-1, w, w), P=np.stack([r[1] for r in recs]).reshape(-1, w, w),
                rounds=np.array([r[2] for r in recs], np.int32), end=np.array([r[3] for r in recs], np.int8))


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    from bench import car_inputs
    out = {}
    with ProcessPoolExecutor(workers) as pool:
        x0, u0 = car_inputs(640, 1234, 40)
        recs, tot = collect(pool, mg.CAR, mg.CAR_SP, x0, u0, 100000)
        print("car", tot, "recorded", len(recs), "stages,", sum(r[2] + 1 for r in recs), "decisions", flush=True)
        assert tot["rebuilt_bad"] == 0, tot
        for k, v in pack(recs).items():
            out["car_" + k] = v
        for tag, name, spec in (("ea", "quad_ea_N100", mg.ea_spec(100)), ("quat", "quad_quat_N80", mg.quat_spec(80))):
            g = np.load(os.path.join(HERE, name + ".npz"))
            sp = dict(N=int(g["N"]), max_iter=int(g["max_iter"]), max_ls=int(g["max_ls"]), d_tol=float(g["d_tol"]))
            recs, tot = collect(pool, spec, sp, g["x0"][:4], g["u_init"][:4], 48)
            print(tag, tot, "recorded", len(recs), "stages,", sum(r[2] + 1 for r in recs), "decisions", flush=True)
            assert tot["rebuilt_bad"] == 0, tot
            for k, v in pack(recs).items():
                out[tag + "_" + k] = v
    np.savez_compressed(os.path.join(HERE, "eig_decisions.npz"), **out)
    print("wrote eig_decisions.npz,", os.path.getsize(os.path.join(HERE, "eig_decisions.npz")) / 1e6, "MB")


if __name__ == "__main__":
    main()
