"""How regularisation is spread over the trajectories of the headline batch (default rule): stages regularised per
backward pass and per trajectory, from solves cut off after 1, 2, ... iterations (solves are deterministic, so the
difference between two cut-offs is what one pass added).  Answers: if wavefronts held only trajectories that
regularised in their previous pass, how many wavefront-stages would still run the group scan?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from ilqr_amd.batch import BatchSolver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
x0, u0 = bench.car_inputs(B, 1234)
prev_s = np.zeros(B, np.int64); prev_it = np.zeros(B, np.int64)
per = np.zeros((T, B), np.int64); ran = np.zeros((T, B), bool)
for t in range(1, T + 1):
    r = BatchSolver(0, [2.0, 2.0], 40, t, 20, 1e-3).solve(x0, u0)
    s = r.regu_stages.cpu().numpy().astype(np.int64); it = r.iters.cpu().numpy().astype(np.int64)
    ran[t - 1] = it > prev_it
    per[t - 1] = s - prev_s
    prev_s, prev_it = s, it
S = 39
for t in range(T):
    m = ran[t]
    n = int(m.sum())
    if n == 0: break
    st = per[t][m]
    frac_any = (st > 0).mean()
    line = "pass %2d: running %7d, stages regularised per pass mean %.2f, trajectories with none %.3f" % (t + 1, n, st.mean(), 1 - frac_any)
    if t > 0:
        both = m & ran[t - 1]
        a = per[t - 1][both] > 0; b = per[t][both] > 0
        line += "; P(any | any before) %.3f, P(any | none before) %.3f" % (b[a].mean() if a.any() else 0, b[~a].mean() if (~a).any() else 0)
        # wavefront-stages that scan: random packing vs packing by the previous pass (needy first)
        k = per[t][both]
        def scan_frac(order):
            kk = k[order]; w = len(kk) // 64
            kk = kk[: w * 64].reshape(w, 64)
            # a wavefront scans at a stage if any of its lanes regularises there; stages independent and uniform: 1 - prod(1 - k/S)
            return (1 - np.prod(1 - kk / S, axis=1)).mean()
        rnd = np.random.default_rng(0).permutation(len(k))
        srt = np.argsort(-per[t - 1][both], kind="stable")
        line += "; wavefront-stages scanning: as is %.3f, sorted by previous pass %.3f" % (scan_frac(rnd), scan_frac(srt))
    print(line)
