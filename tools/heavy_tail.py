import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import bench
from ilqr_amd import _lib
from ilqr_amd.batch import BatchSolver
B = 1 << 20
s = BatchSolver(0, [2.0, 2.0], 40, 50, 20, 1e-3, flags=_lib.FLAG_LAPACK_EIGVALS)
x0, u0 = bench.car_inputs(B, 1234)
r = s.solve(x0, u0)
it = r.iters.cpu().numpy(); rr = r.regu_rounds.cpu().numpy().astype(np.int64); st = r.status.cpu().numpy()
pp = rr / np.maximum(it, 1)
print("rounds per pass quantiles 50/90/99/99.9/99.99/99.999/max:", np.percentile(pp, [50, 90, 99, 99.9, 99.99, 99.999, 100]))
for thr in (512, 2048, 4096, 8192, 16384):
    m = pp > thr
    print("rounds/pass >", thr, ":", int(m.sum()), "solves; exits", np.bincount(st[m], minlength=4), "mean iters %.1f" % (it[m].mean() if m.any() else 0))
top = np.argsort(pp)[-8:]
print("top:", [(int(pp[i]), int(it[i]), int(st[i])) for i in top])
