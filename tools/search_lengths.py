"""Line searches of the headline batch: how many evaluations each takes (control/iLQRController.py:60-79) and how well
the previous search of the same trajectory predicts it.  python tools/search_lengths.py [batch=65536]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from ilqr_amd.batch import BatchSolver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 16
x0, u0 = bench.car_inputs(B, 1234)
r = BatchSolver(0, [2.0, 2.0], 40, 50, 20, 1e-3, trace_len=50).solve(x0, u0)
ls = r.traces["ls_evals"].t().cpu().numpy().astype(np.int64)          # [T][B]
it = r.iters.cpu().numpy().astype(np.int64)
T = ls.shape[0]
ran = (np.arange(T)[:, None] < it[None, :]) & (ls > 0)
allv = ls[ran]
print("searches", allv.size, "evaluations", int(allv.sum()), "mean %.2f" % allv.mean())
h = np.bincount(np.minimum(allv, 21), minlength=22)
print("evaluations per search 1..20:", (h[1:21] / allv.size).round(3).tolist())
for lo, hi in ((0, 11), (11, 50)):
    m = ran[lo:hi]; v = ls[lo:hi][m]
    print("iterations %d-%d: searches %d, mean %.2f, <=2 %.3f, <=4 %.3f, <=8 %.3f" % (lo + 1, hi, v.size, v.mean(), (v <= 2).mean(), (v <= 4).mean(), (v <= 8).mean()))
prev = ls[:-1]; cur = ls[1:]; both = ran[:-1] & ran[1:]
p, c = prev[both], cur[both]
for k in (1, 2, 4):
    a = p <= k
    print("previous <= %d: share %.3f; then now <= %d with prob %.3f (else %.3f)" % (k, a.mean(), k, (c[a] <= k).mean(), (c[~a] <= k).mean()))
# lanes spent by schedules (per search, first pass only + repeats of the same width until stop), side-by-side from evaluation 1
def lanes(width, v):
    return (np.ceil(v / width) * width)
v = c
print("lane-evaluations per reference evaluation: width 8: %.2f, width 4: %.2f, sized by previous (4 if previous <= 4 else 8): %.2f, (2 if <= 2, 4 if <= 4, else 8): %.2f" % (
    lanes(8, v).sum() / v.sum(), lanes(4, v).sum() / v.sum(),
    np.where(p <= 4, lanes(4, v), lanes(8, v)).sum() / v.sum(),
    np.where(p <= 2, lanes(2, v), np.where(p <= 4, lanes(4, v), lanes(8, v))).sum() / v.sum()))
