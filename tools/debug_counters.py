#!/usr/bin/env python3
"""How often the wavefronts of a faithful (ILQR_FLAG_LAPACK_EIGVALS) car solve take LAPACK's path, per site, from a
diagnostics build of the library:
    python -c "from ilqr_amd import build as r; r.build_library('ilqr_amd/lib/libilqr_hip_dbg.so', defines=['ILQR_DEBUG_COUNTERS'])"
    ILQR_HIP_LIB=ilqr_amd/lib/libilqr_hip_dbg.so python tools/debug_counters.py [batch]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ilqr_amd import _lib  # noqa: E402
from ilqr_amd.batch import BatchSolver  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
max_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda:0")
s = BatchSolver(0, [2.0, 2.0], 40, max_iter, 20, 1e-3, device=dev, flags=_lib.FLAG_LAPACK_EIGVALS | _lib.FLAG_SINGLE_STREAM)
x0, u0 = bench.car_inputs(B, 1234)
L = s.lib
out = (C.c_ulonglong * 16)()
hist = (C.c_ulonglong * 32)()
eigc = (C.c_ulonglong * 8)()
eigt = (C.c_ulonglong * 12)()
L.ilqr_debug_eig_cycles(eigc, 1)
L.ilqr_debug_eig_trips(eigt, 1)
L.ilqr_debug_histograms(hist, 1)
L.ilqr_debug_counters(out, 1)
r = s.solve(x0, u0)
torch.cuda.synchronize()
L.ilqr_debug_counters(out, 1)
v = list(out)
for k, name in enumerate(("round 0", "group scan", "long loops")):
    print(f"{name:11s}: wavefront calls {v[2 * k]:10d}, with LAPACK {v[2 * k + 1]:10d} ({v[2 * k + 1] / max(v[2 * k], 1):.3f}), "
          f"lanes in doubt per call {v[8 + k] / max(v[2 * k], 1):.2f}")
L.ilqr_debug_histograms(hist, 1)
h = list(hist)
print("launch for long loops, tests per trajectory-pass,        bins 0, 1, 2, <=4, <=8, ...:", h[:14])
print("launch for long loops, stages by the round the loop stopped at, bins 0, 1, 2, <=4, <=8, ...:", h[16:30])
L.ilqr_debug_eig_cycles(eigc, 1)
e = list(eigc)
calls = max(e[4], 1)
print(f"LAPACK tests of a wavefront: {e[4]}, cycles each: load/scale {e[0] / calls:.0f}, DGEBAL {e[1] / calls:.0f} "
      f"(run in full in {e[5] / calls:.3f} of the tests), DGEHD2 {e[2] / calls:.0f}, DLAHQR {e[3] / calls:.0f}")
L.ilqr_debug_eig_trips(eigt, 1)
t = list(eigt)
nm = max(t[0], 1)
print(f"per matrix on LAPACK's path ({t[0]} matrices): QR iterations {t[1] / nm:.1f}, deflation-search trips {t[2] / nm:.1f}, "
      f"trips of the search for two small subdiagonals {t[3] / nm:.1f}, reflector steps "
      f"{t[5] / nm:.1f}; QR iterations of the slowest matrix of a test {t[6] / calls:.1f}")
print(f"DLAHQR cycles per test (first lane's clock): splits {t[8] / calls:.0f}, shifts {t[9] / calls:.0f}, search for two small "
      f"subdiagonals {t[10] / calls:.0f}, sweep {t[11] / calls:.0f}")
trips = max(v[4], 1)
print(f"long loops: {v[12] / 1e6:.1f} Mcycles in total, {v[12] / trips:.0f} cycles per test of 64 matrices")
print(f"passes of wavefronts: the two launches that drop long loops {v[6]} passes, {v[13] / max(v[6], 1):.0f} cycles each; the launch for long loops {v[7]} passes, "
      f"{v[14] / max(v[7], 1):.0f} cycles each; longest pass {v[15]} cycles")
