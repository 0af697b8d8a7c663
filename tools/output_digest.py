#!/usr/bin/env python3
"""sha256 of every output of one headline car solve (and of a horizon-200 one), to compare two builds of the library bit
for bit:   ILQR_HIP_LIB=<lib.so> python tools/output_digest.py [batch] [flags]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ilqr_amd.batch import BatchSolver  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for N, b in ((40, B), (200, max(B // 16, 256))):
    s = BatchSolver(0, [2.0, 2.0], N, 50, 20, 1e-3, flags=flags, trace_len=50)
    x0, u0 = bench.car_inputs(b, 1234, N)
    r = s.solve(x0, u0)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    parts = {}
    for k in ("x", "u", "J0", "J", "dmax", "max_abs_q", "iters", "status", "regu_rounds", "ls_total", "regu_stages", "last_search"):
        t = getattr(r, k).contiguous().cpu().numpy().tobytes()
        parts[k] = hashlib.sha256(t).hexdigest()[:10]
        h.update(t)
    for k, v in sorted(r.traces.items()):
        h.update(v.contiguous().cpu().numpy().tobytes())
    print(f"N={N} B={b} flags={flags}: {h.hexdigest()[:16]}  " + " ".join(f"{k}:{v}" for k, v in parts.items()), flush=True)
