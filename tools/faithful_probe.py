#!/usr/bin/env python3
"""One solve of the headline car batch under ILQR_FLAG_LAPACK_EIGVALS, for profiling:
    python tools/faithful_probe.py [batch=131072] [single_stream=1] [steps=1]
Prints seconds per solve; under `rocprofv3 --kernel-trace` the trace shows every launch (tools/trace_summary.py)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ilqr_amd import _lib  # noqa: E402
from ilqr_amd.batch import BatchSolver  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
single = int(sys.argv[2]) if len(sys.argv) > 2 else 1
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
flags = _lib.FLAG_LAPACK_EIGVALS | (_lib.FLAG_SINGLE_STREAM if single else 0)
s = BatchSolver(0, [2.0, 2.0], 40, 50, 20, 1e-3, device=dev, flags=flags)
x0, u0 = bench.car_inputs(B, 1234)
x0s = torch.as_tensor(x0, device=dev).t().contiguous()
us = torch.as_tensor(u0, device=dev).permute(1, 2, 0).contiguous()
x = torch.empty((40, 5, B), dtype=torch.float64, device=dev)
out = s.alloc_outputs(B)
s.solve(x0[:256], u0[:256])      # loads the kernels
torch.cuda.synchronize()
for _ in range(steps):
    u = us.clone()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.solve_soa(x0s, u, x, out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = torch.bincount(out["status"], minlength=4).cpu().tolist()
    print(f"B={B} single_stream={single}: {dt:.3f} s, {B / dt:.0f} solves/s, exits {st}, mean iters "
          f"{out['iters'].double().mean().item():.3f}, rounds {out['regu_rounds'].double().mean().item():.1f}", flush=True)
