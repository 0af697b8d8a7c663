"""Time per stage of the 16-lanes-per-trajectory backward kernel on a small plant (the 6-state, 3-control `Tug` of
tests/user_models.py, stated with 12 + 4 like every wide user plant), for comparison with the lane-per-trajectory
kernel's: run under `rocprofv3 --kernel-trace --stats` and read the backward kernel's launches
(tools/trace_summary.py).
    python tools/team_stage_time.py [batch=65536]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

from ilqr_amd.iLQRController import iLQR  # noqa: E402
from user_models import TUG_SP, Tug, tug_inputs  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
ctl = iLQR(Tug().build(), TUG_SP["N"], TUG_SP["max_iter"], TUG_SP["max_ls"], TUG_SP["d_tol"])
x0, u0 = tug_inputs(B, 3)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctl.run_ilqr(x0.copy(), u0.copy())
    torch.cuda.synchronize()
    print(f"B={B}: {(time.perf_counter() - t0) * 1e3:.1f} ms per solve of the batch", flush=True)
