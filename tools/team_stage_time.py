"""Time per stage of the 16-lanes-per-trajectory backward kernel on a small plant (the 6-state, 3-control `Tug` of
tests/user_models.py, stated with 12 + 4 like every wide user plant), for comparison with the lane-per-trajectory
kernel's: run under `rocprofv3 --kernel-trace --stats` and read the backward kernel's launches.
    python tools/team_stage_time.py [batch=65536]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from user_models import Tug, TUG_SP, tug_inputs
from ilqr_amd.batch import BatchSolver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
plant = Tug().build()
s = BatchSolver.for_model(plant, TUG_SP["N"], TUG_SP["max_iter"], TUG_SP["max_ls"], TUG_SP["d_tol"]) if hasattr(BatchSolver, "for_model") else None
if s is None:
    from ilqr_amd.iLQRController import iLQR
    ctl = iLQR(plant, TUG_SP["N"], TUG_SP["max_iter"], TUG_SP["max_ls"], TUG_SP["d_tol"])
x0, u0 = tug_inputs(B, 3)
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if s is not None: r = s.solve(x0, u0)
    else: ctl.run_ilqr(x0.copy(), u0.copy())
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"B={B}: {dt * 1e3:.1f} ms per solve of the batch", flush=True)
