"""Quick GPU sanity run: HIP solver vs golden fixtures (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ilqr_amd.batch import BatchSolver
from ilqr_amd import _lib
MID = dict(car=0, acrobot=1, quad_ea=2, quad_quat=3)
for name in sys.argv[1:] or ["car_readme", "car_mpc50", "car_box256"]:
    g = np.load(f"tests/golden/{name}.npz")
    s = BatchSolver(MID[str(g["model"])], g["par"], int(g["N"]), int(g["max_iter"]), int(g["max_ls"]), float(g["d_tol"]), trace_len=64)
    t = time.time()
    r = s.solve(g["x0"], g["u_init"]); torch.cuda.synchronize()
    dt = time.time() - t
    it = r.iters.cpu().numpy(); stt = r.status.cpu().numpy()
    ok = g["stable"].astype(bool) & (g["status"] != 3)
    same = (it == g["iters"]) & (stt == g["status"]) & (r.traces["ls_evals"].cpu().numpy() == g["ls_evals"]).all(1) & (r.regu_rounds.cpu().numpy() == g["regu_rounds"])
    x = r.x.cpu().numpy(); u = r.u.cpu().numpy()
    relx = np.abs(x - g["x"]).reshape(len(it), -1).max(1) / np.maximum(1e-300, np.abs(g["x"]).reshape(len(it), -1).max(1))
    relJ = np.abs(r.J.cpu().numpy() - g["J_last"]) / np.maximum(1e-300, np.abs(g["J_last"]))
    print(f"{name}: B={len(it)} t={dt:.3f}s  stable={ok.sum()}  decisions-same(stable)={same[ok].sum()}/{ok.sum()}  all={same.sum()}/{len(it)}"
          f"  max relx(stable&same)={relx[ok&same].max() if (ok&same).any() else None:.3e}  relJ={relJ[ok&same].max() if (ok&same).any() else None:.3e}")
    bad = np.where(ok & ~same)[0]
    for b in bad[:5]:
        print("   mismatch", b, "iters", it[b], g["iters"][b], "status", stt[b], g["status"][b], "regu", int(r.regu_rounds[b]), g["regu_rounds"][b], "maxq", g["max_abs_q"][b], "margin", g["lam_margin"][b])
