#!/usr/bin/env python3
"""Headline benchmark: batched CarModel N=40 iLQR solves per second on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]

One step = one pass of the hot path (ilqr_solve_batch) over one batch of B synthetic
problems per GPU whose inputs are already resident in HBM in the library's strided
structure-of-arrays layout.  Workload (per GPU, weak scaling): CarModel(r=2, v_target=2),
N=40, max_iter=50, max_linesearch_iters=20, d_tol=1e-3; x0 ~ U([-4,4]^2) x U(-pi,pi) x
U(0,3) x U(-0.3,0.3), u_init ~ N(0, 1e-4^2), default_rng(1234 + rank); B = 2^20.

For N > 1 the driver starts one process per GPU (torch.distributed.run); every rank solves
its own shard with no data-path collective, then the per-trajectory costs / iteration
counts / exit codes are all-gathered over RCCL, which is the path's only exchange.

Rank 0 prints ONE JSON line.  Besides the contract fields it carries
  roofline:     the dominant kernel, the backward pass (pipe_backward_kernel): algorithmic fp64
                flops of the backward passes actually run (dense formulas, the batch's own
                iteration / regularisation counts) / that kernel's summed device time in one
                solve, timed with HIP events around each of its launches on the solve's stream
                (ILQR_FLAG_TIMING | ILQR_FLAG_SINGLE_STREAM, one extra solve after the timed
                region, kernels back to back), against the fp64 peak.  `line_search` gives the same for the memory-bound forward-pass kernels
                against the HBM peak, `solve` for all kernels of a solve together;
  cpu_baseline: the CPU checker (oracle/, a C restatement of the reference solver) on a
                bounded sample of the same batch, one OpenMP thread per host core, once per eigenvalue
                rule: `rule` "lapack" (np.linalg.eigvals restated: what the reference computes; the
                top-level value) and, under "ldl", the checker running the kernels' default LDL' rule
                (the like-for-like partner of the headline `value`);
  faithful:     the same batch solved once more with ILQR_FLAG_LAPACK_EIGVALS (every test of the
                regularisation loop is LAPACK's, as in control/iLQRController.py:110), timed after
                the headline region: throughput, exits, iterations, and how far the population is
                from the imported reference's own (tests/golden/car_population_ref.npz).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP64_PEAK_TFLOPS = 78.6          # MI355X fp64: matrix (MFMA) peak = vector peak = half the 157.3 TFLOP/s fp32 rate
HBM_PEAK_GBS = 8000.0


def car_inputs(B, seed, N=40):
    rng = np.random.default_rng(seed)
    x0 = np.column_stack([rng.uniform(-4, 4, B), rng.uniform(-4, 4, B), rng.uniform(-np.pi, np.pi, B),
                          rng.uniform(0, 3, B), rng.uniform(-0.3, 0.3, B)])
    u0 = rng.standard_normal((B, N - 1, 2)) * 1e-4
    return x0, u0


def algorithmic_flops(n, m, S, ops, iters, regu_rounds, regu_stages, ls_total, B):
    """Dense flop count of the solves actually performed (FMA = 2, sin/cos/tan/sqrt/div = 1, no
    sparsity or symmetry credit); sums over the batch are passed in.  Returns (initial rollouts,
    backward passes, line-search forward passes)."""
    F_dyn, F_cost, F_costf = ops["dyn"]["flops"], ops["cost_stage"]["flops"], ops["cost_final"]["flops"]
    F_deriv = ops["stage"]["flops"]
    base = (4 * n**3 + 8 * n * n * m + 6 * n * m * m + m**3 / 3 + (n + m) ** 3 / 3 + 6 * n * n + 9 * n * m + 5 * m * m
            + 4 * n + 5 * m + 2 + F_deriv)
    per_round = (n + m) ** 3 / 3 + 2 * (n * n + n * m + m * m)
    per_regu_stage = 2 * n**3 + 2 * n * n * m + 2 * n * m * m
    F_fwd = n + 2 * m * n + 3 * m + F_dyn + F_cost
    init = B * (S * (F_dyn + F_cost) + F_costf)
    backward = iters * (S * base + ops["final"]["flops"]) + regu_rounds * per_round + regu_stages * per_regu_stage
    forward = ls_total * (S * F_fwd + F_costf)
    return float(init), float(backward), float(forward)


def algorithmic_bytes(nx, m, N, B):
    """Bytes a batch has to move whatever the implementation: the caller's x0 and u in, x, u and J plus 12 bytes of
    counters out (8-byte doubles, no padding).  CarModel N = 40: 2 908 bytes per solve.  Gains and intermediate
    trajectories are not counted: a solve that kept them on chip would not move them."""
    S = N - 1
    return float(B * (8 * (nx + S * m) + 8 * (N * nx + S * m + 1) + 12))


def streamed_bytes(nx, n, m, N, B, iters, ls_total, ls_searches):
    """Model of what THIS implementation moves: one kernel per phase of an iteration, every pass streaming its
    operands through HBM once.  The caller's arrays as above; per backward pass x, u in and d, K out; per
    line-search evaluation x, u, d, K in; per search the accepted x, u out.  Returns (solve total, backward,
    line search)."""
    S = N - 1
    traj = N * nx + S * m
    gains = S * (m + m * n)
    backward = iters * 8 * (traj + gains)
    forward = ls_total * 8 * (traj + gains) + ls_searches * 8 * traj
    return float(algorithmic_bytes(nx, m, N, B) + backward + forward), float(backward), float(forward)


def library_fingerprint():
    """sha256 over the kernel sources the solver library is built from: what profiles/traffic_latest.json must have
    been collected on to be reported as this run's traffic."""
    import hashlib
    from ilqr_amd import build as recipe
    h = hashlib.sha256()
    for f in sorted(recipe.sources()):
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def exit_fractions(status_counts):
    tot = float(sum(status_counts)) or 1.0
    return {"d_tol": status_counts[0] / tot, "deltaJ<1": status_counts[1] / tot, "max_iter": status_counts[2] / tot,
            "linalg": status_counts[3] / tot}


def faithful_record(dev, solver_args, x0, u_init, N, steps):
    """The headline batch under the reference's own eigenvalue rule (ILQR_FLAG_LAPACK_EIGVALS), after the headline
    region.  Also solves the 8192 starts of the reference population fixture and compares one by one."""
    from ilqr_amd import _lib
    from ilqr_amd.batch import BatchSolver
    B = x0.shape[-1]
    solver = BatchSolver(*solver_args, device=dev, flags=_lib.FLAG_LAPACK_EIGVALS)
    out = solver.alloc_outputs(B)
    u = torch.empty_like(u_init)
    x = torch.empty((N, 5, B), dtype=torch.float64, device=dev)
    # population fixture first: it also loads the kernels of this mode before anything is timed
    rec = {}
    gpath = os.path.join(ROOT, "tests", "golden", "car_population_ref.npz")
    if N == 40 and os.path.exists(gpath):
        g = np.load(gpath)
        n = int(g["n"])
        px0, pu0 = car_inputs(n, 1234)
        r = solver.solve(px0, pu0)
        torch.cuda.synchronize(dev)
        it, st, rr = (t.cpu().numpy() for t in (r.iters, r.status, r.regu_rounds))
        same = (it == g["iters"]) & (st == g["status"]) & (rr == g["regu_rounds"])
        calm = g["max_abs_q"] < 1e5
        ref_counts = np.bincount(g["status"], minlength=4).tolist()
        rec["reference_population"] = {
            "fixture": "tests/golden/car_population_ref.npz: the imported reference iLQR on %d starts of this workload" % n,
            "ends_like_the_reference": float(same.mean()),
            "ends_like_the_reference_max_abs_q_below_1e5": float(same[calm].mean()), "solves_below_1e5": int(calm.sum()),
            "exit_counts": dict(zip(("d_tol", "deltaJ<1", "max_iter", "linalg"), np.bincount(st, minlength=4).tolist())),
            "reference_exit_counts": dict(zip(("d_tol", "deltaJ<1", "max_iter", "linalg"), ref_counts)),
            "mean_iters": float(it.mean()), "reference_mean_iters": float(g["iters"].mean()),
            "note": "ends like the reference = same iteration count, exit reason and number of regularisation rounds"}
        rec["reference_exit_fractions"] = exit_fractions(ref_counts)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        u.copy_(u_init)
        solver.solve_soa(x0, u, x, out)
    torch.cuda.synchronize(dev)
    dt = (time.perf_counter() - t0) / steps
    status = torch.bincount(out["status"], minlength=4).cpu().tolist()
    rec.update({"flag": "ILQR_FLAG_LAPACK_EIGVALS", "batch": B, "steps": steps, "value": B / dt, "unit": "solves/s",
                "ms_per_step": 1e3 * dt, "mean_iters": float(out["iters"].double().mean().item()),
                "mean_regu_rounds": float(out["regu_rounds"].double().mean().item()),
                "mean_ls_evals": float(out["ls_total"].double().mean().item()),
                "exit_counts": dict(zip(("d_tol", "deltaJ<1", "max_iter", "linalg"), status)),
                "exit_fractions": exit_fractions(status),
                "replayed_iterations": int(out["replayed"].sum().item())})
    return rec


GATHERED = ("J", "iters", "status")      # what the ranks exchange at the end of a step: 16 bytes per trajectory


def gather_buffers(world, B, out, dev):
    return {k: torch.empty(world * B, dtype=out[k].dtype, device=dev) for k in GATHERED}


def gather_step(dist, gathered, out):
    """The path's only exchange: every rank's per-trajectory costs, iteration counts and exit codes to every rank."""
    for k, g in gathered.items():
        dist.all_gather_into_tensor(g, out[k])


def max_over_ranks(dist, seconds, dev):
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=1 << 20, help="trajectories per GPU")
    ap.add_argument("--horizon", type=int, default=40, help="knot points N (40 = the headline workload)")
    ap.add_argument("--cpu-sample", type=int, default=1 << 17, help="trajectories timed on the host cores (0 = skip)")
    ap.add_argument("--lapack-eigvals", action="store_true",
                    help="ILQR_FLAG_LAPACK_EIGVALS: every PSD test the way NumPy makes it (not the headline line)")
    ap.add_argument("--check", type=int, default=2048, help="trajectories checked against the CPU checker (0 = skip)")
    ap.add_argument("--faithful-batch", type=int, default=-1,
                    help="trajectories of the same batch solved again under ILQR_FLAG_LAPACK_EIGVALS for the `faithful` "
                         "record (-1 = the whole batch, 0 = skip; rank 0 of a 1-GPU run only)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist = None
    dev = torch.device("cuda", local)

    from ilqr_amd.batch import BatchSolver
    B, N, nx, m, n = a.batch, a.horizon, 5, 2, 5
    S = N - 1
    from ilqr_amd import _lib as _flags
    solver = BatchSolver(0, [2.0, 2.0], N, 50, 20, 1e-3, device=dev,
                         flags=_flags.FLAG_LAPACK_EIGVALS if a.lapack_eigvals else 0)
    x0_np, u0_np = car_inputs(B, 1234 + rank, N)
    x0 = torch.as_tensor(x0_np).to(dev).t().contiguous()                  # [nx, B]
    u_init = torch.as_tensor(u0_np).to(dev).permute(1, 2, 0).contiguous()  # [S, m, B]
    u = torch.empty_like(u_init)
    x = torch.empty((N, nx, B), dtype=torch.float64, device=dev)
    out = solver.alloc_outputs(B)
    gathered = None
    if dist is not None:
        gathered = gather_buffers(world, B, out, dev)

    def step(ev=None):
        u.copy_(u_init)                       # run_ilqr optimises u in place: restore the initial guess
        if ev:
            ev[0].record()
        solver.solve_soa(x0, u, x, out)
        if ev:
            ev[1].record()
        if dist is not None:
            gather_step(dist, gathered, out)

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    sync()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(evs[k])
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        elapsed = max_over_ranks(dist, elapsed, dev)
    kernel_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in evs]))

    # one more solve with an event pair around every kernel launch (outside the timed region).  It runs the
    # whole batch on one stream, so that a kernel's duration is its own: in the timed region the batch is cut
    # into parts whose kernels overlap on the GPU and stretch each other.
    from ilqr_amd import _lib
    # iterations the timed solves did not run because they repeat the one before them exactly (ILQR_FLAG_NO_REPLAY in
    # include/ilqr_hip.h); the solve below runs every iteration, so that the counts it leaves are the work its kernels did
    replayed = int(out["replayed"].sum().item())
    replayed_solves = int((out["replayed"] > 0).sum().item())
    solver.flags |= _lib.FLAG_TIMING | _lib.FLAG_SINGLE_STREAM | _lib.FLAG_NO_REPLAY
    step()
    sync()
    phases = solver.last_timing()
    solver.flags &= ~(_lib.FLAG_TIMING | _lib.FLAG_SINGLE_STREAM | _lib.FLAG_NO_REPLAY)

    # counts of the work actually done (rank 0's shard; every shard is statistically alike)
    iters = int(out["iters"].sum().item())
    regu_rounds = int(out["regu_rounds"].sum().item())
    regu_stages = int(out["regu_stages"].sum().item())
    ls_total = int(out["ls_total"].sum().item())
    status = torch.bincount(out["status"], minlength=4).cpu().tolist()

    if rank == 0:
        ops = json.load(open(os.path.join(ROOT, "ilqr_amd", "csrc", "gen", "opcounts.json")))["car"]
        # every iteration that got past the deltaJ test ran one search: iterations minus the ones that ended a solve
        # in the backward kernel (deltaJ < 1 or a LinAlgError)
        searches = iters - status[1] - status[3]
        f_init, f_bwd, f_fwd = algorithmic_flops(n, m, S, ops, iters, regu_rounds, regu_stages, ls_total, B)
        b_alg = algorithmic_bytes(nx, m, N, B)
        b_all, b_bwd, b_fwd = streamed_bytes(nx, n, m, N, B, iters, ls_total, searches)
        flops = f_init + f_bwd + f_fwd
        tfl = flops / (kernel_ms * 1e-3) / 1e12
        gbs = b_alg / (kernel_ms * 1e-3) / 1e9
        bw_ms, bw_launches = phases["backward"]
        ls_ms = phases["forward1"][0] + phases["backtrack"][0] + phases["commit"][0]
        bw_tfl = f_bwd / (bw_ms * 1e-3) / 1e12
        ls_gbs = b_fwd / (ls_ms * 1e-3) / 1e9
        # HBM bytes from the hardware counters cannot be read from inside the run: they come from the last
        # rocprofv3 --pmc passes over this same command (tools/collect_traffic.py), kept in profiles/
        traffic, traffic_note = None, "no profiles/traffic_latest.json for this workload"
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath) and B == 1 << 20 and N == 40 and not a.lapack_eigvals:
            try:
                traffic = json.load(open(tpath))
                # counters collected on other kernel sources than the ones this library is built from say nothing
                # about this run: refused
                fp = library_fingerprint()
                if traffic.get("library_fingerprint") != fp:
                    traffic_note = (f"profiles/traffic_latest.json was collected on library sources "
                                    f"{traffic.get('library_fingerprint')}, this run is {fp}: not reported")
                    traffic = None
                else:
                    traffic_note = "library sources match (" + fp + ")"
            except Exception:
                traffic = None
        value = world * B * a.steps / elapsed
        res = {
            "metric": f"iLQR solves/sec (CarModel N={N}, batched)", "value": value, "unit": "solves/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": 1e3 * elapsed / a.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"CarModel(r=2,v_target=2) N={N} max_iter=50 max_ls=20 d_tol=1e-3, x0 uniform box, "
                                   "u_init N(0,1e-8), seed 1234+rank", "batch_per_gpu": B, "global_batch": world * B,
                       "layout": "strided SoA resident in HBM", "parallelism": f"shard{world}"},
            "ms_per_iter": 1e3 * elapsed / a.steps / max(iters / B, 1e-9),
            "mean_iters": iters / B, "mean_ls_evals": ls_total / B, "mean_regu_rounds": regu_rounds / B,
            "exit_counts": {"d_tol": status[0], "deltaJ<1": status[1], "max_iter": status[2], "linalg": status[3]},
            "replay": {"solves": replayed_solves, "iterations": replayed,
                       "note": "iterations of the timed solves that were not run because they repeat the previous one "
                               "exactly (outputs identical bit for bit, tests/test_replay.py); the roofline solve runs "
                               "every iteration"},
            "roofline": {"kernel": "pipe_backward_kernel<CarModel>", "bound": "valu_fp64",
                         "bound_note": "fp64 arithmetic on the vector ALUs (7x7 per-lane matrices, no MFMA instruction "
                                       "in the kernel); 78.6 TFLOP/s is the fp64 vector peak",
                         "achieved": bw_tfl, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": bw_tfl / FP64_PEAK_TFLOPS,
                         "traffic": (traffic or {}).get("backward_hbm_bytes_per_solve"),
                         "traffic_source": "profiles/traffic_latest.json: FETCH_SIZE and WRITE_SIZE of separate rocprofv3 --pmc "
                                           "passes over this command (tools/collect_traffic.py), not collected in this run; "
                                           + traffic_note,
                         "timing": "HIP events around each launch, one extra solve of the same batch on a single stream "
                                   "(ILQR_FLAG_TIMING | ILQR_FLAG_SINGLE_STREAM | ILQR_FLAG_NO_REPLAY); the timed region "
                                   "overlaps 3 parts",
                         "launches_per_solve": bw_launches, "kernel_ms_per_solve": bw_ms,
                         "avg_launch_ms": bw_ms / max(bw_launches, 1), "algorithmic_flops_per_solve": f_bwd,
                         "line_search": {"kernels": "pipe_forward1 / pipe_backtrack_group / pipe_commit", "bound": "hbm",
                                         "achieved": ls_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                         "frac": ls_gbs / HBM_PEAK_GBS,
                                         "traffic": (traffic or {}).get("line_search_hbm_bytes_per_solve"),
                                         "achieved_is": "streamed-bytes model of this implementation (every pass "
                                                        "through HBM once) / kernel time; the algorithmic bytes of a "
                                                        "solve are the caller's arrays only, see solve",
                                         "kernel_ms_per_solve": ls_ms, "streamed_bytes_model": b_fwd},
                         "solve": {"kernel_ms": kernel_ms, "tflops": tfl, "frac_fp64_peak": tfl / FP64_PEAK_TFLOPS,
                                   "algorithmic_flops": flops,
                                   "algorithmic_bytes": b_alg, "algorithmic_bytes_per_solve": b_alg / B,
                                   "gbs": gbs, "frac_hbm_peak": gbs / HBM_PEAK_GBS,
                                   "streamed_bytes_model": b_all, "streamed_gbs": b_all / (kernel_ms * 1e-3) / 1e9,
                                   "traffic": (traffic or {}).get("solve_hbm_bytes"),
                                   "traffic_over_algorithmic": ((traffic or {}).get("solve_hbm_bytes") or 0) / b_alg or None},
                         "phase_ms": {k: v[0] for k, v in phases.items() if k != "iterations"},
                         "phase_launches": {k: v[1] for k, v in phases.items() if k != "iterations"}},
        }
        # (whatever goes wrong in the additions below, the line with the timed result is still printed)
        try:
            # ---- parity spot check and CPU baseline (rank 0, N=1 only for the baseline) ----
            if a.check or (a.cpu_sample and world == 1):
                from oracle import c_oracle as co
                co.build()
                cores = min(len(os.sched_getaffinity(0)), int(os.environ.get("OMP_NUM_THREADS", "16") or 16))
                par = co.pack_car(2.0, 2.0)
                if a.check:
                    # The first c trajectories against the CPU checker.  Which solves CAN be compared one by one is decided
                    # the way the fixtures under tests/golden decide it: the checker solves them a second time with every
                    # input changed in its 14th digit, and a solve counts as reproducible if both runs take the same
                    # decisions and end within 1e-8 of each other (ill-conditioned solves amplify rounding differences
                    # until they do not; those are compared as a population by the tests, not here).
                    c = min(a.check, B)
                    rng = np.random.default_rng(99)
                    ref = co.solve_batch("car", par, N, 50, 20, 1e-3, x0_np[:c], u0_np[:c], nthreads=cores)
                    ref2 = co.solve_batch("car", par, N, 50, 20, 1e-3,
                                          x0_np[:c] * (1.0 + 1e-14 * rng.choice([-1.0, 1.0], size=x0_np[:c].shape)),
                                          u0_np[:c] * (1.0 + 1e-14 * rng.choice([-1.0, 1.0], size=u0_np[:c].shape)), nthreads=cores)
                    i1, i2 = ref["info"], ref2["info"]
                    scale = np.abs(ref["x"]).reshape(c, -1).max(1)
                    with np.errstate(invalid="ignore"):
                        drift = np.abs(ref["x"] - ref2["x"]).reshape(c, -1).max(1) / scale
                    repro = ((i1["iters"] == i2["iters"]) & (i1["status"] == i2["status"]) & (i1["ls_total"] == i2["ls_total"])
                             & (i1["regu_rounds"] == i2["regu_rounds"]) & (drift <= 1e-8))
                    it = out["iters"][:c].cpu().numpy()
                    st = out["status"][:c].cpu().numpy()
                    same = (it == i1["iters"]) & (st == i1["status"])
                    xg = x[:, :, :c].permute(2, 0, 1).cpu().numpy()
                    with np.errstate(invalid="ignore"):
                        err = np.abs(xg - ref["x"]).reshape(c, -1).max(1) / scale
                    sel = err[repro & same] if (repro & same).any() else np.array([np.nan])
                    # the bar of the tests (tests/helpers.py): 1e-8, or 100 times what the checker's own answer moves under
                    # the 1e-14 perturbation where that is more
                    bar = np.maximum(1e-8, 100.0 * drift)[repro & same] if (repro & same).any() else np.array([np.nan])
                    calm = i1["max_abs_q"] < 1e5
                    # the reproducible solves that are NOT within 1e-8, one by one
                    idx = np.where(repro & same)[0]
                    beyond = [dict(index=int(b), rel_x_err=float(err[b]), checker_own_drift=float(drift[b]),
                                   exit=co.EXIT_NAMES[int(st[b])], iters=int(it[b]), max_abs_q=float(i1["max_abs_q"][b]))
                              for b in idx[err[idx] > 1e-8][:16]]
                    res["parity_check"] = {"beyond_1e-8": beyond,"checked": int(c), "reproducible": int(repro.sum()),
                                           "reproducible_same_decisions": int((repro & same).sum()),
                                           "max_rel_x_err_reproducible": float(np.nanmax(sel)),
                                           "median_rel_x_err_reproducible": float(np.nanmedian(sel)),
                                           "frac_rel_x_err_le_1e-8_reproducible": float((sel <= 1e-8).mean()),
                                           "frac_within_parity_bar_reproducible": float((sel <= bar).mean()),
                                           "all_same_iters_and_exit": int(same.sum()),
                                           "max_abs_q_below_1e5": int(calm.sum()),
                                           "max_abs_q_below_1e5_same_decisions": int((calm & same).sum()),
                                           "note": "reproducible = the CPU checker repeats its own decisions and ends within 1e-8 when "
                                                   "its inputs are changed in the 14th digit"}
                if a.cpu_sample and world == 1:
                    c = min(a.cpu_sample, B)
                    t1 = time.perf_counter()
                    co.solve_batch("car", par, N, 50, 20, 1e-3, x0_np[:c], u0_np[:c], nthreads=cores, rule="lapack")
                    dt = time.perf_counter() - t1
                    t1 = time.perf_counter()
                    co.solve_batch("car", par, N, 50, 20, 1e-3, x0_np[:c], u0_np[:c], nthreads=cores, rule="ldl")
                    dt2 = time.perf_counter() - t1
                    res["cpu_baseline"] = {"value": c / dt, "unit": "solves/s", "cores": cores, "kind": "port",
                                           "rule": "lapack",
                                           "rule_note": "np.linalg.eigvals restated (DGEEV): the reference's own test, the "
                                                        "partner of the `faithful` record; the headline `value` runs the "
                                                        "kernels' default LDL' test, whose CPU partner is `ldl`",
                                           "sample": f"first {c} trajectories of the same batch, C restatement of the "
                                                     f"reference solver (oracle/), OpenMP over {cores} threads, {dt:.1f} s",
                                           "ldl": {"value": c / dt2, "unit": "solves/s", "rule": "ldl",
                                                   "sample": f"the same {c} trajectories with the kernels' LDL' rule, {dt2:.1f} s"}}
            if world == 1 and a.faithful_batch != 0 and not a.lapack_eigvals:
                fb = B if a.faithful_batch < 0 else min(a.faithful_batch, B)
                res["faithful"] = faithful_record(dev, (0, [2.0, 2.0], N, 50, 20, 1e-3), x0[:, :fb].contiguous(),
                                                  u_init[:, :, :fb].contiguous(), N, 1)
                res["faithful"]["vs_headline"] = res["faithful"]["value"] / value
        except Exception as e:      # noqa: BLE001
            res["additions_failed"] = f"{type(e).__name__}: {e}"
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
