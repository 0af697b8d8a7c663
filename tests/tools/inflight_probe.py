#!/usr/bin/env python3
"""Throughput of the headline workload with F batches in flight: F host threads, each with its own solver
(workspace), buffers and stream, solve batch after batch; ilqr_solve_batch keeps its state per calling thread.

    python tests/tools/inflight_probe.py [--batch B] [--steps K] [--flights 1,2,3]
"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from bench import car_inputs  # noqa: E402
from ilqr_amd.batch import BatchSolver  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--flights", default="1,2,3")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    B, N = a.batch, 40
    x0_np, u0_np = car_inputs(B, 1234, N)
    x0 = torch.as_tensor(x0_np).to(dev).t().contiguous()
    u_init = torch.as_tensor(u0_np).to(dev).permute(1, 2, 0).contiguous()
    ref = None
    for F in [int(f) for f in a.flights.split(",")]:
        sets = []
        for _ in range(F):
            s = BatchSolver(0, [2.0, 2.0], N, 50, 20, 1e-3, device=dev)
            sets.append(dict(solver=s, u=torch.empty_like(u_init), x=torch.empty((N, 5, B), dtype=torch.float64, device=dev),
                             out=s.alloc_outputs(B), stream=torch.cuda.Stream(dev)))

        def worker(w, steps):
            with torch.cuda.stream(w["stream"]):
                for _ in range(steps):
                    w["u"].copy_(u_init)
                    w["solver"].solve_soa(x0, w["u"], w["x"], w["out"])

        def run(total):
            per = [total // F + (1 if i < total % F else 0) for i in range(F)]
            th = [threading.Thread(target=worker, args=(sets[i], per[i])) for i in range(F)]
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize(dev)
            return time.perf_counter() - t0

        run(F)
        dt = run(a.steps)
        J = sets[-1]["out"]["J"].clone()
        if ref is None:
            ref = J
        same = bool(torch.equal(torch.nan_to_num(J), torch.nan_to_num(ref)))
        print(f"flights {F}: {a.steps} steps {dt * 1e3:.1f} ms, {dt * 1e3 / a.steps:.2f} ms/step, "
              f"{B * a.steps / dt / 1e6:.2f} M solves/s, results identical to 1 flight: {same}", flush=True)
        del sets
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
