"""The N > 1 path on CPU: world size 2 over gloo.  Each rank solves its slice (here with the CPU checker standing in
for the GPU library, which is what runs per rank on a GPU node) and the gathered whole-batch results must equal a
single-process solve of the whole batch bit for bit."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ilqr_amd.sharding import gather_results, shard_bounds, solve_sharded  # noqa: E402


def test_shard_bounds_cover_the_batch():
    for B in (1, 2, 7, 64, 1000, 1 << 20):
        for world in (1, 2, 3, 8):
            spans = [shard_bounds(B, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _inputs(B):
    rng = np.random.default_rng(77)
    x0 = np.column_stack([rng.uniform(-4, 4, B), rng.uniform(-4, 4, B), rng.uniform(-np.pi, np.pi, B),
                          rng.uniform(0, 3, B), rng.uniform(-0.3, 0.3, B)])
    return x0, rng.standard_normal((B, 39, 2)) * 1e-4


def _oracle_solve(x0, u):
    from oracle import c_oracle as co
    r = co.solve_batch("car", co.pack_car(2.0, 2.0), 40, 50, 20, 1e-3, x0.numpy(), u.numpy(), nthreads=2)
    return dict(J=torch.as_tensor(r["info"]["J_last"].copy()), iters=torch.as_tensor(r["info"]["iters"].copy()),
                status=torch.as_tensor(r["info"]["status"].copy()), x=torch.as_tensor(r["x"]), u=torch.as_tensor(r["u"]))


def _worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x0, u = _inputs(B)
        local, whole, (lo, hi) = solve_sharded(_oracle_solve, torch.as_tensor(x0), torch.as_tensor(u))
        full = gather_results({"x_last": local["x"][:, -1, :].contiguous()}, B)
        q.put((rank, lo, hi, {k: v.numpy() for k, v in whole.items()}, full["x_last"].numpy()))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_two_ranks_match_single_process():
    B, world = 101, 2          # odd on purpose: slices of 51 and 50
    from oracle import c_oracle as co
    co.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    x0, u = _inputs(B)
    ref = _oracle_solve(torch.as_tensor(x0), torch.as_tensor(u))
    spans = sorted((lo, hi) for _, lo, hi, _, _ in got)
    assert spans == [(0, 51), (51, 101)]
    for rank, lo, hi, whole, x_last in got:
        for k in ("J", "iters", "status"):
            a, b = whole[k], ref[k].numpy()
            assert a.shape == b.shape
            assert np.array_equal(a, b, equal_nan=True) if a.dtype.kind == "f" else np.array_equal(a, b), (rank, k)
        assert np.array_equal(x_last, ref["x"][:, -1, :].numpy(), equal_nan=True)


def _bench_worker(rank, world, port, B, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        dev = torch.device("cpu")
        out = dict(J=torch.arange(B, dtype=torch.float64) + 1000.0 * rank, iters=torch.arange(B, dtype=torch.int32) + 7 * rank,
                   status=torch.full((B,), rank, dtype=torch.int32))
        g = bench.gather_buffers(world, B, out, dev)
        bench.gather_step(dist, g, out)
        slowest = bench.max_over_ranks(dist, 1.0 + rank, dev)
        q.put((rank, {k: v.numpy() for k, v in g.items()}, slowest))
    finally:
        dist.barrier()
        dist.destroy_process_group()


def test_bench_gather_path_with_two_ranks():
    """bench.py's own exchange (all-gather of J / iters / status, maximum of the ranks' times) with world size 2."""
    B, world = 37, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, g, slowest in got:
        assert slowest == 2.0
        assert np.array_equal(g["J"], np.concatenate([np.arange(B) + 1000.0 * r for r in range(world)]))
        assert np.array_equal(g["iters"], np.concatenate([np.arange(B) + 7 * r for r in range(world)]))
        assert np.array_equal(g["status"], np.repeat(np.arange(world), B))
