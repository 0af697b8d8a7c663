"""ILQR_FLAG_LAPACK_EIGVALS: what a trajectory's solve leaves does not depend on the company it is solved in.  Under the
flag the backward pass of an iteration is spread over launches and wavefronts by what the trajectories have needed so
far (slots from both ends of the buffer, long regularisation loops in a launch of their own, the lanes of a wavefront
shared between the loops still running); none of it may show in the outputs, which are the reference's per-trajectory
arithmetic (control/iLQRController.py:98-123)."""
import os

import numpy as np
import pytest

from test_gpu_parity import torch_gpu   # noqa: F401

pytestmark = pytest.mark.gpu

KEYS = ("x", "u", "J0", "J", "dmax", "max_abs_q", "iters", "status", "regu_rounds", "regu_stages", "ls_total",
        "last_search")


def _solve(x0, u0, N=40, max_iter=50):
    from ilqr_amd import _lib
    from ilqr_amd.batch import BatchSolver
    s = BatchSolver(0, [2.0, 2.0], N, max_iter, 20, 1e-3, flags=_lib.FLAG_LAPACK_EIGVALS)
    r = s.solve(x0, u0)
    return {k: getattr(r, k).cpu().numpy() for k in KEYS}


def _same(a, b, idx, what):
    for k in KEYS:
        assert np.array_equal(a[k][idx], b[k], equal_nan=True), (what, k)


def test_outputs_do_not_depend_on_the_batch_a_trajectory_is_in(torch_gpu):
    import bench
    B = 8192
    x0, u0 = bench.car_inputs(B, 4321)
    whole = _solve(x0, u0)
    # the batch holds every kind: solves that never leave the screen, solves with long loops, LinAlgError exits
    assert (whole["regu_rounds"] > 2000).sum() >= 50 and (whole["status"] == 2).sum() >= 100
    # sub-batches: contiguous slices, the hardest 64 on their own, the easiest 64 on their own, and singles
    order = np.argsort(whole["regu_rounds"])
    groups = [np.arange(0, 64), np.arange(1000, 1700), order[-64:], order[:64], order[-1:], order[B // 2:B // 2 + 1]]
    for gi, idx in enumerate(groups):
        part = _solve(x0[idx], u0[idx])
        _same(whole, part, idx, f"group {gi}")


@pytest.mark.parametrize("N,max_iter", [(2, 50), (3, 50), (40, 1), (40, 3), (12, 50)])
def test_small_and_ragged_batches(torch_gpu, N, max_iter):
    """Batches of 1, 3, 65 and 130 trajectories (launches with a single, partly filled wavefront per range), the
    shortest horizons and cut-off iteration counts: the same bits as the same trajectories inside a batch of 1 000."""
    import bench
    x0, u0 = bench.car_inputs(1000, 7, N)
    x0[::7] *= 3.0                      # some far starts, so that long loops occur at short horizons too
    whole = _solve(x0, u0, N, max_iter)
    for idx in (np.arange(1), np.arange(5, 8), np.arange(100, 165), np.arange(300, 430)):
        part = _solve(x0[idx], u0[idx], N, max_iter)
        _same(whole, part, idx, f"N {N} max_iter {max_iter} batch {len(idx)}")


def test_outputs_do_not_depend_on_how_wavefronts_are_packed(torch_gpu):
    import bench
    B = 16384
    x0, u0 = bench.car_inputs(B, 99)
    base = _solve(x0, u0)
    saved = {k: os.environ.get(k) for k in ("ILQR_HIP_HEAVY_LANES", "ILQR_HIP_DIRTY_LANES", "ILQR_HIP_PARTS")}
    try:
        for env in (dict(ILQR_HIP_HEAVY_LANES="1"), dict(ILQR_HIP_HEAVY_LANES="4", ILQR_HIP_HEAVY_FILL="1"),
                    dict(ILQR_HIP_DIRTY_LANES="1"), dict(ILQR_HIP_DIRTY_LANES="16", ILQR_HIP_DIRTY_WAVES="1"),
                    dict(ILQR_HIP_PARTS="1")):
            for k in ("ILQR_HIP_HEAVY_LANES", "ILQR_HIP_HEAVY_FILL", "ILQR_HIP_DIRTY_LANES", "ILQR_HIP_DIRTY_WAVES",
                      "ILQR_HIP_PARTS"):
                os.environ.pop(k, None)
            os.environ.update(env)
            other = _solve(x0, u0)
            _same(base, other, slice(None), str(env))
    finally:
        for k in ("ILQR_HIP_HEAVY_LANES", "ILQR_HIP_HEAVY_FILL", "ILQR_HIP_DIRTY_LANES", "ILQR_HIP_DIRTY_WAVES",
                  "ILQR_HIP_PARTS"):
            os.environ.pop(k, None)
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v
