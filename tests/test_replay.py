"""Solves that reach a state from which every iteration repeats the last one exactly are not iterated to max_iter.
An abandoned line search keeps the trajectory (control/iLQRController.py:69-79), so the next backward pass and the next
search's candidates are the same numbers; if the search left J = NaN, or followed an abandoned search and left the J
it started from, every later iteration is this one again, and the kernels write what those iterations would leave
behind.  Every output must equal, bit for bit, what ILQR_FLAG_NO_REPLAY gives by running them; the reference's own
records show the pattern this rests on."""
import numpy as np
import pytest

from conftest import load_golden
from test_gpu_parity import _cfg, _inputs, torch_gpu   # noqa: F401

MID = dict(car=0, acrobot=1, quad_ea=2, quad_quat=3)


def test_the_reference_repeats_itself_after_two_abandoned_searches():
    """In the reference's records of 256 headline solves (car_box256.npz: evaluations and accept flag per iteration, J
    after every iteration) an abandoned search that follows an abandoned search and leaves J unchanged is followed by
    the same search, with the same J, in every later iteration."""
    g = load_golden("car_box256")
    ls, acc, J, iters = g["ls_evals"], g["accepted"], g["J"], g["iters"]
    assert J.shape == ls.shape
    frozen_solves = repeats = 0
    for b in range(len(iters)):
        n = int(iters[b])
        for k in range(1, n):
            if ls[b, k] == 0 or ls[b, k - 1] == 0:
                continue
            Jk, Jp = J[b, k], J[b, k - 1]      # J after the searches of iterations k + 1 and k
            same = (Jk == Jp) or (Jk != Jk and Jp != Jp)
            if not acc[b, k] and not acc[b, k - 1] and same:
                rest = slice(k, n)
                searched = ls[b, rest] > 0
                assert (ls[b, rest][searched] == ls[b, k]).all() and not acc[b, rest][searched].any(), b
                frozen_solves += 1
                repeats += int(searched.sum()) - 1
                break
    assert frozen_solves >= 3 and repeats >= 50, (frozen_solves, repeats)


def _all_outputs(kind, cfg, x0, u0, flags, trace):
    from ilqr_amd.batch import BatchSolver
    s = BatchSolver(MID[kind], cfg["par"], cfg["N"], cfg["max_iter"], cfg["max_ls"], cfg["d_tol"], trace_len=trace,
                    flags=flags)
    r = s.solve(x0, u0)
    cp = lambda t: t.cpu().numpy()
    out = dict(x=cp(r.x), u=cp(r.u), J0=cp(r.J0), J=cp(r.J), dmax=cp(r.dmax), maxq=cp(r.max_abs_q), iters=cp(r.iters),
               status=cp(r.status), regu=cp(r.regu_rounds), stages=cp(r.regu_stages), ls_total=cp(r.ls_total),
               last=cp(r.last_search))
    out.update({k: cp(v) for k, v in r.traces.items()})
    return out, cp(r.replayed)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,B,flags,trace", [("car", 16384, 0, 50), ("car", 4096, 2, 0), ("car", 4096, 1, 7),
                                                ("car", 2048, 32, 50),
                                                ("acrobot", 2048, 0, 20), ("acrobot", 1024, 2, 5),
                                                ("acrobot", 1024, 1, 0), ("quad_ea", 256, 0, 20),
                                                ("quad_quat", 256, 0, 20),
                                                # batches that run as several parts on internal streams
                                                ("acrobot", 98304, 0, 0), ("quad_ea", 32768, 0, 3)])
def test_replayed_iterations_leave_what_running_them_leaves(torch_gpu, oracle, kind, B, flags, trace):
    from ilqr_amd import _lib
    rng = np.random.default_rng(dict(car=1234, acrobot=2, quad_ea=3, quad_quat=4)[kind])
    c = _cfg(oracle, kind)
    x0, u0 = _inputs(kind, rng, B, c["N"])
    if kind in ("quad_ea", "quad_quat"):      # far from hover, where searches are abandoned
        u0 = 2.0 * rng.standard_normal(u0.shape)
    fast, replayed = _all_outputs(kind, c, x0, u0, flags, trace)
    full, none = _all_outputs(kind, c, x0, u0, flags | _lib.FLAG_NO_REPLAY, trace)
    assert not none.any()
    for k in full:
        assert np.array_equal(fast[k], full[k], equal_nan=True), (kind, k)
    # the path is exercised: some solves were cut short, each by at least one iteration, and all of them ran out of
    # iterations as far as the outputs say
    cut = replayed > 0
    print(f"{kind}: {int(cut.sum())} of {B} solves replayed, {int(replayed.sum())} of {int(fast['iters'].sum())} iterations")
    if kind != "car" or flags & 32:
        assert cut.any(), "no solve of this batch was replayed: the test compares nothing"
    assert (fast["iters"][cut] == c["max_iter"]).all() and (fast["status"][cut] == 2).all() and (fast["last"][cut] == 2).all()
