"""ilqr_eigvals_negative (the device statement of `np.any(np.linalg.eigvals(Q) < 0)`, control/iLQRController.py:110)
through the C ABI against every (matrix, LAPACK verdict) pair recorded from the imported reference
(tests/golden/eig_decisions.npz), and against NumPy live on random matrices."""
import ctypes as C

import numpy as np
import pytest

from conftest import load_golden
from test_eigvals_restatement import random_matrix, stage_matrices

pytestmark = pytest.mark.gpu


def gpu_verdicts(mats):
    import torch
    from ilqr_amd import _lib
    assert torch.cuda.is_available(), "this test needs a GPU"
    L = _lib.lib()
    B, w = mats.shape[0], mats.shape[1]
    q = torch.as_tensor(np.ascontiguousarray(mats.reshape(B, w * w).T), device="cuda:0")      # [w*w][B]
    out = torch.full((B,), -7, dtype=torch.int32, device="cuda:0")
    _lib.check(L.ilqr_eigvals_negative(w, C.c_void_p(q.data_ptr()), B, B, C.c_void_p(out.data_ptr()), None), L)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("tag,min_decisions", [("car", 120000), ("ea", 9000), ("quat", 6000)])
def test_device_rule_on_every_recorded_decision_of_the_reference(tag, min_decisions):
    g = load_golden("eig_decisions")
    Q0, Pm, rounds, end = g[tag + "_Q0"], g[tag + "_P"], g[tag + "_rounds"], g[tag + "_end"]
    mats, want = [], []
    for s in range(len(Q0)):
        m, w = stage_matrices(Q0[s], Pm[s], int(rounds[s]), int(end[s]))
        if len(m):
            fin = np.isfinite(m).all(axis=(1, 2))
            mats.append(m[fin]); want.append(w[fin])
    mats, want = np.concatenate(mats), np.concatenate(want)
    assert len(mats) >= min_decisions
    got = gpu_verdicts(mats)
    wrong = np.where(got != want.astype(np.int32))[0]
    # the device takes sqrt(sum of squares) of DNRM2 through an emulation of the x87's extended format: allow the rare
    # last-bit difference there to flip a verdict taken in rounding noise
    assert len(wrong) <= 2e-4 * len(mats), (tag, len(wrong), len(mats), wrong[:10], got[wrong[:10]])


def test_device_rule_against_numpy_on_random_matrices():
    rng = np.random.default_rng(21)
    for n in (2, 3, 4, 5, 6, 7, 8, 16):
        mats = np.stack([random_matrix(rng, n, t) for t in range(4000)])
        with np.errstate(all="ignore"):
            want = np.array([bool(np.any(np.linalg.eigvals(m) < 0)) for m in mats], dtype=np.int32)
        got = gpu_verdicts(mats)
        wrong = np.where(got != want)[0]
        assert len(wrong) <= 4, (n, len(wrong), wrong[:10])
    bad = np.eye(7)[None].repeat(3, 0).copy()
    bad[0, 2, 3] = np.nan; bad[1, 0, 0] = np.inf
    assert gpu_verdicts(bad).tolist() == [-1, -1, 0]


def test_device_rule_on_matrices_whose_reflectors_are_not_plain():
    """Zero patterns that keep DGEBAL a no-op but shorten DLARF's products (a Householder vector that ends in zeros, a
    block with a zero last row, matrices that are Hessenberg or triangular already): the orders 3..8 reduce such a
    reflector with the general routines instead of the register version, and the verdicts stay NumPy's."""
    rng = np.random.default_rng(77)
    for n in (3, 4, 5, 6, 7, 8):
        mats = []
        for t in range(1200):
            A = rng.standard_normal((n, n)) + (t % 3 - 1) * 0.5 * np.eye(n)
            kind = t % 6
            if kind == 0:
                A[n - 1, 0] = 0.0                       # first reflector's vector ends in a zero
            elif kind == 1 and n >= 4:
                A[n - 2:, 0] = 0.0; A[n - 1, 1] = 0.0   # ... in two zeros, and the second one's in one
            elif kind == 2:
                A[n - 1, 1:] = 0.0; A[n - 1, 0] = 1.5   # the block the first reflector is applied to has a zero last row
            elif kind == 3:
                A = np.triu(A, -1)                      # Hessenberg already: every x is zero
            elif kind == 4:
                A = np.triu(A); A[n - 1, 0] = 0.3       # triangular but for one corner entry
            else:
                A[2:, 0] = 0.0; A[3:, 1] = 0.0          # the first two columns reduced already
            mats.append(A)
        mats = np.stack(mats)
        with np.errstate(all="ignore"):
            want = np.array([bool(np.any(np.linalg.eigvals(m) < 0)) for m in mats], dtype=np.int32)
        got = gpu_verdicts(mats)
        wrong = np.where(got != want)[0]
        assert len(wrong) <= 2, (n, len(wrong), wrong[:10])
