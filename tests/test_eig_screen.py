"""The certified screen in front of the LAPACK path (ilqr_amd/csrc/eig_screen.hpp), compiled for the host
(tests/tools/screen_check.cpp): whenever it gives a verdict instead of "in doubt", that verdict must be LAPACK's.

Scored against (1) every (matrix, verdict) pair the imported reference recorded (tests/golden/eig_decisions.npz:
NumPy's own np.linalg.eigvals on the matrices its regularisation loop tested), (2) NumPy live on random matrices of the
kinds the solver meets, (3) every test the checker's solver makes on a population of headline car solves (LAPACK
restated, bit for bit NumPy's).  What the screen leaves in doubt goes to the LAPACK path in the kernels, so a doubt
costs time, never parity; the doubt rates asserted here are what keeps ILQR_FLAG_LAPACK_EIGVALS usable."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden

SRC = os.path.join(ROOT, "tests", "tools", "screen_check.cpp")
OUT = os.path.join(ROOT, "oracle", "_build", "libscreen_check.so")


@pytest.fixture(scope="module")
def screen():
    deps = [SRC, os.path.join(ROOT, "ilqr_amd", "csrc", "eig_screen.hpp")]
    if not os.path.exists(OUT) or os.path.getmtime(OUT) < max(os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(OUT), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-o", OUT, SRC])
    S = C.CDLL(OUT)
    S.screen_verdicts.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_void_p]

    def verdicts(Q):
        Q = np.ascontiguousarray(Q, dtype=np.float64)
        out = np.empty(len(Q), np.int32)
        S.screen_verdicts(Q.shape[1], Q.ctypes.data, len(Q), out.ctypes.data)
        return out
    verdicts.lib = S
    return verdicts


def numpy_verdict(Q):
    out = np.empty(len(Q), np.int32)
    for b, q in enumerate(Q):
        try:
            out[b] = 1 if np.any(np.linalg.eigvals(q) < 0) else 0       # control/iLQRController.py:110
        except np.linalg.LinAlgError:
            out[b] = -1
    return out


def test_screen_never_contradicts_the_recorded_reference_decisions(screen):
    g = load_golden("eig_decisions")
    total = certified = 0
    for tag in ("car", "ea", "quat"):
        mats, truth = [], []
        for q0, p, r, e in zip(g[tag + "_Q0"], g[tag + "_P"], g[tag + "_rounds"], g[tag + "_end"]):
            Q = q0.copy()
            for k in range(r + 1 if e == 0 else r):
                mats.append(Q.copy())
                truth.append(1 if k < r else 0)
                with np.errstate(all="ignore"):
                    Q = Q + np.ldexp(p, k)          # the reference's running sum, iLQRController.py:111-115
        mats, truth = np.array(mats), np.array(truth)
        v = screen(mats)
        sure = v != 2
        assert (v[sure] == truth[sure]).all(), (tag, int((v[sure] != truth[sure]).sum()))
        total += len(v)
        certified += int(sure.sum())
        print(tag, len(v), "decisions,", int(sure.sum()), "certified")
    assert total > 140000 and certified > 12000      # (the fixture over-samples the hopeless stages)


def test_screen_against_numpy_on_random_matrices(screen):
    rng = np.random.default_rng(5)
    for w in (2, 3, 5, 7, 8, 16):
        n = 6000 if w <= 8 else 1500
        mats = []
        for k in range(n):
            A = rng.standard_normal((w, w))
            kind = k % 6
            if kind == 0:                                   # positive definite, any conditioning
                Q = A @ A.T + 10.0 ** rng.uniform(-12, 1) * np.eye(w)
            elif kind == 1:                                 # indefinite
                Q = A + A.T
            elif kind == 2:                                 # singular: rank deficient Gram matrix
                Bm = rng.standard_normal((w, max(w - 2, 1)))
                Q = Bm @ Bm.T
            elif kind == 3:                                 # the triangles differ, like the solver's Q in ill-conditioned solves
                Q = A @ A.T + 10.0 ** rng.uniform(-10, 0) * rng.standard_normal((w, w))
            elif kind == 4:                                 # structural zeros: isolated rows and columns
                Q = A @ A.T
                i = rng.integers(w)
                Q[i, :] = 0.0
                Q[:, i] = 0.0
                Q[i, i] = rng.choice([0.0, 0.2, -0.2])
            else:                                           # badly scaled
                D = np.diag(10.0 ** rng.uniform(-6, 6, w))
                Q = D @ (A @ A.T - rng.uniform(0, 2) * np.eye(w)) @ D
            mats.append(Q * 10.0 ** rng.uniform(-3, 8))
        mats = np.array(mats)
        v = screen(mats)
        t = numpy_verdict(mats)
        sure = v != 2
        assert (v[sure] == t[sure]).all(), (w, int((v[sure] != t[sure]).sum()))
        assert sure.mean() > 0.3, (w, sure.mean())
    # entries that are not finite, or far out of range, are never certified
    bad = np.array([np.full((7, 7), np.nan), np.full((7, 7), np.inf), np.eye(7) * 1e200, np.eye(7) * 1e-200, np.zeros((7, 7))])
    assert (screen(bad) == 2).all()


def test_screen_against_the_checkers_lapack_on_a_solve_population(screen, oracle):
    import bench
    L = oracle.lib()
    L.ilqr_oracle_set_eig_observer.argtypes = [C.c_void_p]
    S = screen.lib
    S.screen_tally_reset()
    L.ilqr_oracle_set_eig_observer(C.cast(S.screen_observer, C.c_void_p))
    try:
        x0, u0 = bench.car_inputs(1024, 1234)
        oracle.solve_batch("car", oracle.pack_car(2.0, 2.0), 40, 50, 20, 1e-3, x0, u0)
    finally:
        L.ilqr_oracle_set_eig_observer(None)
    out = (C.c_long * 15)()
    S.screen_tally_get(out)
    t = np.array(list(out)).reshape(5, 3)        # class of max|Q| x [agrees, wrong, in doubt]
    print("tests per class of |Q| [<1e5, <1e8, <1e12, <1e16, beyond] x [agrees with LAPACK, wrong, in doubt]\n", t)
    assert t.sum() > 500000
    assert t[:, 1].sum() == 0
    assert t[0, 2] < 0.10 * t[0].sum()           # well-scaled matrices: few are left to LAPACK
