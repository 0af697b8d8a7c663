"""`np.any(np.linalg.eigvals(Q) < 0)` (control/iLQRController.py:110) as the CPU checker decides it.

oracle/eigvals_lapack.h restates LAPACK's DGEEV (eigenvalues only) routine by routine.  Three pins:
  * every (matrix, LAPACK verdict) pair recorded from the imported reference (tests/golden/eig_decisions.npz, made
    by tests/golden/make_eig_decisions.py): the restatement must give LAPACK's verdict on all of them;
  * NumPy itself, live: eigenvalues bit for bit on random matrices of order 1..16 (well scaled, rank deficient,
    nearly symmetric, with isolated rows, scaled to 1e-200 .. 1e300);
  * the library's own routines one by one (DNRM2, DLAPY2, DLANV2, DLARFG, DLARF, DGEBAL, DGEHD2, DLAHQR, called
    through ctypes from the OpenBLAS that NumPy loads).
The two live pins depend on the BLAS kernels NumPy picked for this CPU (the order of additions in DGEMV/DGER/DNRM2):
they are asserted where NumPy runs OpenBLAS' Haswell/SkylakeX/Zen kernel family and skipped elsewhere.  The last test
scores the LDL' rule the checker used before on the recorded pairs, by decade of max|Q|.
"""
import ctypes as C
import glob
import os

import numpy as np
import pytest

from conftest import load_golden

dp = C.POINTER(C.c_double)
i64 = C.c_int64


def P(a):
    return a.ctypes.data_as(dp)


def I(v):
    return C.byref(i64(v))


def D(v):
    return C.byref(C.c_double(v))


@pytest.fixture(scope="module")
def R(oracle):
    here = os.path.dirname(os.path.abspath(oracle.__file__))
    L = C.CDLL(os.path.join(here, "_build", "libeigvals_lapack.so"))
    L.eigr_nrm2.restype = C.c_double
    L.eigr_nrm2.argtypes = [C.c_int, dp, C.c_int]
    L.eigr_lapy2.restype = C.c_double
    L.eigr_lapy2.argtypes = [C.c_double, C.c_double]
    L.eigr_larfg.argtypes = [C.c_int, dp, dp, C.c_int, dp]
    L.eigr_larf.argtypes = [C.c_int, C.c_int, C.c_int, dp, C.c_double, dp, C.c_int, dp]
    L.eigr_gebal.argtypes = [C.c_int, dp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), dp]
    L.eigr_gehd2.argtypes = [C.c_int, C.c_int, C.c_int, dp, C.c_int, dp]
    L.eigr_lanv2.argtypes = [C.c_double] * 4 + [dp]
    L.eigr_lahqr.argtypes = [C.c_int, C.c_int, C.c_int, dp, C.c_int, dp, dp]
    L.eigr_eigvals.argtypes = [C.c_int, dp, dp, dp]
    L.eigr_eigvals_many.argtypes = [C.c_long, C.c_int, dp, dp, dp, C.POINTER(C.c_int)]
    return L


def numpy_openblas():
    """(library, kernel family) of the OpenBLAS NumPy loaded, or (None, reason)."""
    try:
        from threadpoolctl import threadpool_info
        infos = [i for i in threadpool_info() if i.get("internal_api") == "openblas" and "numpy" in i.get("filepath", "")]
    except Exception as e:   # pragma: no cover
        return None, f"threadpoolctl: {e}"
    if not infos:
        return None, "NumPy does not run on OpenBLAS here"
    arch = infos[0].get("architecture", "?")
    if arch not in ("Haswell", "SkylakeX", "Zen", "Cooperlake", "SapphireRapids"):
        return None, f"OpenBLAS kernel family {arch}: the order of operations was established for the AVX2/AVX-512 kernels"
    try:
        return C.CDLL(infos[0]["filepath"]), arch
    except OSError as e:   # pragma: no cover
        return None, str(e)


def random_matrix(rng, n, t):
    kind = t % 8
    if kind == 0:
        A = rng.standard_normal((n, n))
    elif kind == 1:
        B = rng.standard_normal((n, n)); A = B @ B.T
    elif kind == 2:     # rank deficient times a large factor plus a small general part: what a regularised Q looks like
        B = rng.standard_normal((n, max(1, n - 2)))
        A = B @ B.T * 10.0 ** rng.integers(0, 12) + 1e-3 * rng.standard_normal((n, n))
    elif kind == 3:     # a zero row and column (the car's steering direction at the last stage)
        B = rng.standard_normal((n, n)); A = B + B.T
        z = rng.integers(0, n); A[z, :] = 0; A[:, z] = 0
    elif kind == 4:
        B = rng.standard_normal((n, max(1, n - 2))); A = B @ B.T * 10.0 ** rng.integers(0, 16)
        S = rng.standard_normal((n, n)); A = A * (1 + 1e-16 * rng.standard_normal((n, n))) + S + S.T
    elif kind == 5:     # badly scaled rows, isolated rows
        A = rng.standard_normal((n, n)) * 10.0 ** rng.integers(-8, 9, (n, 1))
        if t % 16 == 5:
            A[rng.integers(0, n), :] = 0
    elif kind == 6:     # beyond DGEEV's scaling thresholds
        B = rng.standard_normal((n, n)); A = (B @ B.T - 0.5 * np.eye(n)) * 10.0 ** rng.integers(139, 300)
    else:
        B = rng.standard_normal((n, n)); A = (B + B.T) * 10.0 ** -rng.integers(140, 250)
    return np.ascontiguousarray(A)


def restated_eigvals(R, A):
    n = A.shape[0]
    wr, wi = np.zeros(n), np.zeros(n)
    rc = R.eigr_eigvals(n, P(np.ascontiguousarray(A)), P(wr), P(wi))
    return rc, wr, wi


def test_eigenvalues_equal_numpys_bit_for_bit(R):
    lib, why = numpy_openblas()
    if lib is None:
        pytest.skip(why)
    rng = np.random.default_rng(12)
    bad = []
    for t in range(24000):
        n = int(rng.integers(1, 17))
        A = random_matrix(rng, n, t)
        w = np.linalg.eigvals(A)
        rc, wr, wi = restated_eigvals(R, A)
        im = w.imag if np.iscomplexobj(w) else np.zeros(n)
        if rc != 0 or not (np.array_equal(w.real, wr) and np.array_equal(im, wi)):
            bad.append((t, n, rc))
    assert not bad, bad[:10]


def test_routines_equal_the_librarys_one_by_one(R):
    OB, why = numpy_openblas()
    if OB is None:
        pytest.skip(why)
    if not hasattr(OB, "scipy_dlahqr_64_"):
        pytest.skip("this OpenBLAS does not export scipy_*_64_ symbols")
    OB.scipy_dnrm2_64_.restype = C.c_double
    OB.scipy_dlapy2_64_.restype = C.c_double
    rng = np.random.default_rng(5)
    T = 4000
    for t in range(T):
        n = int(rng.integers(1, 17))
        x = rng.standard_normal(n) * 10.0 ** rng.integers(-3, 4)
        assert OB.scipy_dnrm2_64_(I(n), P(x), I(1)) == R.eigr_nrm2(n, P(x), 1)
        a, b = (float(v) for v in rng.standard_normal(2) * 10.0 ** rng.integers(-3, 4, 2))
        assert OB.scipy_dlapy2_64_(D(a), D(b)) == R.eigr_lapy2(a, b)
    for t in range(T):      # DLANV2
        v = rng.standard_normal(4) * 10.0 ** rng.integers(-2, 3, 4)
        if t % 7 == 0: v[2] = -v[1] * rng.uniform(0.5, 2)
        if t % 11 == 0: v[3] = v[0]
        if t % 13 == 0: v[2] = 0
        if t % 17 == 0: v[1] = 0
        a, b, c, d = [C.c_double(z) for z in v]
        o = [C.c_double() for _ in range(6)]
        OB.scipy_dlanv2_64_(C.byref(a), C.byref(b), C.byref(c), C.byref(d), *[C.byref(z) for z in o])
        mine = np.zeros(4)
        R.eigr_lanv2(*[float(z) for z in v], P(mine))
        assert np.array_equal(mine, [o[0].value, o[1].value, o[2].value, o[3].value]), v
    for t in range(T):      # DLARFG
        n = int(rng.integers(1, 17))
        alpha = float(rng.standard_normal() * 10.0 ** rng.integers(-3, 4))
        x = rng.standard_normal(max(n - 1, 1)) * 10.0 ** rng.integers(-3, 4)
        if t % 9 == 0: x[:] = 0
        a1, x1, tau1 = C.c_double(alpha), x.copy(), C.c_double()
        OB.scipy_dlarfg_64_(I(n), C.byref(a1), P(x1), I(1), C.byref(tau1))
        a2, x2, tau2 = np.array([alpha]), x.copy(), np.zeros(1)
        R.eigr_larfg(n, P(a2), P(x2), 1, P(tau2))
        assert a1.value == a2[0] and tau1.value == tau2[0] and np.array_equal(x1, x2)
    for side in ("L", "R"):     # DLARF = DGEMV + DGER
        for t in range(T):
            m, n, ld = int(rng.integers(1, 17)), int(rng.integers(1, 17)), 16
            Cm = np.zeros((ld, n), order="F"); Cm[:m, :] = rng.standard_normal((m, n))
            v = rng.standard_normal(m if side == "L" else n); v[0] = 1.0
            if t % 5 == 0: v[-1] = 0
            if t % 7 == 0: Cm[:, -1] = 0
            tau = float(rng.uniform(1, 2))
            C1, w1 = Cm.copy(order="F"), np.zeros(32)
            OB.scipy_dlarf_64_(C.c_char_p(side.encode()), I(m), I(n), P(v), I(1), D(tau), P(C1), I(ld), P(w1), C.c_size_t(1))
            C2, w2 = Cm.copy(order="F"), np.zeros(32)
            R.eigr_larf(1 if side == "L" else 0, m, n, P(v), tau, P(C2), ld, P(w2))
            assert np.array_equal(C1, C2), (side, m, n)
    for t in range(T):      # DGEBAL
        n = int(rng.integers(1, 17)); A = random_matrix(rng, n, t % 6)
        if t % 4 == 0:
            for _ in range(rng.integers(0, 3)):
                z = rng.integers(0, n); keep = A[z, z]
                if rng.random() < 0.5: A[z, :] = 0
                else: A[:, z] = 0
                A[z, z] = keep
        F1, sc1, ilo, ihi, info = np.asfortranarray(A.copy()), np.zeros(n), i64(), i64(), i64()
        OB.scipy_dgebal_64_(C.c_char_p(b"B"), I(n), P(F1), I(n), C.byref(ilo), C.byref(ihi), P(sc1), C.byref(info), C.c_size_t(1))
        F2, sc2, ilo2, ihi2 = np.asfortranarray(A.copy()), np.zeros(n), C.c_int(), C.c_int()
        R.eigr_gebal(n, P(F2), n, C.byref(ilo2), C.byref(ihi2), P(sc2))
        assert (ilo.value, ihi.value) == (ilo2.value, ihi2.value) and np.array_equal(F1, F2) and np.array_equal(sc1, sc2)
    for t in range(T):      # DGEHD2
        n = int(rng.integers(1, 17)); A = random_matrix(rng, n, t % 6)
        ilo = int(rng.integers(1, n + 1)); ihi = int(rng.integers(ilo, n + 1))
        if t % 3: ilo, ihi = 1, n
        F1, tau1, w, info = np.asfortranarray(A.copy()), np.zeros(n), np.zeros(n), i64()
        OB.scipy_dgehd2_64_(I(n), I(ilo), I(ihi), P(F1), I(n), P(tau1), P(w), C.byref(info))
        F2, tau2 = np.asfortranarray(A.copy()), np.zeros(n)
        R.eigr_gehd2(n, ilo, ihi, P(F2), n, P(tau2))
        assert np.array_equal(F1, F2) and np.array_equal(tau1[:max(n - 1, 0)], tau2[:max(n - 1, 0)])
    for t in range(T):      # DLAHQR
        n = int(rng.integers(1, 17)); A = np.triu(random_matrix(rng, n, t % 6), -1)
        F1, wr1, wi1, info, z = np.asfortranarray(A.copy()), np.zeros(n), np.zeros(n), i64(), np.zeros(1)
        OB.scipy_dlahqr_64_(I(0), I(0), I(n), I(1), I(n), P(F1), I(n), P(wr1), P(wi1), I(1), I(n), P(z), I(1), C.byref(info))
        F2, wr2, wi2 = np.asfortranarray(A.copy()), np.zeros(n), np.zeros(n)
        rc = R.eigr_lahqr(n, 1, n, P(F2), n, P(wr2), P(wi2))
        assert info.value == rc and np.array_equal(wr1, wr2) and np.array_equal(wi1, wi2)


def stage_matrices(Q0, Pm, rounds, end):
    """The matrices the reference's loop tested at one recorded stage, and LAPACK's verdicts."""
    mats, want = [], []
    Q = Q0.copy()
    count = rounds + 1 if end == 0 else rounds
    with np.errstate(over="ignore", invalid="ignore"):
        for r in range(count):
            mats.append(Q.copy()); want.append(r < rounds)
            Q = Q + np.ldexp(Pm, r)       # Q += (regu F)'F with regu = 0.1 * 2^r: 2^r times the first update, exactly
    return np.array(mats), np.array(want, bool)


def lapack_verdicts(R, mats):
    cnt, n = mats.shape[0], mats.shape[1]
    wr, wi, rc = np.zeros((cnt, n)), np.zeros((cnt, n)), np.zeros(cnt, np.int32)
    R.eigr_eigvals_many(cnt, n, P(np.ascontiguousarray(mats)), P(wr), P(wi), rc.ctypes.data_as(C.POINTER(C.c_int)))
    assert (rc == 0).all()
    return ((wr < 0) | ((wr == 0) & (wi < 0))).any(axis=1)


@pytest.mark.parametrize("tag,min_decisions", [("car", 120000), ("ea", 9000), ("quat", 6000)])
def test_every_recorded_decision_of_the_reference(R, tag, min_decisions):
    g = load_golden("eig_decisions")
    Q0, Pm, rounds, end = g[tag + "_Q0"], g[tag + "_P"], g[tag + "_rounds"], g[tag + "_end"]
    total = 0
    for s in range(len(Q0)):
        mats, want = stage_matrices(Q0[s], Pm[s], int(rounds[s]), int(end[s]))
        if len(mats) == 0:
            continue
        fin = np.isfinite(mats).all(axis=(1, 2))      # (the reference's last matrix of an `end` = 1 stage is not finite)
        got = lapack_verdicts(R, mats[fin])
        assert np.array_equal(got, want[fin]), (tag, s, int(rounds[s]))
        total += int(fin.sum())
    assert total >= min_decisions, total


def test_score_of_the_ldl_rule_on_the_recorded_decisions(oracle):
    """The cheaper rule (unpivoted LDL' with a 1e-13 pivot tolerance) against the same pairs: right where the smallest
    eigenvalue is resolved (every solve whose |Q| stays below 1e5 ends like the reference's), wrong in a few per cent of the decisions taken in rounding noise and in most stages
    that needed many rounds -- the reason the checker restates LAPACK instead."""
    g = load_golden("eig_decisions")
    Q0, Pm, rounds, end = g["car_Q0"], g["car_P"], g["car_rounds"], g["car_end"]
    by_decade = {}
    stages_right = stages = 0
    for s in range(0, len(Q0), 4):
        mats, want = stage_matrices(Q0[s], Pm[s], int(rounds[s]), int(end[s]))
        fin = np.isfinite(mats).all(axis=(1, 2)) if len(mats) else np.zeros(0, bool)
        if not fin.any():
            continue
        got = np.array([oracle.has_negative_eig(m, rule="ldl") == 1 for m in mats[fin]])
        ok = got == want[fin]
        stages += 1; stages_right += bool(ok.all())
        dec = np.floor(np.log10(np.maximum(np.abs(mats[fin]).max(axis=(1, 2)), 1e-300))).astype(int)
        for d, a in zip(dec, ok):
            e = by_decade.setdefault(int(d) // 4 * 4, [0, 0]); e[0] += 1; e[1] += int(a)
    table = {f"1e{d}..1e{d + 3}": f"{e[1] / e[0]:.4f} of {e[0]}" for d, e in sorted(by_decade.items())}
    print("LDL' rule, agreement with LAPACK by max|Q|:", table, "whole stages right:", stages_right, "of", stages)
    n = sum(e[0] for e in by_decade.values()); right = sum(e[1] for e in by_decade.values())
    assert by_decade[0][1] / by_decade[0][0] > 0.97          # max|Q| below 1e4: nearly always
    assert 0.9 < right / n < 0.995                           # overall: a few per cent off
    assert stages_right / stages < 0.8                       # a stage's round count is often not the reference's


def test_round_search_of_the_16_lane_kernel_on_recorded_stages(oracle):
    """The kernels of the 12-state plants try rounds 0..16 one after the other like the reference, then double the
    round number until the test stops failing and bisect back (csrc/team_backward.hpp).  On the stages the reference
    recorded with more than 16 rounds (quadrotors far from hover, up to 1027 rounds: Q runs into 1e308) this measures
    what that search finds against the one-by-one scan, under the kernels' own LDL' rule and under LAPACK's, and the
    scan against the reference's round count.  Those stages are outside per-trajectory parity whatever the search:
    under the LDL' rule not even the one-by-one scan ends where the reference ends."""
    g = load_golden("eig_decisions")
    for tag, min_same_ldl in (("ea", 0.7), ("quat", 0.9)):
        Q0, Pm, rounds, end = g[tag + "_Q0"], g[tag + "_P"], g[tag + "_rounds"], g[tag + "_end"]
        idx = np.where((rounds > 16) & (end == 0))[0][::3]
        assert len(idx) >= 30
        res = {}
        for rule in ("ldl", "lapack"):
            same = scan_is_reference = 0
            for i in idx:
                cache = {}

                def neg(r):
                    if r not in cache:
                        Q = Q0[i].copy()
                        with np.errstate(over="ignore", invalid="ignore"):
                            for k in range(r):
                                Q = Q + np.ldexp(Pm[i], k)
                        cache[r] = oracle.has_negative_eig(Q, rule=rule) == 1
                    return cache[r]
                scan = 0
                while neg(scan) and scan < 1300:
                    scan += 1
                lo, hi = -1, -1
                while hi < 0 or hi - lo > 1:
                    rt = (lo + 1 if lo < 16 else min(2 * lo, 1300)) if hi < 0 else (lo + hi) // 2
                    if neg(rt) and rt < 1300:
                        lo = rt
                    else:
                        hi = rt
                same += hi == scan
                scan_is_reference += scan == rounds[i]
            res[rule] = (same / len(idx), scan_is_reference / len(idx))
        print(tag, "search == scan, scan == reference:", res)
        assert res["ldl"][0] >= min_same_ldl           # the search finds the scan's round for most stages under its own rule
        assert res["lapack"][1] == 1.0                 # LAPACK's rule, scanned one by one, is the reference
        assert res["ldl"][1] <= 0.1                    # the LDL' rule is not, search or scan
