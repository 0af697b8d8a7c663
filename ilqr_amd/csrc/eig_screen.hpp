// A cheap, certified answer to `np.any(np.linalg.eigvals(Q) < 0)` (control/iLQRController.py:110) for the matrices
// on which the answer does not hang on LAPACK's rounding, so that the LAPACK path of eig_device.hpp only runs
// where it has to.
//
// What is certified.  LAPACK's DGEEV returns the exact eigenvalues of Q + E with |E|_F <= c eps |Q|_F as long as its
// balancing step (DGEBAL) only permutes and does not scale; Householder reduction and the shifted QR sweeps are
// orthogonal similarities.  Write Q = S + A, S symmetric, A antisymmetric (the reference never symmetrises, so A is
// rounding noise in well-conditioned solves and more in others).
//   * none:      every eigenvalue mu of Q + E has Re mu >= lambda_min(S + sym E) >= lambda_min(S) - |E|.  For S = L D L'
//                positive definite, lambda_min(S) >= 1 / trace(S^-1).  So 1 / trace(S^-1) > TOL says "no eigenvalue
//                compares below zero".
//   * negative:  S is Hermitian and Q + E = S + (A + E): the real parts of the eigenvalues of a Hermitian matrix plus
//                ANY perturbation F, in ascending order, stay within sqrt(2) |F|_F of the eigenvalues of S (Kahan
//                1975, spectra of nearly Hermitian matrices).  A vector x with Rayleigh quotient rho = x'Sx / x'x
//                bounds lambda_min(S) <= rho, so rho + KAHAN (|A|_F + TOL) < 0, KAHAN = 2 > sqrt(2), says "some
//                eigenvalue has a negative real part".  x is taken from the first negative pivot j of the
//                factorisation: L'x = e_j.
//   * anything else is IN DOUBT and goes to the LAPACK path: non-finite or extreme entries (DGEEV rescales those),
//     a row whose norm differs from its column's by more than a factor sqrt(3) (DGEBAL would scale), a zero pattern
//     that is not symmetric (DGEBAL's permutation could cascade), pivots lost in rounding.
// Rows and columns that are exactly zero off the diagonal in both directions are isolated by DGEBAL's permutation
// and their eigenvalue is the diagonal entry itself: its sign is read directly.
//
// TOL = SCREEN_C * eps * |Q|_F with SCREEN_C two orders of magnitude above the backward error constants of the
// routines involved; tests/test_eig_screen.py scores the verdicts against NumPy's on every recorded decision of the
// reference's solves and against the restated LAPACK path on the checker's solves.
//
// Plain C++ (no HIP builtins): the same text is compiled for the GPU and, by the tests, for the host.
#pragma once
#include <float.h>
#include <math.h>

#if defined(__HIPCC__)
#define ILQR_SCREEN_FN __host__ __device__ __forceinline__
#else
#define ILQR_SCREEN_FN inline
#endif

namespace ilqr {
namespace eig {

constexpr int SCREEN_NONE = 0, SCREEN_NEGATIVE = 1, SCREEN_DOUBT = 2;
constexpr double SCREEN_C = 4096.0;          // multiples of eps |Q|_F granted to LAPACK's backward error and to this file's own rounding
constexpr double SCREEN_EPS = 2.220446049250313e-16;

// Q given by an accessor q(i, j), 0-based, any i, j (the full matrix as LAPACK sees it).  W is a compile-time order:
// every loop unrolls and the work arrays live in registers.
template <int W, class Acc>
ILQR_SCREEN_FN int screen(const Acc& q)
{
    constexpr int NL = W * (W + 1) / 2;
    auto tri = [](int i, int j) { return i * (i + 1) / 2 + j; };
    double S[NL];                 // symmetric part, lower triangle by rows
    double r2[W], c2[W];          // squared row / column norms (with the diagonal, as DGEBAL takes them)
    bool rz[W], cz[W];            // row / column is exactly zero off the diagonal
    double a2 = 0.0, q2 = 0.0;
#pragma unroll
    for (int i = 0; i < W; ++i) { r2[i] = 0.0; c2[i] = 0.0; rz[i] = true; cz[i] = true; }
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const double dii = q(i, i);
        S[tri(i, i)] = dii;
        r2[i] = fma(dii, dii, r2[i]);
        c2[i] = fma(dii, dii, c2[i]);
#pragma unroll
        for (int j = 0; j < i; ++j) {
            const double lo = q(i, j), up = q(j, i);
            S[tri(i, j)] = 0.5 * (lo + up);
            const double df = lo - up;
            a2 = fma(df, df, a2);
            r2[i] = fma(lo, lo, r2[i]); c2[j] = fma(lo, lo, c2[j]);
            r2[j] = fma(up, up, r2[j]); c2[i] = fma(up, up, c2[i]);
            rz[i] = rz[i] && (lo == 0.0); cz[j] = cz[j] && (lo == 0.0);
            rz[j] = rz[j] && (up == 0.0); cz[i] = cz[i] && (up == 0.0);
        }
    }
#pragma unroll
    for (int i = 0; i < W; ++i) q2 += r2[i];
    // (q2 is NaN or infinite when an entry is, or when squares overflow: comparisons below are then false -> doubt)
    bool sure = (q2 > 1e-200) && (q2 < 1e200);
    const double qF = sqrt(q2);
    const double aF = sqrt(0.5 * a2);                  // |A|_F: (lo - up)/2 twice per pair
    const double tol = SCREEN_C * SCREEN_EPS * qF;
    bool iso_negative = false;
#pragma unroll
    for (int i = 0; i < W; ++i) {
        const bool iso = rz[i] && cz[i];
        sure = sure && (iso || (!rz[i] && !cz[i]));                         // symmetric zero pattern
        sure = sure && (iso || (r2[i] <= 3.0 * c2[i] && c2[i] <= 3.0 * r2[i]));   // DGEBAL leaves the scaling alone
        iso_negative = iso_negative || (iso && S[tri(i, i)] < 0.0);
        if (iso) S[tri(i, i)] = qF;        // taken out of the rest: a positive pivot that couples to nothing
    }
    // ---- L D L' of S, unpivoted; L (unit lower) overwrites S below the diagonal, D its diagonal
    int jneg = W;                 // first negative pivot
    bool lost = false;            // a pivot at or below zero before... (only the first matters)
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const double d = S[tri(j, j)];
        if (jneg == W && !(d > 0.0)) { jneg = j; lost = !(d < 0.0); }
        const double dinv = 1.0 / d;
#pragma unroll
        for (int i = j + 1; i < W; ++i) {
            const double lij = S[tri(i, j)] * dinv;
#pragma unroll
            for (int k = j + 1; k <= i; ++k) S[tri(i, k)] = fma(-lij, S[tri(k, j)], S[tri(i, k)]);   // column j still unscaled
        }
#pragma unroll
        for (int i = j + 1; i < W; ++i) S[tri(i, j)] *= dinv;
    }
    sure = sure && !lost;
    // ---- M = L^-1 (unit lower), row by row: M(i, j) = -sum_{j <= k < i} L(i, k) M(k, j)
    double M[NL];
#pragma unroll
    for (int i = 0; i < W; ++i) {
        M[tri(i, i)] = 1.0;
#pragma unroll
        for (int j = 0; j < i; ++j) {
            double s = S[tri(i, j)];                     // k = j term: L(i, j) * M(j, j)
#pragma unroll
            for (int k = j + 1; k < i; ++k) s = fma(S[tri(i, k)], M[tri(k, j)], s);
            M[tri(i, j)] = -s;
        }
    }
    int verdict;
    if (jneg == W) {
        // positive definite as factorised: 1 / trace(S^-1), trace = sum_k (sum_i M(k, i)^2) / d_k
        double tr = 0.0;
#pragma unroll
        for (int k = 0; k < W; ++k) {
            double s = 0.0;
#pragma unroll
            for (int i = 0; i <= k; ++i) s = fma(M[tri(k, i)], M[tri(k, i)], s);
            tr += s / S[tri(k, k)];
        }
        const bool none = (1.0 > 4.0 * tol * tr) && !iso_negative;       // 1 / tr > 4 tol (false when tr is NaN)
        verdict = iso_negative ? SCREEN_NEGATIVE : (none ? SCREEN_NONE : SCREEN_DOUBT);
    } else {
        // x = row jneg of M (zero beyond jneg); rho = x'Sx / x'x with S taken from q again
        double x[W];
#pragma unroll
        for (int i = 0; i < W; ++i) {
            double v = 0.0;
#pragma unroll
            for (int k = i; k < W; ++k) v = (jneg == k) ? M[tri(k, i)] : v;
            x[i] = v;
        }
        double num = 0.0, den = 0.0;
#pragma unroll
        for (int i = 0; i < W; ++i) {
            double y = 0.0;
#pragma unroll
            for (int j = 0; j < W; ++j) y = fma(0.5 * (q(i, j) + q(j, i)), x[j], y);
            const bool iso = rz[i] && cz[i];
            num = fma(iso ? 0.0 : x[i], y, num);     // (an isolated index has x = 0 unless it is jneg itself, which cannot be: its pivot is qF)
            den = fma(x[i], x[i], den);
        }
        constexpr double KAHAN = 2.0;
        const bool negative = (num < -(KAHAN * (aF + tol) + tol) * den);
        verdict = (negative || iso_negative) ? SCREEN_NEGATIVE : SCREEN_DOUBT;
    }
    return sure ? verdict : SCREEN_DOUBT;
}

}  // namespace eig
}  // namespace ilqr
