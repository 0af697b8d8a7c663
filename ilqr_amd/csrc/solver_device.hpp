// Per-trajectory iLQR building blocks for gfx950, one lane per trajectory.
//
// Each function restates one piece of control/iLQRController.py for a single
// trajectory held by one lane; trajectory-sized data (x, u, d, K) lives in HBM
// as per-stage records (see Rec below), the caller's arrays in a strided
// structure-of-arrays layout, and the small per-stage matrices live in the
// lane's registers / private memory.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include "eig_device.hpp"
#include "eig_screen.hpp"
#include "models.hpp"

namespace ilqr {

// Strided view of one trajectory's slice of a structure-of-arrays buffer:
// element i of the trajectory is p[i * s].
struct Lane {
    double* p;
    long s;
    __device__ __forceinline__ double ld(int i) const { return p[(long)i * s]; }
    __device__ __forceinline__ void st(int i, double v) const { p[(long)i * s] = v; }
};
struct CLane {
    const double* p;
    long s;
    __device__ __forceinline__ double ld(int i) const { return p[(long)i * s]; }
};

template <int C, class L>
__device__ __forceinline__ void load_vec(const L& a, int row, double* v)
{
#pragma unroll
    for (int c = 0; c < C; ++c) v[c] = a.ld(row * C + c);
}
template <int C>
__device__ __forceinline__ void store_vec(const Lane& a, int row, const double* v)
{
#pragma unroll
    for (int c = 0; c < C; ++c) a.st(row * C + c, v[c]);
}

template <int CNT>
__device__ __forceinline__ void load_rec(const double* p, double* v);      // (records: see Rec below)

constexpr double PIVOT_TOL = 1e-13;

// Ordering point for data that the lanes of ONE wavefront exchange through LDS.
// The solver's workgroups are single wavefronts, whose LDS operations execute
// in program order, so nothing has to be waited for: this only stops the
// compiler from moving LDS accesses across it.  (__syncthreads() would also
// drain every outstanding global load and store of the wavefront, which costs
// a memory round trip per call.)
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// np.linalg.inv of an m x m matrix (m <= 4) by Gauss-Jordan with row
// pivoting.  Returns false when a pivot is exactly zero (NumPy: LinAlgError).
template <int m>
__device__ inline bool inv_small(const double* A, double* Ainv)
{
    if constexpr (m == 1) {
        if (A[0] == 0.0) return false;
        Ainv[0] = 1.0 / A[0];
        return true;
    } else {
        double T[m][2 * m];
#pragma unroll
        for (int i = 0; i < m; ++i)
#pragma unroll
            for (int j = 0; j < m; ++j) {
                T[i][j] = A[i * m + j];
                T[i][m + j] = (i == j) ? 1.0 : 0.0;
            }
        bool ok = true;
#pragma unroll
        for (int c = 0; c < m; ++c) {
            // bring the largest |entry| of column c (rows >= c) to row c
#pragma unroll
            for (int r = c + 1; r < m; ++r) {
                bool sw = fabs(T[r][c]) > fabs(T[c][c]);
#pragma unroll
                for (int j = 0; j < 2 * m; ++j) {
                    double a = T[c][j], b = T[r][j];
                    T[c][j] = sw ? b : a;
                    T[r][j] = sw ? a : b;
                }
            }
            ok = ok && !(T[c][c] == 0.0);
            double pinv = 1.0 / T[c][c];
#pragma unroll
            for (int j = 0; j < 2 * m; ++j) T[c][j] *= pinv;
#pragma unroll
            for (int r = 0; r < m; ++r) {
                if (r == c) continue;
                double f = T[r][c];
#pragma unroll
                for (int j = 0; j < 2 * m; ++j) T[r][j] -= f * T[c][j];
            }
        }
#pragma unroll
        for (int i = 0; i < m; ++i)
#pragma unroll
            for (int j = 0; j < m; ++j) Ainv[i * m + j] = T[i][m + j];
        return ok;
    }
}

// ------------------------------------------------------------------
// Storage of the stage's matrices.  The reference never symmetrises V_xx or the
// Q-function Hessian: both triangles are carried and differ by rounding, and in
// ill-conditioned solves that rounding-level asymmetry decides how many
// regularisation rounds a stage takes.  So V_xx is kept as the full n x n
// matrix (row-major), and the Hessian [[Qxx, Qux'], [Qux, Quu]] as
//   QL  its lower triangle packed by rows: entry (i, j), i >= j, at i (i + 1) / 2 + j, in the order
//       [x (n); u (m)] -- this is what the PSD test factorises and what holds Qux;
//   QU  the entries above the diagonal of Qxx (row by row), then those of Quu.
// ------------------------------------------------------------------
__device__ __forceinline__ constexpr int tri(int i, int j) { return i * (i + 1) / 2 + j; }
// position in QU of entry (i, j), i < j, of a k x k block whose upper entries start at `base`
__device__ __forceinline__ constexpr int upp(int k, int i, int j, int base = 0)
{
    return base + i * k - i * (i + 1) / 2 + (j - i - 1);
}

#define ILQR_PRAGMA(x) _Pragma(#x)
// small plants: every matrix in registers, which needs compile-time indices
// everywhere (UF = full unroll)
#define ILQR_UNROLL ILQR_PRAGMA(unroll UF)

template <int n, int m>
struct QPack {
    static constexpr int w = n + m;
    static constexpr int NL = w * (w + 1) / 2;                      // lower triangle of the full Hessian
    static constexpr int NUP = n * (n - 1) / 2 + m * (m - 1) / 2;   // upper entries of Qxx and Quu
    static constexpr int UUB = n * (n - 1) / 2;                     // where Quu's upper entries start in QU
    // entry (i, j) of Qxx, any i, j
    __device__ static double qxx(const double* QL, const double* QU, int i, int j)
    {
        return i >= j ? QL[tri(i, j)] : QU[upp(n, i, j)];
    }
    __device__ static double quu(const double* QL, const double* QU, int a, int b)
    {
        return a >= b ? QL[tri(n + a, n + b)] : QU[upp(m, a, b, UUB)];
    }
};

// Q-function terms of one stage from the plant's dense stage() derivatives
// (control/models/model.py:45-53), in the reference's order of operations:
// Q_xx = l_xx + (f_x' V_xx) f_x, Q_ux = l_ux + (f_u' V_xx) f_x, Q_uu = l_uu + (f_u' V_xx) f_u.
// aux = [fx (n*n), fu (n*m)].
// P = (0.1 [fx fu])'[fx fu]: lower triangle PL (packed by rows) and the entries PU above the diagonal of the xx and uu
// blocks, each entry (i, j) = sum over t ascending of fma(0.1 F[t][i], F[t][j], .)   (iLQRController.py:111-113)
template <int n, int m>
__device__ __forceinline__ void exact_p(const double* fx, const double* fu, double* PL, double* PU)
{
    constexpr int w = n + m;
#pragma unroll
    for (int i = 0; i < w; ++i)
#pragma unroll
        for (int j = 0; j < w; ++j) {
            if (i < j && (i < n) != (j < n)) continue;      // upper right block: the mirror image of Q_ux, exactly
            double s = 0.0;
#pragma unroll
            for (int t = 0; t < n; ++t) {
                const double a = (i < n) ? fx[t * n + i] : fu[t * m + (i - n)];
                const double b = (j < n) ? fx[t * n + j] : fu[t * m + (j - n)];
                s = __builtin_fma(0.1 * a, b, s);
            }
            if (i >= j) PL[tri(i, j)] = s;
            else if (j < n) PU[upp(n, i, j)] = s;
            else PU[upp(m, i - n, j - n, QPack<n, m>::UUB)] = s;
        }
}

template <class M>
struct DenseTerms {
    static constexpr int n = M::NDX, m = M::NU, w = n + m;
    static constexpr int AUX = n * n + n * m;
    static constexpr int UF = 64;
    using QP = QPack<n, m>;
    __device__ static void backward_terms(const double* par, const double* x, const double* u, const double* Vx,
                                          const double* V, double* Qx, double* Qu, double* QL, double* QU, double* aux)
    {
        double lx[n], lu[m], lxx[n * n], lux[m * n], luu[m * m];
        double* const fx = aux;
        double* const fu = aux + n * n;
        M::stage(par, x, u, lx, lu, lxx, lux, luu, fx, fu);
        ILQR_UNROLL
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            ILQR_UNROLL
            for (int t = 0; t < n; ++t) s += fx[t * n + i] * Vx[t];
            Qx[i] = lx[i] + s;
        }
        ILQR_UNROLL
        for (int i = 0; i < m; ++i) {
            double s = 0.0;
            ILQR_UNROLL
            for (int t = 0; t < n; ++t) s += fu[t * m + i] * Vx[t];
            Qu[i] = lu[i] + s;
        }
        {
            double FtV[n * n];     // f_x' V_xx
            ILQR_UNROLL
            for (int i = 0; i < n; ++i)
                ILQR_UNROLL
                for (int j = 0; j < n; ++j) {
                    double s = 0.0;
                    ILQR_UNROLL
                    for (int t = 0; t < n; ++t) s += fx[t * n + i] * V[t * n + j];
                    FtV[i * n + j] = s;
                }
            ILQR_UNROLL
            for (int i = 0; i < n; ++i)
                ILQR_UNROLL
                for (int j = 0; j < n; ++j) {
                    double s = 0.0;
                    ILQR_UNROLL
                    for (int t = 0; t < n; ++t) s += FtV[i * n + t] * fx[t * n + j];
                    const double q = lxx[i * n + j] + s;
                    if (i >= j) QL[tri(i, j)] = q;
                    else QU[upp(n, i, j)] = q;
                }
        }
        {
            double GtV[m * n];     // f_u' V_xx
            ILQR_UNROLL
            for (int i = 0; i < m; ++i)
                ILQR_UNROLL
                for (int j = 0; j < n; ++j) {
                    double s = 0.0;
                    ILQR_UNROLL
                    for (int t = 0; t < n; ++t) s += fu[t * m + i] * V[t * n + j];
                    GtV[i * n + j] = s;
                }
            ILQR_UNROLL
            for (int i = 0; i < m; ++i) {
                ILQR_UNROLL
                for (int j = 0; j < n; ++j) {
                    double s = 0.0;
                    ILQR_UNROLL
                    for (int t = 0; t < n; ++t) s += GtV[i * n + t] * fx[t * n + j];
                    QL[tri(n + i, j)] = lux[i * n + j] + s;
                }
                ILQR_UNROLL
                for (int j = 0; j < m; ++j) {
                    double s = 0.0;
                    ILQR_UNROLL
                    for (int t = 0; t < n; ++t) s += GtV[i * n + t] * fu[t * m + j];
                    const double q = luu[i * m + j] + s;
                    if (i >= j) QL[tri(n + i, n + j)] = q;
                    else QU[upp(m, i, j, QP::UUB)] = q;
                }
            }
        }
    }
    // P = (0.1 [fx fu])'[fx fu], lower triangle, in the reference's operation order (iLQRController.py:111-113)
    __device__ static void regu_p(const double* aux, double* PL)
    {
        const double* const fx = aux;
        const double* const fu = aux + n * n;
        ILQR_UNROLL
        for (int i = 0; i < w; ++i)
            ILQR_UNROLL
            for (int j = 0; j <= i; ++j) {
                double s = 0.0;
                ILQR_UNROLL
                for (int t = 0; t < n; ++t) {
                    const double a = (i < n) ? fx[t * n + i] : fu[t * m + (i - n)];
                    const double b = (j < n) ? fx[t * n + j] : fu[t * m + (j - n)];
                    s += (0.1 * a) * b;
                }
                PL[tri(i, j)] = s;
            }
    }
    // The same with the entries above the diagonal of the xx and uu blocks as the reference has them: its
    // (0.1 f)'f is a matrix product of two different arrays, so entry (i, j) sums (0.1 f[t][i]) f[t][j] and its
    // mirror image (0.1 f[t][j]) f[t][i]: equal up to rounding, and in stages that regularisation cannot rescue
    // (Q(r) dominated by 2^r P) that rounding is what LAPACK's verdict is made of.  Ascending t with fused
    // multiply-adds from zero is NumPy's `@` bit for bit.
    __device__ static void regu_p_exact(const double* aux, double* PL, double* PU)
    {
        exact_p<n, m>(aux, aux + n * n, PL, PU);
    }
    __device__ static void final_full(const double* par, const double* x, double* Vx, double* V)
    {
        M::final(par, x, Vx, V);
    }
};

// Kinematic car: the same terms with the sparsity of f_x, f_u, l_xx folded in
// offline (generated by tools/gen_models.py).  aux = the six state-dependent entries of f_x.
struct CarSparseTerms {
    static constexpr int AUX = 6;
    __device__ static void backward_terms(const double* par, const double* x, const double* u, const double* Vx,
                                          const double* V, double* Qx, double* Qu, double* QL, double* QU, double* aux)
    {
        car_gen_backward_terms(par, x, u, Vx, V, Qx, Qu, QL, QU, aux);
    }
    __device__ static void regu_p(const double* aux, double* PL) { car_gen_regu_p(aux, PL); }
    // (see DenseTerms::regu_p_exact)  f_x = I + the six entries in aux, f_u = dt at (3, 0) and (4, 1)  (car.py:76-79)
    __device__ static void regu_p_exact(const double* aux, double* PL, double* PU)
    {
        double fx[25], fu[10];
#pragma unroll
        for (int e = 0; e < 25; ++e) fx[e] = (e / 5 == e % 5) ? 1.0 : 0.0;
#pragma unroll
        for (int e = 0; e < 10; ++e) fu[e] = 0.0;
        fx[0 * 5 + 2] = aux[0]; fx[0 * 5 + 3] = aux[1]; fx[1 * 5 + 2] = aux[2]; fx[1 * 5 + 3] = aux[3];
        fx[2 * 5 + 3] = aux[4]; fx[2 * 5 + 4] = aux[5];
        fu[3 * 2 + 0] = 0.1; fu[4 * 2 + 1] = 0.1;
        exact_p<5, 2>(fx, fu, PL, PU);
    }
    __device__ static void final_full(const double* par, const double* x, double* Vx, double* V)
    {
        car_gen_final(par, x, Vx, V);
    }
};
template <class M> struct TermsOf { using type = DenseTerms<M>; };
template <> struct TermsOf<CarModel> { using type = CarSparseTerms; };
#if defined(ILQR_USER_MODEL) && defined(USER_HAS_QTERMS)
// A user-defined plant with its own Q_terms (control/models/Quadrotor_Quat.py:267-275 is the reference's example):
// the plant's stage(), then the plant's Q_terms() on what it returned -- the two calls of iLQRController.py:103-104.
// aux = [fx (n*n), fu (n*m)] on the error state, which is what the regularisation update is made of (:111-113).
struct UserOwnTerms {
    using M = UserModel;
    static constexpr int n = M::NDX, m = M::NU, nx = M::NX, w = n + m;
    static constexpr int AUX = n * n + n * m;
    using QP = QPack<n, m>;
    __device__ static void backward_terms(const double* par, const double* x, const double* u, const double* Vx,
                                          const double* V, double* Qx, double* Qu, double* QL, double* QU, double* aux)
    {
        double lx[nx], lu[m], lxx[nx * nx], lux[m * nx], luu[m * m];
        double Qxx[n * n], Qux[m * n], Quu[m * m];
        M::stage_raw(par, x, u, lx, lu, lxx, lux, luu, aux, aux + n * n);
        M::q_terms(par, x, lx, lu, lxx, lux, luu, aux, aux + n * n, Vx, V, Qx, Qu, Qxx, Qux, Quu);
#pragma unroll
        for (int i = 0; i < n; ++i)
#pragma unroll
            for (int j = 0; j < n; ++j) {
                if (i >= j) QL[tri(i, j)] = Qxx[i * n + j];
                else QU[upp(n, i, j)] = Qxx[i * n + j];
            }
#pragma unroll
        for (int i = 0; i < m; ++i) {
#pragma unroll
            for (int j = 0; j < n; ++j) QL[tri(n + i, j)] = Qux[i * n + j];
#pragma unroll
            for (int j = 0; j < m; ++j) {
                if (i >= j) QL[tri(n + i, n + j)] = Quu[i * m + j];
                else QU[upp(m, i, j, QP::UUB)] = Quu[i * m + j];
            }
        }
    }
    __device__ static void regu_p(const double* aux, double* PL) { DenseTerms<M>::regu_p(aux, PL); }
    __device__ static void regu_p_exact(const double* aux, double* PL, double* PU) { DenseTerms<M>::regu_p_exact(aux, PL, PU); }
    __device__ static void final_full(const double* par, const double* x, double* Vx, double* V) { M::final(par, x, Vx, V); }
};
template <> struct TermsOf<UserModel> { using type = UserOwnTerms; };
#endif

// ------------------------------------------------------------------
// `np.any(np.linalg.eigvals(Q) < 0)` of iLQRController.py:110, decided by an
// unpivoted right-looking LDL' of the packed lower triangle L (destroyed):
//   pivot < 0 -> negative eigenvalue; zero pivot with a non-zero column below
//   -> indefinite; zero pivot with a zero column -> zero eigenvalue, skip.
// A pivot that has cancelled to within PIVOT_TOL of the terms it is the
// difference of (acc_j = |Q_jj| + sum_k |l_jk^2 d_k|) is rounding noise and is
// treated as zero (without this, a Q dominated by regu*[fx fu]'[fx fu], which is
// rank deficient, keeps the same noise sign through every further doubling of regu).
// Returns 1 (negative eigenvalue), 0 (none), -1 (not finite: NumPy raises).
// ------------------------------------------------------------------
template <int w, int UF>
__device__ inline int has_negative_eig_packed(double* L, double& maxq)
{
    constexpr int NL = w * (w + 1) / 2;
    // largest |entry| and the finite check (0 * entry summed up stays 0 unless an entry is an infinity or a
    // NaN), each as four independent chains
    double mqs[4] = {maxq, 0.0, 0.0, 0.0}, poisons[4] = {0.0, 0.0, 0.0, 0.0};
    ILQR_UNROLL
    for (int e = 0; e < NL; ++e) {
        mqs[e & 3] = fmax(mqs[e & 3], fabs(L[e]));          // a NaN entry never wins, as in a chain of `a > mq` comparisons
        poisons[e & 3] = __builtin_fma(0.0, L[e], poisons[e & 3]);
    }
    maxq = fmax(fmax(mqs[0], mqs[1]), fmax(mqs[2], mqs[3]));
    const double poison = (poisons[0] + poisons[1]) + (poisons[2] + poisons[3]);
    double acc[w];      // |Q_jj| + sum_k |l_jk^2 d_k|: the size of the terms pivot j is the difference of
    ILQR_UNROLL
    for (int i = 0; i < w; ++i) acc[i] = fabs(L[tri(i, i)]);
    // Straight-line code, no early exit: once the verdict is in the remaining steps still run but
    // can no longer change it; a skipped (zero) pivot eliminates nothing (pinv = 0).
    bool negative = false;
    ILQR_UNROLL
    for (int j = 0; j < w; ++j) {
        const double p = L[tri(j, j)];
        const double thr = PIVOT_TOL * acc[j];
        const bool neg = p < -thr;
        const bool zero = !neg && (p <= thr);
        // a zero pivot over a column that is not zero means indefinite; zero pivots are rare, so the column is
        // only looked at when some lane of the wavefront has one
        bool nz = false;
        if (__ballot(zero)) {
            ILQR_UNROLL
            for (int i = j + 1; i < w; ++i) {
                const double c = L[tri(i, j)];
                nz = nz || (c * c > 4.0 * thr * fabs(L[tri(i, i)]));
            }
        }
        negative = negative || neg || (zero && nz);
        // the factorisation only decides signs: a reciprocal refined to ~1 ulp is enough, and several times
        // cheaper than an IEEE division
        double pinv = __builtin_amdgcn_rcp(p);
        pinv = __builtin_fma(__builtin_fma(-p, pinv, 1.0), pinv, pinv);
        pinv = __builtin_fma(__builtin_fma(-p, pinv, 1.0), pinv, pinv);
        pinv = (zero || negative) ? 0.0 : pinv;
        ILQR_UNROLL
        for (int i = j + 1; i < w; ++i) {
            const double lij = L[tri(i, j)] * pinv;
            ILQR_UNROLL
            for (int k = j + 1; k <= i; ++k) L[tri(i, k)] -= lij * L[tri(k, j)];
            acc[i] += fabs(lij * L[tri(i, j)]);
        }
    }
    return (poison != poison) ? -1 : (negative ? 1 : 0);
}

// ------------------------------------------------------------------
// Regularisation loop of iLQRController.py:106-115,
//     regu = 0.1; while any(eigvals(Q) < 0): Q += (regu [fx fu])'[fx fu]; regu *= 2
// Because regu is 0.1 times a power of two, the update of round r is exactly
// 2^r * P with P = (0.1 [fx fu])'[fx fu] evaluated once, so "after r rounds" is
//     Q(r) = Q + (2^r - 1) P,
// the reference's running sum up to the rounding of its r additions, and a
// matrix any lane can build with one fused multiply-add per entry.
// The result of the loop is the first r = 0, 1, 2, ... whose Q(r) has no negative eigenvalue.
//
// Execution.  Typically a few lanes of a wavefront need 1-8 rounds at a stage
// and, in ill-conditioned solves, one needs ~50.  Running the loop lane by lane
// makes 64 trajectories wait for the slowest, so the wavefront scans instead:
//   group scan: up to 8 lanes that need regularisation publish Q(0) and P in
//     LDS; lane 8g+c builds and tests Q(c+1) of trajectory g; the first passing
//     c of each group is that trajectory's answer (plants with a small Hessian);
//   stuck scan: a trajectory still failing after REGU_BASE rounds publishes
//     Q(REGU_BASE) and P; lane l tests Q(REGU_BASE + 1 + l), then +64, ...
// ------------------------------------------------------------------
#ifndef ILQR_SCAN_LIMIT
#define ILQR_SCAN_LIMIT 1200     // (a smaller value cuts long scans short: timing experiments only)
#endif
constexpr int REGU_BASE = 8;      // rounds covered by the group scan (or lane by lane for the big plants)
constexpr int REGU_GROUPS = 8;    // trajectories per group scan; REGU_GROUPS * REGU_BASE == 64

// 2^r - 1, the factor of P after r rounds (exact up to r = 53, then 2^r)
__device__ __forceinline__ double regu_coef(int r) { return ldexp(1.0, r) - 1.0; }

// Per-wavefront LDS of one backward stage, in doubles:
//   P of up to REGU_GROUPS published trajectories (lower triangles), the lane numbers of those trajectories,
//   and per lane the "park": Q(0) (lower triangle) and the plant's aux values, entry e of lane l at
//   park[e * 64 + l].  Q(0) waits there while the lane factorises its registers in place, the scans read a
//   published trajectory's Q(0) straight from its lane's park, and a trajectory that needed r rounds
//   leaves Q(r) there.
template <int n, int m>
constexpr int scan_lds_doubles() { return REGU_GROUPS * QPack<n, m>::NL + REGU_GROUPS / 2; }
template <int n, int m, int AUX, int PS = 64>
constexpr int stage_lds_doubles() { return scan_lds_doubles<n, m>() + PS * (QPack<n, m>::NL + AUX); }
constexpr int WIDE_TRAJ = 4;      // trajectories per wavefront, at most, in the launch for long regularisation loops
constexpr int WIDE_PARK = 8;      // park places of that kernel

// With ILQR_FLAG_LAPACK_EIGVALS every test is made the way NumPy makes it (eig_device.hpp) on the full matrix, both
// triangles.  That needs, per wavefront, one matrix and its work space per lane, and the entries above the diagonal
// of Q(0) of every published trajectory (they live in the source lane's registers otherwise).
template <int n, int m>
constexpr int exact_lds_doubles()
{
    return ((n + m) * (n + m) + 2 * (n + m)) * 64 + 2 * ((REGU_GROUPS * QPack<n, m>::NUP + 1) / 2 * 2);
}
// the lane's matrix for eig::Lane from the packed lower triangle L and the upper entries U of Qxx and Quu
// (the Qux block is one block: its mirror image fills the upper right corner, as _full_hessian does, :139-147)
template <int n, int m>
__device__ __forceinline__ void fill_full(const eig::Lane<n + m, 64>& E, const double* L, const double* U)
{
    using QP = QPack<n, m>;
    constexpr int w = n + m;
#pragma unroll
    for (int i = 0; i < w; ++i)
#pragma unroll
        for (int j = 0; j < w; ++j) {
            double v;
            if (i >= j) v = L[tri(i, j)];
            else if (j < n) v = U[upp(n, i, j)];
            else if (i >= n) v = U[upp(m, i - n, j - n, QP::UUB)];
            else v = L[tri(j, i)];
            E.A(i + 1, j + 1) = v;
        }
}
// the same matrix entry by entry, for eig::screen
template <int n, int m>
struct PackedFull {
    const double* L;
    const double* U;
    __device__ __forceinline__ double operator()(int i, int j) const
    {
        using QP = QPack<n, m>;
        if (i >= j) return L[tri(i, j)];
        if (j < n) return U[upp(n, i, j)];
        if (i >= n) return U[upp(m, i - n, j - n, QP::UUB)];
        return L[tri(j, i)];
    }
};
// `np.any(np.linalg.eigvals(Q) < 0)` the way NumPy decides it, for the lane's matrix (L, U): the certified screen
// first, LAPACK's path (eig_device.hpp) for the lanes it leaves in doubt.  `doubt` says which way this lane went.
// Called by all lanes together; the LAPACK path is skipped by the wavefront when no lane needs it.
#ifdef ILQR_DEBUG_COUNTERS
// build-time diagnostics (tools/debug_counters.py): [2 site] calls of exact_negative by a wavefront, [2 site + 1] those
// in which some lane took LAPACK's path; sites 0 round 0, 1 group scan, 2 stuck scan; [8] lanes in doubt, summed
static __device__ unsigned long long g_dbg[16];
// [0][b]: trajectory-passes of the launch for long regularisation loops by the trips (LAPACK tests of the wavefront) their
// loops took part in (bin b: 2^(b-1) < trips <= 2^b, bin 0: none); [1][b]: stages of those trajectories by the round their
// loop stopped at, same bins
static __device__ unsigned long long g_hist[2][16];
__device__ __forceinline__ int dbg_bin(int t) { int b = 0; while ((1 << b) < t + 1 && b < 15) ++b; return b; }
#define ILQR_DBG_SITE(k) , (k)
#else
#define ILQR_DBG_SITE(k)
#endif
template <int n, int m>
__device__ __forceinline__ int exact_negative(const eig::Lane<n + m, 64>& E, const double* L, const double* U, bool want,
                                              bool& doubt
#ifdef ILQR_DEBUG_COUNTERS
                                              , int site
#endif
)
{
    // (a lane whose verdict nobody will read -- no trajectory, or one that has dropped out -- never asks for LAPACK's path)
    int v = eig::screen<n + m>(PackedFull<n, m>{L, U});
    doubt = want && (v == eig::SCREEN_DOUBT);
    v = want ? v : eig::SCREEN_NONE;
#ifdef ILQR_DEBUG_COUNTERS
    {
        const unsigned long long db = __ballot(doubt);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&g_dbg[2 * site], 1ull);
            if (db) atomicAdd(&g_dbg[2 * site + 1], 1ull);
            atomicAdd(&g_dbg[8 + site], (unsigned long long)__popcll(db));
        }
    }
#endif
    if (doubt) {
        fill_full<n, m>(E, L, U);
        v = E.has_negative();
    }
    return v;
}
// The same for candidates of one regularisation loop in round order, lane by lane, in the lanes `share` (this lane's
// share of the wavefront: a mask of consecutive lanes; `have`: the share holds a loop): what is wanted is the first
// lane whose matrix has no negative eigenvalue (or is not finite), so LAPACK's path is only taken by the lanes in
// doubt BELOW the first lane of the share the screen is sure about.  Lanes above it return 1.
template <int n, int m>
__device__ __forceinline__ int exact_negative_ordered(const eig::Lane<n + m, 64>& E, const double* L, const double* U,
                                                      bool have, unsigned long long share
#ifdef ILQR_DEBUG_COUNTERS
                                                      , int site
#endif
)
{
    int v = eig::screen<n + m>(PackedFull<n, m>{L, U});
    const unsigned long long sure = __ballot(have && v == eig::SCREEN_NONE) & share;
    const int first = sure ? __ffsll((long long)sure) - 1 : 64;
    const bool doubt = have && (v == eig::SCREEN_DOUBT) && (int)(threadIdx.x & 63) < first;
#ifdef ILQR_DEBUG_COUNTERS
    {
        const unsigned long long db = __ballot(doubt);
        if ((threadIdx.x & 63) == 0) {
            atomicAdd(&g_dbg[2 * site], 1ull);
            if (db) atomicAdd(&g_dbg[2 * site + 1], 1ull);
            atomicAdd(&g_dbg[8 + site], (unsigned long long)__popcll(db));
        }
    }
#endif
    v = (!have || v == eig::SCREEN_DOUBT) ? 1 : v;
    if (doubt) {
        fill_full<n, m>(E, L, U);
        v = E.has_negative();
    }
    return v;
}
// largest |entry| of a packed lower triangle joined to mq (a NaN entry never wins)
template <int NL>
__device__ __forceinline__ double max_abs_packed(const double* L, double mq)
{
    double a[4] = {mq, 0.0, 0.0, 0.0};
#pragma unroll
    for (int e = 0; e < NL; ++e) a[e & 3] = fmax(a[e & 3], fabs(L[e]));
    return fmax(fmax(a[0], a[1]), fmax(a[2], a[3]));
}

#ifndef ILQR_BUILD_CHUNK
#define ILQR_BUILD_CHUNK 8
#endif
// ILQR_PHASE_MARKERS (tools/stage_budget.py): named comments in the assembly at the phase boundaries of a stage
#ifdef ILQR_PHASE_MARKERS
#define ILQR_MARK(name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; ILQR_MARK " #name); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define ILQR_MARK(name) ((void)0)
#endif
#ifdef ILQR_NO_PHASE_BARRIER
#define ILQR_PHASE_BARRIER() ((void)0)
#else
#define ILQR_PHASE_BARRIER() __builtin_amdgcn_sched_barrier(0)
#endif
// Q(r) = Q(0) + coef P of a published trajectory, built in registers in portions (all 2 NL LDS reads issued
// at once would need 2 NL more registers than the factorisation that follows)
template <int NL>
__device__ __forceinline__ void build_regularised(double* L, double coef, const double* P, const double* Qpark)
{
#pragma unroll
    for (int e = 0; e < NL; ++e) {
        L[e] = __builtin_fma(coef, P[e], Qpark[e * 64]);
        if (e % ILQR_BUILD_CHUNK == ILQR_BUILD_CHUNK - 1) ILQR_PHASE_BARRIER();
    }
}

// The reference's own accumulation (iLQRController.py:111-115): Q(r) = Q(r - 1) + 2^(r-1) P, one rounded addition per
// round.  v += 2^k0 p, 2^(k0+1) p, ..., `count` terms (ILQR_FLAG_LAPACK_EIGVALS; the default mode forms
// Q(0) + (2^r - 1) P with one fused multiply-add, which differs from this in the last bits).
__device__ __forceinline__ double running_sum(double v, double p, int k0, int count)
{
#pragma clang fp contract(off)
    for (int i = 0; i < count; ++i) v += ldexp(p, k0 + i);
    return v;
}

// ------------------------------------------------------------------
// One stage of the backward Riccati sweep: iLQRController.py:103-122 with
// Model.Q_terms (model.py:45-53), the regularisation loop (:106-115), _gains
// (:125-129), _V_terms (:131-134) and _expected_cost_reduction (:136-137).
// Vx and V (full V_xx) are updated in place; d_k and K_k are returned.
// xu = [x_k, u_k] is only needed for the Q-function terms; as soon as those are
// done the record at `next` (the stage the caller will pass next) is loaded into
// xu, so that it arrives while the rest of the stage runs and is older than the
// stage's stores of d_k, K_k: memory operations complete in the order they were
// issued, and a load issued after those stores would have the next stage wait for
// them to reach memory first.
// Must be called by all 64 lanes of the wavefront together (`alive` = this
// lane carries a trajectory); `lds` holds stage_lds_doubles() doubles.
// Returns false on the LinAlgError cases.
// ------------------------------------------------------------------
// EXACT with DEFER: a trajectory that is still failing after the group scan is not scanned further here: it is
// reported in `deferred` and drops out of this pass (the caller runs its pass again in a launch whose wavefronts hold few
// trajectories each, so that the long scans of different trajectories do not queue up in one wavefront).
// EXACT without DEFER is the stage of that second kind of launch: no round 0 in the lane and no group scan; the
// rounds 0, 1, 2, ... of every trajectory of the wavefront are tested 64 at a time, one trajectory after the other
// (a trajectory that got here needs ~50 rounds at many stages: one LAPACK test of 64 lanes instead of three or four
// one after the other).  Same matrices, same order, same counts as the other kernels.
template <class M, bool EXACT = false, bool DEFER = false>
__device__ __forceinline__ bool backward_stage(const double* par, double* xu, const double* next, double* Vx, double* V,
                                      double* dk, double* Kk, double& redu, int& regu_rounds, int& regu_stages,
                                      double& maxq, bool alive, double* lds, bool& dirty, bool& deferred
#ifdef ILQR_DEBUG_COUNTERS
                                      , int& dbg_trips
#endif
)
{
    const double* const xk = xu;
    const double* const uk = xu + M::NX;
    using TR = typename TermsOf<M>::type;
    using QP = QPack<M::NDX, M::NU>;
    constexpr int n = M::NDX, m = M::NU, w = n + m;
    constexpr int NL = QP::NL, NUP = QP::NUP;
    constexpr int UF = 64;
    double Qx[n], Qu[m], QL[NL], QU[NUP > 0 ? NUP : 1];
    const int lane = threadIdx.x & 63;
    double* const ldsP = lds;                                                   // P of the published trajectories
    int* const srcs = reinterpret_cast<int*>(lds + REGU_GROUPS * NL);           // their lanes
    // this lane's park, stride PS.  (WIDE: few lanes hold a trajectory; the others, which shadow one and store nothing
    // that is read, share the last place.)
    constexpr bool WIDE = EXACT && !DEFER;
    constexpr int PS = WIDE ? WIDE_PARK : 64;
    const int pslot = (lane < PS - 1) ? lane : PS - 1;
    double* const park = lds + scan_lds_doubles<n, m>() + pslot;
    // EXACT: the lane's matrix and work space for the LAPACK path, and the published trajectories' upper entries
    double* const exact0 = lds + stage_lds_doubles<n, m, TermsOf<M>::type::AUX, PS>();
    const eig::Lane<w, 64> E{exact0 + lane, exact0 + w * w * 64 + lane};
    double* const ldsQU = exact0 + (w * w + 2 * w) * 64;                                  // Q(0) above the diagonal
    double* const ldsPU = ldsQU + (REGU_GROUPS * (NUP > 0 ? NUP : 1) + 1) / 2 * 2;          // P above the diagonal
    double PUown[NUP > 0 ? NUP : 1];       // EXACT: this lane's own P above the diagonal, for the accepted round
    ILQR_MARK(terms);
    {
        double aux[TR::AUX];
        TR::backward_terms(par, xk, uk, Vx, V, Qx, Qu, QL, QU, aux);
        // The phases of a stage are kept apart for the instruction scheduler: interleaving them saves no time
        // (each is a long dependent chain of its own) and makes their register needs add up.
        ILQR_PHASE_BARRIER();
        ILQR_UNROLL
        for (int e = 0; e < NL; ++e) park[e * PS] = QL[e];
        ILQR_UNROLL
        for (int e = 0; e < TR::AUX; ++e) park[(NL + e) * PS] = aux[e];
    }
    ILQR_PHASE_BARRIER();
    ILQR_MARK(load_next);
    load_rec<M::NX + M::NU>(next, xu);
    ILQR_PHASE_BARRIER();
    ILQR_MARK(round0);

    bool ok = true;
    int r = 0;
    bool stuck_here = false;
    if constexpr (WIDE) {
        maxq = max_abs_packed<NL>(QL, maxq);      // round 0's matrix, by every lane as in the other kernels
        stuck_here = alive;
    } else {
    // ---- round 0 in every lane, in place (Q(0) itself is in the park), then the wavefront scans
    bool need;
    {
        int neg;
        if constexpr (EXACT) {
            maxq = max_abs_packed<NL>(QL, maxq);
            bool doubt;
            neg = exact_negative<n, m>(E, QL, QU, alive, doubt ILQR_DBG_SITE(0));
            dirty = dirty || (alive && doubt);
        } else {
            neg = has_negative_eig_packed<w, UF>(QL, maxq);
        }
        ok = ok && (neg >= 0);
        need = alive && (neg == 1);
    }
    ILQR_PHASE_BARRIER();
    if (need) regu_stages += 1;

    // ---- group scan: rounds 1..REGU_BASE of up to REGU_GROUPS trajectories at once
    ILQR_MARK(group_scan);
    unsigned long long pend = __ballot(need);
    const unsigned long long below = (1ull << lane) - 1ull;
    const int g = lane / REGU_BASE, c = lane % REGU_BASE;         // this lane tests trajectory g after c + 1 rounds
    while (pend) {
        const int rank = __popcll(pend & below);
        const bool src = need && rank < REGU_GROUPS;
        const int nsrc = min((int)__popcll(pend), REGU_GROUPS);
        wave_sync();
        if (src) {
            double PL[NL], aux[TR::AUX];
            ILQR_UNROLL
            for (int e = 0; e < TR::AUX; ++e) aux[e] = park[(NL + e) * 64];
            if constexpr (EXACT) TR::regu_p_exact(aux, PL, PUown);
            else TR::regu_p(aux, PL);
            ILQR_UNROLL
            for (int e = 0; e < NL; ++e) ldsP[rank * NL + e] = PL[e];
            srcs[rank] = lane;
            if constexpr (EXACT) {
                ILQR_UNROLL
                for (int e = 0; e < NUP; ++e) {
                    ldsQU[rank * NUP + e] = QU[e];
                    ldsPU[rank * NUP + e] = PUown[e];
                }
            }
        }
        wave_sync();
        ILQR_MARK(group_test);
        int t = 1;
        double mq = 0.0;
        bool gdoubt = false;
        if (g < nsrc) {
            double L[NL];
            if constexpr (!EXACT) build_regularised<NL>(L, regu_coef(c + 1), ldsP + g * NL, park - lane + srcs[g]);
            if constexpr (EXACT) {
                double U[NUP > 0 ? NUP : 1];
                const double* const q0 = park - lane + srcs[g];
                ILQR_UNROLL
                for (int e = 0; e < NL; ++e) L[e] = running_sum(q0[e * 64], ldsP[g * NL + e], 0, c + 1);
                ILQR_UNROLL
                for (int e = 0; e < NUP; ++e) U[e] = running_sum(ldsQU[g * NUP + e], ldsPU[g * NUP + e], 0, c + 1);
                mq = max_abs_packed<NL>(L, mq);
                t = exact_negative<n, m>(E, L, U, true, gdoubt ILQR_DBG_SITE(1));
            } else {
                t = has_negative_eig_packed<w, UF>(L, mq);
            }
        }
        // first passing round of every group, its verdict, and the largest |Q| met up to it
        ILQR_MARK(group_select);
        const unsigned long long passed = __ballot(t != 1);
        const unsigned long long gdoubts = EXACT ? __ballot(gdoubt) : 0ull;
        ILQR_UNROLL
        for (int dlt = 1; dlt < REGU_BASE; dlt *= 2) {
            const double up = __shfl_up(mq, dlt, REGU_BASE);
            if (c >= dlt) mq = fmax(mq, up);
        }
        const int byte = (int)((passed >> (REGU_BASE * (src ? rank : 0))) & ((1u << REGU_BASE) - 1u));
        const int cstar = byte ? __ffs(byte) - 1 : -1;
        const int from = (src ? rank : 0) * REGU_BASE + (cstar >= 0 ? cstar : REGU_BASE - 1);
        const int tt = __shfl(t, from);
        const double mqs = __shfl(mq, from);
        wave_sync();
        if (src) {
            maxq = fmax(maxq, mqs);
            if constexpr (EXACT) dirty = dirty || (((gdoubts >> (REGU_BASE * rank)) & ((1u << REGU_BASE) - 1u)) != 0);
            r = (cstar >= 0) ? cstar + 1 : REGU_BASE;
            if (cstar >= 0) {
                const double coef = regu_coef(r);
                const double* const pl = ldsP + rank * NL;
                // every read before the first write: the writes cannot be shown not to alias the reads, and
                // entry by entry each read would wait for the write before it
                double pe[NL], q0[NL];
                ILQR_UNROLL
                for (int e = 0; e < NL; ++e) pe[e] = pl[e];
                ILQR_UNROLL
                for (int e = 0; e < NL; ++e) q0[e] = park[e * 64];
                ILQR_PHASE_BARRIER();
                if constexpr (EXACT) {
                    ILQR_UNROLL
                    for (int e = 0; e < NL; ++e) park[e * 64] = running_sum(q0[e], pe[e], 0, r);
                    ILQR_UNROLL
                    for (int e = 0; e < NUP; ++e) QU[e] = running_sum(QU[e], PUown[e], 0, r);
                } else {
                    ILQR_UNROLL
                    for (int e = 0; e < NL; ++e) park[e * 64] = __builtin_fma(coef, pe[e], q0[e]);
                    // the entries above the diagonal get the mirrored entries of P (P is symmetric up to rounding)
                    ILQR_UNROLL
                    for (int i = 0; i < n; ++i)
                        ILQR_UNROLL
                        for (int j = i + 1; j < n; ++j)
                            QU[upp(n, i, j)] = __builtin_fma(coef, pl[tri(j, i)], QU[upp(n, i, j)]);
                    ILQR_UNROLL
                    for (int i = 0; i < m; ++i)
                        ILQR_UNROLL
                        for (int j = i + 1; j < m; ++j)
                            QU[upp(m, i, j, QP::UUB)] = __builtin_fma(coef, pl[tri(n + j, n + i)], QU[upp(m, i, j, QP::UUB)]);
                }
                ok = ok && (tt >= 0);
            } else {
                if constexpr (DEFER) deferred = true;
                else stuck_here = true;      // Q(0) stays in the park: the stuck scan builds every round from it
                if constexpr (EXACT) dirty = true;
            }
            need = false;
        }
        pend &= ~__ballot(src);
    }
    }   // !WIDE

    ILQR_MARK(stuck_scan);
#ifdef ILQR_DEBUG_COUNTERS
    const unsigned long long dbg_t0 = __builtin_readcyclecounter();
    const bool dbg_any = __ballot(stuck_here) != 0;
#endif
    if constexpr (WIDE) {
        // ---- the loops of all trajectories of the wavefront (at most WIDE_TRAJ, in lanes 0..) side by side: the 64
        // lanes are split evenly between the trajectories still looping, lane c of a trajectory's share tests round
        // base + c of its running sum, whose state after `base` rounds (the checkpoint) waits in LDS; a trajectory
        // whose share held no stop moves its checkpoint on by the share, one that stopped leaves, and the others get
        // its lanes from the next trip on.
        static_assert(WIDE_TRAJ * 2 <= REGU_GROUPS, "P and the checkpoints share the published slots");
        double* const ckL = ldsP + WIDE_TRAJ * NL;                          // checkpoints, lower triangles
        double* const ckU = ldsQU;                                          // ... entries above the diagonal
        double* const outU = ldsQU + WIDE_TRAJ * (NUP > 0 ? NUP : 1);       // accepted matrix, entries above the diagonal
        wave_sync();
        if (stuck_here) {        // (lane < WIDE_TRAJ: the host launches this kernel with at most that many per wavefront)
            double PL[NL], aux[TR::AUX];
            ILQR_UNROLL
            for (int e = 0; e < TR::AUX; ++e) aux[e] = park[(NL + e) * PS];
            TR::regu_p_exact(aux, PL, PUown);
            ILQR_UNROLL
            for (int e = 0; e < NL; ++e) {
                ldsP[lane * NL + e] = PL[e];
                ckL[lane * NL + e] = park[e * PS];
            }
            ILQR_UNROLL
            for (int e = 0; e < NUP; ++e) {
                ldsPU[lane * NUP + e] = PUown[e];
                ckU[lane * NUP + e] = QU[e];
            }
        }
        int base = 0;
        bool pending = stuck_here;
        const unsigned long long below = (1ull << lane) - 1ull;
        for (;;) {
            const unsigned long long pm = __ballot(pending);
            if (!pm) break;
            const int nu = __popcll(pm);
            const int W = (nu == 1) ? 64 : (nu == 2) ? 32 : 16;      // lanes per trajectory in this trip
            const int g = lane / W, c = lane - g * W;
            int sg = 0;                                              // the lane that owns share g: the g-th pending one
            {
                int seen = 0;
                ILQR_UNROLL
                for (int b2 = 0; b2 < WIDE_TRAJ; ++b2) {
                    const bool is = (pm >> b2) & 1ull;
                    sg = (is && seen == g) ? b2 : sg;
                    seen += is ? 1 : 0;
                }
            }
            const bool have = g < nu;
            const int bs = __shfl(base, sg);
            const unsigned long long gm = ((W == 64) ? ~0ull : ((1ull << W) - 1ull)) << (g * W);
            wave_sync();
#ifdef ILQR_DEBUG_COUNTERS
            if (pending) dbg_trips += 1;
#endif
            double mq = 0.0;
            int t;
            {
                double L[NL], U[NUP > 0 ? NUP : 1];
                ILQR_UNROLL
                for (int e = 0; e < NL; ++e) L[e] = running_sum(ckL[sg * NL + e], ldsP[sg * NL + e], bs, c);
                ILQR_UNROLL
                for (int e = 0; e < NUP; ++e) U[e] = running_sum(ckU[sg * NUP + e], ldsPU[sg * NUP + e], bs, c);
                mq = max_abs_packed<NL>(L, mq);
                t = exact_negative_ordered<n, m>(E, L, U, have, gm ILQR_DBG_SITE(2));
            }
            const unsigned long long allstops = __ballot(have && t != 1);
            const unsigned long long gstops = allstops & gm;
            const int gwin = gstops ? __ffsll((long long)gstops) - 1 : -1;            // lane that stopped this share's loop
            // largest |entry| as the other kernels take it: over rounds 1..REGU_BASE as far as the loop got, and the
            // matrix it stopped at (round 0 was taken above)
            {
                const int rnd = bs + c;
                const int rstar = gstops ? bs + (gwin - g * W) : 0x7fffffff;
                double v = (have && rnd >= 1 && rnd <= rstar && (rnd <= REGU_BASE || rnd == rstar)) ? mq : 0.0;
                ILQR_UNROLL
                for (int dlt = 1; dlt < 64; dlt *= 2) {
                    const double o = __shfl_xor(v, dlt);
                    v = (dlt < W) ? fmax(v, o) : v;
                }
                mq = v;
            }
            wave_sync();
            if (have && lane == gwin) {      // the accepted matrix, built again (the test destroyed it), to its owner
                double* const dst = park - pslot + sg;
                ILQR_UNROLL
                for (int e = 0; e < NL; ++e) dst[e * PS] = running_sum(ckL[sg * NL + e], ldsP[sg * NL + e], bs, c);
                ILQR_UNROLL
                for (int e = 0; e < NUP; ++e) outU[sg * NUP + e] = running_sum(ckU[sg * NUP + e], ldsPU[sg * NUP + e], bs, c);
            }
            if (have && !gstops && c == W - 1) {      // no stop in this share: its checkpoint moves on by W rounds
                ILQR_UNROLL
                for (int e = 0; e < NL; ++e) ckL[sg * NL + e] = running_sum(ckL[sg * NL + e], ldsP[sg * NL + e], bs, W);
                ILQR_UNROLL
                for (int e = 0; e < NUP; ++e) ckU[sg * NUP + e] = running_sum(ckU[sg * NUP + e], ldsPU[sg * NUP + e], bs, W);
            }
            // the owners collect what their shares found
            const int gi = __popcll(pm & below);                      // share of this lane's trajectory, if it has one
            const int first = pending ? gi * W : 0;
            const unsigned long long mine = pending ? (allstops & (((W == 64) ? ~0ull : ((1ull << W) - 1ull)) << first)) : 0ull;
            const int swin = mine ? __ffsll((long long)mine) - 1 : first;
            const int tw = __shfl(t, swin);
            const double mqw = __shfl(mq, first);
            wave_sync();
            if (pending) {
                maxq = fmax(maxq, mqw);
                if (mine) {
                    r = base + (swin - first);
#ifdef ILQR_DEBUG_COUNTERS
                    atomicAdd(&g_hist[1][dbg_bin(r)], 1ull);
#endif
                    if (r >= 1) regu_stages += 1;
                    ILQR_UNROLL
                    for (int e = 0; e < NUP; ++e) QU[e] = outU[lane * NUP + e];
                    ok = ok && (tw >= 0);
                    pending = false;
                } else {
                    base += W;
                    if (base > 1200) {   // 0.1 * 2^r overflowed long before: cannot happen, but never spin
                        ok = false;
                        pending = false;
                    }
                }
            }
        }
    } else if constexpr (!EXACT) {
    // ---- stuck scan: rounds REGU_BASE + 1, + 2, ... of one trajectory, 64 at a time
    unsigned long long stuck = __ballot(stuck_here);
    while (stuck) {
        const int src = __ffsll((long long)stuck) - 1;
        stuck &= stuck - 1;
        wave_sync();
        if (lane == src) {
            double PL[NL], aux[TR::AUX];
            ILQR_UNROLL
            for (int e = 0; e < TR::AUX; ++e) aux[e] = park[(NL + e) * 64];
            TR::regu_p(aux, PL);
            ILQR_UNROLL
            for (int e = 0; e < NL; ++e) ldsP[e] = PL[e];
        }
        for (int base = REGU_BASE + 1;; base += 64) {
            wave_sync();     // P and Q(0) are read from LDS again on every trip, not held in registers across the loop
            double L[NL];
            build_regularised<NL>(L, regu_coef(base + lane), ldsP, park - lane + src);
            double mq = 0.0;
            int t;
            t = has_negative_eig_packed<w, UF>(L, mq);
            const unsigned long long stops = __ballot(t != 1);
            if (stops) {
                const int win = __ffsll((long long)stops) - 1;
                const int tw = __shfl(t, win);
                const double mqw = __shfl(mq, win);
                wave_sync();
                if (lane == src) {
                    r = base + win;
                    const double cw = regu_coef(r);
                    double pe[NL], q0[NL];
                    ILQR_UNROLL
                    for (int e = 0; e < NL; ++e) pe[e] = ldsP[e];
                    ILQR_UNROLL
                    for (int e = 0; e < NL; ++e) q0[e] = park[e * 64];
                    ILQR_PHASE_BARRIER();
                    ILQR_UNROLL
                    for (int e = 0; e < NL; ++e) park[e * 64] = __builtin_fma(cw, pe[e], q0[e]);
                    ILQR_UNROLL
                    for (int i = 0; i < n; ++i)
                        ILQR_UNROLL
                        for (int j = i + 1; j < n; ++j)
                            QU[upp(n, i, j)] = __builtin_fma(cw, ldsP[tri(j, i)], QU[upp(n, i, j)]);
                    ILQR_UNROLL
                    for (int i = 0; i < m; ++i)
                        ILQR_UNROLL
                        for (int j = i + 1; j < m; ++j)
                            QU[upp(m, i, j, QP::UUB)] = __builtin_fma(cw, ldsP[tri(n + j, n + i)], QU[upp(m, i, j, QP::UUB)]);
                    ok = ok && (tw >= 0);
                    maxq = fmax(maxq, mqw);
                }
                break;
            }
            if (base > ILQR_SCAN_LIMIT) {   // 0.1 * 2^r overflowed long before: cannot happen, but never spin
                if (lane == src) ok = false;
                break;
            }
        }
    }
    }   // (EXACT with DEFER: a trajectory that would get here has been dropped from the pass)
#ifdef ILQR_DEBUG_COUNTERS
    if (dbg_any && (threadIdx.x & 63) == 0) atomicAdd(&g_dbg[12], __builtin_readcyclecounter() - dbg_t0);     // cycles in stuck scans
#endif
    regu_rounds += r;
    // No early return on the LinAlgError cases: the lane goes on with numbers nobody will read (the caller
    // drops the trajectory).  With a way out in the middle of the stage the old V_x, V_xx would have to stay
    // in registers through the factorisations for the sake of that path alone.
    ILQR_PHASE_BARRIER();
    ILQR_MARK(gains_v);
    ILQR_UNROLL
    for (int e = 0; e < NL; ++e) QL[e] = park[e * PS];      // Q(r)

    // ---- gains (iLQRController.py:125-129): d = -Quu^-1 Qu, K = -Quu^-1 Qux
    double Quu[m * m], Qinv[m * m];
    ILQR_UNROLL
    for (int a = 0; a < m; ++a)
        ILQR_UNROLL
        for (int b = 0; b < m; ++b) Quu[a * m + b] = QP::quu(QL, QU, a, b);
    ok = inv_small<m>(Quu, Qinv) && ok;
    ILQR_UNROLL
    for (int a = 0; a < m; ++a) {
        double s = 0.0;
        ILQR_UNROLL
        for (int b = 0; b < m; ++b) s += (-Qinv[a * m + b]) * Qu[b];
        dk[a] = s;
        ILQR_UNROLL
        for (int j = 0; j < n; ++j) {
            double rr = 0.0;
            ILQR_UNROLL
            for (int b = 0; b < m; ++b) rr += (-Qinv[a * m + b]) * QL[tri(n + b, j)];
            Kk[a * n + j] = rr;
        }
    }

    // ---- V_xx = Q_xx + K'Q_uu K + K'Q_ux + Q_ux'K ; V_x = Q_x + K'Q_uu d + K'Q_u + Q_ux'd  (:131-134)
    double KtQuu[n * m];
    ILQR_UNROLL
    for (int i = 0; i < n; ++i)
        ILQR_UNROLL
        for (int b = 0; b < m; ++b) {
            double s = 0.0;
            ILQR_UNROLL
            for (int a = 0; a < m; ++a) s += Kk[a * n + i] * Quu[a * m + b];
            KtQuu[i * m + b] = s;
        }
    ILQR_UNROLL
    for (int i = 0; i < n; ++i) {
        ILQR_UNROLL
        for (int j = 0; j < n; ++j) {
            double t1 = 0.0, t2 = 0.0, t3 = 0.0;
            ILQR_UNROLL
            for (int a = 0; a < m; ++a) t1 += KtQuu[i * m + a] * Kk[a * n + j];
            ILQR_UNROLL
            for (int a = 0; a < m; ++a) t2 += Kk[a * n + i] * QL[tri(n + a, j)];
            ILQR_UNROLL
            for (int a = 0; a < m; ++a) t3 += QL[tri(n + a, i)] * Kk[a * n + j];
            V[i * n + j] = ((QP::qxx(QL, QU, i, j) + t1) + t2) + t3;
        }
        double a1 = 0.0, b1 = 0.0, c1 = 0.0;
        ILQR_UNROLL
        for (int a = 0; a < m; ++a) a1 += KtQuu[i * m + a] * dk[a];
        ILQR_UNROLL
        for (int a = 0; a < m; ++a) b1 += Kk[a * n + i] * Qu[a];
        ILQR_UNROLL
        for (int a = 0; a < m; ++a) c1 += QL[tri(n + a, i)] * dk[a];
        Vx[i] = ((Qx[i] + a1) + b1) + c1;
    }

    // ---- expected cost reduction (:136-137): -Q_u'd - 0.5 d'(Q_uu' d)
    double ea = 0.0, eb = 0.0;
    ILQR_UNROLL
    for (int a = 0; a < m; ++a) ea += (-Qu[a]) * dk[a];
    ILQR_UNROLL
    for (int a = 0; a < m; ++a) {
        double s = 0.0;
        ILQR_UNROLL
        for (int b = 0; b < m; ++b) s += Quu[b * m + a] * dk[b];
        eb += (0.5 * dk[a]) * s;
    }
    redu += ea - eb;
    ILQR_MARK(stage_end);
    return ok;
}

// ------------------------------------------------------------------
// Record layout of the solver's own trajectory buffers.  Everything one stage
// of one trajectory needs sits in one contiguous, 64-byte aligned record:
//     T record = [x_k (nx), u_k (nu), pad]        stage k of slot s at T[(k * cap + s) * RT]
//     G record = [d_k (nu), K_k (nu * n), pad]    stage k of slot s at G[(k * cap + s) * RG]
// A lane moves its record with 16-byte loads/stores.  When all 64 lanes of a
// wavefront hold consecutive slots the wavefront touches one contiguous span;
// when only a few scattered slots are still in use (later line-search
// candidates) every 64-byte sector fetched is still fully used, which a
// one-double-per-lane strided layout cannot offer (1/8 of each sector).
// ------------------------------------------------------------------
template <class M>
struct Rec {
    static constexpr int nx = M::NX, m = M::NU, n = M::NDX;
    static constexpr int RT = (nx + m + 7) / 8 * 8;
    static constexpr int RG = (m + m * n + 1) / 2 * 2;
};

struct RecIn {
    const double* p;     // record of stage 0
    long ss;             // doubles between consecutive stages
};
struct RecOut {
    double* p;
    long ss;
};

// CNT doubles from a 16-byte aligned record (reads one double of padding when CNT is odd)
template <int CNT>
__device__ __forceinline__ void load_rec(const double* p, double* v)
{
    const double2* q = reinterpret_cast<const double2*>(p);
#pragma unroll
    for (int i = 0; i < CNT / 2; ++i) {
        const double2 t = q[i];
        v[2 * i] = t.x;
        v[2 * i + 1] = t.y;
    }
    if constexpr (CNT & 1) v[CNT - 1] = q[CNT / 2].x;
}
template <int CNT>
__device__ __forceinline__ void store_rec(double* p, const double* v)
{
    double2* q = reinterpret_cast<double2*>(p);
#pragma unroll
    for (int i = 0; i < CNT / 2; ++i) q[i] = make_double2(v[2 * i], v[2 * i + 1]);
    if constexpr (CNT & 1) q[CNT / 2] = make_double2(v[CNT - 1], 0.0);
}

// Plant parameters as the kernels read them: a small block is copied into
// registers once per lane (a global load in the middle of a stage would make the
// wavefront wait for every store it has in flight), a large one stays in memory.
// A small block may also be given per trajectory (ILQR_FLAG_PARAMS_PER_TRAJECTORY):
// parameter i of trajectory b at p[i * istride + b]; the shared block (istride = 1,
// per_traj = 0) is read through a branch of its own so that it stays a scalar load.
template <class M, bool SMALL = (M::NPAR <= 8)>
struct ParCache;
template <class M>
struct ParCache<M, true> {
    double v[M::NPAR];
    __device__ __forceinline__ explicit ParCache(const double* p, long istride = 1, long per_traj = 0, long b = 0)
    {
        if (per_traj == 0) {
#pragma unroll
            for (int i = 0; i < M::NPAR; ++i) v[i] = p[i];
        } else {
#pragma unroll
            for (int i = 0; i < M::NPAR; ++i) v[i] = p[(long)i * istride + b];
        }
    }
    __device__ __forceinline__ const double* get() const { return v; }
};
template <class M>
struct ParCache<M, false> {
    const double* p;
    __device__ __forceinline__ explicit ParCache(const double* q, long = 1, long = 0, long = 0) : p(q) {}
    __device__ __forceinline__ const double* get() const { return p; }
};

// Where a new trajectory goes: the T records of a slot (of the next iteration's buffer, or of the
// trajectory's own place in the export buffer when the solve ends with it).
template <class M>
struct Dest {
    static constexpr int nx = M::NX, m = M::NU;
    double* t;      // record of stage 0
    long ts;        // doubles between consecutive stages
    __device__ __forceinline__ void stage(int k, const double* xv, const double* uv) const
    {
        double r[nx + m];
#pragma unroll
        for (int i = 0; i < nx; ++i) r[i] = xv[i];
#pragma unroll
        for (int i = 0; i < m; ++i) r[nx + i] = uv[i];
        store_rec<nx + m>(t + (long)k * ts, r);
    }
    __device__ __forceinline__ void last(int k, const double* xv) const { store_rec<nx>(t + (long)k * ts, xv); }
};

// ------------------------------------------------------------------
// Whole backward pass (iLQRController.py:98-123) for one trajectory per lane.
// All 64 lanes of the wavefront call it together; a lane without a trajectory
// (`active` false) only helps the others and must be given a G it may scribble
// on.  Reads the T records, writes the G records; returns deltaJ and max_k |d_k|_2 (with NumPy's NaN propagation, as
// np.max(np.linalg.norm(d, axis=1)) has).  Returns false on LinAlgError.
// ------------------------------------------------------------------
template <class M, bool EXACT = false, bool DEFER = false>
__device__ __forceinline__ bool backward_pass(const double* par, const RecIn& T, const RecOut& G, int S, double& deltaJ,
                                     double& dmax, int& regu_rounds, int& regu_stages, double& maxq, bool active,
                                     double* lds, bool& dirty, bool& deferred)
{
#ifdef ILQR_DEBUG_COUNTERS
    int dbg_trips = 0;
#endif
    using TR = typename TermsOf<M>::type;
    constexpr int n = M::NDX, m = M::NU, nx = M::NX;
    double Vx[n], V[n * n], xu[nx + m], g[m + m * n];
    double* const dk = g;
    double* const Kk = g + m;
    load_rec<nx>(T.p + (long)S * T.ss, xu);
    TR::final_full(par, xu, Vx, V);
    double redu = 0.0, best = -INFINITY;
    bool anynan = false, ok = true;
    load_rec<nx + m>(T.p + (long)(S - 1) * T.ss, xu);
    // Nothing may still be in flight when the loop is entered: otherwise the wait-count bookkeeping of the
    // loop has to allow for these loads on every trip and ends up waiting for the previous stage's stores.
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0)
    for (int k = S - 1; k >= 0; --k) {
        const bool alive = active && ok && !deferred;
        // (the stage leaves the record of stage k - 1 in xu; the last one loads its own again)
        const bool good = backward_stage<M, EXACT, DEFER>(par, xu, T.p + (long)(k > 0 ? k - 1 : 0) * T.ss, Vx, V, dk, Kk, redu,
                                            regu_rounds, regu_stages, maxq, alive, lds, dirty, deferred
#ifdef ILQR_DEBUG_COUNTERS
                                            , dbg_trips
#endif
        );
        ok = ok && good;
        // stored by every lane, whatever its state (a lane without a trajectory is given a scratch record by
        // the caller, a failed one writes gains nobody reads): a fixed number of stores, all younger than the
        // load of the next record, so that waiting for that record does not mean waiting for them
        store_rec<m + m * n>(G.p + (long)k * G.ss, g);
        if (alive && good) {
            double s = 0.0;
            for (int i = 0; i < m; ++i) s += dk[i] * dk[i];
            double r = sqrt(s);
            anynan = anynan || (r != r);
            best = (r > best) ? r : best;
        }
    }
    deltaJ = redu;
    dmax = anynan ? NAN : best;
#ifdef ILQR_DEBUG_COUNTERS
    if (EXACT && !DEFER) {
        if (active) atomicAdd(&g_hist[0][dbg_bin(dbg_trips)], 1ull);
    }
#endif
    return ok;
}

// ------------------------------------------------------------------
// Model.rollout + Model.cost_trj fused (model.py:15-20, :31-37): x written to
// X, cost returned; the sum runs in the reference's order (stages 0..S-1, then
// the final cost).  Strided arrays in and out.
// ------------------------------------------------------------------
template <class M>
__device__ __forceinline__ double rollout_cost(const double* par, const double* x0, const CLane& U, const Lane& X, int S,
                                               int SC)
{
    constexpr int nx = M::NX, m = M::NU;
    double x[nx], xn[nx], u[m];
    for (int i = 0; i < nx; ++i) x[i] = x0[i];
    store_vec<nx>(X, 0, x);
    double total = 0.0;
    for (int k = 0; k < S; ++k) {
        load_vec<m>(U, k, u);
        if (k < SC) total += M::cost_stage(par, x, u);
        M::dynamics(par, x, u, xn);
        for (int i = 0; i < nx; ++i) x[i] = xn[i];
        store_vec<nx>(X, k + 1, x);
    }
    total += M::cost_final(par, x);
    return total;
}

// The same with the trajectory written as T records (u copied along).
template <class M>
__device__ __forceinline__ double rollout_cost_rec(const double* par, const double* x0, const CLane& U, const Dest<M>& out, int S,
                                                   int SC)
{
    constexpr int nx = M::NX, m = M::NU;
    double x[nx], xn[nx], u[m];
    for (int i = 0; i < nx; ++i) x[i] = x0[i];
    double total = 0.0;
    for (int k = 0; k < S; ++k) {
        load_vec<m>(U, k, u);
        out.stage(k, x, u);
        if (k < SC) total += M::cost_stage(par, x, u);
        M::dynamics(par, x, u, xn);
        for (int i = 0; i < nx; ++i) x[i] = xn[i];
    }
    out.last(S, x);
    total += M::cost_final(par, x);
    return total;
}

// ------------------------------------------------------------------
// _forward_pass (iLQRController.py:87-96) fused with cost_trj of the candidate
// (:64): u_new = u + alpha d + K dx, x_new = f(x_new, u_new).  The candidate
// is written to `out` when WRITE; its cost is returned.
// ------------------------------------------------------------------
// (Always inlined, like everything a kernel here calls: a kernel's scratch size is then computed from its own
// body.  Left out of line, the size is propagated from callee to caller through assembler symbols, and builds
// were seen in which that came out as 0 for a callee with a 676-byte frame -- a memory aperture violation.)
template <class M, bool WRITE = true>
__device__ __forceinline__ double forward_pass(const double* par, const RecIn& T, const RecIn& G, double alpha,
                                      const Dest<M>& out, int S, int SC)
{
    constexpr int n = M::NDX, m = M::NU, nx = M::NX;
    double xn[nx], xu[nx + m], xnext[nx], g[m + m * n], dx[n], un[m], xu2[nx + m], g2[m + m * n];
    load_rec<nx>(T.p, xn);
    load_rec<nx + m>(T.p, xu2);
    load_rec<m + m * n>(G.p, g2);
    double total = 0.0;
    for (int k = 0; k < S; ++k) {
        for (int i = 0; i < nx + m; ++i) xu[i] = xu2[i];
        for (int i = 0; i < m + m * n; ++i) g[i] = g2[i];
        if (k + 1 < S) {     // the next stage's records travel while this stage is computed
            load_rec<nx + m>(T.p + (long)(k + 1) * T.ss, xu2);
            load_rec<m + m * n>(G.p + (long)(k + 1) * G.ss, g2);
        }
        const double* const d = g;
        const double* const Kk = g + m;
        M::diff_state(par, xn, xu, dx);
        for (int i = 0; i < m; ++i) {
            double s = 0.0;
            for (int t = 0; t < n; ++t) s += Kk[i * n + t] * dx[t];
            un[i] = (xu[nx + i] + alpha * d[i]) + s;
        }
        if (WRITE) out.stage(k, xn, un);
        if (k < SC) total += M::cost_stage(par, xn, un);
        M::dynamics(par, xn, un, xnext);
        for (int i = 0; i < nx; ++i) xn[i] = xnext[i];
    }
    if (WRITE) out.last(S, xn);
    total += M::cost_final(par, xn);
    return total;
}

// The same evaluation without storing the candidate (cost only).
template <class M>
__device__ __forceinline__ double forward_cost(const double* par, const RecIn& T, const RecIn& G, double alpha, int S, int SC)
{
    return forward_pass<M, false>(par, T, G, alpha, Dest<M>{nullptr, 0}, S, SC);
}

// Copy of a trajectory (all T records) to `out`.
template <class M>
__device__ __forceinline__ void copy_traj(const RecIn& T, const Dest<M>& out, int S)
{
    constexpr int nx = M::NX, m = M::NU;
    double xu[nx + m];
    for (int k = 0; k < S; ++k) {
        load_rec<nx + m>(T.p + (long)k * T.ss, xu);
        out.stage(k, xu, xu + nx);
    }
    load_rec<nx>(T.p + (long)S * T.ss, xu);
    out.last(S, xu);
}

}  // namespace ilqr
