// Per-trajectory iLQR building blocks for gfx950, one lane per trajectory.
//
// Each function restates one piece of control/iLQRController.py for a single
// trajectory held by one lane; trajectory-sized data (x, u, d, K) lives in HBM
// in a strided structure-of-arrays layout so that the 64 lanes of a wavefront
// read and write consecutive doubles, and the small per-stage matrices live in
// the lane's registers / private memory.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

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

// ------------------------------------------------------------------
// `np.any(np.linalg.eigvals(Q) < 0)` of iLQRController.py:110, decided by an
// unpivoted right-looking LDL' of the lower triangle of
// Q = [[Qxx, Qux'], [Qux, Quu]] with deflation of zero pivots:
//   pivot < 0 -> negative eigenvalue; zero pivot with a non-zero column below
//   -> indefinite; zero pivot with a zero column -> zero eigenvalue, skip.
// A pivot that has cancelled to within PIVOT_TOL of its diagonal entry is
// rounding noise and is treated as zero (without this, a Q that is dominated
// by regu*[fx fu]'[fx fu], which is rank deficient, keeps the same noise sign
// through every further doubling of regu).
// Returns 1 (negative eigenvalue), 0 (none), -1 (Q not finite: NumPy raises).
// ------------------------------------------------------------------
constexpr double PIVOT_TOL = 1e-13;

template <int n, int m>
__device__ inline int has_negative_eig(const double* Qxx, const double* Qux, const double* Quu, double& maxq)
{
    constexpr int w = n + m;
    double W[w * w];
    bool finite = true;
    double mq = maxq;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double v = Qxx[i * n + j];
            double a = fabs(v);
            finite = finite && (a <= DBL_MAX);
            mq = (a > mq) ? a : mq;
            if (j <= i) W[i * w + j] = v;
        }
    for (int i = 0; i < m; ++i) {
        for (int j = 0; j < n; ++j) {
            double v = Qux[i * n + j];
            double a = fabs(v);
            finite = finite && (a <= DBL_MAX);
            mq = (a > mq) ? a : mq;
            W[(n + i) * w + j] = v;
        }
        for (int j = 0; j < m; ++j) {
            double v = Quu[i * m + j];
            double a = fabs(v);
            finite = finite && (a <= DBL_MAX);
            mq = (a > mq) ? a : mq;
            if (j <= i) W[(n + i) * w + n + j] = v;
        }
    }
    maxq = mq;
    if (!finite) return -1;
    int result = 0;
    for (int j = 0; j < w; ++j) {
        const double p = W[j * w + j];
        const double ajj = (j < n) ? Qxx[j * n + j] : Quu[(j - n) * m + (j - n)];
        const double thr = PIVOT_TOL * fabs(ajj);
        if (p < -thr) { result = 1; break; }
        if (p <= thr) {
            // (numerically) zero pivot: a PSD matrix then has a (numerically) zero
            // column, |S_ij|^2 <= S_ii S_jj; anything larger is indefinite
            bool nz = false;
            for (int i = j + 1; i < w; ++i) {
                double c = W[i * w + j];
                nz = nz || (c * c > 4.0 * thr * fabs(W[i * w + i]));
            }
            if (nz) { result = 1; break; }
            continue;
        }
        double pinv = 1.0 / p;
        for (int i = j + 1; i < w; ++i) {
            double lij = W[i * w + j] * pinv;
            for (int k = j + 1; k <= i; ++k) W[i * w + k] -= lij * W[k * w + j];
        }
    }
    return result;
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
// Regularisation loop of iLQRController.py:106-115,
//     regu = 0.1; while any(eigvals(Q) < 0): Q += (regu [fx fu])'[fx fu]; regu *= 2
// Because regu is 0.1 times a power of two, the update of round r is exactly
// 2^r * P with P = (0.1 [fx fu])'[fx fu] evaluated once in the reference's
// operation order, so "after r rounds" is a matrix Q(r) every lane can build:
//   r <= REGU_EXACT : Q(r) = Q(r-1) + 2^(r-1) P, accumulated one by one as the reference does;
//   r >  REGU_EXACT : Q(r) = Q(REGU_EXACT) + (2^r - 2^REGU_EXACT) P, the same sum up to rounding
//                     (only ill-conditioned solves need more than ~8 rounds).
// The result of the loop is the first r = 0, 1, 2, ... whose Q(r) has no negative eigenvalue.
//
// Execution: the first REGU_LOCKSTEP rounds run lane by lane.  A lane that
// still fails then is "stuck" (typically ~50 rounds until regu swamps the
// curvature); instead of making the other 63 trajectories of the wavefront
// wait through its scan, the wavefront scans for it: the stuck lane publishes
// Q(REGU_LOCKSTEP) and P in LDS, lane l tests Q(REGU_LOCKSTEP + 1 + l), a ballot
// picks the first r that stops the scan, and that lane hands Q(r) back.
// ------------------------------------------------------------------
constexpr int REGU_LOCKSTEP = 4;
constexpr int REGU_EXACT = 12;

template <int n, int m>
struct QPack {
    static constexpr int NQ = n * n + m * n + m * m;
    static constexpr int LDS_DOUBLES = 2 * NQ + 4;      // per-wavefront record: Q, P, then {test result, r, max|Q|}
};

// Q(r) for one entry, from q = Q(REGU_LOCKSTEP) and p = P
__device__ __forceinline__ double regu_entry(double q, double p, int r, double coef)
{
    double pw = (double)(1 << REGU_LOCKSTEP);
#pragma unroll
    for (int i = REGU_LOCKSTEP; i < REGU_EXACT; ++i) {
        q = (i < r) ? q + pw * p : q;        // pw * p is exact (power of two), so contraction cannot change it
        pw *= 2.0;
    }
    return (r > REGU_EXACT) ? __dadd_rn(q, __dmul_rn(coef, p)) : q;
}

// ------------------------------------------------------------------
// One stage of the backward Riccati sweep: iLQRController.py:103-122 with
// Model.Q_terms (model.py:45-53), the regularisation loop (:106-115), _gains
// (:125-129), _V_terms (:131-134) and _expected_cost_reduction (:136-137).
// Vx/Vxx are updated in place; d_k and K_k are returned in dk/Kk.
// Must be called by all 64 lanes of the wavefront together (`alive` = this
// lane carries a trajectory); `lds` is the wavefront's QPack record.
// Returns false on the LinAlgError cases.
// ------------------------------------------------------------------
template <class M>
__device__ inline bool backward_stage(const double* par, const double* xk, const double* uk, double* Vx, double* Vxx,
                                      double* dk, double* Kk, double& redu, int& regu_rounds, int& regu_stages,
                                      double& maxq, bool alive, double* lds)
{
    constexpr int n = M::NDX, m = M::NU;
    constexpr int NQ = QPack<n, m>::NQ;
    double lx[n], lu[m], lxx[n * n], lux[m * n], luu[m * m], fx[n * n], fu[n * m];
    M::stage(par, xk, uk, lx, lu, lxx, lux, luu, fx, fu);

    double Qx[n], Qu[m], QQ[NQ];
    double* const Qxx = QQ;
    double* const Qux = QQ + n * n;
    double* const Quu = QQ + n * n + m * n;
    {
        // Q_x = l_x + f_x' V_x ; Q_u = l_u + f_u' V_x
        for (int i = 0; i < n; ++i) {
            double s = 0.0;
            for (int t = 0; t < n; ++t) s += fx[t * n + i] * Vx[t];
            Qx[i] = lx[i] + s;
        }
        for (int i = 0; i < m; ++i) {
            double s = 0.0;
            for (int t = 0; t < n; ++t) s += fu[t * m + i] * Vx[t];
            Qu[i] = lu[i] + s;
        }
        // Q_xx = l_xx + (f_x' V_xx) f_x ; Q_ux = l_ux + (f_u' V_xx) f_x ; Q_uu = l_uu + (f_u' V_xx) f_u
        double FtV[n * n];
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int t = 0; t < n; ++t) s += fx[t * n + i] * Vxx[t * n + j];
                FtV[i * n + j] = s;
            }
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int t = 0; t < n; ++t) s += FtV[i * n + t] * fx[t * n + j];
                Qxx[i * n + j] = lxx[i * n + j] + s;
            }
        double GtV[m * n];
        for (int i = 0; i < m; ++i)
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int t = 0; t < n; ++t) s += fu[t * m + i] * Vxx[t * n + j];
                GtV[i * n + j] = s;
            }
        for (int i = 0; i < m; ++i) {
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int t = 0; t < n; ++t) s += GtV[i * n + t] * fx[t * n + j];
                Qux[i * n + j] = lux[i * n + j] + s;
            }
            for (int j = 0; j < m; ++j) {
                double s = 0.0;
                for (int t = 0; t < n; ++t) s += GtV[i * n + t] * fu[t * m + j];
                Quu[i * m + j] = luu[i * m + j] + s;
            }
        }
    }

    // ---- regularisation, rounds 0..REGU_LOCKSTEP lane by lane
    bool ok = true;
    int r = 0;
    int neg = has_negative_eig<n, m>(Qxx, Qux, Quu, maxq);
    ok = ok && (neg >= 0);
    bool need = alive && (neg == 1);
    double PP[NQ];
    if (need) {
        regu_stages += 1;
        double* const Pxx = PP;
        double* const Pux = PP + n * n;
        double* const Puu = PP + n * n + m * n;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int t = 0; t < n; ++t) s += (0.1 * fx[t * n + i]) * fx[t * n + j];
                Pxx[i * n + j] = s;
            }
        for (int i = 0; i < m; ++i) {
            for (int j = 0; j < m; ++j) {
                double s = 0.0;
                for (int t = 0; t < n; ++t) s += (0.1 * fu[t * m + i]) * fu[t * m + j];
                Puu[i * m + j] = s;
            }
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int t = 0; t < n; ++t) s += (0.1 * fu[t * m + i]) * fx[t * n + j];
                Pux[i * n + j] = s;
            }
        }
    }
    double pw = 1.0;
    while (need && r < REGU_LOCKSTEP) {
        for (int e = 0; e < NQ; ++e) QQ[e] += pw * PP[e];
        pw *= 2.0;
        r += 1;
        neg = has_negative_eig<n, m>(Qxx, Qux, Quu, maxq);
        ok = ok && (neg >= 0);
        need = (neg == 1);
    }

    // ---- lanes still failing: the wavefront scans r = REGU_LOCKSTEP+1, +2, ... for them, 64 at a time
    unsigned long long stuck = __ballot(need);
    const int lane = threadIdx.x & 63;
    while (stuck) {
        const int src = __ffsll((long long)stuck) - 1;
        stuck &= stuck - 1;
        __syncthreads();
        if (lane == src)
            for (int e = 0; e < NQ; ++e) {
                lds[e] = QQ[e];
                lds[NQ + e] = PP[e];
            }
        __syncthreads();
        for (int base = REGU_LOCKSTEP + 1;; base += 64) {
            const int rc = base + lane;
            const double coef = ldexp(1.0, rc) - ldexp(1.0, REGU_EXACT);
            double W[NQ];
            for (int e = 0; e < NQ; ++e) W[e] = regu_entry(lds[e], lds[NQ + e], rc, coef);
            double mq = 0.0;
            const int t = has_negative_eig<n, m>(W, W + n * n, W + n * n + m * n, mq);
            const unsigned long long stops = __ballot(t != 1);
            if (stops) {
                const int win = __ffsll((long long)stops) - 1;
                __syncthreads();
                if (lane == win) {
                    for (int e = 0; e < NQ; ++e) lds[e] = regu_entry(lds[e], lds[NQ + e], rc, coef);
                    lds[2 * NQ] = (double)t;
                    lds[2 * NQ + 1] = (double)rc;
                    lds[2 * NQ + 2] = mq;
                }
                __syncthreads();
                if (lane == src) {
                    for (int e = 0; e < NQ; ++e) QQ[e] = lds[e];
                    ok = ok && (lds[2 * NQ] >= 0.0);
                    r = (int)lds[2 * NQ + 1];
                    maxq = (lds[2 * NQ + 2] > maxq) ? lds[2 * NQ + 2] : maxq;
                }
                break;
            }
            if (base > 1200) {   // 0.1 * 2^r overflowed long before: cannot happen, but never spin
                if (lane == src) ok = false;
                break;
            }
        }
    }
    regu_rounds += r;
    if (!ok) return false;

    // gains: d = -Quu^-1 Qu, K = -Quu^-1 Qux
    double Qinv[m * m];
    if (!inv_small<m>(Quu, Qinv)) return false;
    for (int i = 0; i < m; ++i) {
        double s = 0.0;
        for (int t = 0; t < m; ++t) s += (-Qinv[i * m + t]) * Qu[t];
        dk[i] = s;
        for (int j = 0; j < n; ++j) {
            double rr = 0.0;
            for (int t = 0; t < m; ++t) rr += (-Qinv[i * m + t]) * Qux[t * n + j];
            Kk[i * n + j] = rr;
        }
    }

    // V_xx = Q_xx + K'Q_uu K + K'Q_ux + Q_ux'K ; V_x = Q_x + K'Q_uu d + K'Q_u + Q_ux'd
    double KtQuu[n * m];
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < m; ++j) {
            double s = 0.0;
            for (int t = 0; t < m; ++t) s += Kk[t * n + i] * Quu[t * m + j];
            KtQuu[i * m + j] = s;
        }
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) {
            double t1 = 0.0, t2 = 0.0, t3 = 0.0;
            for (int t = 0; t < m; ++t) t1 += KtQuu[i * m + t] * Kk[t * n + j];
            for (int t = 0; t < m; ++t) t2 += Kk[t * n + i] * Qux[t * n + j];
            for (int t = 0; t < m; ++t) t3 += Qux[t * n + i] * Kk[t * n + j];
            Vxx[i * n + j] = ((Qxx[i * n + j] + t1) + t2) + t3;
        }
        double a = 0.0, b = 0.0, c = 0.0;
        for (int t = 0; t < m; ++t) a += KtQuu[i * m + t] * dk[t];
        for (int t = 0; t < m; ++t) b += Kk[t * n + i] * Qu[t];
        for (int t = 0; t < m; ++t) c += Qux[t * n + i] * dk[t];
        Vx[i] = ((Qx[i] + a) + b) + c;
    }

    // expected cost reduction: -Q_u'd - 0.5 d'(Q_uu' d)
    double a = 0.0, b = 0.0;
    for (int i = 0; i < m; ++i) a += (-Qu[i]) * dk[i];
    for (int i = 0; i < m; ++i) {
        double s = 0.0;
        for (int t = 0; t < m; ++t) s += Quu[t * m + i] * dk[t];
        b += (0.5 * dk[i]) * s;
    }
    redu += a - b;
    return true;
}

// ------------------------------------------------------------------
// Whole backward pass (iLQRController.py:98-123) for one trajectory per lane.
// All 64 lanes of the wavefront call it together; a lane without a trajectory
// (`active` false) only helps the others.  Writes d [S][m] and K [S][m*n];
// returns deltaJ and max_k |d_k|_2 (with NumPy's NaN propagation, as
// np.max(np.linalg.norm(d, axis=1)) has).  Returns false on LinAlgError.
// ------------------------------------------------------------------
template <class M>
__device__ inline bool backward_pass(const double* par, const CLane& X, const CLane& U, const Lane& D, const Lane& K,
                                     int S, double& deltaJ, double& dmax, int& regu_rounds, int& regu_stages,
                                     double& maxq, bool active, double* lds)
{
    constexpr int n = M::NDX, m = M::NU, nx = M::NX;
    double Vx[n], Vxx[n * n], xk[nx], uk[m], dk[m], Kk[m * n];
    load_vec<nx>(X, S, xk);
    M::final(par, xk, Vx, Vxx);
    double redu = 0.0, best = -INFINITY;
    bool anynan = false, ok = true;
    for (int k = S - 1; k >= 0; --k) {
        load_vec<nx>(X, k, xk);
        load_vec<m>(U, k, uk);
        const bool alive = active && ok;
        const bool good = backward_stage<M>(par, xk, uk, Vx, Vxx, dk, Kk, redu, regu_rounds, regu_stages, maxq, alive,
                                            lds);
        ok = ok && good;
        if (alive && good) {
            store_vec<m>(D, k, dk);
            store_vec<m * n>(K, k, Kk);
            double s = 0.0;
            for (int i = 0; i < m; ++i) s += dk[i] * dk[i];
            double r = sqrt(s);
            anynan = anynan || (r != r);
            best = (r > best) ? r : best;
        }
    }
    deltaJ = redu;
    dmax = anynan ? NAN : best;
    return ok;
}

// ------------------------------------------------------------------
// Model.rollout + Model.cost_trj fused (model.py:15-20, :31-37): x written to
// X, cost returned; the sum runs in the reference's order (stages 0..S-1, then
// the final cost).
// ------------------------------------------------------------------
template <class M>
__device__ inline double rollout_cost(const double* par, const double* x0, const CLane& U, const Lane& X, int S)
{
    constexpr int nx = M::NX, m = M::NU;
    double x[nx], xn[nx], u[m];
    for (int i = 0; i < nx; ++i) x[i] = x0[i];
    store_vec<nx>(X, 0, x);
    double total = 0.0;
    for (int k = 0; k < S; ++k) {
        load_vec<m>(U, k, u);
        total += M::cost_stage(par, x, u);
        M::dynamics(par, x, u, xn);
        for (int i = 0; i < nx; ++i) x[i] = xn[i];
        store_vec<nx>(X, k + 1, x);
    }
    total += M::cost_final(par, x);
    return total;
}

// ------------------------------------------------------------------
// _forward_pass (iLQRController.py:87-96) fused with cost_trj of the candidate
// (:64): u_new = u + alpha d + K dx, x_new = f(x_new, u_new).  Candidate
// trajectory is written to Xn/Un; its cost is returned.
// ------------------------------------------------------------------
template <class M, bool WRITE = true>
__device__ inline double forward_pass(const double* par, const CLane& X, const CLane& U, const CLane& D,
                                      const CLane& K, double alpha, const Lane& Xn, const Lane& Un, int S)
{
    constexpr int n = M::NDX, m = M::NU, nx = M::NX;
    double xn[nx], xo[nx], xnext[nx], u[m], d[m], Kk[m * n], dx[n], un[m];
    load_vec<nx>(X, 0, xn);
    if (WRITE) store_vec<nx>(Xn, 0, xn);
    double total = 0.0;
    for (int k = 0; k < S; ++k) {
        load_vec<nx>(X, k, xo);
        load_vec<m>(U, k, u);
        load_vec<m>(D, k, d);
        load_vec<m * n>(K, k, Kk);
        M::diff_state(par, xn, xo, dx);
        for (int i = 0; i < m; ++i) {
            double s = 0.0;
            for (int t = 0; t < n; ++t) s += Kk[i * n + t] * dx[t];
            un[i] = (u[i] + alpha * d[i]) + s;
        }
        if (WRITE) store_vec<m>(Un, k, un);
        total += M::cost_stage(par, xn, un);
        M::dynamics(par, xn, un, xnext);
        for (int i = 0; i < nx; ++i) xn[i] = xnext[i];
        if (WRITE) store_vec<nx>(Xn, k + 1, xn);
    }
    total += M::cost_final(par, xn);
    return total;
}

// The same evaluation without storing the candidate (cost only).
template <class M>
__device__ inline double forward_cost(const double* par, const CLane& X, const CLane& U, const CLane& D,
                                      const CLane& K, double alpha, int S)
{
    return forward_pass<M, false>(par, X, U, D, K, alpha, Lane{nullptr, 0}, Lane{nullptr, 0}, S);
}

}  // namespace ilqr
