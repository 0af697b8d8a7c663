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
// One stage of the backward Riccati sweep: iLQRController.py:103-122 with
// Model.Q_terms (model.py:45-53), the regularisation loop (:106-115), _gains
// (:125-129), _V_terms (:131-134) and _expected_cost_reduction (:136-137).
// Vx/Vxx are updated in place; d_k and K_k are returned in dk/Kk.
// Returns false on the LinAlgError cases.
// ------------------------------------------------------------------
template <class M>
__device__ inline bool backward_stage(const double* par, const double* xk, const double* uk, double* Vx, double* Vxx,
                                      double* dk, double* Kk, double& redu, int& regu_rounds, double& maxq)
{
    constexpr int n = M::NDX, m = M::NU;
    double lx[n], lu[m], lxx[n * n], lux[m * n], luu[m * m], fx[n * n], fu[n * m];
    M::stage(par, xk, uk, lx, lu, lxx, lux, luu, fx, fu);

    double Qx[n], Qu[m], Qxx[n * n], Qux[m * n], Quu[m * m];
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

    // regularisation: while any(eigvals(Q) < 0): Q += regu [fx fu]'[fx fu]; regu *= 2
    double regu = 0.1;
    for (;;) {
        int neg = has_negative_eig<n, m>(Qxx, Qux, Quu, maxq);
        if (neg < 0) return false;
        if (neg == 0) break;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int t = 0; t < n; ++t) s += (regu * fx[t * n + i]) * fx[t * n + j];
                Qxx[i * n + j] += s;
            }
        for (int i = 0; i < m; ++i) {
            for (int j = 0; j < m; ++j) {
                double s = 0.0;
                for (int t = 0; t < n; ++t) s += (regu * fu[t * m + i]) * fu[t * m + j];
                Quu[i * m + j] += s;
            }
            for (int j = 0; j < n; ++j) {
                double s = 0.0;
                for (int t = 0; t < n; ++t) s += (regu * fu[t * m + i]) * fx[t * n + j];
                Qux[i * n + j] += s;
            }
        }
        regu *= 2.0;
        regu_rounds += 1;
    }

    // gains: d = -Quu^-1 Qu, K = -Quu^-1 Qux
    double Qinv[m * m];
    if (!inv_small<m>(Quu, Qinv)) return false;
    for (int i = 0; i < m; ++i) {
        double s = 0.0;
        for (int t = 0; t < m; ++t) s += (-Qinv[i * m + t]) * Qu[t];
        dk[i] = s;
        for (int j = 0; j < n; ++j) {
            double r = 0.0;
            for (int t = 0; t < m; ++t) r += (-Qinv[i * m + t]) * Qux[t * n + j];
            Kk[i * n + j] = r;
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
// Whole backward pass (iLQRController.py:98-123) for one trajectory.
// Writes d [S][m] and K [S][m*n]; returns deltaJ and max_k |d_k|_2 (with
// NumPy's NaN propagation, as np.max(np.linalg.norm(d, axis=1)) has).
// ------------------------------------------------------------------
template <class M>
__device__ inline bool backward_pass(const double* par, const CLane& X, const CLane& U, const Lane& D, const Lane& K,
                                     int S, double& deltaJ, double& dmax, int& regu_rounds, double& maxq)
{
    constexpr int n = M::NDX, m = M::NU, nx = M::NX;
    double Vx[n], Vxx[n * n], xk[nx], uk[m], dk[m], Kk[m * n];
    load_vec<nx>(X, S, xk);
    M::final(par, xk, Vx, Vxx);
    double redu = 0.0, best = -INFINITY;
    bool anynan = false;
    for (int k = S - 1; k >= 0; --k) {
        load_vec<nx>(X, k, xk);
        load_vec<m>(U, k, uk);
        if (!backward_stage<M>(par, xk, uk, Vx, Vxx, dk, Kk, redu, regu_rounds, maxq)) return false;
        store_vec<m>(D, k, dk);
        store_vec<m * n>(K, k, Kk);
        double s = 0.0;
        for (int i = 0; i < m; ++i) s += dk[i] * dk[i];
        double r = sqrt(s);
        anynan = anynan || (r != r);
        best = (r > best) ? r : best;
    }
    deltaJ = redu;
    dmax = anynan ? NAN : best;
    return true;
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
template <class M>
__device__ inline double forward_pass(const double* par, const CLane& X, const CLane& U, const CLane& D,
                                      const CLane& K, double alpha, const Lane& Xn, const Lane& Un, int S)
{
    constexpr int n = M::NDX, m = M::NU, nx = M::NX;
    double xn[nx], xo[nx], xnext[nx], u[m], d[m], Kk[m * n], dx[n], un[m];
    load_vec<nx>(X, 0, xn);
    store_vec<nx>(Xn, 0, xn);
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
        store_vec<m>(Un, k, un);
        total += M::cost_stage(par, xn, un);
        M::dynamics(par, xn, un, xnext);
        for (int i = 0; i < nx; ++i) xn[i] = xnext[i];
        store_vec<nx>(Xn, k + 1, xn);
    }
    total += M::cost_final(par, xn);
    return total;
}

}  // namespace ilqr
