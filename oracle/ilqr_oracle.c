/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference iLQR solver.
 *
 * This file is the checker that the HIP path is compared against.  It is never
 * linked into, loaded by, or called from the shipped library or package; only
 * tests/, the smoke check and the benchmark's cpu_baseline leg may use it.
 *
 * It restates, one trajectory at a time, row-major like the NumPy original:
 *   control/iLQRController.py:26-84    run_ilqr       -> oracle_solve()
 *   control/iLQRController.py:87-96    _forward_pass  -> forward_pass()
 *   control/iLQRController.py:98-123   _backward_pass -> backward_pass()
 *   control/iLQRController.py:125-129  _gains         -> gains()
 *   control/iLQRController.py:131-134  _V_terms       -> v_terms()
 *   control/iLQRController.py:136-137  _expected_cost_reduction
 *   control/iLQRController.py:139-147  _full_hessian  -> full_hessian()
 *   control/models/model.py:15-20      rollout        -> rollout()
 *   control/models/model.py:31-37      cost_trj       -> cost_trj()
 *   control/models/model.py:45-53      Q_terms        -> q_terms_default()
 *   control/models/Quadrotor_Quat.py:267-275 Q_terms  -> q_terms_quat()
 * and the four plant models (car.py, acrobot.py, Quadrotor_EA.py,
 * Quadrotor_Quat.py; see oracle_models.h).
 *
 * Pinning: the solver control flow is pinned by golden vectors produced by the
 * reference's own iLQR class (imported from the reference tree) driving NumPy
 * restatements of the plants; see tests/golden/ and tests/golden/make_golden.py.
 * The car's dynamics/cost/derivatives are written here by hand, independently
 * of the generated code the HIP kernels use.  The CasADi-backed plants are
 * "parity unpinned" at model level (CasADi is not available; the reference
 * stores none of their outputs).
 *
 * `np.any(np.linalg.eigvals(Q) < 0)` (iLQRController.py:110) is LAPACK's DGEEV on the general
 * matrix; eigvals_lapack.h restates it routine by routine (DGEBAL, DGEHD2, DLAHQR, DLANV2 and the
 * BLAS calls inside them, in the order the OpenBLAS inside NumPy evaluates them), and on the
 * (matrix, verdict) pairs recorded from the reference (tests/golden/eig_decisions.npz) it gives
 * LAPACK's verdict every time.  The cheaper unpivoted LDL^T rule the checker used before is kept
 * as ilqr_oracle_has_negative_eig_ldl() so that tests can score it against the same pairs.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: no FMA contraction,
 * like NumPy scalar arithmetic; the fused multiply-adds of the BLAS kernels are written out as fma()).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "eigvals_lapack.h"
#include "ilqr_oracle.h"
#include "oracle_models.h"

/* ------------------------------------------------------------------ */
/* small dense helpers, row-major                                      */
/* ------------------------------------------------------------------ */
/* C[m x n] = A^T B, A is [k x m], B is [k x n] */
static void matTmul(const double *A, const double *B, double *C, int k, int m, int n)
{
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int t = 0; t < k; ++t) s = fma(A[t * m + i], B[t * n + j], s);
            C[i * n + j] = s;
        }
}
/* C[m x n] = A B, A is [m x k], B is [k x n] */
static void matmul(const double *A, const double *B, double *C, int m, int k, int n)
{
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < n; ++j) {
            double s = 0.0;
            for (int t = 0; t < k; ++t) s = fma(A[i * k + t], B[t * n + j], s);
            C[i * n + j] = s;
        }
}

/* iLQRController.py:139-147 */
static void full_hessian(const double *Qxx, const double *Qux, const double *Quu, double *Q, int n, int m)
{
    int w = n + m;
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) Q[i * w + j] = Qxx[i * n + j];
        for (int j = 0; j < m; ++j) Q[i * w + n + j] = Qux[j * n + i];
    }
    for (int i = 0; i < m; ++i) {
        for (int j = 0; j < n; ++j) Q[(n + i) * w + j] = Qux[i * n + j];
        for (int j = 0; j < m; ++j) Q[(n + i) * w + n + j] = Quu[i * m + j];
    }
}

/* The rule used before the LAPACK restatement, kept for scoring: `np.any(np.linalg.eigvals(Q) < 0)` decided by a
 * right-looking LDL^T on the lower triangle, natural order, no pivoting.
 *   pivot < -tol*acc_j                  -> a negative eigenvalue exists
 *   |pivot| <= tol*acc_j, column ~ 0    -> zero eigenvalue, deflate, go on
 *   |pivot| <= tol*acc_j, column != 0   -> indefinite 2x2 minor -> negative
 * acc_j = |Q_jj| + sum_k |l_jk^2 d_k| is the size of the terms pivot j is the difference
 * of, tol = ILQR_PIVOT_TOL: a pivot that cancelled to 1e-13 of them is rounding noise.
 * Returns 1 = negative eigenvalue, 0 = none, -1 = Q not finite. `W` is scratch of w*w doubles. */
int ilqr_oracle_has_negative_eig_ldl(const double *Q, int w, double *W)
{
    double acc[ILQR_MAX_NDX + ILQR_MAX_NU];   /* |Q_jj| + sum of |l_jk^2 d_k|: the magnitudes pivot j is a difference of */
    for (int i = 0; i < w * w; ++i) {
        if (!isfinite(Q[i])) return -1;
        W[i] = Q[i];
    }
    for (int j = 0; j < w; ++j) acc[j] = fabs(Q[j * w + j]);
    for (int j = 0; j < w; ++j) {
        double p = W[j * w + j];
        double thr = ILQR_PIVOT_TOL * acc[j];
        if (p < -thr) return 1;
        if (p <= thr) {
            /* (numerically) zero pivot: a positive semi-definite matrix then has a
             * (numerically) zero column, |S_ij|^2 <= S_ii S_jj; anything larger is indefinite */
            for (int i = j + 1; i < w; ++i) {
                double c = W[i * w + j];
                if (c * c > 4.0 * thr * fabs(W[i * w + i])) return 1;
            }
            continue;
        }
        double pinv = 1.0 / p;
        for (int i = j + 1; i < w; ++i) {
            double lij = W[i * w + j] * pinv;
            for (int k = j + 1; k <= i; ++k) W[i * w + k] -= lij * W[k * w + j];
            acc[i] += fabs(lij * W[i * w + j]);
        }
    }
    return 0;
}

/* `np.any(np.linalg.eigvals(Q) < 0)` (iLQRController.py:110): the eigenvalues DGEEV computes for the row-major
 * w x w matrix Q, compared with zero the way NumPy compares (complex numbers order by real part, then imaginary).
 * Returns 1 = some eigenvalue is negative, 0 = none, -1 = Q not finite or the QR iteration did not converge
 * (NumPy raises LinAlgError in both cases). `W` is scratch of w*w doubles. */
int ilqr_oracle_has_negative_eig_lapack(const double *Q, int w, double *W)
{
    double wr[EIG_MAXN], wi[EIG_MAXN];
    for (int i = 0; i < w; ++i)
        for (int j = 0; j < w; ++j) {
            if (!isfinite(Q[i * w + j])) return -1;
            W[i + j * w] = Q[i * w + j];      /* column-major copy of Q itself, as NumPy hands it to LAPACK */
        }
    if (eig_geev_n(w, W, w, wr, wi) != 0) return -1;
    int neg = 0;
    for (int i = 0; i < w; ++i) neg = neg || (wr[i] < 0.0) || (wr[i] == 0.0 && wi[i] < 0.0);
    return neg;
}

/* Which rule the solver applies: 0 (default) = LAPACK restated, the reference's own; 1 = the LDL^T rule, which is what
 * the HIP kernels apply -- selecting it makes this checker the kernels' algorithm on the CPU, for comparing the two
 * statistically on ill-conditioned solves, where the two rules part and no trajectory is reproducible one by one. */
static int g_psd_rule = 0;
void ilqr_oracle_set_psd_rule(int rule) { g_psd_rule = rule; }
/* Diagnostics: a function the tests may register to see every (matrix, verdict) pair the solver decides
 * (row-major w x w, verdict as returned below); called from the solver's OpenMP threads. */
static void (*g_eig_observer)(const double *Q, int w, int verdict) = 0;
void ilqr_oracle_set_eig_observer(void (*fn)(const double *, int, int)) { g_eig_observer = fn; }
int ilqr_oracle_has_negative_eig(const double *Q, int w, double *W)
{
    const int v = g_psd_rule == 1 ? ilqr_oracle_has_negative_eig_ldl(Q, w, W) : ilqr_oracle_has_negative_eig_lapack(Q, w, W);
    if (g_eig_observer) g_eig_observer(Q, w, v);
    return v;
}

/* np.linalg.inv for m <= ILQR_MAX_NU: Gauss-Jordan with partial pivoting.
 * Returns 0, or -1 when a pivot is exactly zero (NumPy: LinAlgError). */
static int inv_small(const double *A, double *Ainv, int m)
{
    double T[ILQR_MAX_NU][2 * ILQR_MAX_NU];
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) {
            T[i][j] = A[i * m + j];
            T[i][m + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < m; ++c) {
        int piv = c;
        double best = fabs(T[c][c]);
        for (int r = c + 1; r < m; ++r)
            if (fabs(T[r][c]) > best) { best = fabs(T[r][c]); piv = r; }
        if (!(best > 0.0) && !isnan(best)) return -1;
        if (piv != c)
            for (int j = 0; j < 2 * m; ++j) { double t = T[c][j]; T[c][j] = T[piv][j]; T[piv][j] = t; }
        double pinv = 1.0 / T[c][c];
        for (int j = 0; j < 2 * m; ++j) T[c][j] *= pinv;
        for (int r = 0; r < m; ++r) {
            if (r == c) continue;
            double f = T[r][c];
            if (f != 0.0)
                for (int j = 0; j < 2 * m; ++j) T[r][j] -= f * T[c][j];
        }
    }
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < m; ++j) Ainv[i * m + j] = T[i][m + j];
    return 0;
}

/* iLQRController.py:125-129 */
static int gains(const double *Quu, const double *Qu, const double *Qux, double *d, double *K, int n, int m)
{
    double Qinv[ILQR_MAX_NU * ILQR_MAX_NU];
    if (inv_small(Quu, Qinv, m)) return -1;
    for (int i = 0; i < m; ++i) {
        double s = 0.0;
        for (int t = 0; t < m; ++t) s += (-Qinv[i * m + t]) * Qu[t];
        d[i] = s;
        for (int j = 0; j < n; ++j) {
            double r = 0.0;
            for (int t = 0; t < m; ++t) r += (-Qinv[i * m + t]) * Qux[t * n + j];
            K[i * n + j] = r;
        }
    }
    return 0;
}

/* iLQRController.py:131-134 (evaluation order of the four-term sums kept) */
static void v_terms(const double *Qx, const double *Qu, const double *Qxx, const double *Qux, const double *Quu,
                    const double *K, const double *d, double *Vx, double *Vxx, int n, int m)
{
    double KtQuu[ILQR_MAX_NDX * ILQR_MAX_NU];      /* n x m */
    double T1[ILQR_MAX_NDX * ILQR_MAX_NDX], T2[ILQR_MAX_NDX * ILQR_MAX_NDX], T3[ILQR_MAX_NDX * ILQR_MAX_NDX];
    matTmul(K, Quu, KtQuu, m, n, m);
    matmul(KtQuu, K, T1, n, m, n);
    matTmul(K, Qux, T2, m, n, n);
    matTmul(Qux, K, T3, m, n, n);
    for (int i = 0; i < n * n; ++i) Vxx[i] = ((Qxx[i] + T1[i]) + T2[i]) + T3[i];
    for (int i = 0; i < n; ++i) {
        double a = 0.0, b = 0.0, c = 0.0;
        for (int t = 0; t < m; ++t) a += KtQuu[i * m + t] * d[t];
        for (int t = 0; t < m; ++t) b += K[t * n + i] * Qu[t];
        for (int t = 0; t < m; ++t) c += Qux[t * n + i] * d[t];
        Vx[i] = ((Qx[i] + a) + b) + c;
    }
}

/* iLQRController.py:136-137 */
static double expected_cost_reduction(const double *Qu, const double *Quu, const double *d, int m)
{
    double a = 0.0, b = 0.0;
    for (int i = 0; i < m; ++i) a += (-Qu[i]) * d[i];
    for (int i = 0; i < m; ++i) {
        double s = 0.0;
        for (int t = 0; t < m; ++t) s += Quu[t * m + i] * d[t];    /* Q_uu^T d */
        b += (0.5 * d[i]) * s;
    }
    return a - b;
}

/* control/models/model.py:45-53 */
static void q_terms_default(const ilqr_oracle_model *M, const double *par, const double *x,
                            const double *lx, const double *lu, const double *lxx, const double *lux,
                            const double *luu, const double *fx, const double *fu, const double *Vx,
                            const double *Vxx, double *Qx, double *Qu, double *Qxx, double *Qux, double *Quu)
{
    (void)par; (void)x;
    int n = M->ndx, m = M->nu;
    double FxtV[ILQR_MAX_NDX * ILQR_MAX_NDX], FutV[ILQR_MAX_NU * ILQR_MAX_NDX], T[ILQR_MAX_NDX * ILQR_MAX_NDX];
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
    matTmul(fx, Vxx, FxtV, n, n, n);
    matmul(FxtV, fx, T, n, n, n);
    for (int i = 0; i < n * n; ++i) Qxx[i] = lxx[i] + T[i];
    matTmul(fu, Vxx, FutV, n, m, n);
    matmul(FutV, fx, T, m, n, n);
    for (int i = 0; i < m * n; ++i) Qux[i] = lux[i] + T[i];
    matmul(FutV, fu, T, m, n, m);
    for (int i = 0; i < m * m; ++i) Quu[i] = luu[i] + T[i];
}

/* control/models/model.py:15-20 */
static void rollout(const ilqr_oracle_model *M, const double *par, const double *x0, const double *u, double *x, int S)
{
    memcpy(x, x0, sizeof(double) * M->nx);
    for (int k = 0; k < S; ++k) M->dynamics(par, x + k * M->nx, u + k * M->nu, x + (k + 1) * M->nx);
}

/* control/models/model.py:31-37 and the overrides acrobot.py:129-142,
 * Quadrotor_EA.py:185-200, Quadrotor_Quat.py:201-216 (same summation order) */
static int g_cost_N = 0;
/* The overrides sum the stage costs of the horizon the PLANT was constructed with (`range(self.N-1)`), then the
 * final cost of the trajectory's last state; 0 = the trajectory's own length (model.py:31-37). */
void ilqr_oracle_set_cost_N(int cost_N) { g_cost_N = cost_N; }
static double cost_trj(const ilqr_oracle_model *M, const double *par, const double *x, const double *u, int S)
{
    double total = 0.0;
    const int SC = g_cost_N > 0 ? g_cost_N - 1 : S;
    for (int k = 0; k < SC; ++k) total += M->cost_stage(par, x + k * M->nx, u + k * M->nu);
    total += M->cost_final(par, x + S * M->nx);
    return total;
}

/* iLQRController.py:87-96 */
static void forward_pass(const ilqr_oracle_model *M, const double *par, const double *x, const double *u,
                         const double *d, const double *K, double alpha, double *xn, double *un, int S)
{
    int nx = M->nx, nu = M->nu, n = M->ndx;
    double dx[ILQR_MAX_NDX];
    memcpy(xn, x, sizeof(double) * nx);
    for (int k = 0; k < S; ++k) {
        M->diff_state(par, xn + k * nx, x + k * nx, dx);
        for (int i = 0; i < nu; ++i) {
            double s = 0.0;
            for (int t = 0; t < n; ++t) s += K[(k * nu + i) * n + t] * dx[t];
            un[k * nu + i] = (u[k * nu + i] + alpha * d[k * nu + i]) + s;
        }
        M->dynamics(par, xn + k * nx, un + k * nu, xn + (k + 1) * nx);
    }
}

/* iLQRController.py:98-123.  Returns 0, or -1 for the LinAlgError cases. */
/* Diagnostics: histogram of regularisation rounds per stage (index 63 = 63 or more). */
static long g_regu_hist[64 + 128];   /* [0..63]: rounds per stage (63 = more); [64 + r / 16]: coarse bins */
static int g_regu_hist_on = 0;
void ilqr_oracle_regu_hist_coarse(long* out128) { memcpy(out128, g_regu_hist + 64, 128 * sizeof(long)); }
void ilqr_oracle_regu_hist(int enable, long* out64)
{
    if (out64) memcpy(out64, g_regu_hist, 64 * sizeof(long));
    if (enable >= 0) {
        g_regu_hist_on = enable & 1;
        memset(g_regu_hist, 0, sizeof(g_regu_hist));
    }
}

static int backward_pass(const ilqr_oracle_model *M, const double *par, const double *x, const double *u,
                         double *d, double *K, double *deltaJ, int S, ilqr_oracle_info *info)
{
    int nx = M->nx, nu = M->nu, n = M->ndx, w = n + nu;
    double Vx[ILQR_MAX_NDX], Vxx[ILQR_MAX_NDX * ILQR_MAX_NDX];
    double Vx2[ILQR_MAX_NDX], Vxx2[ILQR_MAX_NDX * ILQR_MAX_NDX];
    double lx[ILQR_MAX_NX], lu[ILQR_MAX_NU], lxx[ILQR_MAX_NX * ILQR_MAX_NX], lux[ILQR_MAX_NU * ILQR_MAX_NX],
        luu[ILQR_MAX_NU * ILQR_MAX_NU], fx[ILQR_MAX_NDX * ILQR_MAX_NDX], fu[ILQR_MAX_NDX * ILQR_MAX_NU];
    double Qx[ILQR_MAX_NDX], Qu[ILQR_MAX_NU], Qxx[ILQR_MAX_NDX * ILQR_MAX_NDX], Qux[ILQR_MAX_NU * ILQR_MAX_NDX],
        Quu[ILQR_MAX_NU * ILQR_MAX_NU];
    double Q[(ILQR_MAX_NDX + ILQR_MAX_NU) * (ILQR_MAX_NDX + ILQR_MAX_NU)],
        W[(ILQR_MAX_NDX + ILQR_MAX_NU) * (ILQR_MAX_NDX + ILQR_MAX_NU)];
    double redu = 0.0;

    M->final_derivs(par, x + S * nx, Vx, Vxx);
    for (int k = S - 1; k >= 0; --k) {
        const double *xk = x + k * nx, *uk = u + k * nu;
        M->stage_derivs(par, xk, uk, lx, lu, lxx, lux, luu, fx, fu);
        M->q_terms(M, par, xk, lx, lu, lxx, lux, luu, fx, fu, Vx, Vxx, Qx, Qu, Qxx, Qux, Quu);
        /* regularisation loop, iLQRController.py:106-115:
         *   while any(eigvals(Q) < 0): Q += (regu*[fx fu])'[fx fu]; regu *= 2   with regu = 0.1 * 2^r.
         * Because regu is 0.1 times a power of two, the r-th update is exactly 2^r * P0 with
         * P0 = (0.1*[fx fu])'[fx fu] evaluated once in the reference's operation order; the updates are
         * accumulated one round after the other, like the reference's in-place additions. */
        full_hessian(Qxx, Qux, Quu, Q, n, nu);
        int r = 0;
        int have_p = 0;
        double Pxx[ILQR_MAX_NDX * ILQR_MAX_NDX], Pux[ILQR_MAX_NU * ILQR_MAX_NDX], Puu[ILQR_MAX_NU * ILQR_MAX_NU];
        for (;;) {
            for (int i = 0; i < w * w; ++i) {
                double a = fabs(Q[i]);
                if (a > info->max_abs_q) info->max_abs_q = a;   /* NaN never wins: comparisons false */
            }
            int neg = ilqr_oracle_has_negative_eig(Q, w, W);
            if (neg < 0) return -1;
            if (!neg) break;
            if (!have_p) {
                have_p = 1;
                info->regu_stages += 1;
                for (int i = 0; i < n; ++i)
                    for (int j = 0; j < n; ++j) {
                        double s = 0.0;
                        for (int t = 0; t < n; ++t) s = fma(0.1 * fx[t * n + i], fx[t * n + j], s);
                        Pxx[i * n + j] = s;
                    }
                for (int i = 0; i < nu; ++i) {
                    for (int j = 0; j < nu; ++j) {
                        double s = 0.0;
                        for (int t = 0; t < n; ++t) s = fma(0.1 * fu[t * nu + i], fu[t * nu + j], s);
                        Puu[i * nu + j] = s;
                    }
                    for (int j = 0; j < n; ++j) {
                        double s = 0.0;
                        for (int t = 0; t < n; ++t) s = fma(0.1 * fu[t * nu + i], fx[t * n + j], s);
                        Pux[i * n + j] = s;
                    }
                }
            }
            for (int i = 0; i < n * n; ++i) Qxx[i] += ldexp(Pxx[i], r);
            for (int i = 0; i < nu * nu; ++i) Quu[i] += ldexp(Puu[i], r);
            for (int i = 0; i < nu * n; ++i) Qux[i] += ldexp(Pux[i], r);
            full_hessian(Qxx, Qux, Quu, Q, n, nu);
            r += 1;
            info->regu_rounds += 1;
        }
        if (g_regu_hist_on) {
            long* slot = &g_regu_hist[r < 63 ? r : 63];
#pragma omp atomic
            *slot += 1;
            slot = &g_regu_hist[64 + (r / 16 < 127 ? r / 16 : 127)];
#pragma omp atomic
            *slot += 1;
        }
        double *dk = d + k * nu, *Kk = K + k * nu * n;
        if (gains(Quu, Qu, Qux, dk, Kk, n, nu)) return -1;
        v_terms(Qx, Qu, Qxx, Qux, Quu, Kk, dk, Vx2, Vxx2, n, nu);
        memcpy(Vx, Vx2, sizeof(double) * n);
        memcpy(Vxx, Vxx2, sizeof(double) * n * n);
        redu += expected_cost_reduction(Qu, Quu, dk, nu);
    }
    *deltaJ = redu;
    return 0;
}

/* np.max(np.linalg.norm(d, axis=1)) with NumPy's NaN propagation */
static double max_row_norm(const double *d, int S, int m)
{
    double best = -INFINITY;
    int nan = 0;
    for (int k = 0; k < S; ++k) {
        double s = 0.0;
        for (int i = 0; i < m; ++i) s += d[k * m + i] * d[k * m + i];
        double r = sqrt(s);
        if (isnan(r)) nan = 1;
        if (r > best) best = r;
    }
    return nan ? NAN : best;
}

/* iLQRController.py:26-84 */
int ilqr_oracle_solve(const ilqr_oracle_model *M, const double *par, int N, int max_iter, int max_ls, double d_tol,
                      const double *x0, double *u, double *x, ilqr_oracle_info *info)
{
    int nx = M->nx, nu = M->nu, n = M->ndx, S = N - 1;
    double *d = (double *)malloc(sizeof(double) * S * nu);
    double *K = (double *)calloc((size_t)S * nu * n, sizeof(double));
    double *xn = (double *)malloc(sizeof(double) * N * nx);
    double *un = (double *)malloc(sizeof(double) * S * nu);
    memset(info, 0, sizeof(*info));

    rollout(M, par, x0, u, x, S);
    double J = cost_trj(M, par, x, u, S);
    info->J0 = J;
    for (int i = 0; i < S * nu; ++i) d[i] = DBL_MAX;
    int iters = 0;
    int status = ILQR_EXIT_DTOL;
    double dmax = max_row_norm(d, S, nu);
    while (1) {
        if (!(dmax > d_tol)) { status = ILQR_EXIT_DTOL; break; }
        if (!(iters < max_iter)) { status = ILQR_EXIT_MAXITER; break; }
        iters += 1;
        double deltaJ = 0.0;
        if (backward_pass(M, par, x, u, d, K, &deltaJ, S, info)) { status = ILQR_EXIT_LINALG; break; }
        if (iters <= ILQR_TRACE_MAX) info->deltaJ[iters - 1] = deltaJ;
        dmax = max_row_norm(d, S, nu);
        if (deltaJ < 1.0) { status = ILQR_EXIT_SMALL_DELTAJ; break; }
        int ls = 0, complete = 0, abandon = 0;
        double alpha = 1.0, Jn = 0.0;
        while (!complete && !abandon) {
            ls += 1;
            forward_pass(M, par, x, u, d, K, alpha, xn, un, S);
            Jn = cost_trj(M, par, xn, un, S);
            complete = (Jn - J <= -1e-2 * alpha * deltaJ) && (ls < max_ls);
            abandon = isnan(Jn) || (ls >= max_ls);
            alpha *= 0.5;
        }
        if (complete) {
            memcpy(x, xn, sizeof(double) * N * nx);
            memcpy(u, un, sizeof(double) * S * nu);
        }
        if (iters <= ILQR_TRACE_MAX) {
            info->ls_evals[iters - 1] = ls;
            info->accepted[iters - 1] = (uint8_t)complete;
            info->J[iters - 1] = Jn;
        }
        info->ls_total += ls;
        J = Jn;
    }
    info->iters = iters;
    info->status = status;
    info->J_last = J;
    info->dmax = dmax;
    free(d); free(K); free(xn); free(un);
    return status;
}

const ilqr_oracle_model *ilqr_oracle_get_model(int model_id)
{
    return oracle_model_table(model_id);
}

/* Batch driver: AoS per trajectory (x0[B][nx], u[B][S][nu], x[B][N][nx]),
 * one OpenMP thread per core, trajectories independent. */
int ilqr_oracle_solve_batch(int model_id, const double *par, int N, int max_iter, int max_ls, double d_tol,
                            long B, const double *x0, double *u, double *x, ilqr_oracle_info *info, int nthreads)
{
    const ilqr_oracle_model *M = oracle_model_table(model_id);
    if (!M) return -1;
    long S = N - 1;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(nthreads > 0 ? nthreads : 1)
#endif
    for (long b = 0; b < B; ++b)
        ilqr_oracle_solve(M, par, N, max_iter, max_ls, d_tol, x0 + b * M->nx, u + b * S * M->nu,
                          x + b * (long)N * M->nx, info + b);
    return 0;
}

/* Single-point model evaluations for unit tests. */
int ilqr_oracle_eval(int model_id, const double *par, const double *xv, const double *uv, double *xn, double *cs,
                     double *cf, double *lx, double *lu, double *lxx, double *lux, double *luu, double *fx,
                     double *fu, double *Vx, double *Vxx)
{
    const ilqr_oracle_model *M = oracle_model_table(model_id);
    if (!M) return -1;
    M->dynamics(par, xv, uv, xn);
    *cs = M->cost_stage(par, xv, uv);
    *cf = M->cost_final(par, xv);
    M->stage_derivs(par, xv, uv, lx, lu, lxx, lux, luu, fx, fu);
    M->final_derivs(par, xv, Vx, Vxx);
    return 0;
}

void ilqr_oracle_model_dims(int model_id, int *nx, int *nu, int *ndx, int *npar)
{
    const ilqr_oracle_model *M = oracle_model_table(model_id);
    *nx = M ? M->nx : 0; *nu = M ? M->nu : 0; *ndx = M ? M->ndx : 0; *npar = M ? M->npar : 0;
}

/* exposed for unit tests against the imported reference's pure functions */
void ilqr_oracle_unit_qterms(int n, int m, const double *lx, const double *lu, const double *lxx, const double *lux,
                             const double *luu, const double *fx, const double *fu, const double *Vx,
                             const double *Vxx, double *Qx, double *Qu, double *Qxx, double *Qux, double *Quu)
{
    ilqr_oracle_model M;
    memset(&M, 0, sizeof(M));
    M.nx = n; M.ndx = n; M.nu = m;
    q_terms_default(&M, NULL, NULL, lx, lu, lxx, lux, luu, fx, fu, Vx, Vxx, Qx, Qu, Qxx, Qux, Quu);
}
int ilqr_oracle_unit_gains_vterms(int n, int m, const double *Qx, const double *Qu, const double *Qxx,
                                  const double *Qux, const double *Quu, double *d, double *K, double *Vx,
                                  double *Vxx, double *ecr)
{
    if (gains(Quu, Qu, Qux, d, K, n, m)) return -1;
    v_terms(Qx, Qu, Qxx, Qux, Quu, K, d, Vx, Vxx, n, m);
    *ecr = expected_cost_reduction(Qu, Quu, d, m);
    return 0;
}

void q_terms_default_export(const ilqr_oracle_model *M, const double *par, const double *x,
                            const double *lx, const double *lu, const double *lxx, const double *lux,
                            const double *luu, const double *fx, const double *fu, const double *Vx,
                            const double *Vxx, double *Qx, double *Qu, double *Qxx, double *Qux, double *Quu)
{
    q_terms_default(M, par, x, lx, lu, lxx, lux, luu, fx, fu, Vx, Vxx, Qx, Qu, Qxx, Qux, Quu);
}
