/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of what `np.linalg.eigvals(Q)` computes
 * (control/iLQRController.py:110) for a small real matrix: LAPACK's DGEEV with
 * JOBVL = JOBVR = 'N', i.e.
 *
 *     scaling of a badly scaled matrix (DLASCL)  ->  DGEBAL('B')  ->  DGEHRD  ->  DHSEQR('E', 'N')
 *
 * For n <= 16 every blocking crossover of LAPACK is out of reach, so the work is done by the
 * unblocked routines: DGEHD2 (Householder reduction to Hessenberg form, DLARFG + DLARF) and
 * DLAHQR (implicit double-shift QR with the Ahues-Tisseur deflation test, DLANV2 for 2x2 blocks).
 * NumPy does not ship LAPACK's sources; this file restates the published algorithms of
 * Reference-LAPACK 3.11 (the version inside the OpenBLAS 0.3.29 that NumPy 2.2 links), routine by
 * routine, keeping the order of every floating-point operation so that results can be compared
 * bit for bit with the library's own routines (tests/test_eigvals_restatement.py calls them through
 * ctypes).  The level-1/2 BLAS calls inside DLARFG / DLARF are restated in the order OpenBLAS'
 * kernels for this machine class evaluate them (see eig_dot / eig_axpy below).
 *
 * Matrices are column-major with leading dimension `ld`, indices 1-based through the macros, as in
 * the Fortran originals.
 */
#ifndef ORACLE_EIGVALS_LAPACK_H
#define ORACLE_EIGVALS_LAPACK_H
#include <float.h>
#include <math.h>
#include <string.h>

#define EIG_MAXN 16

#define EIG_SAFMIN 2.2250738585072014e-308 /* DLAMCH('S') */
#define EIG_ULP 2.220446049250313e-16      /* DLAMCH('P') = eps * base */
#define EIG_EPS 1.1102230246251565e-16     /* DLAMCH('E') */

static inline double eig_sign(double a, double b) { return signbit(b) ? -fabs(a) : fabs(a); } /* Fortran SIGN(a, b) */
static inline double eig_max(double a, double b) { return (a > b || b != b) ? a : b; } /* Fortran MAX on x86: maxsd-like */
static inline double eig_min(double a, double b) { return (a < b || b != b) ? a : b; }

/* DNRM2 as OpenBLAS evaluates it on x86-64 (its x87 kernel): squares and sums in the 80-bit extended
 * format, four accumulators fed in turn by blocks of eight elements, the remaining n mod 8 elements added to
 * the first one; the square root is taken in extended precision and rounded to double once. */
static inline double eig_nrm2(int n, const double *x, int incx)
{
    if (n <= 0) return 0.0;
    if (n == 1) return fabs(x[0]);
    long double acc[4] = {0.0L, 0.0L, 0.0L, 0.0L};
    int i = 0;
    for (; i + 8 <= n; i += 8)
        for (int t = 0; t < 8; ++t) {
            const long double v = (long double)x[(long)(i + t) * incx];
            acc[t & 3] += v * v;
        }
    for (; i < n; ++i) {
        const long double v = (long double)x[(long)i * incx];
        acc[0] += v * v;
    }
    return (double)sqrtl(((acc[2] + acc[0]) + acc[1]) + acc[3]);
}

/* DLAPY2: sqrt(x^2 + y^2) without unnecessary overflow */
static inline double eig_lapy2(double x, double y)
{
    const int xn = (x != x), yn = (y != y);
    double r = 0.0;
    if (xn) r = x;
    if (yn) r = y;
    if (!(xn || yn)) {
        const double xa = fabs(x), ya = fabs(y);
        const double w = xa > ya ? xa : ya, z = xa < ya ? xa : ya;
        if (z == 0.0 || w > DBL_MAX) r = w;
        else r = w * sqrt(1.0 + (z / w) * (z / w));
    }
    return r;
}

/* DLARFG: elementary reflector H = I - tau [1; v][1; v]', H [alpha; x] = [beta; 0] */
static inline void eig_larfg(int n, double *alpha, double *x, int incx, double *tau)
{
    if (n <= 1) { *tau = 0.0; return; }
    double xnorm = eig_nrm2(n - 1, x, incx);
    if (xnorm == 0.0) { *tau = 0.0; return; }
    double beta = -eig_sign(eig_lapy2(*alpha, xnorm), *alpha);
    const double safmin = EIG_SAFMIN / EIG_EPS;
    int knt = 0;
    if (fabs(beta) < safmin) {
        const double rsafmn = 1.0 / safmin;
        do {
            knt += 1;
            for (int i = 0; i < n - 1; ++i) x[(long)i * incx] *= rsafmn;
            beta *= rsafmn;
            *alpha *= rsafmn;
        } while (fabs(beta) < safmin && knt < 20);
        xnorm = eig_nrm2(n - 1, x, incx);
        beta = -eig_sign(eig_lapy2(*alpha, xnorm), *alpha);
    }
    *tau = (beta - *alpha) / beta;
    const double sc = 1.0 / (*alpha - beta);
    for (int i = 0; i < n - 1; ++i) x[(long)i * incx] *= sc;
    for (int j = 0; j < knt; ++j) beta *= safmin;
    *alpha = beta;
}

#define A_(i, j) a[((i) - 1) + (long)((j) - 1) * lda]

/* DGEBAL('B'): permute to isolate eigenvalues, then scale rows/columns ilo..ihi by powers of two.
 * Returns 0, or -3 when a NaN is met (LAPACK's info). */
static inline int eig_gebal(int n, double *a, int lda, int *ilo, int *ihi, double *scale)
{
    int k = 1, l = n;
    if (n == 0) { *ilo = 1; *ihi = 0; return 0; }
    /* search for rows isolating an eigenvalue and push them down (the sweep goes on below a row it has just
     * moved, and is repeated until it moves none) */
    int noconv = 1;
    while (noconv) {
        noconv = 0;
        for (int i = l; i >= 1; --i) {
            int canswap = 1;
            for (int j = 1; j <= l; ++j)
                if (i != j && A_(i, j) != 0.0) { canswap = 0; break; }
            if (canswap) {
                scale[l - 1] = (double)i;
                if (i != l) {
                    for (int t = 1; t <= l; ++t) { double v = A_(t, i); A_(t, i) = A_(t, l); A_(t, l) = v; }
                    for (int t = k; t <= n; ++t) { double v = A_(i, t); A_(i, t) = A_(l, t); A_(l, t) = v; }
                }
                noconv = 1;
                if (l == 1) { *ilo = 1; *ihi = 1; return 0; }
                l -= 1;
            }
        }
    }
    /* search for columns isolating an eigenvalue and push them left */
    noconv = 1;
    while (noconv) {
        noconv = 0;
        for (int j = k; j <= l; ++j) {
            int canswap = 1;
            for (int i = k; i <= l; ++i)
                if (i != j && A_(i, j) != 0.0) { canswap = 0; break; }
            if (canswap) {
                scale[k - 1] = (double)j;
                if (j != k) {
                    for (int t = 1; t <= l; ++t) { double v = A_(t, j); A_(t, j) = A_(t, k); A_(t, k) = v; }
                    for (int t = k; t <= n; ++t) { double v = A_(j, t); A_(j, t) = A_(k, t); A_(k, t) = v; }
                }
                noconv = 1;
                k += 1;
            }
        }
    }
    for (int i = k; i <= l; ++i) scale[i - 1] = 1.0;
    /* iterative loop for norm reduction */
    const double sclfac = 2.0, factor = 0.95;
    const double sfmin1 = EIG_SAFMIN / EIG_ULP, sfmax1 = 1.0 / sfmin1;
    const double sfmin2 = sfmin1 * sclfac, sfmax2 = 1.0 / sfmin2;
    do {
        noconv = 0;
        for (int i = k; i <= l; ++i) {
            double c = eig_nrm2(l - k + 1, &A_(k, i), 1);
            double r = eig_nrm2(l - k + 1, &A_(i, k), lda);
            double ca = 0.0, ra = 0.0;
            for (int t = 1; t <= l; ++t) { const double v = fabs(A_(t, i)); if (v > ca) ca = v; }
            for (int t = k; t <= n; ++t) { const double v = fabs(A_(i, t)); if (v > ra) ra = v; }
            if (c == 0.0 || r == 0.0) continue;
            double g = r / sclfac, f = 1.0;
            const double s = c + r;
            while (!(c >= g || eig_max(f, eig_max(c, ca)) >= sfmax2 || eig_min(r, eig_min(g, ra)) <= sfmin2)) {
                const double t = c + f + ca + r + g + ra;
                if (t != t) return -3;
                f *= sclfac; c *= sclfac; ca *= sclfac;
                r /= sclfac; g /= sclfac; ra /= sclfac;
            }
            g = c / sclfac;
            while (!(g < r || eig_max(r, ra) >= sfmax2 || eig_min(eig_min(f, c), eig_min(g, ca)) <= sfmin2)) {
                f /= sclfac; c /= sclfac; g /= sclfac; ca /= sclfac;
                r *= sclfac; ra *= sclfac;
            }
            if ((c + r) >= factor * s) continue;
            if (f < 1.0 && scale[i - 1] < 1.0) {
                if (f * scale[i - 1] <= sfmin1) continue;
            }
            if (f > 1.0 && scale[i - 1] > 1.0) {
                if (scale[i - 1] >= sfmax1 / f) continue;
            }
            g = 1.0 / f;
            scale[i - 1] *= f;
            noconv = 1;
            for (int t = k; t <= n; ++t) A_(i, t) *= g;
            for (int t = 1; t <= l; ++t) A_(t, i) *= f;
        }
    } while (noconv);
    *ilo = k;
    *ihi = l;
    return 0;
}

/* The two level-2 BLAS calls of DLARF, DGEMV and DGER, with the order of operations (and the fused
 * multiply-adds) of OpenBLAS 0.3.29's x86-64 kernels for AVX2/AVX-512 cores, established by comparing
 * against the library entry by entry for all sizes up to 16 (tests/test_eigvals_restatement.py):
 *   DGEMV('T'): rows in blocks of four; columns four at a time with one fused 4-lane accumulator, a
 *     remaining pair of columns with an unfused 2-lane accumulator, a last single column with an unfused
 *     4-lane accumulator; the last m mod 4 rows are added as one fused expression.
 *   DGEMV('N'): rows in blocks of four: per group of four columns fma(a3,x3, fma(a2,x2, fma(a0,x0, a1 x1)))
 *     added to y, a remaining pair fma(a0,x0, a1 x1), a last single column unfused; the last m mod 4 rows
 *     are a fused dot product over all columns.
 *   DGER: a_ij = fma(alpha y_j, x_i, a_ij). */
static inline double eig_hsum4(const double *t) { return (t[0] + t[2]) + (t[1] + t[3]); }
/* y = A' x, A is [m x n] column-major: DGEMV('T', m, n, 1, A, lda, x, 1, 0, y, 1) */
static inline void eig_gemv_t(int m, int n, const double *a, int lda, const double *x, double *y)
{
    const int m1 = m & ~3, m3 = m & 3, n4 = n & ~3;
    for (int j = 0; j < n; ++j) {
        const double *c = a + (long)j * lda;
        double yj = 0.0;
        if (m1) {
            double blk;
            if (j < n4) {
                double t[4] = {0.0, 0.0, 0.0, 0.0};
                for (int i = 0; i < m1; i += 4)
                    for (int k = 0; k < 4; ++k) t[k] = fma(c[i + k], x[i + k], t[k]);
                blk = eig_hsum4(t);
            } else if ((n & 2) && j < n4 + 2) {
                double t[2] = {0.0, 0.0};
                for (int i = 0; i < m1; ++i) t[i & 1] = t[i & 1] + c[i] * x[i];
                blk = t[0] + t[1];
            } else {
                double t[4] = {0.0, 0.0, 0.0, 0.0};
                for (int i = 0; i < m1; i += 4)
                    for (int k = 0; k < 4; ++k) t[k] = t[k] + c[i + k] * x[i + k];
                blk = eig_hsum4(t);
            }
            yj = yj + blk;
        }
        const double *ct = c + m1, *xt = x + m1;
        if (m3 == 1) yj = fma(ct[0], xt[0], yj);
        else if (m3 == 2) yj = yj + fma(ct[0], xt[0], ct[1] * xt[1]);
        else if (m3 == 3) yj = yj + fma(ct[2], xt[2], fma(ct[0], xt[0], ct[1] * xt[1]));
        y[j] = yj;
    }
}
/* y = A x: DGEMV('N', m, n, 1, A, lda, x, 1, 0, y, 1) */
static inline void eig_gemv_n(int m, int n, const double *a, int lda, const double *x, double *y)
{
    const int m1 = m & ~3, n4 = n & ~3;
    for (int i = 0; i < m1; ++i) {
        const double *r = a + i;
        double yi = 0.0;
        int j = 0;
        for (; j < n4; j += 4) {
            double t = r[(long)(j + 1) * lda] * x[j + 1];
            t = fma(r[(long)j * lda], x[j], t);
            t = fma(r[(long)(j + 2) * lda], x[j + 2], t);
            t = fma(r[(long)(j + 3) * lda], x[j + 3], t);
            yi = yi + t;
        }
        if (n & 2) {
            yi = yi + fma(r[(long)j * lda], x[j], r[(long)(j + 1) * lda] * x[j + 1]);
            j += 2;
        }
        if (n & 1) yi = yi + r[(long)j * lda] * x[j];
        y[i] = yi;
    }
    for (int i = m1; i < m; ++i) {
        double t = 0.0;
        for (int j = 0; j < n; ++j) t = fma(a[i + (long)j * lda], x[j], t);
        y[i] = 0.0 + t;
    }
}
/* A += alpha x y': DGER(m, n, alpha, x, 1, y, 1, A, lda) */
static inline void eig_ger(int m, int n, double alpha, const double *x, const double *y, double *a, int lda)
{
    for (int j = 0; j < n; ++j) {
        const double t = alpha * y[j];
        for (int i = 0; i < m; ++i) a[i + (long)j * lda] = fma(t, x[i], a[i + (long)j * lda]);
    }
}

/* DLARF: apply H = I - tau v v' to the m x n matrix C from the left ('L') or the right ('R') */
static inline void eig_larf(char side, int m, int n, const double *v, double tau, double *c, int ldc, double *work)
{
    const int left = (side == 'L');
    int lastv = 0, lastc = 0;
    if (tau != 0.0) {
        lastv = left ? m : n;
        while (lastv > 0 && v[lastv - 1] == 0.0) lastv -= 1;
        if (left) {       /* ILADLC: last non-zero column of C(1:lastv, :) */
            for (lastc = n; lastc >= 1; --lastc) {
                int nz = 0;
                for (int i = 0; i < lastv; ++i)
                    if (c[i + (long)(lastc - 1) * ldc] != 0.0) { nz = 1; break; }
                if (nz) break;
            }
        } else {          /* ILADLR: last non-zero row of C(:, 1:lastv) */
            lastc = 0;
            for (int j = 0; j < lastv; ++j) {
                int i = m;
                while (i >= 1 && c[(i - 1) + (long)j * ldc] == 0.0) --i;
                if (i > lastc) lastc = i;
            }
        }
    }
    if (lastv <= 0) return;
    if (left) {
        eig_gemv_t(lastv, lastc, c, ldc, v, work);
        eig_ger(lastv, lastc, -tau, v, work, c, ldc);
    } else {
        eig_gemv_n(lastc, lastv, c, ldc, v, work);
        eig_ger(lastc, lastv, -tau, work, v, c, ldc);
    }
}

/* DGEHD2: reduce A(ilo:ihi, ilo:ihi) to upper Hessenberg form by Householder reflectors */
static inline void eig_gehd2(int n, int ilo, int ihi, double *a, int lda, double *tau, double *work)
{
    for (int i = ilo; i <= ihi - 1; ++i) {
        const int r2 = (i + 2 < n) ? i + 2 : n;
        eig_larfg(ihi - i, &A_(i + 1, i), &A_(r2, i), 1, &tau[i - 1]);
        const double aii = A_(i + 1, i);
        A_(i + 1, i) = 1.0;
        eig_larf('R', ihi, ihi - i, &A_(i + 1, i), tau[i - 1], &A_(1, i + 1), lda, work);
        eig_larf('L', ihi - i, n - i, &A_(i + 1, i), tau[i - 1], &A_(i + 1, i + 1), lda, work);
        A_(i + 1, i) = aii;
    }
}
#undef A_

/* DLANV2: Schur factorisation of a real 2x2 matrix in standard form; only the eigenvalues are returned */
static inline void eig_lanv2(double a, double b, double c, double d, double *rt1r, double *rt1i, double *rt2r, double *rt2i)
{
    const double multpl = 4.0, eps = EIG_ULP;
    /* safmn2 = base ** int(log(safmin / eps) / log(base) / 2) */
    const double safmn2 = ldexp(1.0, (int)(log(EIG_SAFMIN / eps) / log(2.0) / 2.0));
    const double safmx2 = 1.0 / safmn2;
    double cs, sn;
    if (c == 0.0) {
        cs = 1.0; sn = 0.0;
    } else if (b == 0.0) {
        cs = 0.0; sn = 1.0;
        const double temp = d;
        d = a; a = temp; b = -c; c = 0.0;
    } else if ((a - d) == 0.0 && eig_sign(1.0, b) != eig_sign(1.0, c)) {
        cs = 1.0; sn = 0.0;
    } else {
        double temp = a - d;
        double p = 0.5 * temp;
        const double bcmax = eig_max(fabs(b), fabs(c));
        const double bcmis = eig_min(fabs(b), fabs(c)) * eig_sign(1.0, b) * eig_sign(1.0, c);
        double scale = eig_max(fabs(p), bcmax);
        double z = (p / scale) * p + (bcmax / scale) * bcmis;
        if (z >= multpl * eps) {
            /* real eigenvalues */
            z = p + eig_sign(sqrt(scale) * sqrt(z), p);
            a = d + z;
            d = d - (bcmax / z) * bcmis;
            const double tau = eig_lapy2(c, z);
            cs = z / tau; sn = c / tau;
            b = b - c; c = 0.0;
        } else {
            /* complex eigenvalues, or real (almost) equal eigenvalues: make the diagonal elements equal */
            int count = 0;
            double sigma = b + c;
            for (;;) {
                count += 1;
                scale = eig_max(fabs(temp), fabs(sigma));
                if (scale >= safmx2) {
                    sigma *= safmn2; temp *= safmn2;
                    if (count <= 20) continue;
                } else if (scale <= safmn2) {
                    sigma *= safmx2; temp *= safmx2;
                    if (count <= 20) continue;
                } else {
                    p = 0.5 * temp;
                    double tau = eig_lapy2(sigma, temp);
                    cs = sqrt(0.5 * (1.0 + fabs(sigma) / tau));
                    sn = -(p / (tau * cs)) * eig_sign(1.0, sigma);
                    const double aa = a * cs + b * sn, bb = -a * sn + b * cs;
                    const double cc = c * cs + d * sn, dd = -c * sn + d * cs;
                    a = aa * cs + cc * sn; b = bb * cs + dd * sn;
                    c = -aa * sn + cc * cs; d = -bb * sn + dd * cs;
                    temp = 0.5 * (a + d);
                    a = temp; d = temp;
                    if (c != 0.0) {
                        if (b != 0.0) {
                            if (eig_sign(1.0, b) == eig_sign(1.0, c)) {
                                /* real eigenvalues: reduce to upper triangular form */
                                const double sab = sqrt(fabs(b)), sac = sqrt(fabs(c));
                                p = eig_sign(sab * sac, c);
                                tau = 1.0 / sqrt(fabs(b + c));
                                a = temp + p; d = temp - p;
                                b = b - c; c = 0.0;
                            }
                        } else {
                            b = -c; c = 0.0;
                        }
                    }
                }
                break;
            }
        }
    }
    (void)cs; (void)sn;
    *rt1r = a; *rt2r = d;
    if (c == 0.0) { *rt1i = 0.0; *rt2i = 0.0; }
    else { *rt1i = sqrt(fabs(b)) * sqrt(fabs(c)); *rt2i = -*rt1i; }
}

#define H_(i, j) h[((i) - 1) + (long)((j) - 1) * ldh]

/* DLAHQR with WANTT = WANTZ = .FALSE.: eigenvalues of the Hessenberg matrix H(ilo:ihi, ilo:ihi).
 * Returns 0, or i > 0 when the eigenvalues ilo..i were not found in 30 * max(10, nh) iterations. */
static inline int eig_lahqr(int n, int ilo, int ihi, double *h, int ldh, double *wr, double *wi)
{
    const double dat1 = 3.0 / 4.0, dat2 = -0.4375;
    const int kexsh = 10;
    if (n == 0) return 0;
    if (ilo == ihi) { wr[ilo - 1] = H_(ilo, ilo); wi[ilo - 1] = 0.0; return 0; }
    for (int j = ilo; j <= ihi - 3; ++j) { H_(j + 2, j) = 0.0; H_(j + 3, j) = 0.0; }
    if (ilo <= ihi - 2) H_(ihi, ihi - 2) = 0.0;
    const int nh = ihi - ilo + 1;
    const double safmin = EIG_SAFMIN, ulp = EIG_ULP;
    const double smlnum = safmin * ((double)nh / ulp);
    const int itmax = 30 * (nh > 10 ? nh : 10);
    int kdefl = 0;
    int i = ihi;
    while (i >= ilo) {
        int l = ilo;
        int its, converged = 0;
        for (its = 0; its <= itmax; ++its) {
            /* look for a single small subdiagonal element */
            int k;
            for (k = i; k >= l + 1; --k) {
                if (fabs(H_(k, k - 1)) <= smlnum) break;
                double tst = fabs(H_(k - 1, k - 1)) + fabs(H_(k, k));
                if (tst == 0.0) {
                    if (k - 2 >= ilo) tst = tst + fabs(H_(k - 1, k - 2));
                    if (k + 1 <= ihi) tst = tst + fabs(H_(k + 1, k));
                }
                if (fabs(H_(k, k - 1)) <= ulp * tst) {
                    const double ab = eig_max(fabs(H_(k, k - 1)), fabs(H_(k - 1, k)));
                    const double ba = eig_min(fabs(H_(k, k - 1)), fabs(H_(k - 1, k)));
                    const double aa = eig_max(fabs(H_(k, k)), fabs(H_(k - 1, k - 1) - H_(k, k)));
                    const double bb = eig_min(fabs(H_(k, k)), fabs(H_(k - 1, k - 1) - H_(k, k)));
                    const double s = aa + ab;
                    if (ba * (ab / s) <= eig_max(smlnum, ulp * (bb * (aa / s)))) break;
                }
            }
            l = k;
            if (l > ilo) H_(l, l - 1) = 0.0;
            if (l >= i - 1) { converged = 1; break; }
            kdefl += 1;
            const int i1 = l, i2 = i;       /* active block only: the Schur form is not wanted */
            double h11, h12, h21, h22;
            if (kdefl % (2 * kexsh) == 0) {
                const double s = fabs(H_(i, i - 1)) + fabs(H_(i - 1, i - 2));
                h11 = dat1 * s + H_(i, i); h12 = dat2 * s; h21 = s; h22 = h11;
            } else if (kdefl % kexsh == 0) {
                const double s = fabs(H_(l + 1, l)) + fabs(H_(l + 2, l + 1));
                h11 = dat1 * s + H_(l, l); h12 = dat2 * s; h21 = s; h22 = h11;
            } else {
                h11 = H_(i - 1, i - 1); h21 = H_(i, i - 1); h12 = H_(i - 1, i); h22 = H_(i, i);
            }
            double rt1r, rt1i, rt2r, rt2i;
            {
                const double s = fabs(h11) + fabs(h12) + fabs(h21) + fabs(h22);
                if (s == 0.0) {
                    rt1r = 0.0; rt1i = 0.0; rt2r = 0.0; rt2i = 0.0;
                } else {
                    h11 = h11 / s; h21 = h21 / s; h12 = h12 / s; h22 = h22 / s;
                    const double tr = (h11 + h22) / 2.0;
                    const double det = (h11 - tr) * (h22 - tr) - h12 * h21;
                    const double rtdisc = sqrt(fabs(det));
                    if (det >= 0.0) {
                        rt1r = tr * s; rt2r = rt1r; rt1i = rtdisc * s; rt2i = -rt1i;
                    } else {
                        rt1r = tr + rtdisc; rt2r = tr - rtdisc;
                        if (fabs(rt1r - h22) <= fabs(rt2r - h22)) { rt1r = rt1r * s; rt2r = rt1r; }
                        else { rt2r = rt2r * s; rt1r = rt2r; }
                        rt1i = 0.0; rt2i = 0.0;
                    }
                }
            }
            /* look for two consecutive small subdiagonal elements */
            double v[3];
            int m;
            for (m = i - 2; m >= l; --m) {
                double h21s = fabs(H_(m + 1, m));
                double s = fabs(H_(m, m) - rt2r) + fabs(rt2i) + h21s;
                h21s = H_(m + 1, m) / s;
                v[0] = h21s * H_(m, m + 1) + (H_(m, m) - rt1r) * ((H_(m, m) - rt2r) / s) - rt1i * (rt2i / s);
                v[1] = h21s * (H_(m, m) + H_(m + 1, m + 1) - rt1r - rt2r);
                v[2] = h21s * H_(m + 2, m + 1);
                s = fabs(v[0]) + fabs(v[1]) + fabs(v[2]);
                v[0] = v[0] / s; v[1] = v[1] / s; v[2] = v[2] / s;
                if (m == l) break;
                const double h00 = fabs(H_(m - 1, m - 1)), h11a = fabs(H_(m, m)), h22a = fabs(H_(m + 1, m + 1));
                if (fabs(H_(m, m - 1)) * (fabs(v[1]) + fabs(v[2])) <= ulp * fabs(v[0]) * (h00 + h11a + h22a)) break;
            }
            /* double-shift QR step */
            for (k = m; k <= i - 1; ++k) {
                const int nr = (3 < i - k + 1) ? 3 : i - k + 1;
                if (k > m) for (int t = 0; t < nr; ++t) v[t] = H_(k + t, k - 1);
                double t1;
                eig_larfg(nr, &v[0], &v[1], 1, &t1);
                if (k > m) {
                    H_(k, k - 1) = v[0];
                    H_(k + 1, k - 1) = 0.0;
                    if (k < i - 1) H_(k + 2, k - 1) = 0.0;
                } else if (m > l) {
                    H_(k, k - 1) = H_(k, k - 1) * (1.0 - t1);
                }
                const double v2 = v[1], t2 = t1 * v2;
                if (nr == 3) {
                    const double v3 = v[2], t3 = t1 * v3;
                    for (int j = k; j <= i2; ++j) {
                        const double sum = H_(k, j) + v2 * H_(k + 1, j) + v3 * H_(k + 2, j);
                        H_(k, j) = H_(k, j) - sum * t1;
                        H_(k + 1, j) = H_(k + 1, j) - sum * t2;
                        H_(k + 2, j) = H_(k + 2, j) - sum * t3;
                    }
                    const int jend = (k + 3 < i) ? k + 3 : i;
                    for (int j = i1; j <= jend; ++j) {
                        const double sum = H_(j, k) + v2 * H_(j, k + 1) + v3 * H_(j, k + 2);
                        H_(j, k) = H_(j, k) - sum * t1;
                        H_(j, k + 1) = H_(j, k + 1) - sum * t2;
                        H_(j, k + 2) = H_(j, k + 2) - sum * t3;
                    }
                } else if (nr == 2) {
                    for (int j = k; j <= i2; ++j) {
                        const double sum = H_(k, j) + v2 * H_(k + 1, j);
                        H_(k, j) = H_(k, j) - sum * t1;
                        H_(k + 1, j) = H_(k + 1, j) - sum * t2;
                    }
                    for (int j = i1; j <= i; ++j) {
                        const double sum = H_(j, k) + v2 * H_(j, k + 1);
                        H_(j, k) = H_(j, k) - sum * t1;
                        H_(j, k + 1) = H_(j, k + 1) - sum * t2;
                    }
                }
            }
        }
        if (!converged) return i;
        if (l == i) {
            wr[i - 1] = H_(i, i); wi[i - 1] = 0.0;
        } else if (l == i - 1) {
            eig_lanv2(H_(i - 1, i - 1), H_(i - 1, i), H_(i, i - 1), H_(i, i), &wr[i - 2], &wi[i - 2], &wr[i - 1], &wi[i - 1]);
        }
        kdefl = 0;
        i = l - 1;
    }
    return 0;
}
#undef H_

/* DGEEV('N', 'N'): eigenvalues of the n x n matrix `a` (column-major, destroyed), n <= EIG_MAXN.
 * Returns 0; > 0 when the QR iteration did not converge (NumPy raises LinAlgError); -3 for a NaN met in balancing. */
static inline int eig_geev_n(int n, double *a, int lda, double *wr, double *wi)
{
    double scale[EIG_MAXN], tau[EIG_MAXN], work[EIG_MAXN];
    if (n == 0) return 0;
    const double eps = EIG_ULP;
    double smlnum = sqrt(EIG_SAFMIN) / eps;
    double bignum = 1.0 / smlnum;
    /* DLANGE('M'): NaN-propagating maximum of |a_ij| */
    double anrm = 0.0;
    for (int j = 0; j < n; ++j)
        for (int i = 0; i < n; ++i) {
            const double t = fabs(a[i + (long)j * lda]);
            if (anrm < t || t != t) anrm = t;
        }
    int scalea = 0;
    double cscale = 1.0;
    if (anrm > 0.0 && anrm < smlnum) { scalea = 1; cscale = smlnum; }
    else if (anrm > bignum) { scalea = 1; cscale = bignum; }
    if (scalea) {
        /* DLASCL('G', 0, 0, anrm, cscale, n, n, A): multiply by cscale / anrm without over/underflow */
        double cfromc = anrm, ctoc = cscale;
        int done;
        do {
            const double cfrom1 = cfromc * EIG_SAFMIN;
            double mul;
            if (cfrom1 == cfromc) {
                mul = ctoc / cfromc; done = 1;
            } else {
                const double cto1 = ctoc / (1.0 / EIG_SAFMIN);
                if (cto1 == ctoc) { mul = ctoc; done = 1; cfromc = 1.0; }
                else if (fabs(cfrom1) > fabs(ctoc) && ctoc != 0.0) { mul = EIG_SAFMIN; done = 0; cfromc = cfrom1; }
                else if (fabs(cto1) > fabs(cfromc)) { mul = 1.0 / EIG_SAFMIN; done = 0; ctoc = cto1; }
                else { mul = ctoc / cfromc; done = 1; if (mul == 1.0) break; }
            }
            for (int j = 0; j < n; ++j)
                for (int i = 0; i < n; ++i) a[i + (long)j * lda] *= mul;
        } while (!done);
    }
    int ilo, ihi;
    const int rc = eig_gebal(n, a, lda, &ilo, &ihi, scale);
    if (rc) return rc;
    eig_gehd2(n, ilo, ihi, a, lda, tau, work);
    /* DHSEQR('E', 'N'): eigenvalues isolated by the balancing, then DLAHQR on the rest */
    for (int i = 1; i <= ilo - 1; ++i) { wr[i - 1] = a[(i - 1) + (long)(i - 1) * lda]; wi[i - 1] = 0.0; }
    for (int i = ihi + 1; i <= n; ++i) { wr[i - 1] = a[(i - 1) + (long)(i - 1) * lda]; wi[i - 1] = 0.0; }
    int info = 0;
    if (ilo == ihi) { wr[ilo - 1] = a[(ilo - 1) + (long)(ilo - 1) * lda]; wi[ilo - 1] = 0.0; }
    else info = eig_lahqr(n, ilo, ihi, a, lda, wr, wi);
    if (info > 0) return info;
    if (scalea) {
        /* DLASCL('G', 0, 0, cscale, anrm, n, 1, wr) and the same for wi */
        for (int pass = 0; pass < 2; ++pass) {
            double *w = pass ? wi : wr;
            double cfromc = cscale, ctoc = anrm;
            int done;
            do {
                const double cfrom1 = cfromc * EIG_SAFMIN;
                double mul;
                if (cfrom1 == cfromc) {
                    mul = ctoc / cfromc; done = 1;
                } else {
                    const double cto1 = ctoc / (1.0 / EIG_SAFMIN);
                    if (cto1 == ctoc) { mul = ctoc; done = 1; cfromc = 1.0; }
                    else if (fabs(cfrom1) > fabs(ctoc) && ctoc != 0.0) { mul = EIG_SAFMIN; done = 0; cfromc = cfrom1; }
                    else if (fabs(cto1) > fabs(cfromc)) { mul = 1.0 / EIG_SAFMIN; done = 0; ctoc = cto1; }
                    else { mul = ctoc / cfromc; done = 1; if (mul == 1.0) break; }
                }
                for (int i = 0; i < n; ++i) w[i] *= mul;
            } while (!done);
        }
    }
    return 0;
}

#endif
