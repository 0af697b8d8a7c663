/* TEST INFRASTRUCTURE ONLY -- exports the routines of eigvals_lapack.h one by one, so that tests can compare
 * each of them with the LAPACK routine it restates (called through ctypes from the OpenBLAS inside NumPy). */
#include "eigvals_lapack.h"

double eigr_nrm2(int n, const double *x, int incx) { return eig_nrm2(n, x, incx); }
double eigr_lapy2(double x, double y) { return eig_lapy2(x, y); }
void eigr_larfg(int n, double *alpha, double *x, int incx, double *tau) { eig_larfg(n, alpha, x, incx, tau); }
void eigr_larf(int left, int m, int n, const double *v, double tau, double *c, int ldc, double *work)
{
    eig_larf(left ? 'L' : 'R', m, n, v, tau, c, ldc, work);
}
int eigr_gebal(int n, double *a, int lda, int *ilo, int *ihi, double *scale) { return eig_gebal(n, a, lda, ilo, ihi, scale); }
void eigr_gehd2(int n, int ilo, int ihi, double *a, int lda, double *tau)
{
    double work[EIG_MAXN];
    eig_gehd2(n, ilo, ihi, a, lda, tau, work);
}
void eigr_lanv2(double a, double b, double c, double d, double *out4) { eig_lanv2(a, b, c, d, out4, out4 + 1, out4 + 2, out4 + 3); }
int eigr_lahqr(int n, int ilo, int ihi, double *h, int ldh, double *wr, double *wi) { return eig_lahqr(n, ilo, ihi, h, ldh, wr, wi); }
/* a: n x n, ROW-major as NumPy holds it (copied into column-major storage here); wr, wi: n each */
int eigr_eigvals(int n, const double *a, double *wr, double *wi)
{
    double f[EIG_MAXN * EIG_MAXN];
    if (n > EIG_MAXN) return -1;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) f[i + j * n] = a[i * n + j];
    return eig_geev_n(n, f, n, wr, wi);
}
/* many matrices at once: a [count][n][n] -> wr, wi [count][n], rc [count] */
void eigr_eigvals_many(long count, int n, const double *a, double *wr, double *wi, int *rc)
{
    for (long c = 0; c < count; ++c) rc[c] = eigr_eigvals(n, a + c * n * n, wr + c * n, wi + c * n);
}

/* the DGEMV restatements on their own (column-major A [m x n], lda): y = A'x and y = Ax */
void eigr_gemv_t(int m, int n, const double *a, int lda, const double *x, double *y) { eig_gemv_t(m, n, a, lda, x, y); }
void eigr_gemv_n(int m, int n, const double *a, int lda, const double *x, double *y) { eig_gemv_n(m, n, a, lda, x, y); }
