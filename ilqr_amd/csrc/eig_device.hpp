// `np.any(np.linalg.eigvals(Q) < 0)` (control/iLQRController.py:110) decided the way NumPy decides it: by the
// eigenvalues LAPACK's DGEEV computes for the general (not symmetrised) matrix.
//
// DGEEV with JOBVL = JOBVR = 'N' is: scaling of a badly scaled matrix (DLASCL), DGEBAL('B'), DGEHRD, DHSEQR('E','N').
// For the orders met here (n <= 16) every blocking crossover of LAPACK is out of reach, so the work is done by the
// unblocked DGEHD2 (DLARFG + DLARF) and by DLAHQR (implicit double-shift QR, Ahues-Tisseur deflation test, DLANV2
// for 2x2 blocks).  This file states those routines for one matrix per lane, in the operation order of
// Reference-LAPACK 3.12 and of the level-1/2 BLAS kernels (DNRM2, DGEMV, DGER) of the OpenBLAS that NumPy links:
// where the smallest eigenvalue of Q is rounding noise -- and in ill-conditioned solves Q is not even close to
// symmetric -- the verdict is a property of that computation, not of the matrix.
//
// Storage: the lane's matrix lives in LDS, column-major, entry e at a[e * ES] (ES = 64: lanes interleaved, no bank
// conflicts for lane-uniform e); `work` is a second lane-interleaved area of 2 W doubles.  All loops are per lane;
// a wavefront runs as long as its slowest lane.
#pragma once
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>

namespace ilqr {
namespace eig {

#ifdef ILQR_DEBUG_COUNTERS
// cycles of wavefronts in has_negative(): [0] loading, norm and scaling, [1] DGEBAL or its shortcut, [2] DGEHD2,
// [3] DLAHQR, [4] calls, [5] calls in which some lane ran DGEBAL itself, [6] QR sweeps of the slowest lane, summed
static __device__ unsigned long long g_eig_cycles[8];
// per matrix, summed over matrices: [0] matrices, [1] QR iterations, [2] trips of the deflation search, [3] trips of
// the search for two small subdiagonals, [5] reflector steps of the sweeps;
// [6] QR iterations of the matrix with the most in its wavefront's test, summed over tests
static __device__ unsigned long long g_eig_trips[12];   // [8..11] cycles of the first lane: splits, shifts, search for m, sweep
#endif
constexpr double SAFMIN = 2.2250738585072014e-308;   // DLAMCH('S')
constexpr double ULP = 2.220446049250313e-16;        // DLAMCH('P') = eps * base
constexpr double EPS = 1.1102230246251565e-16;       // DLAMCH('E')

__device__ __forceinline__ double sgn(double a, double b) { return __builtin_copysign(__builtin_fabs(a), b); }   // Fortran SIGN
__device__ __forceinline__ double mx(double a, double b) { return (a > b || b != b) ? a : b; }
__device__ __forceinline__ double mn(double a, double b) { return (a < b || b != b) ? a : b; }

// ---- 80-bit extended arithmetic of the x87 DNRM2 kernel, carried as an unevaluated sum hi + lo ----
// (hi, lo) rounded to a 64-bit significand: lo becomes a multiple of ulp(hi) / 2^11, ties to even
__device__ __forceinline__ void round64(double& hi, double& lo)
{
#pragma clang fp contract(off)
    const double s = hi + lo;                 // renormalise so that |lo| <= ulp(hi) / 2
    lo = lo - (s - hi);
    hi = s;
    if (hi == 0.0 || !(__builtin_fabs(hi) <= DBL_MAX)) return;
    const int e = __builtin_amdgcn_frexp_exp(hi);                    // hi = f * 2^e, 0.5 <= |f| < 1: ulp(hi) = 2^(e - 53)
    const double c = __builtin_ldexp(1.5, e - 53 - 11 + 52);          // 1.5 * 2^52 * (ulp / 2^11)
    lo = (lo + c) - c;
}
// sqrt(x1^2 + ... + xn^2): squares and sums in the extended format (n < 8: one accumulator, in order; n >= 8: four
// accumulators fed in turn by blocks of eight, the rest added to the first, summed as ((a2 + a0) + a1) + a3), the
// square root in the extended format, rounded to double once more
__device__ inline double nrm2(int n, const double* x, int stride)
{
#pragma clang fp contract(off)
    if (n <= 0) return 0.0;
    if (n == 1) return __builtin_fabs(x[0]);
    double ah[4] = {0.0, 0.0, 0.0, 0.0}, al[4] = {0.0, 0.0, 0.0, 0.0};
    auto add_square = [&](int k, double v) {
        double p = v * v, pe = __builtin_fma(v, v, -p);
        round64(p, pe);
        // (ah, al) + (p, pe)
        const double s = ah[k] + p;
        const double bb = s - ah[k];
        const double err = (ah[k] - (s - bb)) + (p - bb);
        double lo = err + (al[k] + pe);
        double hi = s;
        round64(hi, lo);
        ah[k] = hi; al[k] = lo;
    };
    int i = 0;
    for (; i + 8 <= n; i += 8)
        for (int t = 0; t < 8; ++t) {
            const int k = t & 3;
            if (k == 0) add_square(0, x[(long)(i + t) * stride]);
            else if (k == 1) add_square(1, x[(long)(i + t) * stride]);
            else if (k == 2) add_square(2, x[(long)(i + t) * stride]);
            else add_square(3, x[(long)(i + t) * stride]);
        }
    for (; i < n; ++i) add_square(0, x[(long)i * stride]);
    double hi = ah[0], lo = al[0];
    if (n >= 8) {
        auto add2 = [&](double bh, double bl) {
            const double s = hi + bh;
            const double bb = s - hi;
            const double err = (hi - (s - bb)) + (bh - bb);
            lo = err + (lo + bl);
            hi = s;
            round64(hi, lo);
        };
        hi = ah[2]; lo = al[2];
        add2(ah[0], al[0]);
        add2(ah[1], al[1]);
        add2(ah[3], al[3]);
    }
    if (!(hi > 0.0) || !(hi <= DBL_MAX)) return __builtin_sqrt(hi);
    // square root of hi + lo to extended precision, then to double
    const double r = __builtin_sqrt(hi);
    const double res = __builtin_fma(-r, r, hi) + lo;
    double rl = res / (2.0 * r);
    double rh = r;
    round64(rh, rl);
    return rh + rl;
}

// nrm2() for a vector of N < 8 entries held in registers (one accumulator, in order: the same operations)
template <int N>
__device__ __forceinline__ double nrm2_small(const double (&x)[N])
{
#pragma clang fp contract(off)
    static_assert(N >= 1 && N < 8, "single-accumulator case of nrm2()");
    if (N == 1) return __builtin_fabs(x[0]);
    double hi = 0.0, lo = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double v = x[i];
        double p = v * v, pe = __builtin_fma(v, v, -p);
        round64(p, pe);
        const double s = hi + p;
        const double bb = s - hi;
        const double err = (hi - (s - bb)) + (p - bb);
        double l2 = err + (lo + pe);
        double h2 = s;
        round64(h2, l2);
        hi = h2; lo = l2;
    }
    if (!(hi > 0.0) || !(hi <= DBL_MAX)) return __builtin_sqrt(hi);
    const double r = __builtin_sqrt(hi);
    const double res = __builtin_fma(-r, r, hi) + lo;
    double rl = res / (2.0 * r);
    double rh = r;
    round64(rh, rl);
    return rh + rl;
}

// DLAPY2
__device__ __forceinline__ double lapy2(double x, double y)
{
#pragma clang fp contract(off)
    const bool xn = (x != x), yn = (y != y);
    double r = 0.0;
    if (xn) r = x;
    if (yn) r = y;
    if (!(xn || yn)) {
        const double xa = __builtin_fabs(x), ya = __builtin_fabs(y);
        const double w = xa > ya ? xa : ya, z = xa < ya ? xa : ya;
        if (z == 0.0 || w > DBL_MAX) r = w;
        else {
            const double q = z / w;
            r = w * __builtin_sqrt(1.0 + q * q);
        }
    }
    return r;
}

// DLARFG: H = I - tau [1; v][1; v]', H [alpha; x] = [beta; 0]; x (n - 1 entries, stride xs) becomes v
__device__ inline void larfg(int n, double& alpha, double* x, int xs, double& tau)
{
#pragma clang fp contract(off)
    if (n <= 1) { tau = 0.0; return; }
    double xnorm = nrm2(n - 1, x, xs);
    if (xnorm == 0.0) { tau = 0.0; return; }
    double beta = -sgn(lapy2(alpha, xnorm), alpha);
    const double safmin = SAFMIN / EPS;
    int knt = 0;
    if (__builtin_fabs(beta) < safmin) {
        const double rsafmn = 1.0 / safmin;
        do {
            knt += 1;
            for (int i = 0; i < n - 1; ++i) x[(long)i * xs] *= rsafmn;
            beta *= rsafmn;
            alpha *= rsafmn;
        } while (__builtin_fabs(beta) < safmin && knt < 20);
        xnorm = nrm2(n - 1, x, xs);
        beta = -sgn(lapy2(alpha, xnorm), alpha);
    }
    tau = (beta - alpha) / beta;
    const double sc = 1.0 / (alpha - beta);
    for (int i = 0; i < n - 1; ++i) x[(long)i * xs] *= sc;
    for (int j = 0; j < knt; ++j) beta *= safmin;
    alpha = beta;
}

// One matrix per lane.  W = order and leading dimension, ES = doubles between consecutive entries.
template <int W, int ES>
struct Lane {
    double* a;        // the matrix, entry (i, j), 1-based, at a[((i - 1) + (j - 1) * W) * ES]
    double* wk;       // 2 W doubles of work space, entry e at wk[e * ES]
    __device__ __forceinline__ double& A(int i, int j) const { return a[((i - 1) + (j - 1) * W) * ES]; }
    __device__ __forceinline__ double* pA(int i, int j) const { return a + ((i - 1) + (j - 1) * W) * ES; }
    __device__ __forceinline__ double& work(int e) const { return wk[e * ES]; }
    __device__ __forceinline__ double& scale(int e) const { return wk[(W + e) * ES]; }

    // DGEBAL('B').  Returns false when a NaN is met.
    __device__ bool gebal(int& ilo, int& ihi) const
    {
#pragma clang fp contract(off)
        const int n = W;
        int k = 1, l = n;
        bool noconv = true;
        while (noconv) {        // rows isolating an eigenvalue are pushed down
            noconv = false;
            for (int i = l; i >= 1; --i) {
                bool canswap = true;
                for (int j = 1; j <= l; ++j)
                    if (i != j && A(i, j) != 0.0) { canswap = false; break; }
                if (canswap) {
                    scale(l - 1) = (double)i;
                    if (i != l) {
                        for (int t = 1; t <= l; ++t) { const double v = A(t, i); A(t, i) = A(t, l); A(t, l) = v; }
                        for (int t = k; t <= n; ++t) { const double v = A(i, t); A(i, t) = A(l, t); A(l, t) = v; }
                    }
                    noconv = true;
                    if (l == 1) { ilo = 1; ihi = 1; return true; }
                    l -= 1;
                }
            }
        }
        noconv = true;
        while (noconv) {        // columns isolating an eigenvalue are pushed left
            noconv = false;
            for (int j = k; j <= l; ++j) {
                bool canswap = true;
                for (int i = k; i <= l; ++i)
                    if (i != j && A(i, j) != 0.0) { canswap = false; break; }
                if (canswap) {
                    scale(k - 1) = (double)j;
                    if (j != k) {
                        for (int t = 1; t <= l; ++t) { const double v = A(t, j); A(t, j) = A(t, k); A(t, k) = v; }
                        for (int t = k; t <= n; ++t) { const double v = A(j, t); A(j, t) = A(k, t); A(k, t) = v; }
                    }
                    noconv = true;
                    k += 1;
                }
            }
        }
        for (int i = k; i <= l; ++i) scale(i - 1) = 1.0;
        const double sclfac = 2.0, factor = 0.95;
        const double sfmin1 = SAFMIN / ULP, sfmax1 = 1.0 / sfmin1;
        const double sfmin2 = sfmin1 * sclfac, sfmax2 = 1.0 / sfmin2;
        do {                    // rows and columns k..l are scaled by powers of two until their norms are close
            noconv = false;
            for (int i = k; i <= l; ++i) {
                double ca = 0.0, ra = 0.0;
                for (int t = 1; t <= l; ++t) { const double v = __builtin_fabs(A(t, i)); if (v > ca) ca = v; }
                for (int t = k; t <= n; ++t) { const double v = __builtin_fabs(A(i, t)); if (v > ra) ra = v; }
                double f = 1.0;
                bool act = false, settled = false;
                {
                    // What the norms c, r of column and row i decide is f, a power of two, and whether the pair is
                    // scaled at all, both through comparisons of c and r times powers of two: functions of c / r, f a
                    // monotone step function of it and the last test linear in it for a given f.  With all sizes far
                    // from the over- and underflow guards, plainly summed norms settle both whenever the two ends of
                    // their error interval (1e-12, a thousand times their rounding) agree; the extended-precision
                    // norms are only taken where they do not.
                    double c2 = 0.0, r2 = 0.0;
                    for (int t = k; t <= l; ++t) {
                        const double v = A(t, i), w2 = A(i, t);
                        c2 = __builtin_fma(v, v, c2);
                        r2 = __builtin_fma(w2, w2, r2);
                    }
                    const bool inrange = c2 > 1e-200 && c2 < 1e200 && r2 > 1e-200 && r2 < 1e200 && ca < 1e100 && ra < 1e100;
                    if (inrange) {
                        const double cp = __builtin_sqrt(c2), rp = __builtin_sqrt(r2);
                        double fe[2];
                        bool ae[2];
                        for (int e = 0; e < 2; ++e) {
                            double c = cp * (e ? 1.0 - 1e-12 : 1.0 + 1e-12), r = rp * (e ? 1.0 + 1e-12 : 1.0 - 1e-12);
                            const double s0 = c + r;
                            double ff = 1.0;
                            while (c < r / sclfac) { ff *= sclfac; c *= sclfac; r /= sclfac; }
                            while (c / sclfac >= r) { ff /= sclfac; c /= sclfac; r *= sclfac; }
                            fe[e] = ff;
                            ae[e] = !((c + r) >= factor * s0);
                        }
                        if (fe[0] == fe[1] && ae[0] == ae[1]) { f = fe[0]; act = ae[0]; settled = true; }
                    }
                }
                if (!settled) {
                    double c = nrm2(l - k + 1, pA(k, i), ES);
                    double r = nrm2(l - k + 1, pA(i, k), W * ES);
                    if (c == 0.0 || r == 0.0) continue;
                    double g = r / sclfac;
                    const double s = c + r;
                    while (!(c >= g || mx(f, mx(c, ca)) >= sfmax2 || mn(r, mn(g, ra)) <= sfmin2)) {
                        const double t = c + f + ca + r + g + ra;
                        if (t != t) return false;
                        f *= sclfac; c *= sclfac; ca *= sclfac;
                        r /= sclfac; g /= sclfac; ra /= sclfac;
                    }
                    g = c / sclfac;
                    while (!(g < r || mx(r, ra) >= sfmax2 || mn(mn(f, c), mn(g, ca)) <= sfmin2)) {
                        f /= sclfac; c /= sclfac; g /= sclfac; ca /= sclfac;
                        r *= sclfac; ra *= sclfac;
                    }
                    act = !((c + r) >= factor * s);
                }
                if (!act) continue;
                if (f < 1.0 && scale(i - 1) < 1.0) {
                    if (f * scale(i - 1) <= sfmin1) continue;
                }
                if (f > 1.0 && scale(i - 1) > 1.0) {
                    if (scale(i - 1) >= sfmax1 / f) continue;
                }
                double g;
                g = 1.0 / f;
                scale(i - 1) *= f;
                noconv = true;
                for (int t = k; t <= n; ++t) A(i, t) *= g;
                for (int t = 1; t <= l; ++t) A(t, i) *= f;
            }
        } while (noconv);
        ilo = k;
        ihi = l;
        return true;
    }

    // True when DGEBAL('B') provably leaves the matrix as it is (ilo = 1, ihi = W, no scaling), decided without its
    // extended-precision norms.  The permutation phase moves nothing unless a row or a column is zero off the diagonal.
    // The scaling phase leaves row/column i alone when c >= r / 2 and c / 2 < r (c, r the 2-norms of column and row i:
    // neither of its loops runs, f stays 1 and c + r >= 0.95 (c + r)); here that is asked of plainly summed squares
    // with room for their rounding, 0.26 r^2 <= c^2 <= 3.9 r^2, both non-zero.  Entries are finite and, after DGEEV's
    // own scaling, far from the overflow of a square.  Anything else takes gebal() itself.
    __device__ bool gebal_is_identity() const
    {
        double r2[W], c2[W];
        bool rz[W], cz[W];
#pragma unroll
        for (int i = 0; i < W; ++i) { r2[i] = 0.0; c2[i] = 0.0; rz[i] = true; cz[i] = true; }
#pragma unroll
        for (int j = 0; j < W; ++j)
#pragma unroll
            for (int i = 0; i < W; ++i) {
                const double v = a[(i + j * W) * ES];
                const double q = v * v;
                r2[i] += q;
                c2[j] += q;
                if (i != j) {
                    rz[i] = rz[i] && (v == 0.0);
                    cz[j] = cz[j] && (v == 0.0);
                }
            }
        bool same = true;
#pragma unroll
        for (int i = 0; i < W; ++i)
            same = same && !rz[i] && !cz[i] && (c2[i] > 0.0) && (r2[i] > 0.0) && (c2[i] >= 0.26 * r2[i]) && (c2[i] <= 3.9 * r2[i]);
        return same;
    }

    // ---- DGEMV / DGER as OpenBLAS' AVX2 / AVX-512 kernels evaluate them (c = first entry, stride ES per row,
    //      W * ES per column) ----
    // work[j] = sum_i C(i, j) v(i), j < n: rows in blocks of four -- columns four at a time with one fused 4-lane
    // accumulator, a remaining pair with an unfused 2-lane accumulator, a last single column with an unfused 4-lane
    // accumulator -- then the last m mod 4 rows as one fused expression
    __device__ void gemv_t(int m, int n, const double* c, const double* v) const
    {
#pragma clang fp contract(off)
        const int m1 = m & ~3, m3 = m & 3, n4 = n & ~3;
        for (int j = 0; j < n; ++j) {
            const double* cj = c + (long)j * W * ES;
            double yj = 0.0;
            if (m1) {
                double blk;
                if (j < n4) {
                    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
                    for (int i = 0; i < m1; i += 4) {
                        t0 = __builtin_fma(cj[(i + 0) * ES], v[(i + 0) * ES], t0);
                        t1 = __builtin_fma(cj[(i + 1) * ES], v[(i + 1) * ES], t1);
                        t2 = __builtin_fma(cj[(i + 2) * ES], v[(i + 2) * ES], t2);
                        t3 = __builtin_fma(cj[(i + 3) * ES], v[(i + 3) * ES], t3);
                    }
                    blk = (t0 + t2) + (t1 + t3);
                } else if ((n & 2) && j < n4 + 2) {
                    double t0 = 0.0, t1 = 0.0;
                    for (int i = 0; i < m1; i += 2) {
                        t0 = t0 + cj[i * ES] * v[i * ES];
                        t1 = t1 + cj[(i + 1) * ES] * v[(i + 1) * ES];
                    }
                    blk = t0 + t1;
                } else {
                    double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
                    for (int i = 0; i < m1; i += 4) {
                        t0 = t0 + cj[(i + 0) * ES] * v[(i + 0) * ES];
                        t1 = t1 + cj[(i + 1) * ES] * v[(i + 1) * ES];
                        t2 = t2 + cj[(i + 2) * ES] * v[(i + 2) * ES];
                        t3 = t3 + cj[(i + 3) * ES] * v[(i + 3) * ES];
                    }
                    blk = (t0 + t2) + (t1 + t3);
                }
                yj = yj + blk;
            }
            const double* ct = cj + m1 * ES;
            const double* vt = v + m1 * ES;
            if (m3 == 1) yj = __builtin_fma(ct[0], vt[0], yj);
            else if (m3 == 2) yj = yj + __builtin_fma(ct[0], vt[0], ct[ES] * vt[ES]);
            else if (m3 == 3) yj = yj + __builtin_fma(ct[2 * ES], vt[2 * ES], __builtin_fma(ct[0], vt[0], ct[ES] * vt[ES]));
            work(j) = yj;
        }
    }
    // work[i] = sum_j C(i, j) v(j), i < m: the first m - m mod 4 rows per group of four columns
    // fma(c3,v3, fma(c2,v2, fma(c0,v0, c1 v1))) added to y, a remaining pair fma(c0,v0, c1 v1), a last column
    // unfused; the last m mod 4 rows a fused dot product over all columns
    __device__ void gemv_n(int m, int n, const double* c, const double* v) const
    {
#pragma clang fp contract(off)
        const int m1 = m & ~3, n4 = n & ~3;
        const long cs = (long)W * ES;
        for (int i = 0; i < m1; ++i) {
            const double* r = c + i * ES;
            double yi = 0.0;
            int j = 0;
            for (; j < n4; j += 4) {
                double t = r[(j + 1) * cs] * v[(j + 1) * ES];
                t = __builtin_fma(r[j * cs], v[j * ES], t);
                t = __builtin_fma(r[(j + 2) * cs], v[(j + 2) * ES], t);
                t = __builtin_fma(r[(j + 3) * cs], v[(j + 3) * ES], t);
                yi = yi + t;
            }
            if (n & 2) {
                yi = yi + __builtin_fma(r[j * cs], v[j * ES], r[(j + 1) * cs] * v[(j + 1) * ES]);
                j += 2;
            }
            if (n & 1) yi = yi + r[j * cs] * v[j * ES];
            work(i) = yi;
        }
        for (int i = m1; i < m; ++i) {
            double t = 0.0;
            for (int j = 0; j < n; ++j) t = __builtin_fma(c[i * ES + j * cs], v[j * ES], t);
            work(i) = 0.0 + t;
        }
    }

    // DLARF: C = H C ('L', v has m entries) or C = C H ('R', v has n entries), H = I - tau v v'; C is m x n at c
    __device__ void larf(bool left, int m, int n, const double* v, double tau, double* c) const
    {
#pragma clang fp contract(off)
        const long cs = (long)W * ES;
        int lastv = 0, lastc = 0;
        if (tau != 0.0) {
            lastv = left ? m : n;
            while (lastv > 0 && v[(lastv - 1) * ES] == 0.0) lastv -= 1;
            if (left) {       // last non-zero column of C(1:lastv, :)
                for (lastc = n; lastc >= 1; --lastc) {
                    bool nz = false;
                    for (int i = 0; i < lastv; ++i)
                        if (c[i * ES + (lastc - 1) * cs] != 0.0) { nz = true; break; }
                    if (nz) break;
                }
            } else {          // last non-zero row of C(:, 1:lastv)
                lastc = 0;
                for (int j = 0; j < lastv; ++j) {
                    int i = m;
                    while (i >= 1 && c[(i - 1) * ES + j * cs] == 0.0) --i;
                    if (i > lastc) lastc = i;
                }
            }
        }
        if (lastv <= 0) return;
        const double mt = -tau;
        if (left) {
            gemv_t(lastv, lastc, c, v);
            for (int j = 0; j < lastc; ++j) {       // DGER: c_ij = fma(-tau w_j, v_i, c_ij)
                const double t = mt * work(j);
                for (int i = 0; i < lastv; ++i) c[i * ES + j * cs] = __builtin_fma(t, v[i * ES], c[i * ES + j * cs]);
            }
        } else {
            gemv_n(lastc, lastv, c, v);
            for (int j = 0; j < lastv; ++j) {
                const double t = mt * v[j * ES];
                for (int i = 0; i < lastc; ++i) c[i * ES + j * cs] = __builtin_fma(t, work(i), c[i * ES + j * cs]);
            }
        }
    }

    // DGEHD2.  `left_only`: reflector ilo has been generated and applied from the right already (its vector is in
    // place); alpha0, tau0 are its beta and tau.
    __device__ void gehd2(int ilo, int ihi, bool left_only = false, double alpha0 = 0.0, double tau0 = 0.0) const
    {
        const int n = W;
        for (int i = ilo; i <= ihi - 1; ++i) {
            const int r2 = (i + 2 < n) ? i + 2 : n;
            double alpha = A(i + 1, i), tau;
            const bool resume = left_only && i == ilo;
            if (resume) { alpha = alpha0; tau = tau0; }
            else larfg(ihi - i, alpha, pA(r2, i), ES, tau);
            A(i + 1, i) = 1.0;
            if (!resume) larf(false, ihi, ihi - i, pA(i + 1, i), tau, pA(1, i + 1));
            larf(true, ihi - i, n - i, pA(i + 1, i), tau, pA(i + 1, i + 1));
            A(i + 1, i) = alpha;
        }
    }

    // ---- DGEHD2 of the whole matrix (ilo = 1, ihi = W) with the matrix in registers, for W <= 8 ----
    // R[r + c W], 0-based.  Reflector I0 (0-based) is done here when nothing about it depends on where zeros happen
    // to sit: DLARF shortens its products to the last non-zero entry of v and the last non-zero row / column of C, and
    // the order of summation in the products depends on those lengths, so a reflector whose v ends in a zero, whose C
    // has a zero last row (from the right) or a zero last column (from the left), or whose beta needs DLARFG's
    // rescaling, is left to the routines above, which continue from the matrix as it then is.  Same operations in the
    // same order as larfg / gemv_n / gemv_t / the DGER loops for these sizes.
    // state: 0 going on, 1 stopped before reflector `at`, 2 stopped before the left application of reflector `at`
    template <int I0>
    __device__ __forceinline__ void hess_step(double (&R)[W * W], int& state, int& at, double& alpha_s, double& tau_s) const
    {
#pragma clang fp contract(off)
        constexpr int NV = W - 1 - I0;       // entries of [alpha; x] = rows I0 + 1 .. W - 1 of column I0
        constexpr int C0 = I0 + 1;           // first row / column of the blocks the reflector is applied to
        static_assert(NV >= 2, "reflectors with a non-empty x");
        if (state != 0) return;
        double alpha = R[C0 + I0 * W];
        double x[NV - 1];
#pragma unroll
        for (int t = 0; t < NV - 1; ++t) x[t] = R[C0 + 1 + t + I0 * W];
        const double xnorm = nrm2_small<NV - 1>(x);
        if (xnorm == 0.0) return;            // tau = 0: H = I, nothing changes
        const double beta = -sgn(lapy2(alpha, xnorm), alpha);
        bool plain = !(__builtin_fabs(beta) < SAFMIN / EPS);
        const double tau = (beta - alpha) / beta;
        const double sc = 1.0 / (alpha - beta);
        double v[NV];
        v[0] = 1.0;
#pragma unroll
        for (int t = 0; t < NV - 1; ++t) v[1 + t] = x[t] * sc;
        plain = plain && (tau != 0.0) && (v[NV - 1] != 0.0);
        {
            bool lastrow = false;            // from the right: C = A(0 .. W-1, C0 .. W-1), its last row
#pragma unroll
            for (int j = 0; j < NV; ++j) lastrow = lastrow || (R[(W - 1) + (C0 + j) * W] != 0.0);
            plain = plain && lastrow;
        }
        if (!plain) { state = 1; at = I0; return; }
#pragma unroll
        for (int t = 0; t < NV - 1; ++t) R[C0 + 1 + t + I0 * W] = v[1 + t];
        const double mt = -tau;
        double wk[W];
        // ---- C = C H from the right: work = C v (gemv_n with m = W rows, n = NV columns), then the rank-one update
        {
            constexpr int m1 = W & ~3, n4 = NV & ~3;
#pragma unroll
            for (int i = 0; i < m1; ++i) {
                double yi = 0.0;
#pragma unroll
                for (int j = 0; j < n4; j += 4) {
                    double t = R[i + (C0 + j + 1) * W] * v[j + 1];
                    t = __builtin_fma(R[i + (C0 + j) * W], v[j], t);
                    t = __builtin_fma(R[i + (C0 + j + 2) * W], v[j + 2], t);
                    t = __builtin_fma(R[i + (C0 + j + 3) * W], v[j + 3], t);
                    yi = yi + t;
                }
                int j = n4;
                if (NV & 2) {
                    yi = yi + __builtin_fma(R[i + (C0 + j) * W], v[j], R[i + (C0 + j + 1) * W] * v[j + 1]);
                    j += 2;
                }
                if (NV & 1) yi = yi + R[i + (C0 + j) * W] * v[j];
                wk[i] = yi;
            }
#pragma unroll
            for (int i = m1; i < W; ++i) {
                double t = 0.0;
#pragma unroll
                for (int j = 0; j < NV; ++j) t = __builtin_fma(R[i + (C0 + j) * W], v[j], t);
                wk[i] = 0.0 + t;
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const double t = mt * v[j];
#pragma unroll
                for (int i = 0; i < W; ++i) R[i + (C0 + j) * W] = __builtin_fma(t, wk[i], R[i + (C0 + j) * W]);
            }
        }
        // ---- C = H C from the left: C = A(C0 .. W-1, C0 .. W-1), NV x NV; its last column must not be zero
        {
            bool lastcol = false;
#pragma unroll
            for (int i = 0; i < NV; ++i) lastcol = lastcol || (R[(C0 + i) + (W - 1) * W] != 0.0);
            if (!lastcol) { state = 2; at = I0; alpha_s = beta; tau_s = tau; return; }
            constexpr int m1 = NV & ~3, m3 = NV & 3, n4 = NV & ~3;
            double y[NV];
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                double yj = 0.0;
                if (m1) {
                    double blk;
                    if (j < n4) {
                        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
                        for (int i = 0; i < m1; i += 4) {
                            t0 = __builtin_fma(R[(C0 + i + 0) + (C0 + j) * W], v[i + 0], t0);
                            t1 = __builtin_fma(R[(C0 + i + 1) + (C0 + j) * W], v[i + 1], t1);
                            t2 = __builtin_fma(R[(C0 + i + 2) + (C0 + j) * W], v[i + 2], t2);
                            t3 = __builtin_fma(R[(C0 + i + 3) + (C0 + j) * W], v[i + 3], t3);
                        }
                        blk = (t0 + t2) + (t1 + t3);
                    } else if ((NV & 2) && j < n4 + 2) {
                        double t0 = 0.0, t1 = 0.0;
#pragma unroll
                        for (int i = 0; i < m1; i += 2) {
                            t0 = t0 + R[(C0 + i) + (C0 + j) * W] * v[i];
                            t1 = t1 + R[(C0 + i + 1) + (C0 + j) * W] * v[i + 1];
                        }
                        blk = t0 + t1;
                    } else {
                        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
                        for (int i = 0; i < m1; i += 4) {
                            t0 = t0 + R[(C0 + i + 0) + (C0 + j) * W] * v[i + 0];
                            t1 = t1 + R[(C0 + i + 1) + (C0 + j) * W] * v[i + 1];
                            t2 = t2 + R[(C0 + i + 2) + (C0 + j) * W] * v[i + 2];
                            t3 = t3 + R[(C0 + i + 3) + (C0 + j) * W] * v[i + 3];
                        }
                        blk = (t0 + t2) + (t1 + t3);
                    }
                    yj = yj + blk;
                }
                if (m3 == 1) yj = __builtin_fma(R[(C0 + m1) + (C0 + j) * W], v[m1], yj);
                else if (m3 == 2)
                    yj = yj + __builtin_fma(R[(C0 + m1) + (C0 + j) * W], v[m1], R[(C0 + m1 + 1) + (C0 + j) * W] * v[m1 + 1]);
                else if (m3 == 3)
                    yj = yj + __builtin_fma(R[(C0 + m1 + 2) + (C0 + j) * W], v[m1 + 2],
                                            __builtin_fma(R[(C0 + m1) + (C0 + j) * W], v[m1], R[(C0 + m1 + 1) + (C0 + j) * W] * v[m1 + 1]));
                y[j] = yj;
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const double t = mt * y[j];
#pragma unroll
                for (int i = 0; i < NV; ++i) R[(C0 + i) + (C0 + j) * W] = __builtin_fma(t, v[i], R[(C0 + i) + (C0 + j) * W]);
            }
        }
        R[C0 + I0 * W] = beta;
    }
    template <int I0>
    __device__ __forceinline__ void hess_steps(double (&R)[W * W], int& state, int& at, double& alpha_s, double& tau_s) const
    {
        if constexpr (I0 <= W - 3) {
            hess_step<I0>(R, state, at, alpha_s, tau_s);
            hess_steps<I0 + 1>(R, state, at, alpha_s, tau_s);
        }
    }
    // DGEHD2(1, W): in registers as far as the reflectors are plain, the rest by gehd2()
    __device__ void gehd2_whole() const
    {
        if constexpr (W >= 3 && W <= 8) {
            double R[W * W];
#pragma unroll
            for (int e = 0; e < W * W; ++e) R[e] = a[e * ES];
            int state = 0, at = W - 2;
            double alpha_s = 0.0, tau_s = 0.0;
            hess_steps<0>(R, state, at, alpha_s, tau_s);
#pragma unroll
            for (int e = 0; e < W * W; ++e) a[e * ES] = R[e];
            if (state == 1) gehd2(at + 1, W);
            else if (state == 2) gehd2(at + 1, W, true, alpha_s, tau_s);
        } else {
            gehd2(1, W);
        }
    }

    // DLAHQR, eigenvalues only (WANTT = WANTZ = false).  `neg` collects "an eigenvalue compares below zero".
    // Returns false when an eigenvalue was not found in 30 * max(10, nh) iterations.
    __device__ bool lahqr(int ilo, int ihi, bool& neg) const
    {
#pragma clang fp contract(off)
        const double dat1 = 3.0 / 4.0, dat2 = -0.4375;
        const int kexsh = 10;
        if (ilo == ihi) { neg = neg || (A(ilo, ilo) < 0.0); return true; }
        for (int j = ilo; j <= ihi - 3; ++j) { A(j + 2, j) = 0.0; A(j + 3, j) = 0.0; }
        if (ilo <= ihi - 2) A(ihi, ihi - 2) = 0.0;
        const int nh = ihi - ilo + 1;
        const double safmin = SAFMIN, ulp = ULP;
        const double smlnum = safmin * ((double)nh / ulp);
        const int itmax = 30 * (nh > 10 ? nh : 10);
        int kdefl = 0;
        int i = ihi;
#ifdef ILQR_DEBUG_COUNTERS
        unsigned dq = 0, dd = 0, dm = 0, dk = 0;
#endif
        // DLAHQR's two nested loops (over the blocks that split off at the bottom, and over the QR iterations that make
        // one split off) run here as ONE loop whose trip is: look for the places where the matrix has split, record
        // every block that has (cheap, repeated until a block needs more work), then one QR iteration.  Per matrix the
        // operations and their order are DLAHQR's; for the wavefront it means that all lanes make their QR iterations
        // together, whichever block each is working on, instead of waiting block by block for the lane that needs
        // the most iterations for its current one.
        int l = ilo, its = 0;
        bool active = true, failed = false;
#ifdef ILQR_DEBUG_COUNTERS
        unsigned long long dbg_t = __builtin_readcyclecounter(), dbg_acc[4] = {0, 0, 0, 0};
        auto dbg_lap = [&](int slot) {
            const unsigned long long now = __builtin_readcyclecounter();
            dbg_acc[slot] += now - dbg_t;
            dbg_t = now;
        };
#endif
        while (active) {
            bool sweep = false;
            int k = 0;
            while (active && !sweep) {
                if (its > itmax) { failed = true; active = false; break; }
                for (k = i; k >= l + 1; --k) {       // look for a single small subdiagonal element
#ifdef ILQR_DEBUG_COUNTERS
                    dd += 1;
#endif
                    const double hkk1 = __builtin_fabs(A(k, k - 1));
                    if (hkk1 <= smlnum) break;
                    const double hk1 = A(k - 1, k - 1), hk = A(k, k);
                    double tst = __builtin_fabs(hk1) + __builtin_fabs(hk);
                    if (tst == 0.0) {
                        if (k - 2 >= ilo) tst = tst + __builtin_fabs(A(k - 1, k - 2));
                        if (k + 1 <= ihi) tst = tst + __builtin_fabs(A(k + 1, k));
                    }
                    if (hkk1 <= ulp * tst) {
                        const double hk1k = __builtin_fabs(A(k - 1, k));
                        const double ab = mx(hkk1, hk1k), ba = mn(hkk1, hk1k);
                        const double dd = __builtin_fabs(hk1 - hk);
                        const double aa = mx(__builtin_fabs(hk), dd), bb = mn(__builtin_fabs(hk), dd);
                        const double s = aa + ab;
                        if (ba * (ab / s) <= mx(smlnum, ulp * (bb * (aa / s)))) break;
                    }
                }
                l = k;
                if (l > ilo) A(l, l - 1) = 0.0;
                if (l >= i - 1) {        // one or two eigenvalues have split off
                    if (l == i) {
                        neg = neg || (A(i, i) < 0.0);
                    } else {
                        double r1r, r1i, r2r, r2i;
                        lanv2(A(i - 1, i - 1), A(i - 1, i), A(i, i - 1), A(i, i), r1r, r1i, r2r, r2i);
                        // complex numbers compare by real part, then by imaginary part
                        neg = neg || (r1r < 0.0) || (r1r == 0.0 && r1i < 0.0) || (r2r < 0.0) || (r2r == 0.0 && r2i < 0.0);
                    }
                    kdefl = 0;
                    i = l - 1;
                    l = ilo;
                    its = 0;
                    if (i < ilo) active = false;
                } else {
                    sweep = true;
                }
            }
#ifdef ILQR_DEBUG_COUNTERS
            dbg_lap(0);
#endif
            if (sweep) {
                kdefl += 1;
#ifdef ILQR_DEBUG_COUNTERS
                dq += 1;
#endif
                const int i1 = l, i2 = i;
                double h11, h12, h21, h22;
                if (kdefl % (2 * kexsh) == 0) {
                    const double s = __builtin_fabs(A(i, i - 1)) + __builtin_fabs(A(i - 1, i - 2));
                    h11 = dat1 * s + A(i, i); h12 = dat2 * s; h21 = s; h22 = h11;
                } else if (kdefl % kexsh == 0) {
                    const double s = __builtin_fabs(A(l + 1, l)) + __builtin_fabs(A(l + 2, l + 1));
                    h11 = dat1 * s + A(l, l); h12 = dat2 * s; h21 = s; h22 = h11;
                } else {
                    h11 = A(i - 1, i - 1); h21 = A(i, i - 1); h12 = A(i - 1, i); h22 = A(i, i);
                }
                double rt1r, rt1i, rt2r, rt2i;
                {
                    const double s = __builtin_fabs(h11) + __builtin_fabs(h12) + __builtin_fabs(h21) + __builtin_fabs(h22);
                    if (s == 0.0) {
                        rt1r = 0.0; rt1i = 0.0; rt2r = 0.0; rt2i = 0.0;
                    } else {
                        h11 = h11 / s; h21 = h21 / s; h12 = h12 / s; h22 = h22 / s;
                        const double tr = (h11 + h22) / 2.0;
                        const double det = (h11 - tr) * (h22 - tr) - h12 * h21;
                        const double rtdisc = __builtin_sqrt(__builtin_fabs(det));
                        if (det >= 0.0) {
                            rt1r = tr * s; rt2r = rt1r; rt1i = rtdisc * s; rt2i = -rt1i;
                        } else {
                            rt1r = tr + rtdisc; rt2r = tr - rtdisc;
                            if (__builtin_fabs(rt1r - h22) <= __builtin_fabs(rt2r - h22)) { rt1r = rt1r * s; rt2r = rt1r; }
                            else { rt2r = rt2r * s; rt1r = rt2r; }
                            rt1i = 0.0; rt2i = 0.0;
                        }
                    }
                }
#ifdef ILQR_DEBUG_COUNTERS
                dbg_lap(1);
#endif
                double v0 = 0.0, v1 = 0.0, v2 = 0.0;
                int m;
                for (m = i - 2; m >= l; --m) {        // look for two consecutive small subdiagonal elements
                    const double hmm = A(m, m), hm1m = A(m + 1, m);
#ifdef ILQR_DEBUG_COUNTERS
                    dm += 1;
#endif
                    double h21s = __builtin_fabs(hm1m);
                    double s = __builtin_fabs(hmm - rt2r) + __builtin_fabs(rt2i) + h21s;
                    h21s = hm1m / s;
                    const double hm1m1 = A(m + 1, m + 1);
                    v0 = h21s * A(m, m + 1) + (hmm - rt1r) * ((hmm - rt2r) / s) - rt1i * (rt2i / s);
                    v1 = h21s * (hmm + hm1m1 - rt1r - rt2r);
                    v2 = h21s * A(m + 2, m + 1);
                    s = __builtin_fabs(v0) + __builtin_fabs(v1) + __builtin_fabs(v2);
                    v0 = v0 / s; v1 = v1 / s; v2 = v2 / s;
                    if (m == l) break;
                    const double h00 = __builtin_fabs(A(m - 1, m - 1)), h11a = __builtin_fabs(hmm), h22a = __builtin_fabs(hm1m1);
                    if (__builtin_fabs(A(m, m - 1)) * (__builtin_fabs(v1) + __builtin_fabs(v2)) <=
                        ulp * __builtin_fabs(v0) * (h00 + h11a + h22a))
                        break;
                }
#ifdef ILQR_DEBUG_COUNTERS
                dbg_lap(2);
#endif
                for (k = m; k <= i - 1; ++k) {        // double-shift QR step
                    const int nr = (3 < i - k + 1) ? 3 : i - k + 1;
#ifdef ILQR_DEBUG_COUNTERS
                    dk += 1;
#endif
                    if (k > m) {
                        v0 = A(k, k - 1); v1 = A(k + 1, k - 1);
                        if (nr == 3) v2 = A(k + 2, k - 1);
                    }
                    double t1;
                    {
                        // DLARFG(nr, v0, [v1, v2]): the reflector's tail stays in registers
                        double tail[2] = {v1, v2};
                        larfg_small(nr, v0, tail, t1);
                        v1 = tail[0]; v2 = tail[1];
                    }
                    if (k > m) {
                        A(k, k - 1) = v0;
                        A(k + 1, k - 1) = 0.0;
                        if (k < i - 1) A(k + 2, k - 1) = 0.0;
                    } else if (m > l) {
                        A(k, k - 1) = A(k, k - 1) * (1.0 - t1);
                    }
                    const double t2 = t1 * v1;
                    if (nr == 3) {
                        const double t3 = t1 * v2;
                        for (int j = k; j <= i2; ++j) {
                            const double a0 = A(k, j), a1 = A(k + 1, j), a2 = A(k + 2, j);
                            const double sum = a0 + v1 * a1 + v2 * a2;
                            A(k, j) = a0 - sum * t1;
                            A(k + 1, j) = a1 - sum * t2;
                            A(k + 2, j) = a2 - sum * t3;
                        }
                        const int jend = (k + 3 < i) ? k + 3 : i;
                        for (int j = i1; j <= jend; ++j) {
                            const double a0 = A(j, k), a1 = A(j, k + 1), a2 = A(j, k + 2);
                            const double sum = a0 + v1 * a1 + v2 * a2;
                            A(j, k) = a0 - sum * t1;
                            A(j, k + 1) = a1 - sum * t2;
                            A(j, k + 2) = a2 - sum * t3;
                        }
                    } else if (nr == 2) {
                        for (int j = k; j <= i2; ++j) {
                            const double a0 = A(k, j), a1 = A(k + 1, j);
                            const double sum = a0 + v1 * a1;
                            A(k, j) = a0 - sum * t1;
                            A(k + 1, j) = a1 - sum * t2;
                        }
                        for (int j = i1; j <= i; ++j) {
                            const double a0 = A(j, k), a1 = A(j, k + 1);
                            const double sum = a0 + v1 * a1;
                            A(j, k) = a0 - sum * t1;
                            A(j, k + 1) = a1 - sum * t2;
                        }
                    }
                }
                its += 1;
#ifdef ILQR_DEBUG_COUNTERS
                dbg_lap(3);
#endif
            }
        }
#ifdef ILQR_DEBUG_COUNTERS
        if ((__ffsll((long long)__ballot(true)) - 1) == (int)(threadIdx.x & 63))
            for (int q = 0; q < 4; ++q) atomicAdd(&g_eig_trips[8 + q], dbg_acc[q]);
#endif
#ifdef ILQR_DEBUG_COUNTERS
        atomicAdd(&g_eig_trips[0], 1ull); atomicAdd(&g_eig_trips[1], (unsigned long long)dq);
        atomicAdd(&g_eig_trips[2], (unsigned long long)dd); atomicAdd(&g_eig_trips[3], (unsigned long long)dm);
        atomicAdd(&g_eig_trips[5], (unsigned long long)dk);
        {
            unsigned mxq = dq;
            for (int d = 32; d >= 1; d /= 2) { const unsigned o = __shfl_xor(mxq, d); mxq = o > mxq ? o : mxq; }
            if ((__ffsll((long long)__ballot(true)) - 1) == (int)(threadIdx.x & 63)) atomicAdd(&g_eig_trips[6], (unsigned long long)mxq);
        }
#endif
        return !failed;
    }

    // DLARFG for the 2- and 3-vectors of the QR sweep, tail in registers
    __device__ __forceinline__ static void larfg_small(int n, double& alpha, double* x, double& tau)
    {
#pragma clang fp contract(off)
        double xnorm;
        if (n == 2) xnorm = __builtin_fabs(x[0]);
        else {
            const double t[2] = {x[0], x[1]};
            xnorm = nrm2_small<2>(t);
        }
        if (n <= 1 || xnorm == 0.0) { tau = 0.0; return; }
        double beta = -sgn(lapy2(alpha, xnorm), alpha);
        const double safmin = SAFMIN / EPS;
        int knt = 0;
        if (__builtin_fabs(beta) < safmin) {
            const double rsafmn = 1.0 / safmin;
            do {
                knt += 1;
                x[0] *= rsafmn;
                if (n == 3) x[1] *= rsafmn;
                beta *= rsafmn;
                alpha *= rsafmn;
            } while (__builtin_fabs(beta) < safmin && knt < 20);
            if (n == 2) xnorm = __builtin_fabs(x[0]);
            else {
                const double t[2] = {x[0], x[1]};
                xnorm = nrm2_small<2>(t);
            }
            beta = -sgn(lapy2(alpha, xnorm), alpha);
        }
        tau = (beta - alpha) / beta;
        const double sc = 1.0 / (alpha - beta);
        x[0] *= sc;
        if (n == 3) x[1] *= sc;
        for (int j = 0; j < knt; ++j) beta *= safmin;
        alpha = beta;
    }

    // DLANV2, eigenvalues only
    __device__ static void lanv2(double a, double b, double c, double d, double& rt1r, double& rt1i, double& rt2r, double& rt2i)
    {
#pragma clang fp contract(off)
        const double multpl = 4.0, eps = ULP;
        const double safmn2 = 0x1p-485, safmx2 = 0x1p485;      // base ** int(log(safmin / eps) / log(base) / 2) and its inverse
        if (c == 0.0) {
        } else if (b == 0.0) {
            const double temp = d;
            d = a; a = temp; b = -c; c = 0.0;
        } else if ((a - d) == 0.0 && sgn(1.0, b) != sgn(1.0, c)) {
        } else {
            double temp = a - d;
            double p = 0.5 * temp;
            const double bcmax = mx(__builtin_fabs(b), __builtin_fabs(c));
            const double bcmis = mn(__builtin_fabs(b), __builtin_fabs(c)) * sgn(1.0, b) * sgn(1.0, c);
            double scale = mx(__builtin_fabs(p), bcmax);
            double z = (p / scale) * p + (bcmax / scale) * bcmis;
            if (z >= multpl * eps) {
                z = p + sgn(__builtin_sqrt(scale) * __builtin_sqrt(z), p);
                a = d + z;
                d = d - (bcmax / z) * bcmis;
                b = b - c; c = 0.0;
            } else {
                int count = 0;
                double sigma = b + c;
                for (;;) {
                    count += 1;
                    scale = mx(__builtin_fabs(temp), __builtin_fabs(sigma));
                    if (scale >= safmx2) {
                        sigma *= safmn2; temp *= safmn2;
                        if (count <= 20) continue;
                    } else if (scale <= safmn2) {
                        sigma *= safmx2; temp *= safmx2;
                        if (count <= 20) continue;
                    } else {
                        p = 0.5 * temp;
                        double tau = lapy2(sigma, temp);
                        const double cs = __builtin_sqrt(0.5 * (1.0 + __builtin_fabs(sigma) / tau));
                        const double sn = -(p / (tau * cs)) * sgn(1.0, sigma);
                        const double aa = a * cs + b * sn, bb = -a * sn + b * cs;
                        const double cc = c * cs + d * sn, dd = -c * sn + d * cs;
                        a = aa * cs + cc * sn; b = bb * cs + dd * sn;
                        c = -aa * sn + cc * cs; d = -bb * sn + dd * cs;
                        temp = 0.5 * (a + d);
                        a = temp; d = temp;
                        if (c != 0.0) {
                            if (b != 0.0) {
                                if (sgn(1.0, b) == sgn(1.0, c)) {
                                    const double sab = __builtin_sqrt(__builtin_fabs(b)), sac = __builtin_sqrt(__builtin_fabs(c));
                                    p = sgn(sab * sac, c);
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
        rt1r = a; rt2r = d;
        if (c == 0.0) { rt1i = 0.0; rt2i = 0.0; }
        else { rt1i = __builtin_sqrt(__builtin_fabs(b)) * __builtin_sqrt(__builtin_fabs(c)); rt2i = -rt1i; }
    }

    // DLASCL('G'): the matrix times cto / cfrom without over- or underflow
    __device__ void lascl(double cfrom, double cto) const
    {
#pragma clang fp contract(off)
        double cfromc = cfrom, ctoc = cto;
        bool done;
        do {
            const double cfrom1 = cfromc * SAFMIN;
            double mul;
            if (cfrom1 == cfromc) {
                mul = ctoc / cfromc; done = true;
            } else {
                const double cto1 = ctoc / (1.0 / SAFMIN);
                if (cto1 == ctoc) { mul = ctoc; done = true; cfromc = 1.0; }
                else if (__builtin_fabs(cfrom1) > __builtin_fabs(ctoc) && ctoc != 0.0) { mul = SAFMIN; done = false; cfromc = cfrom1; }
                else if (__builtin_fabs(cto1) > __builtin_fabs(cfromc)) { mul = 1.0 / SAFMIN; done = false; ctoc = cto1; }
                else { mul = ctoc / cfromc; done = true; if (mul == 1.0) break; }
            }
            for (int e = 0; e < W * W; ++e) a[e * ES] *= mul;
        } while (!done);
    }

    // DGEEV('N', 'N') on the lane's matrix (destroyed), reduced to the one bit the solver needs.
    // Returns 1 = an eigenvalue compares below zero, 0 = none, -1 = NumPy raises LinAlgError (entries not finite,
    // or the QR iteration did not converge).
    __device__ int has_negative() const
    {
#pragma clang fp contract(off)
#ifdef ILQR_DEBUG_COUNTERS
        const bool dbg_lead = (__ffsll((long long)__ballot(true)) - 1) == (int)(threadIdx.x & 63);
        unsigned long long dbg_t = __builtin_readcyclecounter();
        auto dbg_lap = [&](int slot) {
            const unsigned long long now = __builtin_readcyclecounter();
            if (dbg_lead) atomicAdd(&g_eig_cycles[slot], now - dbg_t);
            dbg_t = now;
        };
        if (dbg_lead) atomicAdd(&g_eig_cycles[4], 1ull);
#endif
        double anrm = 0.0;
        bool finite = true;
        for (int e = 0; e < W * W; ++e) {
            const double t = __builtin_fabs(a[e * ES]);
            finite = finite && (t <= DBL_MAX);
            if (anrm < t) anrm = t;
        }
        if (!finite) return -1;
        const double smlnum = __builtin_sqrt(SAFMIN) / ULP, bignum = 1.0 / smlnum;
        // (the eigenvalues are only compared with zero: scaling them back by a positive factor changes nothing)
        if (anrm > 0.0 && anrm < smlnum) lascl(anrm, smlnum);
        else if (anrm > bignum) lascl(anrm, bignum);
        int ilo = 1, ihi = W;
#ifdef ILQR_DEBUG_COUNTERS
        dbg_lap(0);
#endif
        const bool plain = gebal_is_identity();
#ifdef ILQR_DEBUG_COUNTERS
        if (__ballot(!plain) && dbg_lead) atomicAdd(&g_eig_cycles[5], 1ull);
#endif
        if (!plain) {
            if (!gebal(ilo, ihi)) return -1;
        }
#ifdef ILQR_DEBUG_COUNTERS
        dbg_lap(1);
#endif
        if (ilo == 1 && ihi == W) gehd2_whole();
        else gehd2(ilo, ihi);
#ifdef ILQR_DEBUG_COUNTERS
        dbg_lap(2);
#endif
        bool neg = false;
        for (int i = 1; i <= ilo - 1; ++i) neg = neg || (A(i, i) < 0.0);
        for (int i = ihi + 1; i <= W; ++i) neg = neg || (A(i, i) < 0.0);
        const bool conv = lahqr(ilo, ihi, neg);
#ifdef ILQR_DEBUG_COUNTERS
        dbg_lap(3);
#endif
        if (!conv) return -1;
        return neg ? 1 : 0;
    }
};

}  // namespace eig
}  // namespace ilqr
