// libilqr_hip: kernels and C ABI (see include/ilqr_hip.h).
//
// Launch shape: one lane per trajectory, 64-lane workgroups (one wavefront),
// so a trajectory's whole solve runs in one lane and the 64 trajectories of a
// wavefront advance in lockstep; data-dependent trip counts (regularisation
// rounds, backtracking steps, iterations) are handled by predication.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#include "../../include/ilqr_hip.h"
#include "solver_device.hpp"

using namespace ilqr;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, const char* a = "")
{
    snprintf(g_err, sizeof(g_err), fmt, a);
    return code;
}
#define HIP_TRY(expr)                                                             \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) return fail(ILQR_E_HIP, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

struct Dev {
    ilqr_problem p;
    ilqr_outputs o;
};

constexpr int WG = 64;

// ---------------------------------------------------------------- kernels
template <class M>
__global__ __launch_bounds__(WG) void solve_kernel(Dev a, const double* __restrict__ x0, double* __restrict__ u,
                                                   double* __restrict__ x, double* __restrict__ wsD,
                                                   double* __restrict__ wsK, double* __restrict__ wsXn,
                                                   double* __restrict__ wsUn)
{
    constexpr int n = M::NDX, m = M::NU, nx = M::NX;
    const long b = (long)blockIdx.x * WG + threadIdx.x;
    if (b >= a.p.B) return;
    const long st = a.p.stride;
    const int N = a.p.N, S = N - 1;
    const double* par = a.p.params;
    const Lane X{x + b, st}, U{u + b, st}, D{wsD + b, st}, K{wsK + b, st}, Xn{wsXn + b, st}, Un{wsUn + b, st};
    const CLane cX{x + b, st}, cU{u + b, st}, cD{wsD + b, st}, cK{wsK + b, st}, cXn{wsXn + b, st}, cUn{wsUn + b, st};

    double xs[nx];
    for (int i = 0; i < nx; ++i) xs[i] = x0[(long)i * st + b];
    double J = rollout_cost<M>(par, xs, cU, X, S);           // iLQRController.py:27-34
    const double J0 = J;

    int iters = 0, status = ILQR_EXIT_DTOL, regu_rounds = 0, ls_total = 0;
    double dmax = INFINITY, maxq = 0.0;                      // d = DBL_MAX -> norm overflows to +inf (:37, :43)
    const int max_iter = a.p.max_iter, max_ls = a.p.max_ls, T = a.p.trace_len;
    const double d_tol = a.p.d_tol;
    for (;;) {
        if (!(dmax > d_tol)) { status = ILQR_EXIT_DTOL; break; }
        if (!(iters < max_iter)) { status = ILQR_EXIT_MAXITER; break; }
        iters += 1;
        double deltaJ = 0.0;
        if (!backward_pass<M>(par, cX, cU, D, K, S, deltaJ, dmax, regu_rounds, maxq)) {
            status = ILQR_EXIT_LINALG;
            break;
        }
        if (iters <= T && a.o.deltaJ_trace) a.o.deltaJ_trace[(long)(iters - 1) * st + b] = deltaJ;
        if (deltaJ < 1.0) { status = ILQR_EXIT_SMALL_DELTAJ; break; }     // :50-51
        int ls = 0;
        bool complete = false, abandon = false;
        double alpha = 1.0, Jn = 0.0;
        while (!complete && !abandon) {                                   // :60-68
            ls += 1;
            Jn = forward_pass<M>(par, cX, cU, cD, cK, alpha, Xn, Un, S);
            complete = (Jn - J <= -1e-2 * alpha * deltaJ) && (ls < max_ls);
            abandon = (Jn != Jn) || (ls >= max_ls);
            alpha *= 0.5;
        }
        if (complete) {                                                   // :69-72
            for (int i = 0; i < N * nx; ++i) X.st(i, cXn.ld(i));
            for (int i = 0; i < S * m; ++i) U.st(i, cUn.ld(i));
        }
        if (iters <= T) {
            const long q = (long)(iters - 1) * st + b;
            if (a.o.ls_evals) a.o.ls_evals[q] = ls;
            if (a.o.accepted) a.o.accepted[q] = complete ? 1 : 0;
            if (a.o.J_trace) a.o.J_trace[q] = Jn;
        }
        ls_total += ls;
        J = Jn;                                                           // :79
    }
    if (a.o.J0) a.o.J0[b] = J0;
    if (a.o.J) a.o.J[b] = J;
    if (a.o.dmax) a.o.dmax[b] = dmax;
    if (a.o.max_abs_q) a.o.max_abs_q[b] = maxq;
    if (a.o.iters) a.o.iters[b] = iters;
    if (a.o.status) a.o.status[b] = status;
    if (a.o.regu_rounds) a.o.regu_rounds[b] = regu_rounds;
    if (a.o.ls_total) a.o.ls_total[b] = ls_total;
    (void)n;
}

template <class M>
__global__ __launch_bounds__(WG) void rollout_kernel(Dev a, const double* __restrict__ x0,
                                                     const double* __restrict__ u, double* __restrict__ x,
                                                     double* __restrict__ J)
{
    constexpr int nx = M::NX;
    const long b = (long)blockIdx.x * WG + threadIdx.x;
    if (b >= a.p.B) return;
    const long st = a.p.stride;
    double xs[nx];
    for (int i = 0; i < nx; ++i) xs[i] = x0[(long)i * st + b];
    double c = rollout_cost<M>(a.p.params, xs, CLane{u + b, st}, Lane{x + b, st}, a.p.N - 1);
    if (J) J[b] = c;
}

struct EvalOut {
    double *xn, *cs, *cf, *l_x, *l_u, *l_xx, *l_ux, *l_uu, *f_x, *f_u, *V_x, *V_xx;
};

template <class M>
__global__ __launch_bounds__(WG) void eval_kernel(Dev a, const double* __restrict__ xs, const double* __restrict__ us,
                                                  EvalOut o)
{
    constexpr int n = M::NDX, m = M::NU, nx = M::NX;
    const long b = (long)blockIdx.x * WG + threadIdx.x;
    if (b >= a.p.B) return;
    const long st = a.p.stride;
    const double* par = a.p.params;
    double x[nx], u[m];
    for (int i = 0; i < nx; ++i) x[i] = xs[(long)i * st + b];
    for (int i = 0; i < m; ++i) u[i] = us[(long)i * st + b];
    if (o.xn) {
        double xn[nx];
        M::dynamics(par, x, u, xn);
        for (int i = 0; i < nx; ++i) o.xn[(long)i * st + b] = xn[i];
    }
    if (o.cs) o.cs[b] = M::cost_stage(par, x, u);
    if (o.cf) o.cf[b] = M::cost_final(par, x);
    if (o.l_x || o.l_u || o.l_xx || o.l_ux || o.l_uu || o.f_x || o.f_u) {
        double lx[n], lu[m], lxx[n * n], lux[m * n], luu[m * m], fx[n * n], fu[n * m];
        M::stage(par, x, u, lx, lu, lxx, lux, luu, fx, fu);
        auto put = [&](double* dst, const double* src, int cnt) {
            if (dst)
                for (int i = 0; i < cnt; ++i) dst[(long)i * st + b] = src[i];
        };
        put(o.l_x, lx, n); put(o.l_u, lu, m); put(o.l_xx, lxx, n * n); put(o.l_ux, lux, m * n);
        put(o.l_uu, luu, m * m); put(o.f_x, fx, n * n); put(o.f_u, fu, n * m);
    }
    if (o.V_x || o.V_xx) {
        double Vx[n], Vxx[n * n];
        M::final(par, x, Vx, Vxx);
        if (o.V_x)
            for (int i = 0; i < n; ++i) o.V_x[(long)i * st + b] = Vx[i];
        if (o.V_xx)
            for (int i = 0; i < n * n; ++i) o.V_xx[(long)i * st + b] = Vxx[i];
    }
}

// x0 <- f(x0, u[0]); u[k] <- u[k+1] for k < S-1
template <class M>
__global__ __launch_bounds__(WG) void mpc_advance_kernel(Dev a, double* __restrict__ x0, double* __restrict__ u)
{
    constexpr int m = M::NU, nx = M::NX;
    const long b = (long)blockIdx.x * WG + threadIdx.x;
    if (b >= a.p.B) return;
    const long st = a.p.stride;
    const int S = a.p.N - 1;
    double x[nx], xn[nx], u0[m];
    for (int i = 0; i < nx; ++i) x[i] = x0[(long)i * st + b];
    for (int i = 0; i < m; ++i) u0[i] = u[(long)i * st + b];
    M::dynamics(a.p.params, x, u0, xn);
    for (int i = 0; i < nx; ++i) x0[(long)i * st + b] = xn[i];
    for (int k = 0; k < S - 1; ++k)
        for (int i = 0; i < m; ++i) u[(long)(k * m + i) * st + b] = u[(long)((k + 1) * m + i) * st + b];
}

// ---------------------------------------------------------------- host side
struct ModelDims {
    int nx, nu, ndx, npar;
};
bool model_dims(int id, ModelDims& d)
{
    switch (id) {
    case ILQR_MODEL_CAR: d = {CarModel::NX, CarModel::NU, CarModel::NDX, CarModel::NPAR}; return true;
    case ILQR_MODEL_ACROBOT: d = {AcrobotModel::NX, AcrobotModel::NU, AcrobotModel::NDX, AcrobotModel::NPAR}; return true;
    case ILQR_MODEL_QUAD_EA: d = {QuadEAModel::NX, QuadEAModel::NU, QuadEAModel::NDX, QuadEAModel::NPAR}; return true;
    case ILQR_MODEL_QUAD_QUAT: d = {QuadQuatModel::NX, QuadQuatModel::NU, QuadQuatModel::NDX, QuadQuatModel::NPAR}; return true;
    }
    return false;
}

int check_problem(const ilqr_problem* p, ModelDims& d)
{
    if (!p) return fail(ILQR_E_BADARG, "problem is NULL");
    if (!model_dims(p->model_id, d)) return fail(ILQR_E_BADARG, "unknown model_id");
    if (p->N < 2) return fail(ILQR_E_BADARG, "N must be >= 2");
    if (p->B < 1) return fail(ILQR_E_BADARG, "B must be >= 1");
    if (p->stride < p->B) return fail(ILQR_E_BADARG, "stride must be >= B");
    if (!p->params) return fail(ILQR_E_BADARG, "params is NULL");
    if (p->max_iter < 0 || p->max_ls < 0 || p->trace_len < 0) return fail(ILQR_E_BADARG, "negative count");
    return ILQR_OK;
}

struct WsLayout {
    size_t d, k, xn, un, total;
};
WsLayout ws_layout(const ilqr_problem* p, const ModelDims& d)
{
    const size_t st = (size_t)p->stride, S = (size_t)p->N - 1;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    WsLayout w;
    size_t off = 0;
    w.d = off; off += al(S * d.nu * st * 8);
    w.k = off; off += al(S * d.nu * d.ndx * st * 8);
    w.xn = off; off += al((S + 1) * d.nx * st * 8);
    w.un = off; off += al(S * d.nu * st * 8);
    w.total = off;
    return w;
}

template <class F>
int dispatch(int model_id, F&& f)
{
    switch (model_id) {
    case ILQR_MODEL_CAR: return f(CarModel{});
    case ILQR_MODEL_ACROBOT: return f(AcrobotModel{});
    case ILQR_MODEL_QUAD_EA: return f(QuadEAModel{});
    case ILQR_MODEL_QUAD_QUAT: return f(QuadQuatModel{});
    }
    return fail(ILQR_E_BADARG, "unknown model_id");
}

}  // namespace

extern "C" {

const char* ilqr_last_error(void) { return g_err; }
const char* ilqr_version(void) { return "ilqr_hip 0.1 (gfx950)"; }

int ilqr_model_info(int32_t model_id, int32_t* nx, int32_t* nu, int32_t* ndx, int32_t* npar)
{
    ModelDims d;
    if (!model_dims(model_id, d)) return fail(ILQR_E_BADARG, "unknown model_id");
    if (nx) *nx = d.nx;
    if (nu) *nu = d.nu;
    if (ndx) *ndx = d.ndx;
    if (npar) *npar = d.npar;
    return ILQR_OK;
}

size_t ilqr_workspace_bytes(const ilqr_problem* p)
{
    ModelDims d;
    if (check_problem(p, d) != ILQR_OK) return 0;
    return ws_layout(p, d).total;
}

int ilqr_solve_batch(const ilqr_problem* p, const double* x0, double* u, double* x, const ilqr_outputs* out,
                     void* workspace, size_t workspace_bytes, void* stream)
{
    ModelDims d;
    int rc = check_problem(p, d);
    if (rc) return rc;
    if (!x0 || !u || !x) return fail(ILQR_E_BADARG, "x0, u and x must be non-NULL");
    WsLayout w = ws_layout(p, d);
    if (!workspace || workspace_bytes < w.total) return fail(ILQR_E_WORKSPACE, "workspace missing or too small");
    Dev a;
    a.p = *p;
    if (out) a.o = *out; else memset(&a.o, 0, sizeof(a.o));
    char* ws = (char*)workspace;
    dim3 grid((unsigned)((p->B + WG - 1) / WG)), block(WG);
    hipStream_t s = (hipStream_t)stream;
    return dispatch(p->model_id, [&](auto M) {
        using Mt = decltype(M);
        hipLaunchKernelGGL(solve_kernel<Mt>, grid, block, 0, s, a, x0, u, x, (double*)(ws + w.d), (double*)(ws + w.k),
                           (double*)(ws + w.xn), (double*)(ws + w.un));
        HIP_TRY(hipGetLastError());
        return (int)ILQR_OK;
    });
}

int ilqr_rollout_cost(const ilqr_problem* p, const double* x0, const double* u, double* x, double* J, void* stream)
{
    ModelDims d;
    int rc = check_problem(p, d);
    if (rc) return rc;
    if (!x0 || !u || !x) return fail(ILQR_E_BADARG, "x0, u and x must be non-NULL");
    Dev a;
    a.p = *p;
    memset(&a.o, 0, sizeof(a.o));
    dim3 grid((unsigned)((p->B + WG - 1) / WG)), block(WG);
    hipStream_t s = (hipStream_t)stream;
    return dispatch(p->model_id, [&](auto M) {
        using Mt = decltype(M);
        hipLaunchKernelGGL(rollout_kernel<Mt>, grid, block, 0, s, a, x0, u, x, J);
        HIP_TRY(hipGetLastError());
        return (int)ILQR_OK;
    });
}

int ilqr_model_eval(const ilqr_problem* p, const double* xs, const double* us, double* xn, double* cs, double* cf,
                    double* l_x, double* l_u, double* l_xx, double* l_ux, double* l_uu, double* f_x, double* f_u,
                    double* V_x, double* V_xx, void* stream)
{
    ModelDims d;
    int rc = check_problem(p, d);
    if (rc) return rc;
    if (!xs || !us) return fail(ILQR_E_BADARG, "xs and us must be non-NULL");
    Dev a;
    a.p = *p;
    memset(&a.o, 0, sizeof(a.o));
    EvalOut o{xn, cs, cf, l_x, l_u, l_xx, l_ux, l_uu, f_x, f_u, V_x, V_xx};
    dim3 grid((unsigned)((p->B + WG - 1) / WG)), block(WG);
    hipStream_t s = (hipStream_t)stream;
    return dispatch(p->model_id, [&](auto M) {
        using Mt = decltype(M);
        hipLaunchKernelGGL(eval_kernel<Mt>, grid, block, 0, s, a, xs, us, o);
        HIP_TRY(hipGetLastError());
        return (int)ILQR_OK;
    });
}

int ilqr_mpc_advance(const ilqr_problem* p, double* x0, double* u, void* stream)
{
    ModelDims d;
    int rc = check_problem(p, d);
    if (rc) return rc;
    if (!x0 || !u) return fail(ILQR_E_BADARG, "x0 and u must be non-NULL");
    Dev a;
    a.p = *p;
    memset(&a.o, 0, sizeof(a.o));
    dim3 grid((unsigned)((p->B + WG - 1) / WG)), block(WG);
    hipStream_t s = (hipStream_t)stream;
    return dispatch(p->model_id, [&](auto M) {
        using Mt = decltype(M);
        hipLaunchKernelGGL(mpc_advance_kernel<Mt>, grid, block, 0, s, a, x0, u);
        HIP_TRY(hipGetLastError());
        return (int)ILQR_OK;
    });
}

}  // extern "C"
