// libilqr_hip: kernels and C ABI (see include/ilqr_hip.h).
//
// Launch shape: one lane per trajectory, 64-lane workgroups (one wavefront).
// The batched solve is a pipeline of per-iteration kernels (solve_pipeline.hpp).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define ILQR_MAIN_TU
#include "pipeline_host.hpp"
#include "eig_device.hpp"

using namespace ilqr;

namespace {
thread_local char g_err[512] = "";
}
namespace ilqr {
int fail(int code, const char* fmt, const char* a)
{
    snprintf(g_err, sizeof(g_err), fmt, a);
    return code;
}
}  // namespace ilqr

namespace {

struct Dev {
    ilqr_problem p;
    ilqr_outputs o;
};

constexpr int WG = 64;

// the parameter block of trajectory b as the problem describes it (one block for all, or one per trajectory)
template <class M>
__device__ __forceinline__ ParCache<M> par_of(const ilqr_problem& p, long b)
{
    const bool per_traj = (p.flags & ILQR_FLAG_PARAMS_PER_TRAJECTORY) != 0;
    if constexpr (M::NPAR <= 8) return ParCache<M>(p.params, per_traj ? (long)p.stride : 1L, per_traj ? 1L : 0L, b);
    else return ParCache<M>(p.params + (per_traj ? b * M::NPAR : 0L));
}

// ---------------------------------------------------------------- kernels
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
    double c = rollout_cost<M>(par_of<M>(a.p, b).get(), xs, CLane{u + b, st}, Lane{x + b, st}, a.p.N - 1,
                               (a.p.cost_N ? a.p.cost_N : a.p.N) - 1);
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
    const ParCache<M> pc = par_of<M>(a.p, b);
    const double* par = pc.get();
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

// A = d xdot/dx, B = d xdot/du at one point per lane (on the error state for the quaternion plant)
template <class M>
__global__ __launch_bounds__(WG) void lincont_kernel(Dev a, const double* __restrict__ xs, const double* __restrict__ us,
                                                     double* __restrict__ A, double* __restrict__ B)
{
    if constexpr (M::HAS_CONTINUOUS) {
        constexpr int n = M::NDX, m = M::NU, nx = M::NX;
        const long b = (long)blockIdx.x * WG + threadIdx.x;
        if (b >= a.p.B) return;
        const long st = a.p.stride;
        double x[nx], u[m], Ac[n * n], Bc[n * m];
        for (int i = 0; i < nx; ++i) x[i] = xs[(long)i * st + b];
        for (int i = 0; i < m; ++i) u[i] = us[(long)i * st + b];
        M::linearize_continuous(par_of<M>(a.p, b).get(), x, u, Ac, Bc);
        if (A)
            for (int i = 0; i < n * n; ++i) A[(long)i * st + b] = Ac[i];
        if (B)
            for (int i = 0; i < n * m; ++i) B[(long)i * st + b] = Bc[i];
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
    M::dynamics(par_of<M>(a.p, b).get(), x, u0, xn);
    for (int i = 0; i < nx; ++i) x0[(long)i * st + b] = xn[i];
    for (int k = 0; k < S - 1; ++k)
        for (int i = 0; i < m; ++i) u[(long)(k * m + i) * st + b] = u[(long)((k + 1) * m + i) * st + b];
}

// np.any(np.linalg.eigvals(Q) < 0) for one w x w matrix per lane: Q is row-major, entry e of matrix b at Q[e * stride + b]
template <int W>
__global__ __launch_bounds__(WG) void eig_negative_kernel(const double* __restrict__ Q, long B, long stride,
                                                          int32_t* __restrict__ verdict)
{
    __shared__ double lds[(W * W + 2 * W) * WG];
    const long b = (long)blockIdx.x * WG + threadIdx.x;
    const int lane = threadIdx.x & 63;
    if (b >= B) return;
    const eig::Lane<W, WG> m{lds + lane, lds + W * W * WG + lane};
    for (int i = 0; i < W; ++i)
        for (int j = 0; j < W; ++j) m.A(i + 1, j + 1) = Q[(long)(i * W + j) * stride + b];
    verdict[b] = m.has_negative();
}

// ---------------------------------------------------------------- host side
struct ModelDims {
    int nx, nu, ndx, npar;
};
bool model_dims(int id, ModelDims& d)
{
    switch (id) {
#ifdef ILQR_USER_MODEL
    case ILQR_MODEL_USER: d = {UserModel::NX, UserModel::NU, UserModel::NDX, UserModel::NPAR}; return true;
#endif
#ifndef ILQR_ONLY_USER
    case ILQR_MODEL_CAR: d = {CarModel::NX, CarModel::NU, CarModel::NDX, CarModel::NPAR}; return true;
    case ILQR_MODEL_ACROBOT: d = {AcrobotModel::NX, AcrobotModel::NU, AcrobotModel::NDX, AcrobotModel::NPAR}; return true;
    case ILQR_MODEL_QUAD_EA: d = {QuadEAModel::NX, QuadEAModel::NU, QuadEAModel::NDX, QuadEAModel::NPAR}; return true;
    case ILQR_MODEL_QUAD_QUAT: d = {QuadQuatModel::NX, QuadQuatModel::NU, QuadQuatModel::NDX, QuadQuatModel::NPAR}; return true;
#endif
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
    if (p->cost_N < 0 || p->cost_N > p->N) return fail(ILQR_E_BADARG, "cost_N must be 0 (= N) or within 1..N");
    if (p->reserved_ != 0) return fail(ILQR_E_BADARG, "ilqr_problem.reserved_ must be 0 (a caller built against another header?)");
    if ((p->flags & ILQR_FLAG_LS_SEQUENTIAL) && p->max_ls > 60)
        return fail(ILQR_E_BADARG, "ILQR_FLAG_LS_SEQUENTIAL supports max_ls <= 60");
    return ILQR_OK;
}

struct WsLayout {
    size_t TA, TB, TOUT, G, Gscratch, dJ, J0, J, dmax, maxq, iters, status, regu_rounds, regu_stages, ls_total, last_search, idA, idB,
        ls_slot, ls_dest, bt, bt2, ls_from, pred, rr_it, rs_it, dj_it, replayed, dirty, defer, win_ls, win_J, cnt, total;
};
WsLayout ws_layout(const ilqr_problem* p, const ModelDims& d)
{
    const size_t cap = (size_t)p->stride, S = (size_t)p->N - 1, N = (size_t)p->N;
    const size_t RT = (size_t)(d.nx + d.nu + 7) / 8 * 8, RG = (size_t)(d.nu + d.nu * d.ndx + 1) / 2 * 2;   // Rec<M>
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
    WsLayout w;
    w.TA = take(N * RT * cap * 8); w.TB = take(N * RT * cap * 8); w.TOUT = take(N * RT * cap * 8);
    w.G = take(S * RG * cap * 8);
    w.Gscratch = take(S * RG * 8);
    w.dJ = take(cap * 8);
    w.J0 = take(cap * 8); w.J = take(cap * 8); w.dmax = take(cap * 8); w.maxq = take(cap * 8);
    w.iters = take(cap * 4); w.status = take(cap * 4); w.regu_rounds = take(cap * 4); w.regu_stages = take(cap * 4);
    w.ls_total = take(cap * 4); w.last_search = take(cap * 4);
    w.idA = take(cap * 4); w.idB = take(cap * 4); w.ls_slot = take(cap * 4); w.ls_dest = take(cap * 4);
    w.bt = take(cap * 4); w.bt2 = take(cap * 4);
    w.ls_from = take(cap * 4); w.pred = take(cap * 4);
    w.rr_it = take(cap * 4); w.rs_it = take(cap * 4); w.dj_it = take(cap * 8); w.replayed = take(cap * 4); w.dirty = take(cap * 4); w.defer = take(cap * 4);
    w.win_ls = take(cap * 4); w.win_J = take(cap * 8);
    w.cnt = take(64 * 4 * 4);      // one block of counters per part (MAX_PARTS)
    w.total = off;
    return w;
}

thread_local ilqr_timing g_timing;
thread_local bool g_timing_valid = false;

// The same pipe restricted to trajectories [lo, lo + n): every per-trajectory and per-slot pointer moves by lo.
template <class M>
Pipe part_of(Pipe a, long lo, long n, int part)
{
    a.B = n;
    a.x0 += lo; a.u += lo; a.x += lo;
    a.par += (M::NPAR > 8) ? lo * a.par_block : lo * a.par_pt;
    a.J0 += lo; a.J += lo; a.dmax += lo; a.maxq += lo;
    a.iters += lo; a.status += lo; a.regu_rounds += lo; a.regu_stages += lo; a.ls_total += lo; a.last_search += lo;
    if (a.tr_ls) a.tr_ls += lo;
    if (a.tr_acc) a.tr_acc += lo;
    if (a.tr_J) a.tr_J += lo;
    if (a.tr_dJ) a.tr_dJ += lo;
    a.TA += lo * Rec<M>::RT; a.TB += lo * Rec<M>::RT; a.TOUT += lo * Rec<M>::RT; a.G += lo * Rec<M>::RG; a.dJ += lo;
    a.idA += lo; a.idB += lo; a.ls_slot += lo; a.ls_dest += lo; a.bt += lo; a.bt2 += lo;
    a.ls_from += lo; a.pred += lo; a.rr_it += lo; a.rs_it += lo; a.dj_it += lo; a.replayed += lo; a.dirty += lo; a.defer += lo; a.win_ls += lo; a.win_J += lo;
    a.cnt += 64 * part;
    return a;
}

constexpr int MAX_PARTS = 4;
thread_local std::map<int, std::vector<hipStream_t>> g_part_streams;    // per calling thread and device

// `min_part`: the fewest trajectories a part may hold.
int parts_for(const ilqr_problem* p, long min_part)
{
    if (p->flags & ILQR_FLAG_SINGLE_STREAM) return 1;
    int want = 3;
    if (const char* e = getenv("ILQR_HIP_PARTS")) want = atoi(e);
    if (want < 1) want = 1;
    if (want > MAX_PARTS) want = MAX_PARTS;
    while (want > 1 && p->B / want < min_part) --want;
    return want;
}

// plants whose pipeline kernels also exist in the block-per-trajectory variant (large blocks)
template <class M>
constexpr bool PT_OK = (M::NPAR > 8);

// run_ilqr for the whole batch: see solve_pipeline.hpp.  A large batch is cut into parts that run the whole
// pipeline independently, each on its own stream and driven by its own host thread: trajectories never
// interact, and while one part is in a compute-bound phase (backward pass) another is in a memory-bound one
// (line search), or waiting for its host thread.  Work on `st` before the call is waited for, and `st` waits
// for all parts before anything enqueued after the call runs.
template <class M>
int run_pipeline(const ilqr_problem* p, const double* x0, double* u, double* x, const ilqr_outputs* out, char* ws,
                 const WsLayout& w, hipStream_t st)
{
    static_assert(Rec<M>::RT == (M::NX + M::NU + 7) / 8 * 8 && Rec<M>::RG == (M::NU + M::NU * M::NDX + 1) / 2 * 2, "");
    Pipe a;
    memset(&a, 0, sizeof(a));
    a.SC = (p->cost_N ? p->cost_N : p->N) - 1;
    a.N = p->N; a.max_iter = p->max_iter; a.max_ls = p->max_ls; a.trace_len = p->trace_len; a.flags = p->flags;
    a.d_tol = p->d_tol; a.B = p->B; a.cap = p->stride; a.par = p->params;
    const bool per_traj = (p->flags & ILQR_FLAG_PARAMS_PER_TRAJECTORY) != 0;
    a.par_is = per_traj ? p->stride : 1;         // small blocks: strided like the other arrays
    a.par_pt = per_traj ? 1 : 0;
    a.par_block = (per_traj && M::NPAR > 8) ? M::NPAR : 0;      // large blocks: one after the other
    const bool big_pt = a.par_block != 0;
    a.x0 = x0; a.u = u; a.x = x;
    ilqr_outputs o;
    if (out) o = *out; else memset(&o, 0, sizeof(o));
    a.J0 = o.J0 ? o.J0 : (double*)(ws + w.J0);
    a.J = o.J ? o.J : (double*)(ws + w.J);
    a.dmax = o.dmax ? o.dmax : (double*)(ws + w.dmax);
    a.maxq = o.max_abs_q ? o.max_abs_q : (double*)(ws + w.maxq);
    a.iters = o.iters ? o.iters : (int32_t*)(ws + w.iters);
    a.status = o.status ? o.status : (int32_t*)(ws + w.status);
    a.regu_rounds = o.regu_rounds ? o.regu_rounds : (int32_t*)(ws + w.regu_rounds);
    a.regu_stages = o.regu_stages ? o.regu_stages : (int32_t*)(ws + w.regu_stages);
    a.ls_total = o.ls_total ? o.ls_total : (int32_t*)(ws + w.ls_total);
    a.last_search = o.last_search ? o.last_search : (int32_t*)(ws + w.last_search);
    a.tr_ls = o.ls_evals; a.tr_acc = o.accepted; a.tr_J = o.J_trace; a.tr_dJ = o.deltaJ_trace;
    a.TA = (double*)(ws + w.TA); a.TB = (double*)(ws + w.TB); a.TOUT = (double*)(ws + w.TOUT); a.G = (double*)(ws + w.G); a.Gscratch = (double*)(ws + w.Gscratch); a.dJ = (double*)(ws + w.dJ);
    a.idA = (int32_t*)(ws + w.idA); a.idB = (int32_t*)(ws + w.idB);
    a.ls_slot = (int32_t*)(ws + w.ls_slot); a.ls_dest = (int32_t*)(ws + w.ls_dest);
    a.bt = (int32_t*)(ws + w.bt); a.bt2 = (int32_t*)(ws + w.bt2);
    a.ls_from = (int32_t*)(ws + w.ls_from); a.pred = (int32_t*)(ws + w.pred);
    a.rr_it = (int32_t*)(ws + w.rr_it); a.rs_it = (int32_t*)(ws + w.rs_it); a.dj_it = (double*)(ws + w.dj_it);
    a.replayed = o.replayed ? o.replayed : (int32_t*)(ws + w.replayed);
    a.dirty = (int32_t*)(ws + w.dirty); a.defer = (int32_t*)(ws + w.defer);
    a.win_ls = (int32_t*)(ws + w.win_ls); a.win_J = (double*)(ws + w.win_J);
    a.cnt = (int32_t*)(ws + w.cnt);

    const bool timing = (a.flags & ILQR_FLAG_TIMING) != 0;
    // A part should still fill the GPU in the first iterations.  One lane per trajectory: 512 waves; the forward kernels
    // keep two waves per SIMD, and thinner parts of concurrent streams were measured to land on the same SIMDs (acrobot,
    // 65 536 trajectories: 50.1 ms in three parts, 47.0 ms in two).
    const int K = parts_for(p, M::TEAM_BACKWARD ? 16384 : 32768);
    a.parts = K;
    ilqr_timing total;
    memset(&total, 0, sizeof(total));
    if (K == 1) {
        const int rc = big_pt ? run_part<M, PT_OK<M>>(a, st, &total) : run_part<M, false>(a, st, &total);
        if (rc == ILQR_OK && timing) { g_timing = total; g_timing_valid = true; }
        return rc;
    }

    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    std::vector<hipStream_t>& streams = g_part_streams[dev];
    while ((int)streams.size() < K) {
        hipStream_t s2;
        HIP_TRY(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        streams.push_back(s2);
    }
    hipEvent_t start;
    HIP_TRY(hipEventCreateWithFlags(&start, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(start, st));
    for (int k = 0; k < K; ++k) HIP_TRY(hipStreamWaitEvent(streams[k], start, 0));

    std::vector<int> rcs(K, ILQR_OK);
    std::vector<std::string> msgs(K);
    std::vector<ilqr_timing> tms(K);
    auto work = [&](int k) {
        if (hipSetDevice(dev) != hipSuccess) { rcs[k] = ILQR_E_HIP; msgs[k] = "hipSetDevice failed in a part's thread"; return; }
        const long lo = ((a.B * k / K) + 63) / 64 * 64, hi = (k + 1 == K) ? a.B : ((a.B * (k + 1) / K) + 63) / 64 * 64;
        rcs[k] = big_pt ? run_part<M, PT_OK<M>>(part_of<M>(a, lo, hi - lo, k), streams[k], &tms[k])
                        : run_part<M, false>(part_of<M>(a, lo, hi - lo, k), streams[k], &tms[k]);
        if (rcs[k] != ILQR_OK) msgs[k] = g_err;
    };
    std::vector<std::thread> threads;
    for (int k = 1; k < K; ++k) threads.emplace_back(work, k);
    work(0);
    for (auto& t : threads) t.join();

    int rc = ILQR_OK;
    for (int k = 0; k < K; ++k) {
        hipEvent_t done;
        HIP_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(done, streams[k]));
        HIP_TRY(hipStreamWaitEvent(st, done, 0));
        HIP_TRY(hipEventDestroy(done));
        if (rcs[k] != ILQR_OK && rc == ILQR_OK) rc = fail(rcs[k], "%s", msgs[k].c_str());
        for (int ph = 0; ph < ILQR_PHASE_COUNT; ++ph) {
            total.ms[ph] += tms[k].ms[ph];
            total.launches[ph] += tms[k].launches[ph];
        }
        if (tms[k].iterations > total.iterations) total.iterations = tms[k].iterations;
    }
    HIP_TRY(hipEventDestroy(start));
    if (rc == ILQR_OK && timing) { g_timing = total; g_timing_valid = true; }
    return rc;
}

// ---------------------------------------------------------------- built-in workspace (ilqr_solve_batch_static)
// The workspace of a solve (slot buffers, lists, counters: ws_layout) inside the library's own device memory, for
// callers that may not allocate.  The array's name is only a handle in host code: its address on a device comes
// from hipGetSymbolAddress, once per device with that device current, and host code uses that pointer from then on.
static __device__ __align__(256) unsigned char scratch[ILQR_STATIC_WORKSPACE_BYTES];
constexpr int MAX_DEVICES = 64;
std::mutex g_scratch_mutex;
void* g_scratch_ptr[MAX_DEVICES] = {};

int static_workspace(void** ptr)
{
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= MAX_DEVICES) return fail(ILQR_E_BADARG, "device index beyond the library's table");
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void* q = nullptr;
        HIP_TRY(hipGetSymbolAddress(&q, HIP_SYMBOL(scratch)));
        if (!q) return fail(ILQR_E_HIP, "hipGetSymbolAddress returned a null address");
        g_scratch_ptr[dev] = q;
    }
    *ptr = g_scratch_ptr[dev];
    return ILQR_OK;
}

template <class F>
int dispatch(int model_id, F&& f)
{
    switch (model_id) {
#ifdef ILQR_USER_MODEL
    case ILQR_MODEL_USER: return f(UserModel{});
#endif
#ifndef ILQR_ONLY_USER
    case ILQR_MODEL_CAR: return f(CarModel{});
    case ILQR_MODEL_ACROBOT: return f(AcrobotModel{});
    case ILQR_MODEL_QUAD_EA: return f(QuadEAModel{});
    case ILQR_MODEL_QUAD_QUAT: return f(QuadQuatModel{});
#endif
    }
    return fail(ILQR_E_BADARG, "model_id not built into this library");
}

}  // namespace

extern "C" {

const char* ilqr_last_error(void) { return g_err; }
const char* ilqr_version(void) { return "ilqr_hip 0.4 (gfx950)"; }
int32_t ilqr_abi_version(void) { return ILQR_ABI_VERSION; }

int ilqr_last_timing(ilqr_timing* out)
{
    if (!out) return fail(ILQR_E_BADARG, "out is NULL");
    if (!g_timing_valid) return fail(ILQR_E_BADARG, "no ilqr_solve_batch call with ILQR_FLAG_TIMING on this thread yet");
    *out = g_timing;
    return ILQR_OK;
}

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
    if (p->stride > 0x7fffffffL) return fail(ILQR_E_BADARG, "stride must fit in 31 bits");
    WsLayout w = ws_layout(p, d);
    if (!workspace || workspace_bytes < w.total) return fail(ILQR_E_WORKSPACE, "workspace missing or too small");
    hipStream_t s = (hipStream_t)stream;
    return dispatch(p->model_id, [&](auto M) {
        using Mt = decltype(M);
        return run_pipeline<Mt>(p, x0, u, x, out, (char*)workspace, w, s);
    });
}

int ilqr_solve_batch_static(const ilqr_problem* p, const double* x0, double* u, double* x, const ilqr_outputs* out,
                            void* stream)
{
    ModelDims d;
    int rc = check_problem(p, d);
    if (rc) return rc;
    if (!x0 || !u || !x) return fail(ILQR_E_BADARG, "x0, u and x must be non-NULL");
    if (p->stride > 0x7fffffffL) return fail(ILQR_E_BADARG, "stride must fit in 31 bits");
    if (p->flags & ILQR_FLAG_TIMING) return fail(ILQR_E_BADARG, "ILQR_FLAG_TIMING creates events: not available here");
    {   // the solve reads counters back and waits for the stream once per iteration: illegal on a capturing stream
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
            return fail(ILQR_E_BADARG, "the stream is being captured into a graph: the solve blocks the host once per "
                                       "iteration and cannot be captured");
    }
    const WsLayout w = ws_layout(p, d);
    if (w.total > (size_t)ILQR_STATIC_WORKSPACE_BYTES) {
        char need[64];
        snprintf(need, sizeof(need), "%zu", w.total);
        return fail(ILQR_E_WORKSPACE, "problem needs %s bytes of workspace, more than ILQR_STATIC_WORKSPACE_BYTES", need);
    }
    void* ws = nullptr;
    rc = static_workspace(&ws);
    if (rc) return rc;
    ilqr_problem q = *p;
    q.flags |= ILQR_FLAG_SINGLE_STREAM;       // no internal streams, events or host threads
    hipStream_t s = (hipStream_t)stream;
    return dispatch(q.model_id, [&](auto M) {
        using Mt = decltype(M);
        return run_pipeline<Mt>(&q, x0, u, x, out, (char*)ws, w, s);
    });
}

int ilqr_eigvals_negative(int32_t w, const double* Q, int64_t B, int64_t stride, int32_t* verdict, void* stream)
{
    if (!Q || !verdict) return fail(ILQR_E_BADARG, "Q and verdict must be non-NULL");
    if (B < 1 || stride < B) return fail(ILQR_E_BADARG, "B must be >= 1 and stride >= B");
    dim3 grid((unsigned)((B + WG - 1) / WG)), block(WG);
    hipStream_t s = (hipStream_t)stream;
    switch (w) {
#define ILQR_EIG_CASE(W_) case W_: hipLaunchKernelGGL(eig_negative_kernel<W_>, grid, block, 0, s, Q, (long)B, (long)stride, verdict); break;
    ILQR_EIG_CASE(2) ILQR_EIG_CASE(3) ILQR_EIG_CASE(4) ILQR_EIG_CASE(5) ILQR_EIG_CASE(6) ILQR_EIG_CASE(7) ILQR_EIG_CASE(8)
    ILQR_EIG_CASE(16)
#undef ILQR_EIG_CASE
    default: return fail(ILQR_E_BADARG, "ilqr_eigvals_negative is built for orders 2..8 and 16");
    }
    HIP_TRY(hipGetLastError());
    return ILQR_OK;
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

int ilqr_linearize_continuous(const ilqr_problem* p, const double* xs, const double* us, double* A, double* B, void* stream)
{
    ModelDims d;
    int rc = check_problem(p, d);
    if (rc) return rc;
    if (!xs || !us) return fail(ILQR_E_BADARG, "xs and us must be non-NULL");
    Dev a;
    a.p = *p;
    memset(&a.o, 0, sizeof(a.o));
    dim3 grid((unsigned)((p->B + WG - 1) / WG)), block(WG);
    hipStream_t s = (hipStream_t)stream;
    return dispatch(p->model_id, [&](auto M) {
        using Mt = decltype(M);
        if (!Mt::HAS_CONTINUOUS)
            return fail(ILQR_E_BADARG, "this plant states its discrete step only (no continuous-time dynamics to linearise)");
        hipLaunchKernelGGL(lincont_kernel<Mt>, grid, block, 0, s, a, xs, us, A, B);
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
