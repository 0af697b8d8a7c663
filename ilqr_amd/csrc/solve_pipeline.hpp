// Batched run_ilqr as an iteration-synchronous pipeline of gfx950 kernels.
//
// Why not one kernel that runs each trajectory's whole solve in one lane: the
// iteration count of control/iLQRController.py:43 varies from 1 to max_iter
// between trajectories, regularisation rounds (:110) and backtracking steps
// (:60) vary too, and a 64-lane wavefront runs as long as its slowest lane.
// Here every launch handles ONE phase of ONE solver iteration for all
// trajectories that are still running, and the running trajectories are kept
// physically compacted:
//
//   init      rollout + cost of the initial guess            (:27-34)
//   backward  Riccati sweep, exit tests, slot allocation      (:43-51, :98-123)
//   forward1  first line-search candidate, alpha = 1          (:60-68, :87-96)
//   level L   candidate number L (alpha = 0.5^(L-1)) for those still searching, one launch per level
//   parallel  all remaining candidates side by side, first stopping one wins (:60-68)
//
// State of a running trajectory (x, u) lives in one of two slot-indexed
// structure-of-arrays buffers; the forward pass, which has to write the new
// trajectory anyway, writes it into the slot the trajectory will occupy in the
// next iteration (slots are handed out densely with one wave-aggregated atomic
// per wavefront), so every kernel reads and writes 64 consecutive doubles per
// wave instruction although the set of running trajectories keeps shrinking.
// A trajectory that finishes writes its result to the caller's buffers at its
// own index.  Results per trajectory do not depend on the slot order.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/ilqr_hip.h"
#include "solver_device.hpp"

namespace ilqr {

constexpr int PIPE_WG = 64;

struct Pipe {
    // problem
    int N, max_iter, max_ls, trace_len, flags;
    double d_tol;
    long B, cap;               // trajectories, slot capacity (= stride of every buffer)
    const double* par;
    // caller buffers (indexed by trajectory id)
    const double* x0;
    double* u;
    double* x;
    // per-trajectory state, indexed by id (caller's output arrays or workspace)
    double *J0, *J, *dmax, *maxq;
    int32_t *iters, *status, *regu_rounds, *regu_stages, *ls_total;
    // optional traces [trace_len][cap]
    int32_t* tr_ls;
    uint8_t* tr_acc;
    double *tr_J, *tr_dJ;
    // slot-indexed buffers
    double *XA, *XB, *UA, *UB, *D, *K, *dJ;
    int32_t *idA, *idB, *ls_slot, *ls_dest, *bt, *bt2;
    int32_t* cnt;              // [0] slots handed out in B, [1] line-search entries, [1 + L] entries that go on
                               // after evaluation number L (L = 1, 2, ...)
    int count;                 // running trajectories = used slots of buffer A
    int iter;                  // 1-based number of the iteration being run
};

__device__ __forceinline__ int lane_rank(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0));
}

// Wave-aggregated append: every lane with `pred` gets a distinct index from
// *counter; one atomic per wavefront, indices consecutive in lane order.
__device__ __forceinline__ int wave_append(bool pred, int32_t* counter)
{
    const unsigned long long mask = __ballot(pred);
    if (mask == 0) return -1;
    int base = 0;
    const int leader = __ffsll((long long)mask) - 1;
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(counter, __popcll(mask));
    base = __shfl(base, leader);
    return pred ? base + lane_rank(mask) : -1;
}

template <class M>
__device__ inline void copy_traj(const CLane& Xs, const CLane& Us, const Lane& Xd, const Lane& Ud, int N)
{
    for (int i = 0; i < N * M::NX; ++i) Xd.st(i, Xs.ld(i));
    for (int i = 0; i < (N - 1) * M::NU; ++i) Ud.st(i, Us.ld(i));
}

// ---------------------------------------------------------------- init
template <class M>
__global__ __launch_bounds__(PIPE_WG) void pipe_init_kernel(Pipe a)
{
    constexpr int nx = M::NX, m = M::NU;
    const long b = (long)blockIdx.x * PIPE_WG + threadIdx.x;
    if (b >= a.B) return;
    const int S = a.N - 1;
    double xs[nx];
    for (int i = 0; i < nx; ++i) xs[i] = a.x0[(long)i * a.cap + b];
    const Lane UA{a.UA + b, a.cap};
    const CLane Uin{a.u + b, a.cap};
    for (int i = 0; i < S * m; ++i) UA.st(i, Uin.ld(i));
    const double J = rollout_cost<M>(a.par, xs, Uin, Lane{a.XA + b, a.cap}, S);     // iLQRController.py:27-34
    a.idA[b] = (int)b;
    a.J0[b] = J;
    a.J[b] = J;
    a.dmax[b] = INFINITY;      // d = DBL_MAX everywhere: its row norms overflow to +inf (:37, :43)
    a.maxq[b] = 0.0;
    a.iters[b] = 0;
    a.status[b] = ILQR_EXIT_DTOL;
    a.regu_rounds[b] = 0;
    a.regu_stages[b] = 0;
    a.ls_total[b] = 0;
}

// The loop test of :43 failed before the first iteration (d_tol is +inf/NaN or
// max_iter == 0): the initial rollout is the result.
template <class M>
__global__ __launch_bounds__(PIPE_WG) void pipe_finish_all_kernel(Pipe a, int status)
{
    const long b = (long)blockIdx.x * PIPE_WG + threadIdx.x;
    if (b >= a.B) return;
    const CLane Xs{a.XA + b, a.cap};
    const Lane Xd{a.x + b, a.cap};
    for (int i = 0; i < a.N * M::NX; ++i) Xd.st(i, Xs.ld(i));
    a.status[b] = status;
}

// ---------------------------------------------------------------- backward
template <class M>
__global__ __launch_bounds__(PIPE_WG, M::BACKWARD_WAVES) void pipe_backward_kernel(Pipe a)
{
    __shared__ double lds[QPack<M::NDX, M::NU>::LDS_DOUBLES];
    const int s0 = blockIdx.x * PIPE_WG + threadIdx.x;
    const bool live = s0 < a.count;
    const int s = live ? s0 : a.count - 1;      // a lane without a trajectory shadows the last one and stores nothing
    const int S = a.N - 1;
    bool need_ls = false, cont = false;
    const int id = a.idA[s];
    const CLane X{a.XA + s, a.cap}, U{a.UA + s, a.cap};
    const int it = a.iters[id] + 1;                                                  // :46
    double deltaJ = 0.0, dmax = 0.0, maxq = a.maxq[id];
    int rr = 0, rs = 0;
    const bool ok = backward_pass<M>(a.par, X, U, Lane{a.D + s, a.cap}, Lane{a.K + s, a.cap}, S, deltaJ, dmax, rr, rs,
                                     maxq, live, lds);                               // :49
    if (live) {
        a.iters[id] = it;
        a.maxq[id] = maxq;
        a.regu_rounds[id] += rr;
        a.regu_stages[id] += rs;
        if (ok) {
            a.dmax[id] = dmax;
            if (it <= a.trace_len && a.tr_dJ) a.tr_dJ[(long)(it - 1) * a.cap + id] = deltaJ;
        }
        if (!ok) {
            a.status[id] = ILQR_EXIT_LINALG;
        } else if (deltaJ < 1.0) {                                                    // :50-51
            a.status[id] = ILQR_EXIT_SMALL_DELTAJ;
        } else {
            need_ls = true;
            a.dJ[s] = deltaJ;
            // the loop test the trajectory will meet after this iteration's line search (:43)
            cont = (dmax > a.d_tol) && (it < a.max_iter);
            if (!cont) a.status[id] = (dmax > a.d_tol) ? ILQR_EXIT_MAXITER : ILQR_EXIT_DTOL;
        }
        if (!need_ls)   // finished with the trajectory it came in with (:84)
            copy_traj<M>(X, U, Lane{a.x + id, a.cap}, Lane{a.u + id, a.cap}, a.N);
    }
    const int j = wave_append(need_ls, a.cnt + 1);
    const int ns = wave_append(cont, a.cnt + 0);
    if (need_ls) {
        a.ls_slot[j] = s;
        a.ls_dest[j] = cont ? ns : -1;
        if (cont) a.idB[ns] = id;
    }
}

// Where a line-search candidate is written: the trajectory's slot in the next
// iteration's buffer, or straight into the caller's arrays when this was its
// last iteration.
struct Dest {
    Lane X, U;
};
__device__ __forceinline__ Dest make_dest(const Pipe& a, int dest, int id)
{
    return dest >= 0 ? Dest{Lane{a.XB + dest, a.cap}, Lane{a.UB + dest, a.cap}}
                     : Dest{Lane{a.x + id, a.cap}, Lane{a.u + id, a.cap}};
}

__device__ __forceinline__ void record_search(const Pipe& a, int id, int ls, bool complete, double Jn)
{
    const int it = a.iters[id];
    if (it <= a.trace_len) {
        const long q = (long)(it - 1) * a.cap + id;
        if (a.tr_ls) a.tr_ls[q] = ls;
        if (a.tr_acc) a.tr_acc[q] = complete ? 1 : 0;
        if (a.tr_J) a.tr_J[q] = Jn;
    }
    a.ls_total[id] += ls;
    a.J[id] = Jn;                                                                     // :79
}

// ---------------------------------------------------------------- forward, alpha = 1
template <class M>
__global__ __launch_bounds__(PIPE_WG) void pipe_forward1_kernel(Pipe a)
{
    const int j = blockIdx.x * PIPE_WG + threadIdx.x;
    const int nls = a.cnt[1];
    bool need_bt = false;
    if (j < nls) {
        const int s = a.ls_slot[j], dest = a.ls_dest[j], id = a.idA[s];
        const CLane X{a.XA + s, a.cap}, U{a.UA + s, a.cap};
        const Dest d = make_dest(a, dest, id);
        const double J = a.J[id], deltaJ = a.dJ[s];
        const double Jn = forward_pass<M>(a.par, X, U, CLane{a.D + s, a.cap}, CLane{a.K + s, a.cap}, 1.0, d.X, d.U,
                                          a.N - 1);                                    // :63-64
        const bool complete = (Jn - J <= -1e-2 * 1.0 * deltaJ) && (1 < a.max_ls);      // :66
        const bool abandon = (Jn != Jn) || (1 >= a.max_ls);                            // :67
        if (complete || abandon) {
            if (!complete) copy_traj<M>(X, U, d.X, d.U, a.N);    // search abandoned: the trajectory is kept (:69-72)
            record_search(a, id, 1, complete, Jn);
        } else {
            need_bt = true;
        }
    }
    const int q = wave_append(need_bt, a.cnt + 2);
    if (need_bt) a.bt[q] = j;
}

// ---------------------------------------------------------------- one more candidate for the entries of a list
// Evaluation number `ls` (alpha = 0.5^(ls-1)) for every entry of `list_in`;
// entries whose search goes on are appended to `list_out`.  Each level is one
// launch over a compacted list, so a wavefront never waits for a trajectory
// that needs more candidates than its neighbours.
template <class M>
__global__ __launch_bounds__(PIPE_WG) void pipe_forward_level_kernel(Pipe a, const int32_t* __restrict__ list_in,
                                                                     const int32_t* __restrict__ n_in,
                                                                     int32_t* __restrict__ list_out,
                                                                     int32_t* __restrict__ n_out, int ls, double alpha)
{
    const int q = blockIdx.x * PIPE_WG + threadIdx.x;
    bool more = false;
    int j = 0;
    if (q < *n_in) {
        j = list_in[q];
        const int s = a.ls_slot[j], dest = a.ls_dest[j], id = a.idA[s];
        const CLane X{a.XA + s, a.cap}, U{a.UA + s, a.cap};
        const Dest d = make_dest(a, dest, id);
        const double J = a.J[id], deltaJ = a.dJ[s];
        const double Jn = forward_pass<M>(a.par, X, U, CLane{a.D + s, a.cap}, CLane{a.K + s, a.cap}, alpha, d.X, d.U,
                                          a.N - 1);
        const bool complete = (Jn - J <= -1e-2 * alpha * deltaJ) && (ls < a.max_ls);   // :66
        const bool abandon = (Jn != Jn) || (ls >= a.max_ls);                           // :67
        if (complete || abandon) {
            if (!complete) copy_traj<M>(X, U, d.X, d.U, a.N);
            record_search(a, id, ls, complete, Jn);
        } else {
            more = true;
        }
    }
    const int o = wave_append(more, n_out);
    if (more) list_out[o] = j;
}

// ---------------------------------------------------------------- remaining candidates side by side
// G = max_ls - first_ls + 1 lanes per trajectory, lane g evaluates line-search
// evaluation number first_ls + g (alpha = 0.5^(first_ls + g - 1)) without
// storing it.  The search result is that of the first evaluation, in order, at
// which the sequential loop would have stopped (:66-67); a wave-wide ballot
// finds it, and the lane that owns it re-runs its forward pass to write the
// accepted trajectory.
template <class M>
__global__ __launch_bounds__(PIPE_WG) void pipe_backtrack_parallel_kernel(Pipe a, const int32_t* __restrict__ list_in,
                                                                          const int32_t* __restrict__ n_in,
                                                                          int first_ls)
{
    const int G = a.max_ls - first_ls + 1;         // candidates per trajectory, 1 <= G <= 64
    const int per_wave = 64 / G;
    const int lane = threadIdx.x & 63;
    const int grp = lane / G, g = lane - grp * G;
    const int q = blockIdx.x * per_wave + grp;
    const bool live = grp < per_wave && q < *n_in;
    int s = 0, dest = -1, id = 0;
    double J = 0.0, deltaJ = 0.0, Jn = 0.0, alpha = 1.0;
    bool stop = false, complete = false;
    const int ls = first_ls + g;
    if (live) {
        const int j = list_in[q];
        s = a.ls_slot[j]; dest = a.ls_dest[j]; id = a.idA[s];
        J = a.J[id]; deltaJ = a.dJ[s];
        for (int i = 1; i < ls; ++i) alpha *= 0.5;
        Jn = forward_cost<M>(a.par, CLane{a.XA + s, a.cap}, CLane{a.UA + s, a.cap}, CLane{a.D + s, a.cap},
                             CLane{a.K + s, a.cap}, alpha, a.N - 1);
        complete = (Jn - J <= -1e-2 * alpha * deltaJ) && (ls < a.max_ls);
        stop = complete || (Jn != Jn) || (ls >= a.max_ls);
    }
    const unsigned long long stops = __ballot(stop);
    if (!live) return;
    const unsigned long long gmask = ((G >= 64) ? ~0ull : ((1ull << G) - 1ull)) << (grp * G);
    const int first = __ffsll((long long)(stops & gmask)) - 1;     // always exists: the last candidate stops
    if (lane != first) return;
    const CLane X{a.XA + s, a.cap}, U{a.UA + s, a.cap};
    const Dest d = make_dest(a, dest, id);
    if (complete)
        forward_pass<M>(a.par, X, U, CLane{a.D + s, a.cap}, CLane{a.K + s, a.cap}, alpha, d.X, d.U, a.N - 1);
    else
        copy_traj<M>(X, U, d.X, d.U, a.N);
    record_search(a, id, ls, complete, Jn);
}

}  // namespace ilqr
