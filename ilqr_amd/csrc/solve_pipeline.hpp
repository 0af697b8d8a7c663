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
//   forward1  first line-search candidate, alpha = 1, written (:60-68, :87-96)
//   group<8>  the next 8 candidates of the rejected ones side by side, cost only
//   group<16> further candidates of those still searching, 16 at a time, cost only
//   commit    the candidate each backtracking search stopped at is written (:66-72)
//   export    finished trajectories, record layout -> the caller's strided arrays (:84)
// When few trajectories are left the launches are bound by latency; then every
// search is evaluated side by side from its first candidate by one group<8>
// launch that repeats until all have stopped (backward, group, commit).
//
// State of a running trajectory (x, u) lives in one of two slot-indexed record
// buffers (solver_device.hpp, Rec); the forward pass, which has to write the
// new trajectory anyway, writes it into the slot the trajectory will occupy in
// the next iteration (slots are handed out densely with one wave-aggregated
// atomic per wavefront), so the big kernels read and write contiguous spans
// although the set of running trajectories keeps shrinking.  A trajectory
// that finishes writes its result, still as records, to an export buffer at
// its own index, and one last coalesced pass fills the caller's arrays.
// Results per trajectory do not depend on the slot order.
//
// Line search.  The reference tries alpha = 1, 1/2, 1/4, ... one after the
// other and stops at the first candidate that meets the Armijo test (or is
// NaN, or is number max_ls).  Candidates do not depend on each other, so after
// the first one the kernels evaluate several per trajectory in neighbouring
// lanes (which share every load of the trajectory's records) and a ballot picks
// the first one, in the reference's order, at which the sequential loop would
// have stopped.  Only costs are computed there; `commit` re-runs the one
// forward pass whose result is kept, one trajectory per lane.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/ilqr_hip.h"
#include "solver_device.hpp"

namespace ilqr {

#ifndef ILQR_FWD_WAVES
#define ILQR_FWD_WAVES 1
#endif
constexpr int FWD_WAVES = ILQR_FWD_WAVES;   // waves per SIMD the forward-pass kernels are built for
constexpr int CNT_DIRTY = 62;    // index in Pipe::cnt of the number of slots handed out from the back
constexpr int CNT_DEFER = 63;    // ... of the number of trajectories a launch left to the next one (Pipe::defer)
constexpr int PIPE_WG = 64;      // one wavefront per workgroup: backward_stage exchanges data between lanes without barriers
static_assert(PIPE_WG == 64, "backward_stage assumes single-wavefront workgroups");

struct Pipe {
    // problem
    int N, max_iter, max_ls, trace_len, flags;
    int SC;                    // stage costs a trajectory's cost sums (ilqr_problem::cost_N - 1; N - 1 by default)
    double d_tol;
    long B, cap;               // trajectories, slot capacity (= stride of the caller's arrays)
    const double* par;
    long par_is, par_pt;       // small blocks: parameter i of trajectory id at par[i * par_is + id * par_pt]:
                               // (1, 0) one block for all, (cap, 1) per trajectory
    long par_block;            // large blocks: trajectory id's block starts at par + id * par_block (0: one for all)
    // caller buffers (indexed by trajectory id)
    const double* x0;
    double* u;
    double* x;
    // per-trajectory state, indexed by id (caller's output arrays or workspace)
    double *J0, *J, *dmax, *maxq;
    int32_t *iters, *status, *regu_rounds, *regu_stages, *ls_total, *last_search;
    // optional traces [trace_len][cap]
    int32_t* tr_ls;
    uint8_t* tr_acc;
    double *tr_J, *tr_dJ;
    // slot-indexed buffers
    double *TA, *TB, *G, *dJ;  // T records now / next iteration, G records, expected reduction
    double* TOUT;              // T records of finished trajectories, indexed by trajectory (export buffer)
    double* Gscratch;          // G records of one trajectory that lanes without a trajectory may write to
    int32_t *idA, *idB, *ls_slot, *ls_dest, *bt, *bt2;
    int32_t* ls_from;          // per line-search entry: number of the first evaluation still to be made
    int32_t* pred;             // per trajectory id: 1 if its last search rejected alpha = 1
    int32_t *rr_it, *rs_it;    // per trajectory id: regularisation rounds / regularised stages of its last backward pass
    double* dj_it;             // ... and that pass's expected reduction (what replay_rest repeats)
    int32_t* replayed;         // per trajectory id: iterations written by replay_rest instead of being run
    int32_t* dirty;            // per trajectory id: 1 once a test of the trajectory needed LAPACK's path (ILQR_FLAG_LAPACK_EIGVALS)
    int32_t* win_ls;           // per line-search entry: evaluation number the search stopped at (< 0: not accepted)
    double* win_J;             // ... and that candidate's cost
    int32_t* cnt;              // [0] slots handed out in B, [1] line-search entries, [2] rejected at alpha = 1,
                               // [3..] list lengths of the later stages
    int count;                 // running trajectories = used slots of buffer A
    // The slots one backward launch covers, `group` of them, `lanes` per wavefront: entry i < n_range is slot
    // s_first + i * s_step, the entries after those are the slots listed in `defer`.
    // With ILQR_FLAG_LAPACK_EIGVALS slots are handed out from both ends of the buffer: trajectories whose tests the
    // certified screen has settled so far from the front, the others ("dirty": they run LAPACK's path at most
    // stages) from the back, so that a wavefront holds one kind, and an iteration's backward pass covers three ranges:
    // the front, 64 per wavefront; the back, fewer per wavefront; and, at most four per wavefront, the trajectories
    // those two dropped and listed in `defer` because they needed a long regularisation scan (such scans run one
    // after the other in a wavefront, and a launch takes as long as its slowest wavefront).
    int s_first, s_step, group, lanes, n_range;
    // The first two of those launches are one: workgroups 0 .. back_blocks - 1 take the `back_group` slots counted down
    // from the last one, `back_lanes` per wavefront and spread (they run longer and start first); the others take the
    // front as described by the fields above.
    int back_blocks, back_group, back_lanes;
    int spread;                // entry of lane l of wavefront b: l * wavefronts + b instead of b * lanes + l, so that
                               // neighbours in a list, which tend to be alike in how long they take, meet in no wavefront
    int32_t* defer;
    int tail;                  // few trajectories left: every search is evaluated side by side from its first candidate
    int iter;                  // 1-based number of the iteration being run
    int parts;                 // parts of the batch that run concurrently (this is one of them)
};

// Entry of the slot -> trajectory tables idA / idB: the trajectory's index, or -1 - index for a slot retired after
// it was handed out (retire_slot): its records are valid, nobody works on it any more.
struct SlotId {
    int id;
    bool alive;
};
__device__ __forceinline__ SlotId slot_id(int raw) { return raw >= 0 ? SlotId{raw, true} : SlotId{-1 - raw, false}; }

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
__device__ __forceinline__ RecIn slot_T(const double* buf, long cap, int s)
{
    return RecIn{buf + (long)s * Rec<M>::RT, cap * Rec<M>::RT};
}
template <class M>
__device__ __forceinline__ RecIn slot_G(const Pipe& a, int s)
{
    return RecIn{a.G + (long)s * Rec<M>::RG, a.cap * Rec<M>::RG};
}
// A finished trajectory goes to the export buffer, record layout, at its own index; one coalesced pass
// at the end of the solve (pipe_export_kernel) turns that buffer into the caller's strided arrays.
// (Writing the caller's arrays directly would scatter 8-byte stores over the batch.)
template <class M>
__device__ __forceinline__ Dest<M> final_dest(const Pipe& a, int id)
{
    return Dest<M>{a.TOUT + (long)id * Rec<M>::RT, a.cap * Rec<M>::RT};
}
// Where a line-search candidate is written: the trajectory's slot in the next
// iteration's buffer, or the export buffer when this was its last iteration.
template <class M>
__device__ __forceinline__ Dest<M> make_dest(const Pipe& a, int dest, int id)
{
    return dest >= 0 ? Dest<M>{a.TB + (long)dest * Rec<M>::RT, a.cap * Rec<M>::RT} : final_dest<M>(a, id);
}

// The parameter block of trajectory `id` as the kernels hand it to the plant functions.  Small blocks are
// copied into registers (ParCache).  Large ones stay in memory: the kernels of those plants exist twice,
// PT = false reading the one shared block through the kernel argument (scalar loads), PT = true reading a
// block per trajectory.
template <class M, bool PT>
__device__ __forceinline__ ParCache<M> params_of(const Pipe& a, long id)
{
    if constexpr (M::NPAR <= 8) return ParCache<M>(a.par, a.par_is, a.par_pt, id);
    else if constexpr (PT) return ParCache<M>(a.par + id * a.par_block);
    else return ParCache<M>(a.par);
}

// ---------------------------------------------------------------- init
template <class M, bool PT = false>
__global__ __launch_bounds__(PIPE_WG) void pipe_init_kernel(Pipe a)
{
    constexpr int nx = M::NX;
    const long b = (long)blockIdx.x * PIPE_WG + threadIdx.x;
    if (b >= a.B) return;
    const int S = a.N - 1;
    double xs[nx];
    for (int i = 0; i < nx; ++i) xs[i] = a.x0[(long)i * a.cap + b];
    const Dest<M> out{a.TA + b * Rec<M>::RT, a.cap * Rec<M>::RT};
    const double J = rollout_cost_rec<M>(params_of<M, PT>(a, b).get(), xs, CLane{a.u + b, a.cap}, out, S, a.SC);     // iLQRController.py:27-34
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
    a.last_search[b] = 0;
    a.pred[b] = 0;
    a.replayed[b] = 0;
    a.dirty[b] = 0;
}

// The loop test of :43 failed before the first iteration (d_tol is +inf/NaN or
// max_iter == 0): the initial rollout is the result.
template <class M>
__global__ __launch_bounds__(PIPE_WG) void pipe_finish_all_kernel(Pipe a, int status)
{
    const long b = (long)blockIdx.x * PIPE_WG + threadIdx.x;
    if (b >= a.B) return;
    copy_traj<M>(slot_T<M>(a.TA, a.cap, (int)b), final_dest<M>(a, (int)b), a.N - 1);
    a.status[b] = status;
}

// ---------------------------------------------------------------- export
// Record layout -> the caller's strided arrays x [N][nx][cap], u [N-1][nu][cap]; reads and writes of
// consecutive lanes are contiguous.
template <class M>
__global__ __launch_bounds__(PIPE_WG) void pipe_export_kernel(Pipe a)
{
    constexpr int nx = M::NX, m = M::NU;
    const long b = (long)blockIdx.x * PIPE_WG + threadIdx.x;
    if (b >= a.B) return;
    const int S = a.N - 1;
    const RecIn T = slot_T<M>(a.TOUT, a.cap, (int)b);
    double xu[nx + m];
    for (int k = 0; k < S; ++k) {
        load_rec<nx + m>(T.p + (long)k * T.ss, xu);
#pragma unroll
        for (int i = 0; i < nx; ++i) a.x[(long)(k * nx + i) * a.cap + b] = xu[i];
#pragma unroll
        for (int i = 0; i < m; ++i) a.u[(long)(k * m + i) * a.cap + b] = xu[nx + i];
    }
    load_rec<nx>(T.p + (long)S * T.ss, xu);
#pragma unroll
    for (int i = 0; i < nx; ++i) a.x[(long)(S * nx + i) * a.cap + b] = xu[i];
}

// ---------------------------------------------------------------- backward
// EXACT: ILQR_FLAG_LAPACK_EIGVALS, every PSD test through the LAPACK path of eig_device.hpp
// DEFER: the launch for the trajectories that have not needed a long regularisation scan so far (see backward_stage)
template <class M, bool PT = false, bool EXACT = false, bool DEFER = false>
__global__ __launch_bounds__(PIPE_WG, M::BACKWARD_WAVES) void pipe_backward_kernel(Pipe a)
{
    __shared__ double lds[stage_lds_doubles<M::NDX, M::NU, TermsOf<M>::type::AUX, (EXACT && !DEFER) ? WIDE_PARK : 64>() +
                          (EXACT ? exact_lds_doubles<M::NDX, M::NU>() : 0)];
    const bool backs = EXACT && DEFER && (int)blockIdx.x < a.back_blocks;
    const int blk = (EXACT && DEFER && !backs) ? (int)blockIdx.x - a.back_blocks : (int)blockIdx.x;
    const int lanes = backs ? a.back_lanes : a.lanes, group = backs ? a.back_group : a.group;
    const int i0 = backs ? (int)threadIdx.x * a.back_blocks + blk
                         : (EXACT && a.spread) ? (int)(threadIdx.x * gridDim.x + blockIdx.x) : blk * lanes + (int)threadIdx.x;
    bool live = (int)threadIdx.x < lanes && i0 < group;
    const int i1 = live ? i0 : group - 1;      // a lane without a trajectory shadows the last one and stores nothing
    const int s = backs ? (int)a.B - 1 - i1 : (EXACT && i1 >= a.n_range) ? a.defer[i1 - a.n_range] : a.s_first + a.s_step * i1;
    const int S = a.N - 1;
    bool need_ls = false, cont = false;
    const SlotId sid = slot_id(a.idA[s]);
    const int id = sid.id;
    live = live && sid.alive;
    const RecIn T = slot_T<M>(a.TA, a.cap, s);
    const int it = a.iters[id] + 1;                                                  // :46
    double deltaJ = 0.0, dmax = 0.0, maxq = a.maxq[id];
    int rr = 0, rs = 0;
    const ParCache<M> par = params_of<M, PT>(a, id);
    const RecOut G = live ? RecOut{a.G + (long)s * Rec<M>::RG, a.cap * Rec<M>::RG} : RecOut{a.Gscratch, Rec<M>::RG};
    bool dirty = EXACT ? (a.dirty[id] != 0) : false;
    bool deferred = false;
#ifdef ILQR_DEBUG_COUNTERS
    const unsigned long long dbg_k0 = __builtin_readcyclecounter();
#endif
    const bool ok = backward_pass<M, EXACT, DEFER>(par.get(), T, G, S, deltaJ, dmax, rr, rs, maxq, live, lds, dirty, deferred);     // :49
#ifdef ILQR_DEBUG_COUNTERS
    if (EXACT && (threadIdx.x & 63) == 0) {
        const unsigned long long dt = __builtin_readcyclecounter() - dbg_k0;
        atomicAdd(&g_dbg[DEFER ? 13 : 14], dt);            // cycles of all wavefronts' passes, first / second launch
        atomicMax(&g_dbg[15], dt);                         // the longest pass of a wavefront
        atomicAdd(&g_dbg[DEFER ? 6 : 7], 1ull);
    }
#endif
    if constexpr (DEFER) {
        // dropped from this launch: nothing of the pass is kept, the next launch of this iteration runs it again
        const int k = wave_append(live && deferred, a.cnt + CNT_DEFER);
        if (live && deferred) {
            a.defer[k] = s;
            a.dirty[id] = 1;
        }
        live = live && !deferred;
    }
    if (live) {
        if constexpr (EXACT) a.dirty[id] = dirty ? 1 : 0;
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
            a.rr_it[id] = rr; a.rs_it[id] = rs; a.dj_it[id] = deltaJ;      // for replay_rest
            // the loop test the trajectory will meet after this iteration's line search (:43)
            cont = (dmax > a.d_tol) && (it < a.max_iter);
            if (!cont) a.status[id] = (dmax > a.d_tol) ? ILQR_EXIT_MAXITER : ILQR_EXIT_DTOL;
        }
        if (!need_ls)   // finished with the trajectory it came in with (:84)
            copy_traj<M>(T, final_dest<M>(a, id), S);
    }
    const int j = wave_append(need_ls, a.cnt + 1);
    const int nfront = wave_append(cont && !dirty, a.cnt + 0);
    const int nback = wave_append(cont && dirty, a.cnt + CNT_DIRTY);
    const int ns = dirty ? (int)a.B - 1 - nback : nfront;
    bool side = false;
    if (need_ls) {
        a.ls_slot[j] = s;
        a.ls_dest[j] = cont ? ns : -1;
        // A trajectory whose last search rejected alpha = 1 will most likely reject it again: its first
        // candidate is then evaluated together with the next three by the side-by-side kernel instead of
        // being computed and written on its own (same candidates, same order, same result either way).
        // When few trajectories are left the launches are bound by latency, not by throughput, and every
        // search goes that way (a.tail): one kernel less per iteration.
        side = !(a.flags & ILQR_FLAG_LS_SEQUENTIAL) && a.max_ls >= 2 && (a.tail || a.pred[id] != 0);
        a.ls_from[j] = side ? 1 : 2;
        if (cont) a.idB[ns] = id;
    }
    // the side-by-side entries open the list of searches that go on after their first candidate
    const int q = wave_append(side, a.cnt + 2);
    if (side) a.bt[q] = j;
}

// A solve can reach a state from which every further iteration repeats the last one exactly.  When a search is
// abandoned the trajectory is kept (iLQRController.py:69-74), so the next backward pass and every candidate of the next
// search are the same numbers again; only J, which became the last candidate's cost (:79), is new.
//   * Abandoned with a NaN cost: J is NaN from now on, no candidate can be accepted against it (:66), and the next
//     search runs through the same candidates up to the same one (the first that is NaN, or number max_ls): same
//     count, same J, for ever.
//   * Abandoned, after an abandoned search, with the J it started from (the same last candidate, equal bit for bit):
//     the next iteration starts from the same x, u and J and so ends the same way, and so on.
// Either way every iteration up to max_iter (while the loop test of :43 holds) is this one over again.  They are not
// run: replay_rest writes the counters, traces and exit status they would leave; J, dmax, last_search and the
// trajectory stay as they are.  Outputs are the same bit for bit as with ILQR_FLAG_NO_REPLAY, which runs them.
__device__ __forceinline__ void replay_rest(const Pipe& a, int id, int it, int ls, double J)
{
    const int rr = a.rr_it[id], rs = a.rs_it[id];
    const double deltaJ = a.dj_it[id];
    for (int t = it + 1; t <= a.max_iter && t <= a.trace_len; ++t) {
        const long q = (long)(t - 1) * a.cap + id;
        if (a.tr_dJ) a.tr_dJ[q] = deltaJ;
        if (a.tr_ls) a.tr_ls[q] = ls;
        if (a.tr_acc) a.tr_acc[q] = 0;
        if (a.tr_J) a.tr_J[q] = J;
    }
    const int more = a.max_iter - it;
    a.iters[id] = a.max_iter;
    a.regu_rounds[id] += more * rr;
    a.regu_stages[id] += more * rs;
    a.ls_total[id] += more * ls;
    a.replayed[id] = more;
    a.status[id] = ILQR_EXIT_MAXITER;      // the loop test held after this iteration (dest >= 0), so dmax > d_tol
}

// Records what a finished line search leaves (:69-79).  `dest` >= 0: the solve goes on after this iteration (the
// slot it was given).  Returns true if it does not after all, because the rest was replayed: the caller then writes
// the trajectory to the export buffer and retires the slot (retire_slot).
__device__ __forceinline__ bool record_search(const Pipe& a, int id, int ls, bool complete, double Jn, int dest)
{
    const double Jold = a.J[id];
    const bool fixed = !complete && dest >= 0 && !(a.flags & ILQR_FLAG_NO_REPLAY) &&
                       ((Jn != Jn) || (a.last_search[id] == 2 && Jold == Jn));
    const int it = a.iters[id];
    if (it <= a.trace_len) {
        const long q = (long)(it - 1) * a.cap + id;
        if (a.tr_ls) a.tr_ls[q] = ls;
        if (a.tr_acc) a.tr_acc[q] = complete ? 1 : 0;
        if (a.tr_J) a.tr_J[q] = Jn;
    }
    a.ls_total[id] += ls;
    a.last_search[id] = complete ? 1 : 2;                                             // :69-74
    a.J[id] = Jn;                                                                     // :79
    if (fixed) replay_rest(a, id, it, ls, Jn);
    return fixed;
}

// A slot of the next iteration's buffer whose trajectory finished after the slot was handed out: the next backward
// launch still covers it, as a lane that only rides along (slot_id).
template <class M>
__device__ __forceinline__ void retire_slot(const Pipe& a, const RecIn& T, int dest, int id)
{
    copy_traj<M>(T, final_dest<M>(a, id), a.N - 1);
    a.idB[dest] = -1 - id;
}

// ---------------------------------------------------------------- forward, alpha = 1
template <class M, bool PT = false>
__global__ __launch_bounds__(PIPE_WG, FWD_WAVES) void pipe_forward1_kernel(Pipe a)
{
    const int j = blockIdx.x * PIPE_WG + threadIdx.x;
    const int nls = a.cnt[1];
    bool need_bt = false;
    if (j < nls && a.ls_from[j] != 1) {      // (side-by-side entries are in the list already)
        const int s = a.ls_slot[j], dest = a.ls_dest[j], id = a.idA[s];
        const RecIn T = slot_T<M>(a.TA, a.cap, s);
        const Dest<M> d = make_dest<M>(a, dest, id);
        const double J = a.J[id], deltaJ = a.dJ[s];
        const double Jn = forward_pass<M>(params_of<M, PT>(a, id).get(), T, slot_G<M>(a, s), 1.0, d, a.N - 1, a.SC);  // :63-64
        const bool complete = (Jn - J <= -1e-2 * 1.0 * deltaJ) && (1 < a.max_ls);      // :66
        const bool abandon = (Jn != Jn) || (1 >= a.max_ls);                            // :67
        if (complete || abandon) {
            if (!complete) copy_traj<M>(T, d, a.N - 1);    // search abandoned: the trajectory is kept (:69-72)
            if (record_search(a, id, 1, complete, Jn, dest)) retire_slot<M>(a, T, dest, id);
            a.pred[id] = 0;
        } else {
            need_bt = true;
        }
    }
    const int q = wave_append(need_bt, a.cnt + 2);
    if (need_bt) a.bt[q] = j;
}

// ---------------------------------------------------------------- one more candidate for the entries of a list
// Evaluation number `ls` (alpha = 0.5^(ls-1)) for every entry of `list_in`,
// written; entries whose search goes on are appended to `list_out`.  (The
// strictly sequential schedule, ILQR_FLAG_LS_SEQUENTIAL.)
template <class M, bool PT = false>
__global__ __launch_bounds__(PIPE_WG, FWD_WAVES) void pipe_forward_level_kernel(Pipe a, const int32_t* __restrict__ list_in,
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
        const RecIn T = slot_T<M>(a.TA, a.cap, s);
        const Dest<M> d = make_dest<M>(a, dest, id);
        const double J = a.J[id], deltaJ = a.dJ[s];
        const double Jn = forward_pass<M>(params_of<M, PT>(a, id).get(), T, slot_G<M>(a, s), alpha, d, a.N - 1, a.SC);
        const bool complete = (Jn - J <= -1e-2 * alpha * deltaJ) && (ls < a.max_ls);   // :66
        const bool abandon = (Jn != Jn) || (ls >= a.max_ls);                           // :67
        if (complete || abandon) {
            if (!complete) copy_traj<M>(T, d, a.N - 1);
            if (record_search(a, id, ls, complete, Jn, dest)) retire_slot<M>(a, T, dest, id);
        } else {
            more = true;
        }
    }
    const int o = wave_append(more, n_out);
    if (more) list_out[o] = j;
}

// ---------------------------------------------------------------- GL candidates per trajectory side by side
// GL neighbouring lanes share one trajectory; lane g evaluates line-search
// evaluation number ls_from + g (alpha = 0.5^(number - 1)), cost only, where
// ls_from is the entry's first evaluation not made yet.  The
// search result is that of the first evaluation, in order, at which the
// sequential loop would have stopped (:66-67); a ballot finds it and its
// number and cost are left in win_ls / win_J for `commit`.  If none of the GL
// stops: with REPEAT the same lanes go on with the next GL numbers (the last
// number, max_ls, always stops), otherwise the entry is appended to list_out.
template <class M, int GL, bool REPEAT, bool PT = false>
__global__ __launch_bounds__(PIPE_WG, FWD_WAVES) void pipe_backtrack_group_kernel(Pipe a, const int32_t* __restrict__ list_in,
                                                                       const int32_t* __restrict__ n_in,
                                                                       int32_t* __restrict__ list_out,
                                                                       int32_t* __restrict__ n_out)
{
    static_assert(64 % GL == 0, "whole groups per wavefront");
    constexpr int PER_WAVE = 64 / GL;
    const int lane = threadIdx.x & 63;
    const int grp = lane / GL, g = lane % GL;
    const unsigned long long gmask = ((GL >= 64) ? ~0ull : ((1ull << GL) - 1ull)) << (grp * GL);
    const int n = *n_in;
    // grid-stride over the list: the host sizes the grid from an upper bound of the list's length
    for (int q0 = blockIdx.x * PER_WAVE; q0 < n; q0 += gridDim.x * PER_WAVE) {
        const int q = q0 + grp;
        const bool live = q < n;
        int j = 0, s = 0, first_ls = 0, id = 0;
        double J = 0.0, deltaJ = 0.0;
        if (live) {
            j = list_in[q];
            s = a.ls_slot[j];
            first_ls = a.ls_from[j];
            id = a.idA[s];
            J = a.J[id];
            deltaJ = a.dJ[s];
        }
        const ParCache<M> par = params_of<M, PT>(a, id);
        bool pending = live;
        for (int base = first_ls; __ballot(pending) != 0ull; base += GL) {     // base differs between groups
            const int ls = base + g;
            bool stop = false, complete = false;
            double Jn = 0.0;
            if (pending && ls <= a.max_ls) {
                double alpha = 1.0;
                for (int i = 1; i < ls; ++i) alpha *= 0.5;                                 // :73
                Jn = forward_cost<M>(par.get(), slot_T<M>(a.TA, a.cap, s), slot_G<M>(a, s), alpha, a.N - 1, a.SC);
                complete = (Jn - J <= -1e-2 * alpha * deltaJ) && (ls < a.max_ls);          // :66
                stop = complete || (Jn != Jn) || (ls >= a.max_ls);                         // :67
            }
            const unsigned long long stops = __ballot(stop) & gmask;
            if (pending && stops) {
                if (lane == __ffsll((long long)stops) - 1) {
                    a.win_ls[j] = complete ? ls : -ls;
                    a.win_J[j] = Jn;
                }
                pending = false;
            }
            if (!REPEAT) break;
        }
        if (!REPEAT) {
            const int o = wave_append(pending && g == 0, n_out);
            if (pending && g == 0) {
                list_out[o] = j;
                a.ls_from[j] = first_ls + GL;
            }
        }
    }
}

// ---------------------------------------------------------------- write what the backtracking searches decided
// One lane per entry of the list of searches whose first candidate was
// rejected: the accepted candidate is recomputed and written, or, when the
// search was abandoned, the trajectory is carried over unchanged (:69-72).
template <class M, bool PT = false>
__global__ __launch_bounds__(PIPE_WG, FWD_WAVES) void pipe_commit_kernel(Pipe a, const int32_t* __restrict__ list_in,
                                                              const int32_t* __restrict__ n_in)
{
    const int q = blockIdx.x * PIPE_WG + threadIdx.x;
    if (q >= *n_in) return;
    const int j = list_in[q];
    const int s = a.ls_slot[j], dest = a.ls_dest[j], id = a.idA[s];
    const int w = a.win_ls[j];
    const bool complete = w > 0;
    const int ls = complete ? w : -w;
    const RecIn T = slot_T<M>(a.TA, a.cap, s);
    const Dest<M> d = make_dest<M>(a, dest, id);
    if (complete) {
        double alpha = 1.0;
        for (int i = 1; i < ls; ++i) alpha *= 0.5;
        forward_pass<M>(params_of<M, PT>(a, id).get(), T, slot_G<M>(a, s), alpha, d, a.N - 1, a.SC);
    } else {
        copy_traj<M>(T, d, a.N - 1);
    }
    if (record_search(a, id, ls, complete, a.win_J[j], dest)) retire_slot<M>(a, T, dest, id);
    a.pred[id] = (complete && ls == 1) ? 0 : 1;
}

}  // namespace ilqr
