// Host side of the batched solve: the loop that launches the pipeline kernels of solve_pipeline.hpp for one
// part of a batch, iteration by iteration (control/iLQRController.py:43-82 as a schedule of launches).
//
// The library is built from two translation units.  lane_plants.hip holds run_part for the plants whose
// backward pass runs one trajectory per lane and whose policy class can be named in two translation units
// (car, user-defined plants; the acrobot's type is built from file-local generated functions and stays in the other) and is compiled with
// `-mllvm -disable-machine-licm`: otherwise the constants of the trigonometric polynomials are hoisted out of
// the stage loop into ~40 extra registers, which takes the car's backward kernel from two wavefronts per
// SIMD to one (a single wavefront can issue fp64 instructions at about half the SIMD's rate).  ilqr_hip.hip
// holds everything else (the C ABI, the 12-state plants with their 16-lanes-per-trajectory backward kernel).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../include/ilqr_hip.h"
#include "solver_device.hpp"
#include "solve_pipeline.hpp"
#include "team_backward.hpp"

namespace ilqr {

// records the message ilqr_last_error() returns (per calling thread) and returns `code`
int fail(int code, const char* fmt, const char* a = "");

#define HIP_TRY(expr)                                                             \
    do {                                                                          \
        hipError_t e_ = (expr);                                                   \
        if (e_ != hipSuccess) return ::ilqr::fail(ILQR_E_HIP, #expr ": %s", hipGetErrorString(e_)); \
    } while (0)

// ILQR_FLAG_TIMING: a pair of events around every launch, summed per phase when the solve is over.
inline thread_local std::vector<hipEvent_t> g_events;     // per host thread: each part of a solve is driven by its own thread

struct PhaseTimer {
    bool on;
    hipStream_t st;
    size_t used = 0;
    std::vector<int> phase;
    PhaseTimer(bool enable, hipStream_t s) : on(enable), st(s) {}
    void begin(int ph)
    {
        if (!on) return;
        while (g_events.size() < used + 2) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) { on = false; return; }
            g_events.push_back(e);
        }
        phase.push_back(ph);
        (void)hipEventRecord(g_events[used], st);
    }
    void end()
    {
        if (!on) return;
        (void)hipEventRecord(g_events[used + 1], st);
        used += 2;
    }
    int finish(int iterations, ilqr_timing* dst)
    {
        if (dst) memset(dst, 0, sizeof(*dst));
        if (dst) dst->iterations = iterations;
        if (!on || !dst) return ILQR_OK;
        HIP_TRY(hipStreamSynchronize(st));
        for (size_t i = 0; i < phase.size() && 2 * i + 1 < used + 1; ++i) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, g_events[2 * i], g_events[2 * i + 1]));
            dst->ms[phase[i]] += ms;
            dst->launches[phase[i]] += 1;
        }
        return ILQR_OK;
    }
};

inline unsigned blocks_for(long n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// One part of the batch (a contiguous range of trajectories with its own slots, lists and counters) from the
// initial rollout to the last iteration, on one stream.  Blocks the calling thread on `st` once per solver
// iteration to size the next launches (a few bytes read back).
template <class M, bool PT = false>
int run_part(Pipe a, hipStream_t st, ilqr_timing* tout)
{
    PhaseTimer tm((a.flags & ILQR_FLAG_TIMING) != 0, st);
    tm.begin(ILQR_PHASE_INIT);
    hipLaunchKernelGGL((pipe_init_kernel<M, PT>), dim3(blocks_for(a.B, PIPE_WG)), dim3(PIPE_WG), 0, st, a);
    tm.end();
    HIP_TRY(hipGetLastError());
    // the loop test of iLQRController.py:43 before the first iteration: max|d| is +inf there
    if (!(INFINITY > a.d_tol) || !(0 < a.max_iter)) {
        const int status = !(INFINITY > a.d_tol) ? ILQR_EXIT_DTOL : ILQR_EXIT_MAXITER;
        hipLaunchKernelGGL(pipe_finish_all_kernel<M>, dim3(blocks_for(a.B, PIPE_WG)), dim3(PIPE_WG), 0, st, a, status);
        HIP_TRY(hipGetLastError());
        tm.begin(ILQR_PHASE_EXPORT);
        hipLaunchKernelGGL(pipe_export_kernel<M>, dim3(blocks_for(a.B, PIPE_WG)), dim3(PIPE_WG), 0, st, a);
        tm.end();
        HIP_TRY(hipGetLastError());
        return tm.finish(0, tout);
    }
    // Line-search schedule after evaluation 1 (all three give bit-identical results):
    //   default                  the next GA evaluations GA lanes per trajectory, then GB at a time, commit
    //   ILQR_FLAG_PARALLEL_LS    evaluations 2.. sixteen at a time, commit
    //   ILQR_FLAG_LS_SEQUENTIAL  one launch per evaluation over the compacted list of those still searching
    constexpr int GA = 8, GB = 16;
    // side by side means GA lanes per search: affordable while that still fits the GPU in one go
    // (measured on 2^20 car solves in 3 parts: 88.3 ms with 262144, 90.2 ms with 65536, 95 ms with no limit)
    long TAIL_COUNT = M::TAIL_COUNT;
    if (const char* e = getenv("ILQR_HIP_TAIL")) TAIL_COUNT = atol(e);     // for tuning
    int32_t h[64];
    long count = a.B, back = 0;      // running trajectories; how many of them sit in slots handed out from the back
    int iters_run = 0;
    for (int iter = 1; iter <= a.max_iter && count > 0; ++iter) {
        iters_run = iter;
        a.count = (int)count;
        a.iter = iter;
        a.s_first = 0; a.s_step = 1; a.group = (int)(count - back); a.lanes = PIPE_WG; a.n_range = a.group; a.spread = 0;
        a.back_blocks = 0; a.back_group = 0; a.back_lanes = 1;
        // Latency-bound regime (about one wavefront per SIMD or less): fewer, wider launches per iteration.
        // The schedule does not change any result, so neither does the threshold.
        const bool tail = count <= TAIL_COUNT && a.max_ls >= 2 &&
                          !(a.flags & (ILQR_FLAG_LS_SEQUENTIAL | ILQR_FLAG_PARALLEL_LS));
        a.tail = tail ? 1 : 0;
        HIP_TRY(hipMemsetAsync(a.cnt, 0, 64 * sizeof(int32_t), st));
        tm.begin(ILQR_PHASE_BACKWARD);
        if constexpr (M::TEAM_BACKWARD) {
            if constexpr (team_exact_built<M, PT>()) {
                if (a.flags & ILQR_FLAG_LAPACK_EIGVALS)
                    hipLaunchKernelGGL((pipe_backward_team_kernel<M, PT, true>), dim3(blocks_for(count, PIPE_WG / TEAM_LANES)),
                                       dim3(PIPE_WG), 0, st, a);
                else
                    hipLaunchKernelGGL((pipe_backward_team_kernel<M, PT>), dim3(blocks_for(count, PIPE_WG / TEAM_LANES)),
                                       dim3(PIPE_WG), 0, st, a);
            } else {
                hipLaunchKernelGGL((pipe_backward_team_kernel<M, PT>), dim3(blocks_for(count, PIPE_WG / TEAM_LANES)),
                                   dim3(PIPE_WG), 0, st, a);
            }
        } else if (a.flags & ILQR_FLAG_LAPACK_EIGVALS) {
            // Two launches (Pipe::s_first).  The first covers the back and the front of the slot buffer, in that order of
            // workgroups, and drops and lists a trajectory at the stage where it turns out to need a long scan; the second
            // runs the listed ones again from their last stage, few per wavefront.  The back and the list are spread
            // thinly: as few per wavefront as still gives every SIMD its wavefronts (three workgroups of the first
            // kernel fit a CU, four of the second).  Results do not depend on any of it.
            long resident = 768 / (a.parts > 0 ? a.parts : 1);
            if (const char* e = getenv("ILQR_HIP_DIRTY_WAVES")) resident = atol(e);     // for tuning
            {
                Pipe b = a;
                if (back > 0) {
                    long most = 16;
                    if (const char* e = getenv("ILQR_HIP_DIRTY_LANES")) most = atol(e);        // for tuning
                    int lanes = 1;
                    while (lanes < most && (long)blocks_for(back, 2 * lanes) >= resident) lanes *= 2;
                    b.back_lanes = lanes; b.back_group = (int)back; b.back_blocks = (int)blocks_for(back, lanes);
                }
                const unsigned blocks = (unsigned)b.back_blocks + (b.group > 0 ? blocks_for(b.group, PIPE_WG) : 0u);
                if (blocks > 0)
                    hipLaunchKernelGGL((pipe_backward_kernel<M, PT, true, true>), dim3(blocks), dim3(PIPE_WG), 0, st, b);
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(h, a.cnt + CNT_DEFER, sizeof(int32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            const long n_defer = h[0];
            if (n_defer > 0) {
                Pipe b = a;
                b.group = (int)n_defer; b.n_range = 0; b.spread = 1;
                // (the kernel of this launch keeps less in LDS: four of its workgroups fit a CU, and up to WIDE_TRAJ
                // trajectories of a wavefront share each test of 64 matrices)
                long most = WIDE_TRAJ, fill = 1024 / (a.parts > 0 ? a.parts : 1);
                if (const char* e = getenv("ILQR_HIP_HEAVY_LANES")) most = atol(e);        // for tuning
                if (const char* e = getenv("ILQR_HIP_HEAVY_FILL")) fill = atol(e);         // for tuning
                if (most > WIDE_TRAJ) most = WIDE_TRAJ;
                int lanes = 1;
                while (lanes < most && (long)blocks_for(b.group, 2 * lanes) >= fill) lanes *= 2;
                b.lanes = lanes;
                hipLaunchKernelGGL((pipe_backward_kernel<M, PT, true, false>), dim3(blocks_for(b.group, lanes)), dim3(PIPE_WG), 0,
                                   st, b);
            }
        } else {
            hipLaunchKernelGGL((pipe_backward_kernel<M, PT>), dim3(blocks_for(count, PIPE_WG)), dim3(PIPE_WG), 0, st, a);
        }
        tm.end();
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h, a.cnt, 64 * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const long next = h[0] + h[CNT_DIRTY], nls = h[1];
        const long next_back = h[CNT_DIRTY];
        if (nls > 0) {
            if (!tail) {
                tm.begin(ILQR_PHASE_FORWARD1);
                hipLaunchKernelGGL((pipe_forward1_kernel<M, PT>), dim3(blocks_for(nls, PIPE_WG)), dim3(PIPE_WG), 0, st, a);
                tm.end();
                HIP_TRY(hipGetLastError());
            }
            // Trajectories whose first candidate was rejected: the kernels below read the list's length on the
            // device; the host only needs an upper bound for the grids, and waiting for the exact number would
            // put a round trip to the host between forward1 and the backtracking kernels of every iteration.
            long bound = nls;
            if (a.max_ls >= 2 && (a.flags & ILQR_FLAG_LS_SEQUENTIAL)) {
                HIP_TRY(hipMemcpyAsync(h + 2, a.cnt + 2, sizeof(int32_t), hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
                bound = h[2];
                int32_t *lin = a.bt, *lout = a.bt2;
                double alpha = 0.5;
                for (int ls = 2; bound > 0 && ls <= a.max_ls; ++ls, alpha *= 0.5) {
                    tm.begin(ILQR_PHASE_BACKTRACK);
                    hipLaunchKernelGGL((pipe_forward_level_kernel<M, PT>), dim3(blocks_for(bound, PIPE_WG)), dim3(PIPE_WG), 0,
                                       st, a, lin, a.cnt + ls, lout, a.cnt + ls + 1, ls, alpha);
                    tm.end();
                    HIP_TRY(hipGetLastError());
                    int32_t* t = lin; lin = lout; lout = t;
                    if ((ls % 4) == 0) {     // tighten the bound now and then
                        HIP_TRY(hipMemcpyAsync(h + 3, a.cnt + ls + 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
                        HIP_TRY(hipStreamSynchronize(st));
                        bound = h[3];
                    }
                }
            } else if (bound > 0 && a.max_ls >= 2) {
                const int32_t* lin = a.bt;
                const int32_t* nin = a.cnt + 2;
                bool more = true;       // can an entry still be searching after the launches so far?
                if (tail) {
                    // one launch: GA candidates at a time until every search has stopped
                    tm.begin(ILQR_PHASE_BACKTRACK);
                    hipLaunchKernelGGL((pipe_backtrack_group_kernel<M, GA, true, PT>), dim3(blocks_for(bound, 64 / GA)),
                                       dim3(PIPE_WG), 0, st, a, lin, nin, (int32_t*)nullptr, (int32_t*)nullptr);
                    tm.end();
                    HIP_TRY(hipGetLastError());
                    more = false;
                } else if (!(a.flags & ILQR_FLAG_PARALLEL_LS)) {
                    tm.begin(ILQR_PHASE_BACKTRACK);
                    // grid-stride kernels: a capped grid covers a long list in several trips and costs little
                    // when the list turns out short
                    const unsigned ga = blocks_for(bound, 64 / GA);
                    hipLaunchKernelGGL((pipe_backtrack_group_kernel<M, GA, false, PT>), dim3(ga < 8192u ? ga : 8192u),
                                       dim3(PIPE_WG), 0, st, a, lin, nin, a.bt2, a.cnt + 3);
                    tm.end();
                    HIP_TRY(hipGetLastError());
                    lin = a.bt2; nin = a.cnt + 3;
                    more = a.max_ls > GA;     // an entry that started at evaluation 1 goes on with number GA + 1
                }
                if (more) {
                    tm.begin(ILQR_PHASE_BACKTRACK);
                    const unsigned gb = blocks_for(bound, 64 / GB);
                    hipLaunchKernelGGL((pipe_backtrack_group_kernel<M, GB, true, PT>), dim3(gb < 1024u ? gb : 1024u),
                                       dim3(PIPE_WG), 0, st, a, lin, nin, (int32_t*)nullptr, (int32_t*)nullptr);
                    tm.end();
                    HIP_TRY(hipGetLastError());
                }
                tm.begin(ILQR_PHASE_COMMIT);
                hipLaunchKernelGGL((pipe_commit_kernel<M, PT>), dim3(blocks_for(bound, PIPE_WG)), dim3(PIPE_WG), 0, st, a,
                                   a.bt, a.cnt + 2);
                tm.end();
                HIP_TRY(hipGetLastError());
            }
        }
        // the buffer the candidates were written to becomes the current one
        double* t = a.TA; a.TA = a.TB; a.TB = t;
        int32_t* ti = a.idA; a.idA = a.idB; a.idB = ti;
        count = next;
        back = next_back;
    }
    tm.begin(ILQR_PHASE_EXPORT);
    hipLaunchKernelGGL(pipe_export_kernel<M>, dim3(blocks_for(a.B, PIPE_WG)), dim3(PIPE_WG), 0, st, a);
    tm.end();
    HIP_TRY(hipGetLastError());
    return tm.finish(iters_run, tout);
}


// which translation unit instantiates run_part for which plant
#ifdef ILQR_MAIN_TU
#ifndef ILQR_ONLY_USER
extern template int run_part<CarModel, false>(Pipe, hipStream_t, ilqr_timing*);
#endif
#if defined(ILQR_USER_MODEL) && !defined(USER_WIDE)      // (a wide user plant takes the 16-lane kernels of the main unit)
extern template int run_part<UserModel, false>(Pipe, hipStream_t, ilqr_timing*);
#endif
#endif

}  // namespace ilqr
