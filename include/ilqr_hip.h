/*
 * libilqr_hip -- batched iLQR on AMD Instinct MI355X (gfx950), C ABI.
 *
 * Every entry point is a drop-in for one piece of the reference's Python
 * solver, computed for B independent trajectories at once.  All pointers are
 * DEVICE pointers unless a comment says "host".  All floating-point data is
 * IEEE fp64.  No entry point throws; each returns 0 on success or a negative
 * ILQR_E_* code, and ilqr_last_error() gives a message for the calling thread.
 *
 * Data layout ("strided structure of arrays"): a per-trajectory vector field
 * with C components is stored as field[c * stride + b] for trajectory b, and a
 * trajectory of T such vectors as field[(k * C + c) * stride + b].  `stride`
 * is the number of trajectories the buffers were allocated for (>= B), so
 * consecutive lanes of a wavefront touch consecutive doubles.
 */
#ifndef ILQR_HIP_H
#define ILQR_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* plant models; parameter block layouts (doubles, device memory):
 *   CAR        [r, v_target]                               control/models/car.py:55-61
 *   ACROBOT    [dt, Q(4x4), Qf(4x4), R(1x1), xref(4), uref(1), euler]        control/models/acrobot.py:6-19,76-86,101-103
 *   QUAD_EA    [dt, Q(12x12), Qf(12x12), R(4x4), xref(12), uref(4), euler]   control/models/Quadrotor_EA.py:23-43,132-142,157-159
 *   QUAD_QUAT  [dt, Q(12x12), Qf(12x12), R(4x4), xref(13), uref(4), euler]   control/models/Quadrotor_Quat.py:39-59,116-126,150-153,178-180
 *   Q, Qf, R: any matrices, row-major (the reference never asks for symmetry).  euler: 0 = the RK4 step
 *   (integration="RK4"), anything else = one explicit Euler step x + xdot dt (any other `integration`).
 * matrices are row-major. */
enum { ILQR_MODEL_CAR = 0, ILQR_MODEL_ACROBOT = 1, ILQR_MODEL_QUAD_EA = 2, ILQR_MODEL_QUAD_QUAT = 3,
       /* a plant the user wrote (the README.md:25-31 workflow of the reference): present only in a library that
        * was compiled for that plant (ilqr_amd/codegen.py build_user_library); such a library holds this id alone,
        * ilqr_model_info reports its dimensions, and params holds the numbers the user declared as changeable */
       ILQR_MODEL_USER = 4 };

/* why a solve returned: the three ways out of the loop in
 * control/iLQRController.py:43-51, and the numpy.linalg.LinAlgError that
 * np.linalg.eigvals (non-finite Q, :110) or np.linalg.inv (singular Q_uu, :126) raise. */
enum { ILQR_EXIT_DTOL = 0, ILQR_EXIT_SMALL_DELTAJ = 1, ILQR_EXIT_MAXITER = 2, ILQR_EXIT_LINALG = 3 };

enum {
    ILQR_OK = 0,
    ILQR_E_BADARG = -1,    /* null pointer, unknown model, N < 2, B < 1, stride < B ... */
    ILQR_E_WORKSPACE = -2, /* workspace missing or too small */
    ILQR_E_HIP = -3,       /* a HIP runtime call failed */
    ILQR_E_NODEVICE = -4   /* no gfx950 device visible */
};

/* replaces the constructor arguments of control/iLQRController.py:8-20 plus the plant's own parameters */
typedef struct {
    int32_t model_id;     /* ILQR_MODEL_* */
    int32_t N;            /* knot points; S = N-1 stages */
    int32_t max_iter;
    int32_t max_ls;       /* max_linesearch_iters */
    double d_tol;
    int64_t B;            /* trajectories in this call */
    int64_t stride;       /* allocation stride of every strided buffer, >= B */
    const double *params; /* device: parameter block of the model (one per trajectory with
                             ILQR_FLAG_PARAMS_PER_TRAJECTORY) */
    int32_t trace_len;    /* rows of the optional per-iteration traces (0 = none) */
    int32_t flags;        /* ILQR_FLAG_* */
    int32_t cost_N;       /* knot points the trajectory cost sums over: stage costs of stages 0 .. cost_N-2, then the
                             final cost of the LAST state of the trajectory.  0 = N.  The reference's acrobot and
                             quadrotor plants sum over the horizon they were constructed with, whatever horizon the
                             solver runs (acrobot.py:139, Quadrotor_EA.py:199, Quadrotor_Quat.py:212);
                             1 <= cost_N <= N (the reference raises IndexError beyond) */
    int32_t reserved_;    /* must be 0 (refused otherwise, so that it can be given a meaning later) */
} ilqr_problem;

/* Layout version of ilqr_problem / ilqr_outputs / ilqr_timing as this header declares them.  The structs carry no
 * size field: a binding built against another header would hand over shorter or longer structs, so it should
 * compare ilqr_abi_version() with the ILQR_ABI_VERSION it was written for when it loads the library.
 *   1: ilqr_problem 56 bytes, ilqr_outputs 13 pointers      2: ilqr_problem 64 bytes (cost_N, reserved_),
 *   ilqr_outputs 15 pointers (last_search, replayed) */
#define ILQR_ABI_VERSION 2
int32_t ilqr_abi_version(void);

/* Line-search evaluation schedule (the result does not depend on it, bit for bit):
 * default: candidate 1 (alpha = 1) computed and written for every trajectory, except those whose previous search
 * rejected it; for the rejected and the excepted ones the next eight candidates in eight neighbouring lanes, then
 * sixteen at a time (cost only; the first one at which the reference's loop would stop wins), then one pass that
 * writes the trajectory each search ended with.  Once few trajectories are still running, every search goes
 * side by side from its first candidate, eight at a time in one launch. */
enum {
    ILQR_FLAG_NONE = 0,
    ILQR_FLAG_PARALLEL_LS = 1,   /* sixteen at a time right after alpha = 1 was rejected */
    ILQR_FLAG_LS_SEQUENTIAL = 2, /* never side by side: one launch per candidate (max_ls <= 60) */
    ILQR_FLAG_TIMING = 4,        /* bracket every kernel launch with HIP events: see ilqr_last_timing() */
    ILQR_FLAG_SINGLE_STREAM = 8, /* run everything on the caller's stream (default: large batches are cut into
                                    parts that run concurrently on internal streams, one host thread each) */
    /* params holds one parameter block per trajectory.  Batched goals: different references / targets per
     * trajectory, the batched form of set_references (control/models/acrobot.py:101-103) and of the car's
     * (r, v_target) (control/models/car.py:55-59).  Layout: a plant whose block has at most 8 numbers (the car,
     * user-defined plants) takes it strided like every other array of the call, parameter i of trajectory b at
     * params[i * stride + b]; the plants with larger blocks (acrobot, quadrotors) take the blocks one after the
     * other, trajectory b's block at params + b * npar, so that its matrices stay contiguous. */
    ILQR_FLAG_PARAMS_PER_TRAJECTORY = 16,
    /* `np.any(np.linalg.eigvals(Q) < 0)` (control/iLQRController.py:110) decided in every backward stage and every
     * regularisation round the way NumPy decides it -- the eigenvalues of LAPACK's DGEEV on the general matrix, both
     * triangles (ilqr_eigvals_negative) -- instead of the default unpivoted LDL' of the lower triangle.  The two agree
     * wherever the smallest eigenvalue is resolved, i.e. on every solve the reference itself reproduces under a
     * 1e-14 perturbation of its inputs.  They part where Q has been regularised to ~1e16 times its own size or has
     * lost its symmetry (long horizons: CarModel N = 200; the quadrotors far from hover): there LAPACK keeps calling Q
     * indefinite for tens of rounds more, and this flag reproduces the reference's exit classes and iteration
     * counts as a population.  In this mode the regularised matrix is also accumulated round by round like the
     * reference's `Q += regu * f'f` (:111-115) and P = (0.1 f)'f is formed like NumPy's matrix product.  An eigenvalue
     * computation per test wherever a cheaper certified bound does not settle the sign: tens of times slower on
     * the headline batch.  Every plant and every parameter layout the library solves. */
    ILQR_FLAG_LAPACK_EIGVALS = 32,
    /* Run every iteration.  By default a solve that has reached a state from which every further iteration repeats the
     * last one exactly is not iterated to max_iter: the counters, traces and exit status those iterations would leave
     * are written directly.  That state is an abandoned line search (control/iLQRController.py:69-79: the trajectory
     * is kept, so the next backward pass and the next search's candidates are the same numbers) that leaves J = NaN,
     * or that follows an abandoned search and leaves the J it started from.  The outputs are the same bit for bit
     * either way (ilqr_outputs::replayed says how many iterations were written that way); the flag is for checking
     * exactly that. */
    ILQR_FLAG_NO_REPLAY = 64
};

/* Kernel phases of ilqr_solve_batch, for ilqr_timing. */
enum {
    ILQR_PHASE_INIT = 0,      /* rollout + cost of the initial guess      (iLQRController.py:27-34) */
    ILQR_PHASE_BACKWARD = 1,  /* backward passes and exit tests           (:43-51, :98-123) */
    ILQR_PHASE_FORWARD1 = 2,  /* first line-search candidate, alpha = 1   (:60-68, :87-96) */
    ILQR_PHASE_BACKTRACK = 3, /* later candidates (cost only or one launch per candidate) */
    ILQR_PHASE_COMMIT = 4,    /* writing the trajectory a backtracking search ended with */
    ILQR_PHASE_EXPORT = 5,    /* finished trajectories -> the caller's x and u arrays  (:84) */
    ILQR_PHASE_COUNT = 6
};

/* host struct: device time and launch count per phase of the last ilqr_solve_batch call that the calling
 * thread made with ILQR_FLAG_TIMING (HIP events on the call's stream around every launch). */
typedef struct {
    double ms[ILQR_PHASE_COUNT];
    int32_t launches[ILQR_PHASE_COUNT];
    int32_t iterations;   /* solver iterations the batch ran (the longest trajectory's) */
    int32_t pad_;
} ilqr_timing;

/* per-trajectory results; any pointer may be NULL.  Arrays are [B] (index b),
 * traces are [trace_len][stride] (index it * stride + b). */
typedef struct {
    double *J0;          /* cost of the initial rollout            (iLQRController.py:32) */
    double *J;           /* the solver's J at return               (iLQRController.py:79) */
    double *dmax;        /* max_k |d_k|_2 at return                (iLQRController.py:43) */
    double *max_abs_q;   /* largest |Q_ij| met in any backward stage (conditioning flag) */
    int32_t *iters;      /* backward passes started                (iLQRController.py:46) */
    int32_t *status;     /* ILQR_EXIT_* */
    int32_t *regu_rounds;/* total regularisation rounds            (iLQRController.py:110-115) */
    int32_t *ls_total;   /* total forward-pass evaluations         (iLQRController.py:61-64) */
    int32_t *regu_stages;/* backward stages that needed at least one regularisation round */
    int32_t *ls_evals;   /* trace: line-search evaluations of iteration it */
    uint8_t *accepted;   /* trace: 1 if the search of iteration it was accepted (:69) */
    double *J_trace;     /* trace: Jn after the search of iteration it (:79) */
    double *deltaJ_trace;/* trace: expected reduction of iteration it (:49) */
    int32_t *last_search;/* the last line search of this call: 0 no iteration reached it, 1 it accepted a
                          * candidate, 2 it was abandoned: num_ls_success / num_ls_fails as the reference leaves
                          * them after run_ilqr (iLQRController.py:57-58, :69-74) */
    int32_t *replayed;   /* iterations (counted in iters) that were not run because they repeat the one before them
                          * exactly; see ILQR_FLAG_NO_REPLAY.  0 with that flag */
} ilqr_outputs;

/* model.get_dims(): control/models/model.py:9-10.  npar = doubles in the parameter block. */
int ilqr_model_info(int32_t model_id, int32_t *nx, int32_t *nu, int32_t *ndx, int32_t *npar);

/* bytes of device workspace ilqr_solve_batch needs for this problem */
size_t ilqr_workspace_bytes(const ilqr_problem *p);

/* iLQR.run_ilqr for B trajectories: control/iLQRController.py:26-84 with
 * _forward_pass (:87-96), _backward_pass (:98-123), _gains (:125-129),
 * _V_terms (:131-134), _expected_cost_reduction (:136-137), _full_hessian
 * (:139-147), Model.rollout / cost_trj / Q_terms / CalcDifferentialState
 * (control/models/model.py:15-53) and the plant's stage/final derivatives.
 *   x0 [nx][stride]        in
 *   u  [N-1][nu][stride]   in: initial controls; out: optimised controls (in place, like :72)
 *   x  [N][nx][stride]     out: optimised states (:71, :84)
 * `stream` is a hipStream_t (NULL = default stream).  The call blocks: it waits on the GPU a few times per
 * solver iteration to read back how many trajectories are still running.  Work enqueued on `stream` before
 * the call is waited for, and work enqueued after it runs after the solve; a batch of 32768 or more is cut
 * into parts that run on internal streams (see ILQR_FLAG_SINGLE_STREAM).  The result does not depend on that. */
int ilqr_solve_batch(const ilqr_problem *p, const double *x0, double *u, double *x, const ilqr_outputs *out,
                     void *workspace, size_t workspace_bytes, void *stream);

/* No counterpart in the reference.  ilqr_solve_batch for callers that cannot allocate device memory (tight serving
 * loops, harnesses that forbid hipMalloc): the same solve with the workspace in
 * a device array that is part of the library, ILQR_STATIC_WORKSPACE_BYTES per GPU, held from the first call on
 * that GPU until the process ends.  The call allocates nothing: no hipMalloc / hipFree, no streams, events or
 * host threads (it runs on `stream` alone, as with ILQR_FLAG_SINGLE_STREAM; ILQR_FLAG_TIMING is refused).  Like
 * ilqr_solve_batch it blocks the host once per solver iteration (a few bytes are read back to size the next
 * launches), so it cannot run on a stream that is being captured into a HIP graph: such a stream is refused
 * with ILQR_E_BADARG before anything is enqueued.  A
 * problem whose ilqr_workspace_bytes() exceeds ILQR_STATIC_WORKSPACE_BYTES is refused with ILQR_E_WORKSPACE
 * (CarModel N = 40: up to about 2 900 trajectories per call).  Results are those of ilqr_solve_batch, bit for bit.
 * All calls on one GPU share the one array: calls that overlap on the same GPU (two host threads, two streams)
 * race and must be serialised by the caller. */
#define ILQR_STATIC_WORKSPACE_BYTES (32u << 20)
int ilqr_solve_batch_static(const ilqr_problem *p, const double *x0, double *u, double *x, const ilqr_outputs *out,
                            void *stream);

/* Model.rollout + Model.cost_trj (control/models/model.py:15-20, :31-37 and
 * the plants' cost_trj overrides): x [N][nx][stride] out, J [B] out. */
int ilqr_rollout_cost(const ilqr_problem *p, const double *x0, const double *u, double *x, double *J, void *stream);

/* One evaluation per trajectory of the plant interface of control/models/model.py:
 * discrete_dynamics (:12), _cost_stage (:25), _cost_final (:28), stage (:39), final (:42).
 *   xs [nx][stride], us [nu][stride] in; any output may be NULL:
 *   xn [nx][stride]; cs, cf [B];
 *   l_x [ndx] l_u [nu] l_xx [ndx*ndx] l_ux [nu*ndx] l_uu [nu*nu] f_x [ndx*ndx] f_u [ndx*nu]  (x stride, row-major)
 *   V_x [ndx], V_xx [ndx*ndx]
 * For QUAD_QUAT the cost derivatives are returned already projected on the
 * 12-dim error state (l_x G, G' l_xx G, l_ux G: Quadrotor_Quat.py:267-275). */
int ilqr_model_eval(const ilqr_problem *p, const double *xs, const double *us, double *xn, double *cs, double *cf,
                    double *l_x, double *l_u, double *l_xx, double *l_ux, double *l_uu, double *f_x, double *f_u,
                    double *V_x, double *V_xx, void *stream);

/* `np.any(np.linalg.eigvals(Q) < 0)`, the test that drives the regularisation loop of control/iLQRController.py:110,
 * for B general (not symmetrised) w x w matrices at once, decided like NumPy decides it: from the eigenvalues of
 * LAPACK's DGEEV (balancing, Householder reduction to Hessenberg form, implicit double-shift QR), computed in
 * LAPACK's and OpenBLAS' order of operations.  This is the test the solve kernels apply wherever a cheaper, certified
 * bound does not settle the sign.
 *   Q [w*w][stride] in, row-major entries: entry (i, j) of matrix b at Q[(i * w + j) * stride + b]
 *   verdict [B] out: 1 = some eigenvalue compares below zero (complex ones by real part, then imaginary part),
 *                    0 = none, -1 = NumPy raises LinAlgError (an entry is not finite, or no convergence)
 * Built for w = 2..8 and 16. */
int ilqr_eigvals_negative(int32_t w, const double *Q, int64_t B, int64_t stride, int32_t *verdict, void *stream);

/* _linearize_continuous (control/models/acrobot.py:148-150, Quadrotor_EA.py:206-208, Quadrotor_Quat.py:240-246):
 * A = d xdot/dx, B = d xdot/du of the continuous-time dynamics at one point per trajectory; for the quaternion
 * plant on the error state, G'AG and G'B.  What the reference's LQR controllers in mode "continuous" linearise
 * (control/QuadrotorControllers.py:123-124, :228-229).
 *   xs [nx][stride], us [nu][stride] in; A [ndx*ndx][stride], B [ndx*nu][stride] out (row-major, either may be NULL).
 * The car and user-defined plants state their discrete step only: ILQR_E_BADARG. */
int ilqr_linearize_continuous(const ilqr_problem *p, const double *xs, const double *us, double *A, double *B,
                              void *stream);

/* Receding-horizon helper (control/QuadrotorControllers.py:72-76, :185-188;
 * control/AcrobotControllers.py:60-63): x0 <- discrete_dynamics(x0, u[0]) and
 * u[:-1] <- u[1:] (last row kept), in place, for B trajectories. */
int ilqr_mpc_advance(const ilqr_problem *p, double *x0, double *u, void *stream);

/* No counterpart in the reference (measurement aid): copies the timing of the calling thread's last
 * ilqr_solve_batch call made with ILQR_FLAG_TIMING into *out (host pointer).  Returns ILQR_E_BADARG if there is none. */
int ilqr_last_timing(ilqr_timing *out);

/* message for the last error on the calling thread (host pointer, never NULL) */
const char *ilqr_last_error(void);

/* library version string (host pointer) */
const char *ilqr_version(void);

#ifdef __cplusplus
}
#endif
#endif
