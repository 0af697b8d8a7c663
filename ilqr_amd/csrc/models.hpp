// Plant models as compile-time policies for the gfx950 kernels.
//
// Each policy gives the dimensions and the five operations the reference's
// Model interface defines (control/models/model.py:9-53): discrete_dynamics,
// _cost_stage, _cost_final, stage, final, plus CalcDifferentialState.  The
// straight-line arithmetic comes from the generated headers in gen/ (SymPy
// differentiation done offline, in place of the reference's run-time
// pydrake.symbolic / CasADi evaluation).
//
// All matrices are row-major in per-lane private arrays.  stage()/final()
// return cost derivatives in the ndx-dimensional error state, so the generic
// Q-function assembly of model.py:45-53 applies to every plant; for the
// quaternion plant that means the projections l_x G, G' l_xx G, l_ux G of
// Quadrotor_Quat.py:267-275 are applied inside stage().
#pragma once
#include <hip/hip_runtime.h>

#include "gen/acrobot_gen.h"
#include "gen/car_gen.h"
#include "gen/quad_ea_gen.h"
#include "gen/quad_quat_gen.h"
#ifdef ILQR_USER_MODEL
#include "user_gen.h"   // written by ilqr_amd/codegen.py for a plant the user defined; found through -I
#endif

namespace ilqr {

// Loop-unrolling policy: small plants keep every matrix in registers (fully
// unrolled loops, compile-time indices); the 12-state plants keep loops rolled
// so their matrices live in per-lane private memory.

template <int NX>
__device__ __forceinline__ void diff_sub(const double* s, const double* r, double* dx)
{
#pragma unroll
    for (int i = 0; i < NX; ++i) dx[i] = s[i] - r[i];   // model.py:22-23
}

// ---------------------------------------------------------------- car
struct CarModel {
    static constexpr int NX = 5, NU = 2, NDX = 5, NPAR = 2;
    static constexpr bool UNROLL = true;
    static constexpr bool TEAM_BACKWARD = false;
    static constexpr bool HAS_CONTINUOUS = false;     // car.py states the step itself (x + xdot dt folded in)
    // second argument of the backward kernel's __launch_bounds__ (minimum waves per SIMD the register allocator must
    // leave room for).  1 = no cap below the 512 registers of a lane: the kernel comes out at 242 VGPRs, no AGPRs, no
    // scratch, i.e. two resident waves per SIMD; asking for 3 or 4 makes the compiler spill ~1 KB per lane (measured
    // slower than 2 resident waves)
    static constexpr int BACKWARD_WAVES = 1;
    // running trajectories (per part of a batch) below which every line search is evaluated side by side
    static constexpr long TAIL_COUNT = 262144;
    __device__ static void dynamics(const double* par, const double* x, const double* u, double* xn)
    {
        car_gen_dynamics(par, x, u, xn);
    }
    __device__ static double cost_stage(const double* par, const double* x, const double* u)
    {
        double c;
        car_gen_cost_stage(par, x, u, &c);
        return c;
    }
    __device__ static double cost_final(const double* par, const double* x)
    {
        double c;
        car_gen_cost_final(par, x, &c);
        return c;
    }
    __device__ static void stage(const double* par, const double* x, const double* u, double* lx, double* lu,
                                 double* lxx, double* lux, double* luu, double* fx, double* fu)
    {
        car_gen_stage(par, x, u, lx, lu, lxx, lux, luu, fx, fu);
    }
    __device__ static void final(const double* par, const double* x, double* Vx, double* Vxx)
    {
        car_gen_final(par, x, Vx, Vxx);
    }
    __device__ static void diff_state(const double*, const double* s, const double* r, double* dx)
    {
        diff_sub<NX>(s, r, dx);
    }
};

#ifdef ILQR_USER_MODEL
// ------------------------------------------- plant defined by the user
// The reference's own extension point (README.md:25-31): the user writes
// discrete_dynamics, _cost_stage and _cost_final; derivatives are taken
// symbolically (car.py:30-52).  Here those three functions were traced once and
// printed as user_gen_* with the car's signatures; par holds USER_NP numbers
// the user may change between solves without rebuilding.
// A plant with nx + nu > 8 arrives here stated with 12 states and 4 controls (USER_WIDE; the added states are
// constant zero, the added controls cost u^2/2 and move nothing) and takes the 16-lanes-per-trajectory backward
// kernel of the quadrotors.
struct UserModel {
    // USER_NDX differs from USER_NX for a plant on an error state; such a plant states its own stage, final and
    // Q_terms like control/models/Quadrotor_Quat.py:248-275 (USER_HAS_QTERMS: user_gen_stage returns the cost
    // derivatives on the state and f_x, f_u on the error state, user_gen_qterms is the plant's Q_terms)
    static constexpr int NX = USER_NX, NU = USER_NU, NDX = USER_NDX, NPAR = USER_NP;
    static constexpr bool HAS_CONTINUOUS = false;
    static constexpr int BACKWARD_WAVES = 1;
#ifdef USER_WIDE
    static constexpr bool UNROLL = false;
    static constexpr bool TEAM_BACKWARD = true;
    static constexpr long TAIL_COUNT = 8192;
    static_assert(NX == 12 && NU == 4, "wide user plants are stated with 12 states and 4 controls");
    __device__ static bool euler(const double*) { return false; }      // (the plant states its own step)
#else
    static constexpr bool UNROLL = true;
    static constexpr bool TEAM_BACKWARD = false;
    static constexpr long TAIL_COUNT = 16384;
    static_assert(NX + NU <= 8, "the lane-per-trajectory backward stage keeps every matrix in registers");
#endif
#ifdef USER_HAS_QTERMS
    // what the plant's own stage() returns, and its Q_terms applied to it
    __device__ static void stage_raw(const double* par, const double* x, const double* u, double* lx, double* lu,
                                     double* lxx, double* lux, double* luu, double* fx, double* fu)
    {
        user_gen_stage(par, x, u, lx, lu, lxx, lux, luu, fx, fu);
    }
    __device__ static void q_terms(const double* par, const double* x, const double* lx, const double* lu, const double* lxx,
                                   const double* lux, const double* luu, const double* fx, const double* fu,
                                   const double* Vx, const double* V, double* Qx, double* Qu, double* Qxx, double* Qux,
                                   double* Quu)
    {
        user_gen_qterms(par, x, lx, lu, lxx, lux, luu, fx, fu, Vx, V, Qx, Qu, Qxx, Qux, Quu);
    }
#endif
    __device__ static void dynamics(const double* par, const double* x, const double* u, double* xn)
    {
        user_gen_dynamics(par, x, u, xn);
    }
    __device__ static double cost_stage(const double* par, const double* x, const double* u)
    {
        double c;
        user_gen_cost_stage(par, x, u, &c);
        return c;
    }
    __device__ static double cost_final(const double* par, const double* x)
    {
        double c;
        user_gen_cost_final(par, x, &c);
        return c;
    }
    // (for a plant with its own Q_terms the cost derivatives come back on the error state: its Q_terms at V = 0,
    // the convention of the quaternion quadrotor)
    __device__ static void stage(const double* par, const double* x, const double* u, double* lx, double* lu,
                                 double* lxx, double* lux, double* luu, double* fx, double* fu)
    {
#ifdef USER_HAS_QTERMS
        double rx[NX], rxx[NX * NX], rux[NU * NX], Vx0[NDX], V0[NDX * NDX];
        for (int i = 0; i < NDX; ++i) Vx0[i] = 0.0;
        for (int i = 0; i < NDX * NDX; ++i) V0[i] = 0.0;
        double ru[NU], ruu[NU * NU];
        user_gen_stage(par, x, u, rx, ru, rxx, rux, ruu, fx, fu);
        user_gen_qterms(par, x, rx, ru, rxx, rux, ruu, fx, fu, Vx0, V0, lx, lu, lxx, lux, luu);
#else
        user_gen_stage(par, x, u, lx, lu, lxx, lux, luu, fx, fu);
#endif
    }
    __device__ static void final(const double* par, const double* x, double* Vx, double* Vxx)
    {
        user_gen_final(par, x, Vx, Vxx);
    }
    __device__ static void diff_state(const double* par, const double* s, const double* r, double* dx)
    {
#ifdef USER_HAS_DIFF
        user_gen_diff_state(par, s, r, dx);      // the plant's own CalcDifferentialState
#else
        (void)par;
        static_assert(NDX == NX, "an error state of another dimension needs the plant's own CalcDifferentialState");
        diff_sub<NX>(s, r, dx);
#endif
    }
};
#endif

// ------------------------------------------------- RK4 (acrobot.py:76-97)
// `euler`: one explicit Euler step instead, x + xdot dt (`integration` other than "RK4": acrobot.py:83-86,
// Quadrotor_EA.py:139-142, Quadrotor_Quat.py:123-126).  The flag is the same for every lane of a launch that shares a
// parameter block, so the branch is uniform.
template <int NX, void (*XDOT)(const double*, const double*, double*)>
__device__ inline void rk4_step(double dt, bool euler, const double* x, const double* u, double* xn)
{
    double f1[NX], f2[NX], f3[NX], f4[NX], xs[NX];
    XDOT(x, u, f1);
    if (euler) {
        for (int i = 0; i < NX; ++i) xn[i] = x[i] + f1[i] * dt;
        return;
    }
    for (int i = 0; i < NX; ++i) xs[i] = x[i] + 0.5 * f1[i] * dt;
    XDOT(xs, u, f2);
    for (int i = 0; i < NX; ++i) xs[i] = x[i] + 0.5 * f2[i] * dt;
    XDOT(xs, u, f3);
    for (int i = 0; i < NX; ++i) xs[i] = x[i] + 1.0 * f3[i] * dt;
    XDOT(xs, u, f4);
    for (int i = 0; i < NX; ++i) xn[i] = x[i] + (f1[i] + 2 * f2[i] + 2 * f3[i] + f4[i]) * dt / 6.0;
}

// Kx = A (I + c Kxp), Ku = A (c Kup) + B; accumulates w*Kx, w*Ku into Sx, Su.
template <int NX, int NU>
__device__ inline void rk4_chain(double c, double w, const double* A, const double* B, double* Kx, double* Ku,
                                 double* Sx, double* Su)
{
    // Kx/Ku hold the previous stage on entry and the new stage on exit.
    double Tx[NX * NX], Tu[NX * NU];
    for (int i = 0; i < NX; ++i) {
        for (int j = 0; j < NX; ++j) Tx[i * NX + j] = (i == j ? 1.0 : 0.0) + c * Kx[i * NX + j];
        for (int j = 0; j < NU; ++j) Tu[i * NU + j] = c * Ku[i * NU + j];
    }
    for (int i = 0; i < NX; ++i) {
        for (int j = 0; j < NX; ++j) {
            double s = 0.0;
            for (int t = 0; t < NX; ++t) s += A[i * NX + t] * Tx[t * NX + j];
            Kx[i * NX + j] = s;
            Sx[i * NX + j] += w * s;
        }
        for (int j = 0; j < NU; ++j) {
            double s = 0.0;
            for (int t = 0; t < NX; ++t) s += A[i * NX + t] * Tu[t * NU + j];
            s += B[i * NU + j];
            Ku[i * NU + j] = s;
            Su[i * NU + j] += w * s;
        }
    }
}

// d x_next/dx and d x_next/du of one RK4 step: the chain rule through the four
// stages, i.e. what CasADi's ca.jacobian of the RK4 graph evaluates
// (acrobot.py:92-93, Quadrotor_EA.py:148-149, Quadrotor_Quat.py:132-133).
template <int NX, int NU, void (*XDOTJAC)(const double*, const double*, double*, double*, double*)>
__device__ inline void rk4_jac(double dt, bool euler, const double* x, const double* u, double* Fx, double* Fu)
{
    double f[NX], xs[NX], A[NX * NX], B[NX * NU], Kx[NX * NX], Ku[NX * NU];
    XDOTJAC(x, u, f, Kx, Ku);
    if (euler) {      // d(x + xdot dt)/dx = I + A dt, d/du = B dt
        for (int i = 0; i < NX; ++i) {
            for (int j = 0; j < NX; ++j) Fx[i * NX + j] = (i == j ? 1.0 : 0.0) + Kx[i * NX + j] * dt;
            for (int j = 0; j < NU; ++j) Fu[i * NU + j] = Ku[i * NU + j] * dt;
        }
        return;
    }
    for (int i = 0; i < NX * NX; ++i) Fx[i] = Kx[i];
    for (int i = 0; i < NX * NU; ++i) Fu[i] = Ku[i];
    for (int i = 0; i < NX; ++i) xs[i] = x[i] + 0.5 * f[i] * dt;
    XDOTJAC(xs, u, f, A, B);
    rk4_chain<NX, NU>(0.5 * dt, 2.0, A, B, Kx, Ku, Fx, Fu);
    for (int i = 0; i < NX; ++i) xs[i] = x[i] + 0.5 * f[i] * dt;
    XDOTJAC(xs, u, f, A, B);
    rk4_chain<NX, NU>(0.5 * dt, 2.0, A, B, Kx, Ku, Fx, Fu);
    for (int i = 0; i < NX; ++i) xs[i] = x[i] + 1.0 * f[i] * dt;
    XDOTJAC(xs, u, f, A, B);
    rk4_chain<NX, NU>(dt, 1.0, A, B, Kx, Ku, Fx, Fu);
    for (int i = 0; i < NX; ++i) {
        for (int j = 0; j < NX; ++j) Fx[i * NX + j] = (i == j ? 1.0 : 0.0) + Fx[i * NX + j] * dt / 6.0;
        for (int j = 0; j < NU; ++j) Fu[i * NU + j] = Fu[i * NU + j] * dt / 6.0;
    }
}

// Tangent of one RK4 step: d x_next along the direction (vx, vu), the same chain rule as rk4_jac applied
// to one vector instead of the identity: d_i = A_i (vx + c_i d_(i-1)) + B_i vu.
template <int NX, void (*XDOTJVP)(const double*, const double*, const double*, const double*, double*, double*)>
__device__ inline void rk4_jvp(double dt, bool euler, const double* x, const double* u, const double* vx, const double* vu,
                               double* dxn)
{
    double f[NX], d[NX], acc[NX], xs[NX], vs[NX];
    for (int i = 0; i < NX; ++i) { acc[i] = 0.0; f[i] = 0.0; d[i] = 0.0; }
    // one copy of the (large) generated function: the four stages run through the same code (Euler: the first only).
    // The trip count is kept the same for the whole wavefront (with a parameter block per trajectory `euler` is a
    // per-lane value): a lane that integrates by Euler in a wavefront that also holds RK4 lanes idles through the
    // other three stages with weight zero.
    const int stages = (__ballot(!euler) != 0ull) ? 4 : 1;
#pragma unroll 1
    for (int stage = 0; stage < stages; ++stage) {
        const double c = (stage == 0) ? 0.0 : (stage == 3 ? 1.0 : 0.5);
        const double wgt = (stage == 0 || stage == 3) ? 1.0 : 2.0;
        for (int i = 0; i < NX; ++i) {
            xs[i] = (stage == 0) ? x[i] : x[i] + c * f[i] * dt;
            vs[i] = (stage == 0) ? vx[i] : vx[i] + (c * dt) * d[i];
        }
        XDOTJVP(xs, u, vs, vu, f, d);
        const bool counts = !euler || stage == 0;
        for (int i = 0; i < NX; ++i) acc[i] += counts ? wgt * d[i] : 0.0;
    }
    for (int i = 0; i < NX; ++i) dxn[i] = euler ? vx[i] + acc[i] * dt : vx[i] + acc[i] * dt / 6.0;
}

// 0.5 e'We evaluated as ((0.5 e)' W) e  (acrobot.py:119-120, :127)
template <int D>
__device__ __forceinline__ double half_form(const double* W, const double* e)
{
    double s = 0.0;
    for (int j = 0; j < D; ++j) {
        double t = 0.0;
        for (int i = 0; i < D; ++i) t += (0.5 * e[i]) * W[i * D + j];
        s += t * e[j];
    }
    return s;
}

// --------------------- quadratic-cost RK4 plants (acrobot.py, Quadrotor_EA.py)
// parameter block: [dt, Q(nx*nx), Qf(nx*nx), R(nu*nu), xref(nx), uref(nu), euler]   (euler != 0: Euler integration)
template <int NX_, int NU_, bool UNROLL_, void (*XDOT)(const double*, const double*, double*),
          void (*XDOTJAC)(const double*, const double*, double*, double*, double*),
          void (*XDOTJVP)(const double*, const double*, const double*, const double*, double*, double*)>
struct QuadraticRK4Model {
    static constexpr int NX = NX_, NU = NU_, NDX = NX_;
    // 12 states + 4 controls: the backward pass runs with 16 lanes per trajectory (team_backward.hpp)
    static constexpr bool TEAM_BACKWARD = (NX_ == 12 && NU_ == 4);
    static constexpr bool HAS_CONTINUOUS = true;
    // _linearize_continuous (acrobot.py:148-150, Quadrotor_EA.py:206-208): A = d xdot/dx, B = d xdot/du
    __device__ static void linearize_continuous(const double*, const double* x, const double* u, double* A, double* B)
    {
        double f[NX];
        XDOTJAC(x, u, f, A, B);
    }
    // one column (or any combination of columns) of [d x_next/dx, d x_next/du]
    __device__ static void dynamics_jvp(double dt, bool eul, const double* x, const double* u, const double* vx,
                                        const double* vu, double* dxn)
    {
        rk4_jvp<NX, XDOTJVP>(dt, eul, x, u, vx, vu, dxn);
    }
    static constexpr int NPAR = 1 + 2 * NX * NX + NU * NU + NX + NU + 1;
    __device__ static bool euler(const double* p) { return p[NPAR - 1] != 0.0; }
    static constexpr bool UNROLL = UNROLL_;
    static constexpr int BACKWARD_WAVES = 1;
    static constexpr long TAIL_COUNT = 8192;
    __device__ static const double* Qm(const double* p) { return p + 1; }
    __device__ static const double* Qf(const double* p) { return p + 1 + NX * NX; }
    __device__ static const double* Rm(const double* p) { return p + 1 + 2 * NX * NX; }
    __device__ static const double* xref(const double* p) { return p + 1 + 2 * NX * NX + NU * NU; }
    __device__ static const double* uref(const double* p) { return p + 1 + 2 * NX * NX + NU * NU + NX; }

    __device__ static void dynamics(const double* par, const double* x, const double* u, double* xn)
    {
        rk4_step<NX, XDOT>(par[0], euler(par), x, u, xn);
    }
    __device__ static double cost_stage(const double* par, const double* x, const double* u)
    {
        double ex[NX], eu[NU];
        for (int i = 0; i < NX; ++i) ex[i] = x[i] - xref(par)[i];
        for (int i = 0; i < NU; ++i) eu[i] = u[i] - uref(par)[i];
        return half_form<NX>(Qm(par), ex) + half_form<NU>(Rm(par), eu);
    }
    __device__ static double cost_final(const double* par, const double* x)
    {
        double ex[NX];
        for (int i = 0; i < NX; ++i) ex[i] = x[i] - xref(par)[i];
        return half_form<NX>(Qf(par), ex);
    }
    // acrobot.py:152-161 / Quadrotor_EA.py:210-219
    __device__ static void stage(const double* par, const double* x, const double* u, double* lx, double* lu,
                                 double* lxx, double* lux, double* luu, double* fx, double* fu)
    {
        rk4_jac<NX, NU, XDOTJAC>(par[0], euler(par), x, u, fx, fu);
        const double *Q = Qm(par), *R = Rm(par);
        for (int i = 0; i < NX; ++i) {
            double s = 0.0;
            for (int j = 0; j < NX; ++j) s += Q[i * NX + j] * (x[j] - xref(par)[j]);
            lx[i] = s;
        }
        for (int i = 0; i < NU; ++i) {
            double s = 0.0;
            for (int j = 0; j < NU; ++j) s += R[i * NU + j] * (u[j] - uref(par)[j]);
            lu[i] = s;
        }
        for (int i = 0; i < NX * NX; ++i) lxx[i] = Q[i];
        for (int i = 0; i < NU * NU; ++i) luu[i] = R[i];
        for (int i = 0; i < NU * NX; ++i) lux[i] = 0.0;
    }
    // acrobot.py:163-166
    __device__ static void final(const double* par, const double* x, double* Vx, double* Vxx)
    {
        const double* W = Qf(par);
        for (int i = 0; i < NX; ++i) {
            double s = 0.0;
            for (int j = 0; j < NX; ++j) s += W[i * NX + j] * (x[j] - xref(par)[j]);
            Vx[i] = s;
        }
        for (int i = 0; i < NX * NX; ++i) Vxx[i] = W[i];
    }
    __device__ static void diff_state(const double*, const double* s, const double* r, double* dx)
    {
        for (int i = 0; i < NX; ++i) dx[i] = s[i] - r[i];
    }
};

using AcrobotModel = QuadraticRK4Model<4, 1, true, acrobot_gen_xdot, acrobot_gen_xdot_jac, acrobot_gen_xdot_jvp>;
using QuadEAModel = QuadraticRK4Model<12, 4, false, quad_ea_gen_xdot, quad_ea_gen_xdot_jac, quad_ea_gen_xdot_jvp>;

// ------------------------------------ quaternion quadrotor (Quadrotor_Quat.py)
// parameter block: [dt, Q(12x12), Qf(12x12), R(4x4), xref(13), uref(4), euler]; Q, Qf, R any matrices, row-major
struct QuadQuatModel {
    static constexpr int NX = 13, NU = 4, NDX = 12, NPAR = 1 + 144 + 144 + 16 + 13 + 4 + 1;
    static constexpr bool UNROLL = false;
    static constexpr bool TEAM_BACKWARD = true;      // 16 lanes per trajectory (team_backward.hpp)
    static constexpr bool HAS_CONTINUOUS = true;
    static constexpr int BACKWARD_WAVES = 1;
    static constexpr long TAIL_COUNT = 8192;
    __device__ static const double* Qm(const double* p) { return p + 1; }
    __device__ static const double* Qf(const double* p) { return p + 145; }
    __device__ static const double* Rm(const double* p) { return p + 289; }
    __device__ static const double* xref(const double* p) { return p + 305; }
    __device__ static const double* uref(const double* p) { return p + 318; }
    __device__ static bool euler(const double* p) { return p[NPAR - 1] != 0.0; }

    __device__ static void dynamics(const double* par, const double* x, const double* u, double* xn)
    {
        rk4_step<13, quad_quat_gen_xdot>(par[0], euler(par), x, u, xn);
    }
    // state_err = Gx(x)' (x - xref)  (Quadrotor_Quat.py:150): three attitude entries, nine differences
    __device__ static void state_err(const double* par, const double* x, double* e)
    {
        quad_quat_gen_err(xref(par), x, e);
        for (int i = 0; i < 9; ++i) e[3 + i] = x[4 + i] - xref(par)[4 + i];
    }
    __device__ static double cost_stage(const double* par, const double* x, const double* u)
    {
        double e[12], du[4];
        state_err(par, x, e);
        for (int i = 0; i < 4; ++i) du[i] = u[i] - uref(par)[i];
        return half_form<12>(Qm(par), e) + half_form<4>(Rm(par), du);          // Quadrotor_Quat.py:152
    }
    __device__ static double cost_final(const double* par, const double* x)
    {
        double e[12];
        state_err(par, x, e);
        return half_form<12>(Qf(par), e);                                       // Quadrotor_Quat.py:153
    }
    // Derivatives of c = 0.5 e'We (+ 0.5 du'R du unless fin) in the 13-dim state, W = Q or Qf
    // (what ca.jacobian gives, Quadrotor_Quat.py:158-175): with Ws = (W + W')/2, s = Ws e and J = de/dx,
    //   l_x = s'J,  l_xx = J' Ws J + sum_i s_i d2e_i/dx2,  l_u = Rs du,  Rs = (R + R')/2.
    // J = blkdiag(da/dq [3x4], I9); only the three attitude entries of e have second derivatives.
    // Ws [12x12] and Rs [4x4] come in symmetrised (sym_weight below).  `work`: COST_WORK doubles for e, da/dq and the
    // Hessians of a (a caller short of registers hands in memory that is not private).
    static constexpr int COST_WORK = 12 + 12 + 48;
    __device__ static void cost_derivs_sym(const double* Ws, const double* Rs, const double* xr, const double* ur,
                                           bool fin, const double* x, const double* u, double* l13, double* lu,
                                           double* l13x13, double* work)
    {
        double* const e = work;
        double* const Ja = work + 12;
        double* const Ha = work + 24;
        double s[12];
        quad_quat_gen_err_derivs(xr, x, e, Ja, Ha);
        for (int i = 0; i < 9; ++i) e[3 + i] = x[4 + i] - xr[4 + i];
        for (int i = 0; i < 12; ++i) {
            double a = 0.0;
            for (int j = 0; j < 12; ++j) a += Ws[i * 12 + j] * e[j];
            s[i] = a;
        }
        for (int b = 0; b < 4; ++b) l13[b] = s[0] * Ja[b] + s[1] * Ja[4 + b] + s[2] * Ja[8 + b];
        for (int c = 0; c < 9; ++c) l13[4 + c] = s[3 + c];
        double T[12];      // Ws[0..2][0..2] Ja   (3 x 4)
        for (int i = 0; i < 3; ++i)
            for (int b = 0; b < 4; ++b) {
                double t = 0.0;
                for (int j = 0; j < 3; ++j) t += Ws[i * 12 + j] * Ja[j * 4 + b];
                T[i * 4 + b] = t;
            }
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) {
                double t = 0.0;
                for (int i = 0; i < 3; ++i) t += Ja[i * 4 + a] * T[i * 4 + b] + s[i] * Ha[i * 16 + a * 4 + b];
                l13x13[a * 13 + b] = t;
            }
        for (int a = 0; a < 4; ++a)
            for (int c = 0; c < 9; ++c) {
                double t = 0.0;
                for (int i = 0; i < 3; ++i) t += Ja[i * 4 + a] * Ws[i * 12 + 3 + c];
                l13x13[a * 13 + 4 + c] = t;
                l13x13[(4 + c) * 13 + a] = t;
            }
        for (int c = 0; c < 9; ++c)
            for (int d = 0; d < 9; ++d) l13x13[(4 + c) * 13 + 4 + d] = Ws[(3 + c) * 12 + 3 + d];
        for (int i = 0; i < 4; ++i) {
            double t = 0.0;
            if (!fin)
                for (int j = 0; j < 4; ++j) t += Rs[i * 4 + j] * (u[j] - ur[j]);
            lu[i] = t;
        }
    }
    // entry e of [(Q + Q')/2 (144), (Qf + Qf')/2 (144), (R + R')/2 (16)]
    static constexpr int SYM_WEIGHTS = 144 + 144 + 16;
    __device__ static double sym_weight(const double* par, int e)
    {
        if (e < 288) {
            const double* const W = (e < 144) ? Qm(par) : Qf(par);
            const int q = (e < 144) ? e : e - 144, i = q / 12, j = q % 12;
            return 0.5 * (W[i * 12 + j] + W[j * 12 + i]);
        }
        const int q = e - 288;
        return luu(par, q / 4, q % 4);
    }
    __device__ static void cost_derivs(const double* par, bool fin, const double* x, const double* u, double* l13,
                                       double* lu, double* l13x13)
    {
        double Ws[144], Rs[16], work[COST_WORK];
        for (int e = 0; e < 144; ++e) Ws[e] = sym_weight(par, (fin ? 144 : 0) + e);
        for (int e = 0; e < 16; ++e) Rs[e] = sym_weight(par, 288 + e);
        cost_derivs_sym(Ws, Rs, xref(par), uref(par), fin, x, u, l13, lu, l13x13, work);
    }
    // l_uu = (R + R')/2
    __device__ static double luu(const double* par, int i, int j) { return 0.5 * (Rm(par)[i * 4 + j] + Rm(par)[j * 4 + i]); }
    // The state Jacobian Gx = blkdiag(G(q)[4x3], I9) (Quadrotor_Quat.py:142-144)
    // is applied block-wise: only the quaternion rows/columns need arithmetic.
    //   out[12 x C] = Gx' * in[13 x C]
    template <int C>
    __device__ static void projT_left(const double* G, const double* in, double* out)
    {
        for (int j = 0; j < C; ++j) {
            for (int i = 0; i < 3; ++i) {
                double s = 0.0;
                for (int t = 0; t < 4; ++t) s += G[t * 3 + i] * in[t * C + j];
                out[i * C + j] = s;
            }
            for (int i = 0; i < 9; ++i) out[(3 + i) * C + j] = in[(4 + i) * C + j];
        }
    }
    //   out[R x 12] = in[R x 13] * Gx
    template <int R>
    __device__ static void proj_right(const double* G, const double* in, double* out)
    {
        for (int r = 0; r < R; ++r) {
            for (int j = 0; j < 3; ++j) {
                double s = 0.0;
                for (int t = 0; t < 4; ++t) s += in[r * 13 + t] * G[t * 3 + j];
                out[r * 12 + j] = s;
            }
            for (int j = 0; j < 9; ++j) out[r * 12 + 3 + j] = in[r * 13 + 4 + j];
        }
    }
    // Quadrotor_Quat.py:231-238 (f_x = G'AG, f_u = G'B), :248-257, :267-275
    __device__ static void stage(const double* par, const double* x, const double* u, double* lx, double* lu,
                                 double* lxx, double* lux, double* luu, double* fx, double* fu)
    {
        double G[12];
        double l13[13], l13x13[169], T[12 * 13];
        quad_quat_gen_G(x, G);
        cost_derivs(par, false, x, u, l13, lu, l13x13);
        proj_right<1>(G, l13, lx);                 // (l_x G)
        projT_left<13>(G, l13x13, T);              // G' l_xx       [12 x 13]
        proj_right<12>(G, T, lxx);                 // (G' l_xx) G   [12 x 12]
        for (int i = 0; i < 4 * 12; ++i) lux[i] = 0.0;   // l_ux = 0, so l_ux G = 0
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) luu[i * 4 + j] = QuadQuatModel::luu(par, i, j);
        {
            double A[169], B[52];
            rk4_jac<13, 4, quad_quat_gen_xdot_jac>(par[0], euler(par), x, u, A, B);
            projT_left<13>(G, A, T);               // G' A
            proj_right<12>(G, T, fx);              // G' A G
            projT_left<4>(G, B, fu);               // G' B
        }
    }
    // _linearize_continuous (Quadrotor_Quat.py:240-246): G' A G [12x12], G' B [12x4] of the continuous dynamics
    __device__ static void linearize_continuous(const double*, const double* x, const double* u, double* Ac, double* Bc)
    {
        double G[12], f[13], A[169], B[52], T[12 * 13];
        quad_quat_gen_G(x, G);
        quad_quat_gen_xdot_jac(x, u, f, A, B);
        projT_left<13>(G, A, T);
        proj_right<12>(G, T, Ac);
        projT_left<4>(G, B, Bc);
    }
    // Quadrotor_Quat.py:259-265
    __device__ static void final(const double* par, const double* x, double* Vx, double* Vxx)
    {
        double G[12], v13[13], vu[4], v13x13[169], T[12 * 13];
        quad_quat_gen_G(x, G);
        cost_derivs(par, true, x, uref(par), v13, vu, v13x13);
        projT_left<13>(G, v13x13, T);
        proj_right<12>(G, T, Vxx);
        proj_right<1>(G, v13, Vx);
    }
    // Quadrotor_Quat.py:218-228: G(state)' (state - ref)
    __device__ static void diff_state(const double*, const double* s, const double* r, double* dx)
    {
        double G[12], e[13];
        quad_quat_gen_G(s, G);
        for (int i = 0; i < 13; ++i) e[i] = s[i] - r[i];
        projT_left<1>(G, e, dx);
    }
};

}  // namespace ilqr
