// Backward pass of the 12-state / 4-control plants with SIXTEEN LANES PER TRAJECTORY.
//
// One lane per trajectory does not work for these plants: a stage needs f_x (12x12), f_u,
// V_xx, the 16x16 Q-function Hessian and its factorisation, about 1000 doubles per lane that
// end up in scratch memory, and the kernel crawls on scratch latency.  Here a wavefront holds
// four trajectories; lane j of a team of 16 owns column j of [f_x f_u] and row j of Q, V_xx
// and P, everything shared sits in LDS (4 KB per team), and nothing spills.
//
//   column j of [f_x f_u]   one forward-mode derivative through the RK4 step along unit
//                           vector j (12 + 4 = 16 directions = 16 lanes), Model.stage
//   row j of Q, P           Q = l_zz + F'(V F) in the reference's order ((F'V) F), model.py:45-53;
//                           P = (0.1 F)'F, iLQRController.py:111-113
//   PSD test                the same unpivoted LDL' with the same tolerance rule as the
//                           one-lane version, rows distributed over lanes, pivot column
//                           exchanged through LDS (iLQRController.py:110)
//   regularisation          team by team: Q + (2^r - 1) P for r = 1, 2, ... (:106-115)
//   gains, V, reduction     :125-137; V_xx is kept as full rows, as in the reference
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "solve_pipeline.hpp"

namespace ilqr {

constexpr int TEAM_LANES = 16;
constexpr int TEAM_LDS = 512;       // doubles of LDS per team
constexpr int TEAM_REGU_LINEAR = 16; // regularisation rounds tried one by one
constexpr int TEAM_REGU_MAX = 1300;  // 0.1 * 2^r is infinite long before

// Offsets (doubles) inside a team's LDS region
namespace tl {
constexpr int F = 0;        // [12][16]  column c of [f_x f_u] in F[r * 16 + c]
constexpr int V = 192;      // [12][12]  V_xx rows
constexpr int VX = 336;     // [12]
constexpr int QR = 348;     // [4][16]   rows 12..15 of Q: [Q_ux Q_uu]
constexpr int QZ = 412;     // [16]      [Q_x Q_u]
constexpr int GB = 428;     // [4 + 48]  the stage's G record: d, then K row-major
constexpr int COL = 480;    // [16 + 1]  pivot column exchange, then the pivot's tolerance
}  // namespace tl

// `np.any(np.linalg.eigvals(Q) < 0)` for the 16x16 matrix whose row i (entries 0..i-1 in Lr, diagonal in
// diag) is held by lane i of the team.  Same rule as has_negative_eig_packed.  Lr and diag are destroyed.
// Returns 1 / 0 / -1, the same value in all lanes of a team.  All 64 lanes must call it together.
__device__ inline int team_has_negative_eig(double* Lr, double diag, int i, unsigned long long tmask, double* col,
                                            double& maxq)
{
    double mq = fmax(maxq, fabs(diag));
    double poison = __builtin_fma(0.0, diag, 0.0);
#pragma unroll
    for (int c = 0; c < TEAM_LANES - 1; ++c) {
        const double v = (c < i) ? Lr[c] : 0.0;
        mq = fmax(mq, fabs(v));
        poison = __builtin_fma(0.0, v, poison);
    }
    maxq = mq;
    const bool bad = (__ballot(poison != poison) & tmask) != 0ull;
    double acc = fabs(diag);
    bool negative = false;
#pragma unroll
    for (int p = 0; p < TEAM_LANES; ++p) {
        const double c = (p < TEAM_LANES - 1) ? Lr[p < TEAM_LANES - 1 ? p : 0] : 0.0;     // L[i][p], meaningful for i > p
        wave_sync();
        col[i] = (i == p) ? diag : c;
        if (i == p) col[TEAM_LANES] = PIVOT_TOL * acc;
        wave_sync();
        const double d = col[p], thr = col[TEAM_LANES];
        const bool neg = d < -thr;
        const bool zero = !neg && (d <= thr);
        const bool nz = (i > p) && (c * c > 4.0 * thr * fabs(diag));
        const bool any_nz = (__ballot(nz) & tmask) != 0ull;
        negative = negative || neg || (zero && any_nz);
        double pinv = __builtin_amdgcn_rcp(d);
        pinv = __builtin_fma(__builtin_fma(-d, pinv, 1.0), pinv, pinv);
        pinv = __builtin_fma(__builtin_fma(-d, pinv, 1.0), pinv, pinv);
        pinv = (zero || negative) ? 0.0 : pinv;
        if (i > p) {
            const double lij = c * pinv;
#pragma unroll
            for (int k = p + 1; k < TEAM_LANES - 1; ++k)
                if (k < i) Lr[k] -= lij * col[k];
            diag -= lij * c;
            acc += fabs(lij * c);
        }
    }
    return bad ? -1 : (negative ? 1 : 0);
}

// ------------------------------------------------------------------
// What the team kernel asks of a plant (lane j of a team, all 16 lanes call together):
//   SHARED            doubles of LDS per workgroup for data common to all teams (filled by setup)
//   EXTRA             doubles of LDS per team for the plant's own scratch
//   setup             once per kernel
//   column            column j of [f_x f_u] at (x, u), integrator step dt / kind given -> col[12]
//   cost_row          l_z[j] and row j of l_zz = [[l_xx, l_ux'], [l_ux, l_uu]] -> lz, lzz[16]
//   final_row         V_x[j] and row j of V_xx at the last knot point (j < 12)  -> vx, vrow[12]  (shared as above)
// ------------------------------------------------------------------
template <class M>
struct TeamOps;

// quadratic cost, state = error state (acrobot.py:152-166 / Quadrotor_EA.py:210-224)
template <bool U, void (*XD)(const double*, const double*, double*),
          void (*XJ)(const double*, const double*, double*, double*, double*),
          void (*XV)(const double*, const double*, const double*, const double*, double*, double*)>
struct TeamOps<QuadraticRK4Model<12, 4, U, XD, XJ, XV>> {
    using M = QuadraticRK4Model<12, 4, U, XD, XJ, XV>;
    static constexpr int n = 12, m = 4, w = 16;
    static constexpr int SHARED = w * w, EXTRA = 0;
    __device__ static void setup(const double* par, int lane, double* shared)
    {
        for (int e = lane; e < w * w; e += PIPE_WG) {       // l_zz = blkdiag(Q, R)
            const int rj = e / w, c = e % w;
            double v = 0.0;
            if (rj < n && c < n) v = M::Qm(par)[rj * n + c];
            if (rj >= n && c >= n) v = M::Rm(par)[(rj - n) * m + (c - n)];
            shared[e] = v;
        }
    }
    // the same for one team's own copy (a parameter block per trajectory), filled by the team's 16 lanes
    __device__ static void setup_team(const double* par, int j, double* shared)
    {
        for (int e = j; e < w * w; e += TEAM_LANES) {
            const int rj = e / w, c = e % w;
            double v = 0.0;
            if (rj < n && c < n) v = M::Qm(par)[rj * n + c];
            if (rj >= n && c >= n) v = M::Rm(par)[(rj - n) * m + (c - n)];
            shared[e] = v;
        }
    }
    __device__ static void column(const double*, double dt, bool eul, const double* xu, int j, double*, double* col)
    {
        double vx[n], vu[m];
#pragma unroll
        for (int c = 0; c < n; ++c) vx[c] = (c == j) ? 1.0 : 0.0;
#pragma unroll
        for (int c = 0; c < m; ++c) vu[c] = (c + n == j) ? 1.0 : 0.0;
        M::dynamics_jvp(dt, eul, xu, xu + n, vx, vu, col);
    }
    __device__ static void cost_row(const double* par, const double* xu, int j, const double* shared, double*,
                                    double& lz, double* lzz)
    {
        const double* const row = shared + j * w;
        double sx = 0.0, su = 0.0;
#pragma unroll
        for (int c = 0; c < w; ++c) lzz[c] = row[c];
#pragma unroll
        for (int c = 0; c < n; ++c) sx += lzz[c] * (xu[c] - M::xref(par)[c]);
#pragma unroll
        for (int c = 0; c < m; ++c) su += lzz[n + c] * (xu[n + c] - M::uref(par)[c]);
        lz = (j < n) ? sx : su;
    }
    __device__ static void final_row(const double* par, const double* x, int j, const double*, double*, double& vx,
                                     double* vrow)
    {
        double sx = 0.0;
        const int jj = (j < n) ? j : 0;
#pragma unroll
        for (int c = 0; c < n; ++c) {
            const double q = M::Qf(par)[jj * n + c];
            vrow[c] = q;
            sx += q * (x[c] - M::xref(par)[c]);
        }
        vx = sx;
    }
};

// quaternion quadrotor: 13 states, 12-dim error state through Gx = blkdiag(G(q), I9)
// (Quadrotor_Quat.py:142-175, :231-238, :259-275)
template <>
struct TeamOps<QuadQuatModel> {
    using M = QuadQuatModel;
    static constexpr int n = 12, m = 4, w = 16;
    // shared: the symmetrised weights [(Q + Q')/2, (Qf + Qf')/2, (R + R')/2], formed once per kernel, then xref (13)
    // and uref (4): inside the stage loop nothing is read through the parameter pointer
    static constexpr int XREF = M::SYM_WEIGHTS, UREF = XREF + 13, SHARED = UREF + 4 + 1 /* even */;
    // per team: l_x (13), l_u (4), l_xx (13 x 13), then the cost's work arrays at an even offset
    static constexpr int WORK = 188, EXTRA = WORK + M::COST_WORK;
    __device__ static double shared_entry(const double* par, int e)
    {
        if (e < XREF) return M::sym_weight(par, e);
        if (e < UREF) return M::xref(par)[e - XREF];
        return (e < UREF + 4) ? M::uref(par)[e - UREF] : 0.0;
    }
    __device__ static void setup(const double* par, int lane, double* shared)
    {
        for (int e = lane; e < SHARED; e += PIPE_WG) shared[e] = shared_entry(par, e);
    }
    __device__ static void setup_team(const double* par, int j, double* shared)
    {
        for (int e = j; e < SHARED; e += TEAM_LANES) shared[e] = shared_entry(par, e);
    }
    // entry (t, c) of Gx [13 x 12]
    __device__ static double gx(const double* G, int t, int c)
    {
        if (t < 4) return (c < 3) ? G[t * 3 + c] : 0.0;
        return (t - 1 == c) ? 1.0 : 0.0;
    }
    // out[c] = sum_t v[t] Gx[t][c]  (a 13-vector times Gx), entry c
    __device__ static double times_gx(const double* G, const double* v, int c)
    {
        double s = 0.0;
#pragma unroll
        for (int t = 0; t < 4; ++t) s += v[t] * G[t * 3 + (c < 3 ? c : 0)];
        double pick = 0.0;
#pragma unroll
        for (int t = 4; t < 13; ++t) pick = (t - 1 == c) ? v[t] : pick;
        return (c < 3) ? s : pick;
    }
    __device__ static void column(const double*, double dt, bool eul, const double* xu, int j, double*, double* col)
    {
        double G[12], vx[13], vu[m], dxn[13];
        quad_quat_gen_G(xu, G);
#pragma unroll
        for (int t = 0; t < 13; ++t) vx[t] = (j < n) ? gx(G, t, j) : 0.0;
#pragma unroll
        for (int c = 0; c < m; ++c) vu[c] = (c + n == j) ? 1.0 : 0.0;
        rk4_jvp<13, quad_quat_gen_xdot_jvp>(dt, eul, xu, xu + 13, vx, vu, dxn);
        // Gx' (A v): rows 0..2 through G, the rest shifted by one
#pragma unroll
        for (int i = 0; i < n; ++i) {
            if (i < 3) {
                double s = 0.0;
#pragma unroll
                for (int t = 0; t < 4; ++t) s += G[t * 3 + i] * dxn[t];
                col[i] = s;
            } else {
                col[i] = dxn[i + 1];
            }
        }
    }
    // row j (j < 12) of Gx' H Gx and entry j of g Gx for the 13-dim gradient g and Hessian H held in `ex`
    __device__ static void project_row(const double* G, const double* ex, int j, double& gj, double* row)
    {
        const double* const g13 = ex;
        const double* const H = ex + 17;
        gj = times_gx(G, g13, j);
        double T1[13];      // row j of Gx' H
#pragma unroll
        for (int c = 0; c < 13; ++c) {
            double s = 0.0;
#pragma unroll
            for (int t = 0; t < 4; ++t) s += G[t * 3 + (j < 3 ? j : 0)] * H[t * 13 + c];
            const double pick = H[((j < 3 ? 3 : j) + 1) * 13 + c];
            T1[c] = (j < 3) ? s : pick;
        }
#pragma unroll
        for (int c = 0; c < n; ++c) row[c] = times_gx(G, T1, c);
    }
    __device__ static void cost_row(const double* par, const double* xu, int j, const double* shared, double* ex,
                                    double& lz, double* lzz)
    {
        double G[12];
        quad_quat_gen_G(xu, G);
        // every lane of the team writes the same values: l_x (13), l_u (4), l_xx (13 x 13)
        M::cost_derivs_sym(shared, shared + 288, shared + XREF, shared + UREF, false, xu, xu + 13, ex, ex + 13, ex + 17, ex + WORK);
        wave_sync();
        double gj, row[n];
        project_row(G, ex, (j < n) ? j : 0, gj, row);
#pragma unroll
        for (int c = 0; c < n; ++c) lzz[c] = (j < n) ? row[c] : 0.0;                      // l_ux = 0
#pragma unroll
        for (int c = 0; c < m; ++c) lzz[n + c] = (j >= n) ? shared[288 + (j - n) * 4 + c] : 0.0;     // l_uu = (R + R')/2
        lz = (j < n) ? gj : ex[13 + (j - n)];
    }
    __device__ static void final_row(const double* par, const double* x, int j, const double* shared, double* ex,
                                     double& vx, double* vrow)
    {
        double G[12];
        quad_quat_gen_G(x, G);
        M::cost_derivs_sym(shared + 144, shared + 288, shared + XREF, shared + UREF, true, x, shared + UREF, ex, ex + 13, ex + 17, ex + WORK);
        wave_sync();
        project_row(G, ex, (j < n) ? j : 0, vx, vrow);
    }
};

#if defined(ILQR_USER_MODEL) && defined(USER_WIDE)
// a plant the user wrote, stated with 12 states and 4 controls (models.hpp): the generated functions give the
// directional derivative of the step and the cost terms of z = (x, u) in one piece
template <>
struct TeamOps<UserModel> {
    using M = UserModel;
    static constexpr int n = 12, m = 4, w = 16;
    static constexpr int SHARED = 0, EXTRA = w + w * w;
    __device__ static void setup(const double*, int, double*) {}
    __device__ static void setup_team(const double*, int, double*) {}
    __device__ static void column(const double* par, double, bool, const double* xu, int j, double*, double* col)
    {
        double vx[n], vu[m];
#pragma unroll
        for (int c = 0; c < n; ++c) vx[c] = (c == j) ? 1.0 : 0.0;
#pragma unroll
        for (int c = 0; c < m; ++c) vu[c] = (c + n == j) ? 1.0 : 0.0;
        user_gen_dynamics_jvp(par, xu, xu + n, vx, vu, col);
    }
    __device__ static void cost_row(const double* par, const double* xu, int j, const double*, double* ex, double& lz,
                                    double* lzz)
    {
        // every lane of the team writes the same values: l_z (16), l_zz (16 x 16)
        user_gen_cost_terms(par, xu, xu + n, ex, ex + w);
        wave_sync();
        lz = ex[j];
#pragma unroll
        for (int c = 0; c < w; ++c) lzz[c] = ex[w + j * w + c];
    }
    __device__ static void final_row(const double* par, const double* x, int j, const double*, double* ex, double& vx,
                                     double* vrow)
    {
        user_gen_final(par, x, ex, ex + w);       // V_x (12), V_xx (12 x 12)
        wave_sync();
        const int jj = (j < n) ? j : 0;
        vx = ex[jj];
#pragma unroll
        for (int c = 0; c < n; ++c) vrow[c] = ex[w + jj * n + c];
    }
};
#endif

// ------------------------------------------------------------------
// ILQR_FLAG_LAPACK_EIGVALS for the 16-lanes-per-trajectory kernel: the loop of iLQRController.py:106-115 with every
// test made the way NumPy makes it (eig_device.hpp: DGEEV restated) on the full 16 x 16 matrix, one matrix per lane.
// A team's lanes hold the rows of Q and of P = (0.1 F)'F; they are published in LDS, the upper right block as the
// mirror image of the lower left one (the reference assembles Q from Q_xx, Q_ux, Q_uu, :139-147), and then
//   trip 0: lane R0 t + c tests round c of team t (c = 0..R0-1, R0 = TXL / 4),
//   a team still failing is scanned TXL rounds per trip, lane l testing round base + l,
// round r being the reference's running sum Q(r) = Q(r - 1) + 2^(r-1) P (r rounded additions, not one product).
// LDS: TXL matrices with their work space, the four teams' Q(0) and P, two checkpoints of the running sum.  One test
// takes as long for 1 matrix as for 64 (per-lane loops, one wavefront per CU), so TXL is as many matrices as the
// 160 KB of a CU hold next to the rest of the kernel (team_exact_lanes below).
// ------------------------------------------------------------------
constexpr int TX_PUB = (PIPE_WG / TEAM_LANES) * 2 * 256;
constexpr int TX_PER_LANE = 16 * 16 + 2 * 16;
constexpr int team_exact_lds(int txl) { return TX_PER_LANE * txl + TX_PUB + 2 * 256; }
// matrices per test for a kernel whose other LDS arrays hold `other` doubles: a multiple of 4, at most 64
constexpr int team_exact_lanes(int other)
{
    int t = ((160 * 1024 - 256) / 8 - other - TX_PUB - 2 * 256) / TX_PER_LANE;
    t = t / 4 * 4;
    return t > 64 ? 64 : t;
}

// the lane's matrix: q + (2^k0 + ... + 2^(k0 + nr - 1)) p, added one term after the other; q, p row-major 16 x 16
template <int TXL>
__device__ __forceinline__ void team_exact_build(const eig::Lane<16, TXL>& E, const double* q, const double* p,
                                                 int k0, int nr, double& mq)
{
#pragma clang fp contract(off)
#pragma unroll 1
    for (int e = 0; e < 256; ++e) {
        double s = q[e];
        const double pe = p[e];
        for (int i = 0; i < nr; ++i) s += ldexp(pe, k0 + i);
        const double as = __builtin_fabs(s);
        mq = (as > mq) ? as : mq;
        E.a[((e >> 4) + (e & 15) * 16) * TXL] = s;
    }
}

// Qrow, Prow: row j of the team's Q(0) and P (in: as computed; out: Q(r), and P with the mirrored block).
// r: rounds taken; verdict: the last test's (0 = accepted, -1 = NumPy raises LinAlgError).  All 64 lanes call together.
template <int TXL>
__device__ __forceinline__ void team_regularise_exact(double* Qrow, double* Prow, int lane, int team, int j, bool alive,
                                                      double* X, int& r, int& verdict, double& maxq)
{
#pragma clang fp contract(off)
    constexpr int n = 12, m = 4, w = 16;
    static_assert(TXL % 4 == 0 && TXL >= 8 && TXL <= 64, "matrices per test");
    constexpr int R0 = TXL / 4;                  // rounds per team in trip 0
    constexpr int TX_EIG = TX_PER_LANE * TXL;
    const int el = lane < TXL ? lane : TXL - 1;  // (lanes beyond TXL test nothing)
    const eig::Lane<16, TXL> E{X + el, X + 256 * TXL + el};
    double* const pubQ = X + TX_EIG;
    double* const pubP = pubQ + (PIPE_WG / TEAM_LANES) * 256;
    double* ck = X + TX_EIG + TX_PUB;
    double* ckn = ck + 256;
    double* const myQ = pubQ + team * 256;
    double* const myP = pubP + team * 256;
    wave_sync();
#pragma unroll
    for (int c = 0; c < w; ++c) {
        myQ[j * w + c] = Qrow[c];
        myP[j * w + c] = Prow[c];
    }
    wave_sync();
    if (j < n) {
#pragma unroll
        for (int b = 0; b < m; ++b) {
            Qrow[n + b] = myQ[(n + b) * w + j];
            Prow[n + b] = myP[(n + b) * w + j];
        }
#pragma unroll
        for (int b = 0; b < m; ++b) {
            myQ[j * w + n + b] = Qrow[n + b];
            myP[j * w + n + b] = Prow[n + b];
        }
    }
    wave_sync();

    // ---- trip 0: rounds 0..R0-1 of the four teams
    bool stuck = false;
    {
        int t = 1;
        double mq = 0.0;
        const int tt = (lane < TXL) ? lane / R0 : 0, c = lane - tt * R0;
        if (lane < TXL) {
            team_exact_build<TXL>(E, pubQ + tt * 256, pubP + tt * 256, 0, c, mq);
            t = E.has_negative();
        }
        const unsigned long long passed = __ballot(lane < TXL && t != 1);
        const int bits = (int)((passed >> (R0 * team)) & ((1ull << R0) - 1ull));
        const int cstar = bits ? __ffs(bits) - 1 : -1;
        const int lim = (cstar >= 0) ? cstar : R0 - 1;
        const int tv = __shfl(t, R0 * team + lim);
        double mqt = 0.0;
#pragma unroll
        for (int c2 = 0; c2 < R0; ++c2) {
            const double v = __shfl(mq, R0 * team + c2);
            mqt = (c2 <= lim && v > mqt) ? v : mqt;
        }
        if (alive) {
            maxq = (mqt > maxq) ? mqt : maxq;
            r = (cstar >= 0) ? cstar : R0;
            verdict = tv;
            stuck = cstar < 0;
        } else {
            r = 0;
            verdict = 0;
        }
    }
    if (!stuck && r > 0) {
#pragma unroll
        for (int c = 0; c < w; ++c) {
            double s = Qrow[c];
            for (int k = 0; k < r; ++k) s += ldexp(Prow[c], k);
            Qrow[c] = s;
        }
    }

    // ---- teams still failing after round R0 - 1, one after the other, TXL rounds per trip
    unsigned long long todo = __ballot(stuck);
#pragma unroll 1
    while (todo) {
        const int tq = (__ffsll((long long)todo) - 1) >> 4;
        todo &= ~(0xFFFFull << (16 * tq));
        const double* const q0 = pubQ + tq * 256;
        const double* const pp = pubP + tq * 256;
        wave_sync();
        for (int e = lane; e < 256; e += 64) {       // checkpoint: the sum after R0 rounds
            double s = q0[e];
            for (int k = 0; k < R0; ++k) s += ldexp(pp[e], k);
            ck[e] = s;
        }
        double mqs = 0.0;
#pragma unroll 1
        for (int base = R0;; base += TXL) {
            wave_sync();
            int t = 1;
            double mq = 0.0;
            if (lane < TXL) {
                team_exact_build<TXL>(E, ck, pp, base, lane, mq);
                t = E.has_negative();
            }
            const unsigned long long stops = __ballot(lane < TXL && t != 1);
            const int win = stops ? __ffsll((long long)stops) - 1 : TXL - 1;
            double v = (lane <= win) ? mq : 0.0;
#pragma unroll
            for (int dlt = 1; dlt < 64; dlt *= 2) {
                const double o = __shfl_xor(v, dlt);
                v = (o > v) ? o : v;
            }
            mqs = (v > mqs) ? v : mqs;
            if (stops) {
                const int tw = __shfl(t, win);
                // the accepted matrix goes back to the team's rows: built again (the test destroyed it) in Q(0)'s place
                if (lane == win) {
#pragma unroll 1
                    for (int e = 0; e < 256; ++e) {
                        double s = ck[e];
                        for (int i = 0; i < win; ++i) s += ldexp(pp[e], base + i);
                        pubQ[tq * 256 + e] = s;
                    }
                }
                wave_sync();
                if (team == tq) {
                    r = base + win;
                    verdict = tw;
#pragma unroll
                    for (int c = 0; c < w; ++c) Qrow[c] = q0[j * w + c];
                }
                break;
            }
            if (base + TXL > TEAM_REGU_MAX) {      // 0.1 * 2^r overflowed long before: cannot happen, but never spin
                if (team == tq) {
                    r = base + TXL;
                    verdict = -1;
                }
                break;
            }
            for (int e = lane; e < 256; e += 64) {       // next checkpoint: TXL rounds further
                double s = ck[e];
                for (int i = 0; i < TXL; ++i) s += ldexp(pp[e], base + i);
                ckn[e] = s;
            }
            double* const sw = ck;
            ck = ckn;
            ckn = sw;
        }
        if (team == tq) maxq = (mqs > maxq) ? mqs : maxq;
    }
}

// The exact kernel is built for every plant the 16-lane kernel solves, with one parameter block for the batch or one
// per trajectory.
template <class M, bool PT>
constexpr bool team_exact_built()
{
    return true;
}

// PT: one parameter block per trajectory (at a.par + id * a.par_block) instead of one for all.
// EXACT: ILQR_FLAG_LAPACK_EIGVALS (team_regularise_exact above).
template <class M, bool PT = false, bool EXACT = false>
__global__ __launch_bounds__(PIPE_WG) void pipe_backward_team_kernel(Pipe a)
{
    using OPS = TeamOps<M>;
    constexpr int n = 12, m = 4, w = 16, nx = M::NX;
    static_assert(M::NDX == n && M::NU == m, "team backward pass is written for a 12-dim error state and 4 controls");
    static_assert(Rec<M>::RG == 52, "record sizes");
    constexpr int TEAM_ALL = TEAM_LDS + OPS::EXTRA;
    __shared__ double lds_all[(PIPE_WG / TEAM_LANES) * TEAM_ALL];
    __shared__ double shared_all[OPS::SHARED > 0 ? (PT ? PIPE_WG / TEAM_LANES : 1) * OPS::SHARED : 1];
    constexpr int OTHER_LDS = (PIPE_WG / TEAM_LANES) * TEAM_ALL + (OPS::SHARED > 0 ? (PT ? PIPE_WG / TEAM_LANES : 1) * OPS::SHARED : 1);
    constexpr int TXL = team_exact_lanes(OTHER_LDS);
    __shared__ double exact_all[EXACT ? team_exact_lds(TXL) : 1];
    const int lane = threadIdx.x & 63, team = lane / TEAM_LANES, j = lane % TEAM_LANES;
    const unsigned long long tmask = 0xFFFFull << (team * TEAM_LANES);
    double* const L = lds_all + team * TEAM_ALL;
    double* const EX = L + TEAM_LDS;          // the plant's own per-team scratch
    const int s0 = blockIdx.x * (PIPE_WG / TEAM_LANES) + team;
    const int s = s0 < a.count ? s0 : a.count - 1;      // a team without a trajectory shadows the last one
    const int S = a.N - 1;
    const SlotId sid = slot_id(a.idA[s]);
    const int id = sid.id;
    const bool live = s0 < a.count && sid.alive;         // (a retired slot's team only rides along)
    // a small parameter block (user plants) is copied into registers, one per trajectory if the caller gave one per
    // trajectory; a large one (weights and references) stays in memory, shared or per trajectory (PT)
    const ParCache<M> par_regs = params_of<M, PT>(a, id);
    const double* const par = par_regs.get();
    const RecIn T = slot_T<M>(a.TA, a.cap, s);
    double* const Gp = live ? a.G + (long)s * Rec<M>::RG : a.Gscratch;
    const long Gss = live ? a.cap * Rec<M>::RG : Rec<M>::RG;
    const int it = a.iters[id] + 1;                                                  // iLQRController.py:46
    // the integrator's step and kind: read once (with a parameter block per trajectory they are per-lane values)
    const double step_dt = par[0];
    const bool step_euler = M::euler(par);

    if constexpr (PT) OPS::setup_team(par, j, shared_all + team * OPS::SHARED);
    else OPS::setup(par, lane, shared_all);
    wave_sync();

    // terminal condition (final(): V_x, V_xx at the last knot point)
    double xu[nx + m];
    load_rec<nx>(T.p + (long)S * T.ss, xu);
    {
        double vxj, vrow[n];
        OPS::final_row(par, xu, j, PT ? shared_all + team * OPS::SHARED : shared_all, EX, vxj, vrow);
        if (j < n) {
#pragma unroll
            for (int c = 0; c < n; ++c) L[tl::V + j * n + c] = vrow[c];
            L[tl::VX + j] = vxj;
        }
    }
    double redu = 0.0, best = -INFINITY, maxq = a.maxq[id];
    bool anynan = false, ok = true;
    int rr = 0, rs = 0;

    for (int k = S - 1; k >= 0; --k) {
        load_rec<nx + m>(T.p + (long)k * T.ss, xu);
        const bool alive = live && ok;
        // ---- column j of F = [f_x f_u]: derivative of the RK4 step along unit vector j
        double col[n];
        OPS::column(par, step_dt, step_euler, xu, j, EX, col);
        wave_sync();        // everybody is done with the previous stage's LDS data
#pragma unroll
        for (int r = 0; r < n; ++r) L[tl::F + r * w + j] = col[r];
        // ---- l_z[j] and row j of l_zz
        double lz, lzz[w];
        if constexpr (PT) OPS::cost_row(par, xu, j, shared_all + team * OPS::SHARED, EX, lz, lzz);
        else OPS::cost_row(par, xu, j, shared_all, EX, lz, lzz);
        // ---- Q_z[j] = l_z[j] + F[:,j] . V_x ;  srow = F[:,j]' V_xx
        double qz, srow[n];
        {
            double sv = 0.0;
#pragma unroll
            for (int r = 0; r < n; ++r) sv += col[r] * L[tl::VX + r];
            qz = lz + sv;
#pragma unroll
            for (int c = 0; c < n; ++c) {
                double sacc = 0.0;
#pragma unroll
                for (int r = 0; r < n; ++r) sacc += col[r] * L[tl::V + r * n + c];
                srow[c] = sacc;
            }
        }
        wave_sync();        // F complete
        // ---- row j of Q = l_zz + (F'V) F and of P = (0.1 F)'F
        double Qrow[w], Prow[w];
#pragma unroll
        for (int c = 0; c < w; ++c) {
            double sq = 0.0, sp = 0.0;
#pragma unroll
            for (int r = 0; r < n; ++r) {
                const double f = L[tl::F + r * w + c];
                sq += srow[r] * f;
                sp += (0.1 * col[r]) * f;
            }
            Qrow[c] = lzz[c] + sq;
            Prow[c] = sp;
        }
        L[tl::QZ + j] = qz;
        double qdiag = 0.0, pdiag = 0.0;        // own diagonal entries (lane-dependent position)
#pragma unroll
        for (int c = 0; c < w; ++c) {
            qdiag = (c == j) ? Qrow[c] : qdiag;
            pdiag = (c == j) ? Prow[c] : pdiag;
        }

        // ---- regularisation loop (:106-115): first r = 0, 1, 2, ... whose Q + (2^r - 1) P has no negative
        // eigenvalue.  One loop for all four teams; a team that has its answer idles through the others' tests.
        // Rounds up to TEAM_REGU_LINEAR are tried one after the other like the reference does.  A trajectory that
        // needs more (entries of Q around 1e200 can take a thousand doublings) is searched by doubling r until
        // the test stops failing and bisecting back to the first such r: the same answer as the one-by-one
        // scan whenever "has a negative eigenvalue" does not switch back on as r grows, in ~20 tests.
        if constexpr (EXACT) {
            int r = 0, verdict = 0;
            team_regularise_exact<TXL>(Qrow, Prow, lane, team, j, alive, exact_all, r, verdict, maxq);
            ok = ok && (verdict >= 0);
            if (alive && r > 0) rs += 1;
            rr += r;
        } else {
            int r_lo = -1, r_hi = -1, neg_hi = 0;          // last failing round, first known stopping round
            bool testing = true;
#pragma unroll 1
            while (__ballot(testing) != 0ull) {
                int rt;
                if (r_hi < 0) rt = (r_lo < TEAM_REGU_LINEAR) ? r_lo + 1 : min(2 * r_lo, TEAM_REGU_MAX);
                else rt = (r_lo + r_hi) / 2;
                const double coef = regu_coef(rt);
                double Lr[w - 1];
#pragma unroll
                for (int c = 0; c < w - 1; ++c) Lr[c] = (rt == 0) ? Qrow[c] : __builtin_fma(coef, Prow[c], Qrow[c]);
                const double dg = (rt == 0) ? qdiag : __builtin_fma(coef, pdiag, qdiag);
                double mqt = maxq;
                const int neg = team_has_negative_eig(Lr, dg, j, tmask, L + tl::COL, mqt);
                if (testing) {
                    maxq = mqt;
                    if (alive && neg == 1 && rt < TEAM_REGU_MAX) {
                        r_lo = rt;
                    } else {
                        r_hi = rt;
                        neg_hi = (alive && neg == 1) ? -1 : neg;     // still negative at the cap: 0.1 * 2^r overflowed long ago
                    }
                    testing = (r_hi < 0) || (r_hi - r_lo > 1);
                }
            }
            const int r = r_hi;
            ok = ok && (neg_hi >= 0);
            if (alive && r > 0) rs += 1;
            if (r > 0) {
                const double coef = regu_coef(r);
#pragma unroll
                for (int c = 0; c < w; ++c) Qrow[c] = __builtin_fma(coef, Prow[c], Qrow[c]);
            }
            rr += r;
        }

        // ---- gains (:125-129)
        wave_sync();
        if (j >= n) {
#pragma unroll
            for (int c = 0; c < w; ++c) L[tl::QR + (j - n) * w + c] = Qrow[c];
        }
        wave_sync();
        double Quu[m * m], Qinv[m * m], Qu[m], d[m], Kc[m];
#pragma unroll
        for (int aa = 0; aa < m; ++aa) {
#pragma unroll
            for (int b = 0; b < m; ++b) Quu[aa * m + b] = L[tl::QR + aa * w + n + b];
            Qu[aa] = L[tl::QZ + n + aa];
        }
        ok = inv_small<m>(Quu, Qinv) && ok;
#pragma unroll
        for (int aa = 0; aa < m; ++aa) {
            double sd = 0.0, sk = 0.0;
#pragma unroll
            for (int b = 0; b < m; ++b) {
                sd += (-Qinv[aa * m + b]) * Qu[b];
                sk += (-Qinv[aa * m + b]) * L[tl::QR + b * w + (j < n ? j : 0)];
            }
            d[aa] = sd;
            Kc[aa] = sk;
            if (j < n) L[tl::GB + m + aa * n + j] = sk;
        }
        if (j < m) L[tl::GB + j] = d[j];
        wave_sync();
        // ---- V_xx, V_x rows (:131-134), expected reduction (:136-137)
        if (j < n) {
            double KtQuu[m];
#pragma unroll
            for (int b = 0; b < m; ++b) {
                double sacc = 0.0;
#pragma unroll
                for (int aa = 0; aa < m; ++aa) sacc += Kc[aa] * Quu[aa * m + b];
                KtQuu[b] = sacc;
            }
            double Vnew[n];
#pragma unroll
            for (int c = 0; c < n; ++c) {
                double t1 = 0.0, t2 = 0.0, t3 = 0.0;
#pragma unroll
                for (int aa = 0; aa < m; ++aa) t1 += KtQuu[aa] * L[tl::GB + m + aa * n + c];
#pragma unroll
                for (int aa = 0; aa < m; ++aa) t2 += Kc[aa] * L[tl::QR + aa * w + c];
#pragma unroll
                for (int aa = 0; aa < m; ++aa) t3 += L[tl::QR + aa * w + j] * L[tl::GB + m + aa * n + c];
                Vnew[c] = ((Qrow[c] + t1) + t2) + t3;
            }
            double a1 = 0.0, b1 = 0.0, c1 = 0.0;
#pragma unroll
            for (int aa = 0; aa < m; ++aa) a1 += KtQuu[aa] * d[aa];
#pragma unroll
            for (int aa = 0; aa < m; ++aa) b1 += Kc[aa] * Qu[aa];
#pragma unroll
            for (int aa = 0; aa < m; ++aa) c1 += L[tl::QR + aa * w + j] * d[aa];
#pragma unroll
            for (int c = 0; c < n; ++c) L[tl::V + j * n + c] = Vnew[c];      // old V was last read before F was complete
            L[tl::VX + j] = ((qz + a1) + b1) + c1;
        }
        {
            double ea = 0.0, eb = 0.0;
#pragma unroll
            for (int aa = 0; aa < m; ++aa) ea += (-Qu[aa]) * d[aa];
#pragma unroll
            for (int aa = 0; aa < m; ++aa) {
                double sacc = 0.0;
#pragma unroll
                for (int b = 0; b < m; ++b) sacc += Quu[b * m + aa] * d[b];
                eb += (0.5 * d[aa]) * sacc;
            }
            redu += ea - eb;
        }
        // ---- the stage's G record [d, K] leaves through 13 lanes, 32 bytes each
        if (j < 13) {
            const double2* src = reinterpret_cast<const double2*>(L + tl::GB + 4 * j);
            double2* dst = reinterpret_cast<double2*>(Gp + (long)k * Gss + 4 * j);
            dst[0] = src[0];
            dst[1] = src[1];
        }
        if (alive && ok) {
            double sacc = 0.0;
#pragma unroll
            for (int aa = 0; aa < m; ++aa) sacc += d[aa] * d[aa];
            const double nr = sqrt(sacc);
            anynan = anynan || (nr != nr);
            best = (nr > best) ? nr : best;
        }
    }

    // ---- exit tests and slot allocation, one lane per team (iLQRController.py:43-51)
    // largest |Q| entry over the team's rows
#pragma unroll
    for (int dlt = 1; dlt < TEAM_LANES; dlt *= 2) maxq = fmax(maxq, __shfl_xor(maxq, dlt, TEAM_LANES));
    const double deltaJ = redu;
    const double dmax = anynan ? NAN : best;
    bool need_ls = false, cont = false, finished = false;
    if (live && j == 0) {
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
            cont = (dmax > a.d_tol) && (it < a.max_iter);
            if (!cont) a.status[id] = (dmax > a.d_tol) ? ILQR_EXIT_MAXITER : ILQR_EXIT_DTOL;
        }
        finished = !need_ls;
    }
    // a finished trajectory goes to the export buffer: the team copies its records, 16 bytes per lane
    static_assert(Rec<M>::RT / 2 <= TEAM_LANES, "a T record is copied by one team in one go");
    if ((__ballot(finished) & tmask) != 0ull && j < Rec<M>::RT / 2) {
        const Dest<M> out = final_dest<M>(a, id);
        for (int k = 0; k <= S; ++k) {
            const double2 v = reinterpret_cast<const double2*>(T.p + (long)k * T.ss)[j];
            reinterpret_cast<double2*>(out.t + (long)k * out.ts)[j] = v;
        }
    }
    const int jls = wave_append(need_ls, a.cnt + 1);
    const int ns = wave_append(cont, a.cnt + 0);
    bool side = false;
    if (need_ls) {
        a.ls_slot[jls] = s;
        a.ls_dest[jls] = cont ? ns : -1;
        side = !(a.flags & ILQR_FLAG_LS_SEQUENTIAL) && a.max_ls >= 2 && (a.tail || a.pred[id] != 0);
        a.ls_from[jls] = side ? 1 : 2;
        if (cont) a.idB[ns] = id;
    }
    // the side-by-side entries open the list of searches that go on after their first candidate
    const int q = wave_append(side, a.cnt + 2);
    if (side) a.bt[q] = jls;
}

}  // namespace ilqr
