/* TEST INFRASTRUCTURE ONLY -- plant models of the CPU checker.
 *
 * Car: written by hand in the reference's operation order
 *      (control/models/car.py:64-94; derivatives are the closed forms of what
 *      car.py:5-29 obtains symbolically).  Independent of the generated code
 *      the HIP kernels use.
 * Acrobot / quadrotors: continuous dynamics, the attitude Jacobian and the
 *      quaternion cost are written by hand in oracle/plants_ad.cpp from
 *      acrobot.py:20-73, Quadrotor_EA.py:47-128, Quadrotor_Quat.py:61-153 and
 *      maths/casadi_quaternions.py, and differentiated there by forward-mode
 *      automatic differentiation (dual numbers), independently of the
 *      SymPy-generated code the HIP kernels use; RK4 and its derivative are
 *      written here (acrobot.py:76-97 and the same lines in the quadrotor
 *      files).  CasADi differentiates the RK4 graph by AD; the chain rule
 *      below is that derivative in a fixed evaluation order.
 */
#ifndef ORACLE_MODELS_H
#define ORACLE_MODELS_H
#include <math.h>
#include <string.h>

#include "ilqr_oracle.h"

/* oracle/plants_ad.cpp */
void oracle_acrobot_xdot(const double *x, const double *u, double *f);
void oracle_acrobot_xdot_jac(const double *x, const double *u, double *f, double *A, double *B);
void oracle_quad_ea_xdot(const double *x, const double *u, double *f);
void oracle_quad_ea_xdot_jac(const double *x, const double *u, double *f, double *A, double *B);
void oracle_quad_quat_xdot(const double *x, const double *u, double *f);
void oracle_quad_quat_xdot_jac(const double *x, const double *u, double *f, double *A, double *B);
void oracle_quad_quat_G(const double *x, double *G);
void oracle_quad_quat_cost(const double *w, const double *x, const double *u, double *c);
void oracle_quad_quat_cost_derivs(const double *w, const double *x, const double *u, double *lx, double *lu, double *lxx);

void q_terms_default_export(const ilqr_oracle_model *M, const double *par, const double *x, const double *lx,
                            const double *lu, const double *lxx, const double *lux, const double *luu,
                            const double *fx, const double *fu, const double *Vx, const double *Vxx, double *Qx,
                            double *Qu, double *Qxx, double *Qux, double *Quu);

/* model.py:22-23 */
#define DEF_DIFF_SUB(name, NX)                                                                    \
    static void name(const double *par, const double *s, const double *r, double *dx)             \
    {                                                                                             \
        (void)par;                                                                                \
        for (int i = 0; i < NX; ++i) dx[i] = s[i] - r[i];                                         \
    }

/* ------------------------------ car ------------------------------ */
static void car_dynamics(const double *par, const double *x, const double *u, double *xn)
{
    (void)par;
    const double dt = 0.1;                                   /* car.py:77 */
    double heading = x[2], v = x[3], steer = x[4];
    double xd[5] = {v * cos(heading), v * sin(heading), v * tan(steer), u[0], u[1]}; /* car.py:71-73 */
    for (int i = 0; i < 5; ++i) xn[i] = xd[i] * dt + x[i];   /* car.py:78 */
}
static double car_cost_final(const double *par, const double *x)
{
    const double eps = 1e-6;                                 /* car.py:90 */
    double rho = sqrt(x[0] * x[0] + x[1] * x[1] + eps);
    double c_circle = (rho - par[0]) * (rho - par[0]);
    double c_speed = (x[3] - par[1]) * (x[3] - par[1]);
    return c_circle + c_speed;                               /* car.py:94 */
}
static double car_cost_stage(const double *par, const double *x, const double *u)
{
    const double eps = 1e-6;                                 /* car.py:82 */
    double rho = sqrt(x[0] * x[0] + x[1] * x[1] + eps);
    double c_circle = (rho - par[0]) * (rho - par[0]);
    double c_speed = (x[3] - par[1]) * (x[3] - par[1]);
    double c_control = (u[0] * u[0] + u[1] * u[1]) * 0.1;
    return c_circle + c_speed + c_control;                   /* car.py:87 */
}
static void car_circle_derivs(const double *par, const double *x, double *g, double *H)
{
    /* c = (rho - r)^2 + (v - vt)^2, rho = sqrt(x^2+y^2+eps); gradient g[5], Hessian H[25] */
    double px = x[0], py = x[1], r = par[0];
    double rho2 = px * px + py * py + 1e-6, rho = sqrt(rho2), e = rho - r;
    memset(g, 0, 5 * sizeof(double));
    memset(H, 0, 25 * sizeof(double));
    g[0] = 2.0 * e * px / rho;
    g[1] = 2.0 * e * py / rho;
    g[3] = 2.0 * (x[3] - par[1]);
    double a = 1.0 / rho2, b = e / (rho2 * rho);             /* d(rho)/dx = x/rho; d2(rho)/dx2 = 1/rho - x^2/rho^3 */
    H[0] = 2.0 * (px * px * a + e / rho - px * px * b);
    H[6] = 2.0 * (py * py * a + e / rho - py * py * b);
    H[1] = H[5] = 2.0 * px * py * (a - b);
    H[18] = 2.0;
}
static void car_stage_derivs(const double *par, const double *x, const double *u, double *lx, double *lu,
                             double *lxx, double *lux, double *luu, double *fx, double *fu)
{
    const double dt = 0.1;
    car_circle_derivs(par, x, lx, lxx);
    lu[0] = 0.2 * u[0]; lu[1] = 0.2 * u[1];
    memset(lux, 0, 10 * sizeof(double));
    luu[0] = 0.2; luu[1] = 0.0; luu[2] = 0.0; luu[3] = 0.2;
    double th = x[2], v = x[3], t = tan(x[4]);
    memset(fx, 0, 25 * sizeof(double));
    for (int i = 0; i < 5; ++i) fx[i * 5 + i] = 1.0;
    fx[0 * 5 + 2] = -v * sin(th) * dt; fx[0 * 5 + 3] = cos(th) * dt;
    fx[1 * 5 + 2] = v * cos(th) * dt;  fx[1 * 5 + 3] = sin(th) * dt;
    fx[2 * 5 + 3] = t * dt;            fx[2 * 5 + 4] = v * (1.0 + t * t) * dt;
    memset(fu, 0, 10 * sizeof(double));
    fu[3 * 2 + 0] = dt; fu[4 * 2 + 1] = dt;
}
static void car_final_derivs(const double *par, const double *x, double *Vx, double *Vxx)
{
    car_circle_derivs(par, x, Vx, Vxx);
}
DEF_DIFF_SUB(car_diff, 5)

/* ------------------- RK4 plants with quadratic cost ------------------- */
/* parameter block: [dt, Q(nx*nx), Qf(nx*nx), R(nu*nu), xref(nx), uref(nu), euler]; euler != 0: one explicit
 * Euler step instead of RK4 (`integration` other than "RK4": acrobot.py:83-86, Quadrotor_EA.py:139-142) */
#define QP_DT(p) ((p)[0])
#define QP_EULER(p, nx, nu) ((p)[1 + 2 * (nx) * (nx) + (nu) * (nu) + (nx) + (nu)] != 0.0)
#define QP_Q(p, nx, nu) ((p) + 1)
#define QP_QF(p, nx, nu) ((p) + 1 + (nx) * (nx))
#define QP_R(p, nx, nu) ((p) + 1 + 2 * (nx) * (nx))
#define QP_XREF(p, nx, nu) ((p) + 1 + 2 * (nx) * (nx) + (nu) * (nu))
#define QP_UREF(p, nx, nu) ((p) + 1 + 2 * (nx) * (nx) + (nu) * (nu) + (nx))

typedef void (*xdot_fn)(const double *x, const double *u, double *f);
typedef void (*xdot_jac_fn)(const double *x, const double *u, double *f, double *A, double *B);

/* acrobot.py:76-82: f2 = f(x + 0.5*f1*dt), ..., x + (f1 + 2 f2 + 2 f3 + f4)*dt/6.0 */
static void rk4_step(xdot_fn f, int nx, double dt, int euler, const double *x, const double *u, double *xn)
{
    double f1[ILQR_MAX_NX], f2[ILQR_MAX_NX], f3[ILQR_MAX_NX], f4[ILQR_MAX_NX], xs[ILQR_MAX_NX];
    f(x, u, f1);
    if (euler) {      /* acrobot.py:86: x + xdot * dt */
        for (int i = 0; i < nx; ++i) xn[i] = x[i] + f1[i] * dt;
        return;
    }
    for (int i = 0; i < nx; ++i) xs[i] = x[i] + 0.5 * f1[i] * dt;
    f(xs, u, f2);
    for (int i = 0; i < nx; ++i) xs[i] = x[i] + 0.5 * f2[i] * dt;
    f(xs, u, f3);
    for (int i = 0; i < nx; ++i) xs[i] = x[i] + 1.0 * f3[i] * dt;
    f(xs, u, f4);
    for (int i = 0; i < nx; ++i) xn[i] = x[i] + (f1[i] + 2 * f2[i] + 2 * f3[i] + f4[i]) * dt / 6.0;
}

/* Kx_next = A (I + c Kx_prev), Ku_next = A (c Ku_prev) + B */
static void rk4_chain(int nx, int nu, double c, const double *A, const double *B, const double *Kxp,
                      const double *Kup, double *Kx, double *Ku)
{
    double Tx[ILQR_MAX_NX * ILQR_MAX_NX], Tu[ILQR_MAX_NX * ILQR_MAX_NU];
    for (int i = 0; i < nx; ++i) {
        for (int j = 0; j < nx; ++j) Tx[i * nx + j] = (i == j ? 1.0 : 0.0) + c * Kxp[i * nx + j];
        for (int j = 0; j < nu; ++j) Tu[i * nu + j] = c * Kup[i * nu + j];
    }
    for (int i = 0; i < nx; ++i) {
        for (int j = 0; j < nx; ++j) {
            double s = 0.0;
            for (int t = 0; t < nx; ++t) s += A[i * nx + t] * Tx[t * nx + j];
            Kx[i * nx + j] = s;
        }
        for (int j = 0; j < nu; ++j) {
            double s = 0.0;
            for (int t = 0; t < nx; ++t) s += A[i * nx + t] * Tu[t * nu + j];
            Ku[i * nu + j] = s + B[i * nu + j];
        }
    }
}

/* d x_next / d x (nx x nx) and d x_next / d u (nx x nu) of the RK4 step */
static void rk4_jac(xdot_jac_fn fj, int nx, int nu, double dt, int euler, const double *x, const double *u, double *Fx,
                    double *Fu)
{
    double f[ILQR_MAX_NX], xs[ILQR_MAX_NX];
    double A[ILQR_MAX_NX * ILQR_MAX_NX], B[ILQR_MAX_NX * ILQR_MAX_NU];
    double Kx[4][ILQR_MAX_NX * ILQR_MAX_NX], Ku[4][ILQR_MAX_NX * ILQR_MAX_NU];
    fj(x, u, f, Kx[0], Ku[0]);
    if (euler) {      /* d(x + xdot dt)/dx = I + A dt, d/du = B dt */
        for (int i = 0; i < nx; ++i) {
            for (int j = 0; j < nx; ++j) Fx[i * nx + j] = (i == j ? 1.0 : 0.0) + Kx[0][i * nx + j] * dt;
            for (int j = 0; j < nu; ++j) Fu[i * nu + j] = Ku[0][i * nu + j] * dt;
        }
        return;
    }
    for (int i = 0; i < nx; ++i) xs[i] = x[i] + 0.5 * f[i] * dt;
    fj(xs, u, f, A, B);
    rk4_chain(nx, nu, 0.5 * dt, A, B, Kx[0], Ku[0], Kx[1], Ku[1]);
    for (int i = 0; i < nx; ++i) xs[i] = x[i] + 0.5 * f[i] * dt;
    fj(xs, u, f, A, B);
    rk4_chain(nx, nu, 0.5 * dt, A, B, Kx[1], Ku[1], Kx[2], Ku[2]);
    for (int i = 0; i < nx; ++i) xs[i] = x[i] + 1.0 * f[i] * dt;
    fj(xs, u, f, A, B);
    rk4_chain(nx, nu, dt, A, B, Kx[2], Ku[2], Kx[3], Ku[3]);
    for (int i = 0; i < nx; ++i) {
        for (int j = 0; j < nx; ++j) {
            int q = i * nx + j;
            Fx[q] = (i == j ? 1.0 : 0.0) + (Kx[0][q] + 2 * Kx[1][q] + 2 * Kx[2][q] + Kx[3][q]) * dt / 6.0;
        }
        for (int j = 0; j < nu; ++j) {
            int q = i * nu + j;
            Fu[q] = (Ku[0][q] + 2 * Ku[1][q] + 2 * Ku[2][q] + Ku[3][q]) * dt / 6.0;
        }
    }
}

/* acrobot.py:109-127 / Quadrotor_EA.py:165-183:
 * 0.5*xerr^T Q xerr + 0.5*uerr^T R uerr, evaluated as ((0.5 e)^T Q) e */
static double quad_form_half(const double *W, const double *e, int n)
{
    double s = 0.0;
    for (int j = 0; j < n; ++j) {
        double t = 0.0;
        for (int i = 0; i < n; ++i) t += (0.5 * e[i]) * W[i * n + j];
        s += t * e[j];
    }
    return s;
}
#define DEF_QUADRATIC_MODEL(P, NX, NU, XDOT, XDOTJAC)                                                         \
    static void P##_dynamics(const double *par, const double *x, const double *u, double *xn)                 \
    {                                                                                                         \
        rk4_step(XDOT, NX, QP_DT(par), QP_EULER(par, NX, NU), x, u, xn);                                                            \
    }                                                                                                         \
    static double P##_cost_stage(const double *par, const double *x, const double *u)                         \
    {                                                                                                         \
        double ex[NX], eu[NU];                                                                                \
        for (int i = 0; i < NX; ++i) ex[i] = x[i] - QP_XREF(par, NX, NU)[i];                                  \
        for (int i = 0; i < NU; ++i) eu[i] = u[i] - QP_UREF(par, NX, NU)[i];                                  \
        return quad_form_half(QP_Q(par, NX, NU), ex, NX) + quad_form_half(QP_R(par, NX, NU), eu, NU);         \
    }                                                                                                         \
    static double P##_cost_final(const double *par, const double *x)                                          \
    {                                                                                                         \
        double ex[NX];                                                                                        \
        for (int i = 0; i < NX; ++i) ex[i] = x[i] - QP_XREF(par, NX, NU)[i];                                  \
        return quad_form_half(QP_QF(par, NX, NU), ex, NX);                                                    \
    }                                                                                                         \
    static void P##_stage_derivs(const double *par, const double *x, const double *u, double *lx, double *lu, \
                                 double *lxx, double *lux, double *luu, double *fx, double *fu)               \
    {                                                                                                         \
        const double *Q = QP_Q(par, NX, NU), *R = QP_R(par, NX, NU);                                          \
        rk4_jac(XDOTJAC, NX, NU, QP_DT(par), QP_EULER(par, NX, NU), x, u, fx, fu);                                                  \
        for (int i = 0; i < NX; ++i) {                                                                        \
            double s = 0.0;                                                                                   \
            for (int j = 0; j < NX; ++j) s += Q[i * NX + j] * (x[j] - QP_XREF(par, NX, NU)[j]);               \
            lx[i] = s;                                                                                        \
        }                                                                                                     \
        for (int i = 0; i < NU; ++i) {                                                                        \
            double s = 0.0;                                                                                   \
            for (int j = 0; j < NU; ++j) s += R[i * NU + j] * (u[j] - QP_UREF(par, NX, NU)[j]);               \
            lu[i] = s;                                                                                        \
        }                                                                                                     \
        memcpy(lxx, Q, sizeof(double) * NX * NX);                                                             \
        memcpy(luu, R, sizeof(double) * NU * NU);                                                             \
        memset(lux, 0, sizeof(double) * NU * NX);                                                             \
    }                                                                                                         \
    static void P##_final_derivs(const double *par, const double *x, double *Vx, double *Vxx)                 \
    {                                                                                                         \
        const double *Qf = QP_QF(par, NX, NU);                                                                \
        for (int i = 0; i < NX; ++i) {                                                                        \
            double s = 0.0;                                                                                   \
            for (int j = 0; j < NX; ++j) s += Qf[i * NX + j] * (x[j] - QP_XREF(par, NX, NU)[j]);              \
            Vx[i] = s;                                                                                        \
        }                                                                                                     \
        memcpy(Vxx, Qf, sizeof(double) * NX * NX);                                                            \
    }                                                                                                         \
    DEF_DIFF_SUB(P##_diff, NX)

DEF_QUADRATIC_MODEL(acrobot, 4, 1, oracle_acrobot_xdot, oracle_acrobot_xdot_jac)
DEF_QUADRATIC_MODEL(quad_ea, 12, 4, oracle_quad_ea_xdot, oracle_quad_ea_xdot_jac)

/* --------------------------- quaternion quadrotor --------------------------- */
/* parameter block: [dt, Q(12 x 12), Qf(12 x 12), R(4 x 4), xref(13), uref(4), euler] */
#define QQ_DT(p) ((p)[0])
#define QQ_Q(p) ((p) + 1)
#define QQ_QF(p) ((p) + 145)
#define QQ_R(p) ((p) + 289)
#define QQ_XREF(p) ((p) + 305)
#define QQ_UREF(p) ((p) + 318)
#define QQ_EULER(p) ((p)[322] != 0.0)
#define QQ_W 177

static void qq_weights(const double *par, int final, double *w)
{
    /* the cost functions of plants_ad.cpp take w = [xref(13), uref(4), Q(12 x 12), R(4 x 4)] */
    memcpy(w, QQ_XREF(par), 13 * sizeof(double));
    memcpy(w + 13, QQ_UREF(par), 4 * sizeof(double));
    memcpy(w + 17, final ? QQ_QF(par) : QQ_Q(par), 144 * sizeof(double));
    memcpy(w + 161, QQ_R(par), 16 * sizeof(double));
}
/* Quadrotor_Quat.py:142-144: Gx = blkdiag(G(q), I9), 13 x 12 */
static void qq_state_jac(const double *x, double *Gx)
{
    double G[12];
    oracle_quad_quat_G(x, G);
    memset(Gx, 0, sizeof(double) * 13 * 12);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j) Gx[i * 12 + j] = G[i * 3 + j];
    for (int i = 0; i < 9; ++i) Gx[(4 + i) * 12 + 3 + i] = 1.0;
}
static void quad_quat_dynamics(const double *par, const double *x, const double *u, double *xn)
{
    rk4_step(oracle_quad_quat_xdot, 13, QQ_DT(par), QQ_EULER(par), x, u, xn);
}
static double quad_quat_cost_stage(const double *par, const double *x, const double *u)
{
    double w[QQ_W], c;
    qq_weights(par, 0, w);
    oracle_quad_quat_cost(w, x, u, &c);
    return c;
}
static double quad_quat_cost_final(const double *par, const double *x)
{
    /* Quadrotor_Quat.py:153: state part only, weights Qf; u = uref zeroes the action term */
    double w[QQ_W], c;
    qq_weights(par, 1, w);
    oracle_quad_quat_cost(w, x, QQ_UREF(par), &c);
    return c;
}
/* Quadrotor_Quat.py:231-238, 248-257: l_* in the 13-dim state, f_x = G^T A G, f_u = G^T B */
static void quad_quat_stage_derivs(const double *par, const double *x, const double *u, double *lx, double *lu,
                                   double *lxx, double *lux, double *luu, double *fx, double *fu)
{
    double w[QQ_W], A[169], B[52], Gx[156], T[156];
    qq_weights(par, 0, w);
    oracle_quad_quat_cost_derivs(w, x, u, lx, lu, lxx);
    memset(lux, 0, sizeof(double) * 4 * 13);
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) luu[i * 4 + j] = 0.5 * (QQ_R(par)[i * 4 + j] + QQ_R(par)[j * 4 + i]);
    rk4_jac(oracle_quad_quat_xdot_jac, 13, 4, QQ_DT(par), QQ_EULER(par), x, u, A, B);
    qq_state_jac(x, Gx);
    /* T = G^T A (12x13); fx = T G (12x12) */
    for (int i = 0; i < 12; ++i)
        for (int j = 0; j < 13; ++j) {
            double s = 0.0;
            for (int t = 0; t < 13; ++t) s += Gx[t * 12 + i] * A[t * 13 + j];
            T[i * 13 + j] = s;
        }
    for (int i = 0; i < 12; ++i)
        for (int j = 0; j < 12; ++j) {
            double s = 0.0;
            for (int t = 0; t < 13; ++t) s += T[i * 13 + t] * Gx[t * 12 + j];
            fx[i * 12 + j] = s;
        }
    for (int i = 0; i < 12; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int t = 0; t < 13; ++t) s += Gx[t * 12 + i] * B[t * 4 + j];
            fu[i * 4 + j] = s;
        }
}
/* Quadrotor_Quat.py:259-265 */
static void quad_quat_final_derivs(const double *par, const double *x, double *Vx, double *Vxx)
{
    double w[QQ_W], vx[13], vu[4], vxx[169], Gx[156], T[156];
    qq_weights(par, 1, w);
    oracle_quad_quat_cost_derivs(w, x, QQ_UREF(par), vx, vu, vxx);
    qq_state_jac(x, Gx);
    for (int i = 0; i < 12; ++i)
        for (int j = 0; j < 13; ++j) {
            double s = 0.0;
            for (int t = 0; t < 13; ++t) s += Gx[t * 12 + i] * vxx[t * 13 + j];
            T[i * 13 + j] = s;
        }
    for (int i = 0; i < 12; ++i)
        for (int j = 0; j < 12; ++j) {
            double s = 0.0;
            for (int t = 0; t < 13; ++t) s += T[i * 13 + t] * Gx[t * 12 + j];
            Vxx[i * 12 + j] = s;
        }
    for (int j = 0; j < 12; ++j) {
        double s = 0.0;
        for (int t = 0; t < 13; ++t) s += vx[t] * Gx[t * 12 + j];
        Vx[j] = s;
    }
}
/* Quadrotor_Quat.py:218-228: G(state)^T (state - ref) */
static void quad_quat_diff(const double *par, const double *s, const double *r, double *dx)
{
    (void)par;
    double Gx[156];
    qq_state_jac(s, Gx);
    for (int j = 0; j < 12; ++j) {
        double a = 0.0;
        for (int t = 0; t < 13; ++t) a += Gx[t * 12 + j] * (s[t] - r[t]);
        dx[j] = a;
    }
}
/* Quadrotor_Quat.py:267-275 */
static void q_terms_quat(const ilqr_oracle_model *M, const double *par, const double *x, const double *lx,
                         const double *lu, const double *lxx, const double *lux, const double *luu,
                         const double *fx, const double *fu, const double *Vx, const double *Vxx, double *Qx,
                         double *Qu, double *Qxx, double *Qux, double *Quu)
{
    double Gx[156], T[156], plx[12], plxx[144], plux[48];
    qq_state_jac(x, Gx);
    for (int j = 0; j < 12; ++j) {
        double s = 0.0;
        for (int t = 0; t < 13; ++t) s += lx[t] * Gx[t * 12 + j];
        plx[j] = s;
    }
    for (int i = 0; i < 12; ++i)
        for (int j = 0; j < 13; ++j) {
            double s = 0.0;
            for (int t = 0; t < 13; ++t) s += Gx[t * 12 + i] * lxx[t * 13 + j];
            T[i * 13 + j] = s;
        }
    for (int i = 0; i < 12; ++i)
        for (int j = 0; j < 12; ++j) {
            double s = 0.0;
            for (int t = 0; t < 13; ++t) s += T[i * 13 + t] * Gx[t * 12 + j];
            plxx[i * 12 + j] = s;
        }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 12; ++j) {
            double s = 0.0;
            for (int t = 0; t < 13; ++t) s += lux[i * 13 + t] * Gx[t * 12 + j];
            plux[i * 12 + j] = s;
        }
    q_terms_default_export(M, par, x, plx, lu, plxx, plux, luu, fx, fu, Vx, Vxx, Qx, Qu, Qxx, Qux, Quu);
}

#ifdef ORACLE_USER_MODEL
/* A plant the user wrote (README.md:25-31 of the reference; derivatives taken symbolically as in
 * control/models/car.py:30-52).  user_gen.h is the traced, printed form of the user's three functions. */
#include "user_gen.h"
static void user_dynamics(const double *par, const double *x, const double *u, double *xn)
{
    user_gen_dynamics(par, x, u, xn);
}
static double user_cost_stage(const double *par, const double *x, const double *u)
{
    double c;
    user_gen_cost_stage(par, x, u, &c);
    return c;
}
static double user_cost_final(const double *par, const double *x)
{
    double c;
    user_gen_cost_final(par, x, &c);
    return c;
}
#ifdef USER_HAS_DIFF
static void user_diff(const double *par, const double *s, const double *r, double *dx) { user_gen_diff_state(par, s, r, dx); }
#else
DEF_DIFF_SUB(user_diff, USER_NX)
#endif
#ifdef USER_HAS_QTERMS
/* the plant's own Q_terms (control/models/Quadrotor_Quat.py:267-275 is the reference's example), as traced */
static void user_q_terms(const ilqr_oracle_model *M, const double *par, const double *x, const double *lx,
                         const double *lu, const double *lxx, const double *lux, const double *luu, const double *fx,
                         const double *fu, const double *Vx, const double *Vxx, double *Qx, double *Qu, double *Qxx,
                         double *Qux, double *Quu)
{
    (void)M;
    user_gen_qterms(par, x, lx, lu, lxx, lux, luu, fx, fu, Vx, Vxx, Qx, Qu, Qxx, Qux, Quu);
}
#define USER_Q_TERMS user_q_terms
#else
#define USER_Q_TERMS q_terms_default_export
#endif
#ifndef USER_NDX
#define USER_NDX USER_NX
#endif
#endif

static const ilqr_oracle_model *oracle_model_table(int id)
{
#ifdef ORACLE_USER_MODEL
    static const ilqr_oracle_model U = {USER_NX, USER_NU, USER_NDX, USER_NP, user_dynamics, user_cost_stage,
                                        user_cost_final, user_gen_stage, user_gen_final, user_diff,
                                        USER_Q_TERMS};
    if (id == 4) return &U;
#endif
    static const ilqr_oracle_model T[4] = {
        {5, 2, 5, 2, car_dynamics, car_cost_stage, car_cost_final, car_stage_derivs, car_final_derivs, car_diff,
         q_terms_default_export},
        {4, 1, 4, 1 + 16 + 16 + 1 + 4 + 1 + 1, acrobot_dynamics, acrobot_cost_stage, acrobot_cost_final,
         acrobot_stage_derivs, acrobot_final_derivs, acrobot_diff, q_terms_default_export},
        {12, 4, 12, 1 + 144 + 144 + 16 + 12 + 4 + 1, quad_ea_dynamics, quad_ea_cost_stage, quad_ea_cost_final,
         quad_ea_stage_derivs, quad_ea_final_derivs, quad_ea_diff, q_terms_default_export},
        {13, 4, 12, 1 + 144 + 144 + 16 + 13 + 4 + 1, quad_quat_dynamics, quad_quat_cost_stage, quad_quat_cost_final,
         quad_quat_stage_derivs, quad_quat_final_derivs, quad_quat_diff, q_terms_quat},
    };
    return (id >= 0 && id < 4) ? &T[id] : 0;
}
#endif
