/* TEST INFRASTRUCTURE ONLY -- interface of the CPU checker (see ilqr_oracle.c). */
#ifndef ILQR_ORACLE_H
#define ILQR_ORACLE_H
#include <stdint.h>

#define ILQR_MAX_NX 16
#define ILQR_MAX_NDX 16
#define ILQR_MAX_NU 4
#define ILQR_TRACE_MAX 64
#ifndef ILQR_PIVOT_TOL
#define ILQR_PIVOT_TOL 1e-13
#endif

/* model ids and parameter-block layouts (doubles):
 *   0 car        [r, v_target]
 *   1 acrobot    [dt, Q(4x4), Qf(4x4), R(1x1), xref(4), uref(1)]
 *   2 quad_ea    [dt, Q(12x12), Qf(12x12), R(4x4), xref(12), uref(4)]
 *   3 quad_quat  [dt, qdiag(12), qfdiag(12), rdiag(4), xref(13), uref(4)]
 */
enum { ILQR_MODEL_CAR = 0, ILQR_MODEL_ACROBOT = 1, ILQR_MODEL_QUAD_EA = 2, ILQR_MODEL_QUAD_QUAT = 3 };

/* why run_ilqr returned (control/iLQRController.py:43, :50-51, and the
 * LinAlgError that np.linalg.eigvals / inv raise on non-finite / singular input) */
enum { ILQR_EXIT_DTOL = 0, ILQR_EXIT_SMALL_DELTAJ = 1, ILQR_EXIT_MAXITER = 2, ILQR_EXIT_LINALG = 3 };

typedef struct ilqr_oracle_model ilqr_oracle_model;
struct ilqr_oracle_model {
    int nx, nu, ndx, npar;
    void (*dynamics)(const double *par, const double *x, const double *u, double *xn);
    double (*cost_stage)(const double *par, const double *x, const double *u);
    double (*cost_final)(const double *par, const double *x);
    /* l_x[nx] l_u[nu] l_xx[nx*nx] l_ux[nu*nx] l_uu[nu*nu] f_x[ndx*ndx] f_u[ndx*nu] */
    void (*stage_derivs)(const double *par, const double *x, const double *u, double *lx, double *lu, double *lxx,
                         double *lux, double *luu, double *fx, double *fu);
    void (*final_derivs)(const double *par, const double *x, double *Vx, double *Vxx); /* ndx-sized */
    void (*diff_state)(const double *par, const double *state, const double *ref, double *dx);
    void (*q_terms)(const ilqr_oracle_model *M, const double *par, const double *x, const double *lx,
                    const double *lu, const double *lxx, const double *lux, const double *luu, const double *fx,
                    const double *fu, const double *Vx, const double *Vxx, double *Qx, double *Qu, double *Qxx,
                    double *Qux, double *Quu);
};

typedef struct {
    double J0;                       /* cost of the initial rollout */
    double J_last;                   /* the solver's J variable at return (iLQRController.py:79) */
    double dmax;                     /* max_k |d_k|_2 at return */
    double max_abs_q;                /* largest |Q_ij| seen in any backward stage (conditioning flag) */
    double J[ILQR_TRACE_MAX];        /* Jn after the line search of iteration i */
    double deltaJ[ILQR_TRACE_MAX];   /* expected reduction of iteration i */
    int32_t ls_evals[ILQR_TRACE_MAX];
    uint8_t accepted[ILQR_TRACE_MAX];
    int32_t iters;                   /* backward passes run */
    int32_t status;                  /* ILQR_EXIT_* */
    int32_t regu_rounds;             /* total regularisation rounds */
    int32_t ls_total;                /* total forward-pass evaluations */
    int32_t regu_stages;             /* backward stages that needed regularisation */
    int32_t pad_;
} ilqr_oracle_info;

int ilqr_oracle_solve(const ilqr_oracle_model *M, const double *par, int N, int max_iter, int max_ls, double d_tol,
                      const double *x0, double *u, double *x, ilqr_oracle_info *info);
void ilqr_oracle_set_cost_N(int cost_N);   /* knot points cost_trj sums over; 0 = the trajectory's length */
int ilqr_oracle_solve_batch(int model_id, const double *par, int N, int max_iter, int max_ls, double d_tol, long B,
                            const double *x0, double *u, double *x, ilqr_oracle_info *info, int nthreads);
const ilqr_oracle_model *ilqr_oracle_get_model(int model_id);
int ilqr_oracle_has_negative_eig(const double *Q, int w, double *W);      /* the selected rule */
int ilqr_oracle_has_negative_eig_lapack(const double *Q, int w, double *W); /* LAPACK restated (default) */
void ilqr_oracle_set_psd_rule(int rule);                                  /* 0 LAPACK restated, 1 LDL^T */
int ilqr_oracle_has_negative_eig_ldl(const double *Q, int w, double *W);  /* the LDL^T rule, for scoring */
void ilqr_oracle_model_dims(int model_id, int *nx, int *nu, int *ndx, int *npar);
/* Diagnostics: copies the histogram of regularisation rounds per backward stage into out64 (if not NULL),
 * then, when enable >= 0, clears it and switches the counting on (1) or off (0). */
void ilqr_oracle_regu_hist(int enable, long* out64);

#endif
