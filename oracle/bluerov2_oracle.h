/*
 * bluerov2_oracle.h -- CPU restatement (plain C, FP64) of the BlueROV2 NMPC real-time-iteration hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it, and only as the checker / reported CPU baseline.  The product path is
 * bluerov2_amd/csrc (HIP, gfx950) behind include/bluerov2_nmpc.h and fails loudly without a GPU.
 *
 * What it restates (reference = HKPolyU-UAV/bluerov2 @ /root/reference):
 *   - model f(x,u,p):          bluerov2_dobmpc/scripts/bluerov2.py:77-137
 *   - forward sensitivities:   bluerov2_dobmpc/scripts/c_generated_code/bluerov2_model/bluerov2_expl_vde_forw.c:73-4633
 *                              (incl. d|v|v/dv = sign(v)v+|v| with sign(0)=0, :65)
 *   - ERK4, 1 step/interval:   c_generated_code/acados_solver_bluerov2.c:633-641 (num_stages 4, num_steps 1)
 *   - NLS cost, W, scaling Ts: c_generated_code/acados_solver_bluerov2.c:389-493
 *   - bounds / x0 embedding:   c_generated_code/acados_solver_bluerov2.c:501-573
 *   - SQP_RTI full step:       c_generated_code/acados_solver_bluerov2.c:623-672
 *
 * PARITY PINNING.  The model layer (f, A, B, RK4 step) is pinned against the reference's own CasADi-generated C,
 * compiled from /root/reference into oracle/_ref (see oracle/Makefile, tests/test_oracle_model.py and the committed
 * vectors tests/golden/model_vectors.npz).  The solver layer (acados SQP_RTI + HPIPM) is a third-party dependency
 * that is NOT vendored under /root/reference (README.md:41-55 clones acados master, unpinned), is not installed
 * here, and the reference has no tests or recorded outputs:  **solver-level parity is unpinned**.  It is anchored
 * instead on (a) the unique minimiser of the strictly convex QP, cross-checked by an independent condensed
 * bounded-least-squares solve (scripts/make_golden.py -> tests/golden/rti_known_answers.npz), and (b) KKT residuals.
 */
#ifndef BLUEROV2_ORACLE_H_
#define BLUEROV2_ORACLE_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NX 12
#define ORC_NU 4
#define ORC_NP 16
#define ORC_NY 16

/* Solver options; defaults mirror c_generated_code/acados_solver_bluerov2.c (weights :422-481, bounds :559-566,
 * qp_iter_max :668) */
typedef struct orc_opts {
    int    N;             /* shooting intervals (reference: 80, BASELINE headline: 20) */
    double Ts;            /* uniform step, Tf/N (reference 0.0125) */
    double W[ORC_NY];     /* diag of stage weight [x;u] */
    double We[ORC_NX];    /* diag of terminal weight */
    double lbu[ORC_NU], ubu[ORC_NU];
    int    qp_iter_max;   /* 50 */
    double qp_tol_mu;     /* bound resolution: input within this distance of a bound, or its multiplier / input weight below it (1e-7) */
    double qp_tol_stat;   /* stationarity target of the QP (tracked residual, absolute) */
    int    qp_early_exit; /* 1: return the equality-constrained minimiser when it is feasible (exact) */
    const double* ts_vec; /* NULL: uniform grid, step Ts.  Else N time steps (bluerov2_acados_create_with_discretization / _update_time_steps,
                           * acados_solver_bluerov2.c:111-131,375-387): ERK4 step AND cost scaling of stage i = ts_vec[i] */
    const double* W0;     /* NULL: stage 0 weighs like the other stages.  Else the 16 diagonal entries of W_0 (the generated solver keeps a
                           * separate stage-0 weight, acados_solver_bluerov2.c:422-441, set to the same numbers) */
    int    on_failure;    /* failed step (status 1/3/4): 0 keep the iterate (acados: SQP_RTI returns before update_variables),
                           * 1 cold restart at the measured state if it is finite (x_i = x0, u = 0, multipliers 0).  Either way the record's u0
                           * holds the last successfully computed input, clamped to the bounds, NaN -> 0. */
} orc_opts;

void orc_default_opts(orc_opts* o, int N, double Ts);

/* model ------------------------------------------------------------------------------------------------ */
void orc_f(const double* x, const double* u, const double* p, double* xdot);
/* continuous-time Jacobians, row-major A[12][12], B[12][4] */
void orc_jac(const double* x, const double* u, const double* p, double* A, double* B);
/* one explicit RK4 step of length h with forward sensitivities (seed Sx=I,Su=0);
 * xn[12], A[12][12] = d xn/d x, B[12][4] = d xn/d u, row-major */
void orc_rk4_sens(const double* x, const double* u, const double* p, double h, double* xn, double* A, double* B);
/* RK4 step without sensitivities (plant simulation) */
void orc_rk4(const double* x, const double* u, const double* p, double h, double* xn);
/* the 6-disturbance model variant (bluerov2.py:37-38 carries the two symbols commented out, the EKF estimates all
 * six, bluerov2_dob.h:200-205): drp = {d_phi, d_theta} roll / pitch disturbance moments, entering dp += d_phi / Ix, dq += d_theta / Iy
 * the way the other four enter their rows (bluerov2.py:123-128).  drp == NULL is the shipped np = 16 model. */
void orc_f6(const double* x, const double* u, const double* p, const double* drp, double* xdot);
void orc_rk4_sens6(const double* x, const double* u, const double* p, const double* drp, double h, double* xn, double* A, double* B);
void orc_rk4_6(const double* x, const double* u, const double* p, const double* drp, double h, double* xn);

/* QP ------------------------------------------------------------------------------------------------- */
/* box-constrained OCP QP (diagonal cost) in delta variables:
 *   min sum_i 1/2 dx_i'Qd_i dx_i + q_i'dx_i + 1/2 du_i'Rd_i du_i + r_i'du_i  (+ terminal)
 *   s.t. dx_{i+1} = A_i dx_i + B_i du_i + b_i, dx_0 = d0, lb_i <= du_i <= ub_i
 * Solved by a primal-dual interior point method whose Newton systems are solved by a Riccati sweep.
 * Outputs dx[(N+1)*12], du[N*4], pi[N*12] (multiplier of the i-th dynamics equation), lam[N*8] ([lower4, upper4]).
 * stats[0]=Newton systems solved (interior-point iterations + active-set tries), [1]=final mu, [2]=QP complementarity residual (after multiplier recovery), [3]=1 if early exit.
 * returns 0 ok, 2 max iter, 1 NaN, 4 factorisation failure  (acados status codes) */
int orc_qp_solve(const orc_opts* o, const double* A, const double* B, const double* b, const double* Qd,
                 const double* q, const double* Rd, const double* r, const double* d0, const double* lb,
                 const double* ub, double* dx, double* du, double* pi, double* lam, double* stats);

/* doubles of scratch orc_qp_solve_ws / orc_rti_step_ws need for horizon N */
size_t orc_ws_doubles(int N);
int orc_qp_solve_ws(const orc_opts* o, const double* A, const double* B, const double* b, const double* Qd,
                    const double* q, const double* Rd, const double* r, const double* d0, const double* lb,
                    const double* ub, double* dx, double* du, double* pi, double* lam, double* stats, double* mem);

/* RTI ------------------------------------------------------------------------------------------------ */
/* per-instance result record, 104 bytes on the wire */
typedef struct orc_result {
    double u0[ORC_NU]; /* in/out: a failed step holds the value passed in (last successful input) */
    double cost;      /* NLS objective at the updated iterate (failed step: at the entering iterate) */
    double kkt;       /* NLP KKT inf-norm at the iterate ENTERING this step (with the stored multipliers) */
    int    status;
    int    qp_iter;
    double thrust[6]; /* thrust allocation of u0 (bluerov2_dob.cpp:390-395) */
} orc_result;

/* One SQP-RTI step (preparation + feedback) for one OCP instance.
 * in:  x0[12], yref[(N+1)*16], p[(N+1)*16]; in/out iterate: x[(N+1)*12], u[N*4], pi[N*12], lam[N*8].
 * res is in/out: res->u0 must hold the previously applied input (zeros at start) -- a failed step keeps it.
 * optional out (may be NULL): Aout[N*144], Bout[N*48], bout[N*12], qp_stats[4] */
int orc_rti_step(const orc_opts* o, const double* x0, const double* yref, const double* p, double* x, double* u,
                 double* pi, double* lam, orc_result* res, double* Aout, double* Bout, double* bout,
                 double* qp_stats);

/* the same with a caller-provided workspace (orc_ws_doubles(N) doubles) and the optional roll / pitch disturbances drp[(N+1)*2] */
int orc_rti_step_ws(const orc_opts* o, const double* x0, const double* yref, const double* p, const double* drp, double* x,
                    double* u, double* pi, double* lam, orc_result* res, double* Aout, double* Bout, double* bout,
                    double* qp_stats, double* mem);
int orc_rti_step6(const orc_opts* o, const double* x0, const double* yref, const double* p, const double* drp, double* x,
                  double* u, double* pi, double* lam, orc_result* res, double* qp_stats);
int orc_rti_step_batch6(const orc_opts* o, int nb, const double* x0, const double* yref, const double* p, const double* drp,
                        double* x, double* u, double* pi, double* lam, orc_result* res, int nthreads);

/* nb independent instances, instance-major contiguous arrays; nthreads<=0 -> all cores (OpenMP) */
int orc_rti_step_batch(const orc_opts* o, int nb, const double* x0, const double* yref, const double* p,
                       double* x, double* u, double* pi, double* lam, orc_result* res, int nthreads);

/* create-default iterate: x_i=[0,0,-20,0...], u_i=0 (acados_solver_bluerov2.c:681-708) */
void orc_init_iterate(const orc_opts* o, double* x, double* u, double* pi, double* lam);

/* thrust allocation (bluerov2_dob.cpp:390-395) */
void orc_thrust_alloc(const double* u0, double* t6);

int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
