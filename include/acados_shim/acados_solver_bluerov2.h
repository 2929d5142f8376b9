/* acados_shim/acados_solver_bluerov2.h -- drop-in for the reference's generated solver header
 * (/root/reference/bluerov2_dobmpc/scripts/c_generated_code/acados_solver_bluerov2.h:42-167): same macros, same capsule
 * member names, same entry points; implemented by libacados_ocp_solver_bluerov2.so of THIS repository on top of the
 * MI355X batched solver (include/bluerov2_nmpc.h) with batch = 1.  See INTEGRATION.md.
 *
 * Where the drop-in has a choice the generated code does not state, it follows upstream acados, and each
 * choice has an opt-out in the environment, read once at bluerov2_acados_create:
 *   return code    0 = step taken (also when the QP stopped at qp_iter_max: SQP_RTI returns ACADOS_SUCCESS there; "qp_status" /
 *                  "statistics" / print_stats keep the QP's 2), 1 = NaN, 3 / 4 = step not taken.  BROV_SHIM_MAXITER_STATUS=2
 *                  returns the 2 instead.
 *   failed step    the iterate stays as it was (acados returns before update_variables) and ocp_nlp_out_get(.., 0, "u") holds the
 *                  last successfully computed input, clamped into the box.  BROV_ON_FAILURE=restart cold-starts the iterate at
 *                  the measured state instead.
 *   rti_phase 2    without a preparation of the current iterate: preparation + feedback in one call (acados would reuse the QP in
 *                  its memory; the batched brov_solve_phase refuses).
 *   BROV_DEVICE=<i> selects the GPU (default 0); BROV_SHIM_TIMING=1 records kernel times ("time_lin" / "time_qp_sol") from the
 *                  first tick on (default: from the first request on; "time_tot" is always the host wall time of the call). */
#ifndef ACADOS_SOLVER_bluerov2_H_
#define ACADOS_SOLVER_bluerov2_H_

#include "acados/utils/types.h"
#include "acados_c/ocp_nlp_interface.h"
#include "acados_c/external_function_interface.h"

/* dimensions of the OCP (bluerov2.py:24-59, generate_c_code.py:17-27) */
#define BLUEROV2_NX 12
#define BLUEROV2_NZ 0
#define BLUEROV2_NU 4
#define BLUEROV2_NP 16
#define BLUEROV2_NBX 0
#define BLUEROV2_NBX0 12
#define BLUEROV2_NBU 4
#define BLUEROV2_NSBX 0
#define BLUEROV2_NSBU 0
#define BLUEROV2_NSH 0
#define BLUEROV2_NSG 0
#define BLUEROV2_NSPHI 0
#define BLUEROV2_NSHN 0
#define BLUEROV2_NSGN 0
#define BLUEROV2_NSPHIN 0
#define BLUEROV2_NSBXN 0
#define BLUEROV2_NS 0
#define BLUEROV2_NSN 0
#define BLUEROV2_NG 0
#define BLUEROV2_NBXN 0
#define BLUEROV2_NGN 0
#define BLUEROV2_NY0 16
#define BLUEROV2_NY 16
#define BLUEROV2_NYN 12
#ifndef BLUEROV2_N
#define BLUEROV2_N 80 /* horizon baked into the reference's generated code; other N via create_with_discretization */
#endif
#define BLUEROV2_NH 0
#define BLUEROV2_NPHI 0
#define BLUEROV2_NHN 0
#define BLUEROV2_NPHIN 0
#define BLUEROV2_NR 0

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bluerov2_solver_capsule {
    /* the members the callers dereference (bluerov2_dob.cpp:320-388, mpc.cpp:46-78) */
    ocp_nlp_in* nlp_in;
    ocp_nlp_out* nlp_out;
    ocp_nlp_out* sens_out;
    ocp_nlp_solver* nlp_solver;
    void* nlp_opts;
    ocp_nlp_plan_t* nlp_solver_plan;
    ocp_nlp_config* nlp_config;
    ocp_nlp_dims* nlp_dims;
    unsigned int nlp_np;
    /* per-stage parameter views, kept for source compatibility with code that walks them */
    external_function_param_casadi* forw_vde_casadi;
    external_function_param_casadi* expl_ode_fun;
    /* shim state (GPU solver handle, host mirrors) */
    struct brov_shim_state* shim;
} bluerov2_solver_capsule;

ACADOS_SYMBOL_EXPORT bluerov2_solver_capsule* bluerov2_acados_create_capsule(void);
ACADOS_SYMBOL_EXPORT int bluerov2_acados_free_capsule(bluerov2_solver_capsule* capsule);
ACADOS_SYMBOL_EXPORT int bluerov2_acados_create(bluerov2_solver_capsule* capsule);
ACADOS_SYMBOL_EXPORT int bluerov2_acados_reset(bluerov2_solver_capsule* capsule, int reset_qp_solver_mem);
ACADOS_SYMBOL_EXPORT int bluerov2_acados_create_with_discretization(bluerov2_solver_capsule* capsule, int n_time_steps,
                                                                    double* new_time_steps);
ACADOS_SYMBOL_EXPORT int bluerov2_acados_update_time_steps(bluerov2_solver_capsule* capsule, int N, double* new_time_steps);
ACADOS_SYMBOL_EXPORT int bluerov2_acados_update_qp_solver_cond_N(bluerov2_solver_capsule* capsule, int qp_solver_cond_N);
ACADOS_SYMBOL_EXPORT int bluerov2_acados_update_params(bluerov2_solver_capsule* capsule, int stage, double* value, int np);
ACADOS_SYMBOL_EXPORT int bluerov2_acados_update_params_sparse(bluerov2_solver_capsule* capsule, int stage, int* idx, double* p,
                                                              int n_update);
ACADOS_SYMBOL_EXPORT int bluerov2_acados_solve(bluerov2_solver_capsule* capsule);
ACADOS_SYMBOL_EXPORT int bluerov2_acados_free(bluerov2_solver_capsule* capsule);
ACADOS_SYMBOL_EXPORT void bluerov2_acados_print_stats(bluerov2_solver_capsule* capsule);
ACADOS_SYMBOL_EXPORT int bluerov2_acados_custom_update(bluerov2_solver_capsule* capsule, double* data, int data_len);

ACADOS_SYMBOL_EXPORT ocp_nlp_in* bluerov2_acados_get_nlp_in(bluerov2_solver_capsule* capsule);
ACADOS_SYMBOL_EXPORT ocp_nlp_out* bluerov2_acados_get_nlp_out(bluerov2_solver_capsule* capsule);
ACADOS_SYMBOL_EXPORT ocp_nlp_out* bluerov2_acados_get_sens_out(bluerov2_solver_capsule* capsule);
ACADOS_SYMBOL_EXPORT ocp_nlp_solver* bluerov2_acados_get_nlp_solver(bluerov2_solver_capsule* capsule);
ACADOS_SYMBOL_EXPORT ocp_nlp_config* bluerov2_acados_get_nlp_config(bluerov2_solver_capsule* capsule);
ACADOS_SYMBOL_EXPORT void* bluerov2_acados_get_nlp_opts(bluerov2_solver_capsule* capsule);
ACADOS_SYMBOL_EXPORT ocp_nlp_dims* bluerov2_acados_get_nlp_dims(bluerov2_solver_capsule* capsule);
ACADOS_SYMBOL_EXPORT ocp_nlp_plan_t* bluerov2_acados_get_nlp_plan(bluerov2_solver_capsule* capsule);

#ifdef __cplusplus
}
#endif
#endif /* ACADOS_SOLVER_bluerov2_H_ */
