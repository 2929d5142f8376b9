/* acados_shim/acados_c/ocp_nlp_interface.h -- the subset of acados' C interface that the BlueROV2 callers use, backed by
 * the MI355X batched solver (batch = 1).  Call sites this serves (all under /root/reference/bluerov2_dobmpc):
 *   src/bluerov2_dob.cpp:320-321,370-388   src/ctrller/mpc.cpp:46-78,121-137   src/bluerov2_ampc.cpp:320-411
 *   scripts/c_generated_code/main_bluerov2.c:117-247   scripts/c_generated_code/acados_solver_bluerov2.c:1001-1010
 * Struct layouts are the shim's own; only the members the callers touch are promised:
 *   ocp_nlp_out::inf_norm_res (bluerov2_dob.cpp:384, mpc.cpp:73), ocp_nlp_dims::N (main_bluerov2.c:225),
 *   ocp_nlp_plan_t::N (acados_solver_bluerov2.c:113). */
#ifndef BROV_SHIM_OCP_NLP_INTERFACE_H_
#define BROV_SHIM_OCP_NLP_INTERFACE_H_
#include "acados/utils/types.h"
#ifdef __cplusplus
extern "C" {
#endif

struct brov_shim_state; /* opaque: owns the GPU solver handle and the host mirrors */

typedef struct ocp_nlp_plan_t { int N; struct brov_shim_state* shim; } ocp_nlp_plan_t;
typedef struct ocp_nlp_config { int N; struct brov_shim_state* shim; } ocp_nlp_config;
typedef struct ocp_nlp_dims { int N; int nx, nu, np, ny, nyn; struct brov_shim_state* shim; } ocp_nlp_dims;
typedef struct ocp_nlp_in { struct brov_shim_state* shim; } ocp_nlp_in;
typedef struct ocp_nlp_out {
    double inf_norm_res; /* KKT inf-norm of the iterate that entered the last solve (computed on the GPU) */
    struct brov_shim_state* shim;
} ocp_nlp_out;
typedef struct ocp_nlp_solver { struct brov_shim_state* shim; } ocp_nlp_solver;

ACADOS_SYMBOL_EXPORT int ocp_nlp_constraints_model_set(ocp_nlp_config* config, ocp_nlp_dims* dims, ocp_nlp_in* in, int stage,
                                                       const char* field, void* value);
ACADOS_SYMBOL_EXPORT int ocp_nlp_cost_model_set(ocp_nlp_config* config, ocp_nlp_dims* dims, ocp_nlp_in* in, int stage,
                                                const char* field, void* value);
ACADOS_SYMBOL_EXPORT int ocp_nlp_in_set(ocp_nlp_config* config, ocp_nlp_dims* dims, ocp_nlp_in* in, int stage,
                                        const char* field, void* value);
ACADOS_SYMBOL_EXPORT void ocp_nlp_out_set(ocp_nlp_config* config, ocp_nlp_dims* dims, ocp_nlp_out* out, int stage,
                                          const char* field, void* value);
ACADOS_SYMBOL_EXPORT void ocp_nlp_out_get(ocp_nlp_config* config, ocp_nlp_dims* dims, ocp_nlp_out* out, int stage,
                                          const char* field, void* value);
ACADOS_SYMBOL_EXPORT void ocp_nlp_get(ocp_nlp_config* config, ocp_nlp_solver* solver, const char* field, void* value);
ACADOS_SYMBOL_EXPORT int ocp_nlp_solver_opts_set(ocp_nlp_config* config, void* opts, const char* field, void* value);
ACADOS_SYMBOL_EXPORT int ocp_nlp_solve(ocp_nlp_solver* solver, ocp_nlp_in* in, ocp_nlp_out* out);
ACADOS_SYMBOL_EXPORT int ocp_nlp_precompute(ocp_nlp_solver* solver, ocp_nlp_in* in, ocp_nlp_out* out);

#ifdef __cplusplus
}
#endif
#endif
