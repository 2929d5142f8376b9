/* acados_shim: the generated capsule (acados_solver_bluerov2.h:79-127) names this type; the MI355X solver evaluates the
 * model inside its HIP kernels, so the struct only keeps the per-stage parameter vector for introspection. */
#ifndef BROV_SHIM_EXTERNAL_FUNCTION_INTERFACE_H_
#define BROV_SHIM_EXTERNAL_FUNCTION_INTERFACE_H_
#ifdef __cplusplus
extern "C" {
#endif
typedef struct external_function_param_casadi {
    double* p; /* np parameters of the stage (host copy) */
    int np;
} external_function_param_casadi;
#ifdef __cplusplus
}
#endif
#endif
