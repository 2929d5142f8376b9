/* acados_shim: minimal stand-in for acados/utils/types.h -- only what the BlueROV2 callers and the generated-solver
 * header use (return codes as in acados: /root/reference/.../acados_solver_bluerov2.c:726 ACADOS_SUCCESS). */
#ifndef BROV_SHIM_ACADOS_UTILS_TYPES_H_
#define BROV_SHIM_ACADOS_UTILS_TYPES_H_
#ifdef __cplusplus
extern "C" {
#endif
typedef double real_t;
typedef int int_t;
#ifndef ACADOS_SYMBOL_EXPORT
#define ACADOS_SYMBOL_EXPORT __attribute__((visibility("default")))
#endif
enum return_values {
    ACADOS_SUCCESS = 0,
    ACADOS_NAN_DETECTED = 1,
    ACADOS_MAXITER = 2,
    ACADOS_MINSTEP = 3,
    ACADOS_QP_FAILURE = 4,
    ACADOS_READY = 5
};
#ifdef __cplusplus
}
#endif
#endif
