/* acados_shim: MIN/MAX as used by c_generated_code/main_bluerov2.c:220 */
#ifndef BROV_SHIM_ACADOS_UTILS_MATH_H_
#define BROV_SHIM_ACADOS_UTILS_MATH_H_
#ifndef MIN
#define MIN(a, b) (((a) < (b)) ? (a) : (b))
#endif
#ifndef MAX
#define MAX(a, b) (((a) > (b)) ? (a) : (b))
#endif
#endif
