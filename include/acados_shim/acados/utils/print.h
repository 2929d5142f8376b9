/* acados_shim: the callers include acados/utils/print.h but use nothing from it besides the types below. */
#ifndef BROV_SHIM_ACADOS_UTILS_PRINT_H_
#define BROV_SHIM_ACADOS_UTILS_PRINT_H_
#include "acados/utils/types.h"
#endif
