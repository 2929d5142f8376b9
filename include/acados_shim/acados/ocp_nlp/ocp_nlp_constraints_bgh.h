/* acados_shim: included by bluerov2_dobmpc/include/bluerov2_dobmpc/bluerov2_dob.h:27-28; nothing from it is used. */
#ifndef BROV_SHIM_ACADOS_OCP_NLP_OCP_NLP_CONSTRAINTS_BGH_H_
#define BROV_SHIM_ACADOS_OCP_NLP_OCP_NLP_CONSTRAINTS_BGH_H_
#include "acados/utils/types.h"
#endif
