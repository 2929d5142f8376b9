/* acados_shim: included by bluerov2_dob.h:30; nothing from it is used by the callers. */
#ifndef BROV_SHIM_BLASFEO_D_AUX_H_
#define BROV_SHIM_BLASFEO_D_AUX_H_
#endif
