/* acados_shim: bluerov2_dob.h:33 includes the CasADi model header; with the MI355X solver the model lives in
 * bluerov2_amd/csrc/bluerov2_model.hpp (HIP) and nothing is exported to the callers. */
#ifndef BROV_SHIM_BLUEROV2_MODEL_H_
#define BROV_SHIM_BLUEROV2_MODEL_H_
#endif
