/* acados_shim: the two BLASFEO print helpers the generated example uses (main_bluerov2.c:229-231). */
#ifndef BROV_SHIM_BLASFEO_D_AUX_EXT_DEP_H_
#define BROV_SHIM_BLASFEO_D_AUX_EXT_DEP_H_
#ifdef __cplusplus
extern "C" {
#endif
void d_print_mat(int m, int n, double* A, int lda);
void d_print_exp_mat(int m, int n, double* A, int lda);
void d_print_tran_mat(int row, int col, double* A, int lda);
void d_print_exp_tran_mat(int row, int col, double* A, int lda);
#ifdef __cplusplus
}
#endif
#endif
