/* tests/san_batch_driver.c -- the oracle's batch driver (OpenMP over instances) and the EKF oracle under the sanitizers
 * (tests/test_sanitizers.py builds this with -fsanitize=address,undefined -fopenmp).  Input file: N, nb, nticks, nthreads as doubles, then
 * x0[nb][12], p[16], yref[nticks][N+1][16]; output: the records of every tick as text (u0[4] cost kkt status qp_iter per instance). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../oracle/bluerov2_oracle.h"

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    double hd[4];
    if (fread(hd, sizeof(double), 4, f) != 4) return 2;
    const int N = (int)hd[0], nb = (int)hd[1], nt = (int)hd[2], nth = (int)hd[3];
    const size_t n1 = (size_t)N + 1;
    double* x0 = (double*)malloc((size_t)nb * 12 * sizeof(double));
    double p16[16];
    double* yw = (double*)malloc(n1 * 16 * sizeof(double));
    if (fread(x0, sizeof(double), (size_t)nb * 12, f) != (size_t)nb * 12 || fread(p16, sizeof(double), 16, f) != 16) return 2;
    double* yref = (double*)malloc((size_t)nb * n1 * 16 * sizeof(double));
    double* p = (double*)malloc((size_t)nb * n1 * 16 * sizeof(double));
    double* x = (double*)malloc((size_t)nb * n1 * 12 * sizeof(double));
    double* u = (double*)malloc((size_t)nb * N * 4 * sizeof(double));
    double* pi = (double*)malloc((size_t)nb * N * 12 * sizeof(double));
    double* lam = (double*)malloc((size_t)nb * N * 8 * sizeof(double));
    orc_result* res = (orc_result*)calloc((size_t)nb, sizeof(orc_result));
    orc_opts o;
    orc_default_opts(&o, N, 1.0 / N);
    for (int b = 0; b < nb; b++) {
        orc_init_iterate(&o, x + (size_t)b * n1 * 12, u + (size_t)b * N * 4, pi + (size_t)b * N * 12, lam + (size_t)b * N * 8);
        for (size_t i = 0; i < n1; i++) memcpy(p + ((size_t)b * n1 + i) * 16, p16, sizeof p16);
    }
    for (int k = 0; k < nt; k++) {
        if (fread(yw, sizeof(double), n1 * 16, f) != n1 * 16) return 2;
        for (int b = 0; b < nb; b++) memcpy(yref + (size_t)b * n1 * 16, yw, n1 * 16 * sizeof(double));
        const int worst = orc_rti_step_batch(&o, nb, x0, yref, p, x, u, pi, lam, res, nth);
        printf("TICK %d worst %d\n", k, worst);
        for (int b = 0; b < nb; b++)
            printf("REC %d %.17g %.17g %.17g %.17g %.17g %.17g %d %d\n", b, res[b].u0[0], res[b].u0[1], res[b].u0[2], res[b].u0[3], res[b].cost,
                   res[b].kkt, res[b].status, res[b].qp_iter);
    }
    fclose(f);
    free(x0); free(yw); free(yref); free(p); free(x); free(u); free(pi); free(lam); free(res);
    return 0;
}
