/* dev tool: many control ticks through the acados-shaped drop-in at batch 1 (N = 80): the host mailbox's sequence words, the two
 * alternating hand-out counters, the resident windowed kernel with its helper waves -- statuses, finite outputs, worst tick.
 *   gcc -O2 -Iinclude/acados_shim -o /tmp/shim_soak scripts/dev/shim_soak.c -Lbluerov2_amd/lib -lacados_ocp_solver_bluerov2 -Wl,-rpath,$PWD/bluerov2_amd/lib -lm
 *   /tmp/shim_soak [ticks] */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "acados_c/ocp_nlp_interface.h"
#include "acados_solver_bluerov2.h"

int main(int argc, char** argv) {
    const long T = argc > 1 ? atol(argv[1]) : 300000;
    bluerov2_solver_capsule* c = bluerov2_acados_create_capsule();
    if (bluerov2_acados_create(c)) return 1;
    static double yref[BLUEROV2_N + 1][BLUEROV2_NY], par[BLUEROV2_N + 1][BLUEROV2_NP];
    const double pn[16] = {0, 0, 0, 0, 1.7182, 0, 5.468, 0.4006, -11.7391, -20, -31.8678, -5, -18.18, -21.66, -36.99, -1.55};
    double x0[12] = {-2, 0, -20, 0, 0, -1.5707963267948966, 0, 0, 0, 0, 0, 0};
    long bad = 0, nonfinite = 0, loops = 0;
    double worst = 0, sum = 0;
    for (long k = 0; k < T; k++) {
        for (int i = 0; i <= BLUEROV2_N; i++) {
            const double t = 0.05 * (double)(k + i), w = 0.75;   /* (the yaw reference grows without bound, as on the vehicle) */
            for (int j = 0; j < 16; j++) { yref[i][j] = 0.0; par[i][j] = pn[j]; }
            yref[i][0] = -2 * cos(w * t); yref[i][1] = -2 * sin(w * t); yref[i][2] = -20; yref[i][5] = w * t - 1.5707963267948966;
            yref[i][6] = 1.5;
            par[i][0] = 3.0 * sin(0.001 * (double)k);   /* a slowly varying disturbance estimate */
        }
        if (k % 97 == 0) { x0[0] += 4.0; x0[2] -= 3.0; x0[6] = 2.0; }   /* now and then the state jumps: inputs saturate, the QP loop runs */
        struct timespec a, b;
        clock_gettime(CLOCK_MONOTONIC, &a);
        ocp_nlp_constraints_model_set(c->nlp_config, c->nlp_dims, c->nlp_in, 0, "lbx", x0);
        ocp_nlp_constraints_model_set(c->nlp_config, c->nlp_dims, c->nlp_in, 0, "ubx", x0);
        for (int i = 0; i <= BLUEROV2_N; i++) bluerov2_acados_update_params(c, i, par[i], BLUEROV2_NP);
        for (int i = 0; i <= BLUEROV2_N; i++) ocp_nlp_cost_model_set(c->nlp_config, c->nlp_dims, c->nlp_in, i, "yref", yref[i]);
        const int st = bluerov2_acados_solve(c);
        double u0[4];
        int qi = 0;
        ocp_nlp_out_get(c->nlp_config, c->nlp_dims, c->nlp_out, 0, "u", u0);
        ocp_nlp_get(c->nlp_config, c->nlp_solver, "qp_iter", &qi);
        clock_gettime(CLOCK_MONOTONIC, &b);
        const double us = (b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3;
        if (k > 20) { sum += us; if (us > worst) worst = us; }
        if (st != 0) bad++;
        if (qi > 0) loops++;
        for (int j = 0; j < 4; j++) if (!(fabs(u0[j]) <= 50.0 + 1e-9)) nonfinite++;
        x0[0] = yref[1][0]; x0[1] = yref[1][1]; x0[2] = -20; x0[5] = yref[1][5];
        if ((k + 1) % 100000 == 0) { printf("  %ld ticks: status != 0: %ld, outputs outside the box / NaN: %ld, ticks with the QP loop: %ld, mean %.1f us, worst %.1f us\n", k + 1, bad, nonfinite, loops, sum / (double)(k - 20), worst); fflush(stdout); }
    }
    printf("shim soak, N = %d, batch 1: %ld ticks, status != 0: %ld, outputs outside the box / NaN: %ld, ticks with the QP loop: %ld, mean %.1f us, worst %.1f us\n",
           BLUEROV2_N, T, bad, nonfinite, loops, sum / (double)(T - 21), worst);
    bluerov2_acados_free(c);
    bluerov2_acados_free_capsule(c);
    return (bad || nonfinite) ? 3 : 0;
}
