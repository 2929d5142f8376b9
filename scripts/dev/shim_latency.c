/* dev tool: wall time of one control tick through the acados-shaped drop-in at batch 1 -- the node's own call sequence
 * (bluerov2_dob.cpp:306-388: lbx / ubx, (N+1) x update_params, (N+1) x yref, solve, status / kkt / time_tot / u0), N = 80.
 *   gcc -O2 -Iinclude/acados_shim -o /tmp/shim_latency scripts/dev/shim_latency.c -Lbluerov2_amd/lib -lacados_ocp_solver_bluerov2 -Wl,-rpath,$PWD/bluerov2_amd/lib -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "acados_c/ocp_nlp_interface.h"
#include "acados_solver_bluerov2.h"

static int cmp(const void* a, const void* b) { double d = *(const double*)a - *(const double*)b; return d < 0 ? -1 : d > 0; }

int main(int argc, char** argv) {
    const double gap_us = argc > 1 ? atof(argv[1]) : 0.0;   /* idle time between ticks (a control loop does not run back to back) */
    const int split = argc > 2 && atoi(argv[2]) != 0;       /* 1: acados' preparation / feedback split -- rti_phase 1 with the references and parameters
                                                              * of the tick, the idle time, then the measurement and rti_phase 2; timed: the feedback half */
    bluerov2_solver_capsule* c = bluerov2_acados_create_capsule();
    if (bluerov2_acados_create(c)) return 1;
    static double yref[BLUEROV2_N + 1][BLUEROV2_NY], par[BLUEROV2_N + 1][BLUEROV2_NP];
    const double pn[16] = {0, 0, 0, 0, 1.7182, 0, 5.468, 0.4006, -11.7391, -20, -31.8678, -5, -18.18, -21.66, -36.99, -1.55};
    double x0[12] = {-2, 0, -20, 0, 0, -1.5707963267948966, 0, 0, 0, 0, 0, 0};
    enum { T = 400 };
    static double wall[T], tot[T];
    for (int k = 0; k < T; k++) {
        for (int i = 0; i <= BLUEROV2_N; i++) {   /* circle reference, one 0.05 s row per node (as the node feeds it) */
            const double t = 0.05 * (k + i), w = 0.75;
            for (int j = 0; j < 16; j++) yref[i][j] = 0.0;
            yref[i][0] = -2 * cos(w * t); yref[i][1] = -2 * sin(w * t); yref[i][2] = -20; yref[i][5] = w * t - 1.5707963267948966;
            yref[i][6] = 1.5;
            for (int j = 0; j < 16; j++) par[i][j] = pn[j];
        }
        struct timespec a, b;
        int st;
        if (split) {
            int ph = 1;
            ocp_nlp_solver_opts_set(c->nlp_config, c->nlp_opts, "rti_phase", &ph);
            for (int i = 0; i <= BLUEROV2_N; i++) bluerov2_acados_update_params(c, i, par[i], BLUEROV2_NP);
            for (int i = 0; i <= BLUEROV2_N; i++) ocp_nlp_cost_model_set(c->nlp_config, c->nlp_dims, c->nlp_in, i, "yref", yref[i]);
            if (bluerov2_acados_solve(c)) return 2;
            struct timespec g0, g1;
            clock_gettime(CLOCK_MONOTONIC, &g0);
            do clock_gettime(CLOCK_MONOTONIC, &g1); while ((g1.tv_sec - g0.tv_sec) * 1e6 + (g1.tv_nsec - g0.tv_nsec) * 1e-3 < (gap_us > 0 ? gap_us : 100.0));
            ph = 2;
            clock_gettime(CLOCK_MONOTONIC, &a);                      /* the measurement arrives */
            ocp_nlp_solver_opts_set(c->nlp_config, c->nlp_opts, "rti_phase", &ph);
            ocp_nlp_constraints_model_set(c->nlp_config, c->nlp_dims, c->nlp_in, 0, "lbx", x0);
            ocp_nlp_constraints_model_set(c->nlp_config, c->nlp_dims, c->nlp_in, 0, "ubx", x0);
            st = bluerov2_acados_solve(c);
        } else {
        clock_gettime(CLOCK_MONOTONIC, &a);
        ocp_nlp_constraints_model_set(c->nlp_config, c->nlp_dims, c->nlp_in, 0, "lbx", x0);
        ocp_nlp_constraints_model_set(c->nlp_config, c->nlp_dims, c->nlp_in, 0, "ubx", x0);
        for (int i = 0; i <= BLUEROV2_N; i++) bluerov2_acados_update_params(c, i, par[i], BLUEROV2_NP);
        for (int i = 0; i <= BLUEROV2_N; i++) ocp_nlp_cost_model_set(c->nlp_config, c->nlp_dims, c->nlp_in, i, "yref", yref[i]);
        st = bluerov2_acados_solve(c);
        }
        double u0[4], kkt = c->nlp_out->inf_norm_res, tt = 0;
        ocp_nlp_get(c->nlp_config, c->nlp_solver, "time_tot", &tt);
        ocp_nlp_out_get(c->nlp_config, c->nlp_dims, c->nlp_out, 0, "u", u0);
        clock_gettime(CLOCK_MONOTONIC, &b);
        wall[k] = (b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3;
        tot[k] = tt * 1e6;
        if (st != 0 && k > 5) { printf("tick %d status %d kkt %g\n", k, st, kkt); }
        x0[0] = yref[1][0]; x0[1] = yref[1][1]; x0[5] = yref[1][5];   /* a perfect plant: the state follows the reference */
        if (gap_us > 0 && !split) {
            struct timespec g0, g1;
            clock_gettime(CLOCK_MONOTONIC, &g0);
            do clock_gettime(CLOCK_MONOTONIC, &g1); while ((g1.tv_sec - g0.tv_sec) * 1e6 + (g1.tv_nsec - g0.tv_nsec) * 1e-3 < gap_us);
        }
    }
    qsort(wall + 20, T - 20, sizeof(double), cmp);
    qsort(tot + 20, T - 20, sizeof(double), cmp);
    if (split) printf("shim SPLIT tick at N = %d, batch 1 (rti_phase 1, %.0f us, measurement, rti_phase 2): FEEDBACK half, measurement -> u0: ", BLUEROV2_N, gap_us > 0 ? gap_us : 100.0);
    printf("shim tick at N = %d, batch 1, %.0f us idle between ticks: wall median %.1f us, p99 %.1f us;  time_tot median %.1f us\n", BLUEROV2_N, gap_us, wall[20 + (T - 20) / 2],
           wall[20 + (T - 20) * 99 / 100], tot[20 + (T - 20) / 2]);
    double tl = 0, tq = 0;
    ocp_nlp_get(c->nlp_config, c->nlp_solver, "time_lin", &tl);
    ocp_nlp_get(c->nlp_config, c->nlp_solver, "time_qp_sol", &tq);
    printf("   kernels of the last tick: %.1f us (linearise %.1f + qp %.1f)\n", (tl + tq) * 1e6, tl * 1e6, tq * 1e6);
    bluerov2_acados_free(c);
    bluerov2_acados_free_capsule(c);
    return 0;
}
