/* tests/shim_caller.c -- a C caller written against the acados-shaped drop-in headers (include/acados_shim) that makes the
 * same sequence of calls as the reference's control tick (BLUEROV2_DOB::solve, bluerov2_dobmpc/src/bluerov2_dob.cpp:306-388):
 * lbx/ubx <- x0, update_params for stages 0..N, yref for stages 0..N, solve, status / inf_norm_res / time_tot / u0.
 * Inputs come from a binary file written by the test (x0[12], p[16], nticks, yref[nticks][N+1][16]); results go to stdout.
 * A second argument "F" appends a failed step (NaN measurement) and a recovery tick; "G" creates the solver on a non-uniform grid;
 * "M" appends a tick whose QP stops at its iteration limit (qp_iter_max = 1, inputs limited to +-8, the vehicle metres off). */
#include <stdio.h>
#include <stdlib.h>

#include "acados/utils/print.h"
#include "acados_c/ocp_nlp_interface.h"
#include "acados_c/external_function_interface.h"
#include "acados/ocp_nlp/ocp_nlp_constraints_bgh.h"
#include "acados/ocp_nlp/ocp_nlp_cost_ls.h"
#include "blasfeo/include/blasfeo_d_aux.h"
#include "blasfeo/include/blasfeo_d_aux_ext_dep.h"
#include "bluerov2_model/bluerov2_model.h"
#include "acados_solver_bluerov2.h"

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: shim_caller inputs.bin\n"); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("open"); return 2; }
    double x0[BLUEROV2_NX], p[BLUEROV2_NP], nt;
    if (fread(x0, sizeof(double), BLUEROV2_NX, f) != BLUEROV2_NX || fread(p, sizeof(double), BLUEROV2_NP, f) != BLUEROV2_NP ||
        fread(&nt, sizeof(double), 1, f) != 1) return 2;
    const int nticks = (int)nt;
    static double yref[BLUEROV2_N + 1][BLUEROV2_NY];
    static double acados_param[BLUEROV2_N + 1][BLUEROV2_NP];

    bluerov2_solver_capsule* mpc_capsule = bluerov2_acados_create_capsule();
    int create_status;
    if (argc >= 3 && argv[2][0] == 'G') {
        /* non-uniform grid through the reference's own entry point (acados_solver_bluerov2.h:141): a geometric grid, 0.008 s growing
         * by 1 % per interval (the test builds the same numbers) */
        static double new_time_steps[BLUEROV2_N];
        double t = 0.008;
        for (int i = 0; i < BLUEROV2_N; i++) { new_time_steps[i] = t; t *= 1.01; }
        create_status = bluerov2_acados_create_with_discretization(mpc_capsule, BLUEROV2_N, new_time_steps);
    } else {
        create_status = bluerov2_acados_create(mpc_capsule);
    }
    if (create_status != 0) { printf("acados_create() returned status %d. Exiting.\n", create_status); return 1; }

    for (int tick = 0; tick < nticks; tick++) {
        if (fread(yref, sizeof(double), (BLUEROV2_N + 1) * BLUEROV2_NY, f) != (size_t)(BLUEROV2_N + 1) * BLUEROV2_NY) return 2;
        const int split = argc >= 3 && argv[2][0] == 'S';   /* acados' preparation / feedback split (main_bluerov2.c:217 sets the option) */
        if (!split) {
            ocp_nlp_constraints_model_set(mpc_capsule->nlp_config, mpc_capsule->nlp_dims, mpc_capsule->nlp_in, 0, "lbx", x0);
            ocp_nlp_constraints_model_set(mpc_capsule->nlp_config, mpc_capsule->nlp_dims, mpc_capsule->nlp_in, 0, "ubx", x0);
        }
        for (int i = 0; i < BLUEROV2_N + 1; i++) {
            for (int j = 0; j < BLUEROV2_NP; j++) acados_param[i][j] = p[j];
            bluerov2_acados_update_params(mpc_capsule, i, acados_param[i], BLUEROV2_NP);
        }
        for (unsigned int i = 0; i <= BLUEROV2_N; i++)
            ocp_nlp_cost_model_set(mpc_capsule->nlp_config, mpc_capsule->nlp_dims, mpc_capsule->nlp_in, i, "yref", yref[i]);
        if (split) {   /* preparation with references and parameters; the measurement arrives; feedback */
            int ph = 1;
            ocp_nlp_solver_opts_set(mpc_capsule->nlp_config, mpc_capsule->nlp_opts, "rti_phase", &ph);
            if (bluerov2_acados_solve(mpc_capsule) != 0) { printf("preparation failed\n"); return 1; }
            ph = 2;
            ocp_nlp_solver_opts_set(mpc_capsule->nlp_config, mpc_capsule->nlp_opts, "rti_phase", &ph);
            ocp_nlp_constraints_model_set(mpc_capsule->nlp_config, mpc_capsule->nlp_dims, mpc_capsule->nlp_in, 0, "lbx", x0);
            ocp_nlp_constraints_model_set(mpc_capsule->nlp_config, mpc_capsule->nlp_dims, mpc_capsule->nlp_in, 0, "ubx", x0);
        }
        int acados_status = bluerov2_acados_solve(mpc_capsule);
        double kkt_res = (double)mpc_capsule->nlp_out->inf_norm_res, cpu_time = 0.0, u0[BLUEROV2_NU], x1[BLUEROV2_NX];
        ocp_nlp_get(mpc_capsule->nlp_config, mpc_capsule->nlp_solver, "time_tot", &cpu_time);
        ocp_nlp_out_get(mpc_capsule->nlp_config, mpc_capsule->nlp_dims, mpc_capsule->nlp_out, 0, "u", (void*)u0);
        ocp_nlp_out_get(mpc_capsule->nlp_config, mpc_capsule->nlp_dims, mpc_capsule->nlp_out, 1, "x", (void*)x1);
        printf("TICK %d status %d kkt %.17g time %.6g u0 %.17g %.17g %.17g %.17g x1 %.17g %.17g %.17g\n", tick, acados_status, kkt_res,
               cpu_time, u0[0], u0[1], u0[2], u0[3], x1[0], x1[1], x1[2]);
    }
    fclose(f);
    if (argc >= 3 && argv[2][0] == 'F') {
        /* a failed step: a sensor glitch (NaN in the measured state) -> status 1.  The node publishes thrusts from the "u" getter
         * whatever the status (bluerov2_dob.cpp:375-395): it must see the last successfully computed input, as with acados. */
        double bad[BLUEROV2_NX], u0[BLUEROV2_NU];
        for (int j = 0; j < BLUEROV2_NX; j++) bad[j] = x0[j];
        bad[3] = 0.0 / 0.0;
        ocp_nlp_constraints_model_set(mpc_capsule->nlp_config, mpc_capsule->nlp_dims, mpc_capsule->nlp_in, 0, "lbx", bad);
        ocp_nlp_constraints_model_set(mpc_capsule->nlp_config, mpc_capsule->nlp_dims, mpc_capsule->nlp_in, 0, "ubx", bad);
        int st = bluerov2_acados_solve(mpc_capsule);
        ocp_nlp_out_get(mpc_capsule->nlp_config, mpc_capsule->nlp_dims, mpc_capsule->nlp_out, 0, "u", (void*)u0);
        printf("FAILED status %d u0 %.17g %.17g %.17g %.17g\n", st, u0[0], u0[1], u0[2], u0[3]);
        /* ... and the next tick with a sane measurement solves again */
        ocp_nlp_constraints_model_set(mpc_capsule->nlp_config, mpc_capsule->nlp_dims, mpc_capsule->nlp_in, 0, "lbx", x0);
        ocp_nlp_constraints_model_set(mpc_capsule->nlp_config, mpc_capsule->nlp_dims, mpc_capsule->nlp_in, 0, "ubx", x0);
        st = bluerov2_acados_solve(mpc_capsule);
        ocp_nlp_out_get(mpc_capsule->nlp_config, mpc_capsule->nlp_dims, mpc_capsule->nlp_out, 0, "u", (void*)u0);
        printf("RECOVERED status %d u0 %.17g %.17g %.17g %.17g\n", st, u0[0], u0[1], u0[2], u0[3]);
    }
    if (argc >= 3 && argv[2][0] == 'M') {
        /* the QP at its iteration limit: acados' SQP_RTI takes the step and returns ACADOS_SUCCESS, so the
         * node's `if (acados_status != 0) return;` (mpc.cpp:63-68) publishes the new input; the QP's own status stays readable */
        double lb[BLUEROV2_NU] = {-8, -8, -8, -8}, ub[BLUEROV2_NU] = {8, 8, 8, 8}, far[BLUEROV2_NX], u0[BLUEROV2_NU];
        int one = 1, nlp_status = -1, qp_status = -1, qp_iter = -1;
        for (int j = 0; j < BLUEROV2_NX; j++) far[j] = x0[j];
        far[0] += 6.0; far[1] -= 6.0; far[2] += 4.0;
        for (int i = 0; i < BLUEROV2_N; i++) {   /* acados_solver_bluerov2.c:559-573: the same box on every stage */
            ocp_nlp_constraints_model_set(mpc_capsule->nlp_config, mpc_capsule->nlp_dims, mpc_capsule->nlp_in, i, "lbu", lb);
            ocp_nlp_constraints_model_set(mpc_capsule->nlp_config, mpc_capsule->nlp_dims, mpc_capsule->nlp_in, i, "ubu", ub);
        }
        ocp_nlp_solver_opts_set(mpc_capsule->nlp_config, mpc_capsule->nlp_opts, "qp_iter_max", &one);
        ocp_nlp_constraints_model_set(mpc_capsule->nlp_config, mpc_capsule->nlp_dims, mpc_capsule->nlp_in, 0, "lbx", far);
        ocp_nlp_constraints_model_set(mpc_capsule->nlp_config, mpc_capsule->nlp_dims, mpc_capsule->nlp_in, 0, "ubx", far);
        int st = bluerov2_acados_solve(mpc_capsule);
        ocp_nlp_get(mpc_capsule->nlp_config, mpc_capsule->nlp_solver, "status", &nlp_status);
        ocp_nlp_get(mpc_capsule->nlp_config, mpc_capsule->nlp_solver, "qp_status", &qp_status);
        ocp_nlp_get(mpc_capsule->nlp_config, mpc_capsule->nlp_solver, "qp_iter", &qp_iter);
        ocp_nlp_out_get(mpc_capsule->nlp_config, mpc_capsule->nlp_dims, mpc_capsule->nlp_out, 0, "u", (void*)u0);
        printf("MAXITER returned %d status %d qp_status %d qp_iter %d u0 %.17g %.17g %.17g %.17g\n", st, nlp_status, qp_status, qp_iter,
               u0[0], u0[1], u0[2], u0[3]);
    }
    bluerov2_acados_print_stats(mpc_capsule);
    /* misuse that must not kill the process */
    int rc = bluerov2_acados_custom_update(mpc_capsule, NULL, 0);
    printf("custom_update %d\n", rc);
    rc = bluerov2_acados_free(mpc_capsule);
    rc |= bluerov2_acados_free_capsule(mpc_capsule);
    printf("free %d\n", rc);
    return 0;
}
