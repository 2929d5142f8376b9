/*
 * bluerov2_ekf_oracle.c -- see bluerov2_ekf_oracle.h.  TEST INFRASTRUCTURE ONLY; parity with the reference is unpinned
 * (the reference's EKF is not buildable here and has no recorded outputs).
 */
#include "bluerov2_ekf_oracle.h"

#include <math.h>
#include <string.h>

enum { N = ORC_EKF_N };

/* 6x6 inverse by Gauss-Jordan with partial pivoting; only used for the constant mass matrix */
static void inv6(const double* A, double* Ai) {
    double a[6][12];
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) { a[i][j] = A[i * 6 + j]; a[i][6 + j] = (i == j) ? 1.0 : 0.0; }
    for (int k = 0; k < 6; k++) {
        int p = k;
        for (int i = k + 1; i < 6; i++)
            if (fabs(a[i][k]) > fabs(a[p][k])) p = i;
        if (p != k)
            for (int j = 0; j < 12; j++) { const double t = a[k][j]; a[k][j] = a[p][j]; a[p][j] = t; }
        const double ip = 1.0 / a[k][k];
        for (int j = 0; j < 12; j++) a[k][j] *= ip;
        for (int i = 0; i < 6; i++) {
            if (i == k) continue;
            const double f = a[i][k];
            for (int j = 0; j < 12; j++) a[i][j] -= f * a[k][j];
        }
    }
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) Ai[i * 6 + j] = a[i][6 + j];
}

void orc_ekf_derive(orc_ekf_par* c) {
    /* bluerov2_dob.cpp:41-47 */
    double M[36], Mi[36];
    memset(M, 0, sizeof M);
    const double mv[6] = {c->mass + c->added_mass[0], c->mass + c->added_mass[1], c->mass + c->added_mass[2],
                          c->Ix + c->added_mass[3],   c->Iy + c->added_mass[4],   c->Iz + c->added_mass[5]};
    for (int i = 0; i < 6; i++) M[i * 6 + i] = mv[i];
    M[0 * 6 + 4] = c->mass * c->ZG;
    M[1 * 6 + 3] = -c->mass * c->ZG;
    M[3 * 6 + 1] = -c->mass * c->ZG;
    M[4 * 6 + 0] = c->mass * c->ZG;
    inv6(M, Mi);
    for (int i = 0; i < 6; i++) { c->Mdiag[i] = M[i * 6 + i]; c->invMdiag[i] = Mi[i * 6 + i]; }
}

void orc_ekf_default_par(orc_ekf_par* c) {
    /* bluerov2_dob.h:171-183,208; bluerov2_dob.cpp:52-62 */
    static const double am[6] = {1.7182, 0, 5.468, 0, 1.2481, 0.4006};
    static const double dl[6] = {-11.7391, -20, -31.8678, -25, -44.9085, -5};
    static const double dnl[6] = {-18.18, -21.66, -36.99, -1.55, -1.55, -1.55};
    static const double K[36] = {
        0.7071067811847433,   0.7071067811847433,   -0.7071067811919605,  -0.7071067811919605,  0.0,                   0.0,
        0.7071067811883519,   -0.7071067811883519,  0.7071067811811348,   -0.7071067811811348,  0.0,                   0.0,
        0,                    0,                    0,                    0,                    1,                     1,
        0.051265241636155506, -0.05126524163615552, 0.05126524163563227,  -0.05126524163563227, -0.11050000000000001,  0.11050000000000003,
        -0.05126524163589389, -0.051265241635893896, 0.05126524163641713, 0.05126524163641713,  -0.002499999999974481, -0.002499999999974481,
        0.16652364696949604,  -0.16652364696949604, -0.17500892834341342, 0.17500892834341342,  0.0,                   0.0};
    c->dt = 0.05;
    c->mass = 11.26; c->Ix = 0.3; c->Iy = 0.63; c->Iz = 0.58; c->ZG = 0.02; c->g = 9.81; c->bouyancy = 0.661618;
    memcpy(c->added_mass, am, sizeof am);
    memcpy(c->Dl, dl, sizeof dl);
    memcpy(c->Dnl, dnl, sizeof dnl);
    memcpy(c->K, K, sizeof K);
    for (int i = 0; i < 6; i++) c->Q[i] = pow(c->dt, 4) / 4;
    for (int i = 6; i < N; i++) c->Q[i] = pow(c->dt, 2);
    c->R = pow(c->dt, 4) / 4;
    c->fd_step = 1e-6;
    c->compensate_coef = 0.032546960744430276;
    c->rotor_constant = 0.026546960744430276;
    orc_ekf_derive(c);
}

void orc_ekf_init_state(double* x, double* P) {
    /* bluerov2_dob.cpp:64-65, bluerov2_dob.h:203 */
    static const double x0[N] = {0, 0, -20, 0, 0, 0, 0, 0, 0, 0, 0, 0, 6, 6, 6, 0, 0, 0};
    memcpy(x, x0, sizeof x0);
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) P[i * N + j] = (i == j) ? 1.0 : 0.0;
}

/* bluerov2_dob.cpp:637-702: kinematics as in the NMPC model (incl. sin(psi) in phi-dot), dynamics with the rigid-body
 * Coriolis terms on the linear axes and linear + quadratic damping on all six */
void orc_ekf_f(const orc_ekf_par* c, const double* x, const double* tau, double* xd) {
    const double sph = sin(x[3]), cph = cos(x[3]), sth = sin(x[4]), cth = cos(x[4]), sps = sin(x[5]), cps = cos(x[5]);
    const double m = c->mass, bo = c->bouyancy, mzg = c->mass * c->ZG * c->g;
    xd[0] = (cps * cth) * x[6] + (-sps * cph + cps * sth * sph) * x[7] + (sps * sph + cps * cph * sth) * x[8];
    xd[1] = (sps * cth) * x[6] + (cps * cph + sph * sth * sps) * x[7] + (-cps * sph + sth * sps * cph) * x[8];
    xd[2] = (-sth) * x[6] + (cth * sph) * x[7] + (cth * cph) * x[8];
    xd[3] = x[9] + (sps * sth / cth) * x[10] + cph * sth / cth * x[11];
    xd[4] = cph * x[10] + sph * x[11];
    xd[5] = (sph / cth) * x[10] + (cph / cth) * x[11];
    xd[6] = c->invMdiag[0] * (tau[0] + m * x[11] * x[7] - m * x[10] * x[8] - bo * sth + x[12] + c->Dl[0] * x[6] +
                              c->Dnl[0] * fabs(x[6]) * x[6]);
    xd[7] = c->invMdiag[1] * (tau[1] - m * x[11] * x[6] + m * x[9] * x[8] + bo * cth * sph + x[13] + c->Dl[1] * x[7] +
                              c->Dnl[1] * fabs(x[7]) * x[7]);
    xd[8] = c->invMdiag[2] * (tau[2] + m * x[10] * x[6] - m * x[9] * x[7] + bo * cth * cph + x[14] + c->Dl[2] * x[8] +
                              c->Dnl[2] * fabs(x[8]) * x[8]);
    xd[9] = c->invMdiag[3] * (tau[3] + (c->Iy - c->Iz) * x[10] * x[11] - mzg * cth * sph + x[15] + c->Dl[3] * x[9] +
                              c->Dnl[3] * fabs(x[9]) * x[9]);
    xd[10] = c->invMdiag[4] * (tau[4] + (c->Iz - c->Ix) * x[9] * x[11] - mzg * sth + x[16] + c->Dl[4] * x[10] +
                               c->Dnl[4] * fabs(x[10]) * x[10]);
    xd[11] = c->invMdiag[5] * (tau[5] - (c->Iy - c->Ix) * x[9] * x[10] + x[17] + c->Dl[5] * x[11] +
                               c->Dnl[5] * fabs(x[11]) * x[11]);
    for (int i = 12; i < N; i++) xd[i] = 0.0;
}

/* bluerov2_dob.cpp:621-634: classical weights, but the third stage is evaluated at x + k2/3 (sic) */
void orc_ekf_rk4(const orc_ekf_par* c, const double* x, const double* tau, double* xn) {
    double k1[N], k2[N], k3[N], k4[N], xs[N];
    const double dt = c->dt;
    orc_ekf_f(c, x, tau, k1);
    for (int i = 0; i < N; i++) { k1[i] *= dt; xs[i] = x[i] + k1[i] / 2; }
    orc_ekf_f(c, xs, tau, k2);
    for (int i = 0; i < N; i++) { k2[i] *= dt; xs[i] = x[i] + k2[i] / 3; }
    orc_ekf_f(c, xs, tau, k3);
    for (int i = 0; i < N; i++) { k3[i] *= dt; xs[i] = x[i] + k3[i]; }
    orc_ekf_f(c, xs, tau, k4);
    for (int i = 0; i < N; i++) { k4[i] *= dt; xn[i] = x[i] + (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) / 6; }
}

/* bluerov2_dob.cpp:705-727: pose and velocities directly; the generalised thrust is recovered from the measured body
 * accelerations by inverting the dynamics */
void orc_ekf_h(const orc_ekf_par* c, const double* x, const double* acc, double* y) {
    const double sph = sin(x[3]), cph = cos(x[3]), sth = sin(x[4]), cth = cos(x[4]);
    const double m = c->mass, bo = c->bouyancy, mzg = c->mass * c->ZG * c->g;
    for (int i = 0; i < 12; i++) y[i] = x[i];
    y[12] = c->Mdiag[0] * acc[0] - m * x[11] * x[7] + m * x[10] * x[8] + bo * sth - x[12] - c->Dl[0] * x[6] -
            c->Dnl[0] * fabs(x[6]) * x[6];
    y[13] = c->Mdiag[1] * acc[1] + m * x[11] * x[6] - m * x[9] * x[8] - bo * cth * sph - x[13] - c->Dl[1] * x[7] -
            c->Dnl[1] * fabs(x[7]) * x[7];
    y[14] = c->Mdiag[2] * acc[2] - m * x[10] * x[6] + m * x[9] * x[7] - bo * cth * cph - x[14] - c->Dl[2] * x[8] -
            c->Dnl[2] * fabs(x[8]) * x[8];
    y[15] = c->Mdiag[3] * acc[3] - (c->Iy - c->Iz) * x[10] * x[11] + mzg * cth * sph - x[15] - c->Dl[3] * x[9] -
            c->Dnl[3] * fabs(x[9]) * x[9];
    y[16] = c->Mdiag[4] * acc[4] - (c->Iz - c->Ix) * x[9] * x[11] + mzg * sth - x[16] - c->Dl[4] * x[10] -
            c->Dnl[4] * fabs(x[10]) * x[10];
    y[17] = c->Mdiag[5] * acc[5] + (c->Iy - c->Ix) * x[9] * x[10] - x[17] - c->Dl[5] * x[11] -
            c->Dnl[5] * fabs(x[11]) * x[11];
}

/* bluerov2_dob.cpp:730-744: forward differences of the RK4 map */
void orc_ekf_jac_F(const orc_ekf_par* c, const double* x, const double* tau, double* F) {
    double f0[N], f1[N], x1[N];
    const double d = c->fd_step;
    orc_ekf_rk4(c, x, tau, f0);
    for (int i = 0; i < N; i++) {
        memcpy(x1, x, sizeof x1);
        x1[i] += d;
        orc_ekf_rk4(c, x1, tau, f1);
        for (int j = 0; j < N; j++) F[j * N + i] = (f1[j] - f0[j]) / d;
    }
}

/* bluerov2_dob.cpp:747-762 */
void orc_ekf_jac_H(const orc_ekf_par* c, const double* x, const double* acc, double* H) {
    double f0[N], f1[N], x1[N];
    const double d = c->fd_step;
    orc_ekf_h(c, x, acc, f0);
    for (int i = 0; i < N; i++) {
        memcpy(x1, x, sizeof x1);
        x1[i] += d;
        orc_ekf_h(c, x1, acc, f1);
        for (int j = 0; j < N; j++) H[j * N + i] = (f1[j] - f0[j]) / d;
    }
}

static void mat_mul(const double* A, const double* B, double* C) { /* C = A B */
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            double s = 0.0;
            for (int k = 0; k < N; k++) s += A[i * N + k] * B[k * N + j];
            C[i * N + j] = s;
        }
}
static void mat_mul_bt(const double* A, const double* B, double* C) { /* C = A B^T */
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) {
            double s = 0.0;
            for (int k = 0; k < N; k++) s += A[i * N + k] * B[j * N + k];
            C[i * N + j] = s;
        }
}

/* 18x18 inverse by LU with partial pivoting (what Eigen's fixed-size inverse() does beyond 4x4) */
static int inv18(const double* A, double* Ai) {
    double a[N][2 * N];
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) { a[i][j] = A[i * N + j]; a[i][N + j] = (i == j) ? 1.0 : 0.0; }
    for (int k = 0; k < N; k++) {
        int p = k;
        for (int i = k + 1; i < N; i++)
            if (fabs(a[i][k]) > fabs(a[p][k])) p = i;
        if (!(fabs(a[p][k]) > 0.0)) return 1;
        if (p != k)
            for (int j = 0; j < 2 * N; j++) { const double t = a[k][j]; a[k][j] = a[p][j]; a[p][j] = t; }
        const double ip = 1.0 / a[k][k];
        for (int j = 0; j < 2 * N; j++) a[k][j] *= ip;
        for (int i = 0; i < N; i++) {
            if (i == k) continue;
            const double f = a[i][k];
            if (f == 0.0) continue;
            for (int j = 0; j < 2 * N; j++) a[i][j] -= f * a[k][j];
        }
    }
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) Ai[i * N + j] = a[i][N + j];
    return 0;
}

/* bluerov2_dob.cpp:495-545 */
int orc_ekf_update(const orc_ekf_par* c, double* x, double* P, const double* thrust, const double* y12, const double* acc,
                   double* wf, double* mpc_p) {
    double tau[6], y[N], F[N * N], H[N * N], xp[N], Pp[N * N], T1[N * N], T2[N * N], S[N * N], Si[N * N], Kal[N * N];
    double yp[N], ye[N], J[N * N];
    for (int i = 0; i < 6; i++) {
        double s = 0.0;
        for (int j = 0; j < 6; j++) s += c->K[i * 6 + j] * thrust[j];
        tau[i] = s;
    }
    for (int i = 0; i < 12; i++) y[i] = y12[i];
    for (int i = 0; i < 6; i++) y[12 + i] = tau[i];
    /* predict */
    orc_ekf_jac_F(c, x, tau, F);
    orc_ekf_rk4(c, x, tau, xp);
    mat_mul(F, P, T1);
    mat_mul_bt(T1, F, Pp);
    for (int i = 0; i < N; i++) Pp[i * N + i] += c->Q[i];
    /* update */
    orc_ekf_jac_H(c, xp, acc, H);
    orc_ekf_h(c, xp, acc, yp);
    for (int i = 0; i < N; i++) ye[i] = y[i] - yp[i];
    mat_mul_bt(Pp, H, T1);  /* P H^T */
    mat_mul(H, T1, S);
    for (int i = 0; i < N; i++) S[i * N + i] += c->R;
    if (inv18(S, Si)) return 1;
    mat_mul(T1, Si, Kal);
    for (int i = 0; i < N; i++) {
        double s = 0.0;
        for (int k = 0; k < N; k++) s += Kal[i * N + k] * ye[k];
        x[i] = xp[i] + s;
    }
    mat_mul(Kal, H, J);
    for (int i = 0; i < N; i++)
        for (int j = 0; j < N; j++) J[i * N + j] = ((i == j) ? 1.0 : 0.0) - J[i * N + j];
    mat_mul(J, Pp, T1);
    mat_mul_bt(T1, J, T2);
    mat_mul_bt(Kal, Kal, T1);
    for (int i = 0; i < N * N; i++) P[i] = T2[i] + c->R * T1[i];
    /* body-frame disturbance -> inertial frame with the MEASURED attitude (bluerov2_dob.cpp:540-545) */
    {
        const double sph = sin(y[3]), cph = cos(y[3]), sth = sin(y[4]), cth = cos(y[4]), sps = sin(y[5]), cps = cos(y[5]);
        wf[0] = (cps * cth) * x[12] + (-sps * cph + cps * sth * sph) * x[13] + (sps * sph + cps * cph * sth) * x[14];
        wf[1] = (sps * cth) * x[12] + (cps * cph + sph * sth * sps) * x[13] + (-cps * sph + sth * sps * cph) * x[14];
        wf[2] = (-sth) * x[12] + (cth * sph) * x[13] + (cth * cph) * x[14];
        wf[3] = x[15] + (sps * sth / cth) * x[16] + cph * sth / cth * x[17];
        wf[4] = cph * x[16] + sph * x[17];
        wf[5] = (sph / cth) * x[16] + (cph / cth) * x[17];
    }
    /* bluerov2_dob.cpp:334-337 */
    mpc_p[0] = x[12] / c->compensate_coef;
    mpc_p[1] = x[13] / c->compensate_coef;
    mpc_p[2] = x[14] / c->rotor_constant;
    mpc_p[3] = x[17] / c->rotor_constant;
    return 0;
}

int orc_ekf_update_batch(const orc_ekf_par* c, int B, double* x, double* P, const double* thrust, const double* y12,
                         const double* acc, double* wf, double* mpc_p) {
    int bad = 0;
#pragma omp parallel for schedule(static) reduction(| : bad)
    for (int b = 0; b < B; b++)
        bad |= orc_ekf_update(c, x + (size_t)b * N, P + (size_t)b * N * N, thrust + (size_t)b * 6, y12 + (size_t)b * 12,
                              acc + (size_t)b * 6, wf + (size_t)b * 6, mpc_p + (size_t)b * 4);
    return bad;
}
