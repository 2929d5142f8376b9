/*
 * bluerov2_oracle.c -- CPU restatement of the BlueROV2 NMPC RTI hot path (TEST INFRASTRUCTURE; see bluerov2_oracle.h).
 * Plain C99, FP64, no dependencies beyond libm (+ optional OpenMP for the batch driver).
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 */
#include "bluerov2_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef ORC_TRACE   /* development build: gcc -DORC_TRACE ... prints the QP loop's decisions to stderr (scripts/dev/qp_trace.py) */
#include <stdio.h>
#define TRACE(...) fprintf(stderr, __VA_ARGS__)
#else
#define TRACE(...) do { } while (0)
#endif
#ifdef _OPENMP
#include <omp.h>
#endif

#define NX ORC_NX
#define NU ORC_NU
#define NP ORC_NP
#define NY ORC_NY
#define NXU (NX + NU)

/* bluerov2_dobmpc/scripts/bluerov2.py:77-84 */
static const double M_ = 11.26, IX = 0.3, IY = 0.63, IZ = 0.58, ZG = 0.02, GR = 9.81, BOUY = 0.66,
                    ROTOR = 0.026546960744430276;

void orc_default_opts(orc_opts* o, int N, double Ts) {
    /* c_generated_code/acados_solver_bluerov2.c:422-481 (the generated C carries 40 on u,v; generate_c_code.py:34 says 10) */
    static const double W[NY] = {300, 480, 200, 10, 10, 200, 40, 40, 10, 10, 10, 10, 1, 1, 0.1, 0.05};
    o->N = N;
    o->Ts = Ts;
    for (int j = 0; j < NY; j++) o->W[j] = W[j];
    for (int j = 0; j < NX; j++) o->We[j] = W[j];
    for (int j = 0; j < NU; j++) { o->lbu[j] = -50.0; o->ubu[j] = 50.0; } /* :559-566 */
    o->qp_iter_max = 50;                                                     /* :668 */
    o->qp_tol_mu = 1e-7;
    o->qp_tol_stat = 1e-9;
    o->qp_early_exit = 1;
    o->ts_vec = NULL;
    o->W0 = NULL;
    o->on_failure = 1;
}

/* ---------------------------------------------------------------------------------------------------------
 * model: bluerov2.py:103-137.  Thruster map :103-108, K :95-100, dynamics :123-135 (note sin(psi) in dphi, :133)
 * ------------------------------------------------------------------------------------------------------- */
typedef struct trig { double sph, cph, sth, cth, sps, cps; } trig;

static inline void thrust_wrench(const double* u, double* Kt0, double* Kt1, double* Kt2, double* Kt5) {
    const double t0 = (-u[0] + u[1] + u[3]) / ROTOR, t1 = (-u[0] - u[1] - u[3]) / ROTOR;
    const double t2 = (u[0] + u[1] - u[3]) / ROTOR, t3 = (u[0] - u[1] + u[3]) / ROTOR;
    const double t4 = -u[2] / ROTOR, t5 = -u[2] / ROTOR;
    *Kt0 = 0.707 * t0 + 0.707 * t1 - 0.707 * t2 - 0.707 * t3;
    *Kt1 = 0.707 * t0 - 0.707 * t1 + 0.707 * t2 - 0.707 * t3;
    *Kt2 = t4 + t5;
    *Kt5 = 0.167 * t0 - 0.167 * t1 - 0.175 * t2 + 0.175 * t3;
}

void orc_f6(const double* x, const double* u, const double* p, const double* drp, double* f);
void orc_f(const double* x, const double* u, const double* p, double* f) { orc_f6(x, u, p, NULL, f); }
/* drp != NULL: the 6-disturbance variant -- the roll / pitch disturbance moments the reference carries as commented-out symbols
 * (bluerov2.py:37-38) enter dp, dq the way the other four enter their rows (:123-128): + d_phi / Ix, + d_theta / Iy */
void orc_f6(const double* x, const double* u, const double* p, const double* drp, double* f) {
    const double ph = x[3], th = x[4], ps = x[5], vu = x[6], vv = x[7], vw = x[8], wp = x[9], wq = x[10], wr = x[11];
    const double sph = sin(ph), cph = cos(ph), sth = sin(th), cth = cos(th), sps = sin(ps), cps = cos(ps);
    double Kt0, Kt1, Kt2, Kt5;
    thrust_wrench(u, &Kt0, &Kt1, &Kt2, &Kt5);
    f[0] = (cps * cth) * vu + (-sps * cph + cps * sth * sph) * vv + (sps * sph + cps * cph * sth) * vw;
    f[1] = (sps * cth) * vu + (cps * cph + sph * sth * sps) * vv + (-cps * sph + sth * sps * cph) * vw;
    f[2] = (-sth) * vu + (cth * sph) * vv + (cth * cph) * vw;
    f[3] = wp + (sps * sth / cth) * wq + cph * sth / cth * wr; /* sin(psi): reference quirk, bluerov2.py:133 */
    f[4] = cph * wq + sph * wr;
    f[5] = (sph / cth) * wq + (cph / cth) * wr;
    f[6] = (Kt0 - BOUY * sth + p[0] + p[8] * vu + p[12] * fabs(vu) * vu) / (M_ + p[4]);
    f[7] = (Kt1 + BOUY * cth * sph + p[1] + p[9] * vv + p[13] * fabs(vv) * vv) / (M_ + p[5]);
    f[8] = (Kt2 + BOUY * cth * cph + p[2] + p[10] * vw + p[14] * fabs(vw) * vw) / (M_ + p[6]);
    f[9] = ((IY - IZ) * wq * wr - M_ * ZG * GR * cth * sph + (drp ? drp[0] : 0.0)) / IX;
    f[10] = ((IZ - IX) * wp * wr - M_ * ZG * GR * sth + (drp ? drp[1] : 0.0)) / IY;
    f[11] = (Kt5 - (IY - IX) * wp * wq + p[3] + p[11] * wr + p[15] * fabs(wr) * wr) / (IZ + p[7]);
}

/* d f / d x and d f / d u.  Sparsity: 48/144 and 5/48 (matches the structure CasADi differentiates in
 * c_generated_code/bluerov2_model/bluerov2_expl_vde_forw.c; d(|v|v)/dv = sign(v) v + |v| = 2|v|, 0 at v = 0, :65) */
void orc_jac(const double* x, const double* u, const double* p, double* A, double* B) {
    (void)u;
    const double ph = x[3], th = x[4], ps = x[5], vu = x[6], vv = x[7], vw = x[8], wp = x[9], wq = x[10], wr = x[11];
    const double sph = sin(ph), cph = cos(ph), sth = sin(th), cth = cos(th), sps = sin(ps), cps = cos(ps);
    const double tth = sth / cth, icth = 1.0 / cth;
    memset(A, 0, sizeof(double) * NX * NX);
    memset(B, 0, sizeof(double) * NX * NU);
#define A_(r, c) A[(r)*NX + (c)]
#define B_(r, c) B[(r)*NU + (c)]
    /* xdot */
    A_(0, 3) = (cps * sth * cph + sps * sph) * vv + (sps * cph - cps * sph * sth) * vw;
    A_(0, 4) = -cps * sth * vu + cps * cth * sph * vv + cps * cph * cth * vw;
    A_(0, 5) = -sps * cth * vu + (-cps * cph - sps * sth * sph) * vv + (cps * sph - sps * cph * sth) * vw;
    A_(0, 6) = cps * cth;
    A_(0, 7) = -sps * cph + cps * sth * sph;
    A_(0, 8) = sps * sph + cps * cph * sth;
    /* ydot */
    A_(1, 3) = (-cps * sph + cph * sth * sps) * vv + (-cps * cph - sth * sps * sph) * vw;
    A_(1, 4) = -sps * sth * vu + sph * cth * sps * vv + cth * sps * cph * vw;
    A_(1, 5) = cps * cth * vu + (-sps * cph + sph * sth * cps) * vv + (sps * sph + sth * cps * cph) * vw;
    A_(1, 6) = sps * cth;
    A_(1, 7) = cps * cph + sph * sth * sps;
    A_(1, 8) = -cps * sph + sth * sps * cph;
    /* zdot */
    A_(2, 3) = cth * cph * vv - cth * sph * vw;
    A_(2, 4) = -cth * vu - sth * sph * vv - sth * cph * vw;
    A_(2, 6) = -sth;
    A_(2, 7) = cth * sph;
    A_(2, 8) = cth * cph;
    /* phidot = p + sin(psi) tan(th) q + cos(phi) tan(th) r */
    A_(3, 3) = -sph * tth * wr;
    A_(3, 4) = (sps * wq + cph * wr) * icth * icth;
    A_(3, 5) = cps * tth * wq;
    A_(3, 9) = 1.0;
    A_(3, 10) = sps * tth;
    A_(3, 11) = cph * tth;
    /* thetadot = cos(phi) q + sin(phi) r */
    A_(4, 3) = -sph * wq + cph * wr;
    A_(4, 10) = cph;
    A_(4, 11) = sph;
    /* psidot = (sin(phi) q + cos(phi) r)/cos(th) */
    A_(5, 3) = (cph * wq - sph * wr) * icth;
    A_(5, 4) = (sph * wq + cph * wr) * sth * icth * icth;
    A_(5, 10) = sph * icth;
    A_(5, 11) = cph * icth;
    /* udot, vdot, wdot */
    const double imx = 1.0 / (M_ + p[4]), imy = 1.0 / (M_ + p[5]), imz = 1.0 / (M_ + p[6]), imn = 1.0 / (IZ + p[7]);
    A_(6, 4) = -BOUY * cth * imx;
    A_(6, 6) = (p[8] + 2.0 * p[12] * fabs(vu)) * imx;
    A_(7, 3) = BOUY * cth * cph * imy;
    A_(7, 4) = -BOUY * sth * sph * imy;
    A_(7, 7) = (p[9] + 2.0 * p[13] * fabs(vv)) * imy;
    A_(8, 3) = -BOUY * cth * sph * imz;
    A_(8, 4) = -BOUY * sth * cph * imz;
    A_(8, 8) = (p[10] + 2.0 * p[14] * fabs(vw)) * imz;
    /* pdot, qdot, rdot */
    const double mzg = M_ * ZG * GR;
    A_(9, 3) = -mzg * cth * cph / IX;
    A_(9, 4) = mzg * sth * sph / IX;
    A_(9, 10) = (IY - IZ) * wr / IX;
    A_(9, 11) = (IY - IZ) * wq / IX;
    A_(10, 4) = -mzg * cth / IY;
    A_(10, 9) = (IZ - IX) * wr / IY;
    A_(10, 11) = (IZ - IX) * wp / IY;
    A_(11, 9) = -(IY - IX) * wq * imn;
    A_(11, 10) = -(IY - IX) * wp * imn;
    A_(11, 11) = (p[11] + 2.0 * p[15] * fabs(wr)) * imn;
    /* inputs: Kt0 = -4*0.707 u1/c, Kt1 = 4*0.707 u2/c, Kt2 = -2 u3/c, Kt5 = (2*0.167-2*0.175) u2/c + (2*0.167+2*0.175) u4/c */
    B_(6, 0) = (-0.707 - 0.707 - 0.707 - 0.707) / ROTOR * imx;
    B_(7, 1) = (0.707 + 0.707 + 0.707 + 0.707) / ROTOR * imy;
    B_(8, 2) = -2.0 / ROTOR * imz;
    B_(11, 1) = (0.167 + 0.167 - 0.175 - 0.175) / ROTOR * imn;
    B_(11, 3) = (0.167 + 0.167 + 0.175 + 0.175) / ROTOR * imn;
#undef A_
#undef B_
}

/* ---------------------------------------------------------------------------------------------------------
 * ERK4 with forward sensitivities, one step per interval: acados_solver_bluerov2.c:633-641 (4 stages, 1 step),
 * variational equation = bluerov2_expl_vde_forw (Sx' = A Sx, Su' = A Su + B), seeds Sx = I, Su = 0.
 * ------------------------------------------------------------------------------------------------------- */
void orc_rk4_sens(const double* x, const double* u, const double* p, double h, double* xn, double* Aout, double* Bout) {
    orc_rk4_sens6(x, u, p, NULL, h, xn, Aout, Bout);
}
void orc_rk4_sens6(const double* x, const double* u, const double* p, const double* drp, double h, double* xn, double* Aout,
                   double* Bout) {
    static const double ca[4] = {0.0, 0.5, 0.5, 1.0}, cb[4] = {1.0 / 6, 1.0 / 3, 1.0 / 3, 1.0 / 6};
    double S0[NX * NXU], Sacc[NX * NXU], Ks[NX * NXU], Ss[NX * NXU];
    double xs[NX], k[NX], xacc[NX], Ac[NX * NX], Bc[NX * NU];
    memset(S0, 0, sizeof S0);
    for (int i = 0; i < NX; i++) S0[i * NXU + i] = 1.0;
    memcpy(Sacc, S0, sizeof S0);
    memcpy(xacc, x, sizeof xacc);
    memset(Ks, 0, sizeof Ks);
    memset(k, 0, sizeof k);
    for (int s = 0; s < 4; s++) {
        for (int i = 0; i < NX; i++) xs[i] = x[i] + h * ca[s] * k[i];
        for (int i = 0; i < NX * NXU; i++) Ss[i] = S0[i] + h * ca[s] * Ks[i];
        orc_f6(xs, u, p, drp, k);
        orc_jac(xs, u, p, Ac, Bc);
        for (int r = 0; r < NX; r++)
            for (int c = 0; c < NXU; c++) {
                double acc = (c >= NX) ? Bc[r * NU + (c - NX)] : 0.0;
                for (int m = 0; m < NX; m++) acc += Ac[r * NX + m] * Ss[m * NXU + c];
                Ks[r * NXU + c] = acc;
            }
        for (int i = 0; i < NX; i++) xacc[i] += h * cb[s] * k[i];
        for (int i = 0; i < NX * NXU; i++) Sacc[i] += h * cb[s] * Ks[i];
    }
    memcpy(xn, xacc, sizeof xacc);
    for (int r = 0; r < NX; r++) {
        for (int c = 0; c < NX; c++) Aout[r * NX + c] = Sacc[r * NXU + c];
        for (int c = 0; c < NU; c++) Bout[r * NU + c] = Sacc[r * NXU + NX + c];
    }
}

void orc_rk4(const double* x, const double* u, const double* p, double h, double* xn) { orc_rk4_6(x, u, p, NULL, h, xn); }
void orc_rk4_6(const double* x, const double* u, const double* p, const double* drp, double h, double* xn) {
    double k1[NX], k2[NX], k3[NX], k4[NX], xs[NX];
    orc_f6(x, u, p, drp, k1);
    for (int i = 0; i < NX; i++) xs[i] = x[i] + 0.5 * h * k1[i];
    orc_f6(xs, u, p, drp, k2);
    for (int i = 0; i < NX; i++) xs[i] = x[i] + 0.5 * h * k2[i];
    orc_f6(xs, u, p, drp, k3);
    for (int i = 0; i < NX; i++) xs[i] = x[i] + h * k3[i];
    orc_f6(xs, u, p, drp, k4);
    for (int i = 0; i < NX; i++) xn[i] = x[i] + h / 6.0 * (k1[i] + 2.0 * k2[i] + 2.0 * k3[i] + k4[i]);
}

/* ---------------------------------------------------------------------------------------------------------
 * QP: what acados hands to HPIPM (FULL_CONDENSING_HPIPM, acados_solver_bluerov2.c:146,664-669).  The reference
 * condenses and runs a dense IPM; the minimiser of this strictly convex QP is unique, so the restatement solves
 * the same QP in its OCP-structured form: primal-dual Mehrotra IPM, Newton systems by a Riccati sweep.
 * ------------------------------------------------------------------------------------------------------- */
typedef struct qp_ws {
    int N;
    double *P;    /* (N+1)*144 cost-to-go Hessians */
    double *K;    /* N*48 feedback gains, row-major [4][12] */
    double *L;    /* N*16 Cholesky factor of Huu, row-major lower */
    double *pv;   /* (N+1)*12 */
    double *kff;  /* N*4 */
    double *xs;   /* (N+1)*12 */
    double *vs;   /* N*4 */
    double *pis;  /* N*12 */
} qp_ws;

static int chol4(const double* H, double* L) {
    memset(L, 0, 16 * sizeof(double));
    for (int j = 0; j < 4; j++) {
        double d = H[j * 4 + j];
        for (int m = 0; m < j; m++) d -= L[j * 4 + m] * L[j * 4 + m];
        if (!(d > 0.0)) return 1;
        d = sqrt(d);
        L[j * 4 + j] = d;
        for (int i = j + 1; i < 4; i++) {
            double s = H[i * 4 + j];
            for (int m = 0; m < j; m++) s -= L[i * 4 + m] * L[j * 4 + m];
            L[i * 4 + j] = s / d;
        }
    }
    return 0;
}

/* solve L L' y = rhs in place (4-vector) */
static void chol4_solve(const double* L, double* y) {
    for (int i = 0; i < 4; i++) {
        double s = y[i];
        for (int m = 0; m < i; m++) s -= L[i * 4 + m] * y[m];
        y[i] = s / L[i * 4 + i];
    }
    for (int i = 3; i >= 0; i--) {
        double s = y[i];
        for (int m = i + 1; m < 4; m++) s -= L[m * 4 + i] * y[m];
        y[i] = s / L[i * 4 + i];
    }
}

/* backward Riccati factorisation with input-Hessian diagonal Rd + gam */
static int ric_factor(qp_ws* w, const double* A, const double* B, const double* Qd, const double* Rd, const double* gam) {
    const int N = w->N;
    double* PN = w->P + (size_t)N * 144;
    memset(PN, 0, 144 * sizeof(double));
    for (int j = 0; j < NX; j++) PN[j * NX + j] = Qd[N * NX + j];
    for (int i = N - 1; i >= 0; i--) {
        const double* Pn = w->P + (size_t)(i + 1) * 144;
        const double* Ai = A + (size_t)i * 144;
        const double* Bi = B + (size_t)i * 48;
        double PA[144], PB[48], Hxx[144], Hux[48], Huu[16];
        for (int r = 0; r < NX; r++) {
            for (int c = 0; c < NX; c++) {
                double s = 0;
                for (int m = 0; m < NX; m++) s += Pn[r * NX + m] * Ai[m * NX + c];
                PA[r * NX + c] = s;
            }
            for (int c = 0; c < NU; c++) {
                double s = 0;
                for (int m = 0; m < NX; m++) s += Pn[r * NX + m] * Bi[m * NU + c];
                PB[r * NU + c] = s;
            }
        }
        for (int r = 0; r < NX; r++)
            for (int c = 0; c < NX; c++) {
                double s = (r == c) ? Qd[i * NX + r] : 0.0;
                for (int m = 0; m < NX; m++) s += Ai[m * NX + r] * PA[m * NX + c];
                Hxx[r * NX + c] = s;
            }
        for (int r = 0; r < NU; r++) {
            for (int c = 0; c < NX; c++) {
                double s = 0;
                for (int m = 0; m < NX; m++) s += Bi[m * NU + r] * PA[m * NX + c];
                Hux[r * NX + c] = s;
            }
            for (int c = 0; c < NU; c++) {
                double s = (r == c) ? Rd[i * NU + r] + gam[i * NU + r] : 0.0;
                for (int m = 0; m < NX; m++) s += Bi[m * NU + r] * PB[m * NU + c];
                Huu[r * NU + c] = s;
            }
        }
        double* L = w->L + (size_t)i * 16;
        if (chol4(Huu, L)) return 4;
        double* K = w->K + (size_t)i * 48;
        for (int c = 0; c < NX; c++) {
            double col[4] = {Hux[0 * NX + c], Hux[1 * NX + c], Hux[2 * NX + c], Hux[3 * NX + c]};
            chol4_solve(L, col);
            for (int r = 0; r < NU; r++) K[r * NX + c] = -col[r];
        }
        double* Pi = w->P + (size_t)i * 144;
        for (int r = 0; r < NX; r++)
            for (int c = 0; c < NX; c++) {
                double s = Hxx[r * NX + c];
                for (int m = 0; m < NU; m++) s += Hux[m * NX + r] * K[m * NX + c];
                Pi[r * NX + c] = s;
            }
        for (int r = 0; r < NX; r++) /* symmetrise */
            for (int c = r + 1; c < NX; c++) {
                double s = 0.5 * (Pi[r * NX + c] + Pi[c * NX + r]);
                Pi[r * NX + c] = Pi[c * NX + r] = s;
            }
    }
    return 0;
}

/* vector recursion + forward sweep for input gradient rt; fills w->xs, w->vs, w->pis */
static void ric_solve(qp_ws* w, const double* A, const double* B, const double* b, const double* q, const double* rt,
                      const double* d0) {
    const int N = w->N;
    memcpy(w->pv + (size_t)N * NX, q + (size_t)N * NX, NX * sizeof(double));
    for (int i = N - 1; i >= 0; i--) {
        const double* Pn = w->P + (size_t)(i + 1) * 144;
        const double* Ai = A + (size_t)i * 144;
        const double* Bi = B + (size_t)i * 48;
        const double* pn = w->pv + (size_t)(i + 1) * NX;
        double l[NX], gu[NU], gx[NX];
        for (int r = 0; r < NX; r++) {
            double s = pn[r];
            for (int m = 0; m < NX; m++) s += Pn[r * NX + m] * b[i * NX + m];
            l[r] = s;
        }
        for (int c = 0; c < NX; c++) {
            double s = q[i * NX + c];
            for (int m = 0; m < NX; m++) s += Ai[m * NX + c] * l[m];
            gx[c] = s;
        }
        for (int c = 0; c < NU; c++) {
            double s = rt[i * NU + c];
            for (int m = 0; m < NX; m++) s += Bi[m * NU + c] * l[m];
            gu[c] = s;
        }
        const double* K = w->K + (size_t)i * 48;
        double* pi_ = w->pv + (size_t)i * NX;
        for (int c = 0; c < NX; c++) {
            double s = gx[c];
            for (int m = 0; m < NU; m++) s += K[m * NX + c] * gu[m];
            pi_[c] = s;
        }
        double kk[4] = {gu[0], gu[1], gu[2], gu[3]};
        chol4_solve(w->L + (size_t)i * 16, kk);
        for (int m = 0; m < NU; m++) w->kff[i * NU + m] = -kk[m];
    }
    memcpy(w->xs, d0, NX * sizeof(double));
    for (int i = 0; i < N; i++) {
        const double* Ai = A + (size_t)i * 144;
        const double* Bi = B + (size_t)i * 48;
        const double* K = w->K + (size_t)i * 48;
        const double* xi = w->xs + (size_t)i * NX;
        double* vi = w->vs + (size_t)i * NU;
        double* xn = w->xs + (size_t)(i + 1) * NX;
        for (int m = 0; m < NU; m++) {
            double s = w->kff[i * NU + m];
            for (int c = 0; c < NX; c++) s += K[m * NX + c] * xi[c];
            vi[m] = s;
        }
        for (int r = 0; r < NX; r++) {
            double s = b[i * NX + r];
            for (int c = 0; c < NX; c++) s += Ai[r * NX + c] * xi[c];
            for (int c = 0; c < NU; c++) s += Bi[r * NU + c] * vi[c];
            xn[r] = s;
        }
        const double* Pn = w->P + (size_t)(i + 1) * 144;
        const double* pn = w->pv + (size_t)(i + 1) * NX;
        for (int r = 0; r < NX; r++) {
            double s = pn[r];
            for (int c = 0; c < NX; c++) s += Pn[r * NX + c] * xn[c];
            w->pis[i * NX + r] = s;
        }
    }
}

/* given inputs v: roll the linear dynamics out, run the adjoint recursion, return the u-stationarity residual
 * inf-norm || Rd v + r + B'pi - lam_l + lam_u ||; if gout != NULL also store g = Rd v + r + B'pi (N*4) */
static double rollout_adjoint(int N, const double* A, const double* B, const double* b, const double* Qd, const double* q,
                              const double* Rd, const double* r, const double* d0, const double* v, const double* lam,
                              double* xs, double* pis, double* gout) {
    memcpy(xs, d0, NX * sizeof(double));
    for (int i = 0; i < N; i++) {
        const double* Ai = A + (size_t)i * 144;
        const double* Bi = B + (size_t)i * 48;
        for (int rr = 0; rr < NX; rr++) {
            double s = b[i * NX + rr];
            for (int c = 0; c < NX; c++) s += Ai[rr * NX + c] * xs[i * NX + c];
            for (int c = 0; c < NU; c++) s += Bi[rr * NU + c] * v[i * NU + c];
            xs[(i + 1) * NX + rr] = s;
        }
    }
    double res = 0.0;
    for (int i = N - 1; i >= 0; i--) {
        double* pi_ = pis + (size_t)i * NX; /* multiplier of the equation producing x_{i+1} */
        for (int c = 0; c < NX; c++) {
            double s = Qd[(i + 1) * NX + c] * xs[(i + 1) * NX + c] + q[(i + 1) * NX + c];
            if (i + 1 < N) {
                const double* An = A + (size_t)(i + 1) * 144;
                for (int m = 0; m < NX; m++) s += An[m * NX + c] * pis[(i + 1) * NX + m];
            }
            pi_[c] = s;
        }
        const double* Bi = B + (size_t)i * 48;
        for (int c = 0; c < NU; c++) {
            double s = Rd[i * NU + c] * v[i * NU + c] + r[i * NU + c];
            for (int m = 0; m < NX; m++) s += Bi[m * NU + c] * pi_[m];
            if (gout) gout[i * NU + c] = s;
            s += -lam[i * 8 + c] + lam[i * 8 + 4 + c];
            if (fabs(s) > res) res = fabs(s);
        }
    }
    return res;
}

/* IPM constants (the build's own; HPIPM's internals are not mimicked -- the minimiser is unique) */
/* Start and step rule of the interior-point loop.  Chosen on the oracle over the test workloads (mixed 25 %-saturated batches at
 * N = 10..80, the config-4 candidates, forced interior point): against the textbook 0.1 / 0.995 / mu0 = g0 the mean iteration
 * count drops from 7.1 to 4.2 on saturated instances and from 4.4 to 2.3 without active bounds, same minimiser to 1e-8, same
 * status histogram.  The GPU kernel uses the same three numbers (qp_kernel.hip). */
#define IPM_TAU0 0.05    /* interior push of the starting point, fraction of the box width (round 2: 0.003 -- but the loop is now the
                          * fallback for the QPs the active-set rounds do not finish, and those start better from further inside:
                          * worst QP of the test workloads 28 -> 20 Newton systems, means unchanged) */
#define IPM_FTB 0.9999   /* fraction to the boundary of a (nearly) full step */
#define IPM_FTBLO 0.9    /* ... of a blocked step: alpha = a ((1 - a) FTBLO + a FTB), a = min(1, step to the boundary) */
#define IPM_MU0F 0.1     /* initial complementarity target = IPM_MU0F * stationarity residual of the clamped point */
/* Safeguards found with the randomised-options test (tests/test_gpu_parity.py::test_randomised_options_against_oracle):
 *  - a blocked step that goes 99.99 % of the way to the boundary leaves complementarity products 1e-5 of the average behind;
 *    the next predictor is then blocked at once and plain Mehrotra falls into a limit cycle.  A short step now stops 10 % short
 *    of the boundary, a (nearly) full one still takes 99.99 % (the same idea as HPIPM's step-length dependent fraction);
 *  - the interior-point loop ends when every bound is resolved to qp_tol_mu (the input within that distance of it, or its
 *    multiplier too small to move the input that far) and the tracked stationarity residual is below qp_tol_stat.
 *
 * ACTIVE-SET POLISH (round 3).  An interior-point method approaches a degenerate bound (active, vanishing multiplier) like
 * sqrt(mu), and mu has a floor in FP64: round 2 stopped such QPs through a "stall" escape 1e-6 .. 3e-5 away from the minimiser
 * and called that success.  The QP is strictly convex with box constraints only, so its minimiser is characterised exactly by
 * its active set: pin the inputs of a guessed active set at their bounds, solve the remaining EQUALITY-constrained QP with one
 * Riccati factorisation (pinning = a 1e30 entry on the input's Hessian diagonal: the same code path as an interior-point Newton
 * system), and check the two conditions that make the result THE minimiser: free inputs inside the box, multipliers of pinned
 * inputs of the right sign.  A wrong guess is repaired the primal-dual active-set way (violating free inputs are pinned, pinned
 * inputs with a wrong-signed multiplier are released).  Schedule:
 *   - a first round of at most POL_FIRST tries straight from the equality-constrained minimiser (guess: the inputs that violate
 *     their bounds) -- on the standard workloads this finishes 90 % of the QPs with ONE factorisation and the rest with two or
 *     three.  A round ends early when a try had to repair more than POL_NCHG inputs or more than the try before it (measured on
 *     8000 QPs: converging guesses repair 1..9 inputs, fewer each time; hopeless ones 40..120 every time);
 *   - then interior-point iterations from the last active-set point, followed by a round of at most POL_LOOP tries from the
 *     iterate's classification (lambda / R > t <=> active) -- after a failed round only once mu has been halved again.  The
 *     interior-point loop remains the globally convergent fallback, and may still end by its own rule; the stall escape is gone
 *     (a QP that exhausts qp_iter_max Newton systems returns status 2).
 * Against independent BVLS answers (scripts/dev/polish_eval.py): 4e-12 on 48 random QPs, 2e-11 on the hard ones (N = 57 / 80,
 * hundreds of active bounds); the interior-point rule alone: 9e-8 and 1.1e-6.  stats[0] counts Newton systems (interior-point
 * iterations + active-set tries). */
#define POL_BIG 1e30      /* Hessian entry that pins an input */
#define POL_FIRST 5       /* active-set tries before the first interior-point iteration (at most) */
#define POL_LOOP 3        /* ... per round after an interior-point iteration (at most) */
#define POL_NCHG 8        /* a round ends when a try repairs more than this many inputs, or more than the try before it */
#define POL_MU_GATE 0.5   /* after a failed round the next one waits until the interior-point loop has cut mu by this factor ... */
#define POL_ALPHA_GATE 0.9 /* ... and has just taken a (nearly) full step: a blocked iterate classifies the bounds poorly */
#define POL_TOL_G 1e-9    /* wrong-signed multiplier of a pinned input: tolerated up to POL_TOL_G * R (+ POL_TOL_GREL * |g|max) */
#define POL_TOL_GREL 1e-13

size_t orc_ws_doubles(int N) {
    const size_t nv = (size_t)N * NU;
    return /* qp */ (size_t)(N + 1) * 144 + (size_t)N * 48 + (size_t)N * 16 + (size_t)(N + 1) * NX + (size_t)N * NU +
           (size_t)(N + 1) * NX + (size_t)N * NU + (size_t)N * NX + 13 * nv + (size_t)N * 8 +
           /* rti */ (size_t)N * 144 + (size_t)N * 48 + (size_t)N * NX + 2 * (size_t)(N + 1) * NX + 4 * (size_t)N * NU +
           (size_t)(N + 1) * NX + (size_t)N * NU + (size_t)N * NX + (size_t)N * 8 + 64;
}

/* one equality-constrained solve with the inputs act[j] != 0 pinned at their bounds (-1 lower, +1 upper).  Returns 1 when the
 * result is the QP's minimiser (-> vp, with xs / pis / g of that point), 0 when the guess was wrong (act_new = repaired guess,
 * vp = the equality-constrained solution, which may leave the box), -4 / -1 on a factorisation failure / NaN. */
static int polish_try(qp_ws* w, const double* A, const double* B, const double* b, const double* Qd, const double* q,
                      const double* Rd, const double* r, const double* d0, const double* lb, const double* ub, const double* act,
                      double* gam, double* rt, double* vp, double* xs, double* pis, double* g, double* act_new, const double* lamz) {
    const int N = w->N, nv = N * NU;
    for (int j = 0; j < nv; j++) {
        gam[j] = act[j] != 0.0 ? POL_BIG : 0.0;
        rt[j] = r[j] - gam[j] * (act[j] < 0.0 ? lb[j] : ub[j]);
    }
    if (ric_factor(w, A, B, Qd, Rd, gam)) return -4;
    ric_solve(w, A, B, b, q, rt, d0);
    int ok = 1;
    for (int j = 0; j < nv; j++) {
        act_new[j] = act[j];
        const double vj = w->vs[j];
        if (!(vj == vj)) return -1;
        vp[j] = act[j] != 0.0 ? (act[j] < 0.0 ? lb[j] : ub[j]) : vj;   /* pinned inputs exactly onto their bounds */
        if (act[j] == 0.0 && vj < lb[j]) { ok = 0; act_new[j] = -1.0; }   /* a free input that leaves the box is pinned next time */
        if (act[j] == 0.0 && vj > ub[j]) { ok = 0; act_new[j] = 1.0; }
    }
    /* multipliers of this point (the primal-dual active-set method evaluates them at the equality-constrained solution itself) */
    (void)rollout_adjoint(N, A, B, b, Qd, q, Rd, r, d0, vp, lamz, xs, pis, g);
    double gmax = 0.0;
    for (int j = 0; j < nv; j++) if (fabs(g[j]) > gmax) gmax = fabs(g[j]);
    for (int j = 0; j < nv; j++) {
        const double tolg = POL_TOL_G * Rd[j] + POL_TOL_GREL * gmax;
        if ((act[j] < 0.0 && g[j] < -tolg) || (act[j] > 0.0 && g[j] > tolg)) { ok = 0; act_new[j] = 0.0; }
    }
    return ok;
}

int orc_qp_solve_ws(const orc_opts* o, const double* A, const double* B, const double* b, const double* Qd, const double* q,
                    const double* Rd, const double* r, const double* d0, const double* lb, const double* ub, double* dx,
                    double* du, double* pi, double* lam, double* stats, double* mem) {
    const int N = o->N, nv = N * NU;
    qp_ws w;
    w.N = N;
    double* m = mem;
    w.P = m; m += (size_t)(N + 1) * 144;
    w.K = m; m += (size_t)N * 48;
    w.L = m; m += (size_t)N * 16;
    w.pv = m; m += (size_t)(N + 1) * NX;
    w.kff = m; m += (size_t)N * NU;
    w.xs = m; m += (size_t)(N + 1) * NX;
    w.vs = m; m += (size_t)N * NU;
    w.pis = m; m += (size_t)N * NX;
    double* gam = m; m += nv;
    double* rt = m; m += nv;
    double* v = m; m += nv;
    double* tl = m; m += nv;
    double* tu = m; m += nv;
    double* ll = m; m += nv;
    double* lu = m; m += nv;
    double* dva = m; m += nv;
    double* vp = m; m += nv;
    double* gp = m; m += nv;
    double* act = m; m += nv;
    double* act_new = m; m += nv;
    double* lamz = m; m += (size_t)N * 8;
    int status = 0, nsys = 0, early = 0, ipm_on = 0, converged = 0, round_k = 0, round_cap = 0, nchg_prev = nv + 1;
    double mu = 0.0, rho = 0.0, mu_gate = 1e300;
    memset(lamz, 0, (size_t)N * 8 * sizeof(double));

    /* step 0: equality-constrained minimiser (Gamma = 0) */
    memset(gam, 0, nv * sizeof(double));
    if (ric_factor(&w, A, B, Qd, Rd, gam)) { status = 4; goto done; }
    ric_solve(&w, A, B, b, q, r, d0);
    {
        int feas = 1;
        for (int j = 0; j < nv; j++) {
            if (!(w.vs[j] >= lb[j] && w.vs[j] <= ub[j])) feas = 0;
            act[j] = w.vs[j] < lb[j] ? -1.0 : (w.vs[j] > ub[j] ? 1.0 : 0.0);   /* first active-set guess */
            vp[j] = w.vs[j];
        }
        if (feas && o->qp_early_exit) {
            memcpy(du, w.vs, nv * sizeof(double));
            memcpy(dx, w.xs, (size_t)(N + 1) * NX * sizeof(double));
            memcpy(pi, w.pis, (size_t)N * NX * sizeof(double));
            memset(lam, 0, (size_t)N * 8 * sizeof(double));
            early = 1;
            goto done;
        }
    }
    status = 2;
    round_cap = POL_FIRST;
    while (nsys < o->qp_iter_max) {
        if (round_k < round_cap) {   /* ---- active-set try */
            nsys++;
            round_k++;
            const int pr = polish_try(&w, A, B, b, Qd, q, Rd, r, d0, lb, ub, act, gam, rt, vp, w.xs, w.pis, gp, act_new, lamz);
            if (pr < 0) { status = -pr; break; }
            if (pr) { TRACE("  try   nsys %2d  ACCEPTED\n", nsys); memcpy(v, vp, nv * sizeof(double)); status = 0; break; }
            /* repaired guess; the round goes on while the repairs are few and do not grow (a guess that is converging) */
            int nchg = 0;
            for (int j = 0; j < nv; j++) { nchg += act[j] != act_new[j]; act[j] = act_new[j]; }
            TRACE("  try   nsys %2d  repairs %3d  (round %d/%d, ipm %d, mu %.2e)\n", nsys, nchg, round_k, round_cap, ipm_on, mu);
            /* (the first round, straight from the Newton point, is the patient one: a growing but small number of repairs is normal there --
             * 1, 3, 1, 0 -- while a round that starts from an interior-point iterate and gets worse will not recover) */
            if (nchg > POL_NCHG || (nchg > nchg_prev && ipm_on)) round_cap = 0;
            nchg_prev = nchg;
            if (round_k >= round_cap) {   /* the round has failed: the next one waits for the interior-point loop to halve mu */
                if (ipm_on) mu_gate = mu;
                if (converged) break;
            }
            continue;
        }
        if (!ipm_on) {   /* ---- interior start at the last active-set point: clamp into the box, multipliers from mu0 */
            ipm_on = 1;
            for (int j = 0; j < nv; j++) {
                const double wdt = ub[j] - lb[j];
                double vj = vp[j];
                const double lo = lb[j] + IPM_TAU0 * wdt, hi = ub[j] - IPM_TAU0 * wdt;
                if (vj < lo) vj = lo;
                if (vj > hi) vj = hi;
                v[j] = vj;
                tl[j] = vj - lb[j];
                tu[j] = ub[j] - vj;
            }
            /* multiplier scale: stationarity residual of the clamped point without multipliers */
            double g0 = rollout_adjoint(N, A, B, b, Qd, q, Rd, r, d0, v, lamz, w.xs, w.pis, NULL);
            double mu0 = IPM_MU0F * g0;
            if (mu0 < 1e-4) mu0 = 1e-4;
            for (int j = 0; j < nv; j++) { ll[j] = mu0 / tl[j]; lu[j] = mu0 / tu[j]; }
            for (int i = 0; i < N; i++)
                for (int c = 0; c < NU; c++) { lam[i * 8 + c] = ll[i * NU + c]; lam[i * 8 + 4 + c] = lu[i * NU + c]; }
            rho = rollout_adjoint(N, A, B, b, Qd, q, Rd, r, d0, v, lam, w.xs, w.pis, NULL);
        }
        /* ---- interior-point iteration (Mehrotra predictor-corrector) */
        nsys++;
        mu = 0.0;
        for (int j = 0; j < nv; j++) mu += ll[j] * tl[j] + lu[j] * tu[j];
        mu /= (2.0 * nv);
        for (int j = 0; j < nv; j++) gam[j] = ll[j] / tl[j] + lu[j] / tu[j];
        if (ric_factor(&w, A, B, Qd, Rd, gam)) { status = 4; break; }
        /* predictor (sigma = 0) */
        for (int j = 0; j < nv; j++) rt[j] = r[j] - gam[j] * v[j];
        ric_solve(&w, A, B, b, q, rt, d0);
        double aaff = 1.0;
        for (int j = 0; j < nv; j++) {
            const double dv = w.vs[j] - v[j];
            dva[j] = dv;
            const double dll = -ll[j] - ll[j] / tl[j] * dv, dlu = -lu[j] + lu[j] / tu[j] * dv;
            if (dv < 0 && -tl[j] / dv < aaff) aaff = -tl[j] / dv;
            if (dv > 0 && tu[j] / dv < aaff) aaff = tu[j] / dv;
            if (dll < 0 && -ll[j] / dll < aaff) aaff = -ll[j] / dll;
            if (dlu < 0 && -lu[j] / dlu < aaff) aaff = -lu[j] / dlu;
        }
        double muaff = 0.0;
        for (int j = 0; j < nv; j++) {
            const double dv = dva[j];
            const double dll = -ll[j] - ll[j] / tl[j] * dv, dlu = -lu[j] + lu[j] / tu[j] * dv;
            muaff += (ll[j] + aaff * dll) * (tl[j] + aaff * dv) + (lu[j] + aaff * dlu) * (tu[j] - aaff * dv);
        }
        muaff /= (2.0 * nv);
        double sigma = muaff / mu;
        sigma = sigma * sigma * sigma;
        /* corrector */
        for (int j = 0; j < nv; j++) {
            const double dv = dva[j];
            const double dll = -ll[j] - ll[j] / tl[j] * dv, dlu = -lu[j] + lu[j] / tu[j] * dv;
            const double cl = dll * dv, cu = -dlu * dv;
            rt[j] = r[j] - gam[j] * v[j] - (sigma * mu - cl) / tl[j] + (sigma * mu - cu) / tu[j];
        }
        ric_solve(&w, A, B, b, q, rt, d0);
        double amax = 1e300;
        for (int j = 0; j < nv; j++) {
            const double dva_ = dva[j];
            const double dlla = -ll[j] - ll[j] / tl[j] * dva_, dlua = -lu[j] + lu[j] / tu[j] * dva_;
            const double cl = dlla * dva_, cu = -dlua * dva_;
            const double dv = w.vs[j] - v[j];
            const double dll = (sigma * mu - cl) / tl[j] - ll[j] - ll[j] / tl[j] * dv;
            const double dlu = (sigma * mu - cu) / tu[j] - lu[j] + lu[j] / tu[j] * dv;
            if (dv < 0 && -tl[j] / dv < amax) amax = -tl[j] / dv;
            if (dv > 0 && tu[j] / dv < amax) amax = tu[j] / dv;
            if (dll < 0 && -ll[j] / dll < amax) amax = -ll[j] / dll;
            if (dlu < 0 && -lu[j] / dlu < amax) amax = -lu[j] / dlu;
            gam[j] = dll; /* reuse as storage for the dual steps */
            rt[j] = dlu;
        }
        const double a = amax < 1.0 ? amax : 1.0;
        const double alpha = (IPM_FTB * amax >= 1.0) ? 1.0 : a * ((1.0 - a) * IPM_FTBLO + a * IPM_FTB);
        int bad = 0;
        for (int j = 0; j < nv; j++) {
            const double dv = w.vs[j] - v[j];
            v[j] += alpha * dv;
            tl[j] += alpha * dv;
            tu[j] -= alpha * dv;
            ll[j] += alpha * gam[j];
            lu[j] += alpha * rt[j];
            if (!(v[j] == v[j])) bad = 1;
        }
        if (bad) { status = 1; break; }
        rho *= (1.0 - alpha);
        mu = 0.0;
        for (int j = 0; j < nv; j++) mu += ll[j] * tl[j] + lu[j] * tu[j];
        mu /= (2.0 * nv);
        /* the loop's own termination: every bound is resolved -- either the input is within qp_tol_mu of it, or its multiplier is
         * too small to move the input by qp_tol_mu (lambda / R, R = the input's own weight, a lower bound of the curvature) -- and
         * the tracked stationarity residual is below qp_tol_stat.  The same two quantities classify the bounds for the next
         * active-set tries: active <=> the multiplier could move the input further than it is away from the bound. */
        double unres = 0.0;
        for (int j = 0; j < nv; j++) {
            const double al = ll[j] / Rd[j], au = lu[j] / Rd[j];
            const double a_ = al < tl[j] ? al : tl[j], c_ = au < tu[j] ? au : tu[j];
            if (a_ > unres) unres = a_;
            if (c_ > unres) unres = c_;
            act[j] = al > tl[j] ? -1.0 : (au > tu[j] ? 1.0 : 0.0);
            vp[j] = v[j];
        }
        /* the loop's own rule is met: one more round for the exact answer; if that fails too the iterate is the answer (to qp_tol_mu) */
        TRACE("  ipm   nsys %2d  mu %.2e  alpha %.3f  unres %.2e  rho %.2e\n", nsys, mu, alpha, unres, rho);
        if (unres <= o->qp_tol_mu && rho <= o->qp_tol_stat) converged = 1;
        if (converged || (mu <= POL_MU_GATE * mu_gate && alpha >= POL_ALPHA_GATE)) { round_k = 0; round_cap = POL_LOOP; nchg_prev = nv + 1; }
    }
    /* consistent primal/dual output for the final inputs.  The IPM multipliers carry an absolute error ~ eps*Gamma*|v|
     * on active bounds (Gamma = lam/t -> 1e12+), so the multipliers are recovered from the gradient instead:
     * g = Rd v + r + B'pi, lam_l = max(g,0), lam_u = max(-g,0)  (stationarity then holds exactly; what is left of the
     * KKT error shows up as complementarity, reported in stats[2]) */
    if (status == 4 || status == 1) goto done;
    if (converged && status == 2) status = 0;
    if (status == 2 && !ipm_on)   /* limit reached before the first interior-point iteration: the last point, clamped into the box */
        for (int j = 0; j < nv; j++) v[j] = vp[j] < lb[j] ? lb[j] : (vp[j] > ub[j] ? ub[j] : vp[j]);
    (void)rollout_adjoint(N, A, B, b, Qd, q, Rd, r, d0, v, lamz, dx, pi, gam);
    rho = 0.0;
    for (int i = 0; i < N; i++)
        for (int c = 0; c < NU; c++) {
            const double g = gam[i * NU + c];
            const double l_lo = g > 0 ? g : 0.0, l_up = g < 0 ? -g : 0.0;
            lam[i * 8 + c] = l_lo;
            lam[i * 8 + 4 + c] = l_up;
            const double cl = l_lo * (v[i * NU + c] - lb[i * NU + c]), cu = l_up * (ub[i * NU + c] - v[i * NU + c]);
            if (cl > rho) rho = cl;
            if (cu > rho) rho = cu;
        }
    memcpy(du, v, nv * sizeof(double));
done:
    if (stats) { stats[0] = nsys; stats[1] = mu; stats[2] = rho; stats[3] = early; }
    return status;
}

int orc_qp_solve(const orc_opts* o, const double* A, const double* B, const double* b, const double* Qd, const double* q,
                 const double* Rd, const double* r, const double* d0, const double* lb, const double* ub, double* dx,
                 double* du, double* pi, double* lam, double* stats) {
    double* mem = (double*)malloc(orc_ws_doubles(o->N) * sizeof(double));
    const int st = orc_qp_solve_ws(o, A, B, b, Qd, q, Rd, r, d0, lb, ub, dx, du, pi, lam, stats, mem);
    free(mem);
    return st;
}

/* ---------------------------------------------------------------------------------------------------------
 * SQP_RTI step: preparation (linearise) + feedback (QP, full step): acados_solver_bluerov2.c:623-672,945-951.
 * Cost scaling Ts on stages 0..N-1 (:393), terminal unscaled.  Iterate is not shifted between calls.
 * ------------------------------------------------------------------------------------------------------- */
int orc_rti_step_ws(const orc_opts* o, const double* x0, const double* yref, const double* p, const double* drp, double* x,
                    double* u, double* pi, double* lam, orc_result* res, double* Aout, double* Bout, double* bout,
                    double* qp_stats, double* mem) {
    const int N = o->N;
    double* m = mem;
    double* A = m; m += (size_t)N * 144;
    double* B = m; m += (size_t)N * 48;
    double* b = m; m += (size_t)N * NX;
    double* Qd = m; m += (size_t)(N + 1) * NX;
    double* q = m; m += (size_t)(N + 1) * NX;
    double* Rd = m; m += (size_t)N * NU;
    double* r = m; m += (size_t)N * NU;
    double* lb = m; m += (size_t)N * NU;
    double* ub = m; m += (size_t)N * NU;
    double* dx = m; m += (size_t)(N + 1) * NX;
    double* du = m; m += (size_t)N * NU;
    double* pin = m; m += (size_t)N * NX;
    double* lamn = m; m += (size_t)N * 8;
    m += (8 - ((m - mem) & 7)) & 7;   /* the QP's share of the workspace, 64-byte aligned */
    double d0[NX];

    /* preparation: ERK4 + sensitivities on every interval */
    for (int i = 0; i < N; i++) {
        double xn[NX];
        orc_rk4_sens6(x + (size_t)i * NX, u + (size_t)i * NU, p + (size_t)i * NP, drp ? drp + (size_t)i * 2 : NULL,
                      o->ts_vec ? o->ts_vec[i] : o->Ts, xn, A + (size_t)i * 144, B + (size_t)i * 48);
        for (int j = 0; j < NX; j++) b[i * NX + j] = xn[j] - x[(i + 1) * NX + j];
    }
    /* Gauss-Newton LS cost: y = [x;u], J = I, Hess = s W, grad = s W (y - yref) */
    for (int i = 0; i < N; i++) {
        const double sc = o->ts_vec ? o->ts_vec[i] : o->Ts;            /* cost scaling = the stage's time step (:393, :126-127) */
        const double* Wi = (i == 0 && o->W0) ? o->W0 : o->W;           /* separate stage-0 weight (:422-441) */
        for (int j = 0; j < NX; j++) {
            Qd[i * NX + j] = sc * Wi[j];
            q[i * NX + j] = sc * Wi[j] * (x[i * NX + j] - yref[i * NY + j]);
        }
        for (int j = 0; j < NU; j++) {
            Rd[i * NU + j] = sc * Wi[NX + j];
            r[i * NU + j] = sc * Wi[NX + j] * (u[i * NU + j] - yref[i * NY + NX + j]);
            lb[i * NU + j] = o->lbu[j] - u[i * NU + j];
            ub[i * NU + j] = o->ubu[j] - u[i * NU + j];
        }
    }
    for (int j = 0; j < NX; j++) {
        Qd[N * NX + j] = o->We[j];
        q[N * NX + j] = o->We[j] * (x[N * NX + j] - yref[N * NY + j]);
        d0[j] = x0[j] - x[j];
    }
    /* NLP KKT residual of the entering iterate with the stored multipliers */
    double kkt = 0.0;
#define UPD(vv) do { double a_ = fabs(vv); if (a_ > kkt) kkt = a_; if (a_ != a_) kkt = a_; } while (0)
    for (int j = 0; j < NX; j++) UPD(d0[j]);
    for (int i = 0; i < N; i++) {
        for (int j = 0; j < NX; j++) UPD(b[i * NX + j]);
        for (int c = 0; c < NU; c++) {
            double s = r[i * NU + c] - lam[i * 8 + c] + lam[i * 8 + 4 + c];
            for (int k = 0; k < NX; k++) s += B[(size_t)i * 48 + k * NU + c] * pi[i * NX + k];
            UPD(s);
            const double sl = u[i * NU + c] - o->lbu[c], su = o->ubu[c] - u[i * NU + c];
            if (sl < 0) UPD(sl);
            if (su < 0) UPD(su);
            UPD(lam[i * 8 + c] * sl);
            UPD(lam[i * 8 + 4 + c] * su);
        }
        if (i >= 1)
            for (int c = 0; c < NX; c++) {
                double s = q[i * NX + c] - pi[(i - 1) * NX + c];
                for (int k = 0; k < NX; k++) s += A[(size_t)i * 144 + k * NX + c] * pi[i * NX + k];
                UPD(s);
            }
    }
    for (int c = 0; c < NX; c++) UPD(q[N * NX + c] - pi[(N - 1) * NX + c]);
#undef UPD

    double st[4] = {0, 0, 0, 0};
    int status = orc_qp_solve_ws(o, A, B, b, Qd, q, Rd, r, d0, lb, ub, dx, du, pin, lamn, st, m);
    if (status == 0 || status == 2) {
        int nan = 0;
        for (int j = 0; j < (N + 1) * NX; j++) if (dx[j] != dx[j]) nan = 1;
        for (int j = 0; j < N * NU; j++) if (du[j] != du[j]) nan = 1;
        if (nan) status = 1;
    }
    if (status == 0 || status == 2) { /* full step, fixed_step globalisation (:623) */
        for (int j = 0; j < (N + 1) * NX; j++) x[j] += dx[j];
        for (int j = 0; j < N * NU; j++) u[j] += du[j];
        memcpy(pi, pin, (size_t)N * NX * sizeof(double));
        memcpy(lam, lamn, (size_t)N * 8 * sizeof(double));
    }
    double cost = 0.0;
    for (int i = 0; i < N; i++) {
        const double sc = o->ts_vec ? o->ts_vec[i] : o->Ts;
        const double* Wi = (i == 0 && o->W0) ? o->W0 : o->W;
        for (int j = 0; j < NX; j++) { double e = x[i * NX + j] - yref[i * NY + j]; cost += 0.5 * sc * Wi[j] * e * e; }
        for (int j = 0; j < NU; j++) { double e = u[i * NU + j] - yref[i * NY + NX + j]; cost += 0.5 * sc * Wi[NX + j] * e * e; }
    }
    for (int j = 0; j < NX; j++) { double e = x[N * NX + j] - yref[N * NY + j]; cost += 0.5 * o->We[j] * e * e; }
    const int failed = !(status == 0 || status == 2);
    int x0_ok = 1; /* a restart needs a usable measurement */
    for (int j = 0; j < NX; j++) if (!(fabs(x0[j]) < 1e300)) x0_ok = 0;
    if (failed && o->on_failure == 1 && x0_ok) { /* cold restart at the measured state */
        for (int i = 0; i <= N; i++) memcpy(x + (size_t)i * NX, x0, NX * sizeof(double));
        memset(u, 0, (size_t)N * NU * sizeof(double));
        memset(pi, 0, (size_t)N * NX * sizeof(double));
        memset(lam, 0, (size_t)N * 8 * sizeof(double));
    }
    if (res) {
        for (int j = 0; j < NU; j++) {
            if (!failed) res->u0[j] = u[j];
            else { /* hold the last successful input, sanitised */
                double v = res->u0[j];
                if (v != v) v = 0.0;
                v = v < o->lbu[j] ? o->lbu[j] : v;
                v = v > o->ubu[j] ? o->ubu[j] : v;
                res->u0[j] = v;
            }
        }
        orc_thrust_alloc(res->u0, res->thrust);
        res->cost = cost;
        res->kkt = kkt;
        res->status = status;
        res->qp_iter = (int)st[0];
    }
    if (Aout) memcpy(Aout, A, (size_t)N * 144 * sizeof(double));
    if (Bout) memcpy(Bout, B, (size_t)N * 48 * sizeof(double));
    if (bout) memcpy(bout, b, (size_t)N * NX * sizeof(double));
    if (qp_stats) memcpy(qp_stats, st, sizeof st);
    return status;
}

int orc_rti_step(const orc_opts* o, const double* x0, const double* yref, const double* p, double* x, double* u, double* pi,
                 double* lam, orc_result* res, double* Aout, double* Bout, double* bout, double* qp_stats) {
    double* mem = (double*)malloc(orc_ws_doubles(o->N) * sizeof(double));
    const int st = orc_rti_step_ws(o, x0, yref, p, NULL, x, u, pi, lam, res, Aout, Bout, bout, qp_stats, mem);
    free(mem);
    return st;
}

/* the 6-disturbance model variant: drp[(N+1)*2] = per-stage roll / pitch disturbance moments */
int orc_rti_step6(const orc_opts* o, const double* x0, const double* yref, const double* p, const double* drp, double* x,
                  double* u, double* pi, double* lam, orc_result* res, double* qp_stats) {
    double* mem = (double*)malloc(orc_ws_doubles(o->N) * sizeof(double));
    const int st = orc_rti_step_ws(o, x0, yref, p, drp, x, u, pi, lam, res, NULL, NULL, NULL, qp_stats, mem);
    free(mem);
    return st;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* OpenMP over instances.  One preallocated workspace per thread (round 2 did two malloc / free of > 128 KB per instance and step:
 * mmap / munmap and page faults under the process' mmap lock -- 4 % parallel efficiency on 128 threads), static schedule. */
int orc_rti_step_batch6(const orc_opts* o, int nb, const double* x0, const double* yref, const double* p, const double* drp,
                        double* x, double* u, double* pi, double* lam, orc_result* res, int nthreads) {
    const int N = o->N;
    int worst = 0;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
    if (nthreads > nb) nthreads = nb > 0 ? nb : 1;
#pragma omp parallel num_threads(nthreads) reduction(max : worst)
#else
    (void)nthreads;
#endif
    {
        static __thread double* tls_mem = NULL;
        static __thread size_t tls_n = 0;
        const size_t need = orc_ws_doubles(N);
        if (tls_n < need) {
            free(tls_mem);
            tls_mem = (double*)malloc(need * sizeof(double));
            tls_n = need;
        }
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
        for (int k = 0; k < nb; k++) {
            int st = orc_rti_step_ws(o, x0 + (size_t)k * NX, yref + (size_t)k * (N + 1) * NY, p + (size_t)k * (N + 1) * NP,
                                     drp ? drp + (size_t)k * (N + 1) * 2 : NULL, x + (size_t)k * (N + 1) * NX,
                                     u + (size_t)k * N * NU, pi + (size_t)k * N * NX, lam + (size_t)k * N * 8, res + k, NULL, NULL,
                                     NULL, NULL, tls_mem);
            if (st > worst) worst = st;
        }
    }
    return worst;
}

int orc_rti_step_batch(const orc_opts* o, int nb, const double* x0, const double* yref, const double* p, double* x, double* u,
                       double* pi, double* lam, orc_result* res, int nthreads) {
    return orc_rti_step_batch6(o, nb, x0, yref, p, NULL, x, u, pi, lam, res, nthreads);
}

void orc_init_iterate(const orc_opts* o, double* x, double* u, double* pi, double* lam) {
    const int N = o->N;
    memset(x, 0, (size_t)(N + 1) * NX * sizeof(double));
    for (int i = 0; i <= N; i++) x[i * NX + 2] = -20.0; /* acados_solver_bluerov2.c:689-706 */
    memset(u, 0, (size_t)N * NU * sizeof(double));
    memset(pi, 0, (size_t)N * NX * sizeof(double));
    memset(lam, 0, (size_t)N * 8 * sizeof(double));
}

/* bluerov2_dobmpc/src/bluerov2_dob.cpp:390-395 */
void orc_thrust_alloc(const double* u0, double* t) {
    t[0] = (-u0[0] + u0[1] + u0[3]) / ROTOR;
    t[1] = (-u0[0] - u0[1] - u0[3]) / ROTOR;
    t[2] = (u0[0] + u0[1] - u0[3]) / ROTOR;
    t[3] = (u0[0] - u0[1] + u0[3]) / ROTOR;
    t[4] = (-u0[2]) / ROTOR;
    t[5] = (-u0[2]) / ROTOR;
}
