// lin_device.hpp -- building blocks of the wave-wide linearisation (qp_kernel.hip: lin_phase, used by rti_fused_kernel and
// by the streaming path's lin_wave_kernel): ERK4 of the state (gen/acados_solver_bluerov2.c:633-641: 4 stages, 1 step),
// sensitivity columns S[:,c] of S = d x+ / d [x;u] through stage records kept in LDS, NLP KKT rows.
#pragma once
#include "bluerov2_model.hpp"
#include "nmpc_device.hpp"

namespace brov {

__device__ __forceinline__ void kkt_upd(double& kkt, double v) {
    const double a = fabs(v);
    // NaN poisons the norm (oracle/bluerov2_oracle.c UPD macro)
    kkt = (a != a) ? a : ((kkt != kkt) ? kkt : fmax(kkt, a));
}

// inf-norm accumulator of the KKT rows with a separate NaN flag: v_max_f64 drops NaN operands, so the maximum is one
// independent instruction per row (the NaN-propagating kkt_upd is a chain of ~6 dependent ones)
struct KktAcc {
    double mx = 0.0;
    bool nan = false;
    __device__ __forceinline__ void upd(double v) { mx = fmax(mx, fabs(v)); nan = nan || (v != v); }
};

// Stage points of the 4 RK stages + x+
__device__ __forceinline__ void rk4_state(const double* __restrict__ xi, const Wrench& w, const ModelPar& m, double h,
                                          StagePoint (&sp)[4], double (&xn)[NX]) {
    double x[NX], k[NX], xs[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) x[j] = xi[j];
    model_f(x, w, m, k, sp[0]);
#pragma unroll
    for (int j = 0; j < NX; j++) { xn[j] = x[j] + (h / 6.0) * k[j]; xs[j] = x[j] + 0.5 * h * k[j]; }
    model_f(xs, w, m, k, sp[1]);
#pragma unroll
    for (int j = 0; j < NX; j++) { xn[j] += (h / 3.0) * k[j]; xs[j] = x[j] + 0.5 * h * k[j]; }
    model_f(xs, w, m, k, sp[2]);
#pragma unroll
    for (int j = 0; j < NX; j++) { xn[j] += (h / 3.0) * k[j]; xs[j] = x[j] + h * k[j]; }
    model_f(xs, w, m, k, sp[3]);
#pragma unroll
    for (int j = 0; j < NX; j++) xn[j] += (h / 6.0) * k[j];
}

// ---- stage records in LDS ---------------------------------------------------------------------------------------------
// The linearisation runs one wave per SIMD; its column loop is bound by instruction issue, and with the four stage points
// in registers the loop body exceeds the 256 architectural VGPRs (every use then pays v_accvgpr moves).  The stage points
// live in LDS instead: 17 doubles per RK stage = the 13 StagePoint values + the four velocity-damping diagonal
// entries of df/dx (which removes the model parameters from the loop).
typedef __attribute__((address_space(3))) double lds_f64;
constexpr int kRecStage = 17, kRecInterval = 4 * kRecStage;

__device__ __forceinline__ void store_stage_rec(lds_f64* r, const StagePoint& sp, const ModelPar& m) {
    r[0] = sp.sph; r[1] = sp.cph; r[2] = sp.sth; r[3] = sp.cth; r[4] = sp.sps; r[5] = sp.cps; r[6] = sp.icth;
    r[7] = sp.vu; r[8] = sp.vv; r[9] = sp.vw; r[10] = sp.wp; r[11] = sp.wq; r[12] = sp.wr;
    r[13] = (m.lx + 2.0 * m.qx * fabs(sp.vu)) * m.imx;
    r[14] = (m.ly + 2.0 * m.qy * fabs(sp.vv)) * m.imy;
    r[15] = (m.lz + 2.0 * m.qz * fabs(sp.vw)) * m.imz;
    r[16] = (m.ln + 2.0 * m.qn * fabs(sp.wr)) * m.imn;
}

struct StageRec { double v[kRecStage]; };
__device__ __forceinline__ StageRec load_stage_rec(const lds_f64* r) {
    StageRec R;
#pragma unroll
    for (int k = 0; k < kRecStage; k++) R.v[k] = r[k];
    return R;
}

// o = (df/dx)(stage record) * s + kb (kb = the column of df/du for an input column, zero otherwise); same entries as
// model_jvp, bluerov2_model.hpp
__device__ __forceinline__ void jvp_rec(const StageRec& R, double imx, double imy, double imz, double imn, const double (&kb)[4],
                                        const double (&s)[NX], double (&o)[NX]) {
    const double* r = R.v;
    const double sph = r[0], cph = r[1], sth = r[2], cth = r[3], sps = r[4], cps = r[5], icth = r[6];
    const double vu = r[7], vv = r[8], vw = r[9], wp = r[10], wq = r[11], wr = r[12];
    const double r00 = cps * cth, r01 = cps * sth * sph - sps * cph, r02 = sps * sph + cps * cph * sth;
    const double r10 = sps * cth, r11 = cps * cph + sph * sth * sps, r12 = sth * sps * cph - cps * sph;
    const double r21 = cth * sph, r22 = cth * cph;
    const double f0 = r00 * vu + r01 * vv + r02 * vw;
    const double f1 = r10 * vu + r11 * vv + r12 * vw;
    const double f2 = -sth * vu + r21 * vv + r22 * vw;
    o[0] = (r02 * vv - r01 * vw) * s[3] + (cps * f2) * s[4] - f1 * s[5] + r00 * s[6] + r01 * s[7] + r02 * s[8];
    o[1] = (r12 * vv - r11 * vw) * s[3] + (sps * f2) * s[4] + f0 * s[5] + r10 * s[6] + r11 * s[7] + r12 * s[8];
    o[2] = (r22 * vv - r21 * vw) * s[3] - (cth * vu + sth * (sph * vv + cph * vw)) * s[4] - sth * s[6] + r21 * s[7] + r22 * s[8];
    const double tth = sth * icth, ic2 = icth * icth;
    o[3] = (-sph * tth * wr) * s[3] + ((sps * wq + cph * wr) * ic2) * s[4] + (cps * tth * wq) * s[5] + s[9] + (sps * tth) * s[10] +
           (cph * tth) * s[11];
    o[4] = (cph * wr - sph * wq) * s[3] + cph * s[10] + sph * s[11];
    o[5] = ((cph * wq - sph * wr) * s[3] + (sph * wq + cph * wr) * tth * s[4] + sph * s[10] + cph * s[11]) * icth;
    o[6] = (-kBouy * cth * s[4]) * imx + r[13] * s[6] + kb[0];
    o[7] = (kBouy * (r22 * s[3] - sth * sph * s[4])) * imy + r[14] * s[7] + kb[1];
    o[8] = (-kBouy * (r21 * s[3] + sth * cph * s[4])) * imz + r[15] * s[8] + kb[2];
    o[9] = (kMzg * (sth * sph * s[4] - r22 * s[3]) + (kIy - kIz) * (wr * s[10] + wq * s[11])) * (1.0 / kIx);
    o[10] = (-kMzg * cth * s[4] + (kIz - kIx) * (wr * s[9] + wp * s[11])) * (1.0 / kIy);
    o[11] = (-(kIy - kIx) * (wq * s[9] + wp * s[10])) * imn + r[16] * s[11] + kb[3];
}

// two state columns through the same stage record: the ~45 Jacobian entries are derived once and applied twice
__device__ __forceinline__ void jvp_rec2(const StageRec& R, double imx, double imy, double imz, double imn, const double (&s)[NX],
                                         const double (&t)[NX], double (&o)[NX], double (&p)[NX]) {
    const double* r = R.v;
    const double sph = r[0], cph = r[1], sth = r[2], cth = r[3], sps = r[4], cps = r[5], icth = r[6];
    const double vu = r[7], vv = r[8], vw = r[9], wp = r[10], wq = r[11], wr = r[12];
    const double r00 = cps * cth, r01 = cps * sth * sph - sps * cph, r02 = sps * sph + cps * cph * sth;
    const double r10 = sps * cth, r11 = cps * cph + sph * sth * sps, r12 = sth * sps * cph - cps * sph;
    const double r21 = cth * sph, r22 = cth * cph;
    const double f0 = r00 * vu + r01 * vv + r02 * vw;
    const double f1 = r10 * vu + r11 * vv + r12 * vw;
    const double f2 = -sth * vu + r21 * vv + r22 * vw;
    const double tth = sth * icth, ic2 = icth * icth;
    const double a03 = r02 * vv - r01 * vw, a04 = cps * f2;
    const double a13 = r12 * vv - r11 * vw, a14 = sps * f2;
    const double a23 = r22 * vv - r21 * vw, a24 = cth * vu + sth * (sph * vv + cph * vw);
    const double a33 = -sph * tth * wr, a34 = (sps * wq + cph * wr) * ic2, a35 = cps * tth * wq, a3a = sps * tth, a3b = cph * tth;
    const double a43 = cph * wr - sph * wq;
    const double a53 = cph * wq - sph * wr, a54 = (sph * wq + cph * wr) * tth;
    const double a64 = -kBouy * cth * imx;
    const double a73 = kBouy * r22 * imy, a74 = -kBouy * sth * sph * imy;
    const double a83 = -kBouy * r21 * imz, a84 = -kBouy * sth * cph * imz;
    const double a93 = -kMzg * r22 * (1.0 / kIx), a94 = kMzg * sth * sph * (1.0 / kIx);
    const double a9a = (kIy - kIz) * (1.0 / kIx) * wr, a9b = (kIy - kIz) * (1.0 / kIx) * wq;
    const double aa4 = -kMzg * (1.0 / kIy) * cth, aa9 = (kIz - kIx) * (1.0 / kIy) * wr, aab = (kIz - kIx) * (1.0 / kIy) * wp;
    const double ab9 = -(kIy - kIx) * imn * wq, aba = -(kIy - kIx) * imn * wp;
#define BROV_JVP_ROWS(s, o)                                                                                 \
    o[0] = a03 * s[3] + a04 * s[4] - f1 * s[5] + r00 * s[6] + r01 * s[7] + r02 * s[8];                    \
    o[1] = a13 * s[3] + a14 * s[4] + f0 * s[5] + r10 * s[6] + r11 * s[7] + r12 * s[8];                    \
    o[2] = a23 * s[3] - a24 * s[4] - sth * s[6] + r21 * s[7] + r22 * s[8];                                \
    o[3] = a33 * s[3] + a34 * s[4] + a35 * s[5] + s[9] + a3a * s[10] + a3b * s[11];                       \
    o[4] = a43 * s[3] + cph * s[10] + sph * s[11];                                                        \
    o[5] = (a53 * s[3] + a54 * s[4] + sph * s[10] + cph * s[11]) * icth;                                  \
    o[6] = a64 * s[4] + r[13] * s[6];                                                                     \
    o[7] = a73 * s[3] + a74 * s[4] + r[14] * s[7];                                                        \
    o[8] = a83 * s[3] + a84 * s[4] + r[15] * s[8];                                                        \
    o[9] = a93 * s[3] + a94 * s[4] + a9a * s[10] + a9b * s[11];                                           \
    o[10] = aa4 * s[4] + aa9 * s[9] + aab * s[11];                                                        \
    o[11] = ab9 * s[9] + aba * s[10] + r[16] * s[11];
    BROV_JVP_ROWS(s, o)
    BROV_JVP_ROWS(t, p)
#undef BROV_JVP_ROWS
}
__device__ __forceinline__ void sens_column_rec2(const lds_f64* rec, const ModelPar& m, double h, int c0, int c1,
                                                 double (&acc0)[NX], double (&acc1)[NX]) {
    double k0[NX], s0[NX], k1[NX], s1[NX];
    StageRec R = load_stage_rec(rec);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NX; j++) { s0[j] = (j == c0) ? 1.0 : 0.0; s1[j] = (j == c1) ? 1.0 : 0.0; }
    jvp_rec2(R, m.imx, m.imy, m.imz, m.imn, s0, s1, k0, k1);
#pragma unroll
    for (int st = 1; st < 4; st++) {
        R = load_stage_rec(rec + st * kRecStage);
        __builtin_amdgcn_sched_barrier(0);
        const double wprev = (st == 1) ? h / 6.0 : h / 3.0, cs = (st == 3) ? h : 0.5 * h;
#pragma unroll
        for (int j = 0; j < NX; j++) {
            const double e0 = (j == c0) ? 1.0 : 0.0, e1 = (j == c1) ? 1.0 : 0.0;
            acc0[j] = (st == 1 ? e0 : acc0[j]) + wprev * k0[j]; s0[j] = e0 + cs * k0[j];
            acc1[j] = (st == 1 ? e1 : acc1[j]) + wprev * k1[j]; s1[j] = e1 + cs * k1[j];
        }
        __builtin_amdgcn_sched_barrier(0);
        jvp_rec2(R, m.imx, m.imy, m.imz, m.imn, s0, s1, k0, k1);
    }
#pragma unroll
    for (int j = 0; j < NX; j++) { acc0[j] += (h / 6.0) * k0[j]; acc1[j] += (h / 6.0) * k1[j]; }
}

// what the KKT rows of column c need besides the column itself; requested at the top of a trip so that the loads complete
// under the sensitivity arithmetic (indices clamped: every lane issues the same loads, the unused half is discarded)
struct KktOperands { double pm1c, ll, lu, ucur, lbu, ubu, grad, qn; };
// ig = global interval index (HBM arrays), il = index inside the LDS-resident chunk of nl intervals (row nl of q_s holds the
// terminal gradient when the chunk ends the horizon)
// the part of the operands that comes out of HBM / L2 (the rest is LDS or constant memory)
struct KktGlobal { double pm1c, ll, lu, ucur; };
__device__ __forceinline__ KktGlobal load_kkt_global(const DevParams& P, int b, int ig, int c, const double* __restrict__ ui) {
    const int N = P.N;
    const int jc = c - NX, ju = jc & 3, cx = c < NX ? c : NX - 1;
    const double* __restrict__ pim1 = P.pi + ((size_t)b * N + (ig > 0 ? ig - 1 : 0)) * NX;
    const double* __restrict__ lam = P.lam + ((size_t)b * N + ig) * 8;
    KktGlobal G;
    G.pm1c = pim1[cx];
    G.ll = lam[ju];
    G.lu = lam[4 + ju];
    G.ucur = ui[ju];
    return G;
}
__device__ __forceinline__ KktOperands finish_kkt_operands(const KktGlobal& G, const double* __restrict__ cst, int il, int nl, int c,
                                                           const lds_f64* q_s, const lds_f64* r_s) {
    const int jc = c - NX, ju = jc & 3, cx = c < NX ? c : NX - 1;
    KktOperands K;
    K.pm1c = G.pm1c; K.ll = G.ll; K.lu = G.lu; K.ucur = G.ucur;
    K.lbu = cst[32 + ju];
    K.ubu = cst[36 + ju];
    const lds_f64* gp = jc < 0 ? q_s + il * NX + c : r_s + il * NU + jc;
    K.grad = *gp;
    K.qn = q_s[nl * NX + cx];
    return K;
}
__device__ __forceinline__ KktOperands load_kkt_operands(const DevParams& P, const double* __restrict__ cst, int b, int ig, int il,
                                                         int nl, int c, const double* __restrict__ ui, const lds_f64* q_s,
                                                         const lds_f64* r_s) {
    return finish_kkt_operands(load_kkt_global(P, b, ig, c, ui), cst, il, nl, c, q_s, r_s);
}

// sensitivity column c (3..15) of interval record rec through the 4 stage records.  The next record is requested before
// the RK bookkeeping of the current stage, which covers most of the LDS latency.
__device__ __forceinline__ void sens_column_rec(const lds_f64* rec, const ModelPar& m, double h, int c, double (&acc)[NX]) {
    double ks[NX], ss[NX], kb[4];
    const int jc = c - NX;
    constexpr double ir = 1.0 / kRotor;
    // df/du column jc (model_bcol), all zero for a state column
    kb[0] = jc == 0 ? (-4.0 * 0.707) * ir * m.imx : 0.0;
    kb[1] = jc == 1 ? (4.0 * 0.707) * ir * m.imy : 0.0;
    kb[2] = jc == 2 ? -2.0 * ir * m.imz : 0.0;
    kb[3] = jc == 1 ? (0.167 + 0.167 - 0.175 - 0.175) * ir * m.imn : (jc == 3 ? (0.167 + 0.167 + 0.175 + 0.175) * ir * m.imn : 0.0);
    StageRec R = load_stage_rec(rec);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NX; j++) ss[j] = (j == c) ? 1.0 : 0.0;
    jvp_rec(R, m.imx, m.imy, m.imz, m.imn, kb, ss, ks);
    R = load_stage_rec(rec + kRecStage);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NX; j++) { acc[j] = ((j == c) ? 1.0 : 0.0) + (h / 6.0) * ks[j]; ss[j] = ((j == c) ? 1.0 : 0.0) + 0.5 * h * ks[j]; }
    __builtin_amdgcn_sched_barrier(0);
    jvp_rec(R, m.imx, m.imy, m.imz, m.imn, kb, ss, ks);
    R = load_stage_rec(rec + 2 * kRecStage);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NX; j++) { acc[j] += (h / 3.0) * ks[j]; ss[j] = ((j == c) ? 1.0 : 0.0) + 0.5 * h * ks[j]; }
    __builtin_amdgcn_sched_barrier(0);
    jvp_rec(R, m.imx, m.imy, m.imz, m.imn, kb, ss, ks);
    R = load_stage_rec(rec + 3 * kRecStage);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NX; j++) { acc[j] += (h / 3.0) * ks[j]; ss[j] = ((j == c) ? 1.0 : 0.0) + h * ks[j]; }
    __builtin_amdgcn_sched_barrier(0);
    jvp_rec(R, m.imx, m.imy, m.imz, m.imn, kb, ss, ks);
#pragma unroll
    for (int j = 0; j < NX; j++) acc[j] += (h / 6.0) * ks[j];
}

// (df/dx) * s for a vector with only the velocity entries s[6], s[7], s[8], s[11] non-zero (what the first RK stage of an
// input column produces: df/du has rows 6, 7, 8, 11 only)
__device__ __forceinline__ void jvp_rec_vel(const StageRec& R, double imn, const double (&kb)[4], double s6, double s7,
                                            double s8, double s11, double (&o)[NX]) {
    const double* r = R.v;
    const double sph = r[0], cph = r[1], sth = r[2], cth = r[3], sps = r[4], cps = r[5], icth = r[6];
    const double wp = r[10], wq = r[11];
    const double r00 = cps * cth, r01 = cps * sth * sph - sps * cph, r02 = sps * sph + cps * cph * sth;
    const double r10 = sps * cth, r11 = cps * cph + sph * sth * sps, r12 = sth * sps * cph - cps * sph;
    const double r21 = cth * sph, r22 = cth * cph;
    const double tth = sth * icth;
    o[0] = r00 * s6 + r01 * s7 + r02 * s8;
    o[1] = r10 * s6 + r11 * s7 + r12 * s8;
    o[2] = -sth * s6 + r21 * s7 + r22 * s8;
    o[3] = (cph * tth) * s11;
    o[4] = sph * s11;
    o[5] = (cph * s11) * icth;
    o[6] = r[13] * s6 + kb[0];
    o[7] = r[14] * s7 + kb[1];
    o[8] = r[15] * s8 + kb[2];
    o[9] = ((kIy - kIz) * (wq * s11)) * (1.0 / kIx);
    o[10] = ((kIz - kIx) * (wp * s11)) * (1.0 / kIy);
    o[11] = r[16] * s11 + kb[3];
    (void)imn;
}

// sensitivity column of input jc (0..3).  Su starts at zero, so the first RK stage is df/du itself (no Jacobian-vector
// product) and the second one sees a vector with four non-zero entries: an input column costs ~2.4 of the 4 products of a
// state column.
__device__ __forceinline__ void sens_column_rec_u(const lds_f64* rec, const ModelPar& m, double h, int jc, double (&acc)[NX]) {
    double ks[NX], ss[NX], kb[4];
    constexpr double ir = 1.0 / kRotor;
    kb[0] = jc == 0 ? (-4.0 * 0.707) * ir * m.imx : 0.0;
    kb[1] = jc == 1 ? (4.0 * 0.707) * ir * m.imy : 0.0;
    kb[2] = jc == 2 ? -2.0 * ir * m.imz : 0.0;
    kb[3] = jc == 1 ? (0.167 + 0.167 - 0.175 - 0.175) * ir * m.imn : (jc == 3 ? (0.167 + 0.167 + 0.175 + 0.175) * ir * m.imn : 0.0);
    StageRec R = load_stage_rec(rec + kRecStage);
    __builtin_amdgcn_sched_barrier(0);
    // stage 1: k1 = df/du column (rows 6, 7, 8, 11)
#pragma unroll
    for (int j = 0; j < NX; j++) acc[j] = 0.0;
    acc[6] = (h / 6.0) * kb[0]; acc[7] = (h / 6.0) * kb[1]; acc[8] = (h / 6.0) * kb[2]; acc[11] = (h / 6.0) * kb[3];
    jvp_rec_vel(R, m.imn, kb, 0.5 * h * kb[0], 0.5 * h * kb[1], 0.5 * h * kb[2], 0.5 * h * kb[3], ks);
    R = load_stage_rec(rec + 2 * kRecStage);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NX; j++) { acc[j] += (h / 3.0) * ks[j]; ss[j] = 0.5 * h * ks[j]; }
    __builtin_amdgcn_sched_barrier(0);
    jvp_rec(R, m.imx, m.imy, m.imz, m.imn, kb, ss, ks);
    R = load_stage_rec(rec + 3 * kRecStage);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NX; j++) { acc[j] += (h / 3.0) * ks[j]; ss[j] = h * ks[j]; }
    __builtin_amdgcn_sched_barrier(0);
    jvp_rec(R, m.imx, m.imy, m.imz, m.imn, kb, ss, ks);
#pragma unroll
    for (int j = 0; j < NX; j++) acc[j] += (h / 6.0) * ks[j];
}

// Sensitivity columns that never leave span{e_0, e_1, e_2, e_r}, r = 6 + j a linear-velocity row: the state columns of the
// body velocities (c = 6 + j) and the pure force inputs u0 (row 6) and u2 (row 8).  df/dx has no position columns, column
// 6 + j of df/dx is [R(:, j); 0; d_j e_j; 0] (rotation column + damping diagonal, no Coriolis terms in this model), so every
// RK stage maps the scalar sigma = s[r] to k = [R_st(:, j) sigma; d_st sigma + kb]: ~25 operations per stage instead of the
// ~125 of the general Jacobian-vector product.  acc = the full column (rows other than 0..2 and r are e_c resp. 0).
// out = {S[0][c], S[1][c], S[2][c], S[6+j][c]}; expand_cheap() builds the full column
__device__ __forceinline__ void sens_column_cheap(const lds_f64* rec, double h, int j, bool input, double kbv, double (&out)[4]) {
    // all four stage records first (one LDS wait instead of one per stage): the six trig values and this row's damping entry
    double tr[4][6], dd[4];
#pragma unroll
    for (int st = 0; st < 4; st++) {
        const lds_f64* r = rec + st * kRecStage;
#pragma unroll
        for (int k = 0; k < 6; k++) tr[st][k] = r[k];
        dd[st] = r[13 + j];   // one load with a per-lane address: as a select of three loads the compiler branches on j and waits
                              // for the LDS queue inside each branch (four lgkmcnt(0) stalls per column: 1 k cycles per solve)
    }
    const double s0 = input ? 0.0 : 1.0;   // seed: e_c for a state column, 0 for an input column
    double sig = s0, ar = s0, a0 = 0.0, a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int st = 0; st < 4; st++) {
        const double sph = tr[st][0], cph = tr[st][1], sth = tr[st][2], cth = tr[st][3], sps = tr[st][4], cps = tr[st][5];
        const double r00 = cps * cth, r01 = cps * sth * sph - sps * cph, r02 = sps * sph + cps * cph * sth;
        const double r10 = sps * cth, r11 = cps * cph + sph * sth * sps, r12 = sth * sps * cph - cps * sph;
        const double r21 = cth * sph, r22 = cth * cph;
        const double c0 = (j == 0) ? r00 : ((j == 1) ? r01 : r02);
        const double c1 = (j == 0) ? r10 : ((j == 1) ? r11 : r12);
        const double c2 = (j == 0) ? -sth : ((j == 1) ? r21 : r22);
        const double kr = dd[st] * sig + kbv;
        const double w = (st == 0 || st == 3) ? h / 6.0 : h / 3.0;
        a0 += w * (c0 * sig); a1 += w * (c1 * sig); a2 += w * (c2 * sig);
        ar += w * kr;
        sig = s0 + ((st == 2) ? h : 0.5 * h) * kr;   // input of the next stage
    }
    out[0] = a0; out[1] = a1; out[2] = a2; out[3] = ar;
}
__device__ __forceinline__ void expand_cheap(const double (&v)[4], int j, double (&acc)[NX]) {
#pragma unroll
    for (int k = 0; k < NX; k++) acc[k] = 0.0;
    acc[0] = v[0]; acc[1] = v[1]; acc[2] = v[2];
    acc[6] = (j == 0) ? v[3] : 0.0; acc[7] = (j == 1) ? v[3] : 0.0; acc[8] = (j == 2) ? v[3] : 0.0;
}

// stationarity / input-feasibility part of the NLP KKT residual for column c >= 3 of interval i, with the cost gradients
// already in LDS (q_i, r_i); the dynamics gap and the position columns are handled once per interval by the caller.
// Branch-free: rows that do not apply contribute 0 to the max.
__device__ __forceinline__ void lin_kkt_rows(KktAcc& A, const KktOperands& K, int N, int i, int c, double dotpi, double pc) {
    const bool xcol = c < NX;
    const double sx = K.grad + dotpi - K.pm1c;            // x column, i >= 1
    const double su_ = K.grad + dotpi - K.ll + K.lu;      // u column
    A.upd(xcol ? (i >= 1 ? sx : 0.0) : su_);
    A.upd((xcol && i == N - 1) ? K.qn - pc : 0.0);         // terminal: q_N - pi_{N-1}
    const double sl = K.ucur - K.lbu, su = K.ubu - K.ucur;
    A.upd((!xcol && sl < 0) ? sl : 0.0);
    A.upd((!xcol && su < 0) ? su : 0.0);
    A.upd(xcol ? 0.0 : K.ll * sl);
    A.upd(xcol ? 0.0 : K.lu * su);
}
__device__ __forceinline__ void lin_kkt_col(KktAcc& A, const KktOperands& K, int N, int i, int c, const double (&pil)[NX],
                                            const double (&acc)[NX]) {
    double dotpi = 0.0, pc = 0.0;
#pragma unroll
    for (int j = 0; j < NX; j++) dotpi += acc[j] * pil[j];
#pragma unroll
    for (int j = 3; j < NX; j++) pc = (j == c) ? pil[j] : pc;
    lin_kkt_rows(A, K, N, i, c, dotpi, pc);
}

}  // namespace brov
