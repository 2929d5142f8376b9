// lin_device.hpp -- the per-interval linearisation shared by the streaming kernel (lin_kernel.hip) and the fused
// LDS-resident kernel (qp_kernel.hip: rti_fused_kernel).  16 lanes cooperate on one interval; lane c owns column c of
// S = d x+ / d [x;u].
#pragma once
#include "bluerov2_model.hpp"
#include "nmpc_device.hpp"

namespace brov {

__device__ __forceinline__ void kkt_upd(double& kkt, double v) {
    const double a = fabs(v);
    // NaN poisons the norm (oracle/bluerov2_oracle.c UPD macro)
    kkt = (a != a) ? a : ((kkt != kkt) ? kkt : fmax(kkt, a));
}

// ERK4 (4 stages, 1 step: gen/acados_solver_bluerov2.c:633-641) on the state and on sensitivity column c, stage by stage.
// xn = x+ (identical in the 16 lanes of the group), acc = S[:,c].
__device__ __forceinline__ void rk4_sens_column(const double* __restrict__ xi, const double* __restrict__ ui,
                                                const double* __restrict__ pp, double h, int c, double (&xn)[NX],
                                                double (&acc)[NX]) {
    double x[NX], uu[NU];
#pragma unroll
    for (int j = 0; j < NX; j++) x[j] = xi[j];
#pragma unroll
    for (int j = 0; j < NU; j++) uu[j] = ui[j];
    const ModelPar m = make_par(pp);
    const Wrench w = make_wrench(uu);
    const int jc = c - NX;  // >= 0: input column
    double k[NX], ks[NX];
    StagePoint sp;
    double xs[NX], ss[NX];
    // stage 1
    model_f(x, w, m, k, sp);
#pragma unroll
    for (int j = 0; j < NX; j++) ss[j] = (j == c) ? 1.0 : 0.0;
    model_jvp(sp, m, ss, ks);
    if (jc >= 0) model_bcol(m, jc, ks);
#pragma unroll
    for (int j = 0; j < NX; j++) {
        xn[j] = x[j] + (h / 6.0) * k[j];
        acc[j] = ((j == c) ? 1.0 : 0.0) + (h / 6.0) * ks[j];
    }
    // stages 2 and 3
#pragma unroll
    for (int s = 0; s < 2; s++) {
#pragma unroll
        for (int j = 0; j < NX; j++) {
            xs[j] = x[j] + 0.5 * h * k[j];
            ss[j] = ((j == c) ? 1.0 : 0.0) + 0.5 * h * ks[j];
        }
        model_f(xs, w, m, k, sp);
        model_jvp(sp, m, ss, ks);
        if (jc >= 0) model_bcol(m, jc, ks);
#pragma unroll
        for (int j = 0; j < NX; j++) { xn[j] += (h / 3.0) * k[j]; acc[j] += (h / 3.0) * ks[j]; }
    }
    // stage 4
#pragma unroll
    for (int j = 0; j < NX; j++) {
        xs[j] = x[j] + h * k[j];
        ss[j] = ((j == c) ? 1.0 : 0.0) + h * ks[j];
    }
    model_f(xs, w, m, k, sp);
    model_jvp(sp, m, ss, ks);
    if (jc >= 0) model_bcol(m, jc, ks);
#pragma unroll
    for (int j = 0; j < NX; j++) { xn[j] += (h / 6.0) * k[j]; acc[j] += (h / 6.0) * ks[j]; }
}

// Stage points of the 4 RK stages (kept in registers) + x+; used by the column-at-a-time variant below.
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

// The column loop re-derives the ~30 Jacobian entries of a stage from its 13-value StagePoint each time it needs them.
// Left alone, LICM hoists all of them out of the loop (4 stages x 48 entries = 384 registers); making the stage points
// opaque at the top of every iteration keeps the working set at ~120 doubles.
__device__ __forceinline__ void keep_in_loop(StagePoint& s) {
    asm volatile("" : "+v"(s.sph), "+v"(s.cph), "+v"(s.sth), "+v"(s.cth), "+v"(s.sps), "+v"(s.cps), "+v"(s.icth));
    asm volatile("" : "+v"(s.vu), "+v"(s.vv), "+v"(s.vw), "+v"(s.wp), "+v"(s.wq), "+v"(s.wr));
}

// sensitivity column c (3..15) through the 4 stored stage points
__device__ __forceinline__ void sens_column(StagePoint (&sp)[4], const ModelPar& m, double h, int c, double (&acc)[NX]) {
    double ks[NX], ss[NX];
    const int jc = c - NX;
    keep_in_loop(sp[0]); keep_in_loop(sp[1]); keep_in_loop(sp[2]); keep_in_loop(sp[3]);
#pragma unroll
    for (int j = 0; j < NX; j++) ss[j] = (j == c) ? 1.0 : 0.0;
    model_jvp(sp[0], m, ss, ks);
    if (jc >= 0) model_bcol(m, jc, ks);
#pragma unroll
    for (int j = 0; j < NX; j++) { acc[j] = ((j == c) ? 1.0 : 0.0) + (h / 6.0) * ks[j]; ss[j] = ((j == c) ? 1.0 : 0.0) + 0.5 * h * ks[j]; }
    model_jvp(sp[1], m, ss, ks);
    if (jc >= 0) model_bcol(m, jc, ks);
#pragma unroll
    for (int j = 0; j < NX; j++) { acc[j] += (h / 3.0) * ks[j]; ss[j] = ((j == c) ? 1.0 : 0.0) + 0.5 * h * ks[j]; }
    model_jvp(sp[2], m, ss, ks);
    if (jc >= 0) model_bcol(m, jc, ks);
#pragma unroll
    for (int j = 0; j < NX; j++) { acc[j] += (h / 3.0) * ks[j]; ss[j] = ((j == c) ? 1.0 : 0.0) + h * ks[j]; }
    model_jvp(sp[3], m, ss, ks);
    if (jc >= 0) model_bcol(m, jc, ks);
#pragma unroll
    for (int j = 0; j < NX; j++) acc[j] += (h / 6.0) * ks[j];
}

// NLP KKT residual of the entering iterate attributable to variable c of interval i (oracle/bluerov2_oracle.c
// orc_rti_step): dynamics gap, stationarity w.r.t. x_i / u_i (needs pi_i' S[:,c]), input feasibility and
// complementarity, terminal stationarity on the last interval.  Returns b_i[c] (c < 12) through bc.
__device__ __forceinline__ double lin_kkt_lane(const DevParams& P, const double* __restrict__ cst, int b, int i, int c,
                                               const double* __restrict__ xi, const double* __restrict__ ui,
                                               const double (&xn)[NX], const double (&acc)[NX], double& bc) {
    const int N = P.N;
    const int jc = c - NX;
    const double* __restrict__ pil = P.pi + ((size_t)b * N + i) * NX;
    const double* __restrict__ pim1 = P.pi + ((size_t)b * N + (i > 0 ? i - 1 : 0)) * NX;
    const double* __restrict__ yr = P.yref + (size_t)b * P.yref_stride + (size_t)i * NY;
    double kkt = 0.0, dotpi = 0.0;
#pragma unroll
    for (int j = 0; j < NX; j++) dotpi += acc[j] * pil[j];
    const double wc = cst[c];
    bc = 0.0;
    if (jc < 0) {
#pragma unroll
        for (int j = 0; j < NX; j++) bc = (j == c) ? xn[j] : bc;
        bc -= xi[NX + c];
        kkt_upd(kkt, bc);
        if (i >= 1) kkt_upd(kkt, P.Ts * wc * (xi[c] - yr[c]) + dotpi - pim1[c]);
        if (i == N - 1) kkt_upd(kkt, cst[16 + c] * (xi[NX + c] - yr[NY + c]) - pil[c]);  // terminal: q_N - pi_{N-1}
    } else {
        const double* __restrict__ lam = P.lam + ((size_t)b * N + i) * 8;
        const double ucur = ui[jc];
        const double ll = lam[jc], lu = lam[4 + jc];
        kkt_upd(kkt, P.Ts * wc * (ucur - yr[c]) + dotpi - ll + lu);
        const double sl = ucur - cst[32 + jc], su = cst[36 + jc] - ucur;
        if (sl < 0) kkt_upd(kkt, sl);
        if (su < 0) kkt_upd(kkt, su);
        kkt_upd(kkt, ll * sl);
        kkt_upd(kkt, lu * su);
    }
    return kkt;
}

}  // namespace brov
