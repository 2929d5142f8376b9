// lin_kernel.hip -- RTI preparation phase on gfx950: ERK4 + forward sensitivities of every shooting interval of every
// OCP instance, one LANE per (instance, interval).
//
// Replaces, per interval, the 4 calls of bluerov2_expl_vde_forw that acados' ERK integrator makes
// (/root/reference/bluerov2_dobmpc/scripts/c_generated_code/acados_solver_bluerov2.c:310-313,633-641) and the
// residual/KKT bookkeeping of ocp_nlp_sqp_rti's preparation.  Work is embarrassingly parallel over B*N intervals, so
// the mapping is one lane each (no cross-lane traffic, every lane busy).  The RK4 stage points and their trig values
// are computed once (x-trajectory pass) and kept in registers; the 13 non-trivial sensitivity columns (9 state
// columns -- position columns are exactly e_c because f does not depend on position -- and 4 input columns) are then
// propagated one at a time through the 4 stages with the sparse Jacobian-vector product of bluerov2_model.hpp.
//
// Bound: FP64 VALU (about 8 kflop + 12 sincos per interval vs 3.7 KB written) -- not HBM.
#include "bluerov2_model.hpp"
#include "nmpc_device.hpp"

namespace brov {

// The column loop re-derives the ~30 Jacobian entries of a stage from its 13-value StagePoint each time it needs
// them.  Left alone, LICM hoists all of them out of the loop (4 stages x 48 entries) and the kernel spills ~260 VGPRs;
// making the stage points opaque at the top of every iteration keeps the working set at ~120 doubles.
__device__ __forceinline__ void keep_in_loop(StagePoint& s) {
    asm volatile("" : "+v"(s.sph), "+v"(s.cph), "+v"(s.sth), "+v"(s.cth), "+v"(s.sps), "+v"(s.cps), "+v"(s.icth));
    asm volatile("" : "+v"(s.vu), "+v"(s.vv), "+v"(s.vw), "+v"(s.wp), "+v"(s.wq), "+v"(s.wr));
}

__device__ __forceinline__ void kkt_upd(double& kkt, double v) {
    const double a = fabs(v);
    // NaN poisons the norm (oracle/bluerov2_oracle.c UPD macro)
    kkt = (a != a) ? a : ((kkt != kkt) ? kkt : fmax(kkt, a));
}

__global__ __launch_bounds__(256, 2) void lin_kernel(DevParams P) {
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const int N = P.N;
    if (tid >= P.B * N) return;
    const int b = tid / N, i = tid - b * N;
    const double h = P.Ts;

    const double* __restrict__ xi = P.x + ((size_t)b * (N + 1) + i) * NX;
    const double* __restrict__ ui = P.u + ((size_t)b * N + i) * NU;
    const double* __restrict__ pp = P.par + ((size_t)b * (N + 1) + i) * NP;
    double x[NX], uu[NU];
#pragma unroll
    for (int j = 0; j < NX; j++) x[j] = xi[j];
#pragma unroll
    for (int j = 0; j < NU; j++) uu[j] = ui[j];
    const ModelPar m = make_par(pp);
    const Wrench w = make_wrench(uu);

    // ---- pass 1: x-trajectory through the 4 RK stages, keep the stage points --------------------------------
    StagePoint sp[4];
    double xn[NX];
    {
        double k[NX], xs[NX];
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

    // ---- dynamics residual b_i = phi(x_i,u_i) - x_{i+1} ------------------------------------------------------
    double kkt = 0.0;
    {
        const double* __restrict__ xnext = xi + NX;
        double* __restrict__ bv = P.bvec + ((size_t)b * N + i) * NX;
#pragma unroll
        for (int j = 0; j < NX; j++) {
            const double r = xn[j] - xnext[j];
            bv[j] = r;
            kkt_upd(kkt, r);
        }
    }

    // multipliers of the entering iterate, for the NLP stationarity residual (re-read per column: L1-resident)
    const double* __restrict__ pil = P.pi + ((size_t)b * N + i) * NX;
    const double* __restrict__ yr = P.yref + (size_t)b * P.yref_stride + (size_t)i * NY;
    const double* __restrict__ lam = P.lam + ((size_t)b * N + i) * 8;
    const double* __restrict__ pim1 = P.pi + ((size_t)b * N + (i > 0 ? i - 1 : 0)) * NX;

    double* __restrict__ BA = P.BA + ((size_t)b * N + i) * (NX * 16);
    double* __restrict__ BAt = P.BAt + ((size_t)b * N + i) * 256;

    // ---- trivial position columns: S[:,c] = e_c ---------------------------------------------------------------
#pragma unroll
    for (int c = 0; c < 3; c++) {
#pragma unroll
        for (int k = 0; k < NX; k++) BA[k * 16 + c] = (k == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < 16; k++) BAt[c * 16 + k] = (k == c) ? 1.0 : 0.0;
        if (i >= 1) kkt_upd(kkt, P.Ts * P.W[c] * (xi[c] - yr[c]) + pil[c] - pim1[c]);
    }

    // ---- pass 2: one sensitivity column at a time through the 4 stages ---------------------------------------
#pragma unroll 1
    for (int c = 3; c < 16; c++) {
        double s0[NX], acc[NX], ks[NX], ss[NX];
        keep_in_loop(sp[0]); keep_in_loop(sp[1]); keep_in_loop(sp[2]); keep_in_loop(sp[3]);
#pragma unroll
        for (int j = 0; j < NX; j++) { s0[j] = (j == c) ? 1.0 : 0.0; }
        const int jc = c - NX;  // >= 0 for input columns
        // stage 1: S = S0
        model_jvp(sp[0], m, s0, ks);
        if (jc >= 0) model_bcol(m, jc, ks);
#pragma unroll
        for (int j = 0; j < NX; j++) { acc[j] = s0[j] + (h / 6.0) * ks[j]; ss[j] = s0[j] + 0.5 * h * ks[j]; }
        model_jvp(sp[1], m, ss, ks);
        if (jc >= 0) model_bcol(m, jc, ks);
#pragma unroll
        for (int j = 0; j < NX; j++) { acc[j] += (h / 3.0) * ks[j]; ss[j] = s0[j] + 0.5 * h * ks[j]; }
        model_jvp(sp[2], m, ss, ks);
        if (jc >= 0) model_bcol(m, jc, ks);
#pragma unroll
        for (int j = 0; j < NX; j++) { acc[j] += (h / 3.0) * ks[j]; ss[j] = s0[j] + h * ks[j]; }
        model_jvp(sp[3], m, ss, ks);
        if (jc >= 0) model_bcol(m, jc, ks);
        double dotpi = 0.0;
#pragma unroll
        for (int j = 0; j < NX; j++) {
            acc[j] += (h / 6.0) * ks[j];
            BA[j * 16 + c] = acc[j];
            BAt[c * 16 + j] = acc[j];
            dotpi += acc[j] * pil[j];
        }
#pragma unroll
        for (int j = NX; j < 16; j++) BAt[c * 16 + j] = 0.0;
        // NLP stationarity of the entering iterate (oracle/bluerov2_oracle.c orc_rti_step)
        if (jc < 0) {
            if (i >= 1) kkt_upd(kkt, P.Ts * P.W[c] * (xi[c] - yr[c]) + dotpi - pim1[c]);
        } else {
            const double ucur = P.u[((size_t)b * N + i) * NU + jc];
            const double ll = lam[jc], lu = lam[4 + jc];
            kkt_upd(kkt, P.Ts * P.W[c] * (ucur - yr[c]) + dotpi - ll + lu);
            const double sl = ucur - P.lbu[jc], su = P.ubu[jc] - ucur;
            if (sl < 0) kkt_upd(kkt, sl);
            if (su < 0) kkt_upd(kkt, su);
            kkt_upd(kkt, ll * sl);
            kkt_upd(kkt, lu * su);
        }
    }
    // terminal stationarity q_N - pi_{N-1}, attached to the last interval
    if (i == N - 1) {
        const double* __restrict__ xN = xi + NX;
        const double* __restrict__ yN = yr + NY;
#pragma unroll
        for (int j = 0; j < NX; j++) kkt_upd(kkt, P.We[j] * (xN[j] - yN[j]) - pil[j]);
    }
    P.kktp[(size_t)b * N + i] = kkt;
}

void launch_linearise(const DevParams& P, hipStream_t st) {
    const int total = P.B * P.N;
    const int block = 256;
    hipLaunchKernelGGL(lin_kernel, dim3((total + block - 1) / block), dim3(block), 0, st, P);
}

}  // namespace brov
