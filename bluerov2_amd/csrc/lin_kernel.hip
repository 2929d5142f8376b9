// lin_kernel.hip -- RTI preparation phase on gfx950: ERK4 + forward sensitivities of every shooting interval of every
// OCP instance.
//
// Replaces, per interval, the 4 calls of bluerov2_expl_vde_forw that acados' ERK integrator makes
// (/root/reference/bluerov2_dobmpc/scripts/c_generated_code/acados_solver_bluerov2.c:310-313,633-641) and the
// residual/KKT bookkeeping of ocp_nlp_sqp_rti's preparation.
//
// Mapping: 16 lanes per (instance, interval); lane c owns column c of the sensitivity S = d x+ / d [x;u] (12 x 16), so a
// wavefront integrates 4 intervals and a 256-thread block 16.  The consumers (qp_kernel.hip) read [A B] as row-major
// 16-wide tiles, and with this mapping lane c holds S[k][c] for k = 0..11: every store instruction writes 128 contiguous
// bytes per interval (round 1's first version, one lane per interval, scattered 8-byte stores over 3.6 KB strides and
// ran 6x slower, store-bound).  The transposed tile [A B]^T is produced through a padded LDS transpose so that it is
// written row-contiguously as well.  The 4 RK stage points are integrated redundantly by the 16 lanes of a group (the
// state trajectory is 12 values; sharing it would cost more cross-lane traffic than the ~25 % of redundant VALU work it
// saves); position columns need no special case: df/dx has no position dependence, so their stage derivatives are zero
// and S[:,c] = e_c falls out of the same code.
//
// Bound: FP64 VALU (about 1.4 k FP64 instructions per lane, 12 sincos) -- 3.7 KB written per interval is noise.
#include "bluerov2_model.hpp"
#include "nmpc_device.hpp"

namespace brov {

__device__ __forceinline__ void kkt_upd(double& kkt, double v) {
    const double a = fabs(v);
    // NaN poisons the norm (oracle/bluerov2_oracle.c UPD macro)
    kkt = (a != a) ? a : ((kkt != kkt) ? kkt : fmax(kkt, a));
}

constexpr int kGroupsPerBlock = 16;
constexpr int kLdsStride = 17;  // doubles per row of the per-group 12 x 16 staging tile (+1 pad: conflict-free transpose)

#ifndef BROV_LIN_WAVES
#define BROV_LIN_WAVES 1
#endif
__global__ __launch_bounds__(256, BROV_LIN_WAVES) void lin_kernel(DevParams P) {
    __shared__ double lds[kGroupsPerBlock * NX * kLdsStride];
    const int c = threadIdx.x & 15;   // column of S owned by this lane
    const int g = threadIdx.x >> 4;   // group within the block
    const int N = P.N;
    const int total = P.B * N;
    int gi = blockIdx.x * kGroupsPerBlock + g;
    const bool active = gi < total;
    if (!active) gi = total - 1;  // keep the lane alive for the block-wide barrier; it stores nothing
    const int b = gi / N, i = gi - b * N;
    const double h = P.Ts;
    const double* __restrict__ cst = P.cst;

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
    const int jc = c - NX;  // >= 0: input column

    // ---- ERK4 on the state and on this lane's sensitivity column, stage by stage ------------------------------
    double xn[NX], acc[NX], k[NX], ks[NX];
    {
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

    // ---- [A B] row-major, 16 wide: lane c stores S[k][c]; 16 lanes = one 128-byte row ---------------------------
    double* __restrict__ BA = P.BA + (size_t)gi * (NX * 16);
    double* mine = lds + g * (NX * kLdsStride);
    if (active) {
#pragma unroll
        for (int kk = 0; kk < NX; kk++) BA[kk * 16 + c] = acc[kk];
    }
#pragma unroll
    for (int kk = 0; kk < NX; kk++) mine[kk * kLdsStride + c] = acc[kk];

    // ---- NLP KKT residual of the entering iterate (oracle/bluerov2_oracle.c orc_rti_step), per variable c ------
    double kkt = 0.0;
    {
        const double* __restrict__ pil = P.pi + ((size_t)b * N + i) * NX;
        const double* __restrict__ pim1 = P.pi + ((size_t)b * N + (i > 0 ? i - 1 : 0)) * NX;
        const double* __restrict__ yr = P.yref + (size_t)b * P.yref_stride + (size_t)i * NY;
        double dotpi = 0.0;
#pragma unroll
        for (int j = 0; j < NX; j++) dotpi += acc[j] * pil[j];
        const double wc = cst[c];
        if (jc < 0) {
            // dynamics residual b_i = phi(x_i,u_i) - x_{i+1}: lane c owns component c
            double bc = 0.0;
#pragma unroll
            for (int j = 0; j < NX; j++) bc = (j == c) ? xn[j] : bc;
            bc -= xi[NX + c];
            if (active) P.bvec[(size_t)gi * NX + c] = bc;
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
        // max over the 16 lanes of the group, NaN-poisoning
        bool isnan_ = kkt != kkt;
        double kk = isnan_ ? 0.0 : kkt;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            kk = fmax(kk, __shfl_xor(kk, o, 16));
            isnan_ = isnan_ || (__shfl_xor((int)isnan_, o, 16) != 0);
        }
        if (active && c == 0) P.kktp[gi] = isnan_ ? __builtin_nan("") : kk;
    }

    // ---- [A B]^T tile (16 x 16, zero-padded columns 12..15) through the LDS transpose ---------------------------
    __syncthreads();
    if (active) {
        double* __restrict__ BAt = P.BAt + (size_t)gi * 256;
#pragma unroll
        for (int cc = 0; cc < 16; cc++) BAt[cc * 16 + c] = (c < NX) ? mine[(c < NX ? c : 0) * kLdsStride + cc] : 0.0;
    }
}

void launch_linearise(const DevParams& P, hipStream_t st) {
    const int total = P.B * P.N;
    hipLaunchKernelGGL(lin_kernel, dim3((total + kGroupsPerBlock - 1) / kGroupsPerBlock), dim3(256), 0, st, P);
}

}  // namespace brov
