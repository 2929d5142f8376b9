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
#include "lin_device.hpp"

namespace brov {

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
    double xn[NX], acc[NX];
    rk4_sens_column(xi, ui, pp, h, c, xn, acc);

    // ---- [A B] row-major, 16 wide: lane c stores S[k][c]; 16 lanes = one 128-byte row ---------------------------
    double* __restrict__ BA = P.BA + (size_t)gi * (NX * 16);
    double* mine = lds + g * (NX * kLdsStride);
    if (active) {
#pragma unroll
        for (int kk = 0; kk < NX; kk++) BA[kk * 16 + c] = acc[kk];
    }
#pragma unroll
    for (int kk = 0; kk < NX; kk++) mine[kk * kLdsStride + c] = acc[kk];

    // ---- NLP KKT residual of the entering iterate, per variable c, and the dynamics gap b_i ----------------------
    {
        double bc;
        const double kkt = lin_kkt_lane(P, cst, b, i, c, xi, ui, xn, acc, bc);
        if (active && c < NX) P.bvec[(size_t)gi * NX + c] = bc;
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
