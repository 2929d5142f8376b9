// qp/sweeps.hpp -- the Riccati sweeps every kernel family shares: backward (factor / solve-only; step-0, in-loop, robust and accumulating forms of ONE stage body), forward, roll-out, adjoint -- each split into an initialisation and a "stages of the resident window" part.
// Part of ONE translation unit: qp_kernel.hip includes these headers in layer order (tiles -> sweeps -> window -> sched -> qp_body ->
// lin_phase -> fused -> windowed -> pit) and instantiates the kernels between them; see the file map at the head of qp_kernel.hip.
#pragma once

namespace brov {

// Fused path: LDS = 1 is the one-wave-per-SIMD kernel (look-ahead of two stages), LDS = 2 the two-waves-per-SIMD kernel for
// short horizons (one stage: the SIMD's other wave covers the rest, and the third register slot would be spilled).
template <int LDS> constexpr int kLdsDist = LDS == 2 ? 1 : 2;

struct BwdIn {
    d4 ba;          // [A B] tile, rows 0..11
    d4 bv;          // FACTOR: b_i;  else: Pb_i = P_{i+1} b_i   (row-replicated)
    double xv[3], yv[3];  // x_i[row], yref_i[row]
    double rtv, gm;       // rtilde_i[rg], Gamma_i[rg]
    double ks, mt;        // stored factors (only !FACTOR)
};

// Round 6, factor sweeps of the LDS-resident kernels: the second operand image [b_i | A_i(:,1:) B_i] is read as such -- lanes of column 0
// point at b_i, the others where the [A B] image points -- instead of b_i row-replicated and a select per register behind it
struct Ba1Off { int off[3], str; };
template <bool FACTOR, int LDS, bool STEP0 = false, class IT = Inst>
__device__ __forceinline__ BwdIn load_bwd(const IT& I, int i, const double* gam, const double* rt, const Ba1Off* b1 = nullptr) {
    BwdIn s;
    s.ba = get_ba<LDS>(I, i);
    const int ig = I.i0 + i;   // HBM-resident operands are indexed by the global stage
    if constexpr (FACTOR && LDS != 0) {
        if (b1) {
            const lds_f64* t = I.lds_ba + lmul(i, b1->str);
            s.bv = d4{t[b1->off[0]], t[b1->off[1]], t[b1->off[2]], 0.0};
        } else {
            s.bv = get_bv<LDS>(I, i);
        }
    } else
    s.bv = FACTOR ? get_bv<LDS>(I, i) : load_vec12(I.Pb + (size_t)ig * 12, I.rg);
    if constexpr (LDS) {
#pragma unroll
        for (int r = 0; r < 3; r++) { s.xv[r] = I.lds_q[i * 12 + I.rg + 4 * r]; s.yv[r] = 0.0; }
    } else {
        const double* xi = I.x + (size_t)i * 12;
        const double* yi = I.yref + (size_t)i * 16;
#pragma unroll
        for (int r = 0; r < 3; r++) { s.xv[r] = xi[I.rg + 4 * r]; s.yv[r] = yi[I.rg + 4 * r]; }
    }
    if constexpr (STEP0) {  // Gamma = 0, rhs = r_i: no IPM arrays involved
        if constexpr (LDS) s.rtv = I.lds_r[i * 4 + I.rg];
        else s.rtv = I.u[i * 4 + I.rg] - I.yref[(size_t)i * 16 + 12 + I.rg];   // streaming kernel: weighted in the stage body
        s.gm = 0.0;
    } else {
        s.rtv = rt[ig * 4 + I.rg];
        s.gm = FACTOR ? gam[ig * 4 + I.rg] : 0.0;
    }
    s.ks = FACTOR ? 0.0 : I.Ks[(size_t)ig * 64 + I.lane];
    s.mt = 0.0;   // M rides in columns 12..15 of the stored gain operand
    if constexpr (IT::kGrid && LDS != 0 && FACTOR) {
        // general grid on the LDS-resident kernels: the stage's scaled weights ts_i * W (stage 0: W_0) for this lane's four rows,
        // requested with the stage's other operands (the two fields are unused in LDS mode otherwise)
        const double* ws = I.wst + (size_t)ig * 16 + I.rg;
        s.yv[0] = ws[0]; s.yv[1] = ws[4]; s.yv[2] = ws[8]; s.mt = ws[12];
    }
    return s;
}

// backward Riccati sweep.  FACTOR = true: factorise with the current Gamma (ipm[GAM]) and solve for rhs ipm[RT];
// FACTOR = false: reuse the stored factors (Ks, Mt, Pb) and solve for a new rhs.  Returns false if a pivot block is
// not positive definite.  The sweep is split into bwd_init (terminal cost -> P, p) and bwd_chunk (the stages of the resident
// window, state carried in registers) so that the windowed kernel can run it window by window.
// illc: a pivot block of the sweep was ill-conditioned (see kPivotRho); wave-uniform like ok
struct PitAcc { d4 Psi, G; };
struct BwdState { d4 P, pv; bool ok; bool illc = false; PitAcc acc; };   // acc: bwd_chunk<..., ACC = true> only (see PitAcc)
// Parallel-in-time step-0 solve (rti_pit_kernel): what a segment's factor sweep accumulates next to its Riccati recursion, so that the
// segment can be condensed to its two ends -- Psi = Phi' (Phi: closed-loop transition from the current stage to the segment end, 12 x 12
// in columns 0..11), G rows 0..11 = sum Z M Z' (Z = Phi_{i+1} B_i: how the segment's end state answers to a costate at that end), G row 12 =
// c' (c: the forced response of the segment end).  Filled by bwd_chunk<..., ACC = true> (struct PitAcc, a member of BwdState).
// The 4x4 pivot block Huu is inverted EXPLICITLY by 2x2 block elimination (two reciprocals on the serial chain of every stage).  That is
// as accurate as a Cholesky solve while the block is well conditioned after diagonal scaling, and loses a factor cond(Huu) against
// it otherwise (round 4, scripts/dev/riccati_pivot_variants.py: on QPs whose condensed Hessian has cond 1e11..1e13 the explicit
// inverse leaves u 1e-2 off, the Cholesky form 1e-7).  Well conditioned is the rule: the relative pivots rho of the elimination --
// det E / (a00 a11), s00 / a22, s11 / a33, det Sc / (s00 s11) -- are 0.94..1 on every instance of the standard workloads and
// 1e-6..1e-4 on the ill-conditioned ones (iterates of a diverging full-step SQP).  So the fast sweep only WATCHES them (four
// compares per stage, off the chain), and an instance with a relative pivot below kPivotRho repeats the sweep -- and runs all its
// later ones -- in the ROBUST form: Cholesky factor L of Huu (four reciprocal square roots in sequence), its triangular inverse,
//     Y = L^-1 Hu,   S = H - Y'Y,   K = -L^-T Y,   kff = -L^-T (L^-1 gu)        (+2 MFMAs per stage)
// which is the oracle's algebra (chol4 / chol4_solve) in tile form.
constexpr double kPivotRho = 1.0 / 64.0;

template <bool FACTOR, int LDS>
__device__ __forceinline__ void bwd_init(const Inst& I, BwdState& S) {
    const int rg = I.rg, cl = I.cl, N = I.N;
    d4 P = {0, 0, 0, 0}, pv;
    {
        const double* xN = I.x + (size_t)N * 12;
        const double* yN = I.yref + (size_t)N * 16;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int row = rg + 4 * r;
            if (FACTOR) P[r] = (row == cl) ? I.Wer[r] : 0.0;
            if constexpr (LDS) pv[r] = I.lds_q[N * 12 + row];
            else pv[r] = I.Wer[r] * (xN[row] - yN[row]);
        }
        pv[3] = 0.0;
        if (FACTOR) {   // the factor sweep keeps the gradient in column 0 only
#pragma unroll
            for (int r = 0; r < 3; r++) pv[r] = (cl == 0) ? pv[r] : 0.0;
        }
    }
    S.P = P; S.pv = pv; S.ok = true;
}

// Solve-only backward sweep (the corrector solve of an interior-point iteration: stored factors, new right-hand side) of the
// LDS-resident kernels, on the VALU.  It is a pure vector recursion,
//     l = P b + p,   g = [A B]' l + [q; rtilde],   kff = -M g_u,   p <- g_x + K' g_u,
// which round 2 ran as four MFMA tile products per stage (of which 15 of 16 columns are wasted).  Here lane k of ONE 16-lane row
// owns element k of the 16-vector g (12 state rows, 4 input rows) and of p; a product takes the element it needs out of the
// lane that holds it by DPP row broadcast (fmac_bc, see fwd_chunk): 12 fmacs for [A B]' l, 4 for the gain / M column k of the
// stored tile (element (m, k) = K[m][k] for k < 12, M[m][k-12] above: the same address for every lane).  16 lanes = a quarter
// of the LDS clocks of a full-wave read.
struct SolveV { double m[12], ks[4], pb, q, rt; };
template <int LDS>
__device__ __forceinline__ void bwd_solve_v(const Inst& I, BwdState& S) {
    static_assert(LDS != 0, "LDS-resident kernels only");
    const int rg = I.rg, cl = I.cl, N = I.N;
    const int k = I.lane & 15;
    const bool rowx = k < NX, ecol = k < 3;
    const int oc = k >= 3 ? k - 3 : 0, kx = rowx ? k : NX - 1, ku = k & 3;
    const double* rt = I.ipm + (size_t)IPM_RT * I.nv;
    // p of the stage after this window: row-replicated -> lane k (through the transposition scratch; one wave, LDS in order)
    store_vec12_lds(I.lds_tr, S.pv, rg, cl);
    double pcur = I.lds_tr[kx];
    if (I.lane < 16)
    pipelined<(LDS == 3 ? 3 : kLdsDist<LDS>), SolveV>(N, [&](int kk) {   // windowed kernel: the stored factors come out of L2 / HBM
        const int i = N - 1 - kk, ig = I.i0 + i;
        SolveV s;
        const lds_f64* col = I.lds_ba + i * kBaStage + oc;   // column k of [A_i B_i] (columns 0..2 are e_k: loaded, never used)
#pragma unroll
        for (int r = 0; r < 12; r++) s.m[r] = col[r * kBaStride];
        const double* kt = I.Ks + (size_t)ig * 64 + k;        // column k of the stored gain | M tile
#pragma unroll
        for (int t = 0; t < 4; t++) s.ks[t] = kt[16 * t];
        s.pb = I.Pb[(size_t)ig * 12 + kx];
        s.q = I.lds_q[i * 12 + kx];      // both requested by every lane and selected in the body: a load under a divergent
        s.rt = rt[ig * 4 + ku];          // branch is waited for where it is issued
        return s; },
                                     [&](int kk, const SolveV& in) {
        const int i = N - 1 - kk;
        const double l = in.pb + pcur;                        // lanes 12..15: a finite don't-care value, never broadcast
        double d0 = 0.0, d1 = 0.0, d2 = 0.0, d3 = 0.0;
        fmac_bc12(d0, d1, d2, d3, l, in.m[0], in.m[1], in.m[2], in.m[3], in.m[4], in.m[5], in.m[6], in.m[7], in.m[8], in.m[9], in.m[10], in.m[11]);
        const double g = (ecol ? l : (d0 + d1) + (d2 + d3)) + (rowx ? in.q : in.rt);   // columns 0..2 of [A B] are e_k
        double t0 = 0.0, t1 = 0.0;                            // column k of (gain | M) against g_u = lanes 12..15 of g
        fmac_bc4(t0, t1, g, in.ks[0], in.ks[1], in.ks[2], in.ks[3]);
        const double t = t0 + t1;
        lds_f64* kp = rowx ? I.lds_tr + 16 : I.lds_kff + i * 4 + ku;   // rows 12..15: M g_u -> kff = -M g_u; the others park
        *kp = -t;
        pcur = g + t;
    });
    // hand p of this window's first stage on, row-replicated
    lds_f64* tp = (rowx && I.lane < 16) ? I.lds_tr + k : I.lds_tr + 16;
    *tp = pcur;
    const lds_f64* tl = I.lds_tr + rg;
    S.pv = d4{tl[0], tl[4], tl[8], 0.0};
}

// hi / lo: the sweep runs over the stages hi-1 .. lo of the resident block (default: all I.N of them); explicit arguments, not fields of
// Inst -- a horizon that changes under the compiler's eyes costs every sweep of the kernel its loop-invariant addressing
template <bool FACTOR, int LDS, bool STORE_IPM = true, bool STEP0 = false, bool ROBUST = false, class IT = Inst, bool ACC = false>
__device__ __forceinline__ void bwd_chunk(const IT& I, BwdState& S, int hi = -1, int lo = 0) {
    PitAcc* const acc = &S.acc;
    static_assert(!ACC || (FACTOR && LDS == 3 && !ROBUST && !STORE_IPM), "ACC: the factor sweeps of rti_pit_kernel");
    if constexpr (!FACTOR && LDS != 0) {
        bwd_solve_v<LDS>(I, S);
        return;
    }
    const int lane = I.lane, rg = I.rg, cl = I.cl, N = hi < 0 ? I.N : hi;
    const double* gam = I.ipm + (size_t)IPM_GAM * I.nv;
    const double* rt = I.ipm + (size_t)IPM_RT * I.nv;
    BwdIn nx;
    d4& P = S.P;
    d4& pv = S.pv;
    bool& ok = S.ok;
    bool& illc = S.illc;
    const d4 z4 = {0, 0, 0, 0};
    const unsigned mk_col0 = cl == 0 ? ~0u : 0u;
    constexpr bool kMaskPvAtUse = (LDS == 1 || LDS == 2);
    const double col0f = cl == 0 ? 1.0 : 0.0;
    d4 diagm;  // stage cost diag(Ts*Wx, Ts*Wu) in tile layout
#pragma unroll
    for (int r = 0; r < 3; r++) diagm[r] = (rg + 4 * r == cl) ? I.Ts * I.Wr[r] : 0.0;
    diagm[3] = (12 + rg == cl) ? I.Ts * I.Wr[3] : 0.0;
    // Round 6.  (a) LDS-resident kernels: the stage's cost diagonal and the cost gradient enter the H product as its C operand
    // (dgz + [column 0] qr) instead of eight additions behind it; lane 0 holds H[0][0], which is rebuilt from P below, and g_0 -- its
    // diagonal entry stays out of the operand (dgz).  (b) -M = -Huu^-1 as the operand tile, element (rg, cl & 3) per lane, WITHOUT
    // forming the ten elements of M in every lane and selecting one of them (20 operations + a 22-instruction select tree): with
    // Huu = [E F; F' G], X = E^-1 F, M22 = (G - F'X)^-1 and Z = [-X; I] (4 x 2),
    //     M = [E^-1 0; 0 0] + Z M22 Z',      -M[R][C] = -E0[R][C] + (-Z[R][:]) (M22 Z[C][:]')
    // and row R / row C of Z and the entry of E^-1 are sums of the wave-uniform X, E^-1 entries against per-lane 0 / +-1 indicators
    // (loop invariant): 17 multiply-adds per lane, no select, and the part that does not involve M22 runs under the second
    // reciprocal's latency chain.  The negative sign makes T = -M Hu the gain operand itself and M gu the feed-forward term.
    // Only the one-wave fused kernels (LDS = 1) have the registers for all of it.  Measured in the ISA of round 6: with (b) the two-wave kernel (LDS = 2) and the
    // accumulating sweeps of the parallel-in-time kernel go into scratch with it (36 .. 164 bytes per lane), and in the large-batch windowed kernel
    // the loop-invariant operands -- or P and p themselves -- end up in accumulation registers and are moved in every stage (38 .. 71 moves per
    // stage against the 30 instructions saved).  The windowed family also needs its factor sweeps with and without the condensing accumulators
    // to agree bit for bit (the split launches, tests/test_gpu_edge.py), so it stays on one form as a whole.
    constexpr bool kR6 = (LDS == 1 || LDS == 2) && !ROBUST;
    constexpr bool kR6Z = kR6 && LDS == 1;   // (the two-wave kernel, 256 registers: everything but the per-lane constants of (b) -- it keeps the select tree)
    const int pR = rg, pC = cl & 3;
    const double sR0 = pR == 0 ? 1.0 : 0.0, sR1 = pR == 1 ? 1.0 : 0.0, nR2 = pR == 2 ? -1.0 : 0.0, nR3 = pR == 3 ? -1.0 : 0.0;
    const double nC0 = pC == 0 ? -1.0 : 0.0, nC1 = pC == 1 ? -1.0 : 0.0, cC2 = pC == 2 ? 1.0 : 0.0, cC3 = pC == 3 ? 1.0 : 0.0;
    const double nE00 = (pR == 0 && pC == 0) ? -1.0 : 0.0, nE01 = (pR + pC == 1) ? -1.0 : 0.0, nE11 = (pR == 1 && pC == 1) ? -1.0 : 0.0;
    d4 dgz = diagm;
    dgz[0] = lane == 0 ? 0.0 : diagm[0];
    // parked lanes of the per-stage LDS stores write their value to a slot nobody reads (lds_tr[16]) with stride 0
    lds_f64* const kt_st0 = (LDS != 0 && cl < NX) ? I.lds_kt + rg * NX + cl : I.lds_tr + 16;
    const int kt_ststr = (cl < NX) ? kKtStage : 0;
    // (kKffT, below: the feed-forward term comes out of column 12 of the gain product)
#ifndef BROV_EXP_NO_KFF_IN_T
    constexpr bool kKffT = kR6;   // (both fused families: ks carries kff in column 12 with the right sign in either)
#else
    constexpr bool kKffT = false;
#endif
    constexpr int kKffCol = kKffT ? 12 : 0;
    lds_f64* const kf_st0 = (LDS != 0 && cl == kKffCol) ? I.lds_kff + rg : I.lds_tr + 16;
    const int kf_ststr = (cl == kKffCol) ? 4 : 0;
    lds_f64* ktp = kt_st0 + lmul(N - 1, kt_ststr);   // kR6: running store addresses (the stages are visited in order N-1 .. lo)
    lds_f64* kfp = kf_st0 + lmul(N - 1, kf_ststr);
    unsigned long long illm = 0;   // kR6: the watch accumulates in a lane mask (scalar registers)
    Ba1Off b1o;
    if constexpr (LDS != 0) {
        const int bvrel = (int)(I.lds_bv - I.lds_ba);
#pragma unroll
        for (int r = 0; r < 3; r++) b1o.off[r] = cl == 0 ? bvrel + rg + 4 * r : I.ba_off[r];
        b1o.str = cl == 0 ? NX : I.ba_str;
    }
    const Ba1Off* const b1 = kR6 ? &b1o : nullptr;
    auto stage = [&](int i, const BwdIn& in, auto&& mid) __attribute__((always_inline)) {
        // cost gradient [q_i ; rtilde_i], row-replicated
        d4 qr;
#pragma unroll
        for (int r = 0; r < 3; r++) qr[r] = LDS ? in.xv[r] : I.Ts * I.Wr[r] * (in.xv[r] - in.yv[r]);
        qr[3] = (!LDS && STEP0) ? I.Ts * I.Wr[3] * in.rtv : in.rtv;
        d4 dg = diagm;   // stage cost diagonal: loop-invariant ...
        if constexpr (IT::kGrid && LDS != 0 && FACTOR) {   // ... except on a general grid
            dg[0] = (rg == cl) ? in.yv[0] : 0.0; dg[1] = (rg + 4 == cl) ? in.yv[1] : 0.0; dg[2] = (rg + 8 == cl) ? in.yv[2] : 0.0;
            dg[3] = (12 + rg == cl) ? in.mt : 0.0;
        }
        if constexpr (!LDS) {
            // ... except on the streaming kernel's general grid (per-stage time steps / a separate stage-0 weight): the stage's scaled
            // weights come from DevParams::wst, loaded here -- a wave-uniform branch, taken only by solvers that use the feature
            if (I.wst) {
                const double* ws = I.wst + (size_t)i * 16 + rg;
                const double w0 = ws[0], w1 = ws[4], w2 = ws[8], w3 = ws[12];
                qr[0] = w0 * (in.xv[0] - in.yv[0]); qr[1] = w1 * (in.xv[1] - in.yv[1]); qr[2] = w2 * (in.xv[2] - in.yv[2]);
                if (STEP0) qr[3] = w3 * in.rtv;
                dg[0] = (rg == cl) ? w0 : 0.0; dg[1] = (rg + 4 == cl) ? w1 : 0.0; dg[2] = (rg + 8 == cl) ? w2 : 0.0;
                dg[3] = (12 + rg == cl) ? w3 : 0.0;
            }
        }
        if (FACTOR) {
            // One wave's FP64 MFMAs and VALU work do not overlap (scripts/dev/mfma_valu_overlap.hip): a stage costs 64 cycles
            // per MFMA whatever it computes, so the gradient recursion gets no MFMAs of its own -- it rides in column 0 of the
            // two matrix products.  Column 0 of [A B] is e_0 (position x):
            //   P [b | A(:,1:) B]        -> column 0 = P b              (the true column 0, P e_0, is column 0 of P: not needed)
            //   [A B]' [P b + p | ...]   -> column 0 = [A B]'(P b + p)  = g - [q; r]; the true column 0 of H is the transpose of
            //                               its row 0, which this product delivers intact (row 0 of the result = row 0 of the
            //                               right operand, because column 0 of [A B] is e_0); H[0][0] = P[0][0].
            // From here on the gradient recursion (P b, g, p) lives in column 0 of its tiles (lanes cl == 0); the other columns
            // of those tiles carry finite don't-care values.
            d4 ba1, Y2;
#pragma unroll
            for (int r = 0; r < 3; r++) ba1[r] = kR6 ? in.bv[r] : blend(mk_col0, in.bv[r], in.ba[r]);
            ba1[3] = 0.0;
            const d4 Pb = tn<3>(P, ba1, z4);
            d4 Racc = z4;
            if constexpr (ACC) Racc = tn<3>(ba1, acc->Psi, z4);   // [b | A(:,1:) B]' Psi: rows 1..11 = A'Psi, rows 12..15 = Z' = B'Psi, row 0 = b'Psi
            // the operand requests of stage i - 2 go here, into the wait for the product (18 idle cycles otherwise)
            __builtin_amdgcn_sched_barrier(0);
            mid();
            __builtin_amdgcn_sched_barrier(0);
            if (STORE_IPM) store_vec12(I.Pb + (size_t)(I.i0 + i) * 12, Pb, rg, cl);
#pragma unroll
            for (int r = 0; r < 3; r++) {
                // only column 0 of pv is the gradient.  Fused kernels: pv arrives unmasked (finite don't-care values of the previous
                // stage's product elsewhere) and is masked by the multiplication -- 47 cycles per stage less than blending it to zero
                // when it is produced; in the windowed and streaming kernels that form measured slower / spilled, they keep the blend
                if constexpr (kMaskPvAtUse) Y2[r] = fma(pv[r], col0f, Pb[r]);
                else Y2[r] = Pb[r] + pv[r];
            }
            Y2[3] = 0.0;
            d4 H, g;
            if constexpr (kR6) {
                d4 Cq;
                d4 dz = dgz;
                if constexpr (IT::kGrid) { dz = dg; dz[0] = lane == 0 ? 0.0 : dg[0]; }
#pragma unroll
                for (int r = 0; r < 3; r++) Cq[r] = fma(qr[r], col0f, dz[r]);
                Cq[3] = fma(qr[3], col0f, STEP0 ? dz[3] : dz[3] + (12 + rg == cl ? in.gm : 0.0));
                H = tn<3>(in.ba, Y2, Cq);
                g = H;   // column 0: the gradient
            } else {
                H = tn<3>(in.ba, Y2, z4);
#pragma unroll
                for (int r = 0; r < 4; r++) g[r] = H[r] + qr[r];
            }
            // column 0 of H := (row 0 of H)': lanes (0, c) hold H[0][c] in register 0, lane (rg, 0) needs H[rg + 4q][0].  Through
            // LDS; the values are consumed only after the pivot algebra (which touches columns 12..15), so the round trip is
            // off the chain.  No fence: one wave, LDS executes its operations in order.
            lds_f64* tr = I.lds_tr;
            tr[rg == 0 ? cl : 16] = H[0];                 // the other row groups are parked on a spare slot
            const double t0 = tr[rg], t1 = tr[rg + 4], t2 = tr[rg + 8], t3 = tr[rg + 12];
            // + diag(Ts*Wx, Ts*Wu + Gamma_i)
            if constexpr (!kR6) {
#pragma unroll
                for (int r = 0; r < 3; r++) H[r] += dg[r];
                H[3] += STEP0 ? dg[3] : dg[3] + (12 + rg == cl ? in.gm : 0.0);
            }
            // ---- 4x4 pivot block Huu = H[12..15][12..15]: lane 16m+12+n holds Huu[m][n] in H[3]
            const double a00 = readlane_f64(H[3], 12), a10 = readlane_f64(H[3], 28), a11 = readlane_f64(H[3], 29);
            const double a20 = readlane_f64(H[3], 44), a21 = readlane_f64(H[3], 45), a22 = readlane_f64(H[3], 46);
            const double a30 = readlane_f64(H[3], 60), a31 = readlane_f64(H[3], 61), a32 = readlane_f64(H[3], 62),
                         a33 = readlane_f64(H[3], 63);
            double m00 = 0, m10 = 0, m11 = 0, m20 = 0, m21 = 0, m22 = 0, m30 = 0, m31 = 0, m32 = 0, m33 = 0;   // M = Huu^-1 (lower triangle)
            double li00 = 0, li10 = 0, li11 = 0, li20 = 0, li21 = 0, li22 = 0, li30 = 0, li31 = 0, li32 = 0, li33 = 0;   // ROBUST: L^-1
            if constexpr (ROBUST) {
                // Cholesky Huu = L L' (all lanes redundantly), L^-1 by forward substitution, M = L^-T L^-1 for the stored operand
                const double i0 = fast_rsq(a00);
                const double l10 = a10 * i0, l20 = a20 * i0, l30 = a30 * i0;
                const double d1 = a11 - l10 * l10, i1 = fast_rsq(d1);
                const double l21 = (a21 - l20 * l10) * i1, l31 = (a31 - l30 * l10) * i1;
                const double d2 = a22 - (l20 * l20 + l21 * l21), i2 = fast_rsq(d2);
                const double l32 = (a32 - (l30 * l20 + l31 * l21)) * i2;
                const double d3 = a33 - (l30 * l30 + l31 * l31 + l32 * l32), i3 = fast_rsq(d3);
                if (!(a00 > 0.0 && d1 > 0.0 && d2 > 0.0 && d3 > 0.0)) ok = false;
                li00 = i0; li11 = i1; li22 = i2; li33 = i3;
                li10 = -(l10 * li00) * i1;
                li20 = -(l20 * li00 + l21 * li10) * i2; li21 = -(l21 * li11) * i2;
                li30 = -(l30 * li00 + l31 * li10 + l32 * li20) * i3; li31 = -(l31 * li11 + l32 * li21) * i3; li32 = -(l32 * li22) * i3;
                m00 = li00 * li00 + li10 * li10 + li20 * li20 + li30 * li30;
                m10 = li10 * li11 + li20 * li21 + li30 * li31; m11 = li11 * li11 + li21 * li21 + li31 * li31;
                m20 = li20 * li22 + li30 * li32; m21 = li21 * li22 + li31 * li32; m22 = li22 * li22 + li32 * li32;
                m30 = li30 * li33; m31 = li31 * li33; m32 = li32 * li33; m33 = li33 * li33;
            }
            double mt = 0.0, msel = 0.0;   // msel = M[rg][cl & 3] (kR6: -M), mt: the operand tile of T = M Hu
            if constexpr (!ROBUST) {
                // M = Huu^-1 by 2x2 block elimination (the uniform part in all lanes redundantly):
                //   Huu = [E F; F' G],  X = E^-1 F,  Sc = G - F'X,  M22 = Sc^-1,  M12 = -X M22,  M11 = E^-1 - M12 X'
                // Two reciprocals in sequence instead of the four of an LDL^T: this algebra is the serial critical path of
                // every Riccati stage.  SPD <=> e00, det E, s00, det Sc > 0.
                const double aa = a00 * a11;
                const double detE = fma(-a10, a10, aa), iE = fast_rcp(detE);
                const double e00 = a11 * iE, e01 = -a10 * iE, e11 = a00 * iE;           // E^-1
                // F = [a20 a30; a21 a31]^T block: rows 0,1 x cols 2,3 -> F = [[a20, a30], [a21, a31]]
                const double x00 = e00 * a20 + e01 * a21, x01 = e00 * a30 + e01 * a31;   // X = E^-1 F
                const double x10 = e01 * a20 + e11 * a21, x11 = e01 * a30 + e11 * a31;
                double s00, s01, s11;                                                      // Sc = G - F'X
                if constexpr (kR6) {
                    s00 = fma(-a20, x00, fma(-a21, x10, a22)); s01 = fma(-a20, x01, fma(-a21, x11, a32));
                    s11 = fma(-a30, x01, fma(-a31, x11, a33));
                } else {
                    s00 = a22 - (a20 * x00 + a21 * x10); s01 = a32 - (a20 * x01 + a21 * x11);
                    s11 = a33 - (a30 * x01 + a31 * x11);
                }
                const double ss = s00 * s11;
                const double detS = fma(-s01, s01, ss), iS = fast_rcp(detS);
                m22 = s11 * iS; m32 = -s01 * iS; m33 = s00 * iS;            // M22 = Sc^-1
                if (!(a00 > 0.0 && detE > 0.0 && s00 > 0.0 && detS > 0.0)) ok = false;
                if constexpr (kR6Z) {
                    // rows R and C of Z = [-X; I] and the entry of E^-1, per lane (independent of the second reciprocal)
                    const double nzr0 = fma(sR0, x00, fma(sR1, x10, nR2)), nzr1 = fma(sR0, x01, fma(sR1, x11, nR3));   // -Z[R][:]
                    const double zc0 = fma(nC0, x00, fma(nC1, x10, cC2)), zc1 = fma(nC0, x01, fma(nC1, x11, cC3));     //  Z[C][:]
                    const double nE = fma(nE00, e00, fma(nE01, e01, nE11 * e11));                                      // -E0[R][C]
                    const double w0 = fma(m22, zc0, m32 * zc1), w1 = fma(m32, zc0, m33 * zc1);                         // M22 Z[C][:]'
                    msel = fma(nzr0, w0, fma(nzr1, w1, nE));                                                           // -M[R][C]
                    mt = msel;   // columns >= 4 of the operand only reach rows >= 4 of the products, which nobody reads
                } else {
                    m20 = -(x00 * m22 + x01 * m32); m30 = -(x00 * m32 + x01 * m33);  // M12' (rows 2,3 x cols 0,1)
                    m21 = -(x10 * m22 + x11 * m32); m31 = -(x10 * m32 + x11 * m33);
                    m00 = e00 - (m20 * x00 + m30 * x01); m10 = e01 - (m20 * x10 + m30 * x11);
                    m11 = e11 - (m21 * x10 + m31 * x11);                          // M11 = E^-1 - M12 X'
                }
                // the relative pivots of the elimination (kPivotRho): four compares, off the chain
#ifndef BROV_EXP_NO_WATCH
                if constexpr (kR6) illm |= __ballot(detE < kPivotRho * aa) | __ballot(s00 < kPivotRho * a22) | __ballot(s11 < kPivotRho * a33) | __ballot(detS < kPivotRho * ss);
                else illc = illc | (detE < kPivotRho * aa) | (s00 < kPivotRho * a22) | (s11 < kPivotRho * a33) | (detS < kPivotRho * ss);
#endif
            }
            if constexpr (!kR6Z) {
                // Mtile: lane (rg = m, cl = n < 4) = M[m][n]; msel: the same element for every column n = cl & 3
                const int cq = cl & 3;
                const int a = rg > cq ? rg : cq, c = rg > cq ? cq : rg;  // (max, min)
                const double r0 = m00;
                const double r1 = (c == 0) ? m10 : m11;
                const double r2 = (c == 0) ? m20 : ((c == 1) ? m21 : m22);
                const double r3 = (c == 0) ? m30 : ((c == 1) ? m31 : ((c == 2) ? m32 : m33));
                msel = (a == 0) ? r0 : ((a == 1) ? r1 : ((a == 2) ? r2 : r3));
                mt = (kR6 || cl < 4) ? msel : 0.0;   // (kR6: see the Z form)
            }
            H[0] = blend(mk_col0, lane == 0 ? P[0] + dg[0] : t0, H[0]);   // H[0][0] = (P e_0)[0] + Ts W_0
            H[1] = blend(mk_col0, t1, H[1]);
            H[2] = blend(mk_col0, t2, H[2]);
            H[3] = blend(mk_col0, t3, H[3]);
            // T = M Hu (rows 0..3 in reg 0), S = H - Hu^T T, Kt = -(Hu^T M), kff = -M gu, p = gx + K^T gu
            d4 T, S;
            double ks, liT = 0.0, li = 0.0;
            if constexpr (ROBUST) {
                // L^-1 as operand tiles: element L^-1[max][min] selected per lane like M above; liT: (k, m) = L^-1[m][k] (so that the
                // product forms L^-1 y), li: (k, m) = L^-1[k][m] (forms L^-T y)
                const int cq = cl & 3;
                const int a = rg > cq ? rg : cq, c = rg > cq ? cq : rg;
                const double r1 = (c == 0) ? li10 : li11;
                const double r2 = (c == 0) ? li20 : ((c == 1) ? li21 : li22);
                const double r3 = (c == 0) ? li30 : ((c == 1) ? li31 : ((c == 2) ? li32 : li33));
                const double lsel = (a == 0) ? li00 : ((a == 1) ? r1 : ((a == 2) ? r2 : r3));
                liT = (cl < 4 && cl >= rg) ? lsel : 0.0;
                li = (cl < 4 && rg >= cl) ? lsel : 0.0;
                const d4 Y = tn1(liT, H[3], z4);     // Y = L^-1 Hu (rows 0..3 in register 0)
                S = tn1(Y[0], -Y[0], H);             // S = H - Y'Y: a difference of the stage Hessian and a Gram matrix
                T = tn1(li, Y[0], z4);               // L^-T Y = M Hu through the factor, not through the explicit inverse
                ks = -T[0];
            } else {
                if constexpr (kKffT) {
                    // Round 6.  kff = -M g_u rides in the gain product: columns 12..15 of T = -M Hu are -M Huu = -I, known without
                    // computing them, and nothing reads them (S and p use rows / columns 0..11 of what they feed) -- so column 12 of
                    // the right operand carries g_u instead of Huu(:, 0) (one row broadcast of column 0 of g, one select), and the
                    // separate product M g_u -- a whole 64-cycle MFMA for 16 multiply-adds -- is gone: 9 MFMAs per stage instead of 10.
                    // (p = g_x + Hu' kff out of column 12 of the Schur product in the same way -- 8 MFMAs -- was built too: the stage gets
                    // SLOWER, 30.2 -> 33.3 k cycles per sweep, because p then arrives WITH S and has to come back through the vector
                    // unit, where today its own product runs beside the next stage's P [A B]; profiles/r6_fused_experiments.txt.)
                    const double gub = dpp_f64<0x150>(g[3]);   // row_newbcast:0 -- g_u[rg] in every lane of row rg
                    T = tn1(mt, cl == 12 ? gub : H[3], z4);
                } else {
                    T = tn1(mt, H[3], z4);
                }
                ks = kR6Z ? T[0] : -T[0];
                S = tn1(H[3], ks, H);
            }
            // kff = -M gu and p = gx + K^T gu in ONE product: the operand carries the gain in columns 0..11 and M in columns
            // 12..15, so rows 0..11 of the result are p and rows 12..15 are M gu (M is symmetric)
            // (windowed kernel.  In the fused kernels the separate M gu product is what fills the issue slot behind T while S and
            // p wait for the gain: merged, the stage measured 110 cycles SLOWER there and 125 cycles faster in the windowed kernel.)
            const double xt2 = (cl < NX) ? ks : msel;
            d4 pn;
            if constexpr (ROBUST) {
                const d4 yg = tn1(liT, g[3], z4);    // L^-1 gu, then L^-T of it
                const d4 kf = tn1(li, yg[0], z4);
                pn = tn1(ks, g[3], g);
                pn[3] = kf[0];
            } else if constexpr (LDS == 3) {
                const d4 gC = {g[0], g[1], g[2], 0.0};
                pn = tn1(xt2, g[3], gC);
            } else if constexpr (kKffT) {
                pn = tn1(ks, g[3], g);
                pn[3] = ks;   // lanes of column 12: kff (stored from there)
            } else {
                const d4 kf = tn1(mt, g[3], z4);
                pn = tn1(ks, g[3], g);
                pn[3] = kf[0];
            }
            // store factors
            if (STORE_IPM) {  // only the corrector solve of an IPM iteration re-reads this: gain | M as one operand tile
                I.Ks[(size_t)(I.i0 + i) * 64 + lane] = kR6Z ? ((cl < NX) ? ks : -msel) : xt2;
            }
            if constexpr (LDS) {
                // K^T[k][m] = -T[m][k] is ks at lane (rg = m, cl = k): the compact LDS image [12][4] is written straight from
                // that register (no transposing MFMA); lanes cl >= 12 are parked on the constant-zero slot
                // ... as the gain itself, row-major [4][12] (what the VALU forward sweep reads: row m contiguous)
                if constexpr (kR6) {
                    *ktp = ks; ktp -= kt_ststr;
                } else {
                    lds_f64* t = (cl < NX) ? I.lds_kt + i * kKtStage + rg * NX + cl : I.lds_zero;
                    *t = (cl < NX) ? ks : 0.0;
                }
            } else {
                d4 KtT = tn1(H[3], -mt, z4);
                double* kt = I.Kt + (size_t)i * 192;
                kt[lane] = KtT[0]; kt[64 + lane] = KtT[1]; kt[128 + lane] = KtT[2];
            }
            if constexpr (LDS) {  // only column 0 of rows 12..15 is M gu: the other lanes are parked on the constant-zero slot
                if constexpr (kR6) {
                    *kfp = (kR6Z || kKffT) ? pn[3] : -pn[3]; kfp -= kf_ststr;   // (kR6Z: the operand tile carries -M; kKffT: pn[3] is ks, the gain operand itself)
                } else {
                    lds_f64* kp = (cl == 0) ? I.lds_kff + i * 4 + rg : I.lds_zero;
                    *kp = (cl == 0) ? -pn[3] : 0.0;
                }
            } else if (cl == 0) {
                I.kff[i * 4 + rg] = -pn[3];
            }
            if constexpr (ACC) {
                // off the Riccati chain (nothing of it feeds P or p): six products per stage (the first three requested at the head of the stage)
                d4 R = Racc;
                const double bPsi = R[0];
                R[0] = (rg == 0) ? acc->Psi[0] : R[0];         // the true row 0 of A'Psi is row 0 of Psi (column 0 of A is e_0)
                const d4 MZt = tn1(mt, R[3], z4);              // rows 0..3: M Z' -- what a costate at the segment end adds to this stage's feed-forward term
                const double MZ0 = kR6Z ? -MZt[0] : MZt[0];    // (kR6Z: the operand tile carries -M)
                I.Ks[(size_t)(I.i0 + i) * 64 + lane] = MZ0;    // (the gain | M tile of the in-loop sweeps lives there otherwise: no loop in this kernel)
                const double kffb = dpp_f64<0x150>(kR6Z ? pn[3] : -pn[3]);    // row_newbcast:0 -- kff_m in every lane of row m
                const double Xg = (cl < NX) ? MZ0 : ((cl == NX) ? kffb : 0.0);
                d4 Gn = tn1(Xg, R[3], acc->G);                 // rows 0..11 += Z M Z', row 12 += kff' Z'
                Gn[3] += (rg == 0) ? bPsi : 0.0;               // row 12 += b'Psi
                acc->G = Gn;
                d4 Pn = tn1(ks, R[3], d4{R[0], R[1], R[2], 0.0});   // (A + B K)' Psi
                Pn[3] = 0.0;
                acc->Psi = Pn;
            }
            P = S;
#pragma unroll
            for (int r = 0; r < 3; r++) pv[r] = kMaskPvAtUse ? pn[r] : blend(mk_col0, pn[r], 0.0);
            pv[3] = 0.0;
        } else {
            mid();
            d4 l;
#pragma unroll
            for (int r = 0; r < 4; r++) l[r] = in.bv[r] + pv[r];
            d4 g = tn<3>(in.ba, l, qr);
            const d4 gC = {g[0], g[1], g[2], 0.0};
            d4 pn = tn1(in.ks, g[3], gC);      // stored operand = gain | M: rows 0..11 p, rows 12..15 M gu
            if constexpr (LDS) I.lds_kff[i * 4 + rg] = -pn[3]; else if (cl == 0) I.kff[i * 4 + rg] = -pn[3];
            pv = pn;
            pv[3] = 0.0;
        }
    };
    if constexpr (LDS) {
        const int cnt = N - lo;   // stages N-1 .. lo
        if constexpr (LDS == 2) {   // two-wave kernel: the other wave fills those slots, and the variant below costs it registers it does not have
            pipelined<kLdsDist<LDS>, BwdIn>(cnt, [&](int k) { return load_bwd<FACTOR, LDS, STEP0>(I, N - 1 - k, gam, rt, b1); },
                                [&](int k, const BwdIn& in) { stage(N - 1 - k, in, [] {}); });
        } else {
            pipelined_mid<kLdsDist<LDS>, BwdIn>(cnt, [&](int k) { return load_bwd<FACTOR, LDS, STEP0>(I, N - 1 - k, gam, rt, b1); },
                                [&](int k, const BwdIn& in, auto&& issue) { stage(N - 1 - k, in, issue); });
        }
    } else {
        // distance 1 here: a stage is ~2 k cycles of issue per wave (4 k with the SIMD's second wave), enough to cover the
        // round trip, and a second stage in flight (36 VGPRs) pushes the kernel into scratch
        nx = load_bwd<FACTOR, LDS, STEP0>(I, N - 1, gam, rt);
        for (int i = N - 1; i >= 0; i--) {
            const BwdIn in = nx;
            if (i > 0) nx = load_bwd<FACTOR, LDS, STEP0>(I, i - 1, gam, rt);
            stage(i, in, [] {});
        }
    }
    if constexpr (FACTOR && kR6) illc = illc | (illm != 0ull);
}

template <bool FACTOR, int LDS, bool STORE_IPM = true, bool STEP0 = false, bool ROBUST = false, class IT = Inst>
__device__ bool riccati_backward(const IT& I, bool* illc = nullptr) {
    BwdState S;
    wave_fence();
    bwd_init<FACTOR, LDS>(I, S);
    bwd_chunk<FACTOR, LDS, STORE_IPM, STEP0, ROBUST>(I, S);
    if (illc) *illc = S.illc;
    return S.ok;
}

// Partial refactorisation (round 4; fused kernels).  P_i and p_i of the backward sweep depend only on the stages >= i.  An active-set
// try pins inputs of the first few stages almost always (a far-off instance saturates the START of its horizon: measured on the
// mixed batch, the last pinned stage is <= 4 for 98 % of the QPs that run the loop), so everything the step-0 sweep computed for the
// stages >= ckpt -- P, p, the gains and feed-forward terms in LDS -- is what a full sweep of the try would compute again, bit for
// bit (Gamma = 0 and the same right-hand side there).  The try restarts at stage ckpt - 1 from the checkpoint the step-0 sweep
// left in HBM: ckpt of N stages instead of N.  Valid while (a) no pinned input sits at a stage >= ckpt and (b) the LDS gains of
// those stages are still the step-0 ones (no full factor sweep has run inside the QP loop); the feed-forward terms, which every
// adjoint sweep overwrites with the input gradient, are restored from the register copy taken at loop entry.  The K^T area of the
// stages < ckpt = ceil(N / 4) is where the adjoint sweeps stage the multipliers: those stages are refactored in any case.
// part = false: a full sweep (what riccati_backward<true, LDS> does).  ONE call site of the stage loop for both.
template <int LDS, class IT = Inst>
__device__ __forceinline__ bool riccati_backward_tries(const IT& I, bool part, const double (&kff0)[2], bool& illc) {
    static_assert(LDS == 1 || LDS == 2, "fused kernels");
    wave_fence();
    BwdState S;
    if (part) {
#pragma unroll
        for (int t = 0; t < 2; t++) {   // out-of-range lanes rewrite element 0 (stage 0: recomputed by this sweep anyway)
            const int j = I.lane + 64 * t;
            I.lds_kff[j < I.nv ? j : 0] = kff0[t];
        }
        const double* ck = I.Kt;
#pragma unroll
        for (int r = 0; r < 3; r++) { S.P[r] = ck[r * 64 + I.lane]; S.pv[r] = ck[192 + r * 64 + I.lane]; }
        S.P[3] = 0.0; S.pv[3] = 0.0; S.ok = true;
    } else {
        bwd_init<true, LDS>(I, S);
    }
    bwd_chunk<true, LDS, true, false>(I, S, part ? I.ckpt : I.N, 0);
    illc = S.illc;
    return S.ok;
}

struct FwdIn { d4 kt, bat, bb; double kf; };
template <int LDS>
__device__ __forceinline__ FwdIn load_fwd(const Inst& I, int i) {
    FwdIn s;
    if constexpr (LDS) {
        const lds_f64* t = I.lds_ba + lmul(i, I.kt_str);  // offsets are relative to the start of the LDS slice
        s.kt = d4{t[I.kt_off[0]], t[I.kt_off[1]], t[I.kt_off[2]], 0.0};
    } else {
        s.kt = load_tile3(I.Kt + (size_t)i * 192, I.lane);
    }
    s.bat = get_bat<LDS>(I, i);
    s.bb = get_bv<LDS>(I, i);
    if constexpr (LDS) s.kf = I.lds_kff[i * 4 + I.rg]; else s.kf = I.kff[i * 4 + I.rg];
    return s;
}

// forward sweep of the closed loop: vhat_i = K_i dx_i + kff_i, dx_{i+1} = A dx_i + B vhat_i + b_i.
// Leaves vhat in I.vhat and the state steps in I.dxb.  fwd_chunk: the stages of the resident window, dx carried in xx.
// ---- vector recursions on the VALU (LDS-resident kernels) ---------------------------------------------------------------
// The forward, roll-out and adjoint sweeps are matrix-VECTOR recursions.  Round 1 ran them through the 16x16x4 MFMA with the vector
// row-replicated (no cross-lane movement, but 15 of the tile's 16 columns wasted: 7 MFMAs = 448 issue cycles per forward stage for
// 240 multiply-adds).  Here a 16-row x 16-column stage matrix is spread over the wave as lane (k, q) = (lane >> 2, lane & 3) <->
// row k, columns 4q..4q+3: four fmas per lane, a two-step DPP quad reduction, and the result vector goes through LDS (where the
// sweeps store it anyway) to come back as "four elements per lane".  A forward stage is ~60 VALU instructions + two LDS round trips.
__device__ __forceinline__ double quad_sum(double v) {
    v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    return v;
}
// "Vector in scalar registers" form of the recursions: lane k (of every 16-lane row: the four rows of the wave do the same work)
// owns ROW k of the stage matrix -- rows 0..11 = [A_i B_i] rows (x+), rows 12..15 = rows of the gain K_i (inputs) -- and forms the
// whole 12-term dot product itself against the state step held in SGPRs (v_fma with a scalar operand); the result vector goes
// back into SGPRs with v_readlane.  No cross-lane reduction, no LDS round trip on the chain: 12 + 4 fmas and 32 v_readlane per
// forward stage.
struct FwdV { double m[12], b4[4], cv; };
__device__ __forceinline__ FwdV load_fwd_v(const lds_f64* mrow, const lds_f64* klo, const lds_f64* brow, const lds_f64* cvec) {
    FwdV s;
#pragma unroll
    for (int c = 0; c < 3; c++) s.m[c] = klo[c];      // columns 0..2: real data only for the gain rows (A rows: structural e_c)
#pragma unroll
    for (int c = 3; c < 12; c++) s.m[c] = mrow[c];
#pragma unroll
    for (int t = 0; t < 4; t++) s.b4[t] = brow[t];
    s.cv = *cvec;
    return s;
}
// first: where the entering state step is staged for the sweep (default: row 0 of the block's state steps, which it IS; rti_pit_kernel:
// a scratch slot -- a segment's row 0 is the last row of the segment before it and is written by that segment's sweep only)
template <int LDS>
__device__ __forceinline__ void fwd_chunk(const Inst& I, d4& xx, lds_f64* first = nullptr) {
    const int rg = I.rg, cl = I.cl, N = I.N;
    lds_f64* const x_in = first ? first : I.lds_dxb;
    if constexpr (LDS) store_vec12_lds(x_in, xx, rg, cl); else store_vec12(I.dxb, xx, rg, cl);
    if constexpr (LDS) {
        const int k = I.lane & 15;
        const bool rowx = k < NX;
        const int ka = rowx ? k : NX - 1, km = rowx ? 0 : k - NX;
        // row k of [A_i | K_i]: A row k = ba[k*13 + c - 3] (columns 0..2 structural, overridden below), K row m = kt[m*12 + c]
        const lds_f64* mrow0 = rowx ? I.lds_ba + ka * kBaStride - 3 : I.lds_kt + km * NX;
        const int mstr = rowx ? kBaStage : kKtStage;
        const lds_f64* klo0 = I.lds_kt + km * NX;                              // always a valid address (A rows: value unused)
        const lds_f64* brow0 = I.lds_ba + ka * kBaStride + 9;                  // B row k (rows 12..15: unused)
        // (round 6, measured and not kept: reading the structural entries -- columns 0..2 of the A rows, a zero B row for the gain rows -- out of a
        // 4 x 4 table in LDS with stride 0 instead of selecting them removes 8 v_cndmask per stage and makes the sweep 21 cycles per stage SLOWER:
        // profiles/r6_fused_experiments.txt)
        const lds_f64* cvec0 = rowx ? I.lds_bv + k : I.lds_kff + km;          // b_k / kff_m
        const int cstr = rowx ? NX : 4;
        lds_f64* out0 = rowx ? I.lds_dxb + NX + k : I.lds_vhat + km;          // x+_k -> state-step row i+1, v_m -> inputs of stage i
        const int ostr = rowx ? NX : 4;
        const double e0 = (k == 0) ? 1.0 : 0.0, e1 = (k == 1) ? 1.0 : 0.0, e2 = (k == 2) ? 1.0 : 0.0;
        // The state step lives in ONE register, lane k (< 12) of a 16-lane row holding element k; the products take element c
        // straight out of lane c by DPP row broadcast (round 2 kept the vector in SGPRs: 32 v_readlane per stage).  First stage:
        // out of the LDS copy just written (one wave, LDS executes in order).
        // ONE 16-lane row runs the sweep.  Every row would compute the same thing, and every row's LDS reads cost LDS clocks: a
        // 64-lane ds_read_b64 occupies the CU's LDS (shared by the four resident waves, all of them in the same phase) for 4
        // clocks, a 16-lane one for 1 -- with 17 reads per stage that is the difference between 9.2 k and 7.7 k cycles per sweep.
        double xcur = x_in[rowx ? k : 0];
        lds_f64* outp = out0;   // running store address (stages in order 0 .. N-1)
        if (I.lane < 16)
        pipelined<kLdsDist<LDS>, FwdV>(N, [&](int kk) { return load_fwd_v(mrow0 + lmul(kk, mstr), klo0 + kk * kKtStage, brow0 + kk * kBaStage, cvec0 + lmul(kk, cstr)); },
                                       [&](int i, const FwdV& in) {
            const double m0 = rowx ? e0 : in.m[0], m1 = rowx ? e1 : in.m[1], m2 = rowx ? e2 : in.m[2];
            // The sweep is a recurrence on one wave: a dependent FP64 DPP operation issues ~13 cycles behind its producer
            // (measured: 65 cycles per stage for the B v chain), an independent one after ~5.  Four chains of three for the 12-term
            // products, a two-level sum, two chains of two for B v: 8 operations deep (three chains of four + serial B v: 10).
            double d0 = in.cv, d1 = 0.0, d2 = 0.0, d3 = 0.0;
            fmac_bc12(d0, d1, d2, d3, xcur, m0, m1, m2, in.m[3], in.m[4], in.m[5], in.m[6], in.m[7], in.m[8], in.m[9], in.m[10], in.m[11]);
            const double dot = (d0 + d1) + (d2 + d3);   // rows 12..15: v_m = K x + kff; rows 0..11: A x + b
            double xa = dot, xb = 0.0;                  // + B v, the inputs v_m out of lanes 12..15 of the same register
            fmac_bc4(xa, xb, dot, in.b4[0], in.b4[1], in.b4[2], in.b4[3]);
            const double xn = xa + xb;
            *outp = rowx ? xn : dot; outp += ostr;
            xcur = xn;
        });
        // the last state step back into the row-replicated form the callers carry between windows
        const lds_f64* xl = I.lds_dxb + N * NX + rg;
        xx = d4{xl[0], xl[4], xl[8], 0.0};
    }
    if constexpr (!LDS) {
        pipelined<2, FwdIn>(N, [&](int k) { return load_fwd<LDS>(I, k); }, [&](int i, const FwdIn& in) {
            d4 c = {in.kf, 0, 0, 0};
            d4 v = tn<3>(in.kt, xx, c);
            if (cl == 0) I.vhat[i * 4 + rg] = v[0];
            d4 z = {xx[0], xx[1], xx[2], v[0]};
            xx = tn<4>(in.bat, z, in.bb);
            xx[3] = 0.0;
            store_vec12(I.dxb + (size_t)(i + 1) * 12, xx, rg, cl);
        });
    }
}
template <int LDS>
__device__ void riccati_forward(const Inst& I, const d4& d0) {
    wave_fence();
    d4 xx = d0;
    fwd_chunk<LDS>(I, xx);
    wave_fence();
}

struct RollIn { d4 bat, bb; double v; };
template <int LDS>
__device__ __forceinline__ RollIn load_roll(const Inst& I, int i, const double* varr) {
    RollIn s;
    s.bat = get_bat<LDS>(I, i);
    s.bb = get_bv<LDS>(I, i);
    if constexpr (LDS) s.v = I.lds_vhat[i * 4 + I.rg]; else s.v = varr[i * 4 + I.rg];  // LDS path: inputs always staged in vhat
    return s;
}
// roll the linearised dynamics out for the inputs in varr -> I.dxb
struct RollV { double a[4], bk; };
__device__ __forceinline__ RollV load_roll_v(const Inst& I, int oa, int oa3, int ob, int i) {
    RollV s;
    const lds_f64* ba = I.lds_ba + i * kBaStage;
#pragma unroll
    for (int t = 0; t < 3; t++) s.a[t] = ba[oa + t];
    s.a[3] = ba[oa3];
    s.bk = I.lds_bv[i * NX + ob];
    return s;
}
template <int LDS>
__device__ __forceinline__ void roll_chunk(const Inst& I, d4& xx, const double* varr) {
    const int rg = I.rg, cl = I.cl, N = I.N;
    if constexpr (LDS) store_vec12_lds(I.lds_dxb, xx, rg, cl); else store_vec12(I.dxb, xx, rg, cl);
    if constexpr (LDS) {   // VALU form (see fwd_chunk): x+ = [A B] [x; v] + b with the inputs v staged in the LDS copy of vhat
        const int k = I.lane >> 2, q = I.lane & 3;
        const bool rowx = k < NX;
        const double e0 = (q == 0 && k == 0) ? 1.0 : 0.0, e1 = (q == 0 && k == 1) ? 1.0 : 0.0, e2 = (q == 0 && k == 2) ? 1.0 : 0.0;
        lds_f64* xpark = rowx ? I.lds_dxb + NX + k : I.lds_tr + (I.lane & 15);
        const int xstr = rowx ? NX : 0;
        const int ka = rowx ? k : NX - 1, c0 = 4 * q - 3;
        const int oa = ka * kBaStride + (c0 > 0 ? c0 : 0), oa3 = ka * kBaStride + c0 + 3, ob = ka;
        // z = [dx_i ; v_i]: column group q < 3 from the state-step row, q == 3 from the inputs; requested behind the store of the
        // previous stage and ahead of the operand prefetch (see fwd_chunk)
        const lds_f64* zr0 = q < 3 ? I.lds_dxb + 4 * q : I.lds_vhat;
        const int zstr = q < 3 ? NX : 4;
        double z0 = zr0[0], z1 = zr0[1], z2 = zr0[2], z3 = zr0[3];
        pipelined<kLdsDist<LDS>, RollV>(N, [&](int kk) { return load_roll_v(I, oa, oa3, ob, kk); }, [&](int i, const RollV& in) {
            const double a0 = q == 0 ? e0 : in.a[0], a1 = q == 0 ? e1 : in.a[1], a2 = q == 0 ? e2 : in.a[2];
            double pa = a0 * z0;
            pa = fma(a1, z1, pa); pa = fma(a2, z2, pa); pa = fma(in.a[3], z3, pa);
            const double xn = quad_sum(pa) + in.bk;
            xpark[lmul(i, xstr)] = rowx ? xn : 0.0;
            const lds_f64* zr = zr0 + lmul(i + 1 < N ? i + 1 : i, zstr);
            z0 = zr[0]; z1 = zr[1]; z2 = zr[2]; z3 = zr[3];
        });
        const lds_f64* xl = I.lds_dxb + N * NX + rg;
        xx = d4{xl[0], xl[4], xl[8], 0.0};
    } else {
        pipelined<3, RollIn>(N, [&](int k) { return load_roll<LDS>(I, k, varr); }, [&](int i, const RollIn& in) {
            d4 z = {xx[0], xx[1], xx[2], in.v};
            xx = tn<4>(in.bat, z, in.bb);
            xx[3] = 0.0;
            store_vec12(I.dxb + (size_t)(i + 1) * 12, xx, rg, cl);
        });
    }
}
template <int LDS>
__device__ void rollout(const Inst& I, const d4& d0, const double* varr) {
    wave_fence();
    d4 xx = d0;
    roll_chunk<LDS>(I, xx, varr);
    wave_fence();
}

struct AdjIn { d4 ba; double dx[3], xn[3], yn[3]; double v, u, ur; };
template <int LDS>
__device__ __forceinline__ AdjIn load_adj(const Inst& I, int i, const double* varr) {
    AdjIn s;
    s.ba = get_ba<LDS>(I, i);
    const double* dxn = I.dxb + (size_t)(i + 1) * 12;
    if constexpr (LDS) s.v = I.lds_vhat[i * 4 + I.rg]; else s.v = varr[i * 4 + I.rg];
    if constexpr (LDS) {
        const lds_f64* dxl = I.lds_dxb + (i + 1) * 12;
#pragma unroll
        for (int r = 0; r < 3; r++) { s.dx[r] = dxl[I.rg + 4 * r]; s.xn[r] = I.lds_q[(i + 1) * 12 + I.rg + 4 * r]; s.yn[r] = 0.0; }
        s.u = I.lds_r[i * 4 + I.rg];
        s.ur = 0.0;
    } else {
        const double* xn = I.x + (size_t)(i + 1) * 12;
        const double* yn = I.yref + (size_t)(i + 1) * 16;
#pragma unroll
        for (int r = 0; r < 3; r++) { s.dx[r] = dxn[I.rg + 4 * r]; s.xn[r] = xn[I.rg + 4 * r]; s.yn[r] = yn[I.rg + 4 * r]; }
        s.u = I.u[i * 4 + I.rg];
        s.ur = I.yref[(size_t)i * 16 + 12 + I.rg];
    }
    return s;
}
// adjoint recursion for the state steps in I.dxb and inputs varr:
//   pi_i = Qd_{i+1} dx_{i+1} + q_{i+1} + A_{i+1}' pi_{i+1};   g_i = Rd v_i + r_i + B_i' pi_i  -> garr[N*4]
// With COMMIT the multipliers pi are written to pi_out (the iterate).
// LDS path: both outputs go to LDS regions that are dead at this point (g -> the feed-forward array, pi -> the K^T array,
// 12 of its 48 doubles per stage); per-stage global stores would sit on vmcnt in front of every prefetch wait.
struct AdjV { double m[4], dxc, qc, vm, rm, wq, wr; };
template <class IT = Inst>
__device__ __forceinline__ AdjV load_adj_v(const IT& I, int om, int ox, int ou, int i) {
    AdjV s;
    // column c of [A_i B_i], rows 4q..4q+3 (columns 0..2 are e_c: not stored)
    const lds_f64* col = I.lds_ba + i * kBaStage + om;
#pragma unroll
    for (int t = 0; t < 4; t++) s.m[t] = col[t * kBaStride];
    s.dxc = I.lds_dxb[(i + 1) * NX + ox];
    s.qc = I.lds_q[(i + 1) * NX + ox];
    s.vm = I.lds_vhat[i * 4 + ou];
    s.rm = I.lds_r[i * 4 + ou];
    s.wq = 0.0; s.wr = 0.0;
    if constexpr (IT::kGrid) {   // general grid: scaled weights of node i + 1 (row NT = [We | 0]) and of the inputs of stage i
        s.wq = I.wst[(size_t)(I.i0 + i + 1) * 16 + ox];
        s.wr = I.wst[(size_t)(I.i0 + i) * 16 + 12 + ou];
    }
    return s;
}
template <bool COMMIT, int LDS, class IT = Inst>
__device__ __forceinline__ void adj_chunk(const IT& I, d4& atpi, const double* varr, double* garr, double* pi_out) {
    const int rg = I.rg, cl = I.cl, N = I.N;
    if constexpr (LDS) {
        // VALU form (see fwd_chunk): lane (c, q) = (lane >> 2, lane & 3) <-> column c of [A B], rows 4q..4q+3 (q < 3).
        //   pi_i[c] = Qd dx_{i+1}[c] + q_{i+1}[c] + (A_{i+1}' pi_{i+1})[c]     by the quad that owns c, through LDS to every lane,
        //   G = [A_i B_i]' pi_i: rows 0..11 feed the next stage, rows 12..15 are the input gradient.
        // The multipliers' LDS buffer is the K^T area (dead in every adjoint sweep; with COMMIT it is what the caller parks).
        const int c = I.lane >> 2, q = I.lane & 3;
        const bool rowx = c < NX, colx = q < 3;
        const int q3 = colx ? q : 2;
        const double e0 = (colx && 4 * q == c) ? 1.0 : 0.0, e1 = (colx && 4 * q + 1 == c) ? 1.0 : 0.0, e2 = (colx && 4 * q + 2 == c) ? 1.0 : 0.0,
                     e3 = (colx && 4 * q + 3 == c) ? 1.0 : 0.0;
        const bool ecol = c < 3;
        const int om = (4 * q3) * kBaStride + (c >= 3 ? c - 3 : 0), ox = rowx ? c : NX - 1, ou = c & 3;
        // A'pi of the stage after this window: row-replicated -> the quad that owns the row (through the transposition scratch)
        store_vec12_lds(I.lds_tr, atpi, rg, cl);
        double gq = I.lds_tr[rowx ? c : 0];
        lds_f64* ppark = rowx ? I.lds_kt + c : I.lds_tr + (I.lane & 15);
        const int pstr = rowx ? NX : 0;
        lds_f64* gpark = rowx ? I.lds_tr + (I.lane & 15) : I.lds_kff + (c - NX);
        const int gstr = rowx ? 0 : 4;
        const double rd = I.Ts * I.Wuq;
        lds_f64* pp = ppark + lmul(N - 1, pstr);   // running store addresses (stages in order N-1 .. 0)
        lds_f64* gp = gpark + lmul(N - 1, gstr);
        pipelined<kLdsDist<LDS>, AdjV>(N, [&](int kk) { return load_adj_v(I, om, ox, ou, N - 1 - kk); }, [&](int kk, const AdjV& in) {
            const int i = N - 1 - kk;
            const double qd = IT::kGrid ? in.wq : ((I.i0 + i + 1 == I.NT) ? I.Weq : I.Ts * I.Wq);
            const double pic = fma(qd, in.dxc, in.qc + gq);
            *pp = pic; pp -= pstr;   // (parked lanes store what they have: their slots are never read)
            const lds_f64* pr = I.lds_kt + i * NX + 4 * q3;
            const double p0 = pr[0], p1 = pr[1], p2 = pr[2], p3 = pr[3];
            __builtin_amdgcn_sched_barrier(0);
            const double m0 = ecol ? e0 : in.m[0], m1 = ecol ? e1 : in.m[1], m2 = ecol ? e2 : in.m[2], m3 = ecol ? e3 : in.m[3];
            double acc = m0 * p0;
            acc = fma(m1, p1, acc); acc = fma(m2, p2, acc); acc = fma(m3, p3, acc);
            acc = colx ? acc : 0.0;
            const double G = quad_sum(acc);
            *gp = fma(IT::kGrid ? in.wr : rd, in.vm, in.rm + G); gp -= gstr;   // (parked lanes: see above)
            gq = G;
        });
        // hand A'pi of this window's first stage on, row-replicated
        lds_f64* tpark = rowx ? I.lds_tr + c : I.lds_tr + 16;
        *tpark = gq;
        const lds_f64* tl = I.lds_tr + rg;
        atpi = d4{tl[0], tl[4], tl[8], 0.0};
        return;
    }
    const d4 z4 = {0, 0, 0, 0};
    auto stage = [&](int i, const AdjIn& in) __attribute__((always_inline)) {
        d4 pi;
        // scaled weights: of node i + 1 for the states (terminal: We), of stage i for the inputs; per stage on the general grid
        double wq[3], wr;
        if (I.wst) {
            const double* ws = I.wst + (size_t)(i + 1) * 16 + rg;
            wq[0] = ws[0]; wq[1] = ws[4]; wq[2] = ws[8];
            wr = I.wst[(size_t)i * 16 + 12 + rg];
        } else {
#pragma unroll
            for (int r = 0; r < 3; r++) wq[r] = (I.i0 + i + 1 == I.NT) ? I.Wer[r] : I.Ts * I.Wr[r];
            wr = I.Ts * I.Wr[3];
        }
#pragma unroll
        for (int r = 0; r < 3; r++) pi[r] = wq[r] * (in.dx[r] + in.xn[r] - in.yn[r]) + atpi[r];
        pi[3] = 0.0;
        if (COMMIT) store_vec12(pi_out + (size_t)i * 12, pi, rg, cl);
        d4 G = tn<3>(in.ba, pi, z4);
        const double rd = wr;
        if (cl == 0) garr[i * 4 + rg] = rd * in.v + rd * (in.u - in.ur) + G[3];
        atpi = G;
    };
    pipelined<3, AdjIn>(N, [&](int k) { return load_adj<LDS>(I, N - 1 - k, varr); },
                        [&](int k, const AdjIn& in) { stage(N - 1 - k, in); });
}
template <bool COMMIT, int LDS, class IT = Inst>
__device__ void adjoint(const IT& I, const double* varr, double* garr, double* pi_out) {
    wave_fence();
    d4 atpi = {0, 0, 0, 0};  // A_{i+1}' pi_{i+1}, rows 0..11
    adj_chunk<COMMIT, LDS>(I, atpi, varr, garr, pi_out);
    wave_fence();
}

}  // namespace brov
