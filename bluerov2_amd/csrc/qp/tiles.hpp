// qp/tiles.hpp -- tile primitives (v_mfma_f64_16x16x4 images, DPP reductions), the constants of the QP schedule, the per-wave instance record and its per-stage operand access.
// Part of ONE translation unit: qp_kernel.hip includes these headers in layer order (tiles -> sweeps -> window -> sched -> qp_body ->
// lin_phase -> fused -> windowed -> pit) and instantiates the kernels between them; see the file map at the head of qp_kernel.hip.
#pragma once

namespace brov {



typedef double d4 __attribute__((ext_vector_type(4)));
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) dbl2 lds_d2;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_cvoid;

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
// stage index x per-lane element stride (both far below 2^23): the 24-bit multiply issues at full rate, v_mul_lo_u32 -- what `i * stride` compiles
// to when the stride lives in a vector register -- at a quarter of it (16 cycles; round 6: two to four of them per stage of every sweep)
__device__ __forceinline__ int lmul(int i, int stride) { return __mul24(i, stride); }
// j / 12 for 0 <= j < 2^16 (element index -> stage): one full-rate 24-bit multiply and a shift; the compiler's division by a constant is a
// v_mul_hi_u32 (quarter rate).  43691 = ceil(2^19 / 12); exact while j * (43691 * 12 - 2^19) < 2^19, i.e. j < 131072.
__device__ __forceinline__ int div12(int j) { return (int)(__umul24((unsigned)j, 43691u) >> 19); }
// C + Xt^T Y over K4*4 rows
template <int K4>
__device__ __forceinline__ d4 tn(const d4& xt, const d4& y, d4 c) {
#pragma unroll
    for (int kk = 0; kk < K4; kk++) c = mfma(xt[kk], y[kk], c);
    return c;
}
// 4-deep contraction with explicitly chosen registers
__device__ __forceinline__ d4 tn1(double xt, double y, d4 c) { return mfma(xt, y, c); }

// m ? x : y for an all-ones / all-zeros lane mask, as two v_bfi_b32 (never a branch)
__device__ __forceinline__ double blend(unsigned m, double x, double y) {
    const unsigned lo = (__double2loint(x) & m) | (__double2loint(y) & ~m);
    const unsigned hi = (__double2hiint(x) & m) | (__double2hiint(y) & ~m);
    return __hiloint2double((int)hi, (int)lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// Wave reductions.  __shfl_xor is ds_bpermute (an LDS round trip per step, ~700 cycles for the six steps of a 64-lane
// butterfly with nothing to overlap); here the 16 lanes of a row are reduced with DPP moves (xor 1, xor 2, half-row mirror,
// row mirror) and the four row results are combined through v_readlane.  The result is wave-uniform.
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
template <class Op>
__device__ __forceinline__ double wave_reduce(double v, Op op) {
    v = op(v, dpp_f64<0xB1>(v));   // quad_perm [1,0,3,2]
    v = op(v, dpp_f64<0x4E>(v));   // quad_perm [2,3,0,1]
    v = op(v, dpp_f64<0x141>(v));  // row_half_mirror
    v = op(v, dpp_f64<0x140>(v));  // row_mirror
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ double wave_max(double v) { return wave_reduce(v, [](double a, double b) { return fmax(a, b); }); }
__device__ __forceinline__ double wave_min(double v) { return wave_reduce(v, [](double a, double b) { return fmin(a, b); }); }
__device__ __forceinline__ double wave_sum(double v) { return wave_reduce(v, [](double a, double b) { return a + b; }); }

// Data written by some lanes of the wave and read by others goes through global memory (L1/L2 of this CU); a
// workgroup-scope fence (= s_waitcnt, no cache maintenance) orders the two phases.
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }

// Start and step rule of the interior-point loop (the oracle uses the same three numbers; oracle/bluerov2_oracle.c says how they
// were chosen): a start close to the box (0.3 % of its width inside) with a small complementarity target needs 2 iterations
// where no bound is active and 4-5 where inputs saturate, instead of 4 and 7 with the textbook 0.1 / 0.995 / mu0 = g0.
#define IPM_TAU0 0.05   /* interior push of the start point (fraction of the box width; see the oracle) */
#define IPM_FTB 0.9999  /* fraction to the boundary of a (nearly) full step */
#define IPM_FTBLO 0.9   /* ... of a blocked step: alpha = a ((1 - a) FTBLO + a FTB), a = min(1, step to the boundary); see the oracle */
#define IPM_MU0F 0.1    /* mu0 = IPM_MU0F * stationarity residual of the clamped point */
// active-set tries around the interior-point loop: constants and schedule of the oracle (bluerov2_oracle.c "ACTIVE-SET POLISH")
#define POL_BIG 1e30      /* Hessian entry that pins an input */
#define POL_FIRST 5       /* tries before the first interior-point iteration (at most) */
#define POL_LOOP 3        /* ... per round after an interior-point iteration (at most) */
// the parallel-in-time kernel (qp/pit.hpp) runs qp_body's FIRST round of tries itself -- all POL_FIRST of them since round 5 (three before) -- and is
// offered every instance whose previous step succeeded with at most that many Newton systems (more: an interior-point iteration was involved)
#define PIT_TRIES POL_FIRST
#define POL_NCHG 8        /* a round ends when a try repairs more than this many inputs, or more than the try before it */
#define POL_MU_GATE 0.5   /* after a failed round the next one waits until the interior-point loop has cut mu by this factor ... */
#define POL_ALPHA_GATE 0.9 /* ... and has just taken a (nearly) full step */
#define POL_TOL_G 1e-9    /* wrong-signed multiplier of a pinned input: tolerated up to POL_TOL_G * R + POL_TOL_GREL * |g|max */
#define POL_TOL_GREL 1e-13

// everything one wave needs to know about its instance
struct Inst {
    int lane, rg, cl, N, nv;   // N = stages the sweeps run over (the whole horizon, or the resident window of it)
    int i0, NT;                // windowed kernel: global index of the window's first stage, total horizon (else 0, N)
    bool ran_loop;             // set by qp_body: the step ran the QP loop (no early exit)
    int ckpt;                  // fused kernels: the step-0 factor sweep leaves (P, p) entering stage ckpt - 1 in HBM (partial refactorisation); 0 = off
    const double* x;     // [N+1][12] entering iterate
    const double* u;     // [N][4]
    const double* yref;  // [N+1][16]
    const double* BA;    // [N][12][16]
    const double* bvec;  // [N][12]
    double *Ks, *Kt, *Mt, *Pb, *kff, *vhat, *ipm, *dxb;
    const lds_f64* lds_ba;  // fused path: [N][12][kBaStride] (+ b_i behind it), else unused
    const lds_f64* lds_bv;  // fused path: [N][12]
    lds_f64* lds_kt;        // fused path: gain transposed, compact [N][12][4]
    const lds_f64* lds_q;   // fused path: cost gradient q_i = s_i W (x_i - xref_i), [N+1][12] (terminal row N)
    const lds_f64* lds_r;   // fused path: r_i = Ts Wu (u_i - uref_i), [N][4]
    // per-lane element offsets into lds_ba for the three tile images (stage 0) and their per-stage strides: lanes whose
    // element is a structural constant (0 or 1) point at two constant slots with stride 0, so that a tile register is
    // ONE ds_read with an address known before the loop -- no select on the loaded value, which would pull the
    // s_waitcnt of a prefetch to the load itself
    int ba_off[3], ba_str, bat_off[4], bat_str, bat_str0, kt_off[3], kt_str;
    lds_f64* lds_tr;   // 17 doubles of LDS per wave: row -> column transposition in the backward sweep (+ 1 parking slot)
    lds_f64 *lds_kff, *lds_vhat, *lds_dxb, *lds_zero;  // fused path: same arrays as kff/vhat/dxb, typed as LDS so that the sweeps
                                            // issue ds_* instead of flat_*
    double Ts;
    const double* wst;   // streaming kernel, general grid: scaled weights per stage [N+1][16] (else nullptr)
    double Wr[4];   // W[row] for the lane's 4 rows (rows 12..15 = input weights)
    double Wer[3];  // We[row]
    double Wq, Weq, Wuq;  // adjoint sweep (lane = (column c, row group)): W[c], We[c] for c = min(lane >> 2, 11); W[12 + (lane >> 2 & 3)]
    double lbm, ubm;  // bounds of input m = rg
    static constexpr bool kGrid = false;
};
// General grid (round 4: also on the LDS-resident kernels): per-stage time steps and scaled weights (DevParams::tsv / wst).  The sweeps
// are generic in the instance type; where a loop-invariant Ts * W turns into a per-stage load they ask `IT::kGrid`, so the uniform-grid
// kernels are compiled exactly as before.
struct InstGrid : Inst { static constexpr bool kGrid = true; };

__device__ __forceinline__ d4 load_tile3(const double* base, int lane) {  // rows 0..11
    d4 t;
    t[0] = base[lane]; t[1] = base[64 + lane]; t[2] = base[128 + lane]; t[3] = 0.0;
    return t;
}
__device__ __forceinline__ d4 load_tile4(const double* base, int lane) {
    d4 t;
    t[0] = base[lane]; t[1] = base[64 + lane]; t[2] = base[128 + lane]; t[3] = base[192 + lane];
    return t;
}
// row-replicated 12-vector from contiguous memory
__device__ __forceinline__ d4 load_vec12(const double* v, int rg) {
    d4 t;
    t[0] = v[rg]; t[1] = v[rg + 4]; t[2] = v[rg + 8]; t[3] = 0.0;
    return t;
}
__device__ __forceinline__ void store_vec12(double* v, const d4& t, int rg, int cl) {
    if (cl == 0) { v[rg] = t[0]; v[rg + 4] = t[1]; v[rg + 8] = t[2]; }
}

// row-replicated vector -> LDS.  All 16 lanes of a row hold the same value and all of them store it (same address, same
// data): an exec-masked `if (cl == 0)` store becomes a branch, and the compiler then degrades every later lgkmcnt wait in
// the loop to lgkmcnt(0), exposing the LDS write latency once per stage.
__device__ __forceinline__ void store_vec12_lds(lds_f64* v, const d4& t, int rg, int cl) {
    (void)cl;
    v[rg] = t[0]; v[rg + 4] = t[1]; v[rg + 8] = t[2];
}

// ---- where the per-stage linearisation lives -------------------------------------------------------------------------
// LDS = false: streamed from HBM (tiles BA / bvec written by lin_wave_kernel) -- any horizon.
// LDS = true : the whole horizon's [A_i B_i] (row stride kBaStride doubles, padded so that both the row image and the
//              transposed image are read without bank conflicts) and b_i stay in this wave's LDS slice (fused kernel).
constexpr int kBaStride = 13;              // only the 13 non-trivial columns 3..15 are stored (odd stride: no bank conflicts
constexpr int kBaStage = NX * kBaStride;  // for either image); columns 0..2 of [A B] are exactly e_c
constexpr int kKtStage = NX * 4;          // K^T compact [12][4] per stage

template <int LDS>
__device__ __forceinline__ d4 get_ba(const Inst& I, int i) {  // [A B] image: rows k = rg+4r (0..11), cols c = cl
    if constexpr (LDS) {
        const lds_f64* t = I.lds_ba + lmul(i, I.ba_str);
        return d4{t[I.ba_off[0]], t[I.ba_off[1]], t[I.ba_off[2]], 0.0};
    } else {
        return load_tile3(I.BA + (size_t)i * 192, I.lane);
    }
}
template <int LDS>
__device__ __forceinline__ d4 get_bat(const Inst& I, int i) {  // [A B]^T image: rows c = rg+4r (0..15), cols k = cl (< 12)
    if constexpr (LDS) {
        const lds_f64* t = I.lds_ba + lmul(i, I.bat_str);
        return d4{I.lds_ba[lmul(i, I.bat_str0) + I.bat_off[0]], t[I.bat_off[1]], t[I.bat_off[2]], t[I.bat_off[3]]};
    } else {
        // transposed view of the row-major [A B] tile: element (c = rg + 4r, k = cl) = [A B](k, c); lanes cl >= 12 are padding.
        // Four 8-byte gathers that touch the tile's 12 cache lines -- cheaper than writing and re-reading a second, transposed
        // copy of every stage (2 KB per stage in round 1's first streaming version).
        const double* t = I.BA + (size_t)i * 192 + (I.cl < NX ? I.cl * 16 + I.rg : 0);
        const bool in = I.cl < NX;
        return d4{in ? t[0] : 0.0, in ? t[4] : 0.0, in ? t[8] : 0.0, in ? t[12] : 0.0};
    }
}
template <int LDS>
__device__ __forceinline__ d4 get_bv(const Inst& I, int i) {  // b_i, row-replicated
    if constexpr (LDS) {
        const lds_f64* t = I.lds_bv + i * NX + I.rg;
        return d4{t[0], t[4], t[8], 0.0};
    } else {
        return load_vec12(I.bvec + (size_t)i * 12, I.rg);
    }
}

// 1/d for a positive, normal d: v_rcp_f64 seed + 2 Newton steps (~1 ulp).  The pivot recursion below is the serial
// critical path of every Riccati stage; the IEEE-exact division sequence is 3x longer and buys nothing here.
__device__ __forceinline__ double fast_rcp(double d) {
    double y = __builtin_amdgcn_rcp(d);
    double e = fma(-d, y, 1.0);
    y = fma(y, e, y);
    e = fma(-d, y, 1.0);
    return fma(y, e, y);
}

// 1/sqrt(d) for a positive, normal d: v_rsq_f64 seed + 2 Newton steps (the robust pivot path below)
__device__ __forceinline__ double fast_rsq(double d) {
    double y = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    y = y * fma(-h * y, y, 1.5);
    return y * fma(-h * y, y, 1.5);
}

// acc += a * (src of lane K of this lane's 16-lane row): v_fmac_f64_dpp with row_newbcast, the one DPP control gfx950 has for
// 64-bit operands.  The broadcast costs nothing beyond the FMA (5.3 cycles against 4.9, scripts/dev/dpp_fmac_rate.hip) -- a
// v_readlane pair into SGPRs costs 8 plus the SGPR hazard.  A DPP read needs two wait states behind a VALU write of the register
// it reads, and the compiler's hazard recogniser does not see into inline assembly: round 2 issued one asm statement per product
// with an s_nop in front of the first one only, which left any VALU write the compiler might place between two of them (a copy,
// an AGPR reload of a spilled source) unguarded.  A whole chain is now ONE asm block behind one s_nop: nothing can be scheduled
// into it, and the source register is not written inside it.
//   fmac_bc12: d[c & 3] += m[c] * src@lane c, c = 0..11 (four independent chains of three: a dependent FP64 DPP operation issues
//              ~13 cycles behind its producer, an independent one after ~5)
//   fmac_bc4 : da += k0 * src@lane 12 + k1 * src@lane 13,  db += k2 * src@lane 14 + k3 * src@lane 15   (issue order 12, 14, 13, 15)
__device__ __forceinline__ void fmac_bc12(double& d0, double& d1, double& d2, double& d3, double src, double m0, double m1, double m2,
                                          double m3, double m4, double m5, double m6, double m7, double m8, double m9, double m10,
                                          double m11) {
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %4, %5 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %4, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %4, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %4, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %10 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %4, %11 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %4, %12 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %4, %13 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %4, %14 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %2, %4, %15 row_newbcast:10 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %3, %4, %16 row_newbcast:11 row_mask:0xf bank_mask:0xf"
        : "+&v"(d0), "+&v"(d1), "+&v"(d2), "+&v"(d3)   // (early clobber: an input with the same value -- the broadcast source -- must not share a register)
        : "v"(src), "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(m4), "v"(m5), "v"(m6), "v"(m7), "v"(m8), "v"(m9), "v"(m10), "v"(m11));
}
__device__ __forceinline__ void fmac_bc4(double& da, double& db, double src, double k0, double k1, double k2, double k3) {
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %2, %3 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %2, %5 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %2, %4 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %2, %6 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+&v"(da), "+&v"(db)
        : "v"(src), "v"(k0), "v"(k1), "v"(k2), "v"(k3));
}

// ---------------------------------------------------------------------------------------------------------------
// Sweeps.  Every sweep is software-pipelined by hand: all global operands of stage i+-1 are requested (plain loads into
// a second register set) before stage i is computed, so that HBM/L2 latency overlaps the MFMA chain of the current
// stage instead of being exposed once per stage (the in-order wave otherwise stalls ~1-2 us per stage).
// ---------------------------------------------------------------------------------------------------------------
// Streaming path: software pipeline over `count` steps with the HBM operands of step k + D requested before step k is
// computed (D + 1 register slots, rotated by unrolling so that no slot is ever copied).  One stage of a sweep is 0.6-2 k
// cycles of issue and two waves share a SIMD, while an HBM round trip under load is 4-5 k cycles: a prefetch distance of
// one stage leaves the sweeps waiting on memory half of the time.
template <int D, class In, class Load, class Body>
__device__ __forceinline__ void pipelined(int count, Load load, Body body) {
    constexpr int S = D + 1;
    In slot[S];
#pragma unroll
    for (int d = 0; d < D; d++) slot[d] = load(d < count ? d : count - 1);
    for (int k = 0; k < count; k += S) {
#pragma unroll
        for (int d = 0; d < S; d++) {
            const int kn = k + d + D;
            slot[(d + D) % S] = load(kn < count ? kn : count - 1);   // clamped: the tail re-requests the last stage
            if (k + d < count) body(k + d, slot[d]);
        }
    }
}

// The same pipeline with the request for step k + D issued from INSIDE step k: the body calls `issue()` where it has issue slots
// to spare (behind a chain of MFMAs whose result it has to wait for anyway) instead of ahead of its first instruction.
template <int D, class In, class Load, class Body>
__device__ __forceinline__ void pipelined_mid(int count, Load load, Body body) {
    constexpr int S = D + 1;
    In slot[S];
#pragma unroll
    for (int d = 0; d < D; d++) slot[d] = load(d < count ? d : count - 1);
    for (int k = 0; k < count; k += S) {
#pragma unroll
        for (int d = 0; d < S; d++) {
            const int kn = k + d + D;
            auto issue = [&]() __attribute__((always_inline)) { slot[(d + D) % S] = load(kn < count ? kn : count - 1); };
            if (k + d < count) body(k + d, slot[d], issue); else issue();
        }
    }
}

}  // namespace brov
