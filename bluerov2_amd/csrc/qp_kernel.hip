// qp_kernel.hip -- RTI feedback phase on gfx950: box-constrained OCP-QP by Riccati-based active-set rounds around a primal-dual
// interior-point loop (round 3; the schedule is the oracle's, bluerov2_oracle.c "ACTIVE-SET POLISH") + full-step SQP update,
// ONE WAVEFRONT PER OCP INSTANCE.
//
// Replaces what the reference hands to HPIPM through acados (FULL_CONDENSING_HPIPM, qp_iter_max 50;
// /root/reference/bluerov2_dobmpc/scripts/c_generated_code/acados_solver_bluerov2.c:146,664-669) and the
// update_variables step of SQP_RTI (:623,653-654).  Same QP, same unique minimiser; solved in its stage-structured
// form so the cost is O(N) instead of O(N^3).
//
// Mapping to CDNA4.  nx + nu = 12 + 4 = 16 is exactly one v_mfma_f64_16x16x4_f64 tile.  A 16x16 FP64 matrix lives
// in 4 VGPR pairs per lane in the MFMA C/D image  t[r] @ lane l  <->  element (row = (l>>4) + 4r, col = l&15),
// which is also its row-major image in HBM (offset r*64 + l: every tile load/store is a coalesced 512 B access).
// The one primitive is  tn<K4>(Xt, Y, C) = C + Xt^T * Y  (k = rows of Xt and Y): operands are fed to the MFMA
// straight from the C/D image, so chains of products need no cross-lane movement at all:
//     PA = P^T [A B]            H = [A B]^T PA + diag(Q,R+Gamma)        (12-deep contractions, 3 MFMA each)
//     T  = M Hu, S = H - Hu^T T  (Schur complement = next P), K^T = -Hu^T M   (4-deep, 1 MFMA each)
// with [x;u] ordering so that the input block Hu = [Hux Huu] is rows 12..15 = register 3 of the H tile.
// Vectors are carried "row-replicated" (lane holds v[row] for every column), which makes every matrix-vector
// product the same tn<> call.  The 4x4 pivot block is inverted redundantly by all lanes from v_readlane values.
// That is how the factor sweep works in every kernel and how ALL sweeps work in the streaming kernel.  In the LDS-resident
// kernels (rti_fused_kernel*, rti_window_kernel) the pure matrix-VECTOR recursions -- forward, roll-out, adjoint -- run on the
// VALU instead (fwd_chunk / roll_chunk / adj_chunk): a matrix-vector product fills one sixteenth of a 16x16x4 tile.
//
// File map: tile primitives, per-stage operand access -> sweeps (bwd_* / fwd_* / roll_* / adj_*, each split into an
// initialisation and a "stages of the resident window" part) -> window manager of the windowed kernel (Win, win_*, sw_*) ->
// work ordering (sched_*) -> qp_body (QP solve, multiplier recovery, full step; shared by all kernels) -> lin_phase (wave-wide
// linearisation) -> kernels (qp_kernel + lin_wave_kernel[_grid] streaming pair, rti_fused_kernel / _w2, rti_window_kernel and its
// resident-mode instantiation rti_window_kernel_res for small batches) and their launchers.
#include <type_traits>

#include "lin_device.hpp"
#include "nmpc_device.hpp"

namespace brov {


typedef double d4 __attribute__((ext_vector_type(4)));
typedef double dbl2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) dbl2 lds_d2;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_cvoid;

__device__ __forceinline__ d4 mfma(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
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
#define POL_NCHG 8        /* a round ends when a try repairs more than this many inputs, or more than the try before it */
#define POL_MU_GATE 0.5   /* after a failed round the next one waits until the interior-point loop has cut mu by this factor ... */
#define POL_ALPHA_GATE 0.9 /* ... and has just taken a (nearly) full step */
#define POL_TOL_G 1e-9    /* wrong-signed multiplier of a pinned input: tolerated up to POL_TOL_G * R + POL_TOL_GREL * |g|max */
#define POL_TOL_GREL 1e-13

// everything one wave needs to know about its instance
struct Inst {
    int lane, rg, cl, N, nv;   // N = stages the sweeps run over (the whole horizon, or the resident window of it)
    int i0, NT;                // windowed kernel: global index of the window's first stage, total horizon (else 0, N)
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
        const lds_f64* t = I.lds_ba + i * I.ba_str;
        return d4{t[I.ba_off[0]], t[I.ba_off[1]], t[I.ba_off[2]], 0.0};
    } else {
        return load_tile3(I.BA + (size_t)i * 192, I.lane);
    }
}
template <int LDS>
__device__ __forceinline__ d4 get_bat(const Inst& I, int i) {  // [A B]^T image: rows c = rg+4r (0..15), cols k = cl (< 12)
    if constexpr (LDS) {
        const lds_f64* t = I.lds_ba + i * I.bat_str;
        return d4{I.lds_ba[i * I.bat_str0 + I.bat_off[0]], t[I.bat_off[1]], t[I.bat_off[2]], t[I.bat_off[3]]};
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
        : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3)
        : "v"(src), "v"(m0), "v"(m1), "v"(m2), "v"(m3), "v"(m4), "v"(m5), "v"(m6), "v"(m7), "v"(m8), "v"(m9), "v"(m10), "v"(m11));
}
__device__ __forceinline__ void fmac_bc4(double& da, double& db, double src, double k0, double k1, double k2, double k3) {
    asm volatile(
        "s_nop 1\n\t"
        "v_fmac_f64_dpp %0, %2, %3 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %2, %5 row_newbcast:14 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %0, %2, %4 row_newbcast:13 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f64_dpp %1, %2, %6 row_newbcast:15 row_mask:0xf bank_mask:0xf"
        : "+v"(da), "+v"(db)
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

template <bool FACTOR, int LDS, bool STEP0 = false, class IT = Inst>
__device__ __forceinline__ BwdIn load_bwd(const IT& I, int i, const double* gam, const double* rt) {
    BwdIn s;
    s.ba = get_ba<LDS>(I, i);
    const int ig = I.i0 + i;   // HBM-resident operands are indexed by the global stage
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
            for (int r = 0; r < 3; r++) ba1[r] = blend(mk_col0, in.bv[r], in.ba[r]);
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
            d4 H = tn<3>(in.ba, Y2, z4);
            d4 g;
#pragma unroll
            for (int r = 0; r < 4; r++) g[r] = H[r] + qr[r];
            // column 0 of H := (row 0 of H)': lanes (0, c) hold H[0][c] in register 0, lane (rg, 0) needs H[rg + 4q][0].  Through
            // LDS; the values are consumed only after the pivot algebra (which touches columns 12..15), so the round trip is
            // off the chain.  No fence: one wave, LDS executes its operations in order.
            lds_f64* tr = I.lds_tr;
            tr[rg == 0 ? cl : 16] = H[0];                 // the other row groups are parked on a spare slot
            const double t0 = tr[rg], t1 = tr[rg + 4], t2 = tr[rg + 8], t3 = tr[rg + 12];
            // + diag(Ts*Wx, Ts*Wu + Gamma_i)
#pragma unroll
            for (int r = 0; r < 3; r++) H[r] += dg[r];
            H[3] += STEP0 ? dg[3] : dg[3] + (12 + rg == cl ? in.gm : 0.0);
            // ---- 4x4 pivot block Huu = H[12..15][12..15]: lane 16m+12+n holds Huu[m][n] in H[3]
            const double a00 = readlane_f64(H[3], 12), a10 = readlane_f64(H[3], 28), a11 = readlane_f64(H[3], 29);
            const double a20 = readlane_f64(H[3], 44), a21 = readlane_f64(H[3], 45), a22 = readlane_f64(H[3], 46);
            const double a30 = readlane_f64(H[3], 60), a31 = readlane_f64(H[3], 61), a32 = readlane_f64(H[3], 62),
                         a33 = readlane_f64(H[3], 63);
            double m00, m10, m11, m20, m21, m22, m30, m31, m32, m33;   // M = Huu^-1 (lower triangle)
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
            } else {
                // M = Huu^-1 by 2x2 block elimination (all lanes redundantly; the values are wave-uniform):
                //   Huu = [E F; F' G],  X = E^-1 F,  Sc = G - F'X,  M22 = Sc^-1,  M12 = -X M22,  M11 = E^-1 - M12 X'
                // Two reciprocals in sequence instead of the four of an LDL^T: this algebra is the serial critical path of
                // every Riccati stage (~26 dependent FP64 operations instead of ~48).  SPD <=> e00, det E, s00, det Sc > 0.
                const double detE = a00 * a11 - a10 * a10, iE = fast_rcp(detE);
                const double e00 = a11 * iE, e01 = -a10 * iE, e11 = a00 * iE;           // E^-1
                // F = [a20 a30; a21 a31]^T block: rows 0,1 x cols 2,3 -> F = [[a20, a30], [a21, a31]]
                const double x00 = e00 * a20 + e01 * a21, x01 = e00 * a30 + e01 * a31;   // X = E^-1 F
                const double x10 = e01 * a20 + e11 * a21, x11 = e01 * a30 + e11 * a31;
                const double s00 = a22 - (a20 * x00 + a21 * x10), s01 = a32 - (a20 * x01 + a21 * x11);
                const double s11 = a33 - (a30 * x01 + a31 * x11);                          // Sc = G - F'X
                const double detS = s00 * s11 - s01 * s01, iS = fast_rcp(detS);
                m22 = s11 * iS; m32 = -s01 * iS; m33 = s00 * iS;            // M22 = Sc^-1
                m20 = -(x00 * m22 + x01 * m32); m30 = -(x00 * m32 + x01 * m33);  // M12' (rows 2,3 x cols 0,1)
                m21 = -(x10 * m22 + x11 * m32); m31 = -(x10 * m32 + x11 * m33);
                m00 = e00 - (m20 * x00 + m30 * x01); m10 = e01 - (m20 * x10 + m30 * x11);
                m11 = e11 - (m21 * x10 + m31 * x11);                          // M11 = E^-1 - M12 X'
                if (!(a00 > 0.0 && detE > 0.0 && s00 > 0.0 && detS > 0.0)) ok = false;

                // the relative pivots of the elimination (kPivotRho): four compares, off the chain
#ifndef BROV_EXP_NO_WATCH
                illc = illc | (detE < kPivotRho * (a00 * a11)) | (s00 < kPivotRho * a22) | (s11 < kPivotRho * a33) | (detS < kPivotRho * (s00 * s11));
#endif
            }
            // Mtile: lane (rg = m, cl = n < 4) = M[m][n]; msel: the same element for every column n = cl & 3
            double mt = 0.0, msel;
            {
                const int cq = cl & 3;
                const int a = rg > cq ? rg : cq, c = rg > cq ? cq : rg;  // (max, min)
                const double r0 = m00;
                const double r1 = (c == 0) ? m10 : m11;
                const double r2 = (c == 0) ? m20 : ((c == 1) ? m21 : m22);
                const double r3 = (c == 0) ? m30 : ((c == 1) ? m31 : ((c == 2) ? m32 : m33));
                msel = (a == 0) ? r0 : ((a == 1) ? r1 : ((a == 2) ? r2 : r3));
                mt = (cl < 4) ? msel : 0.0;
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
                T = tn1(mt, H[3], z4);
                ks = -T[0];
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
            } else {
                const d4 kf = tn1(mt, g[3], z4);
                pn = tn1(ks, g[3], g);
                pn[3] = kf[0];
            }
            // store factors
            if (STORE_IPM) {  // only the corrector solve of an IPM iteration re-reads this: gain | M as one operand tile
                I.Ks[(size_t)(I.i0 + i) * 64 + lane] = xt2;
            }
            if constexpr (LDS) {
                // K^T[k][m] = -T[m][k] is ks at lane (rg = m, cl = k): the compact LDS image [12][4] is written straight from
                // that register (no transposing MFMA); lanes cl >= 12 are parked on the constant-zero slot
                // ... as the gain itself, row-major [4][12] (what the VALU forward sweep reads: row m contiguous)
                lds_f64* t = (cl < NX) ? I.lds_kt + i * kKtStage + rg * NX + cl : I.lds_zero;
                *t = (cl < NX) ? ks : 0.0;
            } else {
                d4 KtT = tn1(H[3], -mt, z4);
                double* kt = I.Kt + (size_t)i * 192;
                kt[lane] = KtT[0]; kt[64 + lane] = KtT[1]; kt[128 + lane] = KtT[2];
            }
            if constexpr (LDS) {  // only column 0 of rows 12..15 is M gu: the other lanes are parked on the constant-zero slot
                lds_f64* kp = (cl == 0) ? I.lds_kff + i * 4 + rg : I.lds_zero;
                *kp = (cl == 0) ? -pn[3] : 0.0;
            } else if (cl == 0) {
                I.kff[i * 4 + rg] = -pn[3];
            }
            if constexpr (ACC) {
                // off the Riccati chain (nothing of it feeds P or p): six products per stage (the first three requested at the head of the stage)
                d4 R = Racc;
                const double bPsi = R[0];
                R[0] = (rg == 0) ? acc->Psi[0] : R[0];         // the true row 0 of A'Psi is row 0 of Psi (column 0 of A is e_0)
                const d4 MZ = tn1(mt, R[3], z4);               // rows 0..3: M Z' -- what a costate at the segment end adds to this stage's feed-forward term
                I.Ks[(size_t)(I.i0 + i) * 64 + lane] = MZ[0];  // (the gain | M tile of the in-loop sweeps lives there otherwise: no loop in this kernel)
                const double kffb = dpp_f64<0x150>(-pn[3]);    // row_newbcast:0 -- kff_m in every lane of row m
                const double Xg = (cl < NX) ? MZ[0] : ((cl == NX) ? kffb : 0.0);
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
            pipelined<kLdsDist<LDS>, BwdIn>(cnt, [&](int k) { return load_bwd<FACTOR, LDS, STEP0>(I, N - 1 - k, gam, rt); },
                                [&](int k, const BwdIn& in) { stage(N - 1 - k, in, [] {}); });
        } else {
            pipelined_mid<kLdsDist<LDS>, BwdIn>(cnt, [&](int k) { return load_bwd<FACTOR, LDS, STEP0>(I, N - 1 - k, gam, rt); },
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
        const lds_f64* t = I.lds_ba + i * I.kt_str;  // offsets are relative to the start of the LDS slice
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
        if (I.lane < 16)
        pipelined<kLdsDist<LDS>, FwdV>(N, [&](int kk) { return load_fwd_v(mrow0 + kk * mstr, klo0 + kk * kKtStage, brow0 + kk * kBaStage, cvec0 + kk * cstr); },
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
            out0[i * ostr] = rowx ? xn : dot;
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
            xpark[i * xstr] = rowx ? xn : 0.0;
            const lds_f64* zr = zr0 + (i + 1 < N ? i + 1 : i) * zstr;
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
        pipelined<kLdsDist<LDS>, AdjV>(N, [&](int kk) { return load_adj_v(I, om, ox, ou, N - 1 - kk); }, [&](int kk, const AdjV& in) {
            const int i = N - 1 - kk;
            const double qd = IT::kGrid ? in.wq : ((I.i0 + i + 1 == I.NT) ? I.Weq : I.Ts * I.Wq);
            const double pic = fma(qd, in.dxc, in.qc + gq);
            ppark[i * pstr] = rowx ? pic : 0.0;
            const lds_f64* pr = I.lds_kt + i * NX + 4 * q3;
            const double p0 = pr[0], p1 = pr[1], p2 = pr[2], p3 = pr[3];
            __builtin_amdgcn_sched_barrier(0);
            const double m0 = ecol ? e0 : in.m[0], m1 = ecol ? e1 : in.m[1], m2 = ecol ? e2 : in.m[2], m3 = ecol ? e3 : in.m[3];
            double acc = m0 * p0;
            acc = fma(m1, p1, acc); acc = fma(m2, p2, acc); acc = fma(m3, p3, acc);
            acc = colx ? acc : 0.0;
            const double G = quad_sum(acc);
            gpark[i * gstr] = rowx ? 0.0 : fma(IT::kGrid ? in.wr : rd, in.vm, in.rm + G);
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

// ---------------------------------------------------------------------------------------------------------------------
// Windowed LDS residency (rti_window_kernel, horizons that do not fit the LDS slice: N >= 24).  The LDS slice holds the stage
// blocks of ONE window of <= 20 consecutive stages in exactly the layout of the fused kernel; the windows that are not
// resident are parked in a per-block HBM image (flat over the stages, array by array) and move as contiguous pieces:
// HBM -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR staging, one vmcnt wait per window),
// LDS -> HBM through registers in batches of eight 16-byte pieces per lane.  Every sweep is a loop over windows with its
// recursion state (P, p / dx / A'pi) carried in registers; a sweep fetches only the arrays it reads and writes back only the
// arrays it produced.  The window left resident by one sweep is the first window of the next one (sweeps alternate direction).
// LDS slice of the windowed kernel (L = stages per window), in this order:
//     [A B] L x 156 | b L x 12 | q (L+1) x 12 | r L x 4 | K^T L x 48 | kff L x 4 | vhat L x 4 | dx (L+1) x 12 | constants
// The first 236 L + 12 doubles (everything up to and including kff) are what a window parks in HBM: ONE contiguous image per
// window, a verbatim copy of the slice, so that parking and fetching are single contiguous transfers.  WM_LIN = the prefix
// [A B] | b | q | r (184 L + 12 doubles; what the backward, roll-out and adjoint sweeps read), WM_GAIN = K^T | kff (the rest;
// what the forward sweep reads in addition), WM_DX = the state steps (flat array of the whole horizon in HBM).
// (round 3) the prefix is tracked and fetched in three parts, so that a sweep moves only the arrays it reads: the forward sweep
// [A B] | b and K^T | kff (not q | r), the roll-out [A B] | b, the adjoint sweeps [A B] and q | r (not b)
enum : unsigned { WM_AB = 1, WM_BV = 2, WM_QR = 4, WM_LIN = 7, WM_GAIN = 8, WM_DX = 16 };
struct Win {
    int nc, Lc;         // number of windows, stages per window (the last one may be shorter)
    int cur;            // resident window
    unsigned valid;     // parts of the resident window that are valid in LDS
    double* lds;        // slice base (generic pointer)
    double* img;        // parked images of this block: nc x img_doubles(Lc)
#ifdef BROV_DBG_WIN
    unsigned long long t_fetch = 0, n_fetch = 0;   // development build: cycles spent waiting for window fetches, their number
#endif
    bool nan, feas;     // set by the forward / roll-out wrappers: a NaN among the inputs / state steps they produced; all inputs of
                        // the last forward sweep inside their bounds (wave-uniform)
};
__host__ __device__ constexpr int win_lin_doubles(int L) { return 184 * L + 12; }
__host__ __device__ constexpr int win_img_doubles(int L) { return 236 * L + 12; }
__host__ __device__ constexpr int win_off_bv(int L) { return 156 * L; }
__host__ __device__ constexpr int win_off_q(int L) { return 168 * L; }
__host__ __device__ constexpr int win_off_r(int L) { return 180 * L + 12; }
__host__ __device__ constexpr int win_off_kt(int L) { return 184 * L + 12; }
__host__ __device__ constexpr int win_off_kff(int L) { return 232 * L + 12; }
__host__ __device__ constexpr int win_off_vh(int L) { return 236 * L + 12; }
__host__ __device__ constexpr int win_off_dx(int L) { return 240 * L + 12; }
__host__ __device__ constexpr int win_off_const(int L) { return 252 * L + 24; }   // {0.0, 1.0} + 17 doubles of transposition scratch

// nd doubles (even, 16-byte aligned on both sides), HBM -> LDS, asynchronous: wait with s_waitcnt vmcnt(0) before reading
__device__ __forceinline__ void win_fetch(const double* g, double* l, int nd, int lane) {
    for (int o = 0; o < nd; o += 128)
        if (o + lane * 2 < nd) __builtin_amdgcn_global_load_lds((glb_cvoid*)(g + o + lane * 2), (lds_void*)(l + o), 16, 0, 0);
}
// LDS -> HBM; the LDS source may be overwritten as soon as this returns (its reads have landed in registers)
__device__ __forceinline__ void win_flush(double* g, const double* l, int nd, int lane) {
    const lds_d2* lv = (const lds_d2*)l;
    for (int o0 = 0; o0 < nd; o0 += 1024) {
        dbl2 v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int o = o0 + (k * 64 + lane) * 2;
            v[k] = lv[(o < nd ? o : 0) >> 1];
        }
        // opaque from here on: left alone the compiler re-reads each piece inside its store's guard (read, wait, store, eight times
        // over) instead of using the eight reads it has just issued back to back
#pragma unroll
        for (int k = 0; k < 8; k++) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y));
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const int o = o0 + (k * 64 + lane) * 2;
            if (o < nd) *(dbl2*)(g + o) = v[k];
        }
    }
}
// the same for short pieces (a few hundred doubles): batches of two 16-byte pieces per lane, no wasted predicated slots
__device__ __forceinline__ void win_flush_small(double* g, const double* l, int nd, int lane) {
    const lds_d2* lv = (const lds_d2*)l;
    for (int o0 = 0; o0 < nd; o0 += 256) {
        const int oa = o0 + lane * 2, ob = oa + 128;
        dbl2 va = lv[(oa < nd ? oa : 0) >> 1], vb = lv[(ob < nd ? ob : 0) >> 1];
        asm volatile("" : "+v"(va.x), "+v"(va.y), "+v"(vb.x), "+v"(vb.y));   // see win_flush
        if (oa < nd) *(dbl2*)(g + oa) = va;
        if (ob < nd) *(dbl2*)(g + ob) = vb;
    }
}
__device__ __forceinline__ void win_select(Inst& I, Win& W, int c) {
    if (c != W.cur) {
        W.cur = c;
        W.valid = 0;
        I.i0 = c * W.Lc;
        I.N = (I.NT - I.i0 < W.Lc) ? I.NT - I.i0 : W.Lc;
    }
}
// make window c resident with (at least) the parts in `mask`; vh_src != nullptr: the window's candidate inputs are fetched from
// that flat [N][4] array (they are never trusted to be resident: forward / interior-point loop / commit use different arrays)
__device__ __forceinline__ void win_need(Inst& I, Win& W, int c, unsigned mask, const double* vh_src) {
    win_select(I, W, c);
    const unsigned need = mask & ~W.valid;
    const int i0 = I.i0, n = I.N, lane = I.lane, L = W.Lc;
    const double* img = W.img + (size_t)c * win_img_doubles(L);
    __syncthreads();   // single wave: every lane is done with the slice's previous content, earlier stores are issued
#ifdef BROV_DBG_WIN
    const unsigned long long tf0 = __builtin_readcyclecounter();
#endif
    {   // the parts of the image in its order, neighbouring needed parts merged into one contiguous run
        const int beg[5] = {0, win_off_bv(L), win_off_q(L), win_off_kt(L), win_img_doubles(L)};
        const unsigned bit[4] = {WM_AB, WM_BV, WM_QR, WM_GAIN};
        int run0 = -1;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const bool want = (need & bit[k]) != 0;
            if (want && run0 < 0) run0 = beg[k];
            if (run0 >= 0 && (!want || k == 3)) {
                const int end = want ? beg[k + 1] : beg[k];
                win_fetch(img + run0, W.lds + run0, end - run0, lane);
                run0 = -1;
            }
        }
    }
    if (need & WM_DX) win_fetch(I.dxb + i0 * NX, W.lds + win_off_dx(L), (n + 1) * NX, lane);
    if (vh_src) win_fetch(vh_src + i0 * 4, W.lds + win_off_vh(L), n * 4, lane);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
#ifdef BROV_DBG_WIN
    if (need & (WM_LIN | WM_GAIN)) { W.t_fetch += __builtin_readcyclecounter() - tf0; W.n_fetch++; }
#endif
    W.valid |= mask;
}

// windowed kernel: every sweep re-derives the lane index behind an opaque move, so that its per-lane addresses are computed where the
// sweep starts and are not live across the other sweeps of the solve (the register file is full)
__device__ __forceinline__ void opaque_lane(Inst& I) {
    asm volatile("v_mov_b32 %0, %0" : "+v"(I.lane));
    I.rg = I.lane >> 4; I.cl = I.lane & 15;
}
// NaN among the window's candidate inputs / state steps (checked where they are produced, on the LDS copy)
template <bool RES>
__device__ __forceinline__ bool win_nan_check(const Inst& I, const Win& W, bool first) {
    const lds_f64* vh = (const lds_f64*)(W.lds + win_off_vh(W.Lc));
    const lds_f64* dx = (const lds_f64*)(W.lds + win_off_dx(W.Lc));
    // all six elements requested back to back, compared afterwards (and the caller must not short-circuit the call: under a
    // per-lane condition the whole body becomes an exec-masked block with one LDS wait per element)
    double v[6];
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int j = I.lane + 64 * t;
        v[t] = vh[j < I.N * 4 ? j : 0];
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int j = I.lane + 64 * t + (first ? 0 : NX);   // row 0 belongs to the previous window (d0 for the first one)
        v[2 + t] = dx[j < (I.N + 1) * NX ? j : NX];
    }
    bool bad = false;
#pragma unroll
    for (int t = 0; t < 6; t++) bad = bad | !(v[t] == v[t]);
    if constexpr (RES) {   // resident mode: windows longer than 20 stages
        for (int j = I.lane + 128; j < I.N * 4; j += 64) { const double e = vh[j]; bad = bad | !(e == e); }
        for (int j = I.lane + 256 + (first ? 0 : NX); j < (I.N + 1) * NX; j += 64) { const double e = dx[j]; bad = bad | !(e == e); }
    }
    return bad;
}
template <int LDS>
__device__ __forceinline__ void sw_forward(Inst& I, Win* W, const d4& d0, const double* cst = nullptr) {
    if constexpr (LDS < 3) {
        riccati_forward<LDS>(I, d0);
    } else {
        opaque_lane(I);
        wave_fence();
        d4 xx = d0;
        bool bad = false, infeas = false;
        // bound check of the candidate inputs, window by window on the LDS copy: this lane's elements j = lane + 64 t of a
        // window all belong to input m = lane & 3; the iterate's inputs of window c + 1 are requested before window c is swept
        const double lbm = cst ? cst[32 + (I.lane & 3)] : 0.0, ubm = cst ? cst[36 + (I.lane & 3)] : 0.0;
        auto load_u = [&](int c, double (&uw)[2]) __attribute__((always_inline)) {
            const int i0 = c * W->Lc, n = (I.NT - i0 < W->Lc) ? I.NT - i0 : W->Lc;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int j = I.lane + 64 * t;
                uw[t] = I.u[i0 * 4 + (j < n * 4 ? j : 0)];
            }
        };
        double uw[2] = {0.0, 0.0}, un[2] = {0.0, 0.0};
        if (cst) load_u(0, uw);
        for (int c = 0; c < W->nc; c++) {
            // (the last window also takes q | r along: the adjoint sweep that follows starts on it, and a separate fetch of those 332
            // doubles would cost a whole round trip)
            win_need(I, *W, c, c == W->nc - 1 ? (WM_LIN | WM_GAIN) : (WM_AB | WM_BV | WM_GAIN), nullptr);
            if (cst && c + 1 < W->nc) load_u(c + 1, un);
            fwd_chunk<3>(I, xx);
            __syncthreads();
            win_flush_small(I.vhat + I.i0 * 4, W->lds + win_off_vh(W->Lc), I.N * 4, I.lane);
            win_flush_small(I.dxb + I.i0 * NX, W->lds + win_off_dx(W->Lc), (I.N + 1) * NX, I.lane);
            W->valid |= WM_DX;
            bad = bad | win_nan_check<LDS == 4>(I, *W, c == 0);
            if (cst) {
                const lds_f64* vh = (const lds_f64*)(W->lds + win_off_vh(W->Lc));
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int j = I.lane + 64 * t;
                    const double vj = vh[j < I.N * 4 ? j : 0], lb = lbm - uw[t], ub = ubm - uw[t];   // read unconditionally (clamped)
                    infeas = infeas | ((j < I.N * 4) & !(vj >= lb && vj <= ub));
                }
                if constexpr (LDS == 4) {   // resident mode: windows longer than 20 stages
                    for (int j = I.lane + 128; j < I.N * 4; j += 64) {
                        const double vj = vh[j], uj = I.u[I.i0 * 4 + j];
                        infeas = infeas | !(vj >= lbm - uj && vj <= ubm - uj);
                    }
                }
                uw[0] = un[0]; uw[1] = un[1];
            }
        }
        W->nan = __ballot(bad) != 0ull;
        W->feas = __ballot(infeas) == 0ull;
        wave_fence();
    }
}
template <int LDS>
__device__ __forceinline__ void sw_rollout(Inst& I, Win* W, const d4& d0, const double* varr) {
    if constexpr (LDS < 3) {
        rollout<LDS>(I, d0, varr);
    } else {
        opaque_lane(I);
        wave_fence();
        d4 xx = d0;
        bool bad = false;
        for (int c = 0; c < W->nc; c++) {
            win_need(I, *W, c, WM_AB | WM_BV, varr);
            roll_chunk<3>(I, xx, varr);
            __syncthreads();
            win_flush_small(I.dxb + I.i0 * NX, W->lds + win_off_dx(W->Lc), (I.N + 1) * NX, I.lane);
            W->valid |= WM_DX;
            bad = bad | win_nan_check<LDS == 4>(I, *W, c == 0);
        }
        W->nan = __ballot(bad) != 0ull;
        wave_fence();
    }
}
template <bool COMMIT, int LDS, class IT = Inst>
__device__ __forceinline__ void sw_adjoint(IT& I, Win* W, const double* varr, double* garr, double* pi_out) {
    if constexpr (LDS < 3) {
        adjoint<COMMIT, LDS>(I, varr, garr, pi_out);
    } else {
        opaque_lane(I);
        wave_fence();
        d4 atpi = {0, 0, 0, 0};
        for (int c = W->nc - 1; c >= 0; c--) {
            win_need(I, *W, c, WM_AB | WM_QR | WM_DX, varr);
            adj_chunk<COMMIT, 3>(I, atpi, varr, garr, pi_out);
            W->valid &= ~WM_GAIN;   // multipliers / input gradient were staged in the K^T / feed-forward areas
            __syncthreads();
            if (COMMIT) win_flush(pi_out + (size_t)I.i0 * NX, W->lds + win_off_kt(W->Lc), I.N * NX, I.lane);
            win_flush(garr + I.i0 * 4, W->lds + win_off_kff(W->Lc), I.N * 4, I.lane);
        }
        wave_fence();
    }
}
// Windowed kernel: final adjoint sweep and the full step in one pass over the windows.  While a window is resident its state
// steps, inputs, input gradient and multipliers are all in LDS; the iterate rows and the reference of the window are requested
// before the window is fetched and swept, so the step costs no exposed HBM round trip.  cost: this lane's share of the NLS
// objective at the updated iterate; u0v: lanes 0..3 the new first input.
template <bool RES, class Mid, class IT = Inst>
__device__ __forceinline__ void win_adjoint_commit(const DevParams& P, IT& I, Win& W, int b, const double* vfin, bool early,
                                                   double& cost, double& u0v, bool deliver_first, Mid&& mid) {
    const int lane = I.lane, NT = I.NT, L = W.Lc;
    const double* __restrict__ cst = P.cst;
    double* x_it = P.x + (size_t)b * (NT + 1) * 12;
    double* u_it = P.u + (size_t)b * NT * 4;
    double* pi_it = P.pi + (size_t)b * NT * 12;
    double* lam_it = P.lam + (size_t)b * NT * 8;
    const lds_f64* vh = (const lds_f64*)(W.lds + win_off_vh(L));
    const lds_f64* gl = (const lds_f64*)(W.lds + win_off_kff(L));
    const lds_f64* dx = (const lds_f64*)(W.lds + win_off_dx(L));
    wave_fence();
    d4 atpi = {0, 0, 0, 0};
    for (int c = W.nc - 1; c >= 0; c--) {
        win_select(I, W, c);
        const int i0 = I.i0, n = I.N;
        const int nu = n * 4, nxr = (c == W.nc - 1 ? n + 1 : n) * NX;   // the last window also commits the terminal node
        double uo[2], ur[2], wu[2], xo[4], yr[4], wx[4];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int j = lane + 64 * t, jj = j < nu ? j : 0;
            uo[t] = u_it[i0 * 4 + jj];
            ur[t] = I.yref[(size_t)(i0 + (jj >> 2)) * 16 + 12 + (jj & 3)];
            wu[t] = IT::kGrid ? I.wst[(size_t)(i0 + (jj >> 2)) * 16 + 12 + (jj & 3)] : P.Ts * cst[12 + (jj & 3)];   // scaled input weight of the stage
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int j = lane + 64 * t, jj = j < nxr ? j : 0;
            const int i = jj / 12, cc = jj - i * 12;
            xo[t] = x_it[i0 * 12 + jj];
            yr[t] = I.yref[(size_t)(i0 + i) * 16 + cc];
            wx[t] = IT::kGrid ? I.wst[(size_t)(i0 + i) * 16 + cc] : ((i0 + i == NT) ? cst[16 + cc] : P.Ts * cst[cc]);
        }
        auto adjoint_part = [&]() __attribute__((always_inline)) {
            win_need(I, W, c, WM_AB | WM_QR | WM_DX, vfin);
            adj_chunk<true, 3>(I, atpi, vfin, nullptr, nullptr);
            W.valid &= ~WM_GAIN;   // multipliers / input gradient are staged in the K^T / feed-forward areas
            __syncthreads();
            win_flush_small(pi_it + (size_t)i0 * NX, W.lds + win_off_kt(L), n * NX, lane);
        };
        auto update_part = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int j = lane + 64 * t;
                if (j < nu) {
                    const int i = j >> 2, m = j & 3;
                    const double gg = early ? 0.0 : (double)gl[j];
                    lam_it[(size_t)(i0 + i) * 8 + m] = gg > 0 ? gg : 0.0;
                    lam_it[(size_t)(i0 + i) * 8 + 4 + m] = gg < 0 ? -gg : 0.0;
                    const double un = uo[t] + vh[j];
                    u_it[i0 * 4 + j] = un;
                    if (i0 == 0 && j < 4) { P.res[b].u0[j] = un; u0v = un; }
                    const double e = un - ur[t];
                    cost += 0.5 * wu[t] * e * e;
                }
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int j = lane + 64 * t;
                if (j < nxr) {
                    const double xn = xo[t] + dx[j];
                    x_it[i0 * 12 + j] = xn;
                    const double e = xn - yr[t];
                    cost += 0.5 * wx[t] * e * e;
                }
            }
            if constexpr (RES) {   // resident mode: windows longer than 20 stages, the elements beyond the preloaded 128 / 256
                for (int j = lane + 128; j < nu; j += 64) {
                    const int i = j >> 2, m = j & 3;
                    const double gg = early ? 0.0 : (double)gl[j];
                    lam_it[(size_t)(i0 + i) * 8 + m] = gg > 0 ? gg : 0.0;
                    lam_it[(size_t)(i0 + i) * 8 + 4 + m] = gg < 0 ? -gg : 0.0;
                    const double un = u_it[i0 * 4 + j] + vh[j];
                    u_it[i0 * 4 + j] = un;
                    const double e = un - I.yref[(size_t)(i0 + i) * 16 + 12 + m];
                    cost += 0.5 * (IT::kGrid ? I.wst[(size_t)(i0 + i) * 16 + 12 + m] : P.Ts * cst[12 + m]) * e * e;
                }
                for (int j = lane + 256; j < nxr; j += 64) {
                    const int i = j / 12, cc = j - i * 12;
                    const double xn = x_it[i0 * 12 + j] + dx[j];
                    x_it[i0 * 12 + j] = xn;
                    const double e = xn - I.yref[(size_t)(i0 + i) * 16 + cc];
                    cost += 0.5 * (IT::kGrid ? I.wst[(size_t)(i0 + i) * 16 + cc] : ((i0 + i == NT) ? cst[16 + cc] : P.Ts * cst[cc])) * e * e;
                }
            }
        };
        if constexpr (RES) {
            // Resident mode (one window).  An equality-constrained answer needs nothing of the adjoint sweep for its step (its bound
            // multipliers are zero): with deliver_first the step and the record go out first -- mid() hands the record to the host
            // mailbox -- and the multipliers pi of the iterate follow.  The two parts run in either order out of ONE copy each.
            const bool update_first = early && deliver_first;
#pragma clang loop unroll(disable)
            for (int ph = 0; ph < 2; ph++) {
                if ((ph == 0) == update_first) {
                    update_part();
                    if (update_first) mid(cost, u0v);
                } else {
                    adjoint_part();
                }
            }
        } else {
            adjoint_part();
            update_part();
        }
    }
    wave_fence();
}

// STEP0: the equality-constrained system (Gamma = 0, right-hand side r; nothing is read from or stored to the interior-point
// arrays); ROBUST: the Cholesky pivot form (kPivotRho); illc: an ill-conditioned pivot block was seen (fast form only)
// part (windowed kernel): only window 0 is refactorised, from the checkpoint pass 1 left behind (the parked gains of the other windows are
// the step-0 ones, and a try that pins inputs of window 0 only would recompute them bit for bit)
template <bool FACTOR, int LDS, bool STEP0 = false, bool ROBUST = false, class IT = Inst>
__device__ __forceinline__ bool sw_backward(IT& I, Win* W, bool* illc = nullptr, bool part = false) {
    if constexpr (LDS < 3) {
        return riccati_backward<FACTOR, LDS, !STEP0, STEP0, ROBUST>(I, illc);
    } else {
        opaque_lane(I);
        wave_fence();
        BwdState S;
        if (part) {
            const double* ck = I.Kt;
#pragma unroll
            for (int r = 0; r < 3; r++) { S.P[r] = ck[r * 64 + I.lane]; S.pv[r] = ck[192 + r * 64 + I.lane]; }
            S.P[3] = 0.0; S.pv[3] = 0.0; S.ok = true;
            if constexpr (LDS == 4) {
                // resident mode: the stages >= ckpt keep their step-0 gains in LDS (the adjoint sweeps stage the multipliers in the K^T area
                // of the stages < N / 4 <= ckpt only), but their feed-forward terms have been overwritten by an input gradient: back from
                // the copy qp_body took at loop entry
                const double* kf = I.Kt + 384;
                for (int j = I.lane; j < I.NT * 4; j += 64) I.lds_kff[j] = kf[j];
            }
        }
        const int hi = (LDS == 4 && part) ? I.ckpt : -1;   // resident mode: the stages ckpt - 1 .. 0 of the one window
        for (int c = part ? 0 : W->nc - 1; c >= 0; c--) {
            win_need(I, *W, c, WM_LIN, nullptr);
            if (c == W->nc - 1 && !part) bwd_init<FACTOR, 3>(I, S);
            bwd_chunk<FACTOR, 3, !STEP0, STEP0, ROBUST>(I, S, hi, 0);
            __syncthreads();
            // park what the sweep produced: K^T | kff (contiguous), or kff alone after a solve-only sweep.  The resident K^T stays
            // valid in both cases (a solve-only sweep does not touch it) unless an adjoint sweep has overwritten the area since.
            double* img = W->img + (size_t)c * win_img_doubles(W->Lc);
            if constexpr (LDS != 4) {   // (resident mode: the single window keeps what the sweep produced where it is)
                if (FACTOR) win_flush(img + win_off_kt(W->Lc), W->lds + win_off_kt(W->Lc), 52 * W->Lc, I.lane);
                else win_flush(img + win_off_kff(W->Lc), W->lds + win_off_kff(W->Lc), 4 * W->Lc, I.lane);
            }
            if (FACTOR) W->valid |= WM_GAIN;
        }
        wave_fence();
        if (illc) *illc = S.illc;
        return S.ok;
    }
}

// one interior-point vector.  MODE 1 (fused kernels, nv <= 128): two elements per lane, in registers for the whole loop.
// MODE 0 (streaming kernel): an HBM array, read and written element by element.  MODE 2 (windowed kernel): an HBM array with a
// register copy of the lane's T elements that lives for one group of element loops -- fetch() at the head of the group (all the
// group's loads are requested back to back, ahead of its first store: written element by element the compiler has to keep every
// load behind the previous element's stores, which may alias, and the single resident wave then sits through one L2 / HBM
// round trip per element instead of one per group), flush() at its end.  Between groups (across the sweeps) only HBM holds it.
template <int MODE, int T>
struct IpmVec {
    double r[T];
    double* g;
    __device__ __forceinline__ double get(int t, int j) const { return MODE ? r[t % T] : g[j]; }
    __device__ __forceinline__ void set(int t, int j, double v) { if (MODE) r[t % T] = v; else g[j] = v; }
    // element indices are UNSIGNED: base pointer (uniform, SGPR pair) + zero-extended 32-bit offset is one addressing mode of
    // global_load / global_store, so the 8 offsets of a lane serve every vector; with a signed index the compiler forms one 64-bit
    // address per element and vector (160 VGPRs in the windowed kernel) and keeps them all live across the interior-point loop
    __device__ __forceinline__ void fetch(int lane, int nv) {
        if constexpr (MODE == 2) {
#pragma unroll
            for (int t = 0; t < T; t++) { const unsigned j = (unsigned)lane + 64u * t; r[t] = g[j < (unsigned)nv ? j : 0u]; }
        }
    }
    __device__ __forceinline__ void flush(int lane, int nv) const {
        if constexpr (MODE == 2) {
#pragma unroll
            for (int t = 0; t < T; t++) { const unsigned j = (unsigned)lane + 64u * t; if (j < (unsigned)nv) g[j] = r[t]; }
        }
    }
};

// ---- work ordering: expensive instances first ----------------------------------------------------------------------------------
// A launch ends with its slowest instance, and which instances are slow is known in advance with good odds: an instance whose QP had
// active bounds in the previous control tick (it ran active-set tries / interior-point iterations: 2 .. 6 times the cycles of an
// early exit) almost always has them again in this one.  Every solve therefore records the instances that entered the QP loop
// (atomic append to a list L of length n, and pos[b] = position in L or -1), and the next solve hands THOSE out first:
//     (per class of instances, see below)
//     ticket t <  n            -> instance L[t]
//     ticket t >= n, pos[t] < 0 -> instance t
//     ticket t >= n, pos[t] >= 0 (t is in L): the prefix instance its list position names, following pos while that instance is
//                                 itself in L -- the chain ends on a prefix instance outside L, and two chains never meet (pos is
//                                 injective on L), so the map is a bijection of [0, B)
// Tickets are block indices (fused / streaming kernels: the hardware dispatches blocks in index order) or the atomic counter's
// values (windowed kernel).  Only the ORDER of the work changes: every instance is still solved by one wave on its own data, the
// results are bit-identical with and without (tests/test_gpu_edge.py).  Three buffers rotate: read (written by the previous
// solve), written, and zeroed for the next solve.  Measured: mixed batch 17.4 -> see DESIGN.md section 7.
// Contention: all resident waves reach the end of an equally long solve within microseconds of each other, and atomics on ONE
// address serialise (~4 ns each: 4 us per round of 1024 waves when every instance runs the loop, 7 % of that leg).  The instances
// are therefore split into 64 classes (index mod 64), each with its own counter (on its own 128-byte line), list and bijection;
// ticket t is served by class t mod 64, position t / 64.
constexpr int kSchedClasses = 64, kSchedCntStride = 32;
__host__ __device__ inline int sched_class_len(int B) { return (B + kSchedClasses - 1) / kSchedClasses; }
__host__ __device__ inline int sched_buffer_ints(int B) { return kSchedClasses * kSchedCntStride + kSchedClasses * sched_class_len(B) + B; }
__device__ __forceinline__ int sched_map(const DevParams& P, int t) {
    if (!P.sched) return t;
    const int32_t* __restrict__ Rd = P.sched + (size_t)P.sched_r * P.sched_stride;
    const int k = t & (kSchedClasses - 1), i = t >> 6, Bc = sched_class_len(P.B);
    const int n = Rd[k * kSchedCntStride];
    // nothing to gain when most instances of the class are listed (every ticket would pay a dependent look-up for an order that does
    // not matter)
    if (n <= 0 || 2 * n > Bc) return t;
    const int32_t* __restrict__ L = Rd + kSchedClasses * kSchedCntStride + k * Bc;
    const int32_t* __restrict__ pos = Rd + kSchedClasses * kSchedCntStride + kSchedClasses * Bc;
    if (i < n) return L[i];
    int x = pos[t];
    if (x < 0) return t;
    for (int guard = 0; guard < n; guard++) {
        const int y = pos[x * kSchedClasses + k];
        if (y < 0) break;
        x = y;
    }
    return x * kSchedClasses + k;
}
// Ticket and note are WAVE-UNIFORM (round 4).  Round 3 took the ticket under `if (lane == 0)` -- a divergent region ahead of the QP
// loop, next to the place where hipcc's register allocator once put AGPR copies of live registers ahead of the exec restore of a
// join block (scripts/check_exec_restore.py).  Now the atomic is ONE inline-assembly block that narrows exec to lane 0 and restores
// it itself: the compiler sees straight-line code and builds no join block here.  The block waits for the returned value (an
// inline-asm result the compiler might otherwise copy before it has landed): one L2 round trip, ~1.5 us, per instance that runs
// the QP loop (>= 60 us).  Pointers are forced into SGPRs, the two stores of sched_note are issued by all lanes with identical
// address and data.  What round 4 learned about the defect itself: it is NOT tied to this region.  Taking the ticket at the end of
// the wave instead (BROV_SCHED_TICKET_LATE) moves the allocator's copies to the join block of a guarded store of the first-guess
// loop, 40 lines away -- any of the kernel's ~1000 divergent regions can host it when the allocation shifts, which is why the
// link rule runs the checker on every build and tests/test_kernel_resources.py keeps that statement order as a live canary.
__device__ __forceinline__ const int32_t* uniform_ptr(const int32_t* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const int32_t*)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ int wave_atomic_inc(int32_t* addr_uniform) {
    int ret = 0;
    const int zero = 0, one = 1;
    unsigned long long saved;
    asm volatile(
        "s_mov_b64 %[sv], exec\n\t"
        "s_mov_b64 exec, 1\n\t"
        "global_atomic_add %[r], %[off], %[one], %[base] sc0\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "s_mov_b64 exec, %[sv]"
        : [r] "+v"(ret), [sv] "=&s"(saved)
        : [off] "v"(zero), [one] "v"(one), [base] "s"(addr_uniform)
        : "memory");
    return __builtin_amdgcn_readfirstlane(ret);
}
// sched_ticket: where the instance enters the QP loop (wave-uniform result); sched_note: at the end of the wave, all lanes storing
// identical data to identical addresses
__device__ __forceinline__ int sched_ticket(const DevParams& P, int b) {
    if (!P.sched) return -1;
    int32_t* Wr = (int32_t*)uniform_ptr(P.sched + (size_t)P.sched_w * P.sched_stride);
    return wave_atomic_inc(Wr + (b & (kSchedClasses - 1)) * kSchedCntStride);
}
__device__ __forceinline__ void sched_note(const DevParams& P, int b, int p) {
    if (!P.sched) return;
    int32_t* Wr = (int32_t*)uniform_ptr(P.sched + (size_t)P.sched_w * P.sched_stride);
    const int Bc = sched_class_len(P.B), k = b & (kSchedClasses - 1);
    if (p >= 0 && p < Bc) Wr[kSchedClasses * kSchedCntStride + k * Bc + p] = b;
    Wr[kSchedClasses * kSchedCntStride + kSchedClasses * Bc + b] = p;
}
// did instance b run the QP loop in the previous solve?  (pos[b] of the buffer that solve wrote; all zero before the first solve: yes)
__device__ __forceinline__ bool sched_listed(const DevParams& P, int b) {
    if (!P.sched) return true;
    const int32_t* Rd = uniform_ptr(P.sched + (size_t)P.sched_r * P.sched_stride);
    return Rd[kSchedClasses * kSchedCntStride + kSchedClasses * sched_class_len(P.B) + b] >= 0;
}
__device__ __forceinline__ void sched_zero_next(const DevParams& P, int lane) {   // one wave of the launch
    if (P.sched && lane < kSchedClasses) P.sched[(size_t)P.sched_z * P.sched_stride + lane * kSchedCntStride] = 0;
}

// everything after the linearisation: QP solve, multiplier recovery, full step, result record.  lin_part / lin_nan carry this
// lane's share of the linearisation's KKT partials (max / NaN flag), reduced over the wave here.
// developer instrumentation: s_memtime stamps of the phase boundaries (P.dbg == nullptr in normal operation)
// slots 0 / 6 (first and last) also record the 100 MHz real-time counter, which -- unlike the per-XCD cycle counters -- is one
// clock for the whole device: second array, slots 7 (start) and 6 (end), and where the wave ran (slot 5: XCC_ID << 32 | HW_ID); scripts/dev/phase_stamps.py
// draws the launch timeline
#define DBG_STAMP(slot) do { if (P.dbg && lane == 0) {                                                                          \
        P.dbg[(size_t)b * 8 + (slot)] = __builtin_readcyclecounter();                                                           \
        if ((slot) == 0) P.dbg[(size_t)P.B * 8 + (size_t)b * 8 + 7] = __builtin_amdgcn_s_memrealtime();                        \
        if ((slot) == 0) P.dbg[(size_t)P.B * 8 + (size_t)b * 8 + 5] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492); \
        if ((slot) == 6) P.dbg[(size_t)P.B * 8 + (size_t)b * 8 + 6] = __builtin_amdgcn_s_memrealtime(); } } while (0)
// development build only (make EXTRA=-DBROV_DBG_IPM=1, scripts/dev/ipm_phases.py): cycle totals of the interior-point loop's
// phases in a second array, 8 slots per instance: init, element loops, factor sweep, forward, solve-only sweep, forward, iterations
#ifdef BROV_DBG_IPM
#define IPM_T(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); ipm_t[k] += t_ - ipm_last; ipm_last = t_; } while (0)
#else
#define IPM_T(k) do { } while (0)
#endif

// development build only (make EXTRA=-DBROV_DBG_LIN=1, scripts/dev/lin_phases.py): cycle split of the linearisation; the scheduling
// barriers keep the compiler from moving work across the stamps (which also makes this build slower than the product)
#ifdef BROV_DBG_LIN
#define LIN_T(k) do { __builtin_amdgcn_sched_barrier(0); lin_t[k] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define LIN_T(k) do { } while (0)
#endif

// LDS = 0 streaming kernels, 1 / 2 fused kernels (whole horizon resident; element arrays in LDS / registers: EL), 3 windowed
// kernel (sweeps on the resident window through the sw_* wrappers, element loops on the flat HBM arrays like LDS = 0; the
// step-0 factorisation has already run, fused with the linearisation: pre_ok)
// DF (fused kernel of the mailbox ticks, rti_fused_kernel_mail): an early exit sends its record BEFORE the adjoint sweep, as the resident windowed
// kernel does -- nothing in the record depends on the multipliers that sweep computes for the iterate
template <int LDS, class IT = Inst, bool DF = false>
__device__ __forceinline__ void qp_body(const DevParams& P, IT& I, int b, double lin_part, bool lin_nan, Win* W = nullptr,
                                        bool pre_ok = true, bool pre_illc = false) {
    constexpr bool EL = (LDS == 1 || LDS == 2);
    const double* __restrict__ cst = P.cst;
    const int lane = I.lane, N = I.NT, nv = I.nv;
    DBG_STAMP(1);
    const int rg = I.rg;

    double* x_it = P.x + (size_t)b * (N + 1) * 12;
    double* u_it = P.u + (size_t)b * N * 4;
    double* pi_it = P.pi + (size_t)b * N * 12;
    double* lam_it = P.lam + (size_t)b * N * 8;
    double* V = I.ipm + (size_t)IPM_V * nv;
    double* TL = I.ipm + (size_t)IPM_TL * nv;
    double* TU = I.ipm + (size_t)IPM_TU * nv;
    double* LL = I.ipm + (size_t)IPM_LL * nv;
    double* LU = I.ipm + (size_t)IPM_LU * nv;
    double* GAM = I.ipm + (size_t)IPM_GAM * nv;
    double* RT = I.ipm + (size_t)IPM_RT * nv;
    double* DVA = I.ipm + (size_t)IPM_DVA * nv;
    double* ACT = I.ipm + (size_t)IPM_ACT * nv;
    // where adjoint<> leaves the input gradient g: HBM array, or (fused path) the dead feed-forward array in LDS
    const double* GRAD = EL ? (const double*)I.kff : (const double*)DVA;
    // element accessors: LDS-typed on the fused path (a generic pointer into LDS compiles to flat loads / stores)
    auto rd_vhat = [&](int j) -> double { if constexpr (EL) return I.lds_vhat[j]; else return I.vhat[j]; };
    auto wr_vhat = [&](int j, double v) { if constexpr (EL) I.lds_vhat[j] = v; else I.vhat[j] = v; };
    auto rd_dxb = [&](int j) -> double { if constexpr (EL) return I.lds_dxb[j]; else return I.dxb[j]; };
    auto rd_grad = [&](int j) -> double { if constexpr (EL) return I.lds_kff[j]; else return GRAD[j]; };

    // d0 = x0 - x_0, row-replicated; KKT of the entering iterate = max(LIN partials, |d0|).  The six loads are requested here and
    // consumed after the step-0 backward sweep: waited for at once they are an exposed L2 / HBM round trip of the single wave
    double x0v[3], xiv[3];
    {
        const double* x0 = P.x0 + (size_t)b * 12;
#pragma unroll
        for (int r = 0; r < 3; r++) { x0v[r] = x0[rg + 4 * r]; xiv[r] = I.x[rg + 4 * r]; }
    }

    // ---- step 0: equality-constrained minimiser (Gamma = 0, rhs = r) ---------------------------------------
    // fused path: this lane's share of u (needed for the bound check right after the forward sweep) is requested now
    double ureg[2] = {0.0, 0.0};
    if constexpr (EL) {
#pragma unroll
        for (int t = 0; t < 2; t++)
            if (lane + 64 * t < nv) ureg[t] = I.u[lane + 64 * t];
    }
    int status = 0, iters = 0;
    double mu = 0.0, rho = 0.0;
    bool early = false, polished = false, use_vhat = false;
    int sched_p = -1;   // this instance's place in the next solve's list of expensive instances (work ordering; wave-uniform)
    bool ok = pre_ok;
    // partial refactorisation of the active-set tries (fused kernels, riccati_backward_partial): checkpoint stage = ceil(N / 4), off for
    // horizons too short to gain from it
    constexpr bool PART = EL || LDS >= 3;   // fused kernels and the windowed kernel's resident mode: stage checkpoint; windowed kernel: window-0 checkpoint
    bool illc0 = pre_illc;
    bool split0 = false;
#ifndef BROV_EXP_NO_SPLIT
    if constexpr (EL) split0 = I.ckpt > 0;   // set by the kernel body: only instances that ran the QP loop in the previous solve
#endif
    if constexpr (EL) { if (split0) {
        // the step-0 factor sweep in two parts with the checkpoint between them.  Measured: inside the stage loop a wave-uniform
        // `if (i == ckpt)` with the six stores costs the loop 7 % (registers and scheduling, taken or not); the split sweep still
        // costs 2 % (the software pipeline drains and refills once) + 1 % (the stores) -- so only the instances that are LIKELY to run
        // the QP loop pay it: those that ran it in the previous solve (the work ordering's own prediction, sched_listed).  Everybody
        // else runs the unsplit sweep below and, should it enter the loop after all, full factor sweeps as in round 3.
        BwdState S;
        wave_fence();
        bwd_init<true, LDS>(I, S);
        // ... out of ONE copy of the stage loop (a second inlined copy costs instruction-cache misses on every instance)
#pragma clang loop unroll(disable)
        for (int ph = 0; ph < 2; ph++) {
            if (ph == 1) {
                if (I.ckpt == 0) break;
                double* ck = I.Kt;     // the register images of P and p entering stage ckpt - 1: six coalesced 512-byte stores into the
#pragma unroll                         // (otherwise unused) K^T array of the streaming path, never waited for
                for (int r = 0; r < 3; r++) { ck[r * 64 + lane] = S.P[r]; ck[192 + r * 64 + lane] = S.pv[r]; }
            }
            // stages N-1 .. ckpt (all of them when ckpt = 0), then ckpt-1 .. 0
            bwd_chunk<true, LDS, false, true>(I, S, ph == 0 ? N : I.ckpt, ph == 0 ? I.ckpt : 0);
        }
        ok = S.ok;
        illc0 = S.illc;
    } }
    if constexpr (LDS < 3) { if (!split0) ok = riccati_backward<true, LDS, false, true>(I, &illc0); }
    d4 d0;
    double kkt = 0.0;
    {
#pragma unroll
        for (int r = 0; r < 3; r++) {
            d0[r] = x0v[r] - xiv[r];
            kkt_upd(kkt, d0[r]);  // NaN-poisoning max (lin_device.hpp)
        }
        d0[3] = 0.0;
        double part = lin_part;
        bool nanp = lin_nan;
        if (kkt != kkt) nanp = true;
        kkt = wave_max(fmax(part, (kkt != kkt) ? 0.0 : kkt));
        if (__ballot(nanp) != 0ull) kkt = __builtin_nan("");
    }
    // an ill-conditioned pivot block (kPivotRho): this instance repeats the sweep, and runs every later one, in the Cholesky form
    // (not in the two-waves-per-SIMD kernel of the short horizons, N <= 13: its 256 registers do not hold the second pivot form
    // without scratch, which the build forbids in a solver kernel)
    constexpr bool ROB = LDS != 2;
    bool robust = false, robust_ok = false;
    if constexpr (ROB) {
        // Only while the step is numerically meaningful (entering KKT <= 1e6, the bound of the parity rules): the iterate of a diverged
        // full-step SQP is ill-conditioned without end, and with pivots that never fail its interior-point loop grinds through all
        // qp_iter_max systems (measured: 50 instead of the 1..19 after which the fast form gives up or fails -- one such instance
        // made its whole launch 2.6 times as long).
        robust_ok = kkt <= 1e6;
        if ((__ballot(illc0) != 0ull && P.robust_pivot && robust_ok) || P.robust_pivot == 2) {   // (2: development knob, every instance)
            robust = true;
            I.ckpt = 0;   // (no partial refactorisation: the checkpoint belongs to the fast sweep)
            ok = sw_backward<true, LDS, true, true>(I, W);
        }
    }
    DBG_STAMP(2);
    // bounds of this lane's elements of the check below (element j = lane + 64 t belongs to input lane & 3): requested before
    // the forward sweep, not after it
    const double lbc = EL ? cst[32 + (lane & 3)] : 0.0, ubc = EL ? cst[36 + (lane & 3)] : 0.0;
    if (__ballot(!ok) != 0ull) {
        status = BROV_STATUS_QP_FAILURE;
    } else {
        sw_forward<LDS>(I, W, d0, cst);
        DBG_STAMP(3);
        bool feas = true;
        if constexpr (LDS >= 3) {
            feas = W->feas;   // checked window by window inside the sweep wrapper
        } else if constexpr (EL) {  // nv <= 92: two elements per lane, u already in registers
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int j = lane + 64 * t;
                if (j < nv) {
                    const double vj = rd_vhat(j), lb = lbc - ureg[t], ub = ubc - ureg[t];
                    if (!(vj >= lb && vj <= ub)) feas = false;
                }
            }
        } else {
            for (int j = lane; j < nv; j += 64) {
                const int m = j & 3;
                const double vj = rd_vhat(j), lb = cst[32 + m] - I.u[j], ub = cst[36 + m] - I.u[j];
                if (!(vj >= lb && vj <= ub)) feas = false;
            }
        }
        const bool allfeas = (__ballot(!feas) == 0ull);
        if (allfeas && P.early_exit) {
            early = true;  // the accepted inputs stay where the forward sweep left them (I.vhat)
        } else {
            // Active-set tries and interior-point iterations (the oracle's schedule, bluerov2_oracle.c "ACTIVE-SET POLISH"): a round of
            // equality-constrained solves with the guessed active inputs pinned at their bounds -- first from the inputs the
            // Newton point violates, later from the interior-point iterate's classification --, each checked for the two
            // conditions that make it THE minimiser (free inputs inside the box, multipliers of pinned inputs of the right sign)
            // and repaired the primal-dual active-set way if not; interior-point iterations in between as the globally
            // convergent fallback.  iters counts Newton systems (tries + iterations).
            // Fused path: the interior-point vectors (two elements per lane, nv <= 92) live in registers -- at one wave per
            // SIMD every element loop over HBM-resident vectors costs an exposed L2 round trip; only Gamma and the right-hand
            // side, which the backward sweep reads by stage, go through memory.
#ifdef BROV_DBG_IPM
            unsigned long long ipm_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ipm_last = __builtin_readcyclecounter();
#endif
            constexpr bool CACHE = (LDS >= 3);   // windowed kernel: register copies per loop group (IpmVec MODE 2)
            constexpr int kIpmT = EL ? 2 : 8;    // elements per lane; streaming / windowed path: nv <= 512
            using Vec = IpmVec<EL ? 1 : (CACHE ? 2 : 0), kIpmT>;
            Vec vV{{}, V}, vTL{{}, TL}, vTU{{}, TU}, vLL{{}, LL}, vLU{{}, LU}, vDVA{{}, DVA},
                vDLL{{}, GAM}, vDLU{{}, RT},   // dual steps: registers, or (streaming) parked in GAM / RT, both rebuilt every iteration
                vACT{{}, ACT};                 // active-set guess: -1 / +1 pinned at the lower / upper bound, 0 free
#define IPM_FOR(t, j) _Pragma("unroll") for (int t = 0; t < kIpmT; t++) if (const unsigned j = (unsigned)lane + 64u * t; j < (unsigned)nv)
            // MODE 2: the group's other operands (inputs, references, Newton point ...), requested with the fetches
#define IPM_PRE(arr, expr)                                                                     \
            double arr[CACHE ? kIpmT : 1];                                                         \
            if constexpr (CACHE) {                                                                 \
                _Pragma("unroll") for (int t = 0; t < kIpmT; t++) {                                \
                    const unsigned j = ((unsigned)lane + 64u * t < (unsigned)nv) ? (unsigned)lane + 64u * t : 0u; \
                    arr[t] = (expr);                                                               \
                }                                                                                  \
            }
            // windowed kernel: the guess is stored element by element where it is produced (its register copy would be 16 more VGPRs
            // across loops that have none to spare); the loads of a group are all ahead of its first store anyway
            auto set_act = [&](int t, int j, double v) __attribute__((always_inline)) { if constexpr (CACHE) ACT[j] = v; else vACT.set(t, j, v); };
            // Every element group works on its own opaque copy of the lane index.  Element addresses are then formed where they are
            // used (base pointer in SGPRs + 32-bit offset: one addressing mode); computed from the kernel's lane index they are loop
            // invariants, and the compiler hoists one 64-bit address per element and vector out of the loop -- 160 VGPRs live across
            // every sweep of the windowed kernel, which then spills into scratch
#define GROUP_LANE int lane_g_ = I.lane; asm volatile("v_mov_b32 %0, %0" : "+v"(lane_g_)); const int lane = lane_g_
            const int mI = lane & 3;   // input index of every element of this lane (j = lane + 64 t)
            // bounds and weight of that input: loaded once and made opaque, so that the compiler cannot sink the (re-)loads into
            // the guarded element blocks below, where every one of them would be waited for under the exec mask
            double lbI = cst[32 + mI], ubI = cst[36 + mI], wuI = cst[12 + mI];
            asm volatile("" : "+v"(lbI), "+v"(ubI), "+v"(wuI));
            const double rdI = P.Ts * wuI;   // the input's own Hessian entry
            // ... per stage on the streaming kernel's general grid (time steps / stage-0 weight differ from stage to stage)
            auto rd_el = [&](unsigned j) __attribute__((always_inline)) -> double {
                if constexpr (LDS == 0) return I.wst ? I.wst[(size_t)(j >> 2) * 16 + 12 + mI] : rdI;
                else if constexpr (IT::kGrid) return I.wst[(size_t)(j >> 2) * 16 + 12 + mI];
                else return rdI;
            };
            {   // first guess: the inputs of the Newton point that violate their bounds
                GROUP_LANE;
                IPM_PRE(up, I.u[j]);
                IPM_PRE(vh, I.vhat[j]);
                IPM_FOR(t, j) {
                    const double uj = EL ? ureg[t & 1] : (CACHE ? up[CACHE ? t : 0] : I.u[j]);
                    const double vj = CACHE ? vh[CACHE ? t : 0] : rd_vhat(j);
                    set_act(t, j, vj < lbI - uj ? -1.0 : (vj > ubI - uj ? 1.0 : 0.0));
                }
            }
            status = BROV_STATUS_MAXITER;
            // partial refactorisation: the feed-forward terms of the step-0 sweep, before the first adjoint sweep overwrites them
            double kff0[2] = {0.0, 0.0};
            bool hi_step0 = false;   // gains / feed-forward terms of the stages >= ckpt in LDS are the step-0 ones
            if constexpr (PART) hi_step0 = I.ckpt > 0;
            if constexpr (EL) {
#pragma unroll
                for (int t = 0; t < 2; t++) kff0[t] = I.lds_kff[lane + 64 * t < nv ? lane + 64 * t : 0];
            }
            if constexpr (LDS == 4) {   // resident mode: nv <= 324 elements, kept behind the checkpoint in HBM (sw_backward restores them)
                if (I.ckpt > 0) {
                    double* kf = I.Kt + 384;
                    for (int j = lane; j < nv; j += 64) kf[j] = I.lds_kff[j];
                }
            }
            // this instance runs the QP loop: first in line in the next solve.  (BROV_SCHED_TICKET_LATE: the ticket at the end of the wave
            // instead -- the statement order that makes hipcc 7.2 build the exec-restore defect into rti_window_kernel, at a join block of
            // the first-guess stores above; kept as the canary of tests/test_kernel_resources.py: the build gate must reject it.)
#ifndef BROV_SCHED_TICKET_LATE
            sched_p = sched_ticket(P, b);
#else
            sched_p = -2;
#endif
            const double inv2nv = 1.0 / (2.0 * nv);
            int round_k = 0, round_cap = POL_FIRST, nchg_prev = nv + 1;
            double mu_gate = 1e300;
            bool ipm_on = false, converged = false;
            IPM_T(0);
            iters = 0;
            double gam_r[2] = {0.0, 0.0};
            while (iters < P.qp_iter_max) {
                // One Newton system per trip: an active-set try (the guessed active inputs pinned) while a round is on, else an
                // interior-point iteration.  Both factorise and solve through the same pair of sweeps.
                const bool try_mode = round_k < round_cap;
                // nothing lane-dependent may be hoisted out of this loop: the sweeps' per-lane addresses, computed once ahead of the
                // loop, would all be live across all of its sweeps (the register file is full: the kernels then spill into scratch)
                asm volatile("v_mov_b32 %0, %0" : "+v"(I.lane));
                I.rg = I.lane >> 4; I.cl = I.lane & 15;
                if (!try_mode && !ipm_on) {
                    // interior start at the last active-set point: clamp into the box, multipliers from mu0 = stationarity
                    // residual of the clamped point
                    ipm_on = true;
                    {
                        GROUP_LANE;
                        IPM_PRE(up, I.u[j]);
                        IPM_PRE(vh, I.vhat[j]);
                        IPM_FOR(t, j) {
                            const double uj = EL ? ureg[t & 1] : (CACHE ? up[CACHE ? t : 0] : I.u[j]);
                            const double lb = lbI - uj, ub = ubI - uj;
                            const double wdt = ub - lb;
                            double vj = CACHE ? vh[CACHE ? t : 0] : rd_vhat(j);
                            const double lo = lb + IPM_TAU0 * wdt, hi = ub - IPM_TAU0 * wdt;
                            vj = (vj < lo) ? lo : vj;
                            vj = (vj > hi) ? hi : vj;
                            vV.set(t, j, vj); vTL.set(t, j, vj - lb); vTU.set(t, j, ub - vj);
                            if constexpr (EL) wr_vhat(j, vj);  // roll-out / adjoint read their inputs from the LDS copy
                        }
                        vV.flush(lane, nv); vTL.flush(lane, nv); vTU.flush(lane, nv);
                    }
                    sw_rollout<LDS>(I, W, d0, V);
                    sw_adjoint<false, LDS>(I, W, V, DVA, nullptr);
                    {
                        GROUP_LANE;
                        vTL.fetch(lane, nv); vTU.fetch(lane, nv);
                        IPM_PRE(gr, GRAD[j]);
                        double g0 = 0.0;
                        IPM_FOR(t, j) g0 = fmax(g0, fabs(CACHE ? gr[CACHE ? t : 0] : rd_grad(j)));
                        g0 = wave_max(g0);
                        const double mu0 = fmax(IPM_MU0F * g0, 1e-4);
                        double r0 = 0.0;
                        IPM_FOR(t, j) {
                            const double ll = mu0 / vTL.get(t, j), lu = mu0 / vTU.get(t, j);
                            vLL.set(t, j, ll); vLU.set(t, j, lu);
                            r0 = fmax(r0, fabs((CACHE ? gr[CACHE ? t : 0] : rd_grad(j)) - ll + lu));
                        }
                        rho = wave_max(r0);
                        vLL.flush(lane, nv); vLU.flush(lane, nv);
                    }
                    IPM_T(0);
                }
                iters++;
                double s = 0.0;
                bool part = false;   // this Newton system restarts its factor sweep from the step-0 checkpoint
                if (try_mode) {   // pin: Gamma = POL_BIG and a right-hand side that lands the input on its bound
                    round_k++;
                    bool deep = false;   // a pinned input at a stage >= ckpt
                    {
                        GROUP_LANE;
                        vACT.fetch(lane, nv);
                        IPM_PRE(up, I.u[j]);
                        IPM_PRE(yr, I.yref[(size_t)(j >> 2) * 16 + 12 + (j & 3)]);
                        IPM_FOR(t, j) {
                            const double uj = EL ? ureg[t & 1] : (CACHE ? up[CACHE ? t : 0] : I.u[j]);
                            const double rr = EL ? (double)I.lds_r[j]
                                                 : rd_el(j) * (CACHE ? up[CACHE ? t : 0] - yr[CACHE ? t : 0]
                                                                               : I.u[j] - I.yref[(size_t)(j >> 2) * 16 + 12 + mI]);
                            const double ac = vACT.get(t, j);
                            const double gm = ac != 0.0 ? POL_BIG : 0.0;
                            GAM[j] = gm;
                            RT[j] = rr - gm * ((ac < 0.0 ? lbI : ubI) - uj);
                            if constexpr (PART) deep = deep | ((ac != 0.0) & ((int)j >= 4 * I.ckpt));
                        }
                    }
                    if constexpr (PART) part = hi_step0 && __ballot(deep) == 0ull;
                } else {   // group A of an interior-point iteration: Gamma and the predictor's right-hand side
                    {   // group A: Gamma and the predictor's right-hand side
                        GROUP_LANE;
                        vLL.fetch(lane, nv); vLU.fetch(lane, nv); vTL.fetch(lane, nv); vTU.fetch(lane, nv); vV.fetch(lane, nv);
                        IPM_PRE(up, I.u[j]);
                        IPM_PRE(yr, I.yref[(size_t)(j >> 2) * 16 + 12 + (j & 3)]);
                        IPM_FOR(t, j) {
                            const double ll = vLL.get(t, j), lu = vLU.get(t, j), tl = vTL.get(t, j), tu = vTU.get(t, j);
                            s += ll * tl + lu * tu;
                            const double gm = ll / tl + lu / tu;
                            GAM[j] = gm;
                            if constexpr (EL) gam_r[t & 1] = gm;
                            const double rr = EL ? (double)I.lds_r[j]
                                                 : rd_el(j) * (CACHE ? up[CACHE ? t : 0] - yr[CACHE ? t : 0]
                                                                               : I.u[j] - I.yref[(size_t)(j >> 2) * 16 + 12 + mI]);
                            RT[j] = rr - gm * vV.get(t, j);
                        }
                    }
                    mu = wave_sum(s) * inv2nv;
                }
                IPM_T(1);
                if (!robust) {
                    bool ill = false;
                    if constexpr (EL) {
                        ok = riccati_backward_tries<LDS>(I, part, kff0, ill);
                        if (!part) hi_step0 = false;   // a full sweep: the LDS gains are no longer step 0's
                    } else if constexpr (LDS >= 3) {
                        ok = sw_backward<true, LDS>(I, W, &ill, part);
                        if (!part) hi_step0 = false;   // ... the parked gains of the windows >= 1 (resident mode: the LDS gains of the stages >= ckpt)
                    } else {
                        ok = sw_backward<true, LDS>(I, W, &ill);   // (streaming kernel)
                    }
                    if constexpr (ROB) { if (__ballot(ill) != 0ull && P.robust_pivot && robust_ok) { robust = true; hi_step0 = false; } }   // ... and this system is factorised again
                }
                if constexpr (ROB) { if (robust) ok = sw_backward<true, LDS, false, true>(I, W); }
                IPM_T(2);
                if (__ballot(!ok) != 0ull) { status = BROV_STATUS_QP_FAILURE; break; }
                sw_forward<LDS>(I, W, d0);
                IPM_T(3);
                if (try_mode) {
                    bool bad = false;
                    {   // pinned inputs exactly onto their bounds; free inputs that leave the box are marked (+-2: to be pinned)
                        GROUP_LANE;
                        vACT.fetch(lane, nv);
                        IPM_PRE(up, I.u[j]);
                        IPM_PRE(vh, I.vhat[j]);
                        IPM_FOR(t, j) {
                            const double uj = EL ? ureg[t & 1] : (CACHE ? up[CACHE ? t : 0] : I.u[j]);
                            const double lb = lbI - uj, ub = ubI - uj;
                            double vj = CACHE ? vh[CACHE ? t : 0] : rd_vhat(j);
                            double ac = vACT.get(t, j);
                            if (!(vj == vj)) bad = true;
                            if (ac != 0.0) vj = ac < 0.0 ? lb : ub;
                            else ac = vj < lb ? -2.0 : (vj > ub ? 2.0 : 0.0);
                            set_act(t, j, ac);
                            if constexpr (EL) wr_vhat(j, vj); else I.vhat[j] = vj;
                        }
                            }
                    if constexpr (LDS >= 3) bad = bad || W->nan;
                    if (__ballot(bad) != 0ull) { status = BROV_STATUS_NAN; break; }
                    // multipliers of this point: the state steps of the forward sweep are its roll-out (the snap of a pinned input
                    // is a rounding error), so the adjoint recursion alone gives g = R v + r + B'pi
                    sw_adjoint<false, LDS>(I, W, I.vhat, DVA, nullptr);
                    IPM_T(4);
                    int nchg;
                    {
                        GROUP_LANE;
                        vACT.fetch(lane, nv);
                        IPM_PRE(gr, GRAD[j]);
                        double gmx = 0.0;
                        IPM_FOR(t, j) gmx = fmax(gmx, fabs(CACHE ? gr[CACHE ? t : 0] : rd_grad(j)));
                        gmx = wave_max(gmx);
                        double cnt = 0.0;
                        IPM_FOR(t, j) {
                            const double g = CACHE ? gr[CACHE ? t : 0] : rd_grad(j);
                            double ac = vACT.get(t, j);
                            const double tolg = POL_TOL_G * rd_el(j) + POL_TOL_GREL * gmx;
                            if (ac == 2.0 || ac == -2.0) { ac *= 0.5; cnt += 1.0; }                       // newly pinned
                            else if ((ac < 0.0 && g < -tolg) || (ac > 0.0 && g > tolg)) { ac = 0.0; cnt += 1.0; }   // released
                            set_act(t, j, ac);
                        }
                                nchg = (int)wave_sum(cnt);
                    }
                    IPM_T(1);
                    if (nchg == 0) { polished = true; status = BROV_STATUS_SUCCESS; break; }
                    // the round goes on while the repairs are few and do not grow (a guess that is converging)
                    // (the first round is the patient one: see the oracle)
                    if (nchg > POL_NCHG || (nchg > nchg_prev && ipm_on)) round_cap = 0;
                    nchg_prev = nchg;
                    if (round_k >= round_cap) {   // failed round: the next one waits until the interior-point loop has halved mu
                        if (ipm_on) mu_gate = mu;
                        if (converged) break;
                    }
                    continue;
                }
                // ---- the rest of the interior-point iteration (Mehrotra predictor-corrector)
                double smu;
                {   // group B: predictor step length, centering, corrector right-hand side
                    GROUP_LANE;
                    vLL.fetch(lane, nv); vLU.fetch(lane, nv); vTL.fetch(lane, nv); vTU.fetch(lane, nv); vV.fetch(lane, nv);
                    IPM_PRE(vh, I.vhat[j]);
                    IPM_PRE(up, I.u[j]);
                    IPM_PRE(yr, I.yref[(size_t)(j >> 2) * 16 + 12 + (j & 3)]);
                    IPM_PRE(gmp, GAM[j]);
                    double aaff = 1.0;
                    IPM_FOR(t, j) {
                        const double ll = vLL.get(t, j), lu = vLU.get(t, j), tl = vTL.get(t, j), tu = vTU.get(t, j);
                        const double dv = (CACHE ? vh[CACHE ? t : 0] : rd_vhat(j)) - vV.get(t, j);
                        vDVA.set(t, j, dv);
                        const double dll = -ll - ll / tl * dv, dlu = -lu + lu / tu * dv;
                        if (dv < 0) aaff = fmin(aaff, -tl / dv);
                        if (dv > 0) aaff = fmin(aaff, tu / dv);
                        if (dll < 0) aaff = fmin(aaff, -ll / dll);
                        if (dlu < 0) aaff = fmin(aaff, -lu / dlu);
                    }
                    aaff = wave_min(aaff);
                    double sa = 0.0;
                    IPM_FOR(t, j) {
                        const double ll = vLL.get(t, j), lu = vLU.get(t, j), tl = vTL.get(t, j), tu = vTU.get(t, j), dv = vDVA.get(t, j);
                        const double dll = -ll - ll / tl * dv, dlu = -lu + lu / tu * dv;
                        sa += (ll + aaff * dll) * (tl + aaff * dv) + (lu + aaff * dlu) * (tu - aaff * dv);
                    }
                    const double muaff = wave_sum(sa) * inv2nv;
                    double sigma = muaff / mu;
                    sigma = sigma * sigma * sigma;
                    smu = sigma * mu;
                    IPM_FOR(t, j) {
                        const double ll = vLL.get(t, j), lu = vLU.get(t, j), tl = vTL.get(t, j), tu = vTU.get(t, j), dv = vDVA.get(t, j);
                        const double dll = -ll - ll / tl * dv, dlu = -lu + lu / tu * dv;
                        const double cl_ = dll * dv, cu_ = -dlu * dv;
                        const double rr = EL ? (double)I.lds_r[j]
                                             : rd_el(j) * (CACHE ? up[CACHE ? t : 0] - yr[CACHE ? t : 0]
                                                                           : I.u[j] - I.yref[(size_t)(j >> 2) * 16 + 12 + mI]);
                        const double gm = EL ? gam_r[t & 1] : (CACHE ? gmp[CACHE ? t : 0] : GAM[j]);
                        RT[j] = rr - gm * vV.get(t, j) - (smu - cl_) / tl + (smu - cu_) / tu;
                    }
                    vDVA.flush(lane, nv);
                }
                IPM_T(1);
                (void)sw_backward<false, LDS>(I, W);
                IPM_T(4);
                sw_forward<LDS>(I, W, d0);
                IPM_T(5);
                bool bad = false;
                double s2 = 0.0, alpha, unres = 0.0;
                {   // group C: step length of the combined direction, update, classification of the bounds
                    GROUP_LANE;
                    vLL.fetch(lane, nv); vLU.fetch(lane, nv); vTL.fetch(lane, nv); vTU.fetch(lane, nv); vV.fetch(lane, nv);
                    vDVA.fetch(lane, nv);
                    IPM_PRE(vh, I.vhat[j]);
                    double amax = 1e300;
                    IPM_FOR(t, j) {
                        const double ll = vLL.get(t, j), lu = vLU.get(t, j), tl = vTL.get(t, j), tu = vTU.get(t, j), dva = vDVA.get(t, j);
                        const double dlla = -ll - ll / tl * dva, dlua = -lu + lu / tu * dva;
                        const double cl_ = dlla * dva, cu_ = -dlua * dva;
                        const double dv = (CACHE ? vh[CACHE ? t : 0] : rd_vhat(j)) - vV.get(t, j);
                        const double dll = (smu - cl_) / tl - ll - ll / tl * dv;
                        const double dlu = (smu - cu_) / tu - lu + lu / tu * dv;
                        if (dv < 0) amax = fmin(amax, -tl / dv);
                        if (dv > 0) amax = fmin(amax, tu / dv);
                        if (dll < 0) amax = fmin(amax, -ll / dll);
                        if (dlu < 0) amax = fmin(amax, -lu / dlu);
                        if constexpr (!CACHE) { vDLL.set(t, j, dll); vDLU.set(t, j, dlu); }
                    }
                    amax = wave_min(amax);
                    {   // a blocked step stops 10 % short of the boundary, a (nearly) full one goes 99.99 % of the way
                        const double a = amax < 1.0 ? amax : 1.0;
                        alpha = (IPM_FTB * amax >= 1.0) ? 1.0 : a * ((1.0 - a) * IPM_FTBLO + a * IPM_FTB);
                    }
                    IPM_FOR(t, j) {
                        const double dv = (CACHE ? vh[CACHE ? t : 0] : rd_vhat(j)) - vV.get(t, j);
                        const double vj = vV.get(t, j) + alpha * dv;
                        const double tl = vTL.get(t, j) + alpha * dv, tu = vTU.get(t, j) - alpha * dv;
                        double dll, dlu;
                        if constexpr (CACHE) {
                            // windowed kernel: the dual steps are recomputed (a dozen operations per element) instead of held in 32
                            // more registers across the reduction -- the kernel has none to spare
                            const double l0 = vLL.get(t, j), u0_ = vLU.get(t, j), t0_ = vTL.get(t, j), t1_ = vTU.get(t, j), dva = vDVA.get(t, j);
                            const double dlla = -l0 - l0 / t0_ * dva, dlua = -u0_ + u0_ / t1_ * dva;
                            dll = (smu - dlla * dva) / t0_ - l0 - l0 / t0_ * dv;
                            dlu = (smu + dlua * dva) / t1_ - u0_ + u0_ / t1_ * dv;
                        } else {
                            dll = vDLL.get(t, j); dlu = vDLU.get(t, j);
                        }
                        const double ll = vLL.get(t, j) + alpha * dll, lu = vLU.get(t, j) + alpha * dlu;
                        vV.set(t, j, vj); vTL.set(t, j, tl); vTU.set(t, j, tu); vLL.set(t, j, ll); vLU.set(t, j, lu);
                        if (!(vj == vj)) bad = true;
                        s2 += ll * tl + lu * tu;
                        // how far this element's bounds are from resolved: min(distance to the bound, multiplier / input weight);
                        // the same two quantities classify the bound for the next active-set round (active <=> the multiplier
                        // could move the input further than it is away from the bound)
                        const double rde = rd_el(j), al = ll / rde, au = lu / rde;
                        unres = fmax(unres, fmax(fmin(tl, al), fmin(tu, au)));
                        set_act(t, j, al > tl ? -1.0 : (au > tu ? 1.0 : 0.0));
                    }
                    vV.flush(lane, nv); vTL.flush(lane, nv); vTU.flush(lane, nv); vLL.flush(lane, nv); vLU.flush(lane, nv);
                    }
                if (__ballot(bad) != 0ull) { status = BROV_STATUS_NAN; break; }
                rho *= (1.0 - alpha);
                mu = wave_sum(s2) * inv2nv;
                IPM_T(1);
                // the loop's own rule (same as the oracle, bluerov2_oracle.c): every bound resolved to tol_mu -- the input within that
                // distance of it, or its multiplier too small to move the input that far -- and the tracked stationarity residual
                // below tol_stat.  Then one more active-set round for the exact answer; if that fails too the iterate is the answer.
                unres = wave_max(unres);
                if (unres <= P.tol_mu && rho <= P.tol_stat) converged = true;
                if (converged || (mu <= POL_MU_GATE * mu_gate && alpha >= POL_ALPHA_GATE)) { round_k = 0; round_cap = POL_LOOP; nchg_prev = nv + 1; }
            }
            if (converged && status == BROV_STATUS_MAXITER) status = BROV_STATUS_SUCCESS;
#ifdef BROV_DBG_IPM
            if (P.dbg && lane == 0) {
                ipm_t[6] = iters;
                for (int k = 0; k < 7; k++) P.dbg[(size_t)P.B * 8 + (size_t)b * 8 + k] = ipm_t[k];
            }
#endif
            // the final inputs go where the finalisation expects them.  Polished: the LDS copy / I.vhat holds them (with their state
            // steps and, on the fused path, their multipliers).  Otherwise the interior-point iterate: V (streaming / windowed path) /
            // the LDS copy; or, when the limit was reached before the first interior-point iteration, the last active-set point
            // clamped into the box.
            use_vhat = polished || !ipm_on;
            if (!polished) {
                if (ipm_on) {
                    if constexpr (EL) { IPM_FOR(t, j) wr_vhat(j, vV.get(t, j)); }
                } else if (status == BROV_STATUS_MAXITER) {
                    GROUP_LANE;
                    IPM_PRE(up, I.u[j]);
                    IPM_PRE(vh, I.vhat[j]);
                    IPM_FOR(t, j) {
                        const double uj = EL ? ureg[t & 1] : (CACHE ? up[CACHE ? t : 0] : I.u[j]);
                        double vj = CACHE ? vh[CACHE ? t : 0] : rd_vhat(j);
                        vj = fmin(fmax(vj, lbI - uj), ubI - uj);
                        if constexpr (EL) wr_vhat(j, vj); else I.vhat[j] = vj;
                    }
                }
            }
#undef IPM_FOR
#undef IPM_PRE
#undef GROUP_LANE
        }
    }

    // The result record (device copy and, for brov_tick_host at small batches, the host mailbox).  A lambda because the resident
    // windowed kernel sends it BEFORE its last adjoint sweep when the answer is the equality-constrained one: nothing in the record
    // depends on the multipliers that sweep computes for the iterate, and the host gets its input ~20 us earlier at N = 80.
    bool emitted = false;
    auto emit_record = [&](double cost_lane, double u0_lane, bool have_u0) __attribute__((always_inline)) {
        const double cs = wave_sum(cost_lane);
        if (lane == 0) {
            brov_result* r = P.res + b;
            r->cost = cs;
            r->kkt = kkt;
            r->status = status;
            r->qp_iter = early ? 0 : iters;
        }
        // first input of the record.  Failed step: the last successfully computed input is held (clamped into the box, NaN -> 0),
        // so that the plant / thrust consumers never see a diverged iterate's input.
        double u0r = u0_lane;
        if (!have_u0 && lane < 4) {
            u0r = P.res[b].u0[lane];
            u0r = (u0r == u0r) ? u0r : 0.0;
            u0r = fmin(fmax(u0r, cst[32 + lane]), cst[36 + lane]);
            P.res[b].u0[lane] = u0r;
        }
        // thrust allocation epilogue (bluerov2_dob.cpp:390-395), six lanes
        const double a0 = readlane_f64(u0r, 0), a1 = readlane_f64(u0r, 1), a2 = readlane_f64(u0r, 2), a3 = readlane_f64(u0r, 3);
        const double s0 = (lane == 0 || lane == 1) ? -a0 : a0;
        const double s1 = (lane == 0 || lane == 2) ? a1 : -a1;
        const double s3 = (lane == 0 || lane == 3) ? a3 : -a3;
        const double th = ((lane < 4) ? (s0 + s1) + s3 : -a2) / kRotor;   // same operation order as the host helper
        if (lane < 6) P.res[b].thrust[lane] = th;
        if (P.mail) {   // host mailbox: the same record into pinned host memory, then (behind a system-scope fence) the sequence word
            brov_result* m = P.mail + b;
            if (lane < 4) m->u0[lane] = u0r;
            if (lane < 6) m->thrust[lane] = th;
            if (lane == 0) { m->cost = cs; m->kkt = kkt; m->status = status; m->qp_iter = early ? 0 : iters; }
            if (P.mail_flag) {   // (no sequence words: a large batch, the host waits for the launch)
                __threadfence_system();
                if (lane == 0) __hip_atomic_store(P.mail_flag + b, P.mail_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        emitted = true;
    };

    // ---- finalise: consistent primal/dual for the final inputs, multiplier recovery, full step ---------------
    // Element loops issue all their loads before the first use (UX/UU elements per lane per chunk): at one wave per SIMD
    // every dependent global round trip is otherwise fully exposed (~2 us each).
    constexpr int UX = EL ? 5 : 4, UU = EL ? 2 : 4;
    const double* vfin = (early || EL || use_vhat) ? I.vhat : V;   // fused path: the interior-point loop leaves its inputs in the LDS copy
    const int nxe = (N + 1) * 12;
    double cost = 0.0;
    bool wrote_u0 = false;
    double u0v = 0.0;   // lanes 0..3: first input of the result record
    if (status == BROV_STATUS_SUCCESS || status == BROV_STATUS_MAXITER) {
        if (!early && !polished) {  // early exit / accepted active-set point: dxb already holds its state steps
            sw_rollout<LDS>(I, W, d0, vfin);
        }
        DBG_STAMP(4);
        // fused path: the iterate and the reference of the commit loops below are requested before the adjoint sweep, which
        // hides their round trip (the single resident wave has nothing else to switch to)
        if constexpr (LDS >= 3) {
            if (W->nan) {
                status = BROV_STATUS_NAN;
            } else {
                win_adjoint_commit<LDS == 4>(P, I, *W, b, vfin, early, cost, u0v, P.mail != nullptr && P.mail_early != 0,
                                             [&](double cost_lane, double u0_lane) __attribute__((always_inline)) { emit_record(cost_lane, u0_lane, true); });
                wrote_u0 = true;
            }
            DBG_STAMP(5);
        } else {
        double xpre[UX], ypre[UX], urpre[UU];
        // one-wave fused kernel: the cost weights of those elements too (the two-wave variant has no registers for them)
        double wxpre[LDS == 1 ? UX : 1], wupre[LDS == 1 ? UU : 1];
        if constexpr (EL) {
#pragma unroll
            for (int t = 0; t < UX; t++) {
                const int j = lane + 64 * t;
                const int jj = j < nxe ? j : 0;
                const int i = jj / 12, c = jj - i * 12;
                xpre[t] = x_it[jj];
                ypre[t] = I.yref[(size_t)i * 16 + c];
                if constexpr (LDS == 1) wxpre[t] = cst[(i == N) ? 16 + c : c];
            }
#pragma unroll
            for (int t = 0; t < UU; t++) {
                const int j = lane + 64 * t;
                const int jj = j < nv ? j : 0;
                urpre[t] = I.yref[(size_t)(jj >> 2) * 16 + 12 + (jj & 3)];
                if constexpr (LDS == 1) wupre[t] = cst[12 + (jj & 3)];
            }
        }
        // fused kernels, accepted active-set point: the try's own adjoint sweep has left multipliers and input gradient in LDS
        auto copy_pi = [&]() __attribute__((always_inline)) {
            // N * 12 <= 276 elements: five per lane, read back to back, then stored (a guarded copy loop waits for LDS once per element)
            double pv5[5];
#pragma unroll
            for (int t = 0; t < 5; t++) pv5[t] = I.lds_kt[lane + 64 * t < N * 12 ? lane + 64 * t : 0];
#pragma unroll
            for (int t = 0; t < 5; t++) asm volatile("" : "+v"(pv5[t]));
#pragma unroll
            for (int t = 0; t < 5; t++)
                if (lane + 64 * t < N * 12) pi_it[lane + 64 * t] = pv5[t];
        };
        const bool late = DF && EL && early && P.mail != nullptr;   // (constant false outside the mailbox kernel)
        if (!(EL && polished) && !late) sw_adjoint<true, LDS>(I, W, vfin, DVA, pi_it);
        DBG_STAMP(5);
        bool nanv = false;
        // fused kernels: the lane's elements of the accepted inputs and state steps (all of them: nv <= 128, nxe <= 320) are read
        // once, back to back with clamped indices, checked here and reused by the update loops below -- written as guarded
        // element loops every read sits in its own exec-masked block with an LDS wait inside
        double vvp[EL ? UU : 1], djp[EL ? UX : 1];
        if constexpr (EL) {
#pragma unroll
            for (int t = 0; t < UU; t++) vvp[t] = rd_vhat(lane + 64 * t < nv ? lane + 64 * t : 0);
#pragma unroll
            for (int t = 0; t < UX; t++) djp[t] = rd_dxb(lane + 64 * t < nxe ? lane + 64 * t : 0);
#pragma unroll
            for (int t = 0; t < UU; t++) nanv = nanv || !(vvp[t] == vvp[t]);   // clamped slots repeat element 0: same verdict
#pragma unroll
            for (int t = 0; t < UX; t++) nanv = nanv || !(djp[t] == djp[t]);
        } else {
            for (int j = lane; j < nv; j += 64) {
                const double vj = vfin[j];
                if (!(vj == vj)) nanv = true;
            }
            for (int j = lane; j < nxe; j += 64) {
                const double dj = rd_dxb(j);
                if (!(dj == dj)) nanv = true;
            }
        }
        if (__ballot(nanv) != 0ull) {
            status = BROV_STATUS_NAN;
        } else {
            for (int j0 = lane; j0 < nv; j0 += 64 * UU) {
                double uo[UU], vv[UU], gg[UU], ur[UU];
#pragma unroll
                for (int t = 0; t < UU; t++) {
                    const int j = j0 + 64 * t;
                    const bool in = j < nv;
                    const int jj = in ? j : 0;
                    uo[t] = (EL && j0 == lane && t < 2) ? ureg[t] : u_it[jj];
                    vv[t] = EL ? ((j0 == lane) ? vvp[EL ? t : 0] : rd_vhat(jj)) : vfin[jj];
                    gg[t] = early ? 0.0 : rd_grad(jj);
                    ur[t] = (EL && j0 == lane) ? urpre[t] : I.yref[(size_t)(jj >> 2) * 16 + 12 + (jj & 3)];
                }
#pragma unroll
                for (int t = 0; t < UU; t++) {
                    const int j = j0 + 64 * t;
                    if (j < nv) {
                        const int i = j >> 2, m = j & 3;
                        lam_it[i * 8 + m] = gg[t] > 0 ? gg[t] : 0.0;
                        lam_it[i * 8 + 4 + m] = gg[t] < 0 ? -gg[t] : 0.0;
                        const double un = uo[t] + vv[t];
                        u_it[j] = un;
                        if (j < 4) { P.res[b].u0[j] = un; u0v = un; }
                        const double e = un - ur[t];
                        const double wgt = (LDS == 1 && j0 == lane) ? wupre[LDS == 1 ? t : 0] : cst[12 + m];
                        const double sw = (IT::kGrid || (LDS == 0 && I.wst)) ? I.wst[(size_t)i * 16 + 12 + m] : P.Ts * wgt;
                        cost += 0.5 * sw * e * e;
                    }
                }
            }
            wrote_u0 = true;
            if constexpr (EL) {  // multipliers staged in LDS by adjoint<>: [N][12] at the head of the K^T array
                if (!late) copy_pi();
            }
            for (int j0 = lane; j0 < nxe; j0 += 64 * UX) {
                double xo[UX], dj[UX], yr[UX];
#pragma unroll
                for (int t = 0; t < UX; t++) {
                    const int j = j0 + 64 * t;
                    const int jj = j < nxe ? j : 0;
                    const int i = jj / 12, c = jj - i * 12;
                    xo[t] = (EL && j0 == lane) ? xpre[t] : x_it[jj];
                    dj[t] = (EL && j0 == lane) ? djp[EL ? t : 0] : rd_dxb(jj);
                    yr[t] = (EL && j0 == lane) ? ypre[t] : I.yref[(size_t)i * 16 + c];
                }
#pragma unroll
                for (int t = 0; t < UX; t++) {
                    const int j = j0 + 64 * t;
                    if (j < nxe) {
                        const int i = j / 12, c = j - i * 12;
                        const double xn = xo[t] + dj[t];
                        x_it[j] = xn;
                        const double e = xn - yr[t];
                        const double wgt = (LDS == 1 && j0 == lane) ? wxpre[LDS == 1 ? t : 0] : cst[(i == N) ? 16 + c : c];
                        const double sw = (IT::kGrid || (LDS == 0 && I.wst)) ? I.wst[(size_t)i * 16 + c] : ((i == N) ? wgt : P.Ts * wgt);
                        cost += 0.5 * sw * e * e;
                    }
                }
            }
        }
        if constexpr (DF && EL) {
            if (late && status == BROV_STATUS_SUCCESS) {   // record first, then the multipliers of the iterate
                emit_record(cost, u0v, wrote_u0);
                sw_adjoint<true, LDS>(I, W, vfin, DVA, pi_it);
                copy_pi();
            } else if (late) {                             // (a NaN among the inputs: nothing was updated; the sweep the early path skipped is not needed)
            }
        }
        }   // LDS < 3
    }
    if (status != BROV_STATUS_SUCCESS && status != BROV_STATUS_MAXITER) {
        // failed step: report the cost of the entering iterate; the iterate is left as it is (acados: SQP_RTI returns before
        // update_variables) or, with on_failure = RESTART, cold-started at the measured state so that the instance can recover
        const double* x0 = P.x0 + (size_t)b * 12;
        // a restart needs a usable measurement: with a non-finite x0 (sensor glitch) the iterate is kept for the next tick
        const double xl = x0[lane < 12 ? lane : 0];
        const bool restart = P.on_failure == BROV_ON_FAILURE_RESTART && __ballot(!(fabs(xl) < 1e300)) == 0ull;
        for (int j = lane; j < nv; j += 64) {
            const int i = j >> 2, m = j & 3;
            const double e = u_it[j] - I.yref[(size_t)i * 16 + 12 + m];
            cost += 0.5 * ((IT::kGrid || (LDS == 0 && I.wst)) ? I.wst[(size_t)i * 16 + 12 + m] : P.Ts * cst[12 + m]) * e * e;
            if (restart) { u_it[j] = 0.0; lam_it[i * 8 + m] = 0.0; lam_it[i * 8 + 4 + m] = 0.0; }
        }
        for (int j = lane; j < nxe; j += 64) {
            const int i = j / 12, c = j - i * 12;
            const double e = x_it[j] - I.yref[(size_t)i * 16 + c];
            cost += 0.5 * ((IT::kGrid || (LDS == 0 && I.wst)) ? I.wst[(size_t)i * 16 + c] : ((i == N) ? cst[16 + c] : P.Ts * cst[c])) * e * e;
            if (restart) { x_it[j] = x0[c]; if (i < N) pi_it[j] = 0.0; }
        }
    }
    if (!emitted) emit_record(cost, u0v, wrote_u0);
#ifdef BROV_SCHED_TICKET_LATE
    if (sched_p == -2) sched_p = sched_ticket(P, b);
#endif
    sched_note(P, b, sched_p);
    DBG_STAMP(6);
}


// weights and bounds of the lane's rows (cst = [W16 | We12 pad4 | lbu4 | ubu4]).  The LDS-resident kernels request them BEFORE
// the linearisation and hand them to setup_inst afterwards: requested there, the loads would be an exposed L2 round trip
// (the single resident wave has nothing else to run)
struct LaneCst { double Wr[4], Wer[3], lbm, ubm, Wq, Weq, Wuq; };
__device__ __forceinline__ LaneCst load_lane_cst(const double* __restrict__ cst, int lane) {
    const int rg = lane >> 4;
    LaneCst c;
#pragma unroll
    for (int r = 0; r < 4; r++) c.Wr[r] = cst[rg + 4 * r];
#pragma unroll
    for (int r = 0; r < 3; r++) c.Wer[r] = cst[16 + rg + 4 * r];
    c.lbm = cst[32 + rg];
    c.ubm = cst[36 + rg];
    const int cq = (lane >> 2) < NX ? (lane >> 2) : NX - 1;
    c.Wq = cst[cq];
    c.Weq = cst[16 + cq];
    c.Wuq = cst[12 + ((lane >> 2) & 3)];
    return c;
}
__device__ __forceinline__ void setup_inst(const DevParams& P, Inst& I, int b, int lane, const LaneCst* pre = nullptr) {
    const int N = P.N, nv = 4 * N;
    const double* __restrict__ cst = P.cst;
    I.lane = lane; I.rg = lane >> 4; I.cl = lane & 15; I.N = N; I.nv = nv;
    I.i0 = 0; I.NT = N; I.ckpt = 0;
    I.x = P.x + (size_t)b * (N + 1) * 12;
    I.u = P.u + (size_t)b * N * 4;
    I.yref = P.yref + (size_t)b * P.yref_stride;
    I.BA = P.BA + (size_t)b * N * 192;
    I.bvec = P.bvec + (size_t)b * N * 12;
    I.Ks = P.Ks + (size_t)b * N * 64;
    I.Kt = P.Kt + (size_t)b * N * 192;
    I.Mt = P.Mt + (size_t)b * N * 64;
    I.Pb = P.Pb + (size_t)b * N * 12;
    I.kff = P.kff + (size_t)b * N * 4;
    I.vhat = P.vhat + (size_t)b * N * 4;
    I.ipm = P.ipm + (size_t)b * IPM_NARR * nv;
    I.dxb = P.dxb + (size_t)b * (N + 1) * 12;
    I.Ts = P.Ts;
    I.wst = P.wst;
    I.lds_ba = nullptr;
    I.lds_bv = nullptr;
    I.lds_kt = nullptr;
    I.lds_q = nullptr;
    I.lds_r = nullptr;
    I.lds_kff = nullptr;
    I.lds_vhat = nullptr;
    I.lds_dxb = nullptr;
    I.lds_zero = nullptr;
    I.lds_tr = nullptr;
    const LaneCst c = pre ? *pre : load_lane_cst(cst, lane);
#pragma unroll
    for (int r = 0; r < 4; r++) I.Wr[r] = c.Wr[r];
#pragma unroll
    for (int r = 0; r < 3; r++) I.Wer[r] = c.Wer[r];
    I.lbm = c.lbm;
    I.ubm = c.ubm;
    I.Wq = c.Wq; I.Weq = c.Weq; I.Wuq = c.Wuq;
}

#ifndef BROV_QP_WAVES
#define BROV_QP_WAVES 2
#endif
// streaming path: linearisation tiles come from HBM (written by lin_wave_kernel); any horizon
__global__ __launch_bounds__(256, BROV_QP_WAVES) void qp_kernel(DevParams P) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // provably wave-uniform -> SGPR addressing
    const int t = blockIdx.x * (blockDim.x >> 6) + wave;
    if (t >= P.B) return;
    const int b = __builtin_amdgcn_readfirstlane(sched_map(P, t));
    const int lane = threadIdx.x & 63;
    if (t == 0) sched_zero_next(P, lane);
    __shared__ double tr_s[4 * 17];   // per-wave transposition scratch of the backward sweep
    Inst I;
    setup_inst(P, I, b, lane);
    I.lds_tr = (lds_f64*)tr_s + wave * 17;
    double part = 0.0;
    bool nanp = false;
    for (int j = lane; j < P.N; j += 64) {
        const double t = P.kktp[(size_t)b * P.N + j];
        if (t != t) nanp = true;
        part = fmax(part, t);
    }
    qp_body<false>(P, I, b, part, nanp);
}

// coalesced global -> LDS staging of one instance's contiguous input arrays (16 bytes per lane per request).  All requests
// of all arrays are issued before the first LDS write so that they overlap; nd = number of doubles (even).
template <int MAXC>
__device__ __forceinline__ void stage_issue(const double* __restrict__ g, int nd, int lane, dbl2 (&v)[MAXC]) {
#pragma unroll
    for (int k = 0; k < MAXC; k++) {
        const int o = (lane + 64 * k) * 2;
        v[k] = *(const dbl2*)(g + (o < nd ? o : 0));
    }
}
template <int MAXC>
__device__ __forceinline__ void stage_store(double* l, int nd, int lane, const dbl2 (&v)[MAXC]) {
#pragma unroll
    for (int k = 0; k < MAXC; k++) {
        const int o = (lane + 64 * k) * 2;
        if (o < nd) *(lds_d2*)(l + o) = v[k];
    }
}

// Linearisation of the intervals [i0, i0 + n) of instance b by ONE wavefront (n <= 23): ERK4 + forward sensitivities, b_i,
// cost gradients and the NLP KKT partials of the entering iterate.  L = 64/n lanes per interval (3 at n = 20); each lane
// integrates the state once and then walks its share of the 13 non-trivial sensitivity columns; columns land in LDS
// ([A B] compact [n][12][13]), so the scattered 8-byte writes that rule this mapping out against HBM cost nothing.
//   ba_s [n][12][13], bv_s [n][12], q_s [n+1][12] (row n: terminal gradient if the chunk ends the horizon), r_s [n][4];
//   rec_s: scratch for the stage records, n*68 doubles.  part / nanp: this lane's share of the KKT max / NaN flag.
template <bool TWO = true, bool GRID = false>
__device__ __forceinline__ void lin_phase(const DevParams& P, int b, int i0, int n, int lane, double* ba_s, double* bv_s,
                                          double* rec_s, double* q_s, double* r_s, double& part, bool& nanp, bool stamp) {
#ifdef BROV_DBG_LIN
    unsigned long long lin_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    const int N = P.N;
    const double* __restrict__ cst = P.cst;
    const int L = n <= 4 ? 16 : 64 / n;
    const int g = lane / L, j0 = lane - g * L;
    const bool active = g < n;
    const int i = active ? g : n - 1;   // index inside the chunk
    const int ig = i0 + i;              // global interval
    const double* __restrict__ ui = P.u + ((size_t)b * N + ig) * NU;
    // step of this interval and scaled weights of its stage: one number / one weight vector, except on the streaming path's general grid
    const double hstep = GRID ? P.tsv[ig] : P.Ts;
    // 6-disturbance model variant: this interval's roll / pitch disturbance moments.  One-wave kernels request them ahead of the
    // staging below (nothing else would cover the round trip); the two-wave kernel has no registers to carry them that far
    double rp0 = 0.0, rp1 = 0.0;
    if constexpr (TWO) {
        if (P.par_rp) { const double* rp = P.par_rp + ((size_t)b * (N + 1) + ig) * 2; rp0 = rp[0]; rp1 = rp[1]; }
    }
    // the chunk's iterate, parameters, reference and multipliers are contiguous: fetch them with 13 wave-wide 16-byte
    // requests into the (still unused) [A B] area instead of ~90 requests that each touch 20 cache lines, then let every
    // lane pick its interval's operands out of LDS
    const int po = i0 > 0 ? 1 : 0;                // the multipliers of interval i0-1 ride along (pi_{i-1} of the first interval)
    double* sx = ba_s;                            // [n+1][12]
    double* spar = sx + (size_t)(n + 1) * NX;     // [n][16]
    double* syr = spar + (size_t)n * NP;          // [n+1][16]
    double* spi = syr + (size_t)(n + 1) * NY;     // [n+po][12]
    double* su = spi + (size_t)(n + 1) * NX;      // [n][4]
    {
        dbl2 vx[3], vp[3], vy[3], vpi[3], vu[1];
        stage_issue(P.x + ((size_t)b * (N + 1) + i0) * NX, (n + 1) * NX, lane, vx);
        stage_issue(P.par + ((size_t)b * (N + 1) + i0) * NP, n * NP, lane, vp);
        stage_issue(P.yref + (size_t)b * P.yref_stride + (size_t)i0 * NY, (n + 1) * NY, lane, vy);
        stage_issue(P.pi + ((size_t)b * N + i0 - po) * NX, (n + po) * NX, lane, vpi);
        stage_issue(P.u + ((size_t)b * N + i0) * NU, n * NU, lane, vu);
        stage_store(sx, (n + 1) * NX, lane, vx);
        stage_store(spar, n * NP, lane, vp);
        stage_store(syr, (n + 1) * NY, lane, vy);
        stage_store(spi, (n + po) * NX, lane, vpi);
        stage_store(su, n * NU, lane, vu);
    }
    const double* xi = sx + i * NX;
    const double* pp = spar + i * NP;
    const double* yr = syr + i * NY;
    const double* pil = spi + (i + po) * NX;
    const double* pim1 = spi + (ig > 0 ? i + po - 1 : 0) * NX;
    double uu[NU], x0r[NX], yrr[NY], pir[NX], pm1[3];
#pragma unroll
    for (int j = 0; j < NU; j++) uu[j] = su[i * NU + j];
#pragma unroll
    for (int j = 0; j < NX; j++) { x0r[j] = xi[j]; pir[j] = pil[j]; }
#pragma unroll
    for (int j = 0; j < NY; j++) yrr[j] = yr[j];
#pragma unroll
    for (int j = 0; j < 3; j++) pm1[j] = pim1[j];
    const unsigned long long tA = (stamp && P.dbg) ? __builtin_readcyclecounter() : 0;
    LIN_T(0);
    const bool last = ig == N - 1;
    double yrn[NX];
#pragma unroll
    for (int j = 0; j < NX; j++) yrn[j] = yr[NY + j];   // row i+1 of the staged reference: valid for every interval, used by the last one
    const ModelPar m = make_par(pp);
    Wrench w = make_wrench(uu);
    if constexpr (!TWO) {
        if (P.par_rp) { const double* rp = P.par_rp + ((size_t)b * (N + 1) + ig) * 2; rp0 = rp[0]; rp1 = rp[1]; }
    }
    w.k3 = rp0; w.k4 = rp1;
    // cost gradients of this stage (and of the terminal node from the last interval), kept in LDS for all sweeps, and the
    // stationarity rows of the position columns (exactly e_c).  The L lanes of a group write identical values.
    KktAcc ka;
#pragma unroll
    for (int k = 0; k < NX; k++) {
        const double qk = (GRID ? P.wst[(size_t)ig * 16 + k] : P.Ts * cst[k]) * (x0r[k] - yrr[k]);
        q_s[i * NX + k] = qk;
        if (k < 3) ka.upd(ig >= 1 ? qk + pir[k] - pm1[k] : 0.0);
    }
#pragma unroll
    for (int k = 0; k < NU; k++) r_s[i * NU + k] = (GRID ? P.wst[(size_t)ig * 16 + NX + k] : P.Ts * cst[NX + k]) * (uu[k] - yrr[NX + k]);
    if (last) {
#pragma unroll
        for (int k = 0; k < NX; k++) {
            const double qn = cst[16 + k] * (xi[NX + k] - yrn[k]);
            q_s[n * NX + k] = qn;
            if (k < 3) ka.upd(qn - pir[k]);
        }
    }
    const unsigned long long tB = (stamp && P.dbg) ? __builtin_readcyclecounter() : 0;
    LIN_T(1);
    StagePoint sp[4];
    double xn[NX];
    rk4_state(x0r, w, m, hstep, sp, xn);
    const unsigned long long tC = (stamp && P.dbg) ? __builtin_readcyclecounter() : 0;
    LIN_T(2);
    double* tb = ba_s + i * kBaStage;
    // stage records: 4*17 doubles per interval (in the fused kernel they overlay the gain / step arrays, which are dead
    // until the QP phase: 4*17 <= 48+4+4+12)
    lds_f64* rec = (lds_f64*)rec_s + i * kRecInterval;
#pragma unroll
    for (int st = 0; st < 4; st++) store_stage_rec(rec + st * kRecStage, sp[st], m);
    // b_i and the dynamics gap
#pragma unroll
    for (int k = 0; k < NX; k++) {
        const double bk = xn[k] - xi[NX + k];   // x_{i+1}: read from the staging copy here, not carried through the integration in registers
        bv_s[i * NX + k] = bk;
        ka.upd(bk);
    }
    // developer instrumentation, slot 7: loads issued -> cost gradients -> state integrated -> column loop entered
    if (stamp && P.dbg && lane == 0)
        P.dbg[(size_t)b * 8 + 7] = ((tB - tA) & 0xFFFFF) | (((tC - tB) & 0xFFFFF) << 20) | (((__builtin_readcyclecounter() - tC) & 0xFFFFF) << 40);
    LIN_T(3);
    auto finish = [&](int c, const KktOperands& ko, const double (&acc)[NX]) __attribute__((always_inline)) {
        lin_kkt_col(ka, ko, N, ig, c, pir, acc);
#pragma unroll
        for (int k = 0; k < NX; k++) tb[k * kBaStride + (c - 3)] = acc[k];
    };
    // Columns by structure, so that the lanes of a trip run the same code:
    //   (1) attitude angles 3..5 and body rates 9..11: general Jacobian-vector products, 4 per column;
    //   (2) inputs with a yaw-moment component, u1 (rows 7, 11) and u3 (row 11): first stage is df/du itself;
    //   (3) body velocities 6..8 and the pure force inputs u0, u2: closed form (sens_column_cheap).
    // L = 3 at N = 20: 2 + 1 + 2 trips costing about 1, 0.6 and 0.15 of a general one -- 2.9 trip-equivalents (was 5, then 4.3).
    // A lane whose class has run out of columns repeats the class's last column (same values stored twice).
    if constexpr (TWO) {
#pragma unroll 1
        for (int q0 = j0; q0 - j0 < 6; q0 += 2 * L) {   // two general columns per trip: the Jacobian entries of a stage are shared
            const int qa = q0 < 6 ? q0 : 5, qb = q0 + L < 6 ? q0 + L : 5;
            const int ca = qa < 3 ? 3 + qa : 6 + qa, cb = qb < 3 ? 3 + qb : 6 + qb;
            double acc0[NX], acc1[NX];
            const KktOperands koa = load_kkt_operands(P, cst, b, ig, i, n, ca, ui, (const lds_f64*)q_s, (const lds_f64*)r_s);
            const KktOperands kob = load_kkt_operands(P, cst, b, ig, i, n, cb, ui, (const lds_f64*)q_s, (const lds_f64*)r_s);
            sens_column_rec2(rec, m, hstep, ca, cb, acc0, acc1);
            finish(ca, koa, acc0);
            finish(cb, kob, acc1);
        }
    } else {
        // short horizons (L >= 4 lanes per interval, two waves per SIMD): one column per trip -- a pair would mostly repeat
        // column 11, and the second wave covers the latency the pairing is there to hide
#pragma unroll 1
        for (int q0 = j0; q0 - j0 < 6; q0 += L) {
            const int qa = q0 < 6 ? q0 : 5;
            const int ca = qa < 3 ? 3 + qa : 6 + qa;
            double acc0[NX];
            const KktOperands koa = load_kkt_operands(P, cst, b, ig, i, n, ca, ui, (const lds_f64*)q_s, (const lds_f64*)r_s);
            sens_column_rec(rec, m, hstep, ca, acc0);
            finish(ca, koa, acc0);
        }
    }
    LIN_T(4);
    // the closed-form trips are far too short to hide the L2 round trips of their own KKT operands: requested here, under
    // the input-column trip
    constexpr int kCheapTrips = 3;   // ceil(5 / L) <= 3 for L >= 2
    const int nC = (5 + L - 1) / L;
    KktGlobal kg[kCheapTrips];
    int cq[kCheapTrips];
#pragma unroll
    for (int t = 0; t < kCheapTrips; t++) {
        int q = j0 + t * L;
        q = q < 5 ? q : 4;
        cq[t] = q;
        const bool input = q >= 3;
        const int j = input ? (q == 3 ? 0 : 2) : q;
        kg[t] = load_kkt_global(P, b, ig, input ? NX + j : 6 + j, ui);
    }
#pragma unroll 1
    for (int q0 = j0; q0 - j0 < 2; q0 += L) {
        const int q = q0 < 2 ? q0 : 1;
        const int jc = 1 + 2 * q, c = NX + jc;
        double acc[NX];
        const KktOperands ko = load_kkt_operands(P, cst, b, ig, i, n, c, ui, (const lds_f64*)q_s, (const lds_f64*)r_s);
        sens_column_rec_u(rec, m, hstep, jc, acc);
        finish(c, ko, acc);
    }
    LIN_T(5);
    // all closed-form columns of the lane first (independent chains, interleaved by the compiler), then their KKT rows / stores
    double cv[kCheapTrips][4];
#pragma unroll
    for (int t = 0; t < kCheapTrips; t++) {
        if (t < nC) {
            const int q = cq[t];
            const bool input = q >= 3;
            const int j = input ? (q == 3 ? 0 : 2) : q;       // velocity row 6 + j
            constexpr double ir = 1.0 / kRotor;
            const double kbv = !input ? 0.0 : (j == 0 ? (-4.0 * 0.707) * ir * m.imx : -2.0 * ir * m.imz);   // model_bcol rows 6 / 8
            sens_column_cheap(rec, hstep, j, input, kbv, cv[t]);
        }
    }
    LIN_T(6);
#pragma unroll
    for (int t = 0; t < kCheapTrips; t++) {
        if (t < nC) {
            const int q = cq[t];
            const bool input = q >= 3;
            const int j = input ? (q == 3 ? 0 : 2) : q;
            const int c = input ? NX + j : 6 + j;
            double acc[NX];
            expand_cheap(cv[t], j, acc);
            // pi' S[:,c] has four terms here
            const double pr = (j == 0) ? pir[6] : ((j == 1) ? pir[7] : pir[8]);
            const double dotpi = cv[t][0] * pir[0] + cv[t][1] * pir[1] + cv[t][2] * pir[2] + cv[t][3] * pr;
            lin_kkt_rows(ka, finish_kkt_operands(kg[t], cst, i, n, c, (const lds_f64*)q_s, (const lds_f64*)r_s), N, ig, c, dotpi,
                         input ? 0.0 : pr);
#pragma unroll
            for (int k = 0; k < NX; k++) tb[k * kBaStride + (c - 3)] = acc[k];
        }
    }
    LIN_T(7);
#ifdef BROV_DBG_LIN
    if (stamp && P.dbg && lane == 0)
        for (int k = 0; k < 7; k++) P.dbg[(size_t)P.B * 8 + (size_t)b * 8 + (k == 6 ? 7 : k)] = lin_t[k + 1] - lin_t[k];
#endif
    if (active) {
        if (ka.nan) nanp = true;
        part = fmax(part, ka.mx);
    }
}

// coalesced copy of a chunk's linearisation out of LDS into the HBM images of the streaming path: [A B] as [12][16] row-major
// tiles (register image r of the tile = rows rg + 4r, column cl; columns 0..2 are e_c) and b_i.  Also the debug dump of the
// LDS-resident kernels (DevParams::dump_lin), so that tests compare their linearisation with the oracle directly.
__device__ __forceinline__ void copy_out_linearisation(const DevParams& P, int b, int i0, int n, int lane, const double* ba_s,
                                                       const double* bv_s) {
    const int rg = lane >> 4, cl = lane & 15;
    const size_t g0 = (size_t)b * P.N + i0;
    for (int il = 0; il < n; il++) {
        const double* t = ba_s + il * kBaStage;
        double* BA = P.BA + (g0 + il) * 192;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            const int row = rg + 4 * r;
            // every lane reads (clamped to a stored column), then arithmetic instead of a select: written as
            // `cl >= 3 ? t[..] : constant` some builds of the windowed kernel stored the loaded value in the structural columns too
            const double v = t[row * kBaStride + (cl >= 3 ? cl - 3 : 0)];
            const double m = cl >= 3 ? 1.0 : 0.0, c0 = (cl < 3 && row == cl) ? 1.0 : 0.0;
            BA[r * 64 + lane] = fma(m, v, c0);
        }
    }
    for (int j = lane; j < n * NX; j += 64) P.bvec[g0 * NX + j] = bv_s[j];
}

// Streaming path (any horizon): the same wave-wide linearisation, one wavefront per chunk of <= 21 intervals, followed by a
// coalesced copy of the chunk out of LDS into the HBM images qp_kernel reads -- [A B] as [12][16] row-major tiles, b_i, and
// one KKT partial per interval.
constexpr int kLinChunkMax = 21;
__host__ __device__ inline int lin_chunks(int N) { return (N + kLinChunkMax - 1) / kLinChunkMax; }
__host__ __device__ inline int lin_chunk_len(int N) { const int nc = lin_chunks(N); return (N + nc - 1) / nc; }
template <bool GRID>
__device__ __forceinline__ void lin_wave_body(const DevParams& P) {
    extern __shared__ __attribute__((aligned(16))) double lsm[];
    const int N = P.N, lane = threadIdx.x;
    const int nc = lin_chunks(N), C = lin_chunk_len(N);
    const int b = blockIdx.x / nc, ch = blockIdx.x - b * nc;
    const int i0 = ch * C;
    const int n = (N - i0 < C) ? N - i0 : C;
    double* ba_s = lsm;                              // [C][12][13] (also the input staging area: 60 C + 28 doubles)
    double* bv_s = ba_s + (size_t)C * kBaStage;      // [C][12]
    double* rec_s = bv_s + (size_t)C * NX;           // [C][68]
    double* q_s = rec_s + (size_t)C * kRecInterval;  // [C+1][12]
    double* r_s = q_s + (size_t)(C + 1) * NX;        // [C][4]
    double* part_s = r_s + (size_t)C * NU;           // [64]
    double part = 0.0;
    bool nanp = false;
    lin_phase<true, GRID>(P, b, i0, n, lane, ba_s, bv_s, rec_s, q_s, r_s, part, nanp, false);
    part_s[lane] = nanp ? __builtin_nan("") : part;
    __syncthreads();
    const size_t g0 = (size_t)b * N + i0;
    copy_out_linearisation(P, b, i0, n, lane, ba_s, bv_s);
    // one KKT partial per interval: max over the L lanes of its group, NaN-poisoning
    {
        const int L = n <= 4 ? 16 : 64 / n;
        if (lane < n) {
            double m = 0.0;
            bool bad = false;
            for (int j = 0; j < L; j++) {
                const double v = part_s[lane * L + j];
                if (v != v) bad = true; else m = fmax(m, v);
            }
            P.kktp[g0 + lane] = bad ? __builtin_nan("") : m;
        }
    }
}

__global__ __launch_bounds__(64, 1) void lin_wave_kernel(DevParams P) { lin_wave_body<false>(P); }
// the same on a general grid: per-interval time steps, per-stage scaled weights (DevParams::tsv / wst)
__global__ __launch_bounds__(64, 1) void lin_wave_kernel_grid(DevParams P) { lin_wave_body<true>(P); }

// fused path: ONE wavefront owns one OCP instance from linearisation to the updated iterate.  The wave first integrates
// all N intervals at once (64/N lanes per interval, lin_device.hpp) and leaves [A_i B_i] and b_i in its LDS slice
// (N <= kFusedMaxN: 4 waves x 40.5 KB per CU at N = 20), then runs the Riccati IPM on the LDS-resident stage blocks: they
// are read 3-4 times per Newton system and never touch HBM.  One 64-thread block per instance so that a long-running
// (interior-point) instance does not pin the LDS of three finished ones.
constexpr int kFusedMaxN = 23;
template <int W, bool GRID = false, bool DF = false>
__device__ __forceinline__ void rti_fused_body(const DevParams& P) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int b = __builtin_amdgcn_readfirstlane(sched_map(P, blockIdx.x));
    const int lane = threadIdx.x;
    const int N = P.N;
    const bool listed = sched_listed(P, b);   // requested here, used after the linearisation
    if (blockIdx.x == 0) sched_zero_next(P, lane);
    DBG_STAMP(0);
    // LDS slice of this wave: [A B] (13 non-trivial columns) | b | K^T compact | kff | vhat | dx
    double* ba_s = smem;                          // [N][12][13]
    double* bv_s = ba_s + (size_t)N * kBaStage;   // [N][12]
    double* kt_s = bv_s + (size_t)N * NX;         // [N][12][4]
    double* kff_s = kt_s + (size_t)N * kKtStage;  // [N][4]
    double* vh_s = kff_s + (size_t)N * 4;         // [N][4]
    double* dx_s = vh_s + (size_t)N * 4;          // [N+1][12]
    double* q_s = dx_s + (size_t)(N + 1) * NX;    // [N+1][12] cost gradient w.r.t. x (row N = terminal)
    double* r_s = q_s + (size_t)(N + 1) * NX;     // [N][4]    cost gradient w.r.t. u
    double* const_s = r_s + (size_t)N * 4;        // {0.0, 1.0}: targets of structurally constant tile elements
    if (lane == 0) { const_s[0] = 0.0; const_s[1] = 1.0; }
    // ---- preparation: ERK4 + sensitivities of all N intervals at once (lin_phase below)
    double part = 0.0;
    bool nanp = false;
    LaneCst lc;
    if constexpr (W == 1) lc = load_lane_cst(P.cst, lane);   // the two-wave variant has no registers to spare across lin_phase
    lin_phase<W == 1, GRID>(P, b, 0, N, lane, ba_s, bv_s, kt_s, q_s, r_s, part, nanp, true);
    __syncthreads();  // single wave: orders the LDS writes above against the reads below
    if (P.dump_lin) copy_out_linearisation(P, b, 0, N, lane, ba_s, bv_s);
    std::conditional_t<GRID, InstGrid, Inst> I;
    setup_inst(P, I, b, lane, W == 1 ? &lc : nullptr);
    // partial refactorisation of the active-set tries (riccati_backward_tries): checkpoint stage = ceil(N / 4); off for horizons too
    // short to gain from it and for instances the previous solve did not list as expensive
#ifndef BROV_EXP_NO_SPLIT
    I.ckpt = (N >= 8 && P.partial_refactor && listed) ? (N + 3) >> 2 : 0;
#else
    I.ckpt = 0; (void)listed;
#endif
    I.lds_ba = (const lds_f64*)ba_s;
    I.lds_bv = (const lds_f64*)bv_s;
    I.lds_kt = (lds_f64*)kt_s;
    I.lds_q = (const lds_f64*)q_s;
    I.lds_r = (const lds_f64*)r_s;
    I.kff = kff_s;   // generic pointers into LDS (element loops): feed-forward terms, candidate inputs and state steps
    I.vhat = vh_s;   // never leave the CU; the sweeps use the LDS-typed aliases below
    I.dxb = dx_s;
    I.lds_kff = (lds_f64*)kff_s;
    I.lds_vhat = (lds_f64*)vh_s;
    I.lds_dxb = (lds_f64*)dx_s;
    I.lds_zero = (lds_f64*)const_s;
    I.lds_tr = (lds_f64*)const_s + 2;
    {
        const int rg = I.rg, cl = I.cl;
        const int zero = (int)(const_s - ba_s), one = zero + 1, kt0 = (int)(kt_s - ba_s);
        // [A B] image: element (k = rg+4r, c = cl) lives at k*13 + c-3 for c >= 3; columns 0..2 are e_c
        for (int r = 0; r < 3; r++) I.ba_off[r] = cl >= 3 ? (rg + 4 * r) * kBaStride + cl - 3 : ((r == 0 && rg == cl) ? one : zero);
        I.ba_str = cl >= 3 ? kBaStage : 0;
        // [A B]^T image: element (c = rg+4r, k = cl) = [A B](k, c); k >= 12 is padding, c < 3 is e_c
        for (int r = 0; r < 4; r++) {
            const int c = rg + 4 * r;
            I.bat_off[r] = cl >= NX ? zero : (c >= 3 ? cl * kBaStride + c - 3 : (c == cl ? one : zero));
        }
        // lanes cl < 12 read real elements in registers 1..3; register 0 (c = rg) is real only for rg == 3, else e_c
        I.bat_str = cl >= NX ? 0 : kBaStage;
        I.bat_str0 = (cl < NX && rg == 3) ? kBaStage : 0;
        // K^T compact [12][4]: element (c = rg+4r, m = cl < 4)
        for (int r = 0; r < 3; r++) I.kt_off[r] = cl < 4 ? kt0 + (rg + 4 * r) * 4 + cl : zero;
        I.kt_str = cl < 4 ? kKtStage : 0;
    }
    qp_body<W, std::conditional_t<GRID, InstGrid, Inst>, DF>(P, I, b, part, nanp);
}
// One wave per SIMD (up to 512 VGPRs): the variant for horizons whose LDS slice admits only four blocks per CU anyway.
__global__ __launch_bounds__(64, 1) void rti_fused_kernel(DevParams P) { rti_fused_body<1>(P); }
// Two waves per SIMD (256 VGPRs, some spilled): short horizons (N <= 13, at least six blocks per CU by LDS), where the
// second wave fills the first one's MFMA / LDS / dependent-issue waits.
__global__ __launch_bounds__(64, 2) void rti_fused_kernel_w2(DevParams P) { rti_fused_body<2>(P); }
// General grid (per-stage time steps / a separate stage-0 weight), every N <= 23: one wave per SIMD
__global__ __launch_bounds__(64, 1) void rti_fused_kernel_grid(DevParams P) { rti_fused_body<1, true>(P); }
// brov_tick_host at small batches (host mailbox): the record of an early exit goes out ahead of the adjoint sweep (qp_body<.., DF>)
__global__ __launch_bounds__(64, 1) void rti_fused_kernel_mail(DevParams P) { rti_fused_body<1, false, true>(P); }

// function attributes are per device (a process may hold solvers on several GPUs): one flag per (launcher, device)
static bool first_launch_on_device(int which) {
    static bool done[4][64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return true;
    const bool first = !done[which][dev];
    done[which][dev] = true;
    return first;
}

// ---------------------------------------------------------------------------------------------------------------------
// Windowed kernel: horizons that do not fit the LDS slice (N >= 24; the reference ships N = 80, generate_c_code.py:17,24).
// Same algorithm and the same sweep code as rti_fused_kernel, run window by window (Win above).  Pass 1 walks the windows from
// the end of the horizon to its start: linearise the window's intervals into LDS, run the factor sweep over them (P, p carried
// in registers), park the window.  qp_body<3> then runs forward / adjoint (and interior-point) sweeps as loops over windows.
// Persistent blocks: the grid is what fits the chip (one wavefront per SIMD), each block owns one parking image in HBM and
// takes instances from an atomic counter -- the parked working set is (blocks x horizon), not (batch x horizon), and stays
// hot in L2 / Infinity Cache.
constexpr int kWinMaxStages = 20;
constexpr int kLinMaxIntervals = 23;   // lin_phase: 64 / n >= 2 lanes per interval
__host__ __device__ inline int win_chunks(int N) { return (N + kWinMaxStages - 1) / kWinMaxStages; }
__host__ __device__ inline int win_len(int N) { const int nc = win_chunks(N); return (N + nc - 1) / nc; }
// resident split launches: what the preparation parks per quarter of the horizon for a feedback that rolls out the four quarters at once --
// the quarter's closed-loop transition (Psi = Phi', 256), its affine term (row 12 of G as the lanes hold it, 64), and the cost-to-go (P, p) at
// the quarter's END (192 + 192)
constexpr int kSegPark = 704;
__host__ __device__ inline size_t win_ws_doubles(int N, int L) {
    return (size_t)((N + L - 1) / L) * win_img_doubles(L)                  // parked window images
           + (size_t)N * 4 + (size_t)(N + 1) * NX                          // vhat, dx (flat over the horizon)
           + (size_t)N * (64 + 64 + NX) + (size_t)IPM_NARR * 4 * N         // Ks Mt Pb | interior-point vectors
           + 384 + 512 + 4 * kSegPark;                                     // (P, p) entering window 0 (resident mode: stage ckpt): checkpoint of the partial
                                                                           // refactorisation; resident mode: + the step-0 feed-forward terms (4 N <= 512)
}
// RES: resident mode -- one window = the whole horizon (N <= 81) in a slice of up to 160 KB, one block per CU; for batches of at most
// one instance per CU.  Nothing is parked and no window is fetched.  A separate instantiation (rti_window_kernel_res), so that the
// large-batch kernel carries none of its code.
// SPLIT (resident mode only): acados' rti_phase 1 / 2 as two launches (DevParams::rti_split).  The whole backward sweep -- P, p, gains,
// feed-forward terms -- is independent of the measured state (x0 enters with dx_0 = x0 - x_0 in the forward roll-out only), so the
// PREPARATION launch linearises, factorises and parks the LDS image in the instance's workspace, and the FEEDBACK launch fetches it and runs
// qp_body from the forward sweep on: what is left between the arrival of a measurement and u0 is the forward sweep, the bound check, the
// step and the record.  Separate instantiations (rti_window_kernel_res_split, _split_grid).
template <bool RES, bool GRID = false, bool SPLIT = false>
__device__ __forceinline__ void rti_window_body(const DevParams& P) {
    static_assert(!SPLIT || RES, "the split launches exist for the resident mode");
    using InstT = std::conditional_t<GRID, InstGrid, Inst>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane0 = threadIdx.x & 63;       // (RES: four waves per block, see below)
    const int N = P.N, Lc = P.win_L, nc = (N + Lc - 1) / Lc;
    double* ba_s = smem;                      // [Lc][12][13]
    double* bv_s = smem + win_off_bv(Lc);     // [Lc][12]
    double* q_s = smem + win_off_q(Lc);       // [Lc+1][12]
    double* r_s = smem + win_off_r(Lc);       // [Lc][4]
    double* kt_s = smem + win_off_kt(Lc);     // [Lc][12][4]   (kt .. dx double as the linearisation's stage-record scratch)
    double* kff_s = smem + win_off_kff(Lc);   // [Lc][4]
    double* vh_s = smem + win_off_vh(Lc);     // [Lc][4]
    double* dx_s = smem + win_off_dx(Lc);     // [Lc+1][12]
    double* const_s = smem + win_off_const(Lc);
    if constexpr (RES) {
        // Resident mode serves batches of at most one instance per CU: three of the CU's four SIMDs would idle.  The block has FOUR
        // waves; waves 1..3 linearise a quarter of the horizon each for the block's first instance (ticket = block index, known
        // without communication), hand their KKT partials over through the (then dead) stage-record area and end.  The sweeps are
        // serial recursions: wave 0 runs them alone, as it runs everything of any further instance of the block.
        if (threadIdx.x >= 64) {
            if constexpr (SPLIT) {
                if (P.rti_split == 2) {   // feedback: nothing to linearise -- the helper waves fetch their quarters of the parked image and end
                    const int wvf = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
                    const int bf = __builtin_amdgcn_readfirstlane(sched_map(P, (int)blockIdx.x));
                    if (P.pit_done && P.pit_done[bf]) return;   // (rti_pit_kernel_fb has completed this instance's step)
                    const int nd = win_img_doubles(Lc), q = ((nd / 4 + 127) / 128) * 128, o = wvf * q;
                    if (o < nd) win_fetch(P.ws + (size_t)bf * P.ws_stride + o, smem + o, nd - o < q ? nd - o : q, lane0);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    return;
                }
            }
            const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
            const int b = __builtin_amdgcn_readfirstlane(sched_map(P, (int)blockIdx.x));
            if (P.pit_done && P.pit_done[b]) return;   // rti_pit_kernel has completed this instance's step (wave 0 takes the same decision)
            const int lsub = (N + 3) >> 2, j0 = wv * lsub, nj = N - j0 < lsub ? N - j0 : lsub;
            double part = 0.0;
            bool nanp = false;
            __syncthreads();   // wave 0's barrier ahead of the linearisation
            lin_phase<true, GRID>(P, b, j0, nj, lane0, ba_s + (size_t)j0 * kBaStage, bv_s + (size_t)j0 * NX, kt_s + (size_t)j0 * kRecInterval,
                                  q_s + (size_t)j0 * NX, r_s + (size_t)j0 * NU, part, nanp, false);
            ((lds_f64*)kt_s)[(size_t)j0 * kRecInterval + lane0] = nanp ? __builtin_nan("") : part;
            __syncthreads();   // ... and the one behind it
            return;
        }
    }
    if (lane0 == 0) { const_s[0] = 0.0; const_s[1] = 1.0; }
    if (blockIdx.x == 0) {
        sched_zero_next(P, lane0);
        if (lane0 == 0) *P.counter_next = 0;   // the next launch's hand-out counter (this launch uses the other one)
        if constexpr (RES) {
            // batches between one and two instances per CU (pit_rounds_stages): how many instances the parallel-in-time kernel has left to
            // this one -- into a pinned host word the host reads, a solve or two later, when it chooses the mode of a solve
            // (without that kernel in front -- the host has paused it --: how many it WOULD leave, by its own hint: the records of the solve
            // before, read here before any block of this launch can have written one)
            if (P.pit_left_host) {
                int done = 0;
                for (int j = lane0; j < P.B; j += 64)
                    done += P.pit_done ? P.pit_done[j] != 0 : (P.res[j].status == BROV_STATUS_SUCCESS && P.res[j].qp_iter <= 2);
                done = (int)wave_sum((double)done);
                if (lane0 == 0) __hip_atomic_store(P.pit_left_host, ((unsigned long long)(unsigned)P.pit_seq << 32) | (unsigned)(P.B - done), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    double* ws = P.ws + (size_t)blockIdx.x * P.ws_stride;
    Win W;
    W.nc = nc; W.Lc = Lc; W.cur = -1; W.valid = 0;
    W.lds = smem;
    W.img = ws;
    double* ws_vhat = ws + (size_t)nc * win_img_doubles(Lc);
    double* ws_dxb = ws_vhat + (size_t)N * 4;
    double* ws_Ks = ws_dxb + (size_t)(N + 1) * NX;
    double* ws_Mt = ws_Ks + (size_t)N * 64;
    double* ws_Pb = ws_Mt + (size_t)N * 64;
    double* ws_ipm = ws_Pb + (size_t)N * NX;
    double* ws_ck = ws_ipm + (size_t)IPM_NARR * 4 * N;
    for (int trip = 0;; trip++) {
        // the lane index is re-derived behind an opaque move in every iteration: nothing lane-dependent is hoisted out of the
        // instance loop (such loop invariants otherwise sit in VGPRs across lin_phase and push the kernel into scratch)
        int lane;
        asm volatile("v_mov_b32 %0, %1" : "=v"(lane) : "v"(lane0));
        int b = 0;
        if (RES && trip == 0) {
            b = (int)blockIdx.x;   // the helper waves work on this ticket
        } else {
            if (lane == 0) b = atomicAdd(P.counter, 1) + (RES ? (int)gridDim.x : 0);
            b = __builtin_amdgcn_readfirstlane(b);
        }
        if (b >= P.B) break;
        const int ticket = b;
        b = __builtin_amdgcn_readfirstlane(sched_map(P, b));   // expensive instances first
        if constexpr (RES) {
            // (tickets beyond rti_pit_kernel's grid were not its to serve: their flags are stale)
            if (P.pit_done && ticket < P.pit_blocks && P.pit_done[b]) {   // done by rti_pit_kernel: nothing to do but to keep the work-ordering tables consistent
                sched_note(P, b, -1);
                continue;
            }
        }
        // everything per-lane the sweeps need is (re)built AFTER each linearisation call, so that nothing of it is live across
        // lin_phase (which needs the whole architectural register file)
        const LaneCst lc = load_lane_cst(P.cst, lane);
        if constexpr (SPLIT) {   // the parked image belongs to the INSTANCE (the two launches need not give it the same block)
            ws = P.ws + (size_t)b * P.ws_stride;
            W.img = ws;
            ws_vhat = ws + (size_t)nc * win_img_doubles(Lc); ws_dxb = ws_vhat + (size_t)N * 4; ws_Ks = ws_dxb + (size_t)(N + 1) * NX;
            ws_Mt = ws_Ks + (size_t)N * 64; ws_Pb = ws_Mt + (size_t)N * 64; ws_ipm = ws_Pb + (size_t)N * NX; ws_ck = ws_ipm + (size_t)IPM_NARR * 4 * N;
        }
        auto setup = [&](InstT& I) __attribute__((always_inline)) {
            setup_inst(P, I, b, lane, &lc);
            I.Ks = ws_Ks; I.Mt = ws_Mt; I.Pb = ws_Pb; I.ipm = ws_ipm;
            I.vhat = ws_vhat; I.dxb = ws_dxb; I.kff = nullptr; I.Kt = ws_ck;
            // partial refactorisation of the active-set tries: the checkpoint is the state of the factor sweep as it enters window 0
            I.ckpt = !P.partial_refactor ? 0 : (RES ? (N >= 8 ? (N + 3) >> 2 : 0) : (nc >= 2 ? Lc : 0));   // resident mode: a stage, as in the fused kernels
            I.lds_ba = (const lds_f64*)ba_s;
            I.lds_bv = (const lds_f64*)bv_s;
            I.lds_kt = (lds_f64*)kt_s;
            I.lds_q = (const lds_f64*)q_s;
            I.lds_r = (const lds_f64*)r_s;
            I.lds_kff = (lds_f64*)kff_s;
            I.lds_vhat = (lds_f64*)vh_s;
            I.lds_dxb = (lds_f64*)dx_s;
            I.lds_zero = (lds_f64*)const_s;
            I.lds_tr = (lds_f64*)const_s + 2;
            {
                const int rg = I.rg, cl = I.cl;
                const int zero = (int)(const_s - ba_s), one = zero + 1, kt0 = (int)(kt_s - ba_s);
                for (int r = 0; r < 3; r++) I.ba_off[r] = cl >= 3 ? (rg + 4 * r) * kBaStride + cl - 3 : ((r == 0 && rg == cl) ? one : zero);
                I.ba_str = cl >= 3 ? kBaStage : 0;
                for (int r = 0; r < 4; r++) {
                    const int c = rg + 4 * r;
                    I.bat_off[r] = cl >= NX ? zero : (c >= 3 ? cl * kBaStride + c - 3 : (c == cl ? one : zero));
                }
                I.bat_str = cl >= NX ? 0 : kBaStage;
                I.bat_str0 = (cl < NX && rg == 3) ? kBaStage : 0;
                for (int r = 0; r < 3; r++) I.kt_off[r] = cl < 4 ? kt0 + (rg + 4 * r) * 4 + cl : zero;
                I.kt_str = cl < 4 ? kKtStage : 0;
            }
        };
        DBG_STAMP(0);
        // ---- pass 1: linearisation + step-0 factor sweep, last window first
        double part = 0.0;
        bool nanp = false;
        BwdState S;
        W.cur = -1;
        unsigned long long t_lin = 0, t_bwd = 0, t_fl = 0;   // developer instrumentation (P.dbg): pass-1 split, slot 7
        const bool feedback = SPLIT && P.rti_split == 2;
        for (int c = feedback ? -1 : nc - 1; c >= 0; c--) {
            const int i0 = c * Lc, n = (N - i0 < Lc) ? N - i0 : Lc;
            const unsigned long long t0 = P.dbg ? __builtin_readcyclecounter() : 0;
            // cost gradient of the stage after the window (row n of the window's q array; the adjoint sweep reads it): requested
            // here, written after the linearisation.  The last window gets its row n (terminal gradient) from lin_phase.
            double xq = 0.0, yq = 0.0, wq = 0.0;
            if (c < nc - 1 && lane < NX) {
                xq = P.x[((size_t)b * (N + 1) + i0 + n) * NX + lane];
                yq = P.yref[(size_t)b * P.yref_stride + (size_t)(i0 + n) * NY + lane];
                wq = GRID ? P.wst[(size_t)(i0 + n) * 16 + lane] : P.Ts * P.cst[lane];   // scaled state weight of stage i0 + n
            }
            __syncthreads();
            if (RES && trip == 0) {
                // first instance of the block: this wave takes the first quarter of the horizon, waves 1..3 the others
                const int lsub = (n + 3) >> 2;
                lin_phase<true, GRID>(P, b, 0, lsub, lane, ba_s, bv_s, kt_s, q_s, r_s, part, nanp, false);
                __syncthreads();
                for (int wv = 1; wv < 4; wv++) {
                    const double v = ((const lds_f64*)kt_s)[(size_t)wv * lsub * kRecInterval + lane];
                    nanp = nanp | !(v == v);
                    part = fmax(part, v);
                }
            } else if (!RES || n <= kLinMaxIntervals) {
                lin_phase<true, GRID>(P, b, i0, n, lane, ba_s, bv_s, kt_s, q_s, r_s, part, nanp, false);
            } else {
                // resident mode (one window = the whole horizon in a 160 KB slice, small batches): the wave-wide linearisation takes
                // at most 23 intervals at a time -- sub-chunks, each into its own part of the slice (row n_j of a sub-chunk's q is
                // row 0 of the next one's: contiguous)
                const int nsub = (n + kWinMaxStages - 1) / kWinMaxStages, lsub = (n + nsub - 1) / nsub;
                for (int j0 = 0; j0 < n; j0 += lsub) {
                    const int nj = n - j0 < lsub ? n - j0 : lsub;
                    lin_phase<true, GRID>(P, b, i0 + j0, nj, lane, ba_s + (size_t)j0 * kBaStage, bv_s + (size_t)j0 * NX, kt_s, q_s + (size_t)j0 * NX,
                                          r_s + (size_t)j0 * NU, part, nanp, false);
                    __syncthreads();
                }
            }
            if (c < nc - 1 && lane < NX) q_s[n * NX + lane] = wq * (xq - yq);
            __syncthreads();
            if (P.dump_lin) copy_out_linearisation(P, b, i0, n, lane, ba_s, bv_s);
            const unsigned long long t1 = P.dbg ? __builtin_readcyclecounter() : 0;
            InstT I;
            setup(I);
            win_select(I, W, c);
            if (c == nc - 1) bwd_init<true, 3>(I, S);
            if constexpr (SPLIT) {
                // preparation of a split tick: the same sweep in FOUR parts (the quarters the linearisation was made in; the stage checkpoint is
                // the first quarter's end) with the condensing accumulators of the parallel-in-time kernel -- here, with the exact cost-to-go
                // carried from quarter to quarter, they yield each quarter's exact closed-loop transition (Psi, c) --, parked with the
                // cost-to-go at the quarter's end for a feedback launch that rolls out the four quarters at once (rti_pit_kernel_fb)
                const int lsub = (n + 3) >> 2, rg = I.rg, cl = I.cl;
                double* par = ws_ck + 896;
#pragma clang loop unroll(disable)
                for (int j = 3; j >= 0; j--) {
                    const int lo = j * lsub, hi = lo + lsub < n ? lo + lsub : n;
                    if (lo >= n) continue;
                    double* pj = par + (size_t)j * kSegPark;
#pragma unroll
                    for (int r = 0; r < 3; r++) { pj[320 + r * 64 + lane] = S.P[r]; pj[512 + r * 64 + lane] = S.pv[r]; }
                    if (hi == I.ckpt) {
#pragma unroll
                        for (int r = 0; r < 3; r++) { ws_ck[r * 64 + lane] = S.P[r]; ws_ck[192 + r * 64 + lane] = S.pv[r]; }
                    }
#pragma unroll
                    for (int r = 0; r < 3; r++) S.acc.Psi[r] = (rg + 4 * r == cl) ? 1.0 : 0.0;
                    S.acc.Psi[3] = 0.0;
                    S.acc.G = d4{0, 0, 0, 0};
                    bwd_chunk<true, 3, false, true, false, InstT, true>(I, S, hi, lo);
#pragma unroll
                    for (int r = 0; r < 4; r++) pj[r * 64 + lane] = S.acc.Psi[r];
                    pj[256 + lane] = S.acc.G[3];
                }
            } else if constexpr (RES) {
                // resident mode: one window, so the checkpoint of the partial refactorisation is a STAGE (as in the fused kernels): the
                // sweep in two parts out of one copy of the stage loop, (P, p) entering stage ckpt - 1 stored between them
#pragma clang loop unroll(disable)
                for (int ph = 0; ph < 2; ph++) {
                    if (ph == 1) {
                        if (I.ckpt == 0) break;
#pragma unroll
                        for (int r = 0; r < 3; r++) { ws_ck[r * 64 + lane] = S.P[r]; ws_ck[192 + r * 64 + lane] = S.pv[r]; }
                    }
                    bwd_chunk<true, 3, false, true>(I, S, ph == 0 ? n : I.ckpt, ph == 0 ? I.ckpt : 0);
                }
            } else {
                bwd_chunk<true, 3, false, true>(I, S);
            }
            if (!RES && c == 1 && I.ckpt > 0) {   // (P, p) as they enter window 0: six coalesced 512-byte stores, never waited for
#pragma unroll
                for (int r = 0; r < 3; r++) { ws_ck[r * 64 + lane] = S.P[r]; ws_ck[192 + r * 64 + lane] = S.pv[r]; }
            }
            __syncthreads();
            const unsigned long long t2 = P.dbg ? __builtin_readcyclecounter() : 0;
            // park the window: one contiguous image.  Window 0 keeps its K^T | kff in LDS only: the forward sweep starts on the resident
            // copy, and every later factor sweep parks its own gains
            // (a single window is never fetched back: nothing to park)
            if (!RES) win_flush(W.img + (size_t)c * win_img_doubles(Lc), smem, c == 0 ? win_off_kt(Lc) : win_img_doubles(Lc), lane);
            if (P.dbg) { const unsigned long long t3 = __builtin_readcyclecounter(); t_lin += t1 - t0; t_bwd += t2 - t1; t_fl += t3 - t2; }
        }
        if (P.dbg && lane == 0) P.dbg[(size_t)b * 8 + 7] = (t_lin & 0xFFFFF) | ((t_bwd & 0xFFFFF) << 20) | ((t_fl & 0xFFFFF) << 40);
        if constexpr (SPLIT) {
            double* hdr = ws_ck + 384 + 504;   // (behind the resident mode's copy of the feed-forward terms: 4 N <= 320 of its 512 doubles)
            if (P.rti_split == 1) {
                // preparation ends here: the slice as it stands -- [A B] | b | q | r | K^T | kff -- into the instance's workspace, with the KKT
                // partial of the linearisation and the verdicts of the factor sweep
                const double pw = wave_max(part);
                const bool nn = __ballot(nanp) != 0ull;
                win_flush(W.img, smem, win_img_doubles(Lc), lane);
                if (lane == 0) { hdr[0] = nn ? __builtin_nan("") : pw; hdr[1] = S.ok ? 1.0 : 0.0; hdr[2] = S.illc ? 1.0 : 0.0; }
                __syncthreads();
                continue;
            }
            // feedback starts here
            const double h0 = hdr[0], h1 = hdr[1], h2 = hdr[2];
            {   // this wave's quarter of the image (the helper waves fetch the others, see above)
                const int nd = win_img_doubles(Lc), q = ((nd / 4 + 127) / 128) * 128;
                win_fetch(W.img, smem, nd < q ? nd : q, lane);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            nanp = !(h0 == h0); part = nanp ? 0.0 : h0;
            S.ok = h1 != 0.0; S.illc = h2 != 0.0;
        }
        InstT I;
        setup(I);
        W.cur = -1;
        win_select(I, W, 0);
        W.valid = WM_LIN | WM_GAIN;   // window 0 is resident, complete
#ifdef BROV_DBG_WIN
        W.t_fetch = 0; W.n_fetch = 0;
#endif
#if !defined(BROV_WIN_EXP) || BROV_WIN_EXP != 1
        qp_body<(RES ? 4 : 3)>(P, I, b, part, nanp, &W, S.ok, S.illc);
#endif
#ifdef BROV_DBG_WIN
        if (P.dbg && lane == 0) { P.dbg[(size_t)P.B * 8 + (size_t)b * 8 + 3] = W.t_fetch; P.dbg[(size_t)P.B * 8 + (size_t)b * 8 + 4] = W.n_fetch; }
#endif
        __syncthreads();
    }
}
__global__ __launch_bounds__(64, 1) void rti_window_kernel(DevParams P) { rti_window_body<false>(P); }
// the same on a general grid: per-interval time steps, per-stage scaled weights (DevParams::tsv / wst)
__global__ __launch_bounds__(64, 1) void rti_window_kernel_grid(DevParams P) { rti_window_body<false, true>(P); }
__global__ __launch_bounds__(256, 1) void rti_window_kernel_res(DevParams P) { rti_window_body<true>(P); }
__global__ __launch_bounds__(256, 1) void rti_window_kernel_res_grid(DevParams P) { rti_window_body<true, true>(P); }
// rti_phase 1 / 2 as separate launches in the resident mode (see SPLIT above)
__global__ __launch_bounds__(256, 1) void rti_window_kernel_res_split(DevParams P) { rti_window_body<true, false, true>(P); }
__global__ __launch_bounds__(256, 1) void rti_window_kernel_res_split_grid(DevParams P) { rti_window_body<true, true, true>(P); }

// ---------------------------------------------------------------------------------------------------------------------
// Parallel-in-time step-0 solve (round 4): rti_pit_kernel, for the batches the resident mode serves (at most one instance per CU, the
// whole horizon in one LDS slice; the ROS node's batch of one at the shipped N = 80).  There the step is ONE wave's serial chain:
// 80 factor stages + 80 forward stages = 78 us of the 90 us to the record.  Here the block's four waves keep the quarter of the
// horizon they linearised:
//   1. every wave factorises its segment with the ordinary Riccati sweep from a ZERO terminal cost (the last one: the true terminal cost)
//      and accumulates, next to it, how the segment maps to its two ends (PitAcc: Psi, G, c);
//   2. a relay over the three inner boundaries, last to first: the exact cost-to-go (Pc, pc) at a segment's end and the segment's
//      condensed form give the exact cost-to-go at its start,
//          W = (Pc^-1 + G)^-1,  Pc' = P0 + Psi W Psi',  pc' = p0 + Psi (W (c - G pc) + pc)
//      (two 12 x 12 SPD inverses by block sweeps with the factor sweep's own 4 x 4 pivot algebra); then first to last the boundary
//      states and costates,  lam = W (Phi x + c - G pc) + pc,  x' = Phi x + c - G lam;
//   3. every wave adds the costate's share to its feed-forward terms (kff_i -= M_i Z_i' lam: independent per stage) and runs the forward
//      sweep over its own segment from its boundary state.
// Same minimiser as the sequential sweep (scripts/dev/pit_prototype.py: 1e-14 relative on the oracle's linearisations).  Wave 0 then
// checks the bounds; an answer inside them is THE answer (early exit): record, adjoint sweep, full step as in the resident kernel, and
// pit_done[b] = 1 -- the resident kernel, which is launched behind this one in any case, skips the instance.  Anything else (a bound
// active, a pivot block not positive definite or ill-conditioned, a NaN) leaves the iterate untouched and pit_done[b] = 0: the
// resident kernel does the whole step.  Instances whose previous step was not an early exit are not tried (their record says so).
// LDS: the resident slice + 220 doubles (hand-over buffers, one transposition scratch per wave): N <= 80.
constexpr int kPitExtraDoubles = 24 + 144 + 12 + 12 + 3 * 17 + 1;
__host__ __device__ constexpr int pit_off_flags(int L) { return win_off_const(L) + 2 + 17; }   // 6 x 4 doubles: per-wave KKT partial, verdicts, partial sums
__host__ __device__ constexpr int pit_off_P(int L) { return pit_off_flags(L) + 24; }
__host__ __device__ constexpr int pit_off_p(int L) { return pit_off_P(L) + 144; }
__host__ __device__ constexpr int pit_off_x(int L) { return pit_off_p(L) + 12; }
__host__ __device__ constexpr int pit_off_tr(int L) { return pit_off_x(L) + 12; }            // waves 1..3 (wave 0 uses the slice's own)

// inverse of an SPD 4 x 4 block given by its lower triangle (the 2 x 2 block elimination of the factor sweep)
struct Sym4 { double m00, m10, m11, m20, m21, m22, m30, m31, m32, m33; };
__device__ __forceinline__ Sym4 inv4_spd(double a00, double a10, double a11, double a20, double a21, double a22, double a30, double a31, double a32,
                                         double a33, bool& ok) {
    Sym4 m;
    const double detE = a00 * a11 - a10 * a10, iE = fast_rcp(detE);
    const double e00 = a11 * iE, e01 = -a10 * iE, e11 = a00 * iE;
    const double x00 = e00 * a20 + e01 * a21, x01 = e00 * a30 + e01 * a31;
    const double x10 = e01 * a20 + e11 * a21, x11 = e01 * a30 + e11 * a31;
    const double s00 = a22 - (a20 * x00 + a21 * x10), s01 = a32 - (a20 * x01 + a21 * x11);
    const double s11 = a33 - (a30 * x01 + a31 * x11);
    const double detS = s00 * s11 - s01 * s01, iS = fast_rcp(detS);
    m.m22 = s11 * iS; m.m32 = -s01 * iS; m.m33 = s00 * iS;
    m.m20 = -(x00 * m.m22 + x01 * m.m32); m.m30 = -(x00 * m.m32 + x01 * m.m33);
    m.m21 = -(x10 * m.m22 + x11 * m.m32); m.m31 = -(x10 * m.m32 + x11 * m.m33);
    m.m00 = e00 - (m.m20 * x00 + m.m30 * x01); m.m10 = e01 - (m.m20 * x10 + m.m30 * x11);
    m.m11 = e11 - (m.m21 * x10 + m.m31 * x11);
    if (!(a00 > 0.0 && detE > 0.0 && s00 > 0.0 && detS > 0.0)) ok = false;
    return m;
}
// Inverse of an SPD 12 x 12 matrix held as a tile (rows rg + 4r, columns cl < 12; everything else zero) by three symmetric block sweeps:
//   sweep k:  M = S_kk^-1,  Y = M S_k:,  S <- S - S_k:' Y,  block row k <- Y,  block column k <- Y',  S_kk <- -M;     after all three: -S^-1.
// Block row k of the tile is its register k: the products are single 16x16x4 tiles.
__device__ __forceinline__ d4 sweep12(d4 S, int rg, int cl, bool& ok) {
    const d4 z4 = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int c0 = 4 * k;
        const double Rk = S[k];
        const Sym4 m = inv4_spd(readlane_f64(Rk, c0), readlane_f64(Rk, 16 + c0), readlane_f64(Rk, 17 + c0), readlane_f64(Rk, 32 + c0),
                                readlane_f64(Rk, 33 + c0), readlane_f64(Rk, 34 + c0), readlane_f64(Rk, 48 + c0), readlane_f64(Rk, 49 + c0),
                                readlane_f64(Rk, 50 + c0), readlane_f64(Rk, 51 + c0), ok);
        const int cq = cl & 3;
        const int a = rg > cq ? rg : cq, c = rg > cq ? cq : rg;   // element (max, min) of the symmetric block for this lane
        const double r1 = (c == 0) ? m.m10 : m.m11;
        const double r2 = (c == 0) ? m.m20 : ((c == 1) ? m.m21 : m.m22);
        const double r3 = (c == 0) ? m.m30 : ((c == 1) ? m.m31 : ((c == 2) ? m.m32 : m.m33));
        const double msel = (a == 0) ? m.m00 : ((a == 1) ? r1 : ((a == 2) ? r2 : r3));   // M[rg][cl & 3]
        const double mt = (cl < 4) ? msel : 0.0;
        const d4 Y4 = tn1(mt, Rk, z4);                 // rows 0..3: Y = M S_k:
        const double Y = Y4[0];
        d4 Sn = tn1(Rk, -Y, S);                        // S - S_k:' Y
        const double ek = (cl == c0 + rg) ? 1.0 : 0.0;
        const d4 Yt = tn1(Y, ek, z4);                  // Y' placed in block column k
        const bool inblk = (cl >= c0) && (cl < c0 + 4);
#pragma unroll
        for (int r = 0; r < 3; r++) Sn[r] = inblk ? Yt[r] : Sn[r];
        Sn[k] = inblk ? -msel : Y;
        Sn[3] = 0.0;
        S = Sn;
    }
    return d4{-S[0], -S[1], -S[2], 0.0};
}
// (A copy of the record-writing part of qp_body's emit_record, deliberately: with ONE shared device function both call sites compile, pass every
// test -- and the windowed kernel runs 2 % slower (10.69 against 10.89 M solves/s at N = 40, 5.52 against 5.65 M at N = 80, three alternating
// repetitions on one box, scripts/gpu_r4_ao.sh): the register allocation of its tail shifts.  Keep the two in step by hand.)
// the result record of an early exit (what qp_body's emit_record writes): device copy, thrust allocation epilogue
// (bluerov2_dob.cpp:390-395), and -- brov_tick_host -- the host mailbox
__device__ __forceinline__ void pit_emit_record(const DevParams& P, int b, int lane, double cost_lane, double u0_lane, double kkt, int qp_iter) {
    const double cs = wave_sum(cost_lane);
    if (lane == 0) {
        brov_result* r = P.res + b;
        r->cost = cs; r->kkt = kkt; r->status = BROV_STATUS_SUCCESS; r->qp_iter = qp_iter;
    }
    const double a0 = readlane_f64(u0_lane, 0), a1 = readlane_f64(u0_lane, 1), a2 = readlane_f64(u0_lane, 2), a3 = readlane_f64(u0_lane, 3);
    const double s0 = (lane == 0 || lane == 1) ? -a0 : a0;
    const double s1 = (lane == 0 || lane == 2) ? a1 : -a1;
    const double s3 = (lane == 0 || lane == 3) ? a3 : -a3;
    const double th = ((lane < 4) ? (s0 + s1) + s3 : -a2) / kRotor;
    if (lane < 6) P.res[b].thrust[lane] = th;
    if (P.mail) {
        brov_result* m = P.mail + b;
        if (lane < 4) m->u0[lane] = u0_lane;
        if (lane < 6) m->thrust[lane] = th;
        if (lane == 0) { m->cost = cs; m->kkt = kkt; m->status = BROV_STATUS_SUCCESS; m->qp_iter = qp_iter; }
        if (P.mail_flag) {
            __threadfence_system();
            if (lane == 0) __hip_atomic_store(P.mail_flag + b, P.mail_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// FB: the FEEDBACK half of a split tick (rti_phase 2 behind a preparation by rti_window_kernel_res_split, which has parked the factorised LDS
// image and, per quarter of the horizon, the exact closed-loop transition (Psi, c) and the cost-to-go at the quarter's end).  Nothing is
// linearised or factorised for the step-0 answer: the four waves fetch the image, the relay forms the three boundary states and costates from
// the parked quantities (W = Pc, G = 0: x' = Phi x + c, lam = Pc x' + pc), and the quarters are rolled out at once.  The tries -- answers
// that leave the box -- run the kernel's ordinary passes on the fetched image.
template <bool GRID, bool FB = false>
__device__ __forceinline__ void rti_pit_body(const DevParams& P) {
    using InstT = std::conditional_t<GRID, InstGrid, Inst>;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane0 = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int N = P.N, Lc = P.win_L;   // resident mode: Lc = N
    const int b = __builtin_amdgcn_readfirstlane(sched_map(P, (int)blockIdx.x));
    if (b >= P.B) return;
    {   // worth trying?  The previous step of this instance was an early exit (its record says so; a fresh solver: zeros = yes)
        const brov_result* prev = P.res + b;
        const bool try_it = P.pit == 2 || (prev->status == BROV_STATUS_SUCCESS && prev->qp_iter <= (P.pit_try ? 2 : 0));
        if (!try_it) { if (threadIdx.x == 0) P.pit_done[b] = 0; return; }
    }
    double* ba_s = smem;
    double* bv_s = smem + win_off_bv(Lc);
    double* q_s = smem + win_off_q(Lc);
    double* r_s = smem + win_off_r(Lc);
    double* kt_s = smem + win_off_kt(Lc);
    double* kff_s = smem + win_off_kff(Lc);
    double* vh_s = smem + win_off_vh(Lc);
    double* dx_s = smem + win_off_dx(Lc);
    double* const_s = smem + win_off_const(Lc);
    lds_f64* flag_s = (lds_f64*)(smem + pit_off_flags(Lc));
    lds_f64* mailP = (lds_f64*)(smem + pit_off_P(Lc));
    lds_f64* mailp = (lds_f64*)(smem + pit_off_p(Lc));
    lds_f64* mailx = (lds_f64*)(smem + pit_off_x(Lc));
    lds_f64* tr_w = wv == 0 ? (lds_f64*)const_s + 2 : (lds_f64*)(smem + pit_off_tr(Lc)) + (wv - 1) * 17;
    if (threadIdx.x == 0) { const_s[0] = 0.0; const_s[1] = 1.0; }
#define PIT_STAMP(slot) do { if (P.dbg && threadIdx.x == 0) P.dbg[(size_t)b * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
    PIT_STAMP(0);
    int lane;
    asm volatile("v_mov_b32 %0, %1" : "=v"(lane) : "v"(lane0));
    // ---- segments = the quarters of the linearisation
    const int lsub = (N + 3) >> 2;
    const int s0 = wv * lsub, nseg = (N - s0 < lsub) ? N - s0 : lsub;
    const bool last = s0 + nseg == N;
    double part = 0.0;
    bool nanp = false;
    const LaneCst lc = load_lane_cst(P.cst, lane);
    __syncthreads();
    const double* wsb = P.ws + (size_t)(FB ? b : (int)blockIdx.x) * P.ws_stride;   // (a split tick parks by INSTANCE)
    const double* ck_b = wsb + (size_t)1 * win_img_doubles(Lc) + (size_t)N * 4 + (size_t)(N + 1) * NX + (size_t)N * (64 + 64 + NX) + (size_t)IPM_NARR * 4 * N;
    bool pre_bad = false;
    if constexpr (FB) {
        const int nd = win_img_doubles(Lc), q = ((nd / 4 + 127) / 128) * 128, o = wv * q;
        if (o < nd) win_fetch(wsb + o, smem + o, nd - o < q ? nd - o : q, lane);
        const double h0 = ck_b[384 + 504], h1 = ck_b[384 + 505], h2 = ck_b[384 + 506];   // KKT partial of the linearisation, verdicts of the factor sweep
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        nanp = !(h0 == h0);
        part = (wv == 0 && !nanp) ? h0 : 0.0;
        pre_bad = !(h1 != 0.0) || (h2 != 0.0);
    } else {
    lin_phase<true, GRID>(P, b, s0, nseg, lane, ba_s + (size_t)s0 * kBaStage, bv_s + (size_t)s0 * NX, kt_s + (size_t)s0 * kRecInterval, q_s + (size_t)s0 * NX,
                    r_s + (size_t)s0 * NU, part, nanp, false);
    }
    {
        const double pw = wave_max(part);
        const bool nw = __ballot(nanp) != 0ull;
        if (lane == 0) flag_s[wv] = nw ? __builtin_nan("") : pw;
    }
    __syncthreads();   // (the stage-record scratch of the linearisation is the K^T .. dx area the sweeps write next)
    PIT_STAMP(1);
    // ---- this wave's view of its segment
    // (the block's workspace as the resident kernel lays it out: one parked image -- unused here --, candidate inputs, state steps, and the
    // gain | M tiles of the in-loop sweeps, where this kernel keeps its M Z' tiles; nothing else of it is touched)
    double* ws = P.ws + (size_t)(FB ? b : (int)blockIdx.x) * P.ws_stride;
    double* ws_vhat = ws + (size_t)1 * win_img_doubles(Lc);
    double* ws_dxb = ws_vhat + (size_t)N * 4;
    double* ws_Ks = ws_dxb + (size_t)(N + 1) * NX;
    double* ws_ipm = ws_Ks + (size_t)N * (64 + 64 + NX);   // (behind Ks | Mt | Pb) Gamma and the right-hand side of the try
    auto setup = [&](InstT& I, int seg0, int nst, lds_f64* tr) __attribute__((always_inline)) {
        setup_inst(P, I, b, lane, &lc);
        I.Ks = ws_Ks; I.Mt = nullptr; I.Pb = nullptr; I.ipm = ws_ipm;
        I.vhat = ws_vhat; I.dxb = ws_dxb; I.kff = nullptr; I.Kt = nullptr; I.ckpt = 0;
        I.BA = nullptr; I.bvec = nullptr;
        I.i0 = seg0; I.N = nst; I.NT = N;
        const double* ba = ba_s + (size_t)seg0 * kBaStage;
        I.lds_ba = (const lds_f64*)ba;
        I.lds_bv = (const lds_f64*)(bv_s + (size_t)seg0 * NX);
        I.lds_kt = (lds_f64*)(kt_s + (size_t)seg0 * kKtStage);
        I.lds_q = (const lds_f64*)(q_s + (size_t)seg0 * NX);
        I.lds_r = (const lds_f64*)(r_s + (size_t)seg0 * NU);
        I.lds_kff = (lds_f64*)(kff_s + (size_t)seg0 * 4);
        I.lds_vhat = (lds_f64*)(vh_s + (size_t)seg0 * 4);
        I.lds_dxb = (lds_f64*)(dx_s + (size_t)seg0 * NX);
        I.lds_zero = (lds_f64*)const_s;
        I.lds_tr = tr;
        const int rg = I.rg, cl = I.cl;
        const int zero = (int)(const_s - ba), one = zero + 1, kt0 = (int)((kt_s + (size_t)seg0 * kKtStage) - ba);
        for (int r = 0; r < 3; r++) I.ba_off[r] = cl >= 3 ? (rg + 4 * r) * kBaStride + cl - 3 : ((r == 0 && rg == cl) ? one : zero);
        I.ba_str = cl >= 3 ? kBaStage : 0;
        for (int r = 0; r < 4; r++) {
            const int c = rg + 4 * r;
            I.bat_off[r] = cl >= NX ? zero : (c >= 3 ? cl * kBaStride + c - 3 : (c == cl ? one : zero));
        }
        I.bat_str = cl >= NX ? 0 : kBaStage;
        I.bat_str0 = (cl < NX && rg == 3) ? kBaStage : 0;
        for (int r = 0; r < 3; r++) I.kt_off[r] = cl < 4 ? kt0 + (rg + 4 * r) * 4 + cl : zero;
        I.kt_str = cl < 4 ? kKtStage : 0;
    };
    InstT I;
    setup(I, s0, nseg, tr_w);
    const int rg = I.rg, cl = I.cl;
    const d4 z4 = {0, 0, 0, 0};
    // d0 = x0 - x_0 (wave 0 rolls out from it; everybody needs it for nothing else)
    double x0v[3], xiv[3];
    {
        const double* x0 = P.x0 + (size_t)b * 12;
#pragma unroll
        for (int r = 0; r < 3; r++) { x0v[r] = x0[rg + 4 * r]; xiv[r] = I.x[rg + 4 * r]; }
    }
    // the iterate rows and the reference of the segment, for the bound check and the full step behind the forward sweeps: requested here, ahead
    // of the relay (lsub <= 20 stages -> 2 / 4 elements per lane)
    const int nu = nseg * 4, nxr = (last ? nseg + 1 : nseg) * NX;   // the last segment also commits the terminal node
    double uo[2], ur[2], xo[4], yr[4];
    {
        int b2 = b;
        asm volatile("s_mov_b32 %0, %0" : "+s"(b2));
        const double* xr = P.x + ((size_t)b2 * (N + 1) + s0) * NX;
        const double* uu = P.u + ((size_t)b2 * N + s0) * NU;
        const double* yy = P.yref + (size_t)b2 * P.yref_stride + (size_t)s0 * NY;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int j = lane + 64 * t, jj = j < nu ? j : 0;
            uo[t] = uu[jj];
            ur[t] = yy[(size_t)(jj >> 2) * NY + 12 + (jj & 3)];
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int j = lane + 64 * t, jj = j < nxr ? j : 0;
            const int i = jj / 12, cc = jj - i * 12;
            xo[t] = xr[jj];
            yr[t] = yy[(size_t)i * NY + cc];
        }
    }
    // One pass = local factor sweeps, relay, feed-forward correction, forward sweeps (steps 1 - 3 of the header).  Twice at most: the
    // equality-constrained system (step0), and -- when its answer leaves the box -- ONE active-set try with the violated inputs pinned
    // (Gamma = POL_BIG and a right-hand side that lands them on their bounds: qp_body's first try, same arithmetic).
    bool good = !pre_bad;
    d4 lam = z4;   // the costate at this segment's end boundary (the adjoint sweep of the segment enters with it)
    auto solve_pass = [&](const bool step0) __attribute__((always_inline)) {
        // ---- 1. local factor sweep with the condensing accumulators
        BwdState S;
        PitAcc& acc = S.acc;
        wave_fence();
        if (last) bwd_init<true, 3>(I, S);
        else { S.P = z4; S.pv = z4; S.ok = true; }
    #pragma unroll
        for (int r = 0; r < 3; r++) acc.Psi[r] = (rg + 4 * r == cl) ? 1.0 : 0.0;
        acc.Psi[3] = 0.0;
        acc.G = z4;
        const bool parked = FB && step0;   // (uniform over the block)
        const double* pj = ck_b + 896 + (size_t)wv * kSegPark;   // what the preparation parked for this quarter
        if (parked) {
#pragma unroll
            for (int r = 0; r < 4; r++) acc.Psi[r] = pj[r * 64 + lane];
            acc.G = d4{0.0, 0.0, 0.0, pj[256 + lane]};
            S.P = z4; S.pv = z4; S.ok = true;
        } else if (step0) bwd_chunk<true, 3, false, true, false, InstT, true>(I, S);
        else bwd_chunk<true, 3, false, false, false, InstT, true>(I, S);   // (the try: Gamma and its right-hand side from the interior-point arrays)
        wave_fence();
        good = good && S.ok && !S.illc;
        const unsigned long long t_fac = P.dbg ? __builtin_readcyclecounter() : 0;
        // vectors travel row-replicated (lane (rg, cl): elements rg, rg + 4, rg + 8)
        d4 p0;   // p of the segment start: column 0 of S.pv -> every column
    #pragma unroll
        for (int r = 0; r < 3; r++) p0[r] = dpp_f64<0x150>(S.pv[r]);
        p0[3] = 0.0;
        d4 cbar;   // row 12 of G (lanes rg == 0) -> row-replicated
        {
            lds_f64* t = (rg == 0 && cl < NX) ? tr_w + cl : tr_w + 16;
            *t = acc.G[3];
            cbar = d4{tr_w[rg], tr_w[rg + 4], tr_w[rg + 8], 0.0};
        }
        d4 G = acc.G;
        G[3] = 0.0;
        const d4 Psi = acc.Psi;
        d4 idt;
    #pragma unroll
        for (int r = 0; r < 3; r++) idt[r] = (rg + 4 * r == cl) ? 1.0 : 0.0;
        idt[3] = 0.0;
        const d4 Phi = tn<3>(Psi, idt, z4);   // the transpose
        // ---- 2a. coarse relay, last boundary to first: wave j + 1 publishes the cost-to-go at its start, wave j takes it to its own start
        d4 W = z4, vv = z4, pcn = z4, Pcn = z4;   // this segment's W, c - G pc, and the (Pc, pc) it was built with (the forward relay needs them)
        auto publish = [&](const d4& Pt, const d4& pt) __attribute__((always_inline)) {
    #pragma unroll
            for (int r = 0; r < 3; r++) {
                lds_f64* t = cl < NX ? mailP + (rg + 4 * r) * NX + cl : tr_w + 16;   // (the wave's parking slot)
                *t = Pt[r];
            }
            store_vec12_lds(mailp, pt, rg, cl);
        };
        d4 Pst = S.P, pst = p0;   // the exact cost-to-go at this segment's start once the relay has passed (the last segment: already)
        Pst[3] = 0.0;
        if (parked) {   // the cost-to-go at this quarter's end is exact and parked: W = Pc, G = 0 (x' = Phi x + c, lam = Pc x' + pc)
            if (!last) {
                d4 pq;
#pragma unroll
                for (int r = 0; r < 3; r++) { Pcn[r] = cl < NX ? pj[320 + r * 64 + lane] : 0.0; pq[r] = dpp_f64<0x150>(pj[512 + r * 64 + lane]); }
                Pcn[3] = 0.0; pq[3] = 0.0;
                store_vec12_lds(tr_w, pq, rg, cl);
                pcn = d4{tr_w[rg], tr_w[rg + 4], tr_w[rg + 8], 0.0};
                W = Pcn;
                vv = cbar;
            }
        } else
        for (int j = 3; j >= 1; j--) {
            if (wv == j) publish(Pst, pst);
            __syncthreads();
            if (wv == j - 1) {
                d4 Pc;
    #pragma unroll
                for (int r = 0; r < 3; r++) Pc[r] = cl < NX ? (double)mailP[(rg + 4 * r) * NX + cl] : 0.0;
                Pc[3] = 0.0;
                Pcn = Pc;
                pcn = d4{mailp[rg], mailp[rg + 4], mailp[rg + 8], 0.0};
                d4 Pi = sweep12(Pc, rg, cl, good);
    #pragma unroll
                for (int r = 0; r < 3; r++) Pi[r] += G[r];
                W = sweep12(Pi, rg, cl, good);
                const d4 Gp = tn<3>(G, pcn, z4);                    // G pc (G symmetric)
    #pragma unroll
                for (int r = 0; r < 3; r++) vv[r] = cbar[r] - Gp[r];
                d4 Pe, Ce;                                          // [Phi | v] and [0 | pc]: the vectors ride in column 12
    #pragma unroll
                for (int r = 0; r < 3; r++) { Pe[r] = (cl == NX) ? vv[r] : Phi[r]; Ce[r] = (cl == NX) ? pcn[r] : 0.0; }
                Pe[3] = 0.0; Ce[3] = 0.0;
                d4 in = tn<3>(W, Pe, Ce);                           // [W Phi | W v + pc]
                in[3] = 0.0;
                const d4 out = tn<3>(Phi, in, z4);                  // Psi [W Phi | W v + pc]
    #pragma unroll
                for (int r = 0; r < 3; r++) {
                    Pst[r] = S.P[r] + ((cl < NX) ? out[r] : 0.0);
                    pst[r] = p0[r] + dpp_f64<0x15C>(out[r]);        // row_newbcast:12
                }
            }
            __syncthreads();
        }
        const unsigned long long t_cb = P.dbg ? __builtin_readcyclecounter() : 0;
        // ---- 2b. first boundary to last: boundary states and the costates at the segment ends
        d4 xh = z4;
        lam = z4;
        if (wv == 0) {
    #pragma unroll
            for (int r = 0; r < 3; r++) xh[r] = x0v[r] - xiv[r];
        }
        for (int j = 0; j < 3; j++) {
            if (wv == j) {
                const d4 y1 = tn<3>(Psi, xh, z4);                   // Phi x
                d4 y2;
    #pragma unroll
                for (int r = 0; r < 3; r++) y2[r] = y1[r] + vv[r];
                y2[3] = 0.0;
                lam = tn<3>(W, y2, pcn);                            // W (Phi x + c - G pc) + pc
                lam[3] = 0.0;
                const d4 gl = tn<3>(G, lam, z4);
                d4 xn;
    #pragma unroll
                for (int r = 0; r < 3; r++) xn[r] = y1[r] + cbar[r] - gl[r];
                xn[3] = 0.0;
                store_vec12_lds(mailx, xn, rg, cl);
                // What the two explicit inverses behind W are worth on THIS problem: the costate at the boundary must be the gradient of the
                // cost-to-go there, lam = Pc x' + pc -- exactly so for the exact W, and off by (I + Pc G) times the error of lam otherwise.  An
                // iterate on its way out of the physical regime (cond(Pc) 1e8 and more) fails this; its step is left to the resident kernel's
                // sequential sweep, which needs no such inverse (tests/test_gpu_parity.py, the nominal-model fuzz, found such instances).
                const d4 l2 = tn<3>(Pcn, xn, pcn);
                double mis = 0.0, sc = 0.0;
    #pragma unroll
                for (int r = 0; r < 3; r++) { mis = fmax(mis, fabs(l2[r] - lam[r])); sc = fmax(sc, fabs(lam[r])); }
                mis = wave_max(mis); sc = wave_max(sc);
                if (!(mis <= 1e-9 * sc + 1e-300)) good = false;
            }
            __syncthreads();
            if (wv == j + 1) xh = d4{mailx[rg], mailx[rg + 4], mailx[rg + 8], 0.0};
            __syncthreads();
        }
        if (step0) PIT_STAMP(2);
        if (step0 && P.dbg && threadIdx.x == 0) P.dbg[(size_t)b * 8 + 7] = ((t_fac - P.dbg[(size_t)b * 8 + 1]) & 0xFFFFF) | (((t_cb - t_fac) & 0xFFFFF) << 20) | (((__builtin_readcyclecounter() - t_cb) & 0xFFFFF) << 40);
        // ---- 3. the costate's share of the feed-forward terms, then the forward sweep of the segment
        if (!last && !parked) {   // (the parked feed-forward terms are exact: nothing to add)
            store_vec12_lds(tr_w, lam, rg, cl);
            const double lc_ = tr_w[cl < NX ? cl : 0];
            const double lcl = cl < NX ? lc_ : 0.0;
            // (a rolled loop over batches of four stages, the next batch requested before the current one is used: fully unrolled, the 20 stages
            // cost the kernel 18 more SGPR spills than its one lane-spill register holds, and the rest went to scratch)
            const double* kb = I.Ks + (size_t)s0 * 64 + lane;
            const int nlast = nseg - 1;
            double mz[4], mn[4];
    #pragma unroll
            for (int t = 0; t < 4; t++) mz[t] = kb[(t < nlast ? t : nlast) * 64];
    #pragma clang loop unroll(disable)
            for (int i0 = 0; i0 < nseg; i0 += 4) {
    #pragma unroll
                for (int t = 0; t < 4; t++) { const int i = i0 + 4 + t; mn[t] = kb[(i < nlast ? i : nlast) * 64]; }
    #pragma unroll
                for (int t = 0; t < 4; t++) {
                    const int i = i0 + t < nlast ? i0 + t : nlast;       // (past the end: the last stage again, same value written twice)
                    double v = mz[t] * lcl;                              // (M Z')[rg][cl] lam[cl]
                    v += dpp_f64<0xB1>(v);
                    v += dpp_f64<0x4E>(v);
                    v += dpp_f64<0x141>(v);
                    v += dpp_f64<0x140>(v);                              // the row's sum in every lane
                    lds_f64* kp = (cl == 0 && i0 + t < nseg) ? I.lds_kff + i * 4 + rg : tr_w + 16;
                    const double k0 = I.lds_kff[i * 4 + rg];
                    *kp = k0 - v;
                }
    #pragma unroll
                for (int t = 0; t < 4; t++) mz[t] = mn[t];
            }
        }
        wave_fence();
        {
            d4 xx = xh;
            fwd_chunk<3>(I, xx, wv == 0 ? nullptr : tr_w);
        }
        wave_fence();
    };
    solve_pass(true);
    // ---- 4. checks, (one active-set try,) full step and adjoint sweep, every wave on its own segment
    // (everything the record and the full step address is derived from an opaque copy of the instance index HERE: formed from `b` itself the
    // base addresses are loop invariants of the whole kernel, computed up front and spilled -- and the build then reserves scratch)
    int bq = b;
    asm volatile("s_mov_b32 %0, %0" : "+s"(bq));
    lds_f64* vh = I.lds_vhat;          // this segment's candidate inputs [nseg][4] and state steps (row 0 = the boundary it starts from)
    const lds_f64* dxs = I.lds_dxb;
    double* x_it = P.x + ((size_t)bq * (N + 1) + s0) * NX;
    double* u_it = P.u + ((size_t)bq * N + s0) * NU;
    double* pi_it = P.pi + ((size_t)bq * N + s0) * NX;
    double* lam_it = P.lam + ((size_t)bq * N + s0) * 8;
    const int mI = lane & 3;           // input index of this lane's elements j = lane + 64 t of the segment
    const double lbI = P.cst[32 + mI], ubI = P.cst[36 + mI];
    double rd[2];                      // the elements' own Hessian entries (general grid: the scaled input weight of the element's stage)
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int j = lane + 64 * t, jj = j < nu ? j : 0;
        rd[t] = GRID ? P.wst[(size_t)(s0 + (jj >> 2)) * 16 + 12 + mI] : P.Ts * P.cst[12 + mI];
    }
    auto seg_nan = [&]() __attribute__((always_inline)) {   // NaN among what the forward sweep of the segment produced
        bool bad = false;
#pragma unroll
        for (int t = 0; t < 2; t++) { const double vj = vh[lane + 64 * t < nu ? lane + 64 * t : 0]; bad = bad | !(vj == vj); }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const int j = lane + 64 * t + NX;                       // rows 1 .. nseg: the state steps this segment's sweep wrote
            const double e = dxs[j < (nseg + 1) * NX ? j : NX];
            bad = bad | !(e == e);
        }
        return __ballot(bad) != 0ull;
    };
    bool infeas = false;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int j = lane + 64 * t;
        const double vj = vh[j < nu ? j : 0];
        infeas = infeas | ((j < nu) & !(vj >= lbI - uo[t] && vj <= ubI - uo[t]));
    }
    {
        const bool sn = seg_nan(), sf = __ballot(infeas) == 0ull;
        if (lane == 0) { flag_s[4 + wv] = (good && !sn) ? 1.0 : 0.0; flag_s[8 + wv] = sf ? 1.0 : 0.0; }
    }
    __syncthreads();
    PIT_STAMP(3);
    bool all_good = true, all_feas = true;
    double kkt_lin = 0.0;
    bool nan_lin = false;
#pragma unroll
    for (int w = 0; w < 4; w++) {
        all_good = all_good && (flag_s[4 + w] == 1.0);
        all_feas = all_feas && (flag_s[8 + w] == 1.0);
        const double v = flag_s[w];
        nan_lin = nan_lin | !(v == v);
        kkt_lin = fmax(kkt_lin, v);
    }
    double kkt = 0.0;
#pragma unroll
    for (int r = 0; r < 3; r++) kkt_upd(kkt, x0v[r] - xiv[r]);
    bool nanp2 = nan_lin;
    if (kkt != kkt) nanp2 = true;
    kkt = wave_max(fmax(kkt_lin, (kkt != kkt) ? 0.0 : kkt));
    const bool kkt_nan = __ballot(nanp2) != 0ull;
    if (!all_good || kkt_nan) {       // (the same decision in every wave, here and below)
        if (threadIdx.x == 0) P.pit_done[bq] = 0;
        return;
    }
    // the adjoint sweep of the segment.  The multiplier of its last interval is the costate at its end boundary, which the relay has computed
    // (lam; the last segment: the terminal gradient, which the sweep forms itself): the sweep enters with A'pi := lam - (Qd dx_e + q_e).
    // Multipliers -> the head of the segment's K^T area, input gradient -> its feed-forward area (adj_chunk).
    auto seg_adjoint = [&]() __attribute__((always_inline)) {
        d4 atpi = z4;
        if (!last) {
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int row = rg + 4 * r;
                atpi[r] = lam[r] - ((GRID ? P.wst[(size_t)(s0 + nseg) * 16 + row] : P.Ts * I.Wr[r]) * (double)dxs[nseg * NX + row] + (double)I.lds_q[nseg * NX + row]);
            }
        }
        wave_fence();
        adj_chunk<true, 3>(I, atpi, nullptr, nullptr, nullptr);
        wave_fence();
    };
    const bool early = all_feas && P.early_exit;
    double gel[2] = {0.0, 0.0};        // the accepted try's input gradient of this lane's elements (bound multipliers)
    int tries = 0;                     // Newton systems of the QP loop this kernel has solved for the answer it commits
    if (!early) {
        // ---- 5. ONE active-set try (qp_body's first try, element for element): the inputs of the Newton point that violate their bounds are
        // pinned there (Gamma = POL_BIG and the right-hand side that lands them on the bound), the system is solved by a second pass, pinned
        // inputs are snapped onto their bounds, and the point is THE minimiser if no free input leaves the box and no pinned input's multiplier
        // has the wrong sign.  Then it is committed with one Newton system in its record; if not (15 % of the QPs that run the loop on the
        // mixed batch), nothing has been touched and the resident kernel behind this one does the step.
        if (!P.pit_try || P.qp_iter_max < 1) {   // (no try of its own / no Newton system allowed: the resident kernel's)
            if (threadIdx.x == 0) P.pit_done[bq] = 0;
            return;
        }
        // first guess: the inputs of the Newton point that violate their bounds
        double act[2];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int j = lane + 64 * t;
            const double vj = vh[j < nu ? j : 0];
            act[t] = vj < lbI - uo[t] ? -1.0 : (vj > ubI - uo[t] ? 1.0 : 0.0);
        }
        // ... and up to kPitTries - 1 repairs of it, qp_body's first ROUND of tries as far as it goes without an interior-point iteration: a
        // try that asks for more than POL_NCHG repairs ends the round there too
        constexpr int kPitTries = 3;
        bool accepted = false;
#pragma clang loop unroll(disable)
        for (int tk = 0; tk < kPitTries; tk++) {
            if (tk + 1 > P.qp_iter_max) break;
            {
                double* GAM = I.ipm + (size_t)IPM_GAM * I.nv + (size_t)s0 * 4;
                double* RT = I.ipm + (size_t)IPM_RT * I.nv + (size_t)s0 * 4;
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int j = lane + 64 * t;
                    const double uj = uo[t], ac = act[t];
                    const double gm = ac != 0.0 ? POL_BIG : 0.0;
                    const double rr = rd[t] * (uj - ur[t]);
                    if (j < nu) { GAM[j] = gm; RT[j] = rr - gm * ((ac < 0.0 ? lbI : ubI) - uj); }
                }
            }
            __syncthreads();   // (every wave is done with the hand-over buffers and flags of the pass before)
            solve_pass(false);
            bool bad = false;
#pragma unroll
            for (int t = 0; t < 2; t++) {   // pinned inputs exactly onto their bounds; free inputs that leave the box are marked (+-2: to be pinned)
                const int j = lane + 64 * t;
                const double uj = uo[t], lb = lbI - uj, ub = ubI - uj;
                double vj = vh[j < nu ? j : 0];
                bad = bad | ((j < nu) & !(vj == vj));
                if (act[t] != 0.0) vj = act[t] < 0.0 ? lb : ub;
                else act[t] = vj < lb ? -2.0 : (vj > ub ? 2.0 : 0.0);
                lds_f64* o = j < nu ? vh + j : tr_w + 16;
                *o = vj;
            }
            const bool seg_bad = __ballot(bad) != 0ull || seg_nan() || !good;
            seg_adjoint();
            double gmx = 0.0;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int j = lane + 64 * t;
                gel[t] = I.lds_kff[j < nu ? j : 0];
                if (j < nu) gmx = fmax(gmx, fabs(gel[t]));
            }
            gmx = wave_max(gmx);
            if (lane == 0) { flag_s[12 + wv] = seg_bad ? __builtin_nan("") : gmx; }
            __syncthreads();
            bool any_bad = false;
            gmx = 0.0;
#pragma unroll
            for (int w = 0; w < 4; w++) { const double v = flag_s[12 + w]; any_bad = any_bad | !(v == v); gmx = fmax(gmx, v); }
            double cnt = 0.0;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int j = lane + 64 * t;
                if (j < nu) {
                    const double g = gel[t];
                    double ac = act[t];
                    const double tolg = POL_TOL_G * rd[t] + POL_TOL_GREL * gmx;
                    if (ac == 2.0 || ac == -2.0) { ac *= 0.5; cnt += 1.0; }                                          // newly pinned
                    else if ((ac < 0.0 && g < -tolg) || (ac > 0.0 && g > tolg)) { ac = 0.0; cnt += 1.0; }          // released
                    act[t] = ac;
                }
            }
            cnt = wave_sum(cnt);
            if (lane == 0) flag_s[16 + wv] = cnt;
            __syncthreads();
            const double nchg = (flag_s[16] + flag_s[17]) + (flag_s[18] + flag_s[19]);
            if (any_bad) break;
            if (nchg == 0.0) { accepted = true; tries = tk + 1; break; }
            if (nchg > (double)POL_NCHG) break;   // (the round ends: an interior-point iteration is next -- the resident kernel's)
        }
        if (!accepted) {
            if (threadIdx.x == 0) P.pit_done[bq] = 0;
            return;
        }
    }
    PIT_STAMP(4);
    // full step of the segment and its share of the objective at the new iterate
    double cost = 0.0, u0v = 0.0;
#pragma unroll
    for (int t = 0; t < 2; t++) {
        const int j = lane + 64 * t;
        if (j < nu) {
            const int i = j >> 2, m = j & 3;
            const double gg = early ? 0.0 : gel[t];   // no active bound: the bound multipliers are zero
            lam_it[(size_t)i * 8 + m] = gg > 0 ? gg : 0.0;
            lam_it[(size_t)i * 8 + 4 + m] = gg < 0 ? -gg : 0.0;
            const double un = uo[t] + vh[j];
            u_it[j] = un;
            if (wv == 0 && j < 4) { P.res[bq].u0[j] = un; u0v = un; }
            const double e = un - ur[t];
            cost += 0.5 * rd[t] * e * e;
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const int j = lane + 64 * t;
        if (j < nxr) {
            const int i = j / 12, cc = j - i * 12;
            const double xn = xo[t] + dxs[j];
            x_it[j] = xn;
            const double e = xn - yr[t];
            cost += 0.5 * (GRID ? P.wst[(size_t)(s0 + i) * 16 + cc] : ((s0 + i == N) ? P.cst[16 + cc] : P.Ts * P.cst[cc])) * e * e;
        }
    }
    {
        const double cw = wave_sum(cost);
        if (lane == 0) flag_s[20 + wv] = cw;
    }
    __syncthreads();
    if (wv == 0) {   // the record: as soon as the four shares of the objective are in
        const double ctot = ((flag_s[20] + flag_s[21]) + (flag_s[22] + flag_s[23]));
        pit_emit_record(P, bq, lane, lane == 0 ? ctot : 0.0, u0v, kkt, early ? 0 : tries);
        PIT_STAMP(5);
    }
    if (early) seg_adjoint();   // (an accepted try has run it already: its multipliers are the ones to keep)
    win_flush_small(pi_it, (const double*)I.lds_kt, nseg * NX, lane);
    if (threadIdx.x == 0) P.pit_done[bq] = 1;
    if (wv == 0) PIT_STAMP(6);
#undef PIT_STAMP
}
__global__ __launch_bounds__(256, 1) void rti_pit_kernel_fb(DevParams P) { rti_pit_body<false, true>(P); }
__global__ __launch_bounds__(256, 1) void rti_pit_kernel_fb_grid(DevParams P) { rti_pit_body<true, true>(P); }
__global__ __launch_bounds__(256, 1) void rti_pit_kernel(DevParams P) { rti_pit_body<false>(P); }
// the same on a general grid: per-interval time steps, per-stage scaled weights (DevParams::tsv / wst)
__global__ __launch_bounds__(256, 1) void rti_pit_kernel_grid(DevParams P) { rti_pit_body<true>(P); }

void launch_linearise(const DevParams& P, hipStream_t st) {
    const int C = lin_chunk_len(P.N);
    const size_t lds = ((size_t)C * (kBaStage + NX + kRecInterval + NU) + (size_t)(C + 1) * NX + 64) * sizeof(double);
    if (first_launch_on_device(0)) {
        (void)hipFuncSetAttribute((const void*)lin_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        (void)hipFuncSetAttribute((const void*)lin_wave_kernel_grid, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    }
    if (P.tsv) hipLaunchKernelGGL(lin_wave_kernel_grid, dim3(P.B * lin_chunks(P.N)), dim3(64), lds, st, P);
    else hipLaunchKernelGGL(lin_wave_kernel, dim3(P.B * lin_chunks(P.N)), dim3(64), lds, st, P);
}

void launch_qp(const DevParams& P, hipStream_t st) {
    const int waves_per_block = 4;
    const int blocks = (P.B + waves_per_block - 1) / waves_per_block;
    hipLaunchKernelGGL(qp_kernel, dim3(blocks), dim3(64 * waves_per_block), 0, st, P);
}

bool fused_supported(int N) { return N <= kFusedMaxN; }
int sched_buffer_ints_host(int B) { return (sched_buffer_ints(B) + 31) & ~31; }

static size_t windowed_lds_bytes(int L) { return ((size_t)win_off_const(L) + 2 + 17) * sizeof(double); }
// stages per window.  Large batches: windows of <= 20 stages, four blocks per CU.  Batches of at most one instance per CU (the ROS
// node's batch of one, small Monte-Carlo sets): RESIDENT mode -- one window = the whole horizon in a slice of up to 160 KB, one
// block per CU: no parking, no window fetches (N <= 80 fits)
int windowed_stage_count(int N, int B) {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const bool no_resident = getenv("BROV_DEV_NO_RESIDENT") && atoi(getenv("BROV_DEV_NO_RESIDENT")) != 0;   // development knob (tests)
    if (!no_resident && N > kWinMaxStages && B <= cus && windowed_lds_bytes(N) <= 160 * 1024) return N;
    return win_len(N);
}
size_t windowed_ws_doubles(int N, int L) { return win_ws_doubles(N, L); }
static bool windowed_resident(int L) { return L > kWinMaxStages; }
int windowed_blocks(int N, int B, int L) {
    if (first_launch_on_device(2)) {
        (void)hipFuncSetAttribute((const void*)rti_window_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)rti_window_kernel_res, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)rti_window_kernel_grid, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)rti_window_kernel_res_grid, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)rti_window_kernel_res_split, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)rti_window_kernel_res_split_grid, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    int dev = 0, cus = 256, per_cu = 4;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)N;
    const void* fn = windowed_resident(L) ? (const void*)rti_window_kernel_res : (const void*)rti_window_kernel;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, windowed_resident(L) ? 256 : 64, windowed_lds_bytes(L)) != hipSuccess || per_cu < 1)
        per_cu = windowed_resident(L) ? 1 : 4;
    long long fit = (long long)cus * per_cu;
    // development knob (tests/test_gpu_windowed.py): fewer persistent blocks, so that small batches take several instances per block
    if (const char* e = getenv("BROV_DEV_WIN_BLOCKS")) { const long long v = atoll(e); if (v >= 1 && v < fit) fit = v; }
    return (int)(B < fit ? B : fit);
}
// Batches between one and two instances per CU at 48 <= N <= 80: as long as the parallel-in-time kernel can serve a solve (no dumped linearisation,
// BROV_PIT != 0) it runs ONE BLOCK PER INSTANCE -- a CU's second block follows its first -- with the resident kernel behind it for what
// it leaves, instead of the windowed kernel: 512 instances at N = 80 take 0.154 ms against 0.192 ms (N = 60: 0.135 / ~0.153; N = 40:
// 0.120 / 0.114 -- hence the lower limit; scripts/dev/mid_batch_rate.py).  What the parallel kernel leaves (instances with many active
// bounds) starts only then, on one wave: a batch with a quarter of its instances saturated would lose 20 % against the windowed kernel --
// the host follows the number of instances left (nmpc_api.hip, kPitPause; BROV_PIT_ROUNDS=0 keeps the windowed kernel altogether).  Decided per solve: the solver is created for the windowed kernel and with a
// workspace that serves either.  Returns the resident stage count (= N) or 0.
constexpr int kPitRounds = 2, kPitRoundsMinN = 48;
int pit_rounds_stages(int N, int B) {
    int dev = 0, cus = 256;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const bool off = (getenv("BROV_PIT_ROUNDS") && atoi(getenv("BROV_PIT_ROUNDS")) == 0) || (getenv("BROV_DEV_NO_RESIDENT") && atoi(getenv("BROV_DEV_NO_RESIDENT")) != 0);
    return (!off && B > cus && B <= kPitRounds * cus && N >= kPitRoundsMinN && pit_supported(N, N)) ? N : 0;
}
bool pit_supported(int N, int win_L) {
    return windowed_resident(win_L) && win_L == N && N >= 24 && N <= 80 && windowed_lds_bytes(win_L) + kPitExtraDoubles * sizeof(double) <= 160 * 1024;
}
void launch_windowed(const DevParams& P, hipStream_t st) {
    if (windowed_resident(P.win_L) || P.rti_split) {   // (the split launches are the resident mode's at every horizon)
        if (P.pit && P.pit_done && first_launch_on_device(3)) {
            (void)hipFuncSetAttribute((const void*)rti_pit_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)rti_pit_kernel_grid, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)rti_pit_kernel_fb, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute((const void*)rti_pit_kernel_fb_grid, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        if (P.pit && P.pit_done && P.rti_split == 2) {   // feedback of a split tick: the quarters rolled out at once from what the preparation parked
            if (P.tsv) hipLaunchKernelGGL(rti_pit_kernel_fb_grid, dim3(P.pit_blocks), dim3(256), windowed_lds_bytes(P.win_L) + kPitExtraDoubles * sizeof(double), st, P);
            else hipLaunchKernelGGL(rti_pit_kernel_fb, dim3(P.pit_blocks), dim3(256), windowed_lds_bytes(P.win_L) + kPitExtraDoubles * sizeof(double), st, P);
        } else if (P.pit && P.pit_done) {   // parallel-in-time step-0 solve first; the resident kernel skips what it completed
            // (pit_blocks = B where every instance has a workspace of its own: beyond one instance per CU the blocks queue for the CUs)
            if (P.tsv) hipLaunchKernelGGL(rti_pit_kernel_grid, dim3(P.pit_blocks), dim3(256), windowed_lds_bytes(P.win_L) + kPitExtraDoubles * sizeof(double), st, P);
            else hipLaunchKernelGGL(rti_pit_kernel, dim3(P.pit_blocks), dim3(256), windowed_lds_bytes(P.win_L) + kPitExtraDoubles * sizeof(double), st, P);
        }
        if (P.rti_split && P.tsv) hipLaunchKernelGGL(rti_window_kernel_res_split_grid, dim3(P.win_blocks), dim3(256), windowed_lds_bytes(P.win_L), st, P);
        else if (P.rti_split) hipLaunchKernelGGL(rti_window_kernel_res_split, dim3(P.win_blocks), dim3(256), windowed_lds_bytes(P.win_L), st, P);
        else if (P.tsv) hipLaunchKernelGGL(rti_window_kernel_res_grid, dim3(P.win_blocks), dim3(256), windowed_lds_bytes(P.win_L), st, P);
        else hipLaunchKernelGGL(rti_window_kernel_res, dim3(P.win_blocks), dim3(256), windowed_lds_bytes(P.win_L), st, P);
    }
    else if (P.tsv) hipLaunchKernelGGL(rti_window_kernel_grid, dim3(P.win_blocks), dim3(64), windowed_lds_bytes(P.win_L), st, P);
    else hipLaunchKernelGGL(rti_window_kernel, dim3(P.win_blocks), dim3(64), windowed_lds_bytes(P.win_L), st, P);
}
bool windowed_is_resident(int win_L) { return windowed_resident(win_L); }
// the split launches (rti_phase 1 / 2) of the resident kernel at a horizon the FUSED kernels serve in one call (N <= 23): the four quarters the
// block's waves linearise must all hold a stage
bool split_resident_horizon(int N) { return N >= 4 && 3 * ((N + 3) >> 2) < N && windowed_lds_bytes(N) <= 160 * 1024; }


static size_t fused_lds_bytes(int N) { return ((size_t)N * (kBaStage + NX + kKtStage + 4 + 4 + 4) + 2 * (size_t)(N + 1) * NX + 2 + 17) * sizeof(double); }
static bool fused_two_wave(size_t lds, int force) {   // force: DevKnobs::fused_waves (development knob)
    return force ? force == 2 : 6 * lds <= 160 * 1024;   // N <= 13
}
// what the LDS-resident kernel of this horizon asks of a CU: info = {dynamic LDS bytes per block, blocks the occupancy query grants
// per CU, threads per block, 1 fused / 2 fused two-wave / 3 windowed / 4 windowed resident}.  For bench.py's horizon sweep (the
// "LDS-occupancy crossover" of BASELINE configs[4]): which horizon still fits four instances into a CU's 160 KB.
void lds_kernel_info(int N, int win_L, bool windowed, int32_t info[4], const DevKnobs& k) {
    const void* fn;
    size_t lds;
    int threads = 64, kind;
    if (windowed) {
        lds = windowed_lds_bytes(win_L);
        const bool res = windowed_resident(win_L);
        fn = res ? (const void*)rti_window_kernel_res : (const void*)rti_window_kernel;
        threads = res ? 256 : 64;
        kind = res ? 4 : 3;
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    } else {
        lds = fused_lds_bytes(N);
        const bool w2 = fused_two_wave(lds, k.fused_waves);
        fn = w2 ? (const void*)rti_fused_kernel_w2 : (const void*)rti_fused_kernel;
        kind = w2 ? 2 : 1;
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) != hipSuccess) per_cu = -1;
    info[0] = (int32_t)lds; info[1] = per_cu; info[2] = threads; info[3] = kind;
}
void launch_fused(const DevParams& P, hipStream_t st, const DevKnobs& k) {
    const size_t lds = fused_lds_bytes(P.N);
    if (first_launch_on_device(1)) {
        (void)hipFuncSetAttribute((const void*)rti_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)rti_fused_kernel_w2, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)rti_fused_kernel_grid, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)rti_fused_kernel_mail, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    // development knobs (scripts/dev/occupancy_probe.py): pad the LDS request / force a variant (1, 2; default by LDS size)
    const size_t pad = (size_t)k.lds_pad;
    const bool w2 = fused_two_wave(lds, k.fused_waves);
    if (P.mail && P.mail_early && !P.tsv && !w2) {   // mailbox tick (<= 64 instances): the variant that delivers first
        hipLaunchKernelGGL(rti_fused_kernel_mail, dim3(P.B), dim3(64), lds + pad, st, P);
        return;
    }
    if (P.tsv) hipLaunchKernelGGL(rti_fused_kernel_grid, dim3(P.B), dim3(64), lds + pad, st, P);
    else if (w2) hipLaunchKernelGGL(rti_fused_kernel_w2, dim3(P.B), dim3(64), lds + pad, st, P);
    else hipLaunchKernelGGL(rti_fused_kernel, dim3(P.B), dim3(64), lds + pad, st, P);
}

}  // namespace brov

// ---- test hook: the tile primitive alone (tests/test_gpu_tiles.py checks it against numpy with asymmetric data) ----
namespace brov {
__global__ void tile_tn_kernel(const double* xt, const double* y, const double* c, double* out, int k4) {
    const int lane = threadIdx.x;
    d4 X = load_tile4(xt, lane), Y = load_tile4(y, lane), C = load_tile4(c, lane);
    d4 D = C;
    if (k4 == 1) D = tn<1>(X, Y, C);
    if (k4 == 2) D = tn<2>(X, Y, C);
    if (k4 == 3) D = tn<3>(X, Y, C);
    if (k4 == 4) D = tn<4>(X, Y, C);
#pragma unroll
    for (int r = 0; r < 4; r++) out[r * 64 + lane] = D[r];
}
}  // namespace brov

// ---- test hook: the 12 x 12 SPD inverse of the parallel-in-time kernel's relay (tests/test_gpu_pit.py checks it against numpy) ----
namespace brov {
__global__ void sweep12_kernel(const double* a, double* out, int* okf) {   // a, out: row-major [12][12]
    const int lane = threadIdx.x, rg = lane >> 4, cl = lane & 15;
    d4 S;
#pragma unroll
    for (int r = 0; r < 3; r++) S[r] = cl < 12 ? a[(rg + 4 * r) * 12 + cl] : 0.0;
    S[3] = 0.0;
    bool ok = true;
    const d4 R = sweep12(S, rg, cl, ok);
#pragma unroll
    for (int r = 0; r < 3; r++)
        if (cl < 12) out[(rg + 4 * r) * 12 + cl] = R[r];
    if (lane == 0) *okf = ok ? 1 : 0;
}
}  // namespace brov
extern "C" int brov_selftest_sweep12(const double* a, double* out, int* ok) {
    double* d = nullptr;
    if (hipMalloc((void**)&d, (2 * 144 + 1) * sizeof(double)) != hipSuccess) return BROV_ERR_NO_DEVICE;
    hipMemcpy(d, a, 144 * sizeof(double), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(brov::sweep12_kernel, dim3(1), dim3(64), 0, 0, d, d + 144, (int*)(d + 288));
    hipError_t e = hipMemcpy(out, d + 144, 144 * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(ok, d + 288, sizeof(int), hipMemcpyDeviceToHost);
    hipFree(d);
    return e == hipSuccess ? BROV_OK : BROV_ERR_HIP;
}

extern "C" int brov_selftest_tile_tn(const double* xt, const double* y, const double* c, double* out, int k4) {
    double* d = nullptr;
    if (hipMalloc((void**)&d, 4 * 256 * sizeof(double)) != hipSuccess) return BROV_ERR_NO_DEVICE;
    hipMemcpy(d, xt, 256 * sizeof(double), hipMemcpyHostToDevice);
    hipMemcpy(d + 256, y, 256 * sizeof(double), hipMemcpyHostToDevice);
    hipMemcpy(d + 512, c, 256 * sizeof(double), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(brov::tile_tn_kernel, dim3(1), dim3(64), 0, 0, d, d + 256, d + 512, d + 768, k4);
    hipError_t e = hipMemcpy(out, d + 768, 256 * sizeof(double), hipMemcpyDeviceToHost);
    hipFree(d);
    return e == hipSuccess ? BROV_OK : BROV_ERR_HIP;
}
