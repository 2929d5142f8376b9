// qp/pit.hpp -- rti_pit_body: the step-0 solve and the first active-set tries parallel in time on the four wavefronts of a resident-mode block.
// Part of ONE translation unit: qp_kernel.hip includes these headers in layer order (tiles -> sweeps -> window -> sched -> qp_body ->
// lin_phase -> fused -> windowed -> pit) and instantiates the kernels between them; see the file map at the head of qp_kernel.hip.
#pragma once

namespace brov {

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
__device__ __forceinline__ void pit_emit_record(const DevParams& P, int b, int lane, double cost_lane, double u0_lane, double kkt, int qp_iter, int status = BROV_STATUS_SUCCESS) {
    const double cs = wave_sum(cost_lane);
    if (lane == 0) {
        brov_result* r = P.res + b;
        r->cost = cs; r->kkt = kkt; r->status = status; r->qp_iter = qp_iter;
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
        if (lane == 0) { m->cost = cs; m->kkt = kkt; m->status = status; m->qp_iter = qp_iter; }
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
        // (round 5: the kernel runs the whole QP loop -- tries and interior-point iterations -- so every instance is offered; BROV_PIT_TRY=0, the
        // development knob that keeps the loop out of it, leaves it the instances whose previous step was an early exit, as in round 3)
        const bool try_it = P.pit == 2 || P.pit_try || (prev->status == BROV_STATUS_SUCCESS && prev->qp_iter == 0);
        if (!try_it) { if (threadIdx.x == 0) P.pit_done[b] = 0; return; }
    }
    // the step-0 pass keeps what "light" tries reuse (below) only for instances that ran the QP loop in their previous step -- saturated inputs stay
    // saturated for many ticks --: the split sweep and the stores cost a tracking tick 2 us (measured), which is 3 % of what the node pays every tick
    const bool keep_step0 = !FB && P.pit_light && P.res[b].qp_iter > 0;
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
    // What the last-to-first relay left this wave -- W, c - G pc, the (Pc, pc) it was built with, the exact cost-to-go at the segment's start -- in
    // the last pass that ran all three hops with no pinned input behind the first segment.  A try whose pins all sit in the first segment (98 % of
    // them: a far-off instance saturates its first stages) changes nothing behind it: segments 1 .. 3 reproduce their local sweeps bit for bit, so
    // hops 3 -> 2 and 2 -> 1 would reproduce these values, and only hop 1 -> 0 is run (two of the relay's three serial hops, ~13 k cycles per try).
    d4 kW = z4, kvv = z4, kpcn = z4, kPcn = z4, kPst = z4, kpst = z4;
    bool relay_cached = false, only0 = false;   // (block-uniform; only0: this pass pins inputs of the first segment only -- set by the QP loop)
    // ... and a try whose pins all sit below the first segment's checkpoint stage (ceil(nseg / 4): the same 98 %) changes nothing ABOVE it either: what
    // the step-0 pass computed for the stages behind the checkpoint -- and for the whole of segments 1 .. 3 -- is kept and reused ("light" pass):
    // wave 0 refactorises the stages below its checkpoint from the saved (P, p, Psi, G), waves 1 .. 3 do not sweep at all.  Saved by the step-0 pass in
    // the block's (otherwise unused) parked-image area: per wave Psi | G | local feed-forward terms | the head of the K^T area (which the segment's
    // adjoint sweep overwrites with multipliers), and wave 0's checkpoint.  `clean`: every pass since the step-0 pass would have qualified.
    bool clean = false, light_ok = false;
    double* const sv = ws + (size_t)wv * 1024;   // this wave's save area: Psi 256 | G 256 | kff <= 128 | K^T head <= 384
    double* const svck = ws + 4096;              // wave 0: (P, p, Psi, G) entering stage ckst - 1
    const int ckst = (nseg + 3) >> 2;
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
        } else if (step0) {
            // (in two parts with wave 0's checkpoint between them -- out of ONE call site, as in the fused kernels)
            const bool split = keep_step0 && wv == 0 && nseg > ckst;
#pragma clang loop unroll(disable)
            for (int ph = 0; ph < 2; ph++) {
                if (ph == 1) {
                    if (!split) break;
#pragma unroll
                    for (int r = 0; r < 4; r++) { svck[r * 64 + lane] = S.P[r]; svck[256 + r * 64 + lane] = S.pv[r]; svck[512 + r * 64 + lane] = acc.Psi[r]; svck[768 + r * 64 + lane] = acc.G[r]; }
                }
                bwd_chunk<true, 3, false, true, false, InstT, true>(I, S, ph == 0 ? nseg : ckst, (ph == 0 && split) ? ckst : 0);
            }
        } else {
            const bool light = !FB && light_ok;   // (uniform over the block)
            if (light) {
                // what the segment's adjoint sweep and the feed-forward correction of the pass before have overwritten, from the step-0 pass
                const int f0 = wv == 0 ? ckst * 4 : 0;
                for (int j = f0 + lane; j < nseg * 4; j += 64) I.lds_kff[j] = sv[512 + j];
                if (wv != 0) {
                    for (int j = lane; j < nseg * NX; j += 64) I.lds_kt[j] = sv[640 + j];
#pragma unroll
                    for (int r = 0; r < 4; r++) { acc.Psi[r] = sv[r * 64 + lane]; acc.G[r] = sv[256 + r * 64 + lane]; }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; r++) { S.P[r] = svck[r * 64 + lane]; S.pv[r] = svck[256 + r * 64 + lane]; acc.Psi[r] = svck[512 + r * 64 + lane]; acc.G[r] = svck[768 + r * 64 + lane]; }
                }
                wave_fence();
            }
            // (the try: Gamma and its right-hand side from the interior-point arrays)
            if (!light || wv == 0) bwd_chunk<true, 3, false, false, false, InstT, true>(I, S, light ? ckst : nseg, 0);
        }
        if (step0 && keep_step0) {   // the step-0 pass: what a light pass will need of it -- issued AHEAD of the fence the sweep's own tile stores need anyway
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the sweep's LDS writes, read back by other lanes here)
#pragma unroll
            for (int r = 0; r < 4; r++) { sv[r * 64 + lane] = acc.Psi[r]; sv[256 + r * 64 + lane] = acc.G[r]; }
            for (int j = lane; j < nseg * 4; j += 64) sv[512 + j] = I.lds_kff[j];
            for (int j = lane; j < nseg * NX; j += 64) sv[640 + j] = I.lds_kt[j];
        }
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
        }
        const bool short_relay = !parked && !step0 && only0 && relay_cached;
        if (short_relay && wv >= 1) { W = kW; vv = kvv; pcn = kpcn; Pcn = kPcn; Pst = kPst; pst = kpst; }
        if (!parked)
        for (int j = short_relay ? 1 : 3; j >= 1; j--) {
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
        if (step0) clean = keep_step0;
        if (!parked && !short_relay) {
            relay_cached = step0 || only0;
            if (relay_cached) { kW = W; kvv = vv; kpcn = pcn; kPcn = Pcn; kPst = Pst; kpst = pst; }
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
    double gel[2] = {0.0, 0.0};        // the committed point's input gradient of this lane's elements (bound multipliers)
    int tries = 0;                     // Newton systems of the QP loop (tries + interior-point iterations) behind the answer this kernel commits
    int status = BROV_STATUS_SUCCESS;
    if (!early) {
        // ---- 5. qp_body's QP loop, parallel in time (round 5: all of it; round 4: its first three tries).  Every Newton system of the loop --
        // an active-set try with the guessed inputs pinned, the predictor and the corrector of an interior-point iteration -- is ONE more
        // solve_pass over the four segments with its own Gamma and right-hand side; the element loops run on the two elements a lane holds
        // of its segment, their reductions meet in LDS.  The schedule is qp_body's (the oracle's "ACTIVE-SET POLISH"), state for state:
        //   TRY    inputs of the guess pinned (Gamma = POL_BIG, right-hand side landing them on their bounds), solve, snap, adjoint sweep of the
        //          segment (entering with the relay's costate), repairs counted: none -> THE minimiser, committed
        //   START  interior start: the last point clamped 5 % into the box; its state steps and input gradient come from a pass with EVERY
        //          input pinned at that point (zero gains: the closed-loop machinery run open loop) -- where qp_body runs a roll-out and an
        //          adjoint sweep, which have no parallel form; mu0, multipliers, residual
        //   PRED   Gamma = ll / tl + lu / tu, right-hand side of the affine step, solve; step length, centring, corrector right-hand side
        //   CORR   the same Gamma, the corrector's right-hand side: a second factor pass (qp_body runs a solve-only sweep here: 55 k cycles
        //          on one wave; the factor pass of four waves is 48 k + relay), solve; step, update, classification, termination test
        //   FINAL  an answer that is not an accepted try (the converged interior-point iterate, or the last point at the iteration limit):
        //          one pass with every input pinned at it gives the state steps and multipliers the full step commits
        // Anything that goes wrong -- a pivot block not positive definite or ill-conditioned, a NaN, a relay check that fails -- leaves the
        // iterate untouched for the resident kernel behind this one, as before.
        if (!P.pit_try || P.qp_iter_max < 1) {   // (no loop of its own / no Newton system allowed: the resident kernel's)
            if (threadIdx.x == 0) P.pit_done[bq] = 0;
            return;
        }
        int slot = 0;                  // reductions over the block rotate through five groups of four LDS words: a group is rewritten five barriers later
        auto meet = [&](double v) __attribute__((always_inline)) -> lds_f64* {
            lds_f64* g = flag_s + 4 + 4 * slot;
            slot = slot == 4 ? 0 : slot + 1;
            if (lane == 0) g[wv] = v;
            __syncthreads();
            return g;
        };
        auto block_sum = [&](double v) __attribute__((always_inline)) -> double { lds_f64* g = meet(wave_sum(v)); return (g[0] + g[1]) + (g[2] + g[3]); };
        auto block_max = [&](double v) __attribute__((always_inline)) -> double { lds_f64* g = meet(wave_max(v)); return fmax(fmax(g[0], g[1]), fmax(g[2], g[3])); };
        auto block_min = [&](double v) __attribute__((always_inline)) -> double { lds_f64* g = meet(wave_min(v)); return fmin(fmin(g[0], g[1]), fmin(g[2], g[3])); };
        auto block_any = [&](bool f) __attribute__((always_inline)) -> bool { lds_f64* g = meet(__ballot(f) != 0ull ? 1.0 : 0.0); return (g[0] + g[1]) + (g[2] + g[3]) != 0.0; };
        __syncthreads();               // (the verdict flags above have been read by every wave)
        // first guess: the inputs of the Newton point that violate their bounds
        double act[2], V[2] = {0.0, 0.0}, TL[2] = {1.0, 1.0}, TU[2] = {1.0, 1.0}, LL[2] = {0.0, 0.0}, LU[2] = {0.0, 0.0}, DVA[2] = {0.0, 0.0}, GM[2] = {0.0, 0.0},
               RTC[2] = {0.0, 0.0};
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int j = lane + 64 * t;
            const double vj = vh[j < nu ? j : 0];
            act[t] = vj < lbI - uo[t] ? -1.0 : (vj > ubI - uo[t] ? 1.0 : 0.0);
        }
        enum { M_TRY = 0, M_START, M_PRED, M_CORR, M_FINAL };
        const int nvt = 4 * N;
        const double inv2nv = 1.0 / (2.0 * nvt);
        int mode = M_TRY, iters = 0, round_k = 0, round_cap = POL_FIRST, nchg_prev = nvt + 1;
        double mu = 0.0, rho = 0.0, smu = 0.0, mu_gate = 1e300;
        bool ipm_on = false, converged = false, polished = false, give_up = false;
        status = BROV_STATUS_MAXITER;
#pragma clang loop unroll(disable)
        for (;;) {
            if (mode != M_CORR && mode != M_FINAL) {   // which Newton system is next (qp_body's loop head)
                if (iters >= P.qp_iter_max) mode = M_FINAL;
                else mode = round_k < round_cap ? M_TRY : (ipm_on ? M_PRED : M_START);
            }
            if (mode == M_FINAL && polished) break;
            {   // Gamma and right-hand side of this system, into the interior-point arrays the factor stage reads
                double* GAM = I.ipm + (size_t)IPM_GAM * I.nv + (size_t)s0 * 4;
                double* RT = I.ipm + (size_t)IPM_RT * I.nv + (size_t)s0 * 4;
                double ssum = 0.0;
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int j = lane + 64 * t;
                    const bool in = j < nu;
                    const double uj = uo[t], lb = lbI - uj, ub = ubI - uj;
                    const double rr = rd[t] * (uj - ur[t]);
                    double gm, rt;
                    if (mode == M_TRY) {
                        const double ac = act[t];
                        gm = ac != 0.0 ? POL_BIG : 0.0;
                        rt = rr - gm * (ac < 0.0 ? lb : ub);
                    } else if (mode == M_START) {
                        const double wdt = ub - lb;
                        double vj = vh[in ? j : 0];
                        const double lo = lb + IPM_TAU0 * wdt, hi = ub - IPM_TAU0 * wdt;
                        vj = (vj < lo) ? lo : vj;
                        vj = (vj > hi) ? hi : vj;
                        V[t] = vj; TL[t] = vj - lb; TU[t] = ub - vj;
                        gm = POL_BIG; rt = rr - POL_BIG * vj;
                    } else if (mode == M_PRED) {
                        if (in) ssum += LL[t] * TL[t] + LU[t] * TU[t];
                        gm = LL[t] / TL[t] + LU[t] / TU[t];
                        GM[t] = gm;
                        rt = rr - gm * V[t];
                    } else if (mode == M_CORR) {
                        gm = GM[t]; rt = RTC[t];
                    } else {   // M_FINAL: every input pinned at the answer
                        double vj = ipm_on ? V[t] : fmin(fmax((double)vh[in ? j : 0], lb), ub);
                        V[t] = vj;
                        gm = POL_BIG; rt = rr - POL_BIG * vj;
                    }
                    if (in) { GAM[j] = gm; RT[j] = rt; }
                }
                if (mode == M_TRY) {
                    iters++; round_k++;
                    bool deep = false;   // a pinned input behind the first segment?
#pragma unroll
                    for (int t = 0; t < 2; t++) deep = deep | ((lane + 64 * t < nu) & (act[t] != 0.0) & (wv >= 1));
                    bool high = deep;    // ... or at / behind the first segment's checkpoint stage?
#pragma unroll
                    for (int t = 0; t < 2; t++) high = high | ((lane + 64 * t < nu) & (act[t] != 0.0) & (wv == 0) & (lane + 64 * t >= 4 * ckst));
                    lds_f64* g2 = meet((__ballot(deep) != 0ull ? 1.0 : 0.0) + (__ballot(high) != 0ull ? 16.0 : 0.0));
                    const double tot = (g2[0] + g2[1]) + (g2[2] + g2[3]);
                    only0 = ((int)tot & 15) == 0;
                    clean = clean && tot == 0.0;
                    light_ok = clean && relay_cached && only0;
                } else { only0 = false; clean = false; light_ok = false; }
                if (mode == M_PRED) { iters++; mu = block_sum(ssum) * inv2nv; }
            }
            __syncthreads();   // (every wave is done with the hand-over buffers of the pass before)
            solve_pass(false);
            const bool pinned_all = mode == M_START || mode == M_FINAL;
            bool bad = false;
            if (mode == M_TRY || pinned_all) {
#pragma unroll
                for (int t = 0; t < 2; t++) {   // pinned inputs exactly onto their points; (tries) free inputs that leave the box are marked (+-2: to be pinned)
                    const int j = lane + 64 * t;
                    const double uj = uo[t], lb = lbI - uj, ub = ubI - uj;
                    double vj = vh[j < nu ? j : 0];
                    bad = bad | ((j < nu) & !(vj == vj));
                    if (pinned_all) vj = V[t];
                    else if (act[t] != 0.0) vj = act[t] < 0.0 ? lb : ub;
                    else act[t] = vj < lb ? -2.0 : (vj > ub ? 2.0 : 0.0);
                    lds_f64* o = j < nu ? vh + j : tr_w + 16;
                    *o = vj;
                }
                const bool seg_bad = __ballot(bad) != 0ull || seg_nan() || !good;
                seg_adjoint();   // multipliers of this point: its state steps are the forward sweep's (the snap of a pinned input is a rounding error)
                double gmx = 0.0;
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int j = lane + 64 * t;
                    gel[t] = I.lds_kff[j < nu ? j : 0];
                    if (j < nu) gmx = fmax(gmx, fabs(gel[t]));
                }
                gmx = block_max(gmx);
                if (block_any(seg_bad)) { give_up = true; break; }
                if (mode == M_FINAL) break;
                if (mode == M_START) {
                    const double mu0 = fmax(IPM_MU0F * gmx, 1e-4);
                    double r0 = 0.0;
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        LL[t] = mu0 / TL[t]; LU[t] = mu0 / TU[t];
                        if (lane + 64 * t < nu) r0 = fmax(r0, fabs(gel[t] - LL[t] + LU[t]));
                    }
                    rho = block_max(r0);
                    ipm_on = true;
                    continue;
                }
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
                const int nchg = (int)block_sum(cnt);
                if (nchg == 0) { polished = true; status = BROV_STATUS_SUCCESS; break; }
                if (nchg > POL_NCHG || (nchg > nchg_prev && ipm_on)) round_cap = 0;
                nchg_prev = nchg;
                if (round_k >= round_cap) {   // failed round: the next one waits until the interior-point loop has halved mu
                    if (ipm_on) mu_gate = mu;
                    if (converged) { mode = M_FINAL; }
                }
                continue;
            }
            if (block_any(!good || seg_nan())) { give_up = true; break; }
            if (mode == M_PRED) {   // group B: predictor step length, centring, corrector right-hand side
                double aaff = 1.0;
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const int j = lane + 64 * t;
                    const double ll = LL[t], lu = LU[t], tl = TL[t], tu = TU[t];
                    const double dv = (double)vh[j < nu ? j : 0] - V[t];
                    DVA[t] = dv;
                    const double dll = -ll - ll / tl * dv, dlu = -lu + lu / tu * dv;
                    if (j < nu) {
                        if (dv < 0) aaff = fmin(aaff, -tl / dv);
                        if (dv > 0) aaff = fmin(aaff, tu / dv);
                        if (dll < 0) aaff = fmin(aaff, -ll / dll);
                        if (dlu < 0) aaff = fmin(aaff, -lu / dlu);
                    }
                }
                aaff = block_min(aaff);
                double sa = 0.0;
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const double ll = LL[t], lu = LU[t], tl = TL[t], tu = TU[t], dv = DVA[t];
                    const double dll = -ll - ll / tl * dv, dlu = -lu + lu / tu * dv;
                    if (lane + 64 * t < nu) sa += (ll + aaff * dll) * (tl + aaff * dv) + (lu + aaff * dlu) * (tu - aaff * dv);
                }
                const double muaff = block_sum(sa) * inv2nv;
                double sigma = muaff / mu;
                sigma = sigma * sigma * sigma;
                smu = sigma * mu;
#pragma unroll
                for (int t = 0; t < 2; t++) {
                    const double ll = LL[t], lu = LU[t], tl = TL[t], tu = TU[t], dv = DVA[t];
                    const double dll = -ll - ll / tl * dv, dlu = -lu + lu / tu * dv;
                    const double cl_ = dll * dv, cu_ = -dlu * dv;
                    RTC[t] = rd[t] * (uo[t] - ur[t]) - GM[t] * V[t] - (smu - cl_) / tl + (smu - cu_) / tu;
                }
                mode = M_CORR;
                continue;
            }
            // M_CORR, group C: step length of the combined direction, update, classification of the bounds
            double amax = 1e300, dllv[2], dluv[2];
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int j = lane + 64 * t;
                const double ll = LL[t], lu = LU[t], tl = TL[t], tu = TU[t], dva = DVA[t];
                const double dlla = -ll - ll / tl * dva, dlua = -lu + lu / tu * dva;
                const double cl_ = dlla * dva, cu_ = -dlua * dva;
                const double dv = (double)vh[j < nu ? j : 0] - V[t];
                const double dll = (smu - cl_) / tl - ll - ll / tl * dv;
                const double dlu = (smu - cu_) / tu - lu + lu / tu * dv;
                if (j < nu) {
                    if (dv < 0) amax = fmin(amax, -tl / dv);
                    if (dv > 0) amax = fmin(amax, tu / dv);
                    if (dll < 0) amax = fmin(amax, -ll / dll);
                    if (dlu < 0) amax = fmin(amax, -lu / dlu);
                }
                dllv[t] = dll; dluv[t] = dlu;
            }
            amax = block_min(amax);
            double alpha;
            {
                const double a = amax < 1.0 ? amax : 1.0;
                alpha = (IPM_FTB * amax >= 1.0) ? 1.0 : a * ((1.0 - a) * IPM_FTBLO + a * IPM_FTB);
            }
            double s2 = 0.0, unres = 0.0;
            bool nanv = false;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int j = lane + 64 * t;
                const double dv = (double)vh[j < nu ? j : 0] - V[t];
                const double vj = V[t] + alpha * dv;
                const double tl = TL[t] + alpha * dv, tu = TU[t] - alpha * dv;
                const double ll = LL[t] + alpha * dllv[t], lu = LU[t] + alpha * dluv[t];
                V[t] = vj; TL[t] = tl; TU[t] = tu; LL[t] = ll; LU[t] = lu;
                if (j < nu) {
                    nanv = nanv | !(vj == vj);
                    s2 += ll * tl + lu * tu;
                    const double al = ll / rd[t], au = lu / rd[t];
                    unres = fmax(unres, fmax(fmin(tl, al), fmin(tu, au)));
                    act[t] = al > tl ? -1.0 : (au > tu ? 1.0 : 0.0);
                }
            }
            if (block_any(nanv)) { give_up = true; break; }
            rho *= (1.0 - alpha);
            mu = block_sum(s2) * inv2nv;
            unres = block_max(unres);
            if (unres <= P.tol_mu && rho <= P.tol_stat) converged = true;
            if (converged || (mu <= POL_MU_GATE * mu_gate && alpha >= POL_ALPHA_GATE)) { round_k = 0; round_cap = POL_LOOP; nchg_prev = nvt + 1; }
            mode = M_PRED;   // (the loop head decides: a round of tries, or the next iteration)
        }
        __syncthreads();   // (the last reduction's words have been read: the objective's shares go into one of those groups below)
        if (give_up) {
            if (threadIdx.x == 0) P.pit_done[bq] = 0;
            return;
        }
        if (converged && status == BROV_STATUS_MAXITER) status = BROV_STATUS_SUCCESS;
        tries = iters;
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
        pit_emit_record(P, bq, lane, lane == 0 ? ctot : 0.0, u0v, kkt, early ? 0 : tries, status);
        PIT_STAMP(5);
    }
    if (early) seg_adjoint();   // (an accepted try has run it already: its multipliers are the ones to keep)
    win_flush_small(pi_it, (const double*)I.lds_kt, nseg * NX, lane);
    if (threadIdx.x == 0) P.pit_done[bq] = 1;
    if (wv == 0) PIT_STAMP(6);
#undef PIT_STAMP
}

}  // namespace brov
