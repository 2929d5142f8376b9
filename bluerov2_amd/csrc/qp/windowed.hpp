// qp/windowed.hpp -- rti_window_body: horizons beyond the LDS slice window by window (persistent blocks), its resident mode (one window = the horizon, small batches) and the rti_phase 1 / 2 split launches.
// Part of ONE translation unit: qp_kernel.hip includes these headers in layer order (tiles -> sweeps -> window -> sched -> qp_body ->
// lin_phase -> fused -> windowed -> pit) and instantiates the kernels between them; see the file map at the head of qp_kernel.hip.
#pragma once

namespace brov {

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
__host__ __device__ inline size_t win_ws_doubles(int N, int L) {
    return (size_t)((N + L - 1) / L) * win_img_doubles(L)                  // parked window images
           + (size_t)N * 4 + (size_t)(N + 1) * NX                          // vhat, dx (flat over the horizon)
           + (size_t)N * (64 + 64 + NX) + (size_t)IPM_NARR * 4 * N         // Ks Mt Pb | interior-point vectors
           + kWinCk                                                        // (P, p) entering window 0 (resident mode: stage ckpt): checkpoint of the partial
                                                                           // refactorisation; resident mode: + the step-0 feed-forward terms (4 N <= 512)
           + (size_t)((N + L - 1) / L) * 384 + win_stage_doubles(N);       // development build BROV_EXP_WIN_FUSE: (P, p) at the inner window boundaries, staged rows (win_forward_fast)
}
// RES: resident mode -- one window = the whole horizon (N <= 81) in a slice of up to 160 KB, one block per CU; for batches of at most
// one instance per CU.  Nothing is parked and no window is fetched.  A separate instantiation (rti_window_kernel_res), so that the
// large-batch kernel carries none of its code.
// SPLIT (resident mode only): acados' rti_phase 1 / 2 as two launches (DevParams::rti_split).  The whole backward sweep -- P, p, gains,
// feed-forward terms -- is independent of the measured state (x0 enters with dx_0 = x0 - x_0 in the forward roll-out only), so the
// PREPARATION launch linearises, factorises and parks the LDS image in the instance's workspace, and the FEEDBACK launch fetches it and runs
// qp_body from the forward sweep on: what is left between the arrival of a measurement and u0 is the forward sweep, the bound check, the
// step and the record.  Separate instantiations (rti_window_kernel_res_split, _split_grid).
// MULTI (rti_window_kernel_ticks; large batches only): P.ticks RTI steps of an instance back to back once a block has taken it from the counter --
// brov_solve_ticks / brov_closed_loop at the horizons the fused kernels do not serve (see MULTI in qp/fused.hpp)
// LONG (rti_window_kernel_long): horizons beyond BROV_MAX_N_LDS = 128 on the large-batch kernel (see qp_body's LONGV)
template <bool RES, bool GRID = false, bool SPLIT = false, bool MULTI = false, bool LONG = false>
__device__ __forceinline__ void rti_window_body(const DevParams& P) {
    static_assert(!LONG || (!RES && !SPLIT), "long horizons: the large-batch kernel");
    static_assert(!SPLIT || RES, "the split launches exist for the resident mode");
    static_assert(!MULTI || (!RES && !SPLIT && !GRID), "steps in one launch: the large-batch kernel on the uniform grid");
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
        // (steps of a MULTI launch: a backward jump rather than a loop statement around the body -- with a loop, even one of a single trip, the
        // single-step kernels compile 2 - 3 % slower: measured, scripts/dev/ab_libs.sh, N = 40 / 80)
        int tk = 0;
    next_step:
        const size_t yoff = MULTI ? (size_t)tk * (size_t)P.tick_yref : 0;
        if constexpr (MULTI) {   // (opaque per step, as in the instance loop: nothing lane- or instance-dependent is hoisted across the steps)
            asm volatile("v_mov_b32 %0, %0" : "+v"(lane));
            asm volatile("s_mov_b32 %0, %0" : "+s"(b));
        }
        auto setup = [&](InstT& I) __attribute__((always_inline)) {
            setup_inst(P, I, b, lane, &lc, yoff);
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
                yq = P.yref[yoff + (size_t)b * P.yref_stride + (size_t)(i0 + n) * NY + lane];
                wq = GRID ? P.wst[(size_t)(i0 + n) * 16 + lane] : P.Ts * P.cst[lane];   // scaled state weight of stage i0 + n
            }
            __syncthreads();
            if (RES && trip == 0) {
                // first instance of the block: this wave takes the first quarter of the horizon, waves 1..3 the others
                const int lsub = (n + 3) >> 2;
                lin_phase<true, GRID>(P, b, 0, lsub, lane, ba_s, bv_s, kt_s, q_s, r_s, part, nanp, false, yoff);
                __syncthreads();
                for (int wv = 1; wv < 4; wv++) {
                    const double v = ((const lds_f64*)kt_s)[(size_t)wv * lsub * kRecInterval + lane];
                    nanp = nanp | !(v == v);
                    part = fmax(part, v);
                }
            } else if (!RES || n <= kLinMaxIntervals) {
                lin_phase<true, GRID>(P, b, i0, n, lane, ba_s, bv_s, kt_s, q_s, r_s, part, nanp, false, yoff);
            } else {
                // resident mode (one window = the whole horizon in a 160 KB slice, small batches): the wave-wide linearisation takes
                // at most 23 intervals at a time -- sub-chunks, each into its own part of the slice (row n_j of a sub-chunk's q is
                // row 0 of the next one's: contiguous)
                const int nsub = (n + kWinMaxStages - 1) / kWinMaxStages, lsub = (n + nsub - 1) / nsub;
                for (int j0 = 0; j0 < n; j0 += lsub) {
                    const int nj = n - j0 < lsub ? n - j0 : lsub;
                    lin_phase<true, GRID>(P, b, i0 + j0, nj, lane, ba_s + (size_t)j0 * kBaStage, bv_s + (size_t)j0 * NX, kt_s, q_s + (size_t)j0 * NX,
                                          r_s + (size_t)j0 * NU, part, nanp, false, yoff);
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
#if BROV_EXP_WIN_FUSE
            if (!RES && c >= 1) {   // round 6: (P, p) as they cross the boundary into window c - 1 (win_forward_fast: the costate there)
                double* bd = ws_ck + kWinCk + (size_t)(c - 1) * 384;
#pragma unroll
                for (int r = 0; r < 3; r++) { bd[r * 64 + lane] = S.P[r]; bd[192 + r * 64 + lane] = S.pv[r]; }
            }
#endif
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
        qp_body<(RES ? 4 : 3), InstT, false, LONG>(P, I, b, part, nanp, &W, S.ok, S.illc);
#endif
#ifdef BROV_DBG_WIN
        if (P.dbg && lane == 0) { P.dbg[(size_t)P.B * 8 + (size_t)b * 8 + 3] = W.t_fetch; P.dbg[(size_t)P.B * 8 + (size_t)b * 8 + 4] = W.n_fetch; }
#endif
        __syncthreads();
        if constexpr (MULTI) {
            if (P.tick_status && lane == 0) P.tick_status[(size_t)tk * P.B + b] = P.res[b].status;   // (lane 0 wrote the record itself)
            wave_fence();
            if (P.plant_pp) {      // brov_closed_loop: the plant moves on with the step's first input
                plant_step_wave(P, b, lane, tk);
                __syncthreads();
                wave_fence();
            }
            if (++tk < P.ticks) goto next_step;
        }
    }
}

}  // namespace brov
