// qp/fused.hpp -- rti_fused_body: one wavefront owns one instance from the linearisation to the updated iterate, the whole horizon in LDS (N <= 23).
// Part of ONE translation unit: qp_kernel.hip includes these headers in layer order (tiles -> sweeps -> window -> sched -> qp_body ->
// lin_phase -> fused -> windowed -> pit) and instantiates the kernels between them; see the file map at the head of qp_kernel.hip.
#pragma once

namespace brov {

// fused path: ONE wavefront owns one OCP instance from linearisation to the updated iterate.  The wave first integrates
// all N intervals at once (64/N lanes per interval, lin_device.hpp) and leaves [A_i B_i] and b_i in its LDS slice
// (N <= kFusedMaxN: 4 waves x 40.5 KB per CU at N = 20), then runs the Riccati IPM on the LDS-resident stage blocks: they
// are read 3-4 times per Newton system and never touch HBM.  One 64-thread block per instance so that a long-running
// (interior-point) instance does not pin the LDS of three finished ones.
constexpr int kFusedMaxN = 23;
// MULTI (rti_fused_kernel_ticks, brov_solve_ticks): P.ticks RTI steps of the instance back to back -- linearise, QP, full step, again -- with the
// reference window moving on P.tick_yref doubles per step.  The same code on the same data as P.ticks launches (bit-identical records and
// iterates, tests/test_gpu_ticks.py); what changes is the scheduling: an instance goes on to its next step when ITS step is done, so a step
// with a slow instance (13 .. 47 Newton systems on the mixed batch) no longer holds the whole batch at a launch boundary.
template <int W, bool GRID, bool DF>
__device__ __forceinline__ bool rti_fused_step(const DevParams& P, int b, int lane, bool listed, size_t yoff) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int N = P.N;
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
    LaneCst lc;
    if constexpr (W == 1) lc = load_lane_cst(P.cst, lane);   // the two-wave variant has no registers to spare across lin_phase
    // ---- preparation: ERK4 + sensitivities of all N intervals at once (lin_phase below)
    double part = 0.0;
    bool nanp = false;
    lin_phase<W == 1, GRID>(P, b, 0, N, lane, ba_s, bv_s, kt_s, q_s, r_s, part, nanp, true, yoff);
    __syncthreads();  // single wave: orders the LDS writes above against the reads below
    if (P.dump_lin) copy_out_linearisation(P, b, 0, N, lane, ba_s, bv_s);
    std::conditional_t<GRID, InstGrid, Inst> I;
    setup_inst(P, I, b, lane, W == 1 ? &lc : nullptr, yoff);
    // partial refactorisation of the active-set tries (riccati_backward_tries): checkpoint stage = ceil(N / 4); off for horizons too
    // short to gain from it and for instances the previous solve did not list as expensive
#ifndef BROV_EXP_NO_SPLIT
    // (`listed` arrives in a vector register -- a plain load, requested ahead of the linearisation.  It is the same in every lane, and saying so
    // here keeps the checkpoint, hence the bounds of the factor sweep's stage loop, in scalar registers: with a vector bound the compiler
    // drives that loop, and every guard inside it, through exec masks)
    const bool listed_u = __builtin_amdgcn_readfirstlane((int)listed) != 0;
    I.ckpt = (N >= 8 && P.partial_refactor && listed_u) ? (N + 3) >> 2 : 0;
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
    return I.ran_loop;
}
// the plant update of brov_closed_loop inside the step loop (traj_kernel.hip, plant_kernel: the same arithmetic, one lane per instance there;
// here every lane of the instance's wave computes it -- a wave's issue slots cost the same for one lane as for sixty-four -- and lane 0 stores)
__device__ __forceinline__ void plant_step_wave(const DevParams& P, int b, int lane, int tk) {
    double x[NX], u[NU], k[NX], xs[NX], acc[NX];
    double* x0 = P.x0_rw + (size_t)b * NX;
#pragma unroll
    for (int j = 0; j < NX; j++) x[j] = x0[j];
#pragma unroll
    for (int j = 0; j < NU; j++) u[j] = P.res[b].u0[j];
    const ModelPar m = make_par(P.plant_pp + (size_t)b * NP);
    Wrench w = make_wrench(u);
    if (P.plant_rp) { w.k3 = P.plant_rp[(size_t)b * P.plant_rp_stride]; w.k4 = P.plant_rp[(size_t)b * P.plant_rp_stride + 1]; }
    const double h = P.plant_dt / P.plant_substeps;
    StagePoint sp;
    for (int s = 0; s < P.plant_substeps; s++) {
        model_f(x, w, m, k, sp);
#pragma unroll
        for (int j = 0; j < NX; j++) { acc[j] = x[j] + (h / 6.0) * k[j]; xs[j] = x[j] + 0.5 * h * k[j]; }
        model_f(xs, w, m, k, sp);
#pragma unroll
        for (int j = 0; j < NX; j++) { acc[j] += (h / 3.0) * k[j]; xs[j] = x[j] + 0.5 * h * k[j]; }
        model_f(xs, w, m, k, sp);
#pragma unroll
        for (int j = 0; j < NX; j++) { acc[j] += (h / 3.0) * k[j]; xs[j] = x[j] + h * k[j]; }
        model_f(xs, w, m, k, sp);
#pragma unroll
        for (int j = 0; j < NX; j++) x[j] = acc[j] + (h / 6.0) * k[j];
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < NX; j++) x0[j] = x[j];
        if (P.plant_xlog) {
#pragma unroll
            for (int j = 0; j < NX; j++) P.plant_xlog[((size_t)tk * P.B + b) * NX + j] = x[j];
        }
        if (P.plant_ulog) {
#pragma unroll
            for (int j = 0; j < NU; j++) P.plant_ulog[((size_t)tk * P.B + b) * NU + j] = u[j];
        }
    }
}

template <int W, bool GRID = false, bool DF = false, bool MULTI = false>
__device__ __forceinline__ void rti_fused_body(const DevParams& P) {
    const int b = __builtin_amdgcn_readfirstlane(sched_map(P, blockIdx.x));
    const int lane = threadIdx.x;
    bool listed = sched_listed(P, b);   // requested here, used after the linearisation
    if (blockIdx.x == 0) sched_zero_next(P, lane);
    DBG_STAMP(0);
    if constexpr (!MULTI) {
        (void)rti_fused_step<W, GRID, DF>(P, b, lane, listed, 0);
    } else {
#pragma clang loop unroll(disable)
        for (int tk = 0; tk < P.ticks; tk++) {
            // opaque copies per step: nothing derived from the instance or lane index is a loop invariant of the step loop (hoisted, the base
            // addresses of every phase would be live across all of them -- the register file is full, the build forbids scratch)
            int bq = b, lq = lane;
            asm volatile("s_mov_b32 %0, %0" : "+s"(bq));
            asm volatile("v_mov_b32 %0, %0" : "+v"(lq));
            // the instance's own history replaces the work ordering's list (which such a launch neither reads nor writes)
            listed = rti_fused_step<W, GRID, DF>(P, bq, lq, listed, (size_t)tk * (size_t)P.tick_yref);
            if (P.tick_status && lq == 0) P.tick_status[(size_t)tk * P.B + bq] = P.res[bq].status;   // (lane 0 wrote the record itself)
            __syncthreads();       // single wave: the step's stores (iterate, record) against the next step's loads
            wave_fence();
            if (P.plant_pp) {      // brov_closed_loop: the plant moves on with the step's first input; the next step measures the new state
                plant_step_wave(P, bq, lq, tk);
                __syncthreads();
                wave_fence();
            }
        }
    }
}

}  // namespace brov
