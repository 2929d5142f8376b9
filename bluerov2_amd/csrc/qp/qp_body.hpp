// qp/qp_body.hpp -- qp_body: everything after the linearisation -- step-0 solve, active-set tries around the interior-point loop, multiplier recovery, full step, result record (shared by all kernels); setup_inst.
// Part of ONE translation unit: qp_kernel.hip includes these headers in layer order (tiles -> sweeps -> window -> sched -> qp_body ->
// lin_phase -> fused -> windowed -> pit) and instantiates the kernels between them; see the file map at the head of qp_kernel.hip.
#pragma once

namespace brov {

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
// LONGV (rti_window_kernel_long, 128 < N <= 256): the interior-point vectors of the windowed kernel without their per-group register copies --
// 16 elements a lane, element by element out of HBM as in the streaming kernel (the copies of 8 a lane are what limits the other LDS-resident
// kernels to nv <= 512)
template <int LDS, class IT = Inst, bool DF = false, bool LONGV = false>
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
#if BROV_EXP_WIN_FUSE
    WinFast wfast;   // (large-batch windowed kernel: see win_forward_fast)
#endif
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
    // (not in the two-waves-per-SIMD kernel of the short horizons, N <= 11: its 256 registers do not hold the second pivot form
    // without scratch, which the build forbids in a solver kernel)
    constexpr bool ROB = LDS != 2;
    bool robust = false, robust_ok = false;
    if constexpr (ROB) {
        // Only while the step is numerically meaningful (entering KKT <= 1e6, the bound of the parity rules): the iterate of a diverged
        // full-step SQP is ill-conditioned without end, and with pivots that never fail its interior-point loop grinds through all
        // qp_iter_max systems (measured: 50 instead of the 1..19 after which the fast form gives up or fails -- one such instance
        // made its whole launch 2.6 times as long).
        robust_ok = kkt <= P.robust_kkt_max || P.robust_pivot == 3;   // (3: development knob -- no limit: the oracle's behaviour, profiles/r5_status_direction.txt)
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
        // (large-batch windowed kernel, development build -DBROV_EXP_WIN_FUSE=1: forward sweep, check, adjoint sweep and full step in ONE pass over
        // the windows when the answer stays inside the box -- win_forward_fast in qp/window.hpp, measured and not shipped)
        bool fast_pass = false;
#if BROV_EXP_WIN_FUSE
        if constexpr (LDS == 3) fast_pass = P.early_exit != 0 && !robust;
        if constexpr (LDS == 3) { if (fast_pass) wfast = win_forward_fast(P, I, *W, b, d0); }
#endif
        if (!fast_pass) sw_forward<LDS>(I, W, d0, cst);
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
            constexpr bool CACHE = (LDS >= 3) && !LONGV;   // windowed kernel: register copies per loop group (IpmVec MODE 2)
            constexpr int kIpmT = EL ? 2 : ((LDS == 0 || LONGV) ? 16 : 8);   // elements per lane; windowed kernels: nv <= 512 (register copies per loop group); streaming kernel (vectors in HBM): nv <= 1024
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
#if BROV_EXP_WIN_FUSE
            } else if (LDS == 3 && wfast.committed) {   // win_forward_fast has taken the step window by window
                cost = wfast.cost_lane; u0v = wfast.u0_lane;
                wrote_u0 = true;
#endif
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
                const int i = div12(jj), c = jj - i * 12;
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
                    const int i = div12(jj), c = jj - i * 12;
                    xo[t] = (EL && j0 == lane) ? xpre[t] : x_it[jj];
                    dj[t] = (EL && j0 == lane) ? djp[EL ? t : 0] : rd_dxb(jj);
                    yr[t] = (EL && j0 == lane) ? ypre[t] : I.yref[(size_t)i * 16 + c];
                }
#pragma unroll
                for (int t = 0; t < UX; t++) {
                    const int j = j0 + 64 * t;
                    if (j < nxe) {
                        const int i = div12(j), c = j - i * 12;
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
    I.ran_loop = !early;   // (rti_fused_kernel_ticks: the instance's own hint for its next step; dead everywhere else)
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
__device__ __forceinline__ void setup_inst(const DevParams& P, Inst& I, int b, int lane, const LaneCst* pre = nullptr, size_t yoff = 0) {
    const int N = P.N, nv = 4 * N;
    const double* __restrict__ cst = P.cst;
    I.lane = lane; I.rg = lane >> 4; I.cl = lane & 15; I.N = N; I.nv = nv;
    I.i0 = 0; I.NT = N; I.ckpt = 0;
    I.x = P.x + (size_t)b * (N + 1) * 12;
    I.u = P.u + (size_t)b * N * 4;
    I.yref = P.yref + yoff + (size_t)b * P.yref_stride;
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

}  // namespace brov
