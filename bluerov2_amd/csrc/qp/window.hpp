// qp/window.hpp -- window manager of the windowed kernel (Win, win_*: LDS slice <-> parked HBM images) and the sw_* wrappers through which qp_body runs its sweeps window by window.
// Part of ONE translation unit: qp_kernel.hip includes these headers in layer order (tiles -> sweeps -> window -> sched -> qp_body ->
// lin_phase -> fused -> windowed -> pit) and instantiates the kernels between them; see the file map at the head of qp_kernel.hip.
#pragma once

namespace brov {

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
// what win_forward_fast hands back to qp_body (by value: as fields of Win -- which lives across the kernel's instance loop -- they would be live
// across every phase of every instance): the step-0 answer was accepted AND the full step taken window by window; this lane's share of the
// objective at the new iterate; lanes 0..3: its first input
struct WinFast { bool committed = false; double cost_lane = 0.0, u0_lane = 0.0; };
// per-block scratch behind the windows' images (I.Kt points at it in the windowed kernels): [0, 384) (P, p) entering window 0 / the checkpoint stage,
// [384, 896) resident mode: step-0 feed-forward terms + split header, then 4 x kSegPark of the split launches' quarters (kWinCk in all); round 6:
// (P, p) at every inner window boundary (384 each: as they enter window c - 1, c = 1 .. nc - 1) and the staged rows of win_forward_fast
constexpr int kSegPark = 704;
constexpr int kWinCk = 384 + 512 + 4 * kSegPark;
__host__ __device__ constexpr int win_stage_doubles(int N) { return (N + 1) * 12 + N * 4 + N * 12 + N * 8; }
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
#ifndef BROV_EXP_WIN_FUSE
#define BROV_EXP_WIN_FUSE 0
#endif
#if BROV_EXP_WIN_FUSE
// Round 6 (large-batch windowed kernel, verdict item 2) -- MEASURED AND NOT SHIPPED, kept as a development build (make EXTRA=-DBROV_EXP_WIN_FUSE=1;
// profiles/r6_window_traffic.txt has the numbers): forward sweep, bound check, ADJOINT SWEEP AND FULL STEP of a window while it is resident.
// The regular schedule walks the windows 0 .. nc-1 forward (fetching [A B] | b | K' | kff of all but the first), learns that the step-0 answer is
// inside the box, and walks back nc-1 .. 0 for the multipliers and the step -- fetching [A B] | q | r of all but the last AGAIN, with the state steps
// and inputs going through HBM in between.  Here the adjoint sweep of window c runs right behind its forward sweep: it does not need its neighbour,
// because the costate at the window's last node IS the gradient of the cost-to-go there, lambda = P x + p with the (P, p) the factor sweep carried
// across that boundary in pass 1 (parked per boundary, 3 KB) -- the algebra the parallel-in-time kernel uses between its segments:
//     A' pi entering the window's sweep  :=  (P dx_e + p) - (Q dx_e + q_e).
// The step of a window cannot be taken before the LAST window has passed the bound check, so the new rows of x, u, pi, lambda of the windows before
// it are STAGED in the block's scratch and copied into the iterate when it has (the last window commits directly); an answer that leaves the box in
// window c simply stops the extra work -- windows c .. nc-1 get the plain forward sweep, the iterate was never touched, qp_body goes on as before.
// What it saves per early-exit solve at nc = 4: three prefix fetches (172 L doubles each) and their waits, against 36 doubles per stage staged and
// copied.  The multipliers differ from the plain recursion's by rounding (the boundary relation is exact in exact arithmetic).
template <class IT>
__device__ __forceinline__ WinFast win_forward_fast(const DevParams& P, IT& I, Win& W, int b, const d4& d0) {
    WinFast out;
    const double* __restrict__ cst = P.cst;
    opaque_lane(I);
    wave_fence();
    const int lane = I.lane, NT = I.NT, L = W.Lc, nc = W.nc, rg = I.rg;
    double* x_it = P.x + (size_t)b * (NT + 1) * 12;
    double* u_it = P.u + (size_t)b * NT * 4;
    double* pi_it = P.pi + (size_t)b * NT * 12;
    double* lam_it = P.lam + (size_t)b * NT * 8;
    const double* bnd = I.Kt + kWinCk;
    double* sx = I.Kt + kWinCk + (size_t)(nc - 1) * 384;
    double* su = sx + (size_t)(NT + 1) * 12;
    double* spi = su + (size_t)NT * 4;
    double* slam = spi + (size_t)NT * 12;
    const lds_f64* vh = (const lds_f64*)(W.lds + win_off_vh(L));
    const lds_f64* dx = (const lds_f64*)(W.lds + win_off_dx(L));
    const lds_f64* ql = (const lds_f64*)(W.lds + win_off_q(L));
    d4 xx = d0;
    bool bad = false, infeas = false;
    bool fast = true;   // wave-uniform: every window so far inside the box, nothing NaN
    double cost = 0.0, u0v = 0.0;
    const double lbm = cst[32 + (lane & 3)], ubm = cst[36 + (lane & 3)];
    for (int c = 0; c < nc; c++) {
        const bool last = c == nc - 1;
        const int i0 = c * L, n = (NT - i0 < L) ? NT - i0 : L;
        const int nu = n * 4, nxr = (last ? n + 1 : n) * NX;
        // the window's rows of the iterate and of the reference: requested before the window is fetched and swept
        double uo[2], ur[2], wu[2], xo[4], yr[4], wx[4];
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int j = lane + 64 * t, jj = j < nu ? j : 0;
            uo[t] = u_it[i0 * 4 + jj];
            ur[t] = I.yref[(size_t)(i0 + (jj >> 2)) * 16 + 12 + (jj & 3)];
            wu[t] = IT::kGrid ? I.wst[(size_t)(i0 + (jj >> 2)) * 16 + 12 + (jj & 3)] : P.Ts * cst[12 + (jj & 3)];
        }
        if (fast) {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int j = lane + 64 * t, jj = j < nxr ? j : 0;
                const int i = div12(jj), cc = jj - i * 12;
                xo[t] = x_it[i0 * 12 + jj];
                yr[t] = I.yref[(size_t)(i0 + i) * 16 + cc];
                wx[t] = IT::kGrid ? I.wst[(size_t)(i0 + i) * 16 + cc] : ((i0 + i == NT) ? cst[16 + cc] : P.Ts * cst[cc]);
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; t++) { xo[t] = 0.0; yr[t] = 0.0; wx[t] = 0.0; }
        }
        win_need(I, W, c, (fast || last) ? (WM_LIN | WM_GAIN) : (WM_AB | WM_BV | WM_GAIN), nullptr);
        fwd_chunk<3>(I, xx);
        __syncthreads();
        win_flush_small(I.vhat + I.i0 * 4, W.lds + win_off_vh(L), I.N * 4, lane);
        win_flush_small(I.dxb + I.i0 * NX, W.lds + win_off_dx(L), (I.N + 1) * NX, lane);
        W.valid |= WM_DX;
        bad = bad | win_nan_check<false>(I, W, c == 0);
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const int j = lane + 64 * t;
            const double vj = vh[j < nu ? j : 0], lb = lbm - uo[t], ub = ubm - uo[t];
            infeas = infeas | ((j < nu) & !(vj >= lb && vj <= ub));
        }
        if (fast) fast = __ballot(bad | infeas) == 0ull;
        if (fast) {
            // costate at the window's last node -> A' pi entering the sweep (row-replicated); zero behind the terminal node
            d4 atpi = {0, 0, 0, 0};
            if (!last) {
                const double* ck = bnd + (size_t)c * 384;
                d4 Pc, pr;
#pragma unroll
                for (int r = 0; r < 3; r++) { Pc[r] = ck[r * 64 + lane]; pr[r] = dpp_f64<0x150>(ck[192 + r * 64 + lane]); }   // (p sits in column 0: row_newbcast:0)
                Pc[3] = 0.0; pr[3] = 0.0;
                const d4 lam = tn<3>(Pc, xx, pr);
#pragma unroll
                for (int r = 0; r < 3; r++) {
                    const int row = rg + 4 * r;
                    const double qd = IT::kGrid ? I.wst[(size_t)(i0 + n) * 16 + row] : I.Ts * I.Wr[r];
                    atpi[r] = lam[r] - fma(qd, xx[r], (double)ql[n * NX + row]);
                }
            }
            adj_chunk<true, 3>(I, atpi, nullptr, nullptr, nullptr);
            W.valid &= ~WM_GAIN;   // multipliers / input gradient are staged in the K^T / feed-forward areas
            __syncthreads();
            win_flush_small((last ? pi_it : spi) + (size_t)i0 * NX, W.lds + win_off_kt(L), n * NX, lane);
            double* xd = last ? x_it : sx;
            double* ud = last ? u_it : su;
            double* ld = last ? lam_it : slam;
#pragma unroll
            for (int t = 0; t < 2; t++) {
                const int j = lane + 64 * t;
                if (j < nu) {
                    const int i = j >> 2, m = j & 3;
                    ld[(size_t)(i0 + i) * 8 + m] = 0.0;        // (an answer inside the box: no bound multipliers)
                    ld[(size_t)(i0 + i) * 8 + 4 + m] = 0.0;
                    const double un = uo[t] + vh[j];
                    ud[i0 * 4 + j] = un;
                    if (i0 == 0 && j < 4) u0v = un;
                    const double e = un - ur[t];
                    cost += 0.5 * wu[t] * e * e;
                }
            }
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const int j = lane + 64 * t;
                if (j < nxr) {
                    const double xn = xo[t] + dx[j];
                    xd[i0 * 12 + j] = xn;
                    const double e = xn - yr[t];
                    cost += 0.5 * wx[t] * e * e;
                }
            }
        }
    }
    W.nan = __ballot(bad) != 0ull;
    W.feas = __ballot(infeas) == 0ull;
    if (fast) {
        // every window passed: the staged rows of the windows 0 .. nc-2 into the iterate (all loads requested before the first store)
        wave_fence();
        const int ns = (nc - 1) * L;   // staged stages
        auto copy_rows = [&](double* dst, const double* src, int nd) __attribute__((always_inline)) {
            for (int o0 = 0; o0 < nd; o0 += 1024) {
                dbl2 v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int o = o0 + (k * 64 + lane) * 2;
                    v[k] = *(const dbl2*)(src + (o < nd ? o : 0));
                }
#pragma unroll
                for (int k = 0; k < 8; k++) asm volatile("" : "+v"(v[k].x), "+v"(v[k].y));
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int o = o0 + (k * 64 + lane) * 2;
                    if (o < nd) *(dbl2*)(dst + o) = v[k];
                }
            }
        };
        copy_rows(x_it, sx, ns * 12);
        copy_rows(u_it, su, ns * 4);
        copy_rows(pi_it, spi, ns * 12);
        copy_rows(lam_it, slam, ns * 8);
        if (lane < 4) P.res[b].u0[lane] = u0v;
        out.committed = true;
        out.cost_lane = cost;
        out.u0_lane = u0v;
    }
    wave_fence();
    return out;
}
#endif   // BROV_EXP_WIN_FUSE
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

}  // namespace brov
