// qp/lin_phase.hpp -- the wave-wide linearisation (lin_phase: ERK4 + forward sensitivities of all intervals of a chunk at once), its staging helpers and the streaming path's lin_wave_body.
// Part of ONE translation unit: qp_kernel.hip includes these headers in layer order (tiles -> sweeps -> window -> sched -> qp_body ->
// lin_phase -> fused -> windowed -> pit) and instantiates the kernels between them; see the file map at the head of qp_kernel.hip.
#pragma once

namespace brov {

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
                                          double* rec_s, double* q_s, double* r_s, double& part, bool& nanp, bool stamp, size_t yoff = 0) {   // yoff: rti_fused_kernel_ticks, the window of the step in hand
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
        stage_issue(P.yref + yoff + (size_t)b * P.yref_stride + (size_t)i0 * NY, (n + 1) * NY, lane, vy);
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

}  // namespace brov
