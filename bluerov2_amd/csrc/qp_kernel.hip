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
// File map -- ONE translation unit, in layers (round 5: the 4 100-line file of rounds 1-4 cut into headers along its own map; the
// cut is textual, the device code is byte-identical, profiles/r5_split_isa.txt):
//     qp/tiles.hpp      tile primitives, per-stage operand access, the instance record (Inst)
//     qp/sweeps.hpp     bwd_* / fwd_* / roll_* / adj_*: the Riccati sweeps, each as an initialisation and a "stages of the resident window" part
//     qp/window.hpp     window manager of the windowed kernel (Win, win_*) and the sw_* wrappers qp_body calls
//     qp/sched.hpp      interior-point vectors (IpmVec), work ordering (sched_*)
//     qp/qp_body.hpp    qp_body: QP solve, multiplier recovery, full step, record -- shared by all kernels; setup_inst
//     qp/lin_phase.hpp  lin_phase (wave-wide linearisation), lin_wave_body
//     qp/fused.hpp      rti_fused_body                      qp/windowed.hpp  rti_window_body (windowed, resident, split)
//     qp/pit.hpp        rti_pit_body (parallel in time)
// and this file: the __global__ instantiations between the layers they need (qp_kernel + lin_wave_kernel[_grid] streaming pair,
// rti_fused_kernel / _w2 / _grid / _mail, rti_window_kernel[_grid], its resident mode rti_window_kernel_res[_grid][_split], rti_pit_kernel[_fb][_grid]),
// their launchers and two test hooks.
#include <mutex>
#include <string>
#include <type_traits>

#include "lin_device.hpp"
#include "nmpc_device.hpp"

#include "qp/tiles.hpp"
#include "qp/sweeps.hpp"
#include "qp/window.hpp"
#include "qp/sched.hpp"
#include "qp/qp_body.hpp"

namespace brov {
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
}  // namespace brov

#include "qp/lin_phase.hpp"

namespace brov {
__global__ __launch_bounds__(64, 1) void lin_wave_kernel(DevParams P) { lin_wave_body<false>(P); }
// the same on a general grid: per-interval time steps, per-stage scaled weights (DevParams::tsv / wst)
__global__ __launch_bounds__(64, 1) void lin_wave_kernel_grid(DevParams P) { lin_wave_body<true>(P); }
}  // namespace brov

#include "qp/fused.hpp"

namespace brov {
// One wave per SIMD (up to 512 VGPRs): the variant for horizons whose LDS slice admits only four blocks per CU anyway.
__global__ __launch_bounds__(64, 1) void rti_fused_kernel(DevParams P) { rti_fused_body<1>(P); }
// Two waves per SIMD (256 VGPRs): short horizons (N <= 11, seven blocks per CU by LDS), where the second wave fills the first one's
// MFMA / LDS / dependent-issue waits.  (Until round 6: N <= 13, six blocks.  With the round-6 factor stage -- which the 256-register
// form only takes in part, qp/sweeps.hpp kR6Z -- the one-wave kernel is ahead from N = 12: scripts/dev/w2_crossover.sh, N = 10 / 11 / 12
// / 13 / 14: 43.4 / 38.9 / 34.9 / 32.8 / 29.3 M solves/s on two waves against 38.6 / 37.6 / 36.3 / 34.1 / 33.3 M on one.)
__global__ __launch_bounds__(64, 2) void rti_fused_kernel_w2(DevParams P) { rti_fused_body<2>(P); }
// General grid (per-stage time steps / a separate stage-0 weight), every N <= 23: one wave per SIMD
__global__ __launch_bounds__(64, 1) void rti_fused_kernel_grid(DevParams P) { rti_fused_body<1, true>(P); }
// brov_tick_host at small batches (host mailbox): the record of an early exit goes out ahead of the adjoint sweep (qp_body<.., DF>)
__global__ __launch_bounds__(64, 1) void rti_fused_kernel_mail(DevParams P) { rti_fused_body<1, false, true>(P); }
// brov_solve_ticks: P.ticks steps per instance in one launch (see MULTI in qp/fused.hpp)
__global__ __launch_bounds__(64, 1) void rti_fused_kernel_ticks(DevParams P) { rti_fused_body<1, false, false, true>(P); }
// (N <= 11: the code of rti_fused_kernel_w2 -- qp_body<2> -- so that the steps are bit-identical to single launches; compiled for one wave per
// SIMD, because the step loop's few live values no longer fit the 256 registers of the two-wave form without scratch)
__global__ __launch_bounds__(64, 1) void rti_fused_kernel_ticks_w2(DevParams P) { rti_fused_body<2, false, false, true>(P); }

}  // namespace brov

#include "qp/windowed.hpp"

namespace brov {
__global__ __launch_bounds__(64, 1) void rti_window_kernel(DevParams P) { rti_window_body<false>(P); }
// the same on a general grid: per-interval time steps, per-stage scaled weights (DevParams::tsv / wst)
__global__ __launch_bounds__(64, 1) void rti_window_kernel_grid(DevParams P) { rti_window_body<false, true>(P); }
// brov_solve_ticks / brov_closed_loop at N >= 24, large batches: P.ticks steps per instance in one launch (see MULTI in qp/windowed.hpp)
__global__ __launch_bounds__(64, 1) void rti_window_kernel_ticks(DevParams P) { rti_window_body<false, false, false, true>(P); }
__global__ __launch_bounds__(256, 1) void rti_window_kernel_res(DevParams P) { rti_window_body<true>(P); }
__global__ __launch_bounds__(256, 1) void rti_window_kernel_res_grid(DevParams P) { rti_window_body<true, true>(P); }
// rti_phase 1 / 2 as separate launches in the resident mode (see SPLIT above)
__global__ __launch_bounds__(256, 1) void rti_window_kernel_res_split(DevParams P) { rti_window_body<true, false, true>(P); }
__global__ __launch_bounds__(256, 1) void rti_window_kernel_res_split_grid(DevParams P) { rti_window_body<true, true, true>(P); }
}  // namespace brov

#include "qp/pit.hpp"

namespace brov {
__global__ __launch_bounds__(256, 1) void rti_pit_kernel_fb(DevParams P) { rti_pit_body<false, true>(P); }
__global__ __launch_bounds__(256, 1) void rti_pit_kernel_fb_grid(DevParams P) { rti_pit_body<true, true>(P); }
__global__ __launch_bounds__(256, 1) void rti_pit_kernel(DevParams P) { rti_pit_body<false>(P); }
// the same on a general grid: per-interval time steps, per-stage scaled weights (DevParams::tsv / wst)
__global__ __launch_bounds__(256, 1) void rti_pit_kernel_grid(DevParams P) { rti_pit_body<true>(P); }
// 128 < N <= 256 (round 5): the large-batch windowed kernel with its interior-point vectors element by element out of HBM (qp_body's LONGV);
// defined behind the other kernels, whose device code stays byte for byte what it was
__global__ __launch_bounds__(64, 1) void rti_window_kernel_long(DevParams P) { rti_window_body<false, false, false, false, true>(P); }
__global__ __launch_bounds__(64, 1) void rti_window_kernel_long_grid(DevParams P) { rti_window_body<false, true, false, false, true>(P); }
__global__ __launch_bounds__(64, 1) void rti_window_kernel_long_ticks(DevParams P) { rti_window_body<false, false, false, true, true>(P); }

// The dynamic-LDS limit of every solver kernel (anything above 64 KB has to be asked for, per function and PER DEVICE), set ONCE per device and
// CHECKED: brov_create calls this with its device current, so a runtime that refuses a limit fails the create with its reason instead of the
// first launch with "invalid argument" -- and no launcher carries a lazy "first launch" flag any more.  Round 5's flags were plain statics,
// flipped BEFORE the attribute calls ran: a second host thread making its first launch on the same device (one rank object per thread,
// tests/test_gpu_group_loopback.py) could launch with > 64 KB ahead of the attribute.  std::call_once blocks that thread until the calls are over.
int prepare_kernels_on_device(std::string* why) {
    struct KernelLimit { const void* fn; int bytes; const char* name; };
#define BROV_KL(k, kb) {(const void*)k, kb * 1024, #k}
    static const KernelLimit limits[] = {
        BROV_KL(lin_wave_kernel, 64), BROV_KL(lin_wave_kernel_grid, 64),
        BROV_KL(rti_fused_kernel, 160), BROV_KL(rti_fused_kernel_w2, 160), BROV_KL(rti_fused_kernel_grid, 160), BROV_KL(rti_fused_kernel_mail, 160),
        BROV_KL(rti_fused_kernel_ticks, 160), BROV_KL(rti_fused_kernel_ticks_w2, 160),
        BROV_KL(rti_window_kernel, 160), BROV_KL(rti_window_kernel_grid, 160), BROV_KL(rti_window_kernel_ticks, 160),
        BROV_KL(rti_window_kernel_res, 160), BROV_KL(rti_window_kernel_res_grid, 160), BROV_KL(rti_window_kernel_res_split, 160),
        BROV_KL(rti_window_kernel_res_split_grid, 160),
        BROV_KL(rti_window_kernel_long, 160), BROV_KL(rti_window_kernel_long_grid, 160), BROV_KL(rti_window_kernel_long_ticks, 160),
        BROV_KL(rti_pit_kernel, 160), BROV_KL(rti_pit_kernel_grid, 160), BROV_KL(rti_pit_kernel_fb, 160), BROV_KL(rti_pit_kernel_fb_grid, 160),
    };
#undef BROV_KL
    constexpr int kMaxDev = 64;
    static std::once_flag once[kMaxDev];
    static std::string failure[kMaxDev];   // written inside call_once, read after it: ordered by call_once itself
    auto set_all = [](std::string& err) {
        for (const KernelLimit& k : limits) {
            const hipError_t e = hipFuncSetAttribute(k.fn, hipFuncAttributeMaxDynamicSharedMemorySize, k.bytes);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                err = std::string("hipFuncSetAttribute(") + k.name + ", MaxDynamicSharedMemorySize, " + std::to_string(k.bytes) + "): " + hipGetErrorString(e);
                return;
            }
        }
    };
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { if (why) *why = "hipGetDevice failed"; return BROV_ERR_HIP; }
    std::string local;
    if (dev >= 0 && dev < kMaxDev) {
        std::call_once(once[dev], [&] { set_all(failure[dev]); });
        local = failure[dev];
    } else {
        static std::mutex mu;   // (a device index beyond the table: no cache, one thread at a time)
        std::lock_guard<std::mutex> g(mu);
        set_all(local);
    }
    if (!local.empty()) { if (why) *why = local; return BROV_ERR_HIP; }
    return BROV_OK;
}

void launch_linearise(const DevParams& P, hipStream_t st) {
    const int C = lin_chunk_len(P.N);
    const size_t lds = ((size_t)C * (kBaStage + NX + kRecInterval + NU) + (size_t)(C + 1) * NX + 64) * sizeof(double);
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
// since round 5 the parallel kernel runs the whole QP loop itself (qp/pit.hpp), so what it leaves are only the instances it gives up on
// (BROV_PIT_ROUNDS=0 keeps the windowed kernel altogether).  Decided per solve: the solver is created for the windowed kernel and with a
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
    // development knob, read once per process: the long-horizon instantiations (interior-point vectors without register copies) at every horizon (A/B)
    static const bool force_long = getenv("BROV_DEV_WIN_LONG") && atoi(getenv("BROV_DEV_WIN_LONG")) != 0;
    const bool long_h = P.N > BROV_MAX_N_LDS || force_long;
    if (windowed_resident(P.win_L) || P.rti_split) {   // (the split launches are the resident mode's at every horizon)
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
    else if (P.ticks > 0) {
        if (long_h) hipLaunchKernelGGL(rti_window_kernel_long_ticks, dim3(P.win_blocks), dim3(64), windowed_lds_bytes(P.win_L), st, P);
        else hipLaunchKernelGGL(rti_window_kernel_ticks, dim3(P.win_blocks), dim3(64), windowed_lds_bytes(P.win_L), st, P);
    }
    else if (P.tsv && long_h) hipLaunchKernelGGL(rti_window_kernel_long_grid, dim3(P.win_blocks), dim3(64), windowed_lds_bytes(P.win_L), st, P);
    else if (P.tsv) hipLaunchKernelGGL(rti_window_kernel_grid, dim3(P.win_blocks), dim3(64), windowed_lds_bytes(P.win_L), st, P);
    else if (long_h) hipLaunchKernelGGL(rti_window_kernel_long, dim3(P.win_blocks), dim3(64), windowed_lds_bytes(P.win_L), st, P);
    else hipLaunchKernelGGL(rti_window_kernel, dim3(P.win_blocks), dim3(64), windowed_lds_bytes(P.win_L), st, P);
}
bool windowed_is_resident(int win_L) { return windowed_resident(win_L); }
// the split launches (rti_phase 1 / 2) of the resident kernel at a horizon the FUSED kernels serve in one call (N <= 23): the four quarters the
// block's waves linearise must all hold a stage
bool split_resident_horizon(int N) { return N >= 4 && 3 * ((N + 3) >> 2) < N && windowed_lds_bytes(N) <= 160 * 1024; }


static size_t fused_lds_bytes(int N) { return ((size_t)N * (kBaStage + NX + kKtStage + 4 + 4 + 4) + 2 * (size_t)(N + 1) * NX + 2 + 17) * sizeof(double); }
static bool fused_two_wave(size_t lds, int force) {   // force: DevKnobs::fused_waves (development knob)
    return force ? force == 2 : 7 * lds <= 160 * 1024;   // N <= 11 (seven slices per CU)
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
    } else {
        lds = fused_lds_bytes(N);
        const bool w2 = fused_two_wave(lds, k.fused_waves);
        fn = w2 ? (const void*)rti_fused_kernel_w2 : (const void*)rti_fused_kernel;
        kind = w2 ? 2 : 1;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, threads, lds) != hipSuccess) per_cu = -1;
    info[0] = (int32_t)lds; info[1] = per_cu; info[2] = threads; info[3] = kind;
}
void launch_fused_ticks(const DevParams& P, hipStream_t st, const DevKnobs& k) {
    const size_t lds = fused_lds_bytes(P.N);
    // the variant a single step of this solver runs (two waves per SIMD for short horizons): the steps are then the same code on the same data
    if (fused_two_wave(lds, k.fused_waves)) hipLaunchKernelGGL(rti_fused_kernel_ticks_w2, dim3(P.B), dim3(64), lds + (size_t)k.lds_pad, st, P);
    else hipLaunchKernelGGL(rti_fused_kernel_ticks, dim3(P.B), dim3(64), lds + (size_t)k.lds_pad, st, P);
}
void launch_fused(const DevParams& P, hipStream_t st, const DevKnobs& k) {
    const size_t lds = fused_lds_bytes(P.N);
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
