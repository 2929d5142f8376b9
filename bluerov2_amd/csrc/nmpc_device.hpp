// nmpc_device.hpp -- device-side parameter block and HBM layout shared by the kernels and the host API.
#pragma once
#include <string>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bluerov2_nmpc.h"

namespace brov {

// HBM layout (all FP64, instance-major; "tile" = 16x16 doubles stored row-major = MFMA C/D register image,
// see qp_kernel.hip):
//   x0    [B][12]             yref [B][N+1][16] or shared [N+1][16]        par [B][N+1][16]
//   x     [B][N+1][12]        u    [B][N][4]      pi [B][N][12]            lam [B][N][8]
//   BA    [B][N][12][16]      row k, col c:  d x+_k / d (x,u)_c            (3/4 tile: rows 0..11)
//   bvec  [B][N][12]          phi(x_i,u_i) - x_{i+1}
//   kktp  [B][N]              per-interval partial of the NLP KKT inf-norm
//   Riccati by-products per stage:  Ks [B][N][4][16] (gain, row m, col c), Kt [B][N][12][16] (gain^T, row c, col m),
//                                   Mt [B][N][4][16] (Huu^-1, cols 0..3), Pb [B][N][12], kff [B][N][4]
//   IPM state per instance:         ipm [B][IPM_NARR][N*4]
struct DevParams {
    int32_t B, N;
    int32_t qp_iter_max, early_exit;
    int32_t pit;             // parallel-in-time step-0 solve ahead of the resident windowed kernel: 0 off, 1 instances whose previous step was an early exit, 2 every instance (tests)
    int32_t pit_light;       // ... tries whose pins sit below the first segment's checkpoint reuse the step-0 pass (default 1; BROV_PIT_LIGHT=0: A/B, tests)
    int32_t pit_try;         // ... and, when the step-0 answer leaves the box, ONE active-set try parallel in time as well (default 1; BROV_PIT_TRY=0: A/B)
    int32_t pit_blocks;      // blocks of rti_pit_kernel = tickets it serves (win_blocks; B where every instance has a workspace of its own: pit_rounds_stages)
    int32_t* pit_done;       // [B]: rti_pit_kernel has completed the instance's step (the resident kernel behind it skips it); nullptr when pit = 0
    double robust_kkt_max;   // robust pivot form on demand only while the entering KKT is at most this (1e6; BROV_ROBUST_KKT_MAX: development knob)
    int32_t partial_refactor, robust_pivot;   // robust_pivot: ill-conditioned instances refactorise in the Cholesky pivot form (default 1; BROV_ROBUST_PIVOT=0: A/B);   // active-set tries restart their factor sweep from the step-0 checkpoint where they may (default 1; BROV_PARTIAL_REFACTOR=0: A/B)
    int32_t rti_split;       // resident windowed kernel, rti_phase 1 / 2 as separate launches: 1 = preparation (linearise + step-0 factor sweep, LDS image parked per instance), 2 = feedback (image fetched, everything that depends on x0); 0 = one launch
    int32_t on_failure, dump_lin;   // BROV_ON_FAILURE_*; dump_lin != 0: LDS-resident kernels copy [A B | b] out to BA / bvec (tests)
    double Ts, tol_mu, tol_stat;
    double W[16], We[12], lbu[4], ubu[4];
    // inputs
    const double* x0;
    const double* yref;
    int64_t yref_stride;  // doubles between instances (0 when one window is shared)
    const double* par;    // always [B][N+1][16]
    const double* par_rp; // 6-disturbance model variant: roll / pitch disturbance moments [B][N+1][2]; nullptr = the shipped model
    // iterate
    double* x;
    double* u;
    double* pi;
    double* lam;
    // linearisation
    double* BA;
    double* bvec;
    double* kktp;
    // Riccati storage
    double* Ks;
    double* Kt;
    double* Mt;
    double* Pb;
    double* kff;
    double* vhat;   // [B][N][4] candidate inputs of the current Newton solve
    double* ipm;    // [B][IPM_NARR][N*4]
    double* dxb;    // [B][N+1][12] QP primal step of the states
    const double* cst;  // [W16 | We12 pad4 | lbu4 | ubu4]
    // general grid (streaming kernels only; both nullptr = uniform step Ts and one stage weight): tsv [N] time steps = ERK4 step and
    // cost scaling per stage (acados_solver_bluerov2.c:111-131), wst [N+1][16] = the scaled weights per stage, ts_i * (i == 0 ? W_0 : W),
    // row N = [We | 0] (separate stage-0 weight: :422-441)
    const double* tsv;
    const double* wst;
    brov_result* res;
    // host mailbox (brov_tick_host, small batches): the record additionally goes straight into pinned host memory, followed by the
    // instance's sequence word -- the host polls that word instead of waiting for a copy and a stream synchronisation
    brov_result* mail;
    int32_t* mail_flag;
    int32_t mail_seq;
    int32_t mail_early;      // resident windowed kernel: send the record ahead of the last adjoint sweep when the answer needs none of it
    // windowed kernel (N >= 24): per-block parking image + scratch, instance hand-out counter, stages per window
    double* ws;
    int64_t ws_stride;   // doubles per block
    int32_t* counter;        // this launch's hand-out counter (zero on entry) ...
    int32_t* counter_next;   // ... and the next launch's, which block 0 of this launch zeroes (no memset node between the launches)
    int32_t win_L, win_blocks;
    // work ordering (qp_kernel.hip, sched_map): three rotating buffers of [64 class counters | class lists | pos[B]] int32 -- the
    // instances whose QP had active bounds in the previous solve are handed out first in this one; nullptr = instances in index order
    int32_t* sched;
    int32_t sched_stride, sched_r, sched_w, sched_z;   // buffer read / written / zeroed by this launch
    // rti_fused_kernel_ticks (brov_solve_ticks): `ticks` RTI steps of an instance back to back inside one launch; the shared reference window
    // moves on tick_yref doubles per step (row stride x 16 inside the resident trajectory table; 0: the window stays); optional status log
    int32_t ticks;
    int64_t tick_yref;
    int32_t* tick_status;    // [ticks][B] or nullptr
    // ... and, for brov_closed_loop, the plant update behind every step: x0 <- ERK4(x0, u0 of the step, plant parameters) (nullptr: no plant, x0 held)
    const double* plant_pp;  // [B][16] true plant parameters
    const double* plant_rp;  // 6-disturbance variant: roll / pitch moments, instance b at plant_rp + b * plant_rp_stride (or nullptr)
    int32_t plant_rp_stride, plant_substeps;
    double plant_dt;
    double* x0_rw;           // = x0, writable
    double* plant_xlog;      // [ticks][B][12] states after every step, or nullptr
    double* plant_ulog;      // [ticks][B][4] applied inputs, or nullptr
    unsigned long long* dbg;  // optional per-instance phase timestamps (s_memtime), 8 slots per instance; nullptr = off
};

// development knobs (BROV_* environment variables), read once per solver by the host API (nmpc_api.hip, read_knobs)
struct DevKnobs {
    double robust_kkt_max = 1e6;
    int robust_pivot = 1, partial_refactor = 1, mail_early = 1, split_resident = 1, pit = 1, split_parallel = 1, pit_try = 1, pit_light = 1;
    int tick_mailbox = 1, tick_bulk = 1, tick_zerocopy = 1, sched = 1, force_windowed = 0, fused_waves = 0, lds_pad = 0, tick_breakdown = 0, closed_loop_fused = 1;
};

enum { IPM_V = 0, IPM_TL, IPM_TU, IPM_LL, IPM_LU, IPM_GAM, IPM_RT, IPM_DVA, IPM_ACT, IPM_NARR };

int sched_buffer_ints_host(int B);   // int32 per work-ordering buffer (three of them)
int prepare_kernels_on_device(std::string* why);   // dynamic-LDS limits of every solver kernel on the CURRENT device, once per device, checked (brov_create)
void launch_linearise(const DevParams& P, hipStream_t st);
void launch_qp(const DevParams& P, hipStream_t st);
void launch_fused(const DevParams& P, hipStream_t st, const DevKnobs& k);
void launch_fused_ticks(const DevParams& P, hipStream_t st, const DevKnobs& k);   // P.ticks steps per instance in one launch (uniform grid, N <= 23)  // linearise + QP in one kernel, stage blocks in LDS
bool fused_supported(int N);      // whole horizon fits the LDS slice (N <= 23)
// windowed LDS-resident kernel for longer horizons: persistent blocks (one wavefront each) that take instances from a counter
void launch_windowed(const DevParams& P, hipStream_t st);
void lds_kernel_info(int N, int win_L, bool windowed, int32_t info[4], const DevKnobs& k);   // LDS bytes per block, blocks per CU, threads, kernel kind
bool pit_supported(int N, int win_L);       // rti_pit_kernel ahead of the resident kernel
bool windowed_is_resident(int win_L);      // one window = the whole horizon (small batches)
int windowed_stage_count(int N, int B);   // stages per window (= N for batches of at most one instance per CU: resident mode)
int windowed_blocks(int N, int B, int L); // persistent blocks that will be launched on the current device
size_t windowed_ws_doubles(int N, int L); // per-block workspace
bool split_resident_horizon(int N);           // rti_phase 1 / 2 on the resident kernel's split launches at a fused-kernel horizon
int pit_rounds_stages(int N, int B);          // batches of up to two instances per CU: resident stage count for parallel-in-time rounds, or 0
void launch_window(const double* traj, int rows, const int* lines, int line0, int B, int N, int ncols, double* out, hipStream_t st);
void launch_plant(double* x0, const brov_result* res, const double* pplant, const double* prp, int rp_stride, int B, double dt, int substeps,
                  double* xlog, double* ulog, hipStream_t st);   // prp: roll / pitch disturbance moments, instance b at prp + b * rp_stride (or nullptr)
void launch_candidates(int kind, const double* p0, const double* p1, const double* phase, double t0, double dt, int B, int N,
                       double* out, hipStream_t st);

}  // namespace brov
