// nmpc_api.hip -- host side of the batched solver: owns the HBM state of B OCP instances and implements the C ABI of
// include/bluerov2_nmpc.h (each entry point there cites the reference call it replaces).  No CPU compute path: without a
// usable HIP device every entry point fails with BROV_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>
#include <atomic>
#include <chrono>
#include <vector>

#include "nmpc_device.hpp"

using namespace brov;

#define kTickMailboxMaxBatch 64          /* brov_tick_host: up to this many instances deliver their records through the host mailbox */

static thread_local std::string g_err;
extern "C" const char* brov_last_error(void) { return g_err.c_str(); }

#define HIPCHK(call)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (call);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            g_err = std::string(#call) + ": " + hipGetErrorString(e_);                                 \
            (void)hipGetLastError(); /* reported here: not left behind as the "last error" of a later call */ \
            return (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice || e_ == hipErrorInsufficientDriver) \
                       ? BROV_ERR_NO_DEVICE                                                            \
                       : BROV_ERR_HIP;                                                                 \
        }                                                                                              \
    } while (0)

struct brov_solver {
    int device = 0, B = 0, N = 0;
    brov_opts opts{};
    bool yref_shared = false;
    const double* yref_view = nullptr;   // shared window = rows of the resident trajectory table, used in place (no copy)
    int traj_line = -1, traj_ncols = 0;  // the shared window in force was built from this row of the table by brov_set_yref_from_traj (-1: by something else)
    // device buffers
    double *x0 = nullptr, *yref = nullptr, *yref_sh = nullptr, *par = nullptr;
    double *x = nullptr, *u = nullptr, *pi = nullptr, *lam = nullptr;
    double *BA = nullptr, *bvec = nullptr, *kktp = nullptr;
    double *Ks = nullptr, *Kt = nullptr, *Mt = nullptr, *Pb = nullptr, *kff = nullptr, *vhat = nullptr, *ipm = nullptr,
           *dxb = nullptr, *cst = nullptr;
    brov_result* res = nullptr;
    int* best = nullptr;
    double* traj = nullptr;
    int traj_rows = 0;
    double* scratch3 = nullptr;  // [3][B] candidate parameters
    double* pplant = nullptr;    // [B][16] true plant parameters
    double* par_rp = nullptr;    // 6-disturbance variant: roll / pitch disturbance moments [B][N+1][2] (allocated by brov_enable_dist6)
    double* prp_plant = nullptr; // ... of the plant [B][2] (brov_plant_set_rp_disturbance_host); else the controller's stage 0
    bool dist6 = false, prp_plant_set = false;
    bool pplant_set = false;     // explicit plant parameters given (brov_plant_set_params_host)
    bool pplant_stale = true;    // controller parameters changed since the plant's copy of them was taken
    bool cand_set = false;       // candidate shape parameters resident in scratch3
    int cand_kind = 0;
    bool dump_lin = false;
    int* lines = nullptr;        // [B]
    size_t bytes = 0;
    std::vector<void*> allocs;
    hipStream_t last_stream = nullptr;
    bool timing = false;
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    bool ev_valid = false;
    bool last_fused = false, last_windowed = false;
    double* ws = nullptr;        // windowed kernel: per-block parking images
    int32_t* counter = nullptr;
    unsigned win_tick = 0;           // windowed launches so far: which of the two hand-out counters the next one uses
    int win_blocks = 0, win_L = 0;
    double* ws_split = nullptr;      // fused-kernel horizons, at most one instance per CU: per-instance workspace of the resident kernel's split launches (rti_phase 1 / 2)
    int alt_blocks = 0, alt_L = 0;   // parallel-in-time rounds (pit_rounds_stages): the resident configuration a solve may use instead
    int prep_path = 0;               // the last rti_phase-1 call: 1 streaming pair (linearisation in HBM), 2 resident split (factorised LDS image parked), 3 the latter, invalidated by a setter
    bool force_windowed = false;
    unsigned long long* dbg = nullptr;
    // general grid (streaming kernels): per-stage time steps and / or a separate stage-0 weight
    std::vector<double> ts_host;     // N time steps, empty = uniform
    double W0_host[16] = {0};
    bool has_W0 = false;
    double* tsv = nullptr;           // device [N]
    double* wst = nullptr;           // device [N+1][16] scaled weights per stage
    int32_t* sched = nullptr;        // work ordering: 3 rotating buffers of 64 class counters | lists | pos[B] (qp_kernel.hip, sched_map)
    unsigned sched_tick = 0;
    bool sched_on = true;
    hipStream_t tick_stream = nullptr;   // brov_tick_host: the solver's own stream and pinned staging buffer
    double* pin = nullptr;
    size_t pin_doubles = 0;
    // brov_tick_host, mailbox path: inputs passed with the tick are read by THIS launch straight from the pinned staging buffer (no copy
    // command ahead of the kernel); their device copies are refreshed behind the kernel.  Non-null only while that launch is built.
    const double *tick_x0 = nullptr, *tick_yref = nullptr, *tick_par = nullptr;
    hipEvent_t ev_up = nullptr;          // a preparation tick (rti_phase 1): its uploads have left the pinned staging buffer
    hipEvent_t ev_tick = nullptr;        // ticks with inputs read in place: the kernel's end (neither the host nor the next tick's kernel waits for the copies behind it)
    hipEvent_t ev_pre = nullptr;         // large batches: recorded AHEAD of the tick's kernel -- its refresh copies run next to the kernel, not behind it
    hipStream_t copy_stream = nullptr;   // ... those copies (pinned staging buffer -> the device arrays every other entry point works on) run here, behind ev_tick
    hipEvent_t ev_copy = nullptr;        // ... and end here (alias of the ev_set[] recorded last)
    hipEvent_t ev_set[2] = {nullptr, nullptr};   // the refresh copies out of input set 0 / 1 of the pinned staging buffer (see brov_tick_host)
    bool set_pending[2] = {false, false};
    int pin_sel = 0;                     // input set the last copying tick used
    bool buffers_out = false;            // brov_tick_buffers has handed set 0 out
    bool copies_pending = false;         // ev_copy recorded and not yet waited for by the host
    int copy_mask = 0;                   // ... which device arrays those copies write: 1 x0, 2 shared window, 4 stage parameters
    bool in_tick = false;                // brov_tick_host is calling brov_solve_phase (whose kernel reads the pinned inputs: no need to order it behind the copies)
    brov_result* mail = nullptr;     // host mailbox of the tick in flight (device-visible pinned memory), else nullptr
    int32_t* mail_flag = nullptr;
    bool pit_ran = false;            // the last solve launched rti_pit_kernel
    int32_t* pit_done = nullptr;     // [B]: written by rti_pit_kernel (parallel-in-time step-0 solve), read by the resident kernel launched behind it
    int32_t mail_seq = 0;
    double tick_us[5] = {0, 0, 0, 0, 0};   // BROV_TICK_BREAKDOWN=1: host time of the last brov_tick_host by part (brov_dev_tick_breakdown)
    DevKnobs k;                      // development knobs (BROV_* environment), read once in brov_create: no getenv on the path of a solve
};

// The solver's development knobs: A/B switches and test hooks, all of them BROV_* environment variables.  Read ONCE per solver (brov_create;
// brov_dev_reload_knobs re-reads them for tests that flip a switch between two solves of one solver) -- a linear scan of the environment
// per variable has no place inside a 29 us feedback call (round 4 did up to 16 of them per solve).
static int env_int(const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; }
static DevKnobs read_knobs() {
    DevKnobs k;
    k.robust_pivot = env_int("BROV_ROBUST_PIVOT", 1);            // 0 off, 1 on demand (default), 2 every instance, 3 on demand without the KKT <= 1e6 limit
    if (const char* v = getenv("BROV_ROBUST_KKT_MAX")) k.robust_kkt_max = atof(v);   // entering KKT up to which the robust form is taken on demand
    k.partial_refactor = env_int("BROV_PARTIAL_REFACTOR", 1) != 0;
    k.mail_early = env_int("BROV_DEV_NO_EARLY_RECORD", 0) == 0;
    k.split_resident = env_int("BROV_SPLIT_RESIDENT", 1) != 0;
    k.pit = env_int("BROV_PIT", 1);                               // 0 off, 1 product rule, 2 every instance is tried
    k.split_parallel = env_int("BROV_SPLIT_PARALLEL", 1) != 0;
    k.pit_try = env_int("BROV_PIT_TRY", 1) != 0;
    k.pit_light = env_int("BROV_PIT_LIGHT", 1) != 0;
    k.tick_mailbox = env_int("BROV_TICK_MAILBOX", 1) != 0;
    k.tick_bulk = env_int("BROV_TICK_BULK", 1) != 0;
    k.tick_zerocopy = env_int("BROV_TICK_ZEROCOPY", 1) != 0;
    k.sched = env_int("BROV_SCHED", 1) != 0;
    k.force_windowed = env_int("BROV_DEV_FORCE_WINDOWED", 0) != 0;
    k.fused_waves = env_int("BROV_DEV_FUSED_WAVES", 0);           // 1 / 2: force a variant of the fused kernel (default by LDS size)
    k.lds_pad = env_int("BROV_DEV_LDS_PAD", 0);
    k.closed_loop_fused = env_int("BROV_CLOSED_LOOP_FUSED", 1) != 0;   // 0: brov_closed_loop as three launches per tick (A/B, tests)
    k.tick_breakdown = env_int("BROV_TICK_BREAKDOWN", 0) != 0;
    return k;
}
extern "C" int brov_dev_reload_knobs(brov_solver* s) {
    if (!s) return BROV_ERR_ARG;
    const bool fw = s->force_windowed;
    s->k = read_knobs();
    s->k.force_windowed = fw;        // (the workspaces were allocated for the create-time choice)
    s->sched_on = s->k.sched != 0;
    return BROV_OK;
}

extern "C" void brov_default_opts(brov_opts* o, int N, double Ts) {
    // /root/reference/bluerov2_dobmpc/scripts/c_generated_code/acados_solver_bluerov2.c:422-481 (W), :559-566 (bounds), :668
    static const double W[16] = {300, 480, 200, 10, 10, 200, 40, 40, 10, 10, 10, 10, 1, 1, 0.1, 0.05};
    std::memset(o, 0, sizeof(*o));
    o->N = N;
    o->Ts = Ts;
    for (int j = 0; j < 16; j++) o->W[j] = W[j];
    for (int j = 0; j < 12; j++) o->We[j] = W[j];
    for (int j = 0; j < 4; j++) { o->lbu[j] = -50.0; o->ubu[j] = 50.0; }
    o->qp_iter_max = 50;
    o->qp_tol_mu = 1e-7;
    o->qp_tol_stat = 1e-9;
    o->qp_early_exit = 1;
    o->kernel_path = BROV_PATH_AUTO;
    o->on_failure = BROV_ON_FAILURE_RESTART;
}

// what is wrong with a set of options (nullptr = nothing).  The QP must be strictly convex in the inputs (R > 0), the box must
// have an interior (the interior-point start divides by its width), limits and tolerances must be usable.
static const char* opts_problem(const brov_opts* o) {
    if (o->N < 1 || o->N > BROV_MAX_N) return "N out of range";
    if (!(o->Ts > 0.0) || !(o->Ts < 1e6)) return "Ts must be positive and finite";
    if (o->kernel_path < 0 || o->kernel_path > 2) return "kernel_path must be BROV_PATH_AUTO / _STREAMING / _FUSED";
    if (o->on_failure < 0 || o->on_failure > 1) return "on_failure must be BROV_ON_FAILURE_KEEP / _RESTART";
    for (int j = 0; j < 16; j++)
        if (!(o->W[j] >= 0.0) || !(o->W[j] < 1e300)) return "stage weights must be finite and >= 0";
    for (int j = 12; j < 16; j++)
        if (!(o->W[j] > 0.0)) return "input weights W[12..15] must be > 0 (strictly convex QP)";
    for (int j = 0; j < 12; j++)
        if (!(o->We[j] >= 0.0) || !(o->We[j] < 1e300)) return "terminal weights must be finite and >= 0";
    for (int j = 0; j < 4; j++)
        if (!(o->lbu[j] < o->ubu[j]) || !(o->lbu[j] > -1e300) || !(o->ubu[j] < 1e300)) return "input bounds need lbu < ubu, both finite";
    if (o->qp_iter_max < 1) return "qp_iter_max must be >= 1";
    if (!(o->qp_tol_mu > 0.0) || !(o->qp_tol_stat > 0.0)) return "qp tolerances must be > 0";
    return nullptr;
}

template <typename T>
static int dalloc(brov_solver* s, T** p, size_t n) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, n * sizeof(T));
    if (e != hipSuccess) {
        g_err = std::string("hipMalloc: ") + hipGetErrorString(e);
        return BROV_ERR_ALLOC;
    }
    s->allocs.push_back(q);
    s->bytes += n * sizeof(T);
    *p = (T*)q;
    return BROV_OK;
}

static bool general_grid(const brov_solver* s) { return !s->ts_host.empty() || s->has_W0; }
// per-stage time steps and scaled weights of the general grid: wst[i] = ts_i * (i == 0 ? W_0 : W) for i < N, wst[N] = [We | 0]
static int upload_grid(brov_solver* s) {
    if (!general_grid(s)) return BROV_OK;
    const int N = s->N;
    if (!s->tsv) {
        if (int rc = dalloc(s, &s->tsv, (size_t)N)) return rc;
        if (int rc = dalloc(s, &s->wst, (size_t)(N + 1) * 16)) return rc;
    }
    std::vector<double> ts(N), w((size_t)(N + 1) * 16, 0.0);
    for (int i = 0; i < N; i++) {
        ts[i] = s->ts_host.empty() ? s->opts.Ts : s->ts_host[i];
        const double* Wi = (i == 0 && s->has_W0) ? s->W0_host : s->opts.W;
        for (int j = 0; j < 16; j++) w[(size_t)i * 16 + j] = ts[i] * Wi[j];
    }
    for (int j = 0; j < 12; j++) w[(size_t)N * 16 + j] = s->opts.We[j];
    HIPCHK(hipMemcpy(s->tsv, ts.data(), ts.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->wst, w.data(), w.size() * sizeof(double), hipMemcpyHostToDevice));
    return BROV_OK;
}
static int upload_cst(brov_solver* s) {
    double c[40];
    std::memset(c, 0, sizeof c);
    for (int j = 0; j < 16; j++) c[j] = s->opts.W[j];
    for (int j = 0; j < 12; j++) c[16 + j] = s->opts.We[j];
    for (int j = 0; j < 4; j++) { c[32 + j] = s->opts.lbu[j]; c[36 + j] = s->opts.ubu[j]; }
    HIPCHK(hipMemcpy(s->cst, c, sizeof c, hipMemcpyHostToDevice));
    return upload_grid(s);
}

extern "C" int brov_init_iterate_default(brov_solver* s) {
    if (s && s->prep_path == 2) s->prep_path = 3;   // a parked preparation (rti_phase 1, resident kernel) does not survive this call
    if (!s) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    const int B = s->B, N = s->N;
    std::vector<double> hx((size_t)B * (N + 1) * 12, 0.0);
    for (size_t k = 0; k < (size_t)B * (N + 1); k++) hx[k * 12 + 2] = -20.0;  // acados_solver_bluerov2.c:689-706
    HIPCHK(hipMemcpy(s->x, hx.data(), hx.size() * sizeof(double), hipMemcpyHostToDevice));
    HIPCHK(hipMemset(s->u, 0, (size_t)B * N * 4 * sizeof(double)));
    HIPCHK(hipMemset(s->pi, 0, (size_t)B * N * 12 * sizeof(double)));
    HIPCHK(hipMemset(s->lam, 0, (size_t)B * N * 8 * sizeof(double)));
    return BROV_OK;
}

extern "C" int brov_create(brov_solver** out, int device, int B, const brov_opts* opts) {
    if (!out || !opts || B < 1) {
        g_err = "brov_create: bad argument";
        return BROV_ERR_ARG;
    }
    if (const char* why = opts_problem(opts)) {
        g_err = std::string("brov_create: ") + why;
        return BROV_ERR_ARG;
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1 || device < 0 || device >= ndev) {
        g_err = "brov_create: no usable HIP device (this library has no CPU fallback)";
        return BROV_ERR_NO_DEVICE;
    }
    HIPCHK(hipSetDevice(device));
    {   // the kernels' dynamic-LDS limits on this device (once per device, checked): a refusal fails the create, not the first launch
        std::string why;
        if (int prc = prepare_kernels_on_device(&why)) { g_err = "brov_create: " + why; return prc; }
    }
    brov_solver* s = new brov_solver();
    s->device = device;
    s->B = B;
    s->N = opts->N;
    s->opts = *opts;
    const size_t N = opts->N, Bz = B;
    int rc = BROV_OK;
#define AL(ptr, n) if (rc == BROV_OK) rc = dalloc(s, &s->ptr, (n))
    // x0 | shared reference window | stage parameters in ONE allocation, in the order of brov_tick_host's staging buffer: a tick that
    // rewrites all three (the ROS node does) uploads them with one copy
    AL(x0, Bz * 12 + (size_t)(N + 1) * 16 + Bz * (N + 1) * 16);
    if (rc == BROV_OK) { s->yref_sh = s->x0 + Bz * 12; s->par = s->yref_sh + (size_t)(N + 1) * 16; }
    AL(yref, Bz * (N + 1) * 16);
    AL(x, Bz * (N + 1) * 12);
    AL(u, Bz * N * 4);
    AL(pi, Bz * N * 12);
    AL(lam, Bz * N * 8);
    AL(BA, Bz * N * 192);
    AL(bvec, Bz * N * 12);
    AL(kktp, Bz * N);
    AL(Ks, Bz * N * 64);
    AL(Kt, Bz * N * 192);
    AL(Mt, Bz * N * 64);
    AL(Pb, Bz * N * 12);
    AL(kff, Bz * N * 4);
    AL(vhat, Bz * N * 4);
    AL(ipm, Bz * IPM_NARR * N * 4);
    AL(dxb, Bz * (N + 1) * 12);
    AL(cst, 40);
    AL(res, Bz);
    AL(best, 2);
    AL(scratch3, 3 * Bz);
    AL(lines, Bz);
    AL(pplant, Bz * 16);
    AL(counter, 64);   // two hand-out counters of the windowed kernel, 128 bytes apart, used alternately
    AL(pit_done, Bz);   // rti_pit_kernel's per-instance verdict
    AL(sched, 3 * (size_t)sched_buffer_ints_host(B));
    // development knob: BROV_DEV_FORCE_WINDOWED=1 runs the windowed kernel for every horizon (one window when N <= 20)
    s->k = read_knobs();
    s->force_windowed = s->k.force_windowed != 0;
    // BROV_PATH_AUTO beyond the horizons the fused kernels serve: the windowed kernel at every batch size -- its resident mode where the whole horizon
    // fits one window (N <= 81, at most one instance per CU), windows of <= 20 stages otherwise.  (Rounds 3-4 sent up to eight instances at N > 81 to
    // the streaming pair, then as fast; since round 5 the windowed kernel is ahead there too -- one instance at N = 82 / 128 / 160 / 256: 0.196 / 0.278 /
    // 0.342 / 0.535 ms per step against 0.207 / 0.301 / 0.366 / 0.558, eight instances 0.198 / 0.281 / 0.344 / 0.539 against 0.241 / 0.343 / 0.421 /
    // 0.686: scripts/dev/long_horizon_small_batches.py.)
    if ((!fused_supported(opts->N) || s->force_windowed) && opts->kernel_path != BROV_PATH_STREAMING) {
        s->win_L = windowed_stage_count(opts->N, B);
        s->win_blocks = windowed_blocks(opts->N, B, s->win_L);
        size_t ws_doubles = (size_t)s->win_blocks * windowed_ws_doubles(opts->N, s->win_L);
        if (!windowed_is_resident(s->win_L) && (s->alt_L = pit_rounds_stages(opts->N, B)) != 0) {   // one workspace per instance for rti_pit_kernel's blocks
            s->alt_blocks = windowed_blocks(opts->N, B, s->alt_L);
            const size_t alt = (size_t)B * windowed_ws_doubles(opts->N, s->alt_L);
            ws_doubles = alt > ws_doubles ? alt : ws_doubles;
        }
        AL(ws, ws_doubles);
    }
    // rti_phase 1 / 2 at a horizon the fused kernels serve: the resident kernel's split launches need a workspace per instance
    if (fused_supported(opts->N) && !s->force_windowed && opts->kernel_path != BROV_PATH_STREAMING && split_resident_horizon(opts->N) &&
        s->k.split_resident) {
        int dev_ = 0, cus_ = 256;
        (void)hipGetDevice(&dev_);
        (void)hipDeviceGetAttribute(&cus_, hipDeviceAttributeMultiprocessorCount, dev_);
        if (B <= cus_) {
            AL(ws_split, (size_t)B * windowed_ws_doubles(opts->N, opts->N));
        }
    }
#undef AL
    if (rc != BROV_OK) { brov_destroy(s); return rc; }
    // create defaults: yref = 0, p = 0, x0 = [0,0,-20,0..] (acados_solver_bluerov2.c:355-364, 405-420, 520-527)
    hipMemset(s->yref, 0, Bz * (N + 1) * 16 * sizeof(double));
    hipMemset(s->yref_sh, 0, (N + 1) * 16 * sizeof(double));
    hipMemset(s->par, 0, Bz * (N + 1) * 16 * sizeof(double));
    hipMemset(s->res, 0, Bz * sizeof(brov_result));
    hipMemset(s->sched, 0, 3 * (size_t)sched_buffer_ints_host(B) * sizeof(int32_t));
    hipMemset(s->counter, 0, 64 * sizeof(int32_t));
    // development knob: BROV_SCHED=0 hands the instances out in index order (A/B of the work ordering)
    s->sched_on = s->k.sched != 0;
    {
        std::vector<double> h0(Bz * 12, 0.0);
        for (size_t k = 0; k < Bz; k++) h0[k * 12 + 2] = -20.0;
        hipMemcpy(s->x0, h0.data(), h0.size() * sizeof(double), hipMemcpyHostToDevice);
    }
    rc = upload_cst(s);
    if (rc == BROV_OK) rc = brov_init_iterate_default(s);
    if (rc != BROV_OK) { brov_destroy(s); return rc; }
    for (int k = 0; k < 3; k++) hipEventCreate(&s->ev[k]);
    *out = s;
    return BROV_OK;
}

extern "C" void brov_destroy(brov_solver* s) {
    if (!s) return;
    hipSetDevice(s->device);
    // nothing of this solver may still be in flight when its memory goes: the tail of a tick's kernel, the input copies behind it
    if (s->tick_stream) hipStreamSynchronize(s->tick_stream);
    if (s->copy_stream) hipStreamSynchronize(s->copy_stream);
    // (a caller's own stream is the caller's to drain -- it may not exist any more; hipFree below waits for the device in any case)
    for (void* p : s->allocs) hipFree(p);
    if (s->traj) hipFree(s->traj);
    if (s->dbg) hipFree(s->dbg);
    if (s->pin) hipHostFree(s->pin);
    if (s->copy_stream) hipStreamDestroy(s->copy_stream);
    if (s->ev_tick) hipEventDestroy(s->ev_tick);
    if (s->ev_pre) hipEventDestroy(s->ev_pre);
    if (s->ev_up) hipEventDestroy(s->ev_up);
    for (int k = 0; k < 2; k++) if (s->ev_set[k]) hipEventDestroy(s->ev_set[k]);
    if (s->tick_stream) hipStreamDestroy(s->tick_stream);
    for (int k = 0; k < 3; k++)
        if (s->ev[k]) hipEventDestroy(s->ev[k]);
    delete s;
}

extern "C" int brov_batch(const brov_solver* s) { return s ? s->B : 0; }
extern "C" int brov_horizon(const brov_solver* s) { return s ? s->N : 0; }
extern "C" size_t brov_device_bytes(const brov_solver* s) { return s ? s->bytes : 0; }

// Stream ordering.  Everything a solver enqueues runs on the stream its caller names, and brov_tick_host uses the solver's own
// non-blocking stream and (mailbox path) returns while the tail of its kernel is still running.  A call that arrives on ANOTHER
// stream than the one last used would overlap that work and race on the iterate / work-ordering buffers / hand-out counters: the
// host waits for the earlier stream first.  Same stream (every loop in bench.py, the closed loop, tick after tick): a pointer
// comparison, no cost.
// host-side wait for everything the solver has in flight: the last stream, and the input copies a tick left running on the copy stream
static hipError_t sync_last(brov_solver* s) {
    hipError_t e = hipStreamSynchronize(s->last_stream);
    if (e == hipSuccess && s->copies_pending) { e = hipEventSynchronize(s->ev_copy); s->copies_pending = false; }
    return e;
}
static int order_behind_last(brov_solver* s, hipStream_t st) {
    if (s->last_stream != st) HIPCHK(hipStreamSynchronize(s->last_stream));
    // a tick's input copies (copy stream) write the device arrays this stream's next command may read or overwrite
    if (s->copies_pending && !s->in_tick) HIPCHK(hipStreamWaitEvent(st, s->ev_copy, 0));
    return BROV_OK;
}
extern "C" int brov_order_stream(brov_solver* s, void* stream) {
    if (!s) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    return order_behind_last(s, (hipStream_t)stream);
}

static int copy_in(brov_solver* s, double* dst, const double* src, size_t n, bool host, void* stream) {
    if (!s || !src) { g_err = "null argument"; return BROV_ERR_ARG; }
    HIPCHK(hipSetDevice(s->device));
    if (host) {
        // a solve may still be running on the caller's (possibly non-blocking) stream: the blocking copy on the null stream does not
        // wait for such a stream by itself
        HIPCHK(sync_last(s));
        HIPCHK(hipMemcpy(dst, src, n * sizeof(double), hipMemcpyHostToDevice));
    } else {
        if (int rc = order_behind_last(s, (hipStream_t)stream)) return rc;
        HIPCHK(hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    }
    return BROV_OK;
}

extern "C" int brov_set_x0_host(brov_solver* s, const double* x0) { return copy_in(s, s ? s->x0 : nullptr, x0, s ? (size_t)s->B * 12 : 0, true, nullptr); }
extern "C" int brov_set_x0_device(brov_solver* s, const double* x0, void* st) { return copy_in(s, s ? s->x0 : nullptr, x0, s ? (size_t)s->B * 12 : 0, false, st); }

static const double* shared_window(const brov_solver* s) { return s->yref_view ? s->yref_view : s->yref_sh; }

static int set_yref(brov_solver* s, const double* y, int shared, bool host, void* st) {
    if (!s) return BROV_ERR_ARG;
    s->yref_view = nullptr; s->traj_line = -1;
    s->yref_shared = shared != 0;
    const size_t n = (size_t)(s->N + 1) * 16;
    return shared ? copy_in(s, s->yref_sh, y, n, host, st) : copy_in(s, s->yref, y, n * s->B, host, st);
}
extern "C" int brov_set_yref_host(brov_solver* s, const double* y, int shared) { return set_yref(s, y, shared, true, nullptr); }
extern "C" int brov_set_yref_device(brov_solver* s, const double* y, int shared, void* st) { return set_yref(s, y, shared, false, st); }

__global__ void bcast_par_kernel(const double* __restrict__ p16, double* __restrict__ par, int B, int N1) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t tot = (size_t)B * N1 * 16;
    if (t >= tot) return;
    const size_t b = t / ((size_t)N1 * 16);
    par[t] = p16[b * 16 + (t & 15)];
}

static int set_par(brov_solver* s, const double* p, int per_stage, bool host, void* st) {
    if (!s || !p) return BROV_ERR_ARG;
    s->pplant_stale = true;
    const size_t N1 = s->N + 1;
    if (per_stage) return copy_in(s, s->par, p, (size_t)s->B * N1 * 16, host, st);
    HIPCHK(hipSetDevice(s->device));
    const double* src = p;
    double* tmp = nullptr;
    if (!host) { if (int rc = order_behind_last(s, (hipStream_t)st)) return rc; }
    if (host) {
        HIPCHK(sync_last(s));
        HIPCHK(hipMalloc((void**)&tmp, (size_t)s->B * 16 * sizeof(double)));
        hipError_t e = hipMemcpy(tmp, p, (size_t)s->B * 16 * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) { hipFree(tmp); g_err = hipGetErrorString(e); return BROV_ERR_HIP; }
        src = tmp;
    }
    const size_t tot = (size_t)s->B * N1 * 16;
    hipLaunchKernelGGL(bcast_par_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)st, src, s->par, s->B, (int)N1);
    if (host) {
        hipStreamSynchronize((hipStream_t)st);
        hipFree(tmp);
    }
    HIPCHK(hipGetLastError());
    return BROV_OK;
}
extern "C" int brov_set_params_host(brov_solver* s, const double* p, int per_stage) { return set_par(s, p, per_stage, true, nullptr); }
extern "C" int brov_set_params_device(brov_solver* s, const double* p, int per_stage, void* st) { return set_par(s, p, per_stage, false, st); }

extern "C" int brov_set_param_stage_host(brov_solver* s, int inst, int stage, const double* p16) {
    if (!s || !p16 || inst < 0 || inst >= s->B || stage < 0 || stage > s->N) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    HIPCHK(hipMemcpy(s->par + ((size_t)inst * (s->N + 1) + stage) * 16, p16, 16 * sizeof(double), hipMemcpyHostToDevice));
    if (stage == 0) s->pplant_stale = true;
    return BROV_OK;
}
// ---- boundary corners of the reference API: non-uniform grids and a separate stage-0 weight ------------------------------------
extern "C" int brov_set_time_steps(brov_solver* s, const double* ts) {
    if (s && s->prep_path == 2) s->prep_path = 3;   // a parked preparation (rti_phase 1, resident kernel) does not survive this call
    if (!s) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    if (!ts) s->ts_host.clear();
    else {
        for (int i = 0; i < s->N; i++)
            if (!(ts[i] > 0.0) || !(ts[i] < 1e6)) { g_err = "brov_set_time_steps: every time step must be positive and finite"; return BROV_ERR_ARG; }
        bool uniform = true;
        for (int i = 0; i < s->N; i++) uniform = uniform && std::fabs(ts[i] - ts[0]) <= 1e-12 * std::fabs(ts[0]);
        if (uniform) { s->ts_host.clear(); s->opts.Ts = ts[0]; }     // a uniform vector is the uniform grid: every kernel path stays open
        else s->ts_host.assign(ts, ts + s->N);
    }
    return upload_cst(s);
}
extern "C" int brov_set_stage0_weight(brov_solver* s, const double* W0) {
    if (s && s->prep_path == 2) s->prep_path = 3;   // a parked preparation (rti_phase 1, resident kernel) does not survive this call
    if (!s) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    s->has_W0 = false;
    if (W0) {
        bool same = true;
        for (int j = 0; j < 16; j++) {
            if (!(W0[j] >= 0.0) || !(W0[j] < 1e300) || (j >= 12 && !(W0[j] > 0.0))) { g_err = "brov_set_stage0_weight: weights must be finite, >= 0 (inputs > 0)"; return BROV_ERR_ARG; }
            same = same && W0[j] == s->opts.W[j];
        }
        if (!same) { std::memcpy(s->W0_host, W0, 16 * sizeof(double)); s->has_W0 = true; }
    }
    return upload_cst(s);
}
extern "C" int brov_general_grid(const brov_solver* s) { return s ? (general_grid(s) ? 1 : 0) : BROV_ERR_ARG; }

// ---- 6-disturbance model variant ----------------------------------------------------------------
__global__ void bcast_rp_kernel(const double* __restrict__ d2, double* __restrict__ rp, int B, int N1) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (size_t)B * N1 * 2) return;
    rp[t] = d2[(t / ((size_t)N1 * 2)) * 2 + (t & 1)];
}
__global__ void split_p18_kernel(const double* __restrict__ p18, double* __restrict__ par, double* __restrict__ rp, size_t rows, int N1, int per_stage) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (instance, stage)
    if (t >= rows) return;
    const double* src = p18 + (per_stage ? t : t / N1) * 18;
    double* p = par + t * 16;
    p[0] = src[0]; p[1] = src[1]; p[2] = src[2]; p[3] = src[5];
#pragma unroll
    for (int j = 0; j < 12; j++) p[4 + j] = src[6 + j];
    rp[t * 2] = src[3]; rp[t * 2 + 1] = src[4];
}
extern "C" int brov_enable_dist6(brov_solver* s, int on) {
    if (!s) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    if (on && !s->par_rp) {
        const size_t n = (size_t)s->B * (s->N + 1) * 2;
        int rc = dalloc(s, &s->par_rp, n);
        if (rc == BROV_OK) rc = dalloc(s, &s->prp_plant, (size_t)s->B * 2);
        if (rc != BROV_OK) return rc;
        HIPCHK(hipMemset(s->par_rp, 0, n * sizeof(double)));
        HIPCHK(hipMemset(s->prp_plant, 0, (size_t)s->B * 2 * sizeof(double)));
    }
    s->dist6 = on != 0;
    return BROV_OK;
}
extern "C" int brov_dist6_enabled(const brov_solver* s) { return s ? (s->dist6 ? 1 : 0) : BROV_ERR_ARG; }
static int need_dist6(brov_solver* s, const char* who) {
    if (!s) return BROV_ERR_ARG;
    if (!s->dist6) { g_err = std::string(who) + ": the 6-disturbance model variant is off (brov_enable_dist6)"; return BROV_ERR_ARG; }
    return BROV_OK;
}
static int set_rp(brov_solver* s, const double* d, int per_stage, bool host, void* st) {
    if (int rc = need_dist6(s, "brov_set_rp_disturbance")) return rc;
    if (!d) return BROV_ERR_ARG;
    const size_t N1 = s->N + 1;
    if (per_stage) return copy_in(s, s->par_rp, d, (size_t)s->B * N1 * 2, host, st);
    HIPCHK(hipSetDevice(s->device));
    const double* src = d;
    double* tmp = nullptr;
    if (!host) { if (int rc = order_behind_last(s, (hipStream_t)st)) return rc; }
    if (host) {
        HIPCHK(sync_last(s));
        HIPCHK(hipMalloc((void**)&tmp, (size_t)s->B * 2 * sizeof(double)));
        hipError_t e = hipMemcpy(tmp, d, (size_t)s->B * 2 * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) { hipFree(tmp); g_err = hipGetErrorString(e); return BROV_ERR_HIP; }
        src = tmp;
    }
    const size_t tot = (size_t)s->B * N1 * 2;
    hipLaunchKernelGGL(bcast_rp_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)st, src, s->par_rp, s->B, (int)N1);
    if (host) { hipStreamSynchronize((hipStream_t)st); hipFree(tmp); }
    HIPCHK(hipGetLastError());
    return BROV_OK;
}
extern "C" int brov_set_rp_disturbance_host(brov_solver* s, const double* d, int per_stage) { return set_rp(s, d, per_stage, true, nullptr); }
extern "C" int brov_set_rp_disturbance_device(brov_solver* s, const double* d, int per_stage, void* st) { return set_rp(s, d, per_stage, false, st); }
extern "C" double* brov_rp_disturbance_device(brov_solver* s) { return (s && s->dist6) ? s->par_rp : nullptr; }
extern "C" int brov_get_rp_disturbance_host(brov_solver* s, double* d) {
    if (int rc = need_dist6(s, "brov_get_rp_disturbance_host")) return rc;
    if (!d) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(d, s->par_rp, (size_t)s->B * (s->N + 1) * 2 * sizeof(double), hipMemcpyDeviceToHost));
    return BROV_OK;
}
extern "C" int brov_set_params18_host(brov_solver* s, const double* p18, int per_stage) {
    if (int rc = need_dist6(s, "brov_set_params18_host")) return rc;
    if (!p18) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    const size_t N1 = s->N + 1, rows = (size_t)s->B * N1, nsrc = (per_stage ? rows : (size_t)s->B) * 18;
    double* tmp = nullptr;
    HIPCHK(hipMalloc((void**)&tmp, nsrc * sizeof(double)));
    hipError_t e = hipMemcpy(tmp, p18, nsrc * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(split_p18_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, nullptr, tmp, s->par, s->par_rp, rows, (int)N1, per_stage);
        e = hipDeviceSynchronize();
    }
    hipFree(tmp);
    if (e != hipSuccess) { g_err = hipGetErrorString(e); return BROV_ERR_HIP; }
    s->pplant_stale = true;
    return BROV_OK;
}
extern "C" int brov_plant_set_rp_disturbance_host(brov_solver* s, const double* d) {
    if (int rc = need_dist6(s, "brov_plant_set_rp_disturbance_host")) return rc;
    if (!d) { s->prp_plant_set = false; return BROV_OK; }
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    HIPCHK(hipMemcpy(s->prp_plant, d, (size_t)s->B * 2 * sizeof(double), hipMemcpyHostToDevice));
    s->prp_plant_set = true;
    return BROV_OK;
}

extern "C" int brov_set_yref_stage_host(brov_solver* s, int inst, int stage, const double* y, int ny) {
    if (!s || !y || inst < 0 || inst >= s->B || stage < 0 || stage > s->N || ny < 1 || ny > 16) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    if (s->yref_shared) {  // materialise the shared window per instance first
        for (int b = 0; b < s->B; b++)
            HIPCHK(hipMemcpy(s->yref + (size_t)b * (s->N + 1) * 16, shared_window(s), (size_t)(s->N + 1) * 16 * sizeof(double), hipMemcpyDeviceToDevice));
        s->yref_shared = false;
        s->yref_view = nullptr; s->traj_line = -1;
    }
    HIPCHK(hipMemcpy(s->yref + ((size_t)inst * (s->N + 1) + stage) * 16, y, (size_t)ny * sizeof(double), hipMemcpyHostToDevice));
    return BROV_OK;
}

extern "C" int brov_traj_set_host(brov_solver* s, const double* traj, int rows) {
    if (!s || !traj || rows < 1) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    if (s->yref_view)     // the window in force is a view into the table that is about to go: keep a copy
        HIPCHK(hipMemcpy(s->yref_sh, s->yref_view, (size_t)(s->N + 1) * 16 * sizeof(double), hipMemcpyDeviceToDevice));
    // whatever window is in force -- a view, or one launch_window built from the OLD table (12 columns, end padding) -- no longer names a
    // line of the table that is coming: brov_solve_ticks(row_stride > 0) must not walk the new table from the old one's line
    s->yref_view = nullptr; s->traj_line = -1;
    if (s->traj) { hipFree(s->traj); s->traj = nullptr; }
    HIPCHK(hipMalloc((void**)&s->traj, (size_t)rows * 16 * sizeof(double)));
    HIPCHK(hipMemcpy(s->traj, traj, (size_t)rows * 16 * sizeof(double), hipMemcpyHostToDevice));
    s->traj_rows = rows;
    return BROV_OK;
}
extern "C" int brov_traj_rows(const brov_solver* s) { return s ? s->traj_rows : 0; }
extern "C" int brov_set_yref_from_traj(brov_solver* s, int line, int ncols, void* stream) {
    if (!s || !s->traj || (ncols != 12 && ncols != 16)) { g_err = "brov_set_yref_from_traj: no trajectory or bad ncols"; return BROV_ERR_ARG; }
    HIPCHK(hipSetDevice(s->device));
    if (int rc = order_behind_last(s, (hipStream_t)stream)) return rc;
    if (ncols == 16 && line >= 0 && line + s->N <= s->traj_rows - 1) {
        // the window is N+1 consecutive whole rows of the resident table: use them where they lie (no kernel, no copy)
        s->yref_view = s->traj + (size_t)line * 16;
    } else {
        s->yref_view = nullptr;
        launch_window(s->traj, s->traj_rows, nullptr, line, 1, s->N, ncols, s->yref_sh, (hipStream_t)stream);
    }
    s->traj_line = line; s->traj_ncols = ncols;
    s->yref_shared = true;
    HIPCHK(hipGetLastError());
    return BROV_OK;
}
extern "C" int brov_set_yref_from_traj_lines_host(brov_solver* s, const int32_t* lines, int ncols) {
    if (!s || !s->traj || !lines || (ncols != 12 && ncols != 16)) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    HIPCHK(hipMemcpy(s->lines, lines, (size_t)s->B * sizeof(int), hipMemcpyHostToDevice));
    launch_window(s->traj, s->traj_rows, s->lines, 0, s->B, s->N, ncols, s->yref, nullptr);
    s->yref_shared = false;
    s->yref_view = nullptr; s->traj_line = -1;
    HIPCHK(hipGetLastError());
    return BROV_OK;
}
extern "C" int brov_set_candidate_params_host(brov_solver* s, int kind, const double* p0, const double* p1, const double* phase) {
    if (!s || !p0 || !p1 || !phase || kind < 0 || kind > 1) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    const size_t nb = (size_t)s->B * sizeof(double);
    HIPCHK(hipMemcpy(s->scratch3, p0, nb, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->scratch3 + s->B, p1, nb, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(s->scratch3 + 2 * (size_t)s->B, phase, nb, hipMemcpyHostToDevice));
    s->cand_set = true;
    s->cand_kind = kind;
    return BROV_OK;
}
extern "C" int brov_set_yref_candidates(brov_solver* s, double t0, double dt, void* stream) {
    if (!s || !s->cand_set) { g_err = "brov_set_yref_candidates: no candidate parameters (brov_set_candidate_params_host)"; return BROV_ERR_ARG; }
    HIPCHK(hipSetDevice(s->device));
    if (int rc = order_behind_last(s, (hipStream_t)stream)) return rc;
    launch_candidates(s->cand_kind, s->scratch3, s->scratch3 + s->B, s->scratch3 + 2 * (size_t)s->B, t0, dt, s->B, s->N, s->yref,
                      (hipStream_t)stream);
    s->yref_shared = false;
    s->yref_view = nullptr; s->traj_line = -1;
    HIPCHK(hipGetLastError());
    return BROV_OK;
}
extern "C" int brov_set_yref_candidates_host(brov_solver* s, int kind, const double* p0, const double* p1, const double* phase,
                                             double t0, double dt) {
    const int rc = brov_set_candidate_params_host(s, kind, p0, p1, phase);
    return rc != BROV_OK ? rc : brov_set_yref_candidates(s, t0, dt, nullptr);
}
extern "C" int brov_get_yref_host(brov_solver* s, double* yref) {
    if (!s || !yref) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipDeviceSynchronize());
    const size_t per = (size_t)(s->N + 1) * 16;
    if (s->yref_shared) {
        for (int b = 0; b < s->B; b++) HIPCHK(hipMemcpy(yref + (size_t)b * per, shared_window(s), per * sizeof(double), hipMemcpyDeviceToHost));
    } else {
        HIPCHK(hipMemcpy(yref, s->yref, per * s->B * sizeof(double), hipMemcpyDeviceToHost));
    }
    return BROV_OK;
}
// model parameters currently in force, [B][N+1][16]
extern "C" int brov_get_params_host(brov_solver* s, double* par) {
    if (!s || !par) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(par, s->par, (size_t)s->B * (s->N + 1) * 16 * sizeof(double), hipMemcpyDeviceToHost));
    return BROV_OK;
}

__global__ void copy_stage0_par_kernel(const double* __restrict__ par, double* __restrict__ pp, int B, int N1) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < B * 16) pp[t] = par[(size_t)(t >> 4) * N1 * 16 + (t & 15)];
}
__global__ void gather_status_kernel(const brov_result* __restrict__ res, int* __restrict__ out, int B) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < B) out[t] = res[t].status;
}
extern "C" int brov_plant_set_params_host(brov_solver* s, const double* p) {
    if (!s || !p) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipMemcpy(s->pplant, p, (size_t)s->B * 16 * sizeof(double), hipMemcpyHostToDevice));
    s->pplant_set = true;
    return BROV_OK;
}
// Without explicit plant parameters the plant is the controller's own model: stage-0 parameters, re-read whenever the
// controller's parameters may have changed (setters, brov_params_device() hand-outs, the EKF's write-back).
// roll / pitch disturbance moments the plant integrates (6-disturbance variant): its own, else the controller's stage 0, else none
static const double* plant_rp(const brov_solver* s) { return s->prp_plant_set ? s->prp_plant : (s->dist6 && !s->pplant_set ? s->par_rp : nullptr); }
static int plant_rp_stride(const brov_solver* s) { return s->prp_plant_set ? 2 : (s->N + 1) * 2; }
static int ensure_plant_params(brov_solver* s, hipStream_t st) {
    if (!s->pplant_set && s->pplant_stale) {
        hipLaunchKernelGGL(copy_stage0_par_kernel, dim3((s->B * 16 + 255) / 256), dim3(256), 0, st, s->par, s->pplant, s->B, s->N + 1);
        s->pplant_stale = false;
    }
    return BROV_OK;
}
extern "C" int brov_plant_step(brov_solver* s, double dt, int substeps, void* stream) {
    if (!s || !(dt > 0.0) || substeps < 1) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    if (int rc = order_behind_last(s, (hipStream_t)stream)) return rc;
    ensure_plant_params(s, (hipStream_t)stream);
    launch_plant(s->x0, s->res, s->pplant, plant_rp(s), plant_rp_stride(s), s->B, dt, substeps, nullptr, nullptr, (hipStream_t)stream);
    HIPCHK(hipGetLastError());
    return BROV_OK;
}
extern "C" int brov_get_x0_host(brov_solver* s, double* x0) {
    if (!s || !x0) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(x0, s->x0, (size_t)s->B * 12 * sizeof(double), hipMemcpyDeviceToHost));
    return BROV_OK;
}
static DevParams make_params(const brov_solver* s);
static int ticks_kernel(const brov_solver* s);
static void launch_ticks(brov_solver* s, DevParams& P, hipStream_t st, int which);
extern "C" int brov_closed_loop(brov_solver* s, int ticks, int line0, int ncols, double dt, int substeps, double* u_log, double* x_log,
                                int32_t* st_log) {
    if (!s || ticks < 1 || !s->traj || (ncols != 12 && ncols != 16) || !(dt > 0.0) || substeps < 1) {
        g_err = "brov_closed_loop: bad argument (needs a trajectory table, see brov_traj_set_host)";
        return BROV_ERR_ARG;
    }
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = s->last_stream;
    const size_t B = s->B;
    double *dx = nullptr, *du = nullptr;
    int* dst = nullptr;
    int rc = BROV_OK;
    auto alloc = [&](void** p, size_t bytes) {
        if (rc != BROV_OK) return;
        const hipError_t e = hipMalloc(p, bytes);
        if (e != hipSuccess) { g_err = std::string("brov_closed_loop: hipMalloc: ") + hipGetErrorString(e); rc = BROV_ERR_ALLOC; }
    };
    if (x_log) alloc((void**)&dx, (size_t)(ticks + 1) * B * 12 * sizeof(double));
    if (u_log) alloc((void**)&du, (size_t)ticks * B * 4 * sizeof(double));
    if (st_log) alloc((void**)&dst, (size_t)ticks * B * sizeof(int));
    if (rc == BROV_OK) ensure_plant_params(s, st);
    if (rc == BROV_OK && dx && hipMemcpyAsync(dx, s->x0, B * 12 * sizeof(double), hipMemcpyDeviceToDevice, st) != hipSuccess) {
        g_err = "brov_closed_loop: log copy failed";
        rc = BROV_ERR_HIP;
    }
    // One launch for the whole loop where the fused kernels serve the solver and every window is rows of the table in place (round 5,
    // rti_fused_kernel_ticks with the plant update behind every step): every instance runs its own closed loop at its own pace -- no launch
    // boundaries, three launches per tick saved, and a tick on which one instance grinds through the QP loop holds nobody else.
    const int which = ticks_kernel(s);
    const bool one_launch = rc == BROV_OK && s->k.closed_loop_fused && ncols == 16 && line0 >= 0 && line0 + (ticks - 1) + s->N <= s->traj_rows - 1 && which != 0;
    if (one_launch) {
        rc = brov_set_yref_from_traj(s, line0, 16, st);
        if (rc == BROV_OK) rc = order_behind_last(s, st);
        if (rc == BROV_OK) {
            DevParams P = make_params(s);
            P.sched = nullptr;
            P.ticks = ticks; P.tick_yref = 16; P.tick_status = dst;
            P.plant_pp = s->pplant; P.plant_rp = plant_rp(s); P.plant_rp_stride = plant_rp_stride(s); P.plant_substeps = substeps; P.plant_dt = dt;
            P.x0_rw = s->x0; P.plant_xlog = dx ? dx + B * 12 : nullptr; P.plant_ulog = du;
            if (s->timing) { hipEventRecord(s->ev[0], st); hipEventRecord(s->ev[1], st); }   // (as brov_solve_ticks: brov_last_solve_seconds then reports THIS launch)
            launch_ticks(s, P, st, which);
            if (s->timing) { hipEventRecord(s->ev[2], st); s->ev_valid = true; }
            s->traj_line = line0 + ticks - 1; s->yref_view = s->traj + (size_t)s->traj_line * 16;
        }
    }
    for (int k = 0; k < ticks && rc == BROV_OK && !one_launch; k++) {
        s->yref_view = nullptr; s->traj_line = -1;
        launch_window(s->traj, s->traj_rows, nullptr, line0 + k, 1, s->N, ncols, s->yref_sh, st);
        s->yref_shared = true;
        rc = brov_solve_phase(s, st, 0);
        if (dst) hipLaunchKernelGGL(gather_status_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, s->res, dst + (size_t)k * B, (int)B);
        launch_plant(s->x0, s->res, s->pplant, plant_rp(s), plant_rp_stride(s), s->B, dt, substeps, dx ? dx + (size_t)(k + 1) * B * 12 : nullptr,
                     du ? du + (size_t)k * B * 4 : nullptr, st);
    }
    hipError_t e = hipStreamSynchronize(st);
    if (rc == BROV_OK && e != hipSuccess) { g_err = hipGetErrorString(e); rc = BROV_ERR_HIP; }
    if (rc == BROV_OK && dx) hipMemcpy(x_log, dx, (size_t)(ticks + 1) * B * 12 * sizeof(double), hipMemcpyDeviceToHost);
    if (rc == BROV_OK && du) hipMemcpy(u_log, du, (size_t)ticks * B * 4 * sizeof(double), hipMemcpyDeviceToHost);
    if (rc == BROV_OK && dst) hipMemcpy(st_log, dst, (size_t)ticks * B * sizeof(int), hipMemcpyDeviceToHost);
    if (dx) hipFree(dx);
    if (du) hipFree(du);
    if (dst) hipFree(dst);
    return rc;
}

extern "C" int brov_set_iterate_host(brov_solver* s, const double* x, const double* u, const double* pi, const double* lam) {
    if (s && s->prep_path == 2) s->prep_path = 3;   // a parked preparation (rti_phase 1, resident kernel) does not survive this call
    if (!s) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));   // a solve on a non-blocking stream may still be writing the iterate
    const size_t B = s->B, N = s->N;
    if (x) HIPCHK(hipMemcpy(s->x, x, B * (N + 1) * 12 * sizeof(double), hipMemcpyHostToDevice));
    if (u) HIPCHK(hipMemcpy(s->u, u, B * N * 4 * sizeof(double), hipMemcpyHostToDevice));
    if (pi) HIPCHK(hipMemcpy(s->pi, pi, B * N * 12 * sizeof(double), hipMemcpyHostToDevice));
    if (lam) HIPCHK(hipMemcpy(s->lam, lam, B * N * 8 * sizeof(double), hipMemcpyHostToDevice));
    return BROV_OK;
}
extern "C" int brov_get_iterate_host(brov_solver* s, double* x, double* u, double* pi, double* lam) {
    if (!s) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    const size_t B = s->B, N = s->N;
    if (x) HIPCHK(hipMemcpy(x, s->x, B * (N + 1) * 12 * sizeof(double), hipMemcpyDeviceToHost));
    if (u) HIPCHK(hipMemcpy(u, s->u, B * N * 4 * sizeof(double), hipMemcpyDeviceToHost));
    if (pi) HIPCHK(hipMemcpy(pi, s->pi, B * N * 12 * sizeof(double), hipMemcpyDeviceToHost));
    if (lam) HIPCHK(hipMemcpy(lam, s->lam, B * N * 8 * sizeof(double), hipMemcpyDeviceToHost));
    return BROV_OK;
}
extern "C" int brov_reset(brov_solver* s) {  // acados_solver_bluerov2.c:797-830: everything to zero
    if (s && s->prep_path == 2) s->prep_path = 3;   // a parked preparation (rti_phase 1, resident kernel) does not survive this call
    if (!s) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    const size_t B = s->B, N = s->N;
    HIPCHK(hipMemset(s->x, 0, B * (N + 1) * 12 * sizeof(double)));
    HIPCHK(hipMemset(s->u, 0, B * N * 4 * sizeof(double)));
    HIPCHK(hipMemset(s->pi, 0, B * N * 12 * sizeof(double)));
    HIPCHK(hipMemset(s->lam, 0, B * N * 8 * sizeof(double)));
    // the records too: their u0 / thrust is what a failed step holds, and a reset must not hand the previous run's input on
    HIPCHK(hipMemset(s->res, 0, B * sizeof(brov_result)));
    return BROV_OK;
}

static DevParams make_params(const brov_solver* s) {
    DevParams P;
    std::memset(&P, 0, sizeof P);
    P.B = s->B; P.N = s->N;
    P.qp_iter_max = s->opts.qp_iter_max; P.early_exit = s->opts.qp_early_exit;
    P.on_failure = s->opts.on_failure; P.dump_lin = s->dump_lin ? 1 : 0;
    P.robust_kkt_max = s->k.robust_kkt_max;
    P.robust_pivot = s->k.robust_pivot;            // development knob: 0 off, 1 on demand (default), 2 every instance
    P.partial_refactor = s->k.partial_refactor;    // development knob (A/B, tests)
    P.Ts = s->opts.Ts; P.tol_mu = s->opts.qp_tol_mu; P.tol_stat = s->opts.qp_tol_stat;
    for (int j = 0; j < 16; j++) P.W[j] = s->opts.W[j];
    for (int j = 0; j < 12; j++) P.We[j] = s->opts.We[j];
    for (int j = 0; j < 4; j++) { P.lbu[j] = s->opts.lbu[j]; P.ubu[j] = s->opts.ubu[j]; }
    P.x0 = s->tick_x0 ? s->tick_x0 : s->x0;
    P.yref = s->tick_yref ? s->tick_yref : (s->yref_shared ? shared_window(s) : s->yref);
    P.yref_stride = s->yref_shared ? 0 : (int64_t)(s->N + 1) * 16;
    P.par = s->tick_par ? s->tick_par : s->par;
    P.par_rp = s->dist6 ? s->par_rp : nullptr;
    P.tsv = general_grid(s) ? s->tsv : nullptr;
    P.sched = s->sched_on ? s->sched : nullptr;
    P.sched_stride = sched_buffer_ints_host(s->B);
    P.sched_r = (int)(s->sched_tick % 3); P.sched_w = (int)((s->sched_tick + 1) % 3); P.sched_z = (int)((s->sched_tick + 2) % 3);
    P.wst = general_grid(s) ? s->wst : nullptr;
    P.x = s->x; P.u = s->u; P.pi = s->pi; P.lam = s->lam;
    P.BA = s->BA; P.bvec = s->bvec; P.kktp = s->kktp;
    P.Ks = s->Ks; P.Kt = s->Kt; P.Mt = s->Mt; P.Pb = s->Pb; P.kff = s->kff; P.vhat = s->vhat; P.ipm = s->ipm;
    P.dxb = s->dxb; P.cst = s->cst; P.res = s->res;
    P.mail = s->mail; P.mail_flag = s->mail_flag; P.mail_seq = s->mail_seq;
    P.mail_early = s->k.mail_early;   // development knob (A/B)
    P.ws = s->ws; P.ws_stride = s->win_L ? (int64_t)windowed_ws_doubles(s->N, s->win_L) : 0; P.counter = s->counter + 32 * (s->win_tick & 1u);
    P.counter_next = s->counter + 32 * ((s->win_tick + 1u) & 1u);
    P.win_L = s->win_L; P.win_blocks = s->win_blocks;
    P.dbg = s->dbg;
    return P;
}

extern "C" int brov_solve_phase(brov_solver* s, void* stream, int rti_phase) {
    if (!s || rti_phase < 0 || rti_phase > 2) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    if (int rc = order_behind_last(s, st)) return rc;   // e.g. a brov_tick_host whose kernel is still finishing on the solver's own stream
    DevParams P = make_params(s);
    const int path = s->opts.kernel_path;
    // LDS-resident kernels (one launch): whole horizon for N <= 23, windowed above.  rti_phase 1 / 2 (preparation and feedback as
    // separate calls) need the linearisation in HBM between the calls: streaming kernels.
    // a general grid (per-stage time steps / separate stage-0 weight) runs on the LDS-resident kernels too (round 4: rti_fused_kernel_grid,
    // rti_window_kernel_grid, rti_window_kernel_res_grid, rti_pit_kernel_grid)
    // rti_phase 1 / 2 in the windowed kernel's resident mode (at most one instance per CU at 24 <= N <= 81): the split
    // launches of rti_window_kernel_res_split -- preparation parks the factorised LDS image per instance, feedback runs from the forward
    // sweep on.  A feedback call follows the path its preparation took (prep_path); BROV_SPLIT_RESIDENT=0: the streaming pair as before.
    const bool fused_h = fused_supported(s->N) && !s->force_windowed;
    const bool split_env = s->k.split_resident != 0;
    const bool split_fused_h = fused_h && s->ws_split != nullptr && path != BROV_PATH_STREAMING && !s->dump_lin && split_env;   // (N <= 23: see brov_create)
    const bool split_res_ok = split_fused_h || (path != BROV_PATH_STREAMING && !fused_h && s->ws != nullptr &&
                              windowed_is_resident(s->win_L) && s->win_blocks == (int)s->B && !s->dump_lin && split_env);
    const bool split_res = (rti_phase == 1 && split_res_ok) || (rti_phase == 2 && split_res_ok && s->prep_path == 2);
    if (rti_phase == 2 && s->prep_path == 0) {   // no preparation, or one that a later step (rti_phase 0, an earlier feedback) has used up: the iterate it linearised is gone
        g_err = "brov_solve: rti_phase 2 needs a preparation (rti_phase 1) of the CURRENT iterate: none since the last step";
        return BROV_ERR_ARG;
    }
    if (rti_phase == 2 && (s->prep_path == 3 || (s->prep_path == 2 && !split_res_ok))) {   // (a grid / option / iterate / path change between the two calls)
        g_err = "brov_solve: rti_phase 2 after a preparation on the resident kernel, which the solver's settings no longer allow: repeat rti_phase 1";
        return BROV_ERR_ARG;
    }
    if (rti_phase == 1) s->prep_path = split_res ? 2 : 1;
    // (128 < N <= 256: rti_window_kernel_long / _long_grid)
    const bool lds_path = (rti_phase == 0 || split_res) && path != BROV_PATH_STREAMING;
    const bool fused = lds_path && fused_supported(s->N) && !s->force_windowed && !(split_res && split_fused_h);
    const bool windowed = lds_path && !fused && (s->ws != nullptr || (split_res && split_fused_h));
    if (split_res && split_fused_h) {   // the resident configuration of a fused-kernel horizon: one window = the horizon, one block per instance
        P.ws = s->ws_split; P.ws_stride = (int64_t)windowed_ws_doubles(s->N, s->N); P.win_L = s->N; P.win_blocks = (int32_t)s->B;
    }
    s->pit_ran = false;
    if (s->timing) hipEventRecord(s->ev[0], st);
    if (fused || windowed) {
        if (s->timing) hipEventRecord(s->ev[1], st);
        if (fused) launch_fused(P, st, s->k);
        else {
            // batches the resident mode serves: the parallel-in-time step-0 solve goes first (rti_pit_kernel; BROV_PIT=0 off, 2: every
            // instance is tried, not only those whose previous step was an early exit)
            const int pit = s->k.pit;
            const bool pit_can = pit && s->pit_done && !s->dump_lin && rti_phase == 0;
            P.rti_split = rti_phase;
            // the feedback half of a split tick: the parallel-in-time kernel's feedback instantiation first, rolling out the
            // four quarters at once from what the preparation parked; the resident feedback launch behind it for what it leaves
            if (rti_phase == 2 && pit && s->pit_done && pit_supported(s->N, P.win_L) && s->k.split_parallel) {
                P.pit = pit; P.pit_done = s->pit_done; P.pit_blocks = P.win_blocks;
                P.pit_try = s->k.pit_try; P.pit_light = s->k.pit_light;
            }
            P.pit_blocks = P.win_blocks;
            // Round 5: a constant rule.  The parallel kernel runs the whole QP loop itself (qp/pit.hpp), so what it leaves to the resident kernel
            // behind it are the instances it gives up on -- a NaN, a pivot block that fails or is ill-conditioned: verdicts of its first pass --, and
            // those cost it next to nothing.  Round 4's kernel left every instance that needed a fourth try or an interior-point iteration, which then
            // STARTED only when it was over; the host followed a pinned report word, paused the kernel for eight solves and probed (forty lines here).
            // All of that is gone: the kernel runs whenever it can serve the solve.
            const bool pit_now = pit_can && (s->alt_L != 0 || pit_supported(s->N, P.win_L));
            if (pit_now && s->alt_L) {   // between one and two instances per CU: the resident configuration, one rti_pit_kernel block per instance
                P.win_L = s->alt_L; P.win_blocks = s->alt_blocks; P.ws_stride = (int64_t)windowed_ws_doubles(s->N, s->alt_L);
                P.pit_blocks = (int32_t)s->B;
            }
            if (pit_now && pit_supported(s->N, P.win_L)) {
                P.pit = pit; P.pit_done = s->pit_done;
                P.pit_try = s->k.pit_try; P.pit_light = s->k.pit_light;
            }
            s->pit_ran = P.pit != 0;
            launch_windowed(P, st); s->win_tick++;   // persistent blocks; the two hand-out counters alternate
        }
    } else {
        if (rti_phase != 2) launch_linearise(P, st);
        if (s->timing) hipEventRecord(s->ev[1], st);
        if (rti_phase != 1) launch_qp(P, st);
    }
    if (s->timing) { hipEventRecord(s->ev[2], st); s->ev_valid = true; }
    if (rti_phase != 1) s->sched_tick++;   // a QP kernel ran: it wrote the next ordering
    if (rti_phase != 1) s->prep_path = 0;  // ... and the iterate moved (and the per-block workspace was rewritten): whatever preparation there was is used up
    s->last_fused = fused;
    s->last_windowed = windowed;
    s->last_stream = st;
    HIPCHK(hipGetLastError());
    return BROV_OK;
}
extern "C" int brov_solve(brov_solver* s, void* stream) { return brov_solve_phase(s, stream, 0); }

// steps in one launch: 1 = rti_fused_kernel_ticks (N <= 23), 2 = rti_window_kernel_ticks (longer horizons, large batches: the windowed kernel's
// persistent blocks), 0 = neither (general grid, streaming pair, a dumped linearisation, the resident / parallel-in-time configurations of small
// batches -- those are latency paths: a launch per step)
static int ticks_kernel(const brov_solver* s) {
    if (s->opts.kernel_path == BROV_PATH_STREAMING || general_grid(s) || s->dump_lin) return 0;
    if (fused_supported(s->N) && !s->force_windowed) return 1;
    if (s->ws != nullptr && !windowed_is_resident(s->win_L) && s->alt_L == 0) return 2;
    return 0;
}
static void launch_ticks(brov_solver* s, DevParams& P, hipStream_t st, int which) {
    if (which == 1) { launch_fused_ticks(P, st, s->k); s->last_fused = true; s->last_windowed = false; }
    else { launch_windowed(P, st); s->win_tick++; s->last_fused = false; s->last_windowed = true; }
    s->pit_ran = false; s->prep_path = 0; s->last_stream = st;
}

// `ticks` RTI steps of every instance with ONE launch where the fused kernels serve the solver (N <= 23, uniform grid): rti_fused_kernel_ticks,
// every instance going on to its next step as soon as its own is done.  Elsewhere (and when the moving window would leave the resident
// table): the same steps as `ticks` launches.  Either way the result equals `ticks` x { brov_set_yref_from_traj(line + k row_stride); brov_solve }.
extern "C" int brov_solve_ticks(brov_solver* s, void* stream, int ticks, int row_stride, int32_t* status_log) {
    if (!s || ticks < 1 || row_stride < 0) { g_err = "brov_solve_ticks: bad argument"; return BROV_ERR_ARG; }
    if (row_stride > 0 && !(s->yref_shared && s->traj && s->traj_line >= 0)) {
        g_err = "brov_solve_ticks: a moving window (row_stride > 0) needs the window in force to come from brov_set_yref_from_traj";
        return BROV_ERR_ARG;
    }
    HIPCHK(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    const int line0 = s->traj_line, ncols = s->traj_ncols;
    const int which = ticks_kernel(s);
    const bool one_launch = which != 0 && (row_stride == 0 || (s->yref_view != nullptr && line0 + (ticks - 1) * row_stride + s->N <= s->traj_rows - 1));
    if (!one_launch) {
        for (int k = 0; k < ticks; k++) {
            if (k > 0 && row_stride > 0)
                if (int rc = brov_set_yref_from_traj(s, line0 + k * row_stride, ncols, stream)) return rc;
            if (int rc = brov_solve_phase(s, stream, 0)) return rc;
            if (status_log) hipLaunchKernelGGL(gather_status_kernel, dim3((unsigned)((s->B + 255) / 256)), dim3(256), 0, st, s->res, status_log + (size_t)k * s->B, (int)s->B);
        }
        HIPCHK(hipGetLastError());
        return BROV_OK;
    }
    if (int rc = order_behind_last(s, st)) return rc;
    DevParams P = make_params(s);
    P.sched = nullptr;                 // a launch of many steps neither reads nor writes the work ordering: every instance follows its own history
    P.ticks = ticks; P.tick_yref = (int64_t)row_stride * 16; P.tick_status = status_log;
    if (s->timing) { hipEventRecord(s->ev[0], st); hipEventRecord(s->ev[1], st); }
    launch_ticks(s, P, st, which);
    if (s->timing) { hipEventRecord(s->ev[2], st); s->ev_valid = true; }
    if (row_stride > 0) {              // the window in force is the last step's
        s->traj_line = line0 + (ticks - 1) * row_stride;
        s->yref_view = s->traj + (size_t)s->traj_line * 16;
    }
    HIPCHK(hipGetLastError());
    return BROV_OK;
}

// One control tick with the fewest host round trips (what the acados-shaped drop-in calls per bluerov2_acados_solve): the inputs that
// changed go through ONE pinned staging buffer and asynchronous copies on the solver's own stream, the step is enqueued behind them,
// the result records come back the same way, and the host waits once.  The separate setters + brov_solve + brov_get_results_host
// cost five blocking pageable copies and three synchronisations per tick -- 0.2 .. 0.4 ms at batch 1, more than the kernels.
// the pinned staging buffer of brov_tick_host (x0 | shared window | stage parameters | records | sequence words), (re)allocated on demand
static int tick_pin(brov_solver* s) {
    const size_t B = s->B, N1 = s->N + 1;
    const size_t n_x0 = B * 12, n_y = N1 * 16, n_p = B * N1 * 16, n_r = (B * sizeof(brov_result) + 7) / 8, n_f = (B * sizeof(int32_t) + 7) / 8;
    // (x0 | window | parameters) | records | sequence words | a second (x0 | window | parameters): ticks that copy their arguments in alternate
    // between the two input sets
    const size_t n_all = 2 * (n_x0 + n_y + n_p) + n_r + n_f;
    if (s->pin_doubles < n_all) {
        if (s->pin) hipHostFree(s->pin);
        s->pin = nullptr; s->pin_doubles = 0;
        HIPCHK(hipHostMalloc((void**)&s->pin, n_all * sizeof(double), hipHostMallocDefault));
        s->pin_doubles = n_all;
        std::memset(s->pin, 0, s->pin_doubles * sizeof(double));
    }
    return BROV_OK;
}
// A caller that builds its inputs and reads its records IN these buffers (device-visible pinned host memory, valid until brov_destroy)
// saves brov_tick_host the host-side copies: pass the pointers returned here as x0 / yref_shared / par_stage and NULL as res.
extern "C" int brov_tick_buffers(brov_solver* s, double** x0, double** yref_shared, double** par_stage, const brov_result** res) {
    if (!s) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    if (int rc = tick_pin(s)) return rc;
    const size_t B = s->B, N1 = s->N + 1;
    // input set 0 is the caller's from here on and "may be rewritten freely" (header): refresh copies an EARLIER copying tick left running out
    // of set 0 (copy_stream; they read it) must be over before the pointers go out -- the wait at the top of the next brov_tick_host comes
    // after the caller's writes (round-5 advisor)
    if (s->set_pending[0]) { HIPCHK(hipEventSynchronize(s->ev_set[0])); s->set_pending[0] = false; }
    s->buffers_out = true;
    if (x0) *x0 = s->pin;
    if (yref_shared) *yref_shared = s->pin + B * 12;
    if (par_stage) *par_stage = s->pin + B * 12 + N1 * 16;
    if (res) *res = (const brov_result*)(s->pin + B * 12 + N1 * 16 + B * N1 * 16);
    return BROV_OK;
}

extern "C" int brov_tick_host(brov_solver* s, const double* x0, const double* yref_shared, const double* par_stage, int rti_phase,
                              brov_result* res) {
    if (!s || rti_phase < 0 || rti_phase > 2) return BROV_ERR_ARG;
    using clk = std::chrono::steady_clock;
    const bool brk = s->k.tick_breakdown != 0;
    clk::time_point tb0, tb1, tb2, tb3, tb4;
    if (brk) tb0 = clk::now();
    HIPCHK(hipSetDevice(s->device));
    const size_t B = s->B, N1 = s->N + 1;
    const size_t n_x0 = B * 12, n_y = N1 * 16, n_p = B * N1 * 16, n_r = (B * sizeof(brov_result) + 7) / 8;
    if (!s->tick_stream) HIPCHK(hipStreamCreateWithFlags(&s->tick_stream, hipStreamNonBlocking));
    if (int rc = tick_pin(s)) return rc;
    hipStream_t st = s->tick_stream;
    if (s->last_stream != st) HIPCHK(sync_last(s));   // an earlier solve on the caller's stream
    double* pr = s->pin + n_x0 + n_y + n_p;
    volatile int32_t* pf = (volatile int32_t*)(pr + n_r);
    // The refresh copies a zero-copy tick leaves running behind its kernel (pinned staging buffer -> device arrays, copy_stream) still READ the
    // input set they were enqueued for (round-4 advisor: rewriting it under them leaves a torn or already-next-tick x0 in the device array).
    // Waiting for them at the top of the next tick costs a back-to-back control loop 30 us (the kernel's tail and the copies behind it; measured,
    // round 5) -- so the staging buffer holds TWO input sets and ticks that copy their arguments in alternate: a set is rewritten two ticks
    // after its copies were enqueued, and the check below almost always finds them done.  Set 0 is the one brov_tick_buffers hands out.
    const bool in_place = (x0 && x0 == s->pin) || (yref_shared && yref_shared == s->pin + n_x0) || (par_stage && par_stage == s->pin + n_x0 + n_y);
    // (a caller that holds set 0 through brov_tick_buffers keeps it to itself: its copying ticks, if any, all use set 1)
    const int sel = in_place ? 0 : (s->buffers_out ? 1 : (s->pin_sel ^= 1));
    double* px = sel ? pr + n_r + (B * sizeof(int32_t) + 7) / 8 : s->pin; double* py = px + n_x0; double* pp = py + n_y;
    if (s->set_pending[sel] && hipEventQuery(s->ev_set[sel]) != hipSuccess) {
        (void)hipGetLastError();
        HIPCHK(hipEventSynchronize(s->ev_set[sel]));
    }
    s->set_pending[sel] = false;
    if (x0 && x0 != px) std::memcpy(px, x0, n_x0 * sizeof(double));   // (equal: the caller wrote into the staging buffer, brov_tick_buffers)
    if (yref_shared) {
        if (yref_shared != py) std::memcpy(py, yref_shared, n_y * sizeof(double));
        s->yref_view = nullptr; s->traj_line = -1;
        s->yref_shared = true;
    }
    if (par_stage) { if (par_stage != pp) std::memcpy(pp, par_stage, n_p * sizeof(double)); s->pplant_stale = true; }
    // Small batches with the mailbox: the kernel reads the inputs of this tick where the host has just put them (pinned, device-visible
    // memory: 21 KB over PCIe inside the linearisation's staging loads) instead of waiting for a copy command ahead of it; the device
    // copies every other entry point works on are refreshed by the same copies, enqueued BEHIND the launch (BROV_TICK_ZEROCOPY=0: ahead).
    const bool mailbox = rti_phase != 1 && B <= kTickMailboxMaxBatch && s->k.tick_mailbox;
    // Larger batches whose records the kernel writes into the pinned buffer itself (bulk, below): the same for x0 and a shared window (96 bytes
    // per instance over PCIe inside the linearisation's staging loads) -- not for per-stage parameters passed with the tick (2.7 KB per instance
    // at N = 20: those go through the copy engine ahead of the launch, and the other inputs with them).
    const bool bulk = !mailbox && rti_phase != 1 && B > kTickMailboxMaxBatch && s->k.tick_mailbox && s->k.tick_bulk;
    // (a feedback call, rti_phase 2, too: its kernel reads the new measurement where the host has just put it)
    const bool zerocopy = (mailbox || (bulk && !par_stage)) && rti_phase != 1 && s->k.tick_zerocopy;
    auto upload = [&](hipStream_t cs) -> int {
        if (x0 && yref_shared && par_stage) {   // device side: one allocation in the same order (brov_create)
            HIPCHK(hipMemcpyAsync(s->x0, px, (n_x0 + n_y + n_p) * sizeof(double), hipMemcpyHostToDevice, cs));
        } else {
            if (x0) HIPCHK(hipMemcpyAsync(s->x0, px, n_x0 * sizeof(double), hipMemcpyHostToDevice, cs));
            if (yref_shared) HIPCHK(hipMemcpyAsync(s->yref_sh, py, n_y * sizeof(double), hipMemcpyHostToDevice, cs));
            if (par_stage) HIPCHK(hipMemcpyAsync(s->par, pp, n_p * sizeof(double), hipMemcpyHostToDevice, cs));
        }
        return BROV_OK;
    };
    if (!zerocopy) {
        if (s->copies_pending) HIPCHK(hipStreamWaitEvent(st, s->ev_copy, 0));   // (an earlier tick's copies into the same device arrays)
        if (int rc = upload(st)) return rc;
        if (rti_phase == 1) {   // a preparation delivers nothing: the call returns when the staging buffer is free again (below), not when the kernel ends
            if (!s->ev_up) HIPCHK(hipEventCreateWithFlags(&s->ev_up, hipEventDisableTiming));
            HIPCHK(hipEventRecord(s->ev_up, st));
        }
    } else {
        s->tick_x0 = x0 ? px : nullptr; s->tick_yref = yref_shared ? py : nullptr; s->tick_par = par_stage ? pp : nullptr;
        if (!s->copy_stream) {
            HIPCHK(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&s->ev_tick, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&s->ev_pre, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&s->ev_set[0], hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&s->ev_set[1], hipEventDisableTiming));
            s->ev_copy = s->ev_set[0];
        }
    }
    // Results.  Small batches (the ROS node's batch of one): the kernel writes every record into the pinned buffer itself and then the
    // instance's sequence word; the host polls those words -- no copy command, no stream synchronisation on the way back.  The stream
    // is queried now and then: a launch that ended without delivering (a device fault) falls back to the synchronous path's error.
    // Larger batches: the kernel still writes the records into the pinned buffer itself (no copy command behind it), without the
    // per-instance sequence words -- the host waits for the stream once.
    if (mailbox) {
        s->mail_seq = s->mail_seq == 0x7fffffff ? 1 : s->mail_seq + 1;
        s->mail = (brov_result*)pr; s->mail_flag = (int32_t*)pf;
    } else if (bulk) {
        s->mail = (brov_result*)pr; s->mail_flag = nullptr;
    }
    // (inputs NOT passed with this tick are read from their device arrays: if the copies an earlier tick left running write one of those, the
    // kernel is ordered behind them after all -- a loop that passes the same inputs every tick never is)
    const int passed = (x0 ? 1 : 0) | (yref_shared ? 2 : 0) | (par_stage ? 4 : 0);
    const bool behind_copies = s->copies_pending && (s->copy_mask & ~passed) != 0;
    s->in_tick = zerocopy && !behind_copies;
    if (brk) tb1 = clk::now();
    // Large batches (records written by the kernel, the host waits for the launch once): the refresh copies of the tick's inputs run NEXT TO the
    // kernel -- it reads the pinned copies, they write the device arrays, which nothing of this launch reads -- ordered only behind what was
    // enqueued ahead of it.  They are over long before the kernel is (0.4 MB against 0.16 ms), so a caller that owns the staging buffers
    // (brov_tick_buffers) does not pay for them at the end of the call.  Small batches keep them BEHIND the kernel: next to it they would share
    // the PCIe reads of a 50 us launch whose latency is the point.
    const bool copies_beside = zerocopy && bulk && !behind_copies;
    if (copies_beside) HIPCHK(hipEventRecord(s->ev_pre, st));
    const int rc = brov_solve_phase(s, st, rti_phase);
    if (brk) tb2 = clk::now();
    s->in_tick = false;
    const int32_t seq = s->mail_seq;
    s->mail = nullptr; s->mail_flag = nullptr;
    s->tick_x0 = s->tick_yref = s->tick_par = nullptr;
    if (rc) return rc;
    if (zerocopy) {
        // the device copies every other entry point works on: refreshed BEHIND the kernel that has read the pinned ones, on the copy stream --
        // neither the host (which waits for the kernel's end / the mailbox) nor the next tick's kernel waits for them; whatever else touches
        // those arrays is ordered behind ev_copy (order_behind_last, sync_last)
        HIPCHK(hipEventRecord(s->ev_tick, st));
        HIPCHK(hipStreamWaitEvent(s->copy_stream, copies_beside ? s->ev_pre : s->ev_tick, 0));
        if (int rc2 = upload(s->copy_stream)) return rc2;
        s->ev_copy = s->ev_set[sel];
        HIPCHK(hipEventRecord(s->ev_copy, s->copy_stream));
        s->set_pending[sel] = true;
        s->copy_mask = (s->copies_pending && !behind_copies) ? (s->copy_mask | passed) : passed;   // (arrays written by copies no kernel on st is ordered behind yet)
        s->copies_pending = true;
    }
    if (brk) tb3 = clk::now();
    if (mailbox) {
        size_t done = 0;
        for (unsigned long spin = 1; done < B; spin++) {
            while (done < B && pf[done] == seq) done++;
            if (done < B && (spin & 0x3ff) == 0) {
                const hipError_t q = hipStreamQuery(st);
                if (q == hipSuccess) {   // the launch is over: everything it wrote is visible
                    while (done < B && pf[done] == seq) done++;
                    if (done < B) { g_err = "brov_tick_host: the solve ended without delivering its records"; return BROV_ERR_HIP; }
                } else if (q != hipErrorNotReady) {
                    g_err = std::string("brov_tick_host: ") + hipGetErrorString(q);
                    return BROV_ERR_HIP;
                }
            }
        }
        std::atomic_thread_fence(std::memory_order_acquire);
    } else if (rti_phase == 1) {
        HIPCHK(hipEventSynchronize(s->ev_up));   // the inputs have left the pinned buffer; the preparation itself runs on (stream-ordered ahead of
        return BROV_OK;                          // whatever follows; no record: `res` is left alone)
    } else {
        if (!bulk) HIPCHK(hipMemcpyAsync(pr, s->res, B * sizeof(brov_result), hipMemcpyDeviceToHost, st));
        if (zerocopy) HIPCHK(hipEventSynchronize(s->ev_tick));
        else HIPCHK(hipStreamSynchronize(st));
    }
    if (brk) {
        tb4 = clk::now();
        auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        s->tick_us[0] = us(tb0, tb1); s->tick_us[1] = us(tb1, tb2); s->tick_us[2] = us(tb2, tb3); s->tick_us[3] = us(tb3, tb4); s->tick_us[4] = us(tb0, tb4);
    }
    if (res && res != (brov_result*)pr) std::memcpy(res, pr, B * sizeof(brov_result));
    // a caller that builds its inputs IN the staging buffers (brov_tick_buffers) is free to write the next tick's as soon as this call is back:
    // the refresh copies out of them are over by then
    if (zerocopy && in_place) HIPCHK(hipEventSynchronize(s->ev_copy));
    return BROV_OK;
}

extern "C" int brov_set_opts(brov_solver* s, const brov_opts* o) {
    if (s && s->prep_path == 2) s->prep_path = 3;   // a parked preparation (rti_phase 1, resident kernel) does not survive this call
    if (!s || !o || o->N != s->N) { g_err = "brov_set_opts: bad argument (N is fixed at create)"; return BROV_ERR_ARG; }
    if (const char* why = opts_problem(o)) { g_err = std::string("brov_set_opts: ") + why; return BROV_ERR_ARG; }
    if (o->kernel_path == BROV_PATH_FUSED && !fused_supported(o->N) && !s->ws) {
        // the windowed kernel's workspace is allocated at create, from the path and batch asked for then
        g_err = "brov_set_opts: BROV_PATH_FUSED at this horizon needs the windowed kernel's workspace, which this solver was created without "
                "(created with BROV_PATH_STREAMING): create it with BROV_PATH_AUTO or BROV_PATH_FUSED";
        return BROV_ERR_ARG;
    }
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    s->opts = *o;
    return upload_cst(s);
}
extern "C" int brov_get_opts(const brov_solver* s, brov_opts* o) {
    if (!s || !o) return BROV_ERR_ARG;
    *o = s->opts;
    return BROV_OK;
}
extern "C" int brov_synchronize(brov_solver* s, void* stream) {
    if (!s) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return BROV_OK;
}
extern "C" int brov_last_kernel_path(const brov_solver* s) {
    return s ? (s->last_fused ? BROV_PATH_FUSED : (s->last_windowed ? BROV_PATH_WINDOWED : BROV_PATH_STREAMING)) : BROV_ERR_ARG;
}
// which instances of the LAST solve were completed by the parallel-in-time kernel (rti_pit_kernel, batches the resident windowed mode
// serves): done[b] = 1, else 0 -- all zero when that kernel did not run.  Test / bench instrumentation.
// development (BROV_TICK_BREAKDOWN=1 at create): host time of the last brov_tick_host in microseconds -- [0] entry to launch (device selection,
// staging: copies into the pinned buffer, waits for earlier refresh copies), [1] brov_solve_phase (parameter block, kernel launch(es)), [2] what is
// enqueued behind the launch (events, refresh copies), [3] the wait for the records (mailbox poll / event / stream), [4] the whole call
extern "C" int brov_dev_tick_breakdown(brov_solver* s, double us[5]) {
    if (!s || !us) return BROV_ERR_ARG;
    for (int k = 0; k < 5; k++) us[k] = s->tick_us[k];
    return BROV_OK;
}
extern "C" int brov_pit_last(brov_solver* s, int32_t* done) {
    if (!s || !done) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    if (!s->pit_ran) { std::memset(done, 0, (size_t)s->B * sizeof(int32_t)); return BROV_OK; }
    HIPCHK(hipMemcpy(done, s->pit_done, (size_t)s->B * sizeof(int32_t), hipMemcpyDeviceToHost));
    return BROV_OK;
}
extern "C" int brov_lds_kernel_info(const brov_solver* s, int32_t info[4]) {
    if (!s || !info) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    const bool fused = fused_supported(s->N) && !s->force_windowed;
    // streaming kernels (asked for, or forced by a general grid, or no windowed workspace): stage blocks in HBM, nothing to report
    if (s->opts.kernel_path == BROV_PATH_STREAMING || (!fused && !s->ws)) { info[0] = info[1] = info[2] = info[3] = 0; return BROV_OK; }
    lds_kernel_info(s->N, s->win_L, !fused, info, s->k);
    return BROV_OK;
}
extern "C" int brov_window_stages(const brov_solver* s) { return s ? (s->ws ? s->win_L : 0) : BROV_ERR_ARG; }
extern "C" int brov_debug_dump_linearisation(brov_solver* s, int enable) {
    if (!s) return BROV_ERR_ARG;
    s->dump_lin = enable != 0;
    return BROV_OK;
}
// developer hook (not in the public header): per-instance phase timestamps of the last solve, 8 x uint64 per instance
extern "C" int brov_debug_phase_stamps(brov_solver* s, int enable, unsigned long long* out_host) {
    if (!s) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    // [B][8] phase stamps followed by [B][8] interior-point phase totals (the latter only filled by a -DBROV_DBG_IPM build)
    if (enable && !s->dbg) { HIPCHK(hipMalloc((void**)&s->dbg, (size_t)s->B * 128)); HIPCHK(hipMemset(s->dbg, 0, (size_t)s->B * 128)); }
    if (out_host && s->dbg) HIPCHK(hipMemcpy(out_host, s->dbg, (size_t)s->B * (enable == 2 ? 128 : 64), hipMemcpyDeviceToHost));
    if (!enable && s->dbg) { hipFree(s->dbg); s->dbg = nullptr; }
    return BROV_OK;
}
extern "C" int brov_enable_timing(brov_solver* s, int on) { if (!s) return BROV_ERR_ARG; s->timing = on != 0; s->ev_valid = false; return BROV_OK; }
extern "C" int brov_last_solve_seconds(brov_solver* s, double* total, double* k2) {
    if (!s || !s->ev_valid) return BROV_ERR_ARG;
    HIPCHK(hipEventSynchronize(s->ev[2]));
    float a = 0, b = 0;
    HIPCHK(hipEventElapsedTime(&a, s->ev[0], s->ev[1]));
    HIPCHK(hipEventElapsedTime(&b, s->ev[1], s->ev[2]));
    if (total) *total = (a + b) * 1e-3;
    if (k2) { k2[0] = a * 1e-3; k2[1] = b * 1e-3; }
    return BROV_OK;
}

extern "C" int brov_get_results_host(brov_solver* s, brov_result* res) {
    if (!s || !res) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    HIPCHK(hipMemcpy(res, s->res, (size_t)s->B * sizeof(brov_result), hipMemcpyDeviceToHost));
    return BROV_OK;
}
extern "C" int brov_get_u0_host(brov_solver* s, double* u0) {
    if (!s || !u0) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    HIPCHK(hipMemcpy2D(u0, 4 * sizeof(double), s->res, sizeof(brov_result), 4 * sizeof(double), s->B, hipMemcpyDeviceToHost));
    return BROV_OK;
}
extern "C" const brov_result* brov_results_device(const brov_solver* s) { return s ? s->res : nullptr; }
extern "C" double* brov_x0_device(brov_solver* s) { return s ? s->x0 : nullptr; }
extern "C" double* brov_yref_device(brov_solver* s) { if (!s) return nullptr; s->yref_shared = false; s->yref_view = nullptr; s->traj_line = -1; return s->yref; }
extern "C" double* brov_params_device(brov_solver* s) { if (!s) return nullptr; s->pplant_stale = true; return s->par; }
extern "C" double* brov_x_device(brov_solver* s) { return s ? s->x : nullptr; }
extern "C" double* brov_u_device(brov_solver* s) { return s ? s->u : nullptr; }

extern "C" int brov_get_linearisation_host(brov_solver* s, double* AB, double* b) {
    if (!s) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    if (AB) HIPCHK(hipMemcpy(AB, s->BA, (size_t)s->B * s->N * 192 * sizeof(double), hipMemcpyDeviceToHost));
    if (b) HIPCHK(hipMemcpy(b, s->bvec, (size_t)s->B * s->N * 12 * sizeof(double), hipMemcpyDeviceToHost));
    return BROV_OK;
}

// arg-min of cost over successful instances: one block, strided scan + LDS tree
__global__ void select_best_kernel(const brov_result* __restrict__ res, int B, int* __restrict__ out) {
    __shared__ double sc[256];
    __shared__ int si[256];
    double best = 1e300;
    int bi = -1;
    for (int k = threadIdx.x; k < B; k += blockDim.x) {
        const double c = res[k].cost;
        if (res[k].status == BROV_STATUS_SUCCESS && c == c && (c < best || (c == best && k < bi))) { best = c; bi = k; }
    }
    sc[threadIdx.x] = best; si[threadIdx.x] = bi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            const double c2 = sc[threadIdx.x + o];
            const int i2 = si[threadIdx.x + o];
            if (i2 >= 0 && (si[threadIdx.x] < 0 || c2 < sc[threadIdx.x] || (c2 == sc[threadIdx.x] && i2 < si[threadIdx.x]))) {
                sc[threadIdx.x] = c2; si[threadIdx.x] = i2;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = si[0];
}

extern "C" int brov_select_best_host(brov_solver* s, int* best_index, brov_result* best) {
    if (!s || !best_index) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    hipLaunchKernelGGL(select_best_kernel, dim3(1), dim3(256), 0, s->last_stream, s->res, s->B, s->best);
    HIPCHK(hipGetLastError());
    HIPCHK(sync_last(s));
    int idx = -1;
    HIPCHK(hipMemcpy(&idx, s->best, sizeof(int), hipMemcpyDeviceToHost));
    *best_index = idx;
    if (best && idx >= 0) HIPCHK(hipMemcpy(best, s->res + idx, sizeof(brov_result), hipMemcpyDeviceToHost));
    return BROV_OK;
}

extern "C" int brov_get_thrusts_host(brov_solver* s, double* t6) {
    if (!s || !t6) return BROV_ERR_ARG;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(sync_last(s));
    HIPCHK(hipMemcpy2D(t6, 6 * sizeof(double), (const char*)s->res + offsetof(brov_result, thrust), sizeof(brov_result), 6 * sizeof(double),
                       s->B, hipMemcpyDeviceToHost));
    return BROV_OK;
}
