/*
 * bluerov2_nmpc.h -- C ABI of the MI355X-native batched BlueROV2 NMPC (SQP real-time-iteration) solver.
 *
 * One handle owns B independent OCP instances (12 states / 4 inputs / 16 parameters, N shooting intervals) resident
 * in the HBM of ONE GPU.  brov_solve() performs, for every instance, exactly what ONE call of the reference's
 *     bluerov2_acados_solve(capsule)      bluerov2_dobmpc/scripts/c_generated_code/acados_solver_bluerov2.c:945-951
 * performs for its single instance (ocp_nlp_solve with SQP_RTI: ERK4+sensitivities, Gauss-Newton LS cost, box QP,
 * full step), and the setters replace the per-tick acados calls of the callers:
 *     brov_set_x0      <- ocp_nlp_constraints_model_set(..,0,"lbx"/"ubx",x0)   bluerov2_dob.cpp:320-321, ctrller/mpc.cpp:121-137
 *     brov_set_params  <- bluerov2_acados_update_params(capsule,i,p,16)         bluerov2_dob.cpp:324-355, acados_solver_bluerov2.c:835-883
 *     brov_set_yref    <- ocp_nlp_cost_model_set(..,i,"yref",yref[i])           bluerov2_dob.cpp:370-372, ctrller/mpc.cpp:46-58
 *     brov_get_u0      <- ocp_nlp_out_get(..,0,"u",u0)                          bluerov2_dob.cpp:388
 *     brov_get_results <- status / nlp_out->inf_norm_res / "time_tot"           bluerov2_dob.cpp:377-386
 *     brov_set/get_iterate <- ocp_nlp_out_set/get(..,"x"/"u")                   main_bluerov2.c:211-247
 *     brov_reset       <- bluerov2_acados_reset                                 acados_solver_bluerov2.c:797-830
 * The batch=1 acados-shaped shim (include/acados_shim/, libacados_ocp_solver_bluerov2.so) is a veneer over this API.
 *
 * Plain C: pointers + sizes, no C++/torch types.  Every pointer argument is tagged HOST or DEVICE below; DEVICE
 * pointers must be valid on the handle's GPU.  All calls on one handle must come from one thread at a time
 * (same contract as an acados capsule).  Layouts are instance-major, row-major, FP64:
 *     x0   [B][12]            yref [B][N+1][16] (or shared [N+1][16])      p [B][16] or [B][N+1][16]
 *     x    [B][N+1][12]       u    [B][N][4]      pi [B][N][12]            lam [B][N][8] = [lower4 | upper4]
 * There is no CPU fallback: every entry point that needs the GPU returns BROV_ERR_NO_DEVICE when none is usable.
 */
#ifndef BLUEROV2_NMPC_H_
#define BLUEROV2_NMPC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BROV_NX 12
#define BROV_NU 4
#define BROV_NP 16
#define BROV_NY 16

/* error codes of the API itself (solver status per instance uses the acados codes below) */
#define BROV_OK 0
#define BROV_ERR_ARG -1
#define BROV_ERR_NO_DEVICE -2
#define BROV_ERR_HIP -3
#define BROV_ERR_ALLOC -4

/* per-instance solver status == acados return codes (acados/utils/types.h as used by the callers) */
#define BROV_STATUS_SUCCESS 0
#define BROV_STATUS_NAN 1
#define BROV_STATUS_MAXITER 2
#define BROV_STATUS_MINSTEP 3
#define BROV_STATUS_QP_FAILURE 4

/* options; brov_default_opts() fills the values baked into the reference's generated solver
 * (acados_solver_bluerov2.c: W :422-481, bounds :559-566, qp_iter_max :668, Ts :389) */
typedef struct brov_opts {
    int32_t N;              /* shooting intervals, 1..BROV_MAX_N */
    int32_t qp_iter_max;    /* 50 */
    double  Ts;             /* uniform interval length [s]; cost scaling of stages 0..N-1 */
    double  W[BROV_NY];     /* diagonal stage weight on y = [x;u] */
    double  We[BROV_NX];    /* diagonal terminal weight */
    double  lbu[BROV_NU];
    double  ubu[BROV_NU];
    double  qp_tol_mu;      /* interior point, bound resolution (1e-7): at termination every input is within this distance of a bound or
                             * the bound's multiplier divided by the input's weight is below it */
    double  qp_tol_stat;    /* interior point, stationarity target (1e-9, tracked residual, absolute) */
    int32_t qp_early_exit;  /* 1: accept the equality-constrained minimiser when it satisfies the bounds (exact) */
    int32_t kernel_path;    /* BROV_PATH_AUTO (LDS-resident kernels: whole horizon for N <= 23, windowed above), _STREAMING, _FUSED */
    int32_t on_failure;     /* what happens to an instance whose step fails (status NAN / MINSTEP / QP_FAILURE):
                             *   BROV_ON_FAILURE_KEEP    iterate left untouched -- what acados' SQP_RTI does (it returns before
                             *                           update_variables); a diverged iterate then fails again every tick
                             *   BROV_ON_FAILURE_RESTART (default) cold restart at the measured state (if x0 is finite): x_i = x0 for all i, u = 0,
                             *                           multipliers 0, so that the instance can recover on the next tick
                             * In both cases the record's u0 is the last successfully computed input (zero-order hold), clamped
                             * to [lbu, ubu] with NaN -> 0: plant / thrust consumers never see a diverged input. */
    int32_t reserved_;
} brov_opts;

#define BROV_ON_FAILURE_KEEP 0
#define BROV_ON_FAILURE_RESTART 1

#define BROV_PATH_AUTO 0
#define BROV_PATH_STREAMING 1 /* lin_wave_kernel + qp_kernel, stage blocks streamed through HBM; any N <= BROV_MAX_N */
#define BROV_PATH_FUSED 2     /* one kernel, one wavefront per instance, stage blocks in LDS: the whole horizon (N <= 23) or a
                               * window of <= 20 stages at a time with the other windows parked in a per-instance HBM image */

#define BROV_PATH_WINDOWED 3  /* reported by brov_last_kernel_path only: the windowed flavour of BROV_PATH_FUSED / _AUTO (N >= 24) */

/* Longest horizon.  The reference's create_with_discretization takes any N (acados_solver_bluerov2.c:734-783).  Here the QP loop keeps the 4 N
 * inputs of an instance as elements per lane of its wavefront: 8 per lane as register copies in the LDS-resident kernels (N <= 128 =
 * BROV_MAX_N_LDS), 16 per lane read from HBM element by element beyond (N <= 256; round 5): the streaming pair and the large-batch windowed
 * kernel's long-horizon instantiations (rti_window_kernel_long, _long_grid, _long_ticks), which BROV_PATH_AUTO runs there.  N > 256 is refused by brov_create with BROV_ERR_ARG (the drop-in's create
 * returns non-zero).  (The reference ships N = 80.) */
#define BROV_MAX_N 256
#define BROV_MAX_N_LDS 128

/* 104-byte per-instance result record; this is also the record all-gathered across GPUs ("optimal
 * thrusts/costs for selection") */
typedef struct brov_result {
    double  u0[BROV_NU];    /* optimal first input after the step  (ocp_nlp_out_get(..,0,"u")) */
    double  cost;           /* NLS objective at the updated iterate */
    double  kkt;            /* NLP KKT inf-norm of the iterate entering the step (-> ocp_nlp_out::inf_norm_res) */
    int32_t status;         /* BROV_STATUS_* */
    int32_t qp_iter;        /* interior-point iterations used (0 = early exit) */
    double  thrust[6];      /* thrust allocation of u0, written by the solve kernel (bluerov2_dob.cpp:390-395) */
} brov_result;

typedef struct brov_solver brov_solver; /* opaque */

void brov_default_opts(brov_opts* o, int N, double Ts);

/* lifecycle.  device = HIP device ordinal.  Allocates all HBM state for B instances; initial iterate, yref, p and
 * x0 are the reference's create defaults (x_i=[0,0,-20,0..], u_i=0, yref=0, p=0; acados_solver_bluerov2.c:355-364,
 * 681-708). */
int  brov_create(brov_solver** out, int device, int B, const brov_opts* opts);
void brov_destroy(brov_solver* s);
int  brov_batch(const brov_solver* s);
int  brov_horizon(const brov_solver* s);
size_t brov_device_bytes(const brov_solver* s);
const char* brov_last_error(void);

/* inputs.  *_host variants copy from HOST memory (blocking on the solver's stream); *_device variants take DEVICE
 * pointers and enqueue a device-to-device copy on `stream` (NULL = default stream); no host sync. */
int brov_set_x0_host(brov_solver* s, const double* x0 /*[B][12]*/);
int brov_set_x0_device(brov_solver* s, const double* x0, void* stream);
/* shared != 0: one [N+1][16] window used by every instance; else per-instance [B][N+1][16] */
int brov_set_yref_host(brov_solver* s, const double* yref, int shared);
int brov_set_yref_device(brov_solver* s, const double* yref, int shared, void* stream);
/* per_stage == 0: p[B][16] applied to all stages (what every reference caller does); else p[B][N+1][16] */
int brov_set_params_host(brov_solver* s, const double* p, int per_stage);
int brov_set_params_device(brov_solver* s, const double* p, int per_stage, void* stream);
/* single-instance, single-stage setters (the shim's update_params / cost_model_set); stage in [0,N] */
int brov_set_param_stage_host(brov_solver* s, int instance, int stage, const double* p16);
int brov_set_yref_stage_host(brov_solver* s, int instance, int stage, const double* y, int ny);

/* ---- reference windows built on the device (the step before the path: bluerov2_path/src/bluerov2_path.cpp:79-118,
 * bluerov2_dobmpc/src/bluerov2_dob.cpp:218-265).  A trajectory table [rows][16] (the reference's txt format: x y z phi theta
 * psi u v w p q r u1..u4, one row per 0.05 s) is kept in HBM; node i of a window that starts at `line` is row
 * min(line + i, rows - 1).  ncols = 16 copies whole rows (DOB node), ncols = 12 leaves the input reference at zero (CTRL
 * node, ctrller/mpc.cpp:242-262). */
int brov_traj_set_host(brov_solver* s, const double* traj /*[rows][16]*/, int rows);
int brov_traj_rows(const brov_solver* s);
/* one shared window; with ncols = 16 and the window inside the table its rows are used where they lie (no copy, no kernel) */
int brov_set_yref_from_traj(brov_solver* s, int line, int ncols, void* stream);
int brov_set_yref_from_traj_lines_host(brov_solver* s, const int32_t* lines /*[B]*/, int ncols);     /* per instance      */
/* analytic per-instance candidate windows (bluerov2_path/config/traj/lemniscate.py:18-39, circle.py:22-56 evaluated at
 * t0 + i*dt with per-instance shape parameters and phase): kind 0 lemniscate (p0 = amplitude, p1 = frequency),
 * kind 1 circle (p0 = radius, p1 = speed) */
int brov_set_yref_candidates_host(brov_solver* s, int kind, const double* p0, const double* p1, const double* phase /*[B]*/,
                                  double t0, double dt);
/* the same in two steps for per-tick use: shape parameters uploaded once (HOST pointers), then one kernel per tick on `stream`
 * (no host traffic) that rebuilds every instance's window at t0 + i*dt */
int brov_set_candidate_params_host(brov_solver* s, int kind, const double* p0, const double* p1, const double* phase /*[B]*/);
int brov_set_yref_candidates(brov_solver* s, double t0, double dt, void* stream);
/* read back the reference windows currently in force, as [B][N+1][16] (shared windows are replicated) */
int brov_get_yref_host(brov_solver* s, double* yref);
/* read back the model parameters currently in force, [B][N+1][16] */
int brov_get_params_host(brov_solver* s, double* par);

/* ---- closed-loop roll-outs on the device (the step after the path): plant update + reference advance ------------------
 * Plant = the OCP's own 12-state model (bluerov2.py:103-137) integrated with ERK4 over one control period with u0 of the
 * last solve and per-instance TRUE parameters (Monte-Carlo disturbance draws / model mismatch); the thrusters see
 * bluerov2_dob.cpp:390-395's allocation of that u0 (brov_get_thrusts_host). */
/* Without brov_plant_set_params_host the plant uses the controller's own stage-0 parameters, re-read whenever they may have
 * changed (any brov_set_params_* call, brov_params_device() hand-out) */
int brov_plant_set_params_host(brov_solver* s, const double* p /*[B][16]*/);
int brov_plant_step(brov_solver* s, double dt, int substeps, void* stream);   /* x0 <- ERK4(x0, u0, p_plant, dt) */
/* `ticks` control ticks entirely on the device: window(line0 + k) -> RTI step -> plant step; like the nodes, the window
 * advances one trajectory row per tick (bluerov2_dob.cpp:367-368) and the iterate is not shifted.  Optional HOST logs:
 * u_log [ticks][B][4] (applied inputs), x_log [ticks+1][B][12] (plant state before the first and after every tick),
 * st_log [ticks][B] (solver status). */
int brov_closed_loop(brov_solver* s, int ticks, int line0, int ncols, double dt, int substeps, double* u_log, double* x_log,
                     int32_t* st_log);
int brov_get_x0_host(brov_solver* s, double* x0 /*[B][12]*/);

/* iterate (warm start) access; any pointer may be NULL to skip that block */
int brov_set_iterate_host(brov_solver* s, const double* x, const double* u, const double* pi, const double* lam);
int brov_get_iterate_host(brov_solver* s, double* x, double* u, double* pi, double* lam);
int brov_reset(brov_solver* s); /* zero iterate like bluerov2_acados_reset */
int brov_init_iterate_default(brov_solver* s); /* create-time default iterate */

/* one RTI step for all B instances, enqueued on `stream` (NULL = default).  Asynchronous: returns after launch. */
int brov_solve(brov_solver* s, void* stream);
/* acados' rti_phase (main_bluerov2.c:217): 0 = preparation + feedback (== brov_solve), 1 = preparation only
 * (linearise at the current iterate), 2 = feedback only (QP + step with the current x0; needs a prior phase 1).  Batches of at most one
 * instance per CU at N <= 81: the preparation also runs the step-0 factor sweep -- it does not depend on x0 -- and
 * parks the factorised LDS image; the feedback is forward sweep + bound check + step + record (a batch of one at N = 80 through
 * brov_tick_host: 33 us from the measurement to u0, against 71 us for rti_phase 0); a call that changes the iterate, the grid or the options
 * between the two makes the feedback fail (BROV_ERR_ARG: repeat the preparation) -- and so does a feedback call with no fresh preparation
 * at all (none since the last step, e.g. a second rti_phase 2 on one preparation): the iterate that was linearised is gone.  acados itself
 * would re-solve the QP in its memory; the acados-shaped drop-in turns such a call into a whole step (rti_phase 0) instead of failing.
 * Elsewhere: linearisation / QP on the streaming pair. */
int brov_solve_phase(brov_solver* s, void* stream, int rti_phase);
/* `ticks` consecutive RTI steps of every instance, the measured state held, the shared reference window moving on `row_stride` rows of the
 * resident trajectory table per step (0: the window in force stays -- several SQP iterations on one problem, e.g. candidates iterated to
 * convergence; > 0 needs the window to come from brov_set_yref_from_traj).  Result: exactly that of
 *     for k in 0 .. ticks-1: brov_set_yref_from_traj(s, line + k * row_stride, ncols, stream); brov_solve(s, stream)
 * (records of the last step; status_log, DEVICE [ticks][B] or NULL, keeps every step's status) -- but on the uniform grid it is ONE launch in
 * which every instance goes on to its next step as soon as its own is done (rti_fused_kernel_ticks for N <= 23, rti_window_kernel_ticks for
 * longer horizons and batches beyond two instances per CU; smaller batches at long horizons, general grids and the streaming pair take a
 * launch per step): a step with a slow instance -- 13 .. 47 Newton systems on a diverging one -- no longer holds the whole batch at a
 * launch boundary, and there are no launch gaps.  For workloads whose steps do not depend on each other through the host (parameter sweeps,
 * candidate libraries, Monte-Carlo draws at a fixed measurement); a control loop that measures between two steps calls brov_solve. */
int brov_solve_ticks(brov_solver* s, void* stream, int ticks, int row_stride, int32_t* status_log /*DEVICE or NULL*/);
int brov_synchronize(brov_solver* s, void* stream);
/* Stream ordering.  One solver, one stream at a time: every call that enqueues work (brov_solve*, brov_plant_step, the *_device
 * setters, the trajectory / candidate window builders, brov_ekf_*_solver) first waits -- on the host -- for the stream the solver
 * used LAST when that is a different one (e.g. brov_tick_host's private stream, whose kernel may still be finishing when the tick
 * returns its records).  Calls on the same stream cost nothing.  brov_order_stream does the same wait explicitly, for callers that
 * work on the DEVICE pointers handed out below from a stream of their own. */
int brov_order_stream(brov_solver* s, void* stream);
/* One control tick with ONE host synchronisation: the inputs that changed since the last tick (NULL = unchanged; x0 [B][12], ONE
 * reference window shared by the batch [N+1][16], per-stage parameters [B][N+1][16]) are staged through a pinned buffer and copied
 * asynchronously on the solver's own stream, the step (rti_phase as brov_solve_phase) runs behind them, the result records come
 * back the same way -- for up to 64 instances without a copy: the kernel writes each record into the pinned buffer itself, followed
 * by a sequence word the host polls (BROV_TICK_MAILBOX=0 in the environment at brov_create: copy + stream synchronisation, as for larger
 * batches).  A tick that passes all three inputs uploads them with one copy.  This is what the acados-shaped drop-in makes of one
 * bluerov2_acados_solve (bluerov2_dob.cpp:306-388: lbx / ubx, (N+1) x update_params, (N+1) x yref, solve, u0 / status / kkt).
 * rti_phase 1 (a preparation) delivers no records: the call returns as soon as the inputs have left the pinned buffer, `res` is left alone,
 * and the preparation runs on, stream-ordered ahead of whatever follows.  HOST pointers. */
int brov_tick_host(brov_solver* s, const double* x0, const double* yref_shared, const double* par_stage, int rti_phase,
                   brov_result* res /*[B] or NULL*/);
/* The staging buffers brov_tick_host copies its arguments into / its records out of: device-visible pinned host memory, valid until
 * brov_destroy.  A caller that builds its inputs in them and reads its records from them -- pass exactly these pointers as x0 /
 * yref_shared / par_stage, and NULL as res -- saves the tick its host-side copies (0.8 MB per step at a batch of 4096).  The result
 * records are written there by the solve kernel itself (no copy command behind the launch; batches <= 64: with a sequence word per
 * instance that the host polls, larger ones: the host waits for the launch once; BROV_TICK_BULK=0: the copy command of round 3).
 * The input buffers may be rewritten as soon as brov_tick_host has returned (a tick that was handed these pointers waits for the copies it
 * enqueued out of them; a tick that copies its arguments in waits, before it does, for what an earlier tick left reading them). */
/* instrumentation: which instances of the last solve were completed by the parallel-in-time kernel */
int brov_pit_last(brov_solver* s, int32_t* done /*[B]*/);
int brov_tick_buffers(brov_solver* s, double** x0 /*[B][12]*/, double** yref_shared /*[N+1][16]*/, double** par_stage /*[B][N+1][16]*/,
                      const brov_result** res /*[B]*/);
/* Development knobs.  The solver's A/B switches and test hooks are BROV_* environment variables (BROV_PIT, BROV_TICK_MAILBOX,
 * BROV_PARTIAL_REFACTOR, ...: DESIGN.md section 4.6 lists them); a solver reads them ONCE, in brov_create -- no call on the path of a
 * solve touches the environment.  brov_dev_reload_knobs reads them again (tests that flip a switch between two solves of one solver). */
int brov_dev_reload_knobs(brov_solver* s);
/* BROV_TICK_BREAKDOWN=1: host microseconds of the last brov_tick_host -- staging, launch, post-launch enqueues, wait for the records, total */
int brov_dev_tick_breakdown(brov_solver* s, double us[5]);
/* replace weights / bounds / Ts / QP options of an existing solver (N must not change) */
int brov_set_opts(brov_solver* s, const brov_opts* opts);
int brov_get_opts(const brov_solver* s, brov_opts* opts);

/* outputs */
int brov_get_results_host(brov_solver* s, brov_result* res /*[B]*/); /* syncs the last solve's stream */
int brov_get_u0_host(brov_solver* s, double* u0 /*[B][4]*/);
const brov_result* brov_results_device(const brov_solver* s);          /* DEVICE pointer, [B] records */
/* DEVICE pointers to the resident state for zero-copy producers/consumers (layouts above) */
double* brov_x0_device(brov_solver* s);
double* brov_yref_device(brov_solver* s);   /* [B][N+1][16] (always allocated; used when shared == 0) */
double* brov_params_device(brov_solver* s); /* [B][N+1][16] */
double* brov_x_device(brov_solver* s);
double* brov_u_device(brov_solver* s);
/* linearisation of the LAST solve (row-major per stage: [A|B] as [12][16], b as [12]).  The streaming path always leaves it
 * in HBM; the LDS-resident kernels write it out only after brov_debug_dump_linearisation(s, 1).  For tests. */
int brov_get_linearisation_host(brov_solver* s, double* AB /*[B][N][12][16]*/, double* b /*[B][N][12]*/);
int brov_debug_dump_linearisation(brov_solver* s, int enable);

/* ---- boundary corners of the reference API -----------------------------------------------------------------------------------
 * Non-uniform grids: bluerov2_acados_create_with_discretization(capsule, N, new_time_steps) / bluerov2_acados_update_time_steps
 * (c_generated_code/acados_solver_bluerov2.h:141,146; .c:111-131) set the ERK4 step AND the cost scaling of stage i to
 * new_time_steps[i].  brov_set_time_steps does the same for the whole batch (ts[N]; NULL or a uniform vector = the uniform grid
 * with brov_opts::Ts).  Separate stage-0 weight: the generated solver carries W_0 next to W (.c:422-441, same numbers as
 * shipped); brov_set_stage0_weight(W0[16]) gives stage 0 its own (NULL or W itself = one stage weight).
 * Round 4: either feature runs on the LDS-resident kernels (grid instantiations of the fused kernel, of the windowed kernel, of its
 * resident mode and of the parallel-in-time step-0 kernel of small batches) as well as on the streaming pair. */
int brov_set_time_steps(brov_solver* s, const double* ts /*[N] or NULL*/);
int brov_set_stage0_weight(brov_solver* s, const double* W0 /*[16] or NULL*/);
int brov_general_grid(const brov_solver* s);   /* 1 while either feature is in force */

/* ---- 6-disturbance model variant (BASELINE configs[2] "6 disturbance states") ------------------
 * The reference's EKF estimates six disturbances (bluerov2_dob.h:200-205), its OCP model takes four: the roll / pitch symbols are
 * there but commented out (bluerov2_dobmpc/scripts/bluerov2.py:37-38), so the shipped solver has np = 16.  This switch (default
 * OFF = the shipped model) adds the two terms the way the other four enter their rows (bluerov2.py:123-128):
 *     dp += d_phi / Ix,   dq += d_theta / Iy            (additive: df/dx and df/du are untouched)
 * with d_phi, d_theta per instance and stage, next to p[16].  brov_set_params18_host takes the parameter vector the uncommented
 * model would have, [dx dy dz d_phi d_theta d_psi | added mass 4 | linear damping 4 | quadratic damping 4]; with the variant on,
 * brov_ekf_apply_to_solver fills all six disturbances from the estimate, and brov_plant_step integrates the controller's stage-0
 * values unless brov_plant_set_rp_disturbance_host gave the plant its own. */
int brov_enable_dist6(brov_solver* s, int on);
int brov_dist6_enabled(const brov_solver* s);
int brov_set_rp_disturbance_host(brov_solver* s, const double* d /*[B][2] or [B][N+1][2]*/, int per_stage);
int brov_set_rp_disturbance_device(brov_solver* s, const double* d, int per_stage, void* stream);
int brov_set_params18_host(brov_solver* s, const double* p18 /*[B][18] or [B][N+1][18]*/, int per_stage);
double* brov_rp_disturbance_device(brov_solver* s);   /* [B][N+1][2]; NULL while the variant is off */
int brov_get_rp_disturbance_host(brov_solver* s, double* d /*[B][N+1][2]*/);
int brov_plant_set_rp_disturbance_host(brov_solver* s, const double* d /*[B][2]; NULL: back to the controller's stage 0*/);

/* argmin of cost over instances with status SUCCESS (config 4 "best-trajectory select"); writes the winning index
 * and its record; runs on the GPU, result copied to HOST */
int brov_select_best_host(brov_solver* s, int* best_index, brov_result* best);

/* ---- several GPUs in ONE process (BASELINE.json configs[3]: 65 536 candidates over 8 GPUs, all-gather of the
 * optimal cost / u* for best-trajectory select) ------------------------------------------------------------------------------------
 * A group owns one brov_solver per device -- a contiguous shard of `total_instances`, the first total % n shards one instance
 * larger --, one stream and one RCCL communicator per device.  There is no communication during the solve; brov_group_gather is ONE
 * ncclAllGather per device (inside ncclGroupStart / ncclGroupEnd, on the devices' own streams, behind the solve) of
 *     BROV_GATHER_RECORDS  every instance's 104-byte result record, to every device (uneven shards: padded with never-selectable slots)
 *     BROV_GATHER_PACKED   one (cost, global index) pair per device (16 B): the local arg-min -- when only the winner is wanted
 * and brov_group_select_best returns the global arg-min of cost over the instances with status SUCCESS (ties: lowest index) and waits
 * for it -- the only host wait of a step.  With BROV_GATHER_RECORDS the winner comes back through a pinned host mailbox the select kernel
 * (up to 64 blocks over the gathered records) writes itself: no copy command and no stream synchronisation on the way back; the call
 * returns when local device 0 has delivered, the other devices may still be finishing their gather (their streams order whatever is
 * enqueued next behind it; brov_group_synchronize waits for all of them).  Per-shard inputs go through the shard's own handle (brov_group_solver: every brov_set_* /
 * brov_get_* above works on it, with brov_group_stream(rank) as the stream) or through the whole-batch setters below, which slice a
 * global HOST array.  RCCL is loaded at run time (dlopen) by brov_group_create; a process that never creates a group never loads it.
 * The reference's callers live in one C++ process (bluerov2_dob.cpp:270-451): this is their route to several GPUs; one process per
 * GPU on torch.distributed (bluerov2_amd/distributed.py, bench.py --gpus N) is the other. */
typedef struct brov_group brov_group;
#define BROV_GATHER_RECORDS 0
#define BROV_GATHER_PACKED 1
int  brov_group_create(brov_group** out, const int* devices /*[n] distinct HIP ordinals*/, int n, int total_instances, const brov_opts* opts);
/* Which collective carries the gather.  BROV_COLLECTIVE_RCCL (what brov_group_create / brov_group_create_rank use): ncclAllGather.
 * BROV_COLLECTIVE_COPY: the all-gather as device-to-device copies -- every rank publishes its contribution behind an event on its own
 * stream, every rank pulls the others' into its own gathered array on its own stream.  RCCL is not loaded, and `devices` may name a GPU
 * more than once: W ranks on ONE GPU run the whole bookkeeping of W GPUs (shard bounds, padded staging, rank-major gathered layout, packed
 * pairs, the select over W x slots) -- the development and test route on a 1-GPU box, and a fallback where RCCL is not installed.  The
 * ranks of a copy group live in ONE process; brov_group_gather blocks the calling thread until every rank of the group has entered the
 * same gather (ranks held by other brov_group objects: call from one thread per object, as RCCL asks of ncclCommInitRank ranks that share
 * a process), and the id of brov_group_create_rank_ex may be any 128 bytes the ranks agree on.  The entry points put the calling thread's
 * current HIP device back before they return. */
#define BROV_COLLECTIVE_RCCL 0
#define BROV_COLLECTIVE_COPY 1
int  brov_group_create_ex(brov_group** out, const int* devices /*[n]*/, int n, int total_instances, const brov_opts* opts, int collective);
/* The same group with ONE PROCESS PER GPU (the launcher's route, e.g. torchrun): every process creates its rank of the group on its own
 * device.  Rank 0 calls brov_group_unique_id and hands the 128 bytes to the other ranks by whatever means the launcher has (a file, MPI,
 * torch.distributed.broadcast); counts[world] = instances per rank.  All entry points below then act on the local shard, the gather
 * spans all ranks; brov_group_select_best returns the global index everywhere and the full record where the winner is local or the
 * 104-byte records were gathered (with BROV_GATHER_PACKED and a remote winner: cost and status only). */
int  brov_group_unique_id(char id[128]);
int  brov_group_create_rank(brov_group** out, int device, int rank, int world, const char id[128], const int* counts /*[world]*/, const brov_opts* opts);
int  brov_group_create_rank_ex(brov_group** out, int device, int rank, int world, const char id[128], const int* counts /*[world]*/, const brov_opts* opts,
                               int collective);
int  brov_group_collective(const brov_group* g);     /* BROV_COLLECTIVE_* */
int  brov_group_set_copy_wait_seconds(int seconds);  /* BROV_COLLECTIVE_COPY: how long a gather waits for a rank that has not entered it (60) */
void brov_group_destroy(brov_group* g);
const char* brov_group_last_error(void);
int  brov_group_rccl_version(int* version);          /* loads RCCL if need be; BROV_ERR_HIP when it cannot be loaded */
int  brov_group_size(const brov_group* g);           /* devices held by THIS process (local indices 0 .. size-1 address solver / stream below) */
int  brov_group_world(const brov_group* g);          /* ranks of the whole group (= size in the one-process form) */
int  brov_group_first_rank(const brov_group* g);     /* global rank of local device 0 */
int  brov_group_total(const brov_group* g);
int  brov_group_shard(const brov_group* g, int rank, int* lo, int* hi);   /* global instance range [lo, hi) of GLOBAL rank `rank` */
brov_solver* brov_group_solver(brov_group* g, int local);   /* shard handle of local device `local` */
void* brov_group_stream(brov_group* g, int local);    /* its hipStream_t */
int brov_group_set_x0_host(brov_group* g, const double* x0 /*[total][12]*/);
int brov_group_set_params_host(brov_group* g, const double* p /*[total][16] or [total][N+1][16]*/, int per_stage);
int brov_group_set_yref_host(brov_group* g, const double* yref /*shared [N+1][16] or [total][N+1][16]*/, int shared);
int brov_group_set_candidate_params_host(brov_group* g, int kind, const double* p0, const double* p1, const double* phase /*[total]*/);
int brov_group_set_yref_candidates(brov_group* g, double t0, double dt);   /* one window kernel per device, no host traffic */
int brov_group_solve(brov_group* g);                 /* one RTI step of every shard, enqueued; returns after the launches */
int brov_group_gather(brov_group* g, int mode);      /* enqueued behind the solve */
int brov_group_select_best(brov_group* g, int* best_index /*global; -1: no instance qualifies*/, brov_result* best /*or NULL*/);
int brov_group_synchronize(brov_group* g);
int brov_group_get_results_host(brov_group* g, brov_result* res /*[total]*/);   /* after a BROV_GATHER_RECORDS gather: device 0's copy */
const brov_result* brov_group_gathered_device(const brov_group* g, int rank);   /* DEVICE: [n][slots_per_rank] records on shard `rank`'s GPU */
int brov_group_slots_per_rank(const brov_group* g);
int brov_group_enable_timing(brov_group* g, int on);  /* HIP events around solve / gather / select (default on) */
int brov_group_last_seconds(brov_group* g, double* solve, double* gather, double* select);   /* slowest device each */

/* thrust allocation (bluerov2_dob.cpp:390-395): t[B][6] = the `thrust` field the solve kernel wrote into every result record */
int brov_get_thrusts_host(brov_solver* s, double* t6 /*[B][6]*/);

/* timing of the last brov_solve (HIP events on its stream), seconds: total and per kernel [linearise, qp] */
int brov_last_solve_seconds(brov_solver* s, double* total, double* kernels2);
int brov_enable_timing(brov_solver* s, int on);
/* which kernels the last brov_solve launched: BROV_PATH_FUSED, BROV_PATH_WINDOWED or BROV_PATH_STREAMING */
int brov_last_kernel_path(const brov_solver* s);
/* stages per LDS window of the windowed kernel this solver was created for (0: no windowed workspace).  Equal to N: resident mode,
 * the whole horizon in one window (batches of at most one instance per CU at N <= 81); otherwise <= 20 */
int brov_window_stages(const brov_solver* s);
/* what the LDS-resident kernel this solver launches asks of a compute unit: info = {dynamic LDS bytes per block (= per instance in
 * flight), blocks per CU granted by the occupancy query, threads per block, kind: 1 whole horizon / 2 whole horizon, two waves per
 * SIMD / 3 windowed / 4 windowed, resident mode}; all 0 when the solver runs on the streaming kernels */
int brov_lds_kernel_info(const brov_solver* s, int32_t info[4]);

/* ---------------------------------------------------------------------------------------------------------------------
 * Batched EKF disturbance observer: B independent copies of the reference's 18-state filter
 * BLUEROV2_DOB::EKF() (bluerov2_dobmpc/src/bluerov2_dob.cpp:495-545) -- forward-difference Jacobians of the RK4 map
 * (:730-744, with the k2/3 stage quirk of :630) and of the measurement model (:747-762), gain through the explicit
 * inverse of the innovation covariance, Joseph-form covariance update.  One call = one EKF tick of every instance.
 * State [pose(6) | body velocities(6) | body-frame disturbance wrench(6)], instance-major, FP64.
 * ------------------------------------------------------------------------------------------------------------------- */
typedef struct brov_ekf brov_ekf;
typedef struct brov_ekf_params { /* defaults: bluerov2_dob.h:171-183,208 and bluerov2_dob.cpp:41-62 */
    double dt;
    double mass, Ix, Iy, Iz, ZG, g, bouyancy;
    double added_mass[6], Dl[6], Dnl[6];
    double K[36];          /* propulsion matrix, row-major: tau = K * thrust */
    double Q[18];          /* process noise (diagonal) */
    double R;              /* measurement noise R * I */
    double fd_step;        /* finite-difference step of both Jacobians (1e-6) */
    double compensate_coef, rotor_constant; /* scaling of the estimate into the NMPC parameters p[0..3] (:334-337) */
} brov_ekf_params;
void brov_ekf_default_params(brov_ekf_params* p);
const char* brov_ekf_last_error(void); /* message of the last failing brov_ekf_* call on this thread */
int  brov_ekf_create(brov_ekf** out, int device, int batch, const brov_ekf_params* p /* NULL: defaults */);
void brov_ekf_destroy(brov_ekf* e);
int  brov_ekf_batch(const brov_ekf* e);
/* every instance := (x0, P0); NULL = the reference's start [0,0,-20,0..0,6,6,6,0,0,0], P0 = I (bluerov2_dob.cpp:64-65) */
int brov_ekf_reset(brov_ekf* e, const double* x0 /*[18]*/, const double* P0 /*[18][18]*/);
int brov_ekf_set_state_host(brov_ekf* e, const double* x /*[B][18]*/, const double* P /*[B][18][18]*/);
int brov_ekf_get_state_host(brov_ekf* e, double* x /*[B][18] or NULL*/, double* P /*[B][18][18] or NULL*/);
/* one EKF tick.  thrust = meas_u (thruster outputs, :498), y12 = measured pose + body velocities (:501-502), acc = body
 * accelerations (finite differences of the velocities, :148-153).  HOST or DEVICE pointers respectively. */
int brov_ekf_update_host(brov_ekf* e, const double* thrust /*[B][6]*/, const double* y12 /*[B][12]*/,
                         const double* acc /*[B][6]*/, void* stream);
int brov_ekf_update_device(brov_ekf* e, const double* thrust, const double* y12, const double* acc, void* stream);
/* outputs of the last tick: world-frame disturbance (:540-545), NMPC parameters p[0..3] (:334-337), per-instance status
 * (0 ok; 1 innovation covariance not positive definite or NaN: estimate and covariance left at the prediction;
 *  2 non-finite estimate, e.g. a NaN measurement: propagated as the reference would) */
int brov_ekf_get_outputs_host(brov_ekf* e, double* wf /*[B][6] or NULL*/, double* mpc_p /*[B][4] or NULL*/,
                              int* status /*[B] or NULL*/);
const double* brov_ekf_x_device(const brov_ekf* e);      /* [B][18] */
const double* brov_ekf_P_device(const brov_ekf* e);      /* [B][18][18] */
const double* brov_ekf_mpc_p_device(const brov_ekf* e);  /* [B][4] */
/* close the DOB-MPC loop on the device (BASELINE config 3), no host round trip:
 *   brov_ekf_update_from_solver: measurement = the solver's x0 (the plant state after brov_plant_step), thrusts = thrust
 *     allocation of the solver's last u0 (:390-395) -- the thruster vector brov_plant_step applies --, accelerations =
 *     (v - v_prev)/dt with v_prev kept in the observer (zero after brov_ekf_reset, like the reference's pre_body_pos);
 *   brov_ekf_apply_to_solver: p[0..3] of every stage of instance b := the estimate of instance b (:332-337).
 * Units: the reference's plant is Gazebo, whose thrusters turn a command w into rotor_constant*w|w| newtons, and :334-337
 * rescale the estimated wrench by 1/compensate_coef resp. 1/rotor_constant into the OCP model's force units.  The device
 * plant is the OCP model itself (wrench = K * allocation(u)/rotor_constant), so an observer that is to compensate THAT
 * plant is created with compensate_coef = rotor_constant = 1 in brov_ekf_params. */
int brov_ekf_update_from_solver(brov_ekf* e, brov_solver* s, void* stream);
int brov_ekf_apply_to_solver(brov_ekf* e, brov_solver* s, void* stream);
/* seconds of the last update kernel (HIP events on its stream) */
int brov_ekf_last_update_seconds(brov_ekf* e, double* seconds);

#ifdef __cplusplus
}
#endif
#endif
