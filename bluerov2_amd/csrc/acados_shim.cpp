// acados_shim.cpp -- libacados_ocp_solver_bluerov2.so: the reference's acados-shaped call surface
// (/root/reference/bluerov2_dobmpc/scripts/c_generated_code/acados_solver_bluerov2.{h,c} plus the handful of ocp_nlp_*
// functions its callers use) implemented as a batch-of-one veneer over the MI355X solver (include/bluerov2_nmpc.h).
// Semantics follow the generated C, cited per function.  The capsule keeps HOST mirrors of x0 / yref / p / iterate so that
// the ~3(N+1) tiny setter calls a ROS node makes per tick cost nothing; brov_* uploads them once inside solve().
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "acados_solver_bluerov2.h"
#include "blasfeo/include/blasfeo_d_aux_ext_dep.h"
#include "bluerov2_nmpc.h"

struct brov_shim_state {
    brov_solver* solver = nullptr;
    brov_opts opts{};
    int N = 0;
    std::vector<double> x0, yref, par, x, u, pi, lam;
    std::vector<double> ts;        // per-stage time steps (create_with_discretization / update_time_steps); uniform ones collapse to opts.Ts
    double W0[16] = {0};           // stage-0 weight ("W" of stage 0)
    bool has_W0 = false, dirty_grid = false;
    bool dirty_x0 = true, dirty_yref = true, dirty_par = true, dirty_iter = false, dirty_opts = false;
    bool iter_host_valid = true;  // host mirror of the iterate is current
    int rti_phase = 0;
    int last_phase = 0;       // the phase the last call actually ran (a feedback call without a preparation runs phase 0)
    bool prepared = false;    // a preparation (rti_phase 1) of the current iterate, under the current options, is parked in the solver
    double time_tot = 0.0, time_lin = 0.0, time_qp = 0.0;
    bool times_valid = false, timing_on = false;
    brov_result last{};
    int last_status = 0;      // NLP status of the last call as acados reports it (ocp_nlp_get "status")
    int last_qp_status = 0;   // the QP's own verdict (ocp_nlp_get "qp_status", the qp_stat column of "statistics")
    bool strict_maxiter = false;   // BROV_SHIM_MAXITER_STATUS=2: hand the batched API's status 2 through instead of acados' 0
};

static brov_shim_state* st_of(bluerov2_solver_capsule* c) { return c ? c->shim : nullptr; }

static void pull_iterate(brov_shim_state* s) {
    if (s->iter_host_valid) return;
    brov_get_iterate_host(s->solver, s->x.data(), s->u.data(), s->pi.data(), s->lam.data());
    s->iter_host_valid = true;
}

extern "C" {

bluerov2_solver_capsule* bluerov2_acados_create_capsule(void) {  // acados_solver_bluerov2.c:87-93
    bluerov2_solver_capsule* c = (bluerov2_solver_capsule*)std::calloc(1, sizeof(bluerov2_solver_capsule));
    return c;
}
int bluerov2_acados_free_capsule(bluerov2_solver_capsule* c) {  // :96-100
    std::free(c);
    return 0;
}

int bluerov2_acados_create_with_discretization(bluerov2_solver_capsule* c, int N, double* new_time_steps) {  // :734-783
    if (!c) return 1;
    if (N != BLUEROV2_N && !new_time_steps) {  // :737-743
        std::fprintf(stderr,
                     "bluerov2_acados_create_with_discretization: new_time_steps is NULL but the number of shooting intervals "
                     "(= %d) differs from the number of shooting intervals (= %d) during code generation! Please provide a new "
                     "vector of time_stamps!\n", N, BLUEROV2_N);
        return 1;
    }
    double Ts = 0.0125;  // :389
    if (new_time_steps) {
        Ts = new_time_steps[0];
        for (int i = 0; i < N; i++)
            if (!(new_time_steps[i] > 0.0)) { std::fprintf(stderr, "bluerov2 (MI355X shim): time steps must be positive\n"); return 1; }
    }
    brov_shim_state* s = new brov_shim_state();
    s->N = N;
    if (new_time_steps) { s->ts.assign(new_time_steps, new_time_steps + N); s->dirty_grid = true; }   // :375-387 -> update_time_steps
    brov_default_opts(&s->opts, N, Ts);
    // the generated solver carries W_0 next to W from the start (:422-441, same numbers as shipped): stage 0 keeps ITS weight when a
    // caller later changes "W" of the stages 1..N-1 only
    std::memcpy(s->W0, s->opts.W, sizeof s->W0);
    s->has_W0 = true;
    // a failed step: acados' SQP_RTI returns before update_variables and leaves everything as it is -- so does the drop-in by
    // default (BROV_ON_FAILURE_KEEP).  BROV_ON_FAILURE=restart in the environment opts in to the batched API's default, a cold
    // start of the failed iterate at the measured state ("Failed steps").
    {
        const char* of = std::getenv("BROV_ON_FAILURE");
        s->opts.on_failure = (of && (!std::strcmp(of, "restart") || !std::strcmp(of, "RESTART") || !std::strcmp(of, "1")))
                                 ? BROV_ON_FAILURE_RESTART : BROV_ON_FAILURE_KEEP;
    }
    // SQP_RTI's return code when the QP stopped at its iteration limit (qp_iter_max, :668): upstream takes the step and returns
    // ACADOS_SUCCESS -- the node's `if (acados_status != 0) return;` (mpc.cpp:63-68) therefore publishes
    // on such a tick -- and so does the drop-in; the QP's own status 2 stays visible through "qp_status" / "statistics" /
    // print_stats.  BROV_SHIM_MAXITER_STATUS=2 hands the batched API's 2 through instead (a stricter caller's choice).
    {
        const char* ms = std::getenv("BROV_SHIM_MAXITER_STATUS");
        s->strict_maxiter = ms && std::atoi(ms) == ACADOS_MAXITER;
    }
    // which GPU: the reference has no such notion; BROV_DEVICE selects one on a multi-GPU host (default 0)
    const char* dev_env = std::getenv("BROV_DEVICE");
    const int device = dev_env ? std::atoi(dev_env) : 0;
    int rc = brov_create(&s->solver, device, 1, &s->opts);
    if (rc != BROV_OK) {
        std::fprintf(stderr, "bluerov2_acados_create: MI355X solver unavailable (%d): %s\n", rc, brov_last_error());
        delete s;
        return rc == BROV_ERR_NO_DEVICE ? ACADOS_QP_FAILURE : 1;  // non-zero: the callers exit(1) (bluerov2_dob.cpp:35-38)
    }
    // kernel times (time_lin / time_qp_sol) come from HIP events around the kernels, which cost ~10 us of every tick: no reference
    // caller asks for them (they read time_tot, host wall time), so the events are recorded only from the first request on -- or
    // from the start with BROV_SHIM_TIMING=1
    s->timing_on = std::getenv("BROV_SHIM_TIMING") && std::atoi(std::getenv("BROV_SHIM_TIMING")) != 0;
    brov_enable_timing(s->solver, s->timing_on ? 1 : 0);
    const size_t n1 = (size_t)N + 1;
    s->x0.assign(12, 0.0);
    s->x0[2] = -20.0;  // :520-527 (lbx0 = ubx0 = [0,0,-20,0..])
    s->yref.assign(n1 * 16, 0.0);  // :405-420
    s->par.assign(n1 * 16, 0.0);   // :355-364
    s->x.assign(n1 * 12, 0.0);
    for (size_t i = 0; i < n1; i++) s->x[i * 12 + 2] = -20.0;  // :689-706
    s->u.assign((size_t)N * 4, 0.0);
    s->pi.assign((size_t)N * 12, 0.0);
    s->lam.assign((size_t)N * 8, 0.0);
    c->shim = s;
    c->nlp_np = BLUEROV2_NP;
    c->nlp_solver_plan = new ocp_nlp_plan_t{N, s};
    c->nlp_config = new ocp_nlp_config{N, s};
    c->nlp_dims = new ocp_nlp_dims{N, 12, 4, 16, 16, 12, s};
    c->nlp_in = new ocp_nlp_in{s};
    c->nlp_out = new ocp_nlp_out{0.0, s};
    c->sens_out = new ocp_nlp_out{0.0, s};
    c->nlp_solver = new ocp_nlp_solver{s};
    c->nlp_opts = (void*)s;
    c->forw_vde_casadi = (external_function_param_casadi*)std::calloc(N, sizeof(external_function_param_casadi));
    c->expl_ode_fun = (external_function_param_casadi*)std::calloc(N, sizeof(external_function_param_casadi));
    for (int i = 0; i < N; i++) {
        c->forw_vde_casadi[i].p = c->expl_ode_fun[i].p = &s->par[(size_t)i * 16];
        c->forw_vde_casadi[i].np = c->expl_ode_fun[i].np = 16;
    }
    return 0;
}

int bluerov2_acados_create(bluerov2_solver_capsule* c) {  // :103-108
    return bluerov2_acados_create_with_discretization(c, BLUEROV2_N, nullptr);
}

int bluerov2_acados_update_time_steps(bluerov2_solver_capsule* c, int N, double* ts) {  // :111-131
    brov_shim_state* s = st_of(c);
    if (!s || !ts) return 1;
    if (N != s->N) {
        std::fprintf(stderr,
                     "bluerov2_acados_update_time_steps: given number of time steps (= %d) differs from the currently allocated "
                     "number of time steps (= %d)!\nPlease recreate with new discretization and provide a new vector of time_stamps!\n",
                     N, s->N);
        return 1;
    }
    for (int i = 0; i < N; i++)
        if (!(ts[i] > 0.0)) { std::fprintf(stderr, "bluerov2 (MI355X shim): time steps must be positive\n"); return 1; }
    // :122-127: "Ts" and the cost "scaling" of every stage := new_time_steps[i] (a uniform vector is the uniform grid again)
    s->ts.assign(ts, ts + N);
    s->dirty_grid = true;
    return 0;
}

int bluerov2_acados_update_qp_solver_cond_N(bluerov2_solver_capsule*, int) {  // :788-794
    std::printf("\nacados_update_qp_solver_cond_N() failed, since no partial condensing solver is used!\n\n");
    std::exit(1);
    return -1;
}

int bluerov2_acados_reset(bluerov2_solver_capsule* c, int) {  // :797-830: iterate and multipliers to zero
    brov_shim_state* s = st_of(c);
    if (!s) return 1;
    std::fill(s->x.begin(), s->x.end(), 0.0);
    std::fill(s->u.begin(), s->u.end(), 0.0);
    std::fill(s->pi.begin(), s->pi.end(), 0.0);
    std::fill(s->lam.begin(), s->lam.end(), 0.0);
    // ... and the result record: its u0 is what the stage-0 "u" getter returns after a failed step (the held input), and a reset
    // must not hand the previous run's input on
    if (brov_reset(s->solver) != BROV_OK) return 1;
    s->last = brov_result{};
    s->last_status = 0;
    s->last_qp_status = 0;
    s->prepared = false;
    s->iter_host_valid = true;
    s->dirty_iter = false;   // brov_reset has zeroed the device iterate too
    return 0;
}

int bluerov2_acados_update_params(bluerov2_solver_capsule* c, int stage, double* p, int np) {  // :835-883
    const int casadi_np = 16;
    if (casadi_np != np) {
        std::printf("acados_update_params: trying to set %i parameters for external functions. External function has %i parameters. "
                    "Exiting.\n", np, casadi_np);
        std::exit(1);
    }
    brov_shim_state* s = st_of(c);
    if (!s || !p) return 1;
    if (stage < 0 || stage > s->N) stage = s->N;  // the generated code treats everything else as the terminal node
    std::memcpy(&s->par[(size_t)stage * 16], p, 16 * sizeof(double));
    s->dirty_par = true;
    return 0;
}

int bluerov2_acados_update_params_sparse(bluerov2_solver_capsule* c, int stage, int* idx, double* p, int n_update) {  // :886-942
    const int casadi_np = 16;
    if (casadi_np < n_update) {
        std::printf("bluerov2_acados_update_params_sparse: trying to set %d parameters for external functions. External function has "
                    "%d parameters. Exiting.\n", n_update, casadi_np);
        std::exit(1);
    }
    brov_shim_state* s = st_of(c);
    if (!s || !p || !idx) return 1;
    if (stage < 0 || stage > s->N) stage = s->N;
    for (int k = 0; k < n_update; k++)
        if (idx[k] >= 0 && idx[k] < 16) s->par[(size_t)stage * 16 + idx[k]] = p[k];
    s->dirty_par = true;
    return 0;
}

// options and a caller-written iterate (rare) go through their blocking setters; x0 / reference / parameters -- what the node
// rewrites every tick -- ride with the solve itself (brov_tick_host: one pinned staging buffer, asynchronous copies, one wait)
static int push_rare_inputs(brov_shim_state* s) {
    int rc = BROV_OK;
    if (s->dirty_opts) { rc = brov_set_opts(s->solver, &s->opts); s->dirty_opts = false; if (rc) return rc; }
    if (s->dirty_grid) {
        rc = brov_set_time_steps(s->solver, s->ts.empty() ? nullptr : s->ts.data());
        if (rc == BROV_OK) rc = brov_set_stage0_weight(s->solver, s->has_W0 ? s->W0 : nullptr);
        if (rc == BROV_OK) rc = brov_get_opts(s->solver, &s->opts);   // a uniform vector has become opts.Ts
        s->dirty_grid = false;
        if (rc) return rc;
    }
    if (s->dirty_iter) {
        rc = brov_set_iterate_host(s->solver, s->x.data(), s->u.data(), s->pi.data(), s->lam.data());
        s->dirty_iter = false;
        if (rc) return rc;
    }
    return rc;
}

int ocp_nlp_solve(ocp_nlp_solver* solver, ocp_nlp_in*, ocp_nlp_out* out) {  // what :948 calls
    brov_shim_state* s = solver ? solver->shim : nullptr;
    if (!s) return ACADOS_QP_FAILURE;
    const auto t0 = std::chrono::steady_clock::now();
    // A FEEDBACK call (rti_phase 2) without a preparation of the current iterate -- none yet, a second feedback on one preparation, or
    // options / grid / iterate rewritten in between: acados would solve the QP left in its memory, i.e. step on a linearisation of an
    // iterate that is gone.  The batched API refuses that (BROV_ERR_ARG); the drop-in takes the whole step (preparation + feedback in
    // one launch) on the CURRENT iterate instead -- the call does not fail and the step is never staler than acados' would be.
    // (Documented in INTEGRATION.md section 1; the reference's callers only use phase 0.)
    int phase = s->rti_phase;
    if (phase == 2 && (!s->prepared || s->dirty_opts || s->dirty_grid || s->dirty_iter)) phase = 0;
    int rc = push_rare_inputs(s);
    brov_result r{};
    if (rc == BROV_OK) {
        rc = brov_tick_host(s->solver, s->dirty_x0 ? s->x0.data() : nullptr, s->dirty_yref ? s->yref.data() : nullptr,
                            s->dirty_par ? s->par.data() : nullptr, phase, &r);
        if (rc == BROV_OK) s->dirty_x0 = s->dirty_yref = s->dirty_par = false;
    }
    s->prepared = rc == BROV_OK && phase == 1;
    if (rc != BROV_OK) {
        std::fprintf(stderr, "bluerov2_acados_solve: MI355X solver error %d: %s\n", rc, brov_last_error());
        return ACADOS_QP_FAILURE;
    }
    s->iter_host_valid = false;
    s->last_phase = phase;
    if (phase != 1) s->last = r;   // (a preparation moves nothing: the record of the last step stays what the getters answer from)
    s->times_valid = false;   // kernel times are read from the events when (if) the caller asks for them: not a wait of every tick
    s->time_tot = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (phase == 1) return ACADOS_SUCCESS;  // preparation only: nothing to report
    if (out) out->inf_norm_res = r.kkt;
    // status mapping (see create): QP at its iteration limit, step taken -> ACADOS_SUCCESS as upstream SQP_RTI; 1 / 3 / 4 as they are
    s->last_qp_status = r.status;
    s->last_status = (r.status == ACADOS_MAXITER && !s->strict_maxiter) ? ACADOS_SUCCESS : r.status;
    return s->last_status;
}

int ocp_nlp_precompute(ocp_nlp_solver*, ocp_nlp_in*, ocp_nlp_out*) { return ACADOS_SUCCESS; }

int bluerov2_acados_solve(bluerov2_solver_capsule* c) {  // :945-951
    if (!c || !c->shim) return ACADOS_QP_FAILURE;
    return ocp_nlp_solve(c->nlp_solver, c->nlp_in, c->nlp_out);
}

int bluerov2_acados_free(bluerov2_solver_capsule* c) {  // :954-998
    brov_shim_state* s = st_of(c);
    if (!s) return 0;
    brov_destroy(s->solver);
    delete c->nlp_solver_plan; delete c->nlp_config; delete c->nlp_dims; delete c->nlp_in; delete c->nlp_out;
    delete c->sens_out; delete c->nlp_solver;
    std::free(c->forw_vde_casadi); std::free(c->expl_ode_fun);
    delete s;
    std::memset(c, 0, sizeof(*c));
    return 0;
}

void bluerov2_acados_print_stats(bluerov2_solver_capsule* c) {  // :1001-1028 (RTI: one row)
    brov_shim_state* s = st_of(c);
    if (!s) return;
    std::printf("iter\tqp_stat\tqp_iter\n");
    std::printf("%d\t%d\t%d\n", 1, s->last_qp_status, s->last.qp_iter);
}

int bluerov2_acados_custom_update(bluerov2_solver_capsule*, double*, int) {  // :1030-1036
    std::printf("\ndummy function that can be called in between solver calls to update parameters or numerical data efficiently in C.\n");
    std::printf("nothing set yet..\n");
    return 1;
}

ocp_nlp_in* bluerov2_acados_get_nlp_in(bluerov2_solver_capsule* c) { return c->nlp_in; }
ocp_nlp_out* bluerov2_acados_get_nlp_out(bluerov2_solver_capsule* c) { return c->nlp_out; }
ocp_nlp_out* bluerov2_acados_get_sens_out(bluerov2_solver_capsule* c) { return c->sens_out; }
ocp_nlp_solver* bluerov2_acados_get_nlp_solver(bluerov2_solver_capsule* c) { return c->nlp_solver; }
ocp_nlp_config* bluerov2_acados_get_nlp_config(bluerov2_solver_capsule* c) { return c->nlp_config; }
void* bluerov2_acados_get_nlp_opts(bluerov2_solver_capsule* c) { return c->nlp_opts; }
ocp_nlp_dims* bluerov2_acados_get_nlp_dims(bluerov2_solver_capsule* c) { return c->nlp_dims; }
ocp_nlp_plan_t* bluerov2_acados_get_nlp_plan(bluerov2_solver_capsule* c) { return c->nlp_solver_plan; }

// ---- the ocp_nlp_* setters/getters the callers use --------------------------------------------------------------
int ocp_nlp_constraints_model_set(ocp_nlp_config*, ocp_nlp_dims*, ocp_nlp_in* in, int stage, const char* field, void* value) {
    brov_shim_state* s = in ? in->shim : nullptr;
    if (!s || !field || !value) return 1;
    const double* v = (const double*)value;
    if (!std::strcmp(field, "lbx") || !std::strcmp(field, "ubx")) {
        // stage 0 only (nbx = 0 elsewhere); all 12 components are equalities (idxbxe, :528-543): x0 := lbx = ubx
        if (stage != 0) return 1;
        std::memcpy(s->x0.data(), v, 12 * sizeof(double));
        s->dirty_x0 = true;
        return 0;
    }
    if (!std::strcmp(field, "lbu") || !std::strcmp(field, "ubu")) {  // :559-573 sets the same box on every stage
        double* dst = !std::strcmp(field, "lbu") ? s->opts.lbu : s->opts.ubu;
        std::memcpy(dst, v, 4 * sizeof(double));
        s->dirty_opts = true;
        return 0;
    }
    if (!std::strcmp(field, "idxbx") || !std::strcmp(field, "idxbu") || !std::strcmp(field, "idxbxe")) return 0;  // fixed structure
    std::fprintf(stderr, "ocp_nlp_constraints_model_set (MI355X shim): unsupported field '%s'\n", field);
    return 1;
}

int ocp_nlp_cost_model_set(ocp_nlp_config*, ocp_nlp_dims*, ocp_nlp_in* in, int stage, const char* field, void* value) {
    brov_shim_state* s = in ? in->shim : nullptr;
    if (!s || !field || !value || stage < 0 || stage > s->N) return 1;
    const double* v = (const double*)value;
    if (!std::strcmp(field, "yref") || !std::strcmp(field, "y_ref")) {
        const int ny = stage == s->N ? 12 : 16;  // NYN = 12: only the first 12 of the caller's 16-wide row are read
        std::memcpy(&s->yref[(size_t)stage * 16], v, ny * sizeof(double));
        s->dirty_yref = true;
        return 0;
    }
    if (!std::strcmp(field, "W") && stage == 0 && s->N > 1) {
        // the generated solver keeps a separate stage-0 weight W_0 (:422-441).  Stage 0 has its own here too, from create on; the
        // solver runs its single-weight kernels while the two are equal (brov_set_stage0_weight compares them on every push)
        for (int j = 0; j < 16; j++) s->W0[j] = v[j + 16 * j];
        s->has_W0 = true;
        s->dirty_grid = true;
        return 0;
    }
    if (!std::strcmp(field, "W")) {  // column-major ny x ny, diagonal (:422-481)
        const int ny = stage == s->N ? 12 : 16;
        double* dst = stage == s->N ? s->opts.We : s->opts.W;
        // the generated solver gives every stage 0..N-1 the same W (:422-481) and so does this solver: a stage weight that differs
        // from the one in force changes ALL stages -- said once, loudly, because acados itself would keep them apart
        bool differs = false;
        for (int j = 0; j < ny; j++) differs = differs || dst[j] != v[j + ny * j];
        static bool warned = false;
        if (differs && stage > 1 && stage < s->N && !warned) {
            std::fprintf(stderr, "ocp_nlp_cost_model_set (MI355X shim): \"W\" of stage %d differs from the shared stage weight; this "
                                 "solver keeps ONE weight for stages 1..N-1 (and W_0 for stage 0), it now applies to all of them\n", stage);
            warned = true;
        }
        for (int j = 0; j < ny; j++) dst[j] = v[j + ny * j];
        s->dirty_opts = true;
        if (stage < s->N) s->dirty_grid = true;   // W_0 is compared against the NEW shared weight when the options are pushed
        return 0;
    }
    if (!std::strcmp(field, "scaling")) {
        // :393 sets the stage cost scaling to the time step, and this solver has ONE number for both (brov_opts::Ts is the ERK4
        // step and the cost scaling).  Re-stating the current value is accepted; a scaling that differs from the integrator's step
        // is refused instead of silently changing the discretisation (use ocp_nlp_in_set "Ts" / update_time_steps for that).
        const double tsi = (stage < s->N && !s->ts.empty()) ? s->ts[stage] : s->opts.Ts;
        if (stage < s->N && std::fabs(v[0] - tsi) > 1e-12 * std::fabs(tsi)) {
            std::fprintf(stderr, "ocp_nlp_cost_model_set (MI355X shim): cost scaling %g of stage %d differs from its time step %g; the two "
                                 "are one parameter here (bluerov2_acados_update_time_steps sets both) -- refused\n", v[0], stage, tsi);
            return 1;
        }
        return 0;
    }
    std::fprintf(stderr, "ocp_nlp_cost_model_set (MI355X shim): unsupported field '%s'\n", field);
    return 1;
}

int ocp_nlp_in_set(ocp_nlp_config*, ocp_nlp_dims*, ocp_nlp_in* in, int stage, const char* field, void* value) {
    brov_shim_state* s = in ? in->shim : nullptr;
    if (!s || !field || !value) return 1;
    if (!std::strcmp(field, "Ts")) {   // one stage's time step (step and cost scaling together, as update_time_steps sets them)
        if (stage < 0 || stage >= s->N || !(*(const double*)value > 0.0)) return 1;
        if (s->ts.empty()) s->ts.assign((size_t)s->N, s->opts.Ts);
        s->ts[stage] = *(const double*)value;
        s->dirty_grid = true;
        return 0;
    }
    return 1;
}

void ocp_nlp_out_set(ocp_nlp_config*, ocp_nlp_dims*, ocp_nlp_out* out, int stage, const char* field, void* value) {
    brov_shim_state* s = out ? out->shim : nullptr;
    if (!s || !field || !value || stage < 0 || stage > s->N) return;
    pull_iterate(s);
    const double* v = (const double*)value;
    if (!std::strcmp(field, "x")) std::memcpy(&s->x[(size_t)stage * 12], v, 12 * sizeof(double));
    else if (!std::strcmp(field, "u")) { if (stage < s->N) std::memcpy(&s->u[(size_t)stage * 4], v, 4 * sizeof(double)); }
    else if (!std::strcmp(field, "pi")) { if (stage < s->N) std::memcpy(&s->pi[(size_t)stage * 12], v, 12 * sizeof(double)); }
    else if (!std::strcmp(field, "lam")) { if (stage < s->N) std::memcpy(&s->lam[(size_t)stage * 8], v, 8 * sizeof(double)); }
    else return;  // sl, su, t, z: this OCP has no slacks / algebraic variables (acados_solver_bluerov2.c:812-818)
    s->dirty_iter = true;
}

void ocp_nlp_out_get(ocp_nlp_config*, ocp_nlp_dims*, ocp_nlp_out* out, int stage, const char* field, void* value) {
    brov_shim_state* s = out ? out->shim : nullptr;
    if (!s || !field || !value) return;
    double* v = (double*)value;
    if (!std::strcmp(field, "kkt_norm_inf")) { *v = out->inf_norm_res; return; }
    if (stage < 0 || stage > s->N) return;
    // the node's per-tick read (bluerov2_dob.cpp:388: "u" of stage 0) is answered from the result record of the last solve: after a
    // successful step its u0 IS the new first input, after a failed one the held input (below) -- no copy of the iterate off the device
    if (stage == 0 && !std::strcmp(field, "u") && !s->iter_host_valid && s->last_phase != 1) { std::memcpy(v, s->last.u0, 4 * sizeof(double)); return; }
    pull_iterate(s);
    if (!std::strcmp(field, "x")) std::memcpy(v, &s->x[(size_t)stage * 12], 12 * sizeof(double));
    else if (!std::strcmp(field, "u")) {
        if (stage >= s->N) return;
        // The node publishes thrusts from this getter whatever the status was (bluerov2_dob.cpp:375-395).  After a failed step
        // (status 1 / 3 / 4) acados has not touched the iterate, so stage 0 still holds the last input it computed; here the iterate
        // of a failed instance may have diverged (KEEP) or been cold-started (RESTART: u = 0), and the input to apply is the one
        // the result record holds: the last successfully computed one, clamped into the box, NaN -> 0.
        const bool failed = s->last_status == ACADOS_NAN_DETECTED || s->last_status == ACADOS_MINSTEP || s->last_status == ACADOS_QP_FAILURE;
        if (stage == 0 && failed) std::memcpy(v, s->last.u0, 4 * sizeof(double));
        else std::memcpy(v, &s->u[(size_t)stage * 4], 4 * sizeof(double));
    }
    else if (!std::strcmp(field, "pi")) { if (stage < s->N) std::memcpy(v, &s->pi[(size_t)stage * 12], 12 * sizeof(double)); }
    else if (!std::strcmp(field, "lam")) { if (stage < s->N) std::memcpy(v, &s->lam[(size_t)stage * 8], 8 * sizeof(double)); }
}

void ocp_nlp_get(ocp_nlp_config*, ocp_nlp_solver* solver, const char* field, void* value) {
    brov_shim_state* s = solver ? solver->shim : nullptr;
    if (!s || !field || !value) return;
    if (!std::strcmp(field, "time_tot")) *(double*)value = s->time_tot;          // bluerov2_dob.cpp:386
    else if (!std::strcmp(field, "time_lin") || !std::strcmp(field, "time_qp_sol") || !std::strcmp(field, "time_qp")) {
        if (!s->timing_on) { s->timing_on = true; brov_enable_timing(s->solver, 1); }   // measured from the next solve on (0 until then)
        else if (!s->times_valid) {
            double k2[2] = {0, 0}, tot = 0;
            if (brov_last_solve_seconds(s->solver, &tot, k2) == BROV_OK) { s->time_lin = k2[0]; s->time_qp = k2[1]; }
            s->times_valid = true;
        }
        *(double*)value = field[5] == 'l' ? s->time_lin : s->time_qp;
    }
    else if (!std::strcmp(field, "sqp_iter")) *(int*)value = 1;                   // RTI: one iteration per call
    else if (!std::strcmp(field, "qp_iter")) *(int*)value = s->last.qp_iter;
    else if (!std::strcmp(field, "status")) *(int*)value = s->last_status;
    else if (!std::strcmp(field, "qp_status") || !std::strcmp(field, "qp_stat")) *(int*)value = s->last_qp_status;
    else if (!std::strcmp(field, "stat_n")) *(int*)value = 2;
    else if (!std::strcmp(field, "stat_m")) *(int*)value = 2;
    else if (!std::strcmp(field, "statistics")) {  // (stat_n+1) x nrow, column-major: [iter, qp_stat, qp_iter]
        double* st = (double*)value;
        st[0] = 0; st[1] = 1; st[2] = 0; st[3] = s->last_qp_status; st[4] = 0; st[5] = s->last.qp_iter;
    } else if (!std::strcmp(field, "cost_value")) *(double*)value = s->last.cost;
}

int ocp_nlp_solver_opts_set(ocp_nlp_config*, void* opts, const char* field, void* value) {
    brov_shim_state* s = (brov_shim_state*)opts;
    if (!s || !field || !value) return 1;
    if (!std::strcmp(field, "rti_phase")) {  // main_bluerov2.c:217
        const int ph = *(const int*)value;
        if (ph < 0 || ph > 2) return 1;
        s->rti_phase = ph;
        return 0;
    }
    if (!std::strcmp(field, "qp_iter_max")) { s->opts.qp_iter_max = *(const int*)value; s->dirty_opts = true; return 0; }
    if (!std::strcmp(field, "print_level") || !std::strcmp(field, "qp_warm_start") || !std::strcmp(field, "step_length") ||
        !std::strcmp(field, "levenberg_marquardt") || !std::strcmp(field, "globalization") || !std::strcmp(field, "qp_hpipm_mode"))
        return 0;  // fixed by construction (full step, GN Hessian, cold-started QP)
    return 1;
}

// ---- BLASFEO print helpers (main_bluerov2.c:229-231) --------------------------------------------------------------
void d_print_mat(int m, int n, double* A, int lda) {
    for (int i = 0; i < m; i++) { for (int j = 0; j < n; j++) std::printf("%9.5f ", A[i + lda * j]); std::printf("\n"); }
    std::printf("\n");
}
void d_print_exp_mat(int m, int n, double* A, int lda) {
    for (int i = 0; i < m; i++) { for (int j = 0; j < n; j++) std::printf("%e\t", A[i + lda * j]); std::printf("\n"); }
    std::printf("\n");
}
void d_print_tran_mat(int row, int col, double* A, int lda) {
    for (int j = 0; j < col; j++) { for (int i = 0; i < row; i++) std::printf("%9.5f ", A[i + lda * j]); std::printf("\n"); }
    std::printf("\n");
}
void d_print_exp_tran_mat(int row, int col, double* A, int lda) {
    for (int j = 0; j < col; j++) { for (int i = 0; i < row; i++) std::printf("%e\t", A[i + lda * j]); std::printf("\n"); }
    std::printf("\n");
}

}  // extern "C"
