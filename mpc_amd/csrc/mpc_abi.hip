// extern "C" boundary of libmpc_amd.so (declared in include/mpc_amd.h).
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "mpc_kernels.h"
#include "mpc_device.hpp"      // CL_STAT_W

struct mpc_handle {
    mpc_config cfg;
    mpc::DevCfg dev;
    bool lanes = false;          // one lane per scenario (mpc_config.solver_layout resolved)
    double *lane_ws = nullptr;   // lanes: per-stage workspace, also the warm-start record
    double *Hws = nullptr;       // B * N * N condensed Hessians (wave layout)
    int *sqp = nullptr;          // B SQP iteration counts of the last NL call
    double *ipm_ws = nullptr;    // B interior-point warm-start records (NULL: cold starts only)
    double *stat = nullptr;      // B * STAT_W solver records and closed-loop statistics
    int *iters = nullptr, *status = nullptr;   // used when the caller passes no output arrays
    int *work = nullptr, *order = nullptr;     // load balance of chained launches: last launch's iterations, scenario order
    int *queue = nullptr;                      // work queue counters of the fixed-model kernel (two ints, zero between launches)
    int order_B = 0;                           // batch size `order` is a permutation for (0: none yet)
    double *mp_scen = nullptr;   // optional B * 6
    double *cost_scen = nullptr; // optional B * 43 per-scenario cost weights
    double *term_C = nullptr;    // optional terminal set rows [term_m][5] (lane layout)
    int term_m = 0;
    bool px_free = false;        // nothing weighs the longitudinal position (controller weights, or every row of cost_scen)
    double *stats_out = nullptr; // device result of the statistics reduction (MPC_STATS_LEN) and its per-workgroup partial vectors
    bool cl_valid = false;       // the last call was a closed-loop call (the closed-loop part of `stat` belongs to it)
    double *sh_H = nullptr, *sh_Kref = nullptr, *sh_Kx = nullptr;   // fixed-model cache (mpc_condense_fixed)
    double *sqp_x = nullptr;     // emulate_acados: SQP iterates [B][6][N+1] (zero = a new solver object)
    double *sh_T = nullptr;      // table of the active-set solve (part of the cache; NULL with qp_method = interior point)
    double sh_T_resid = 0.0;     // largest entry of |H inv(H) - I| of that table (checked when it is built)
    bool sh_T_ok = false;        // the table is accurate enough for the active-set solve to certify its points with
    int *as2 = nullptr;          // B * 6: active-set counters of the last call (SolveArgs::as_out)
    double sh_vx = 0.0;
    int sh_mask = 0;
    bool sh_on = false, sh_exclusive = false;
    int last_B = 0;
    // staging of mpc_optimize_host: one block of pinned host memory that the GPU addresses directly (inputs, the
    // controller's carried memory, outputs), allocated by the first call
    double *host_stage = nullptr, *host_stage_dev = nullptr;
    std::string err;
};

static thread_local std::string g_create_err;

// Largest entry of |H inv(H) - I| the cached table of the active-set solve may show.  A point the method returns is exact
// on its active bounds and stationary up to the table's relative error times the distance from the unconstrained
// minimiser (at most twice the input bound): 1e-9 keeps that four orders below the 1e-5 the inputs are held to.
static const double ASET_TABLE_RESID_MAX = 1e-9;

static int fail(mpc_handle *h, int code, const std::string &msg)
{
    if (h) h->err = msg; else g_create_err = msg;
    return code;
}

static int hip_fail(mpc_handle *h, hipError_t e, const char *what)
{
    return fail(h, MPC_EHIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(h, call)                                        \
    do {                                                        \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return hip_fail(h, e_, #call);    \
    } while (0)

// Makes a handle's device current for the duration of an entry point and gives the caller's device back afterwards.
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    hipError_t err;
    explicit DeviceScope(int dev)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};

#define ON_HANDLE_DEVICE(h)                                                       \
    DeviceScope dev_scope_((h)->cfg.device);                                      \
    if (dev_scope_.err != hipSuccess) return hip_fail(h, dev_scope_.err, "selecting the handle's device")

extern "C" {

const char *mpc_last_error(const mpc_handle *h) { return h ? h->err.c_str() : g_create_err.c_str(); }

// Argument checks shared by mpc_create and the handle-free entry point; returns NULL when the definition is usable.
static const char *check_config(const mpc_config *cfg, bool with_workspace = true)
{
    static thread_local char buf[160];
    if (cfg->kind < MPC_KIND_L || cfg->kind > MPC_KIND_NL) return "unknown kind";
    if (!mpc::horizon_supported(cfg->N)) {
        snprintf(buf, sizeof buf, "horizon N=%d is outside 2..%d", cfg->N, MPC_MAX_HORIZON);
        return buf;
    }
    if ((with_workspace && cfg->B < 1) || !(cfg->dt > 0.0)) return "need B >= 1 and dt > 0";
    if (!(cfg->step_ramp_length >= 0.0)) return "step_ramp_length must be >= 0";
    if (cfg->kind == MPC_KIND_NL) {
        if (!mpc::nl_horizon_capacity(cfg->N)) return "the NL controller needs a horizon of 2..64";
        if (!(cfg->w_nl[7] > 0.0)) return "input weight w_nl[7] must be positive (strict convexity)";
    } else if (!(cfg->w[6] > 0.0)) return "input weight R*r must be positive (strict convexity)";
    if (cfg->solver_layout < MPC_LAYOUT_AUTO || cfg->solver_layout > MPC_LAYOUT_LANE) return "unknown solver_layout";
    if (cfg->solver_layout == MPC_LAYOUT_LANE && cfg->kind == MPC_KIND_NL) return "the NL controller has no lane-per-scenario kernel";
    if (cfg->qp_method < MPC_QP_AUTO || cfg->qp_method > MPC_QP_ACTIVE_SET) return "unknown qp_method";
    if (!(cfg->max_rate > 0.0) || !(cfg->max_steering > 0.0)) return "max_rate and max_steering must be positive";
    if (cfg->emulate_acados) {
        if (cfg->kind == MPC_KIND_NL) return "emulate_acados: the NL controller already runs the SQP iteration of its solver";
        if (!mpc::horizon_templated(cfg->N)) return "emulate_acados: horizons 10, 20, 30, 40, 50, 64 only";
        if (cfg->solver_layout == MPC_LAYOUT_LANE) return "emulate_acados: wave layout only";
        if (cfg->reg_epsilon < 0.0 || cfg->sqp_kkt_tol < 0.0) return "emulate_acados: reg_epsilon and sqp_kkt_tol must not be negative";
    }
    return nullptr;
}

static void fill_devcfg(const mpc_config *cfg, mpc::DevCfg &d)
{
    d.kind = cfg->kind; d.N = cfg->N; d.lpv_lin_prev = cfg->lpv_lin_prev; d.max_iter = cfg->max_iter > 0 ? cfg->max_iter : 80;
    d.dt = cfg->dt; d.eps = cfg->eps; d.max_steering = cfg->max_steering; d.max_rate = cfg->max_rate;
    d.tol_mu = cfg->tol_mu > 0 ? cfg->tol_mu : 1e-15; d.tol_res = cfg->tol_res > 0 ? cfg->tol_res : 1e-12;
    memcpy(d.mp, cfg->mp, sizeof d.mp); memcpy(d.w, cfg->w, sizeof d.w); memcpy(d.We, cfg->We, sizeof d.We);
    memcpy(d.w_nl, cfg->w_nl, sizeof d.w_nl); memcpy(d.we_nl, cfg->we_nl, sizeof d.we_nl);
    d.reg_eps = cfg->reg_epsilon > 0 ? cfg->reg_epsilon : 1e-4; d.kkt_tol = cfg->sqp_kkt_tol > 0 ? cfg->sqp_kkt_tol : 1e-4;
    d.eps_nl = cfg->eps_nl; d.sqp_tol = cfg->sqp_tol > 0 ? cfg->sqp_tol : 1e-9; d.sqp_max_iter = cfg->sqp_max_iter > 0 ? cfg->sqp_max_iter : 30;
}

// true when a cost definition (diag W [7], W_e 6x6 row-major) puts nothing on px: stage weight zero, terminal row and
// column zero
static bool weights_px_free(const double *w, const double *We)
{
    bool ok = w[0] == 0.0;
    for (int j = 0; j < 6 && ok; j++) ok = We[j] == 0.0 && We[6 * j] == 0.0;
    return ok;
}

int mpc_create(const mpc_config *cfg, mpc_handle **out)
{
    if (!cfg || !out) return fail(nullptr, MPC_EINVAL, "mpc_create: null argument");
    *out = nullptr;
    if (const char *why = check_config(cfg)) return fail(nullptr, MPC_EINVAL, std::string("mpc_create: ") + why);
    mpc_handle *h = new (std::nothrow) mpc_handle();
    if (!h) return fail(nullptr, MPC_ENOMEM, "mpc_create: out of host memory");
    h->cfg = *cfg;
    h->px_free = weights_px_free(cfg->w, cfg->We);
    // AUTO: the lane layout where it is the faster one (measured, closed-loop skidpad runs): horizons without a
    // register-resident wave kernel from 2048 scenarios up (L at N = 80: 1.5 against 0.8 M scenario-steps/s at 4096
    // scenarios, 2.7 against 0.8 M at 8192; below 2048 scenarios a wavefront per scenario has the shorter step).  At the
    // instantiated horizons the wave kernels win at every batch size (LPV at N = 50: 6.9 against 4.4 M at 8192).
    h->lanes = cfg->solver_layout == MPC_LAYOUT_LANE ||
               (cfg->solver_layout == MPC_LAYOUT_AUTO && !cfg->emulate_acados && (cfg->kind == MPC_KIND_LPV || cfg->kind == MPC_KIND_L) &&
                !mpc::horizon_templated(cfg->N) && cfg->B >= 2048);
    fill_devcfg(cfg, h->dev);
    DeviceScope scope(cfg->device);
    hipError_t e = scope.err;
    // condensed Hessians: N x N per scenario; the NL kernels lay theirs out for the instantiation that runs the horizon
    const size_t B = (size_t)cfg->B, N = (size_t)(cfg->kind == MPC_KIND_NL ? mpc::nl_horizon_capacity(cfg->N) : cfg->N);
    if (e == hipSuccess && !h->lanes) e = hipMalloc(&h->Hws, B * N * N * sizeof(double));
    if (e == hipSuccess && h->lanes) e = hipMalloc(&h->lane_ws, mpc::lanes_workspace_doubles(cfg->N, cfg->B, 0) * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&h->stat, B * mpc::STAT_W * sizeof(double));
    if (e == hipSuccess) e = hipMemset(h->stat, 0, B * mpc::STAT_W * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&h->iters, B * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&h->status, B * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&h->work, B * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&h->order, B * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&h->queue, 2 * sizeof(int));
    if (e == hipSuccess) e = hipMemset(h->queue, 0, 2 * sizeof(int));
    if (e == hipSuccess && !cfg->ipm_cold_start) {
        e = hipMalloc(&h->ipm_ws, B * mpc::IPM_WARM_DOUBLES * sizeof(double));
        if (e == hipSuccess) e = hipMemset(h->ipm_ws, 0, B * mpc::IPM_WARM_DOUBLES * sizeof(double));
    }
    if (e == hipSuccess && cfg->emulate_acados) {
        e = hipMalloc(&h->sqp_x, B * 6 * ((size_t)cfg->N + 1) * sizeof(double));
        if (e == hipSuccess) e = hipMemset(h->sqp_x, 0, B * 6 * ((size_t)cfg->N + 1) * sizeof(double));
    }
    if (e == hipSuccess) e = hipMalloc(&h->sqp, B * sizeof(int));
    if (e == hipSuccess) e = hipMemset(h->sqp, 0, B * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(&h->stats_out, (size_t)(1 + mpc::STATS_BLOCKS) * MPC_STATS_LEN * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&h->as2, B * 6 * sizeof(int));
    if (e == hipSuccess) e = hipMemset(h->as2, 0, B * 6 * sizeof(int));
    if (e != hipSuccess) {
        int rc = hip_fail(nullptr, e, "mpc_create: workspace allocation");
        mpc_destroy(h);
        return rc;
    }
    *out = h;
    return MPC_OK;
}

void mpc_destroy(mpc_handle *h)
{
    if (!h) return;
    DeviceScope scope(h->cfg.device);
    (void)hipFree(h->Hws); (void)hipFree(h->lane_ws); (void)hipFree(h->term_C); (void)hipFree(h->stat); (void)hipFree(h->iters); (void)hipFree(h->status);
    (void)hipFree(h->stats_out); (void)hipFree(h->mp_scen); (void)hipFree(h->cost_scen); (void)hipFree(h->sqp); (void)hipFree(h->ipm_ws);
    (void)hipFree(h->sh_H); (void)hipFree(h->sh_Kref); (void)hipFree(h->sh_Kx); (void)hipFree(h->sh_T); (void)hipFree(h->as2);
    (void)hipFree(h->work); (void)hipFree(h->order); (void)hipFree(h->queue); (void)hipFree(h->sqp_x);
    if (h->host_stage) (void)hipHostFree(h->host_stage);
    delete h;
}

int mpc_set_model_params(mpc_handle *h, const double *mp_per_scenario)
{
    if (!h) return MPC_EINVAL;
    ON_HANDLE_DEVICE(h);
    if (!mp_per_scenario) { (void)hipFree(h->mp_scen); h->mp_scen = nullptr; return MPC_OK; }
    const size_t bytes = (size_t)h->cfg.B * 6 * sizeof(double);
    if (!h->mp_scen) HIP_TRY(h, hipMalloc(&h->mp_scen, bytes));
    HIP_TRY(h, hipMemcpy(h->mp_scen, mp_per_scenario, bytes, hipMemcpyDeviceToDevice));
    return MPC_OK;
}

int mpc_set_cost_weights(mpc_handle *h, const double *cost_per_scenario)
{
    if (!h) return MPC_EINVAL;
    ON_HANDLE_DEVICE(h);
    if (h->cfg.kind == MPC_KIND_NL) return fail(h, MPC_EINVAL, "mpc_set_cost_weights: the NL controller's weights are fixed by its definition");
    if (!cost_per_scenario) {
        (void)hipFree(h->cost_scen); h->cost_scen = nullptr;
        h->px_free = weights_px_free(h->cfg.w, h->cfg.We);
        return MPC_OK;
    }
    const size_t bytes = (size_t)h->cfg.B * 43 * sizeof(double);
    if (!h->cost_scen) HIP_TRY(h, hipMalloc(&h->cost_scen, bytes));
    HIP_TRY(h, hipMemcpy(h->cost_scen, cost_per_scenario, bytes, hipMemcpyDeviceToDevice));
    if (h->lanes) {      // which build of the lane solver the table allows (a set-up call: one copy to the host)
        std::vector<double> host((size_t)h->cfg.B * 43);
        HIP_TRY(h, hipMemcpy(host.data(), h->cost_scen, bytes, hipMemcpyDeviceToHost));
        bool ok = true;
        for (int b = 0; b < h->cfg.B && ok; b++) ok = weights_px_free(&host[(size_t)b * 43], &host[(size_t)b * 43 + 7]);
        h->px_free = ok;
    }
    return MPC_OK;
}

int mpc_set_terminal_set(mpc_handle *h, int m, const double *C_host)
{
    if (!h) return MPC_EINVAL;
    ON_HANDLE_DEVICE(h);
    if (m < 0 || m > MPC_MAX_TERMINAL_ROWS || (m > 0 && !C_host)) return fail(h, MPC_EINVAL, "mpc_set_terminal_set: 0 <= m <= MPC_MAX_TERMINAL_ROWS rows of 5 doubles");
    if (m > 0 && h->cfg.kind == MPC_KIND_NL) return fail(h, MPC_EINVAL, "mpc_set_terminal_set: linear controllers only");
    if (m > 0 && !h->lanes) {
        // General terminal rows are solved stage-wise.  A handle whose layout was left to the library moves to the
        // lane-per-scenario kernel here (it stays there when the set is removed again); one created with
        // MPC_LAYOUT_WAVE keeps what its caller asked for and refuses the rows.
        if (h->cfg.solver_layout != MPC_LAYOUT_AUTO || h->sqp_x)
            return fail(h, MPC_EINVAL, "mpc_set_terminal_set: general terminal rows are solved stage-wise; create the controller with solver_layout = MPC_LAYOUT_AUTO or MPC_LAYOUT_LANE");
        HIP_TRY(h, hipDeviceSynchronize());
        double *ws = nullptr;
        HIP_TRY(h, hipMalloc(&ws, mpc::lanes_workspace_doubles(h->cfg.N, h->cfg.B, 0) * sizeof(double)));
        h->lane_ws = ws; h->lanes = true; h->sh_on = false; h->order_B = 0;
        if (h->ipm_ws) HIP_TRY(h, hipMemset(h->ipm_ws, 0, (size_t)h->cfg.B * mpc::IPM_WARM_DOUBLES * sizeof(double)));
    }
    if (!h->lanes) return MPC_OK;                          // m == 0 on a wave-layout handle: nothing to remove
    HIP_TRY(h, hipDeviceSynchronize());                    // the workspace changes shape: no launch of this handle may be in flight
    // rows, then a workspace with room for their slacks and multipliers; on failure the handle keeps what it had
    double *C_new = nullptr, *ws_new = nullptr;
    if (m > 0) {
        hipError_t e = hipMalloc(&C_new, (size_t)m * 5 * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy(C_new, C_host, (size_t)m * 5 * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) { (void)hipFree(C_new); return hip_fail(h, e, "mpc_set_terminal_set: rows"); }
    }
    if (m != h->term_m) {
        hipError_t e = hipMalloc(&ws_new, mpc::lanes_workspace_doubles(h->cfg.N, h->cfg.B, m) * sizeof(double));
        if (e != hipSuccess) { (void)hipFree(C_new); return hip_fail(h, e, "mpc_set_terminal_set: workspace"); }
        (void)hipFree(h->lane_ws);
        h->lane_ws = ws_new;
    }
    (void)hipFree(h->term_C);
    h->term_C = C_new; h->term_m = m;
    // the records of earlier solves belong to another problem (and another workspace shape)
    if (h->ipm_ws) HIP_TRY(h, hipMemset(h->ipm_ws, 0, (size_t)h->cfg.B * mpc::IPM_WARM_DOUBLES * sizeof(double)));
    return MPC_OK;
}

int mpc_condense_fixed(mpc_handle *h, void *stream, double vx, int exclusive)
{
    if (!h) return MPC_EINVAL;
    ON_HANDLE_DEVICE(h);
    if (h->cfg.kind == MPC_KIND_LPV || h->cfg.kind == MPC_KIND_NL) return fail(h, MPC_EINVAL, "mpc_condense_fixed: the LPV and NL models are re-linearised every step");
    if (!(vx == vx)) { h->sh_on = false; return MPC_OK; }   // NaN switches the cache off
    if (h->lanes) { h->sh_on = false; return MPC_OK; }      // the lane-per-scenario kernel never condenses: nothing to cache
    // the cache belongs to the register-resident kernels; other horizons keep condensing per scenario (same results)
    if (!mpc::horizon_templated(h->cfg.N)) { h->sh_on = false; return MPC_OK; }
    const size_t N = (size_t)h->cfg.N;
    const bool want_T = h->cfg.qp_method != MPC_QP_INTERIOR_POINT;
    if (!h->sh_H) {      // all buffers or none: a partial set would hand the kernels a null pointer on the next call
        hipError_t e = hipMalloc(&h->sh_H, N * N * sizeof(double));
        if (e == hipSuccess) e = hipMalloc(&h->sh_Kref, 3 * N * N * sizeof(double));
        if (e == hipSuccess) e = hipMalloc(&h->sh_Kx, 5 * N * sizeof(double));
        if (e == hipSuccess && want_T) e = hipMalloc(&h->sh_T, 4 * N * N * sizeof(double));
        if (e != hipSuccess) {
            (void)hipFree(h->sh_H); (void)hipFree(h->sh_Kref); (void)hipFree(h->sh_Kx); (void)hipFree(h->sh_T);
            h->sh_H = h->sh_Kref = h->sh_Kx = h->sh_T = nullptr; h->sh_on = false;
            return hip_fail(h, e, "mpc_condense_fixed: cache allocation");
        }
    }
    // The active-set solve refines and certifies its points through the table alone; the table is therefore checked
    // against H once, here (a set-up call: it waits for the kernel).  Beyond ASET_TABLE_RESID_MAX the method stays off
    // for this cache and every constrained solve goes to the interior point, which works from H itself.
    HIP_TRY(h, mpc::launch_condense_fixed(h->dev, vx, h->sh_H, h->sh_Kref, h->sh_Kx, h->sh_T, h->stats_out, (hipStream_t)stream));
    h->sh_T_resid = 0.0; h->sh_T_ok = false;
    if (h->sh_T) {
        HIP_TRY(h, hipMemcpyAsync(&h->sh_T_resid, h->stats_out, sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream));
        HIP_TRY(h, hipStreamSynchronize((hipStream_t)stream));
        h->sh_T_ok = h->sh_T_resid <= ASET_TABLE_RESID_MAX;      // (a NaN fails the test)
    }
    h->sh_vx = vx;
    h->sh_mask = 0;
    for (int c = 0; c < 3; c++) {
        bool used = h->cfg.w[c] != 0.0;
        for (int d = 0; d < 6; d++) used = used || h->cfg.We[d * 6 + c] != 0.0 || h->cfg.We[c * 6 + d] != 0.0;
        if (used) h->sh_mask |= 1 << c;
    }
    h->sh_on = true;
    h->sh_exclusive = exclusive != 0;
    return MPC_OK;
}

static void attach_cache(const mpc_handle *h, mpc::SolveArgs &a)
{
    a.sqp_x = h->sqp_x;                  // emulate_acados (NULL otherwise): the regularised problem is condensed per scenario
    if (h->sqp_x) a.sqp_out = h->sqp;
    if (h->sh_on && !h->mp_scen && !h->cost_scen && !h->sqp_x) {
        a.sh_H = h->sh_H; a.sh_Kref = h->sh_Kref; a.sh_Kx = h->sh_Kx; a.sh_vx = h->sh_vx; a.sh_mask = h->sh_mask; a.sh_exclusive = h->sh_exclusive ? 1 : 0;
        a.sh_T = h->sh_T_ok ? h->sh_T : nullptr;
    }
    a.as_out = h->as2;
    a.general_aset = h->cfg.qp_method != MPC_QP_INTERIOR_POINT ? 1 : 0;
}

static hipError_t launch_for_kind(const mpc_handle *h, mpc::SolveArgs &a, hipStream_t st)
{
    if (h->cfg.kind == MPC_KIND_NL) { a.sqp_out = h->sqp; return mpc::launch_mpc_nl(h->dev, a, st); }
    if (h->lanes) return mpc::launch_mpc_lanes(h->dev, a, h->lane_ws, (size_t)h->cfg.B, st);
    return mpc::launch_mpc(h->dev, a, st);
}

// Load balance of the closed-loop launches.  A launch ends with its slowest scenario, and scenarios differ widely in
// how often their QP hits a bound (each such solve is a long chain of dependent work for one wave).  Every launch of
// at least BAL_MIN_STEPS steps records the solver iterations per scenario, and the next launch maps its waves to the
// scenarios in descending order of that work: the long chains start first and the short ones fill in behind them
// (measured on 4096 skidpad scenarios: 1000 steps in 28 ms instead of 39 ms).  A run longer than BAL_CHUNK steps goes
// out as a chain of launches so that the order can follow the work.  Per scenario nothing changes: its state lives in
// the caller's arrays between launches exactly as between two calls, and which wave advances it has no influence on
// its arithmetic.
static const int BAL_CHUNK = 1024, BAL_MIN_B = 1024, BAL_MIN_STEPS = 4;

// true when the launch goes to the work-queue form of the fixed-model kernel (the only reader of SolveArgs::queue)
static bool launches_work_queue(const mpc_handle *h, const mpc::SolveArgs &a)
{
    return h->cfg.kind != MPC_KIND_NL && !h->lanes && a.sh_H && a.sh_exclusive && mpc::horizon_templated(h->cfg.N);
}

static hipError_t launch_chain(mpc_handle *h, const mpc::SolveArgs &a, hipStream_t st)
{
    mpc::SolveArgs c = a;
    const int CHUNK = BAL_CHUNK;
    if (a.B < BAL_MIN_B || h->lanes) return launch_for_kind(h, c, st);   // lanes advance in lock step: no order to balance
    int shift = 0;
    const int span = a.nsteps < CHUNK ? a.nsteps : CHUNK;
    while (((span * 24) >> shift) > 1023) shift++;             // up to ~24 iterations per step spread over the 1024 bins
    for (int t0 = 0; t0 < a.nsteps; t0 += CHUNK) {
        c = a;
        c.nsteps = a.nsteps - t0 < CHUNK ? a.nsteps - t0 : CHUNK;
        const size_t off = (size_t)t0 * (size_t)a.B;
        if (a.u_hist) c.u_hist = a.u_hist + off;
        if (a.x_hist) c.x_hist = a.x_hist + off * a.nx_plant;
        if (a.est_hist) c.est_hist = a.est_hist + off * 10;
        if (a.meas_noise) c.meas_noise = a.meas_noise + off * 7;
        c.accumulate = t0 > 0;
        c.order = h->order_B == a.B ? h->order : nullptr;
        const bool rank = c.nsteps >= BAL_MIN_STEPS;            // a short launch says little and the sort is not free
        c.work_out = rank ? h->work : nullptr;
        c.queue = c.nsteps >= 2 ? h->queue : nullptr;           // one step per launch: every wave has the same short job
        hipError_t e = hipSuccess;
        // The last wave to leave a work-queue launch puts the counters back to zero; a launch that was cut short (a fault,
        // a reset) would leave them set and the next launch's waves would find nothing to do.  Start every such launch
        // from zero whatever came before.
        if (c.queue && launches_work_queue(h, c)) e = hipMemsetAsync(h->queue, 0, 2 * sizeof(int), st);
        if (e != hipSuccess) return e;
        e = launch_for_kind(h, c, st);
        if (e != hipSuccess) return e;
        if (rank) {
            e = mpc::launch_balance_order(st, a.B, h->work, shift, h->order);
            if (e != hipSuccess) return e;
            h->order_B = a.B;
        }
    }
    return hipSuccess;
}

static int check_batch(mpc_handle *h, int B, const char *who)
{
    if (!h) return MPC_EINVAL;
    if (B < 1 || B > h->cfg.B) return fail(h, MPC_EINVAL, std::string(who) + ": B outside 1..cfg.B");
    return MPC_OK;
}

int mpc_solve_only(mpc_handle *h, void *stream, int B, const double *x0, const double *wp, const double *p,
                   const double *d_hat, double *lin_x, double *lin_u, double *U_out, double *traj_out,
                   int *status_out, int *iters_out)
{
    if (int rc = check_batch(h, B, "mpc_solve_only")) return rc;
    ON_HANDLE_DEVICE(h);
    if (!x0 || !wp || !p || !U_out) return fail(h, MPC_EINVAL, "mpc_solve_only: x0, wp, p and U_out are required");
    if (h->cfg.kind == MPC_KIND_LPV && (!lin_x || !lin_u)) return fail(h, MPC_EINVAL, "mpc_solve_only: LPV needs lin_x and lin_u");
    if (h->cfg.kind == MPC_KIND_NL && !lin_u) return fail(h, MPC_EINVAL, "mpc_solve_only: NL needs lin_u (the SQP iterate)");
    mpc::SolveArgs a = {};
    a.mode = mpc::MODE_SOLVE; a.B = B; a.nsteps = 1; a.nx_plant = 8; a.x0 = x0; a.wp = wp; a.p = p; a.d_hat = d_hat; a.lin_x = lin_x; a.lin_u = lin_u;
    a.mp_scen = h->mp_scen; a.cost_scen = h->cost_scen; a.Hws = h->Hws; a.ipm_ws = h->ipm_ws; a.U_out = U_out; a.traj_out = traj_out;
    a.status_out = status_out; a.iters_out = iters_out; a.status_own = h->status; a.iters_own = h->iters; a.term_m = h->term_m; a.term_C = h->term_C; a.px_free = h->px_free;
    a.stat_out = h->stat;
    attach_cache(h, a);
    h->last_B = B; h->cl_valid = false;
    HIP_TRY(h, launch_for_kind(h, a, (hipStream_t)stream));
    return MPC_OK;
}

// optimize() with host arrays in and out (the reference's own calling convention: numpy in, numpy out, blocking).
// The arguments are packed into a block of pinned host memory that the kernels read and write in place over the bus (a
// few KB per scenario): one kernel launch and one stream synchronisation per call, no copy commands.
static size_t host_stage_doubles(int N, int B)
{
    const size_t n1 = (size_t)N + 1, b = (size_t)B;
    return b * (7 + 4 * n1 + n1 + 1 + 6 * n1 + (size_t)N + (size_t)N + 7 * n1 + 1);       // the last 1: status and iterations (two ints)
}

int mpc_optimize_host(mpc_handle *h, void *stream, int B, const double *x0, const double *wp, const double *p,
                      const double *d_hat, double *lin_x, double *lin_u, double *U_out, double *traj_out, int *status_out,
                      int *iters_out)
{
    if (!h) return MPC_EINVAL;
    const int cap = h->cfg.B < MPC_HOST_MAX_B ? h->cfg.B : MPC_HOST_MAX_B;
    if (B < 1 || B > cap) return fail(h, MPC_EINVAL, "mpc_optimize_host: B outside 1..min(cfg.B, MPC_HOST_MAX_B)");
    ON_HANDLE_DEVICE(h);
    if (!x0 || !wp || !p || !U_out) return fail(h, MPC_EINVAL, "mpc_optimize_host: x0, wp, p and U_out are required");
    const int kind = h->cfg.kind, N = h->cfg.N;
    if (kind == MPC_KIND_LPV && (!lin_x || !lin_u)) return fail(h, MPC_EINVAL, "mpc_optimize_host: LPV needs lin_x and lin_u");
    if (kind == MPC_KIND_NL && !lin_u) return fail(h, MPC_EINVAL, "mpc_optimize_host: NL needs lin_u (the SQP iterate)");
    if (!h->host_stage) {
        HIP_TRY(h, hipHostMalloc((void **)&h->host_stage, host_stage_doubles(N, cap) * sizeof(double), hipHostMallocMapped));
        hipError_t e = hipHostGetDevicePointer((void **)&h->host_stage_dev, h->host_stage, 0);
        if (e != hipSuccess) { (void)hipHostFree(h->host_stage); h->host_stage = nullptr; return hip_fail(h, e, "mpc_optimize_host: hipHostGetDevicePointer"); }
    }
    const size_t n1 = (size_t)N + 1, b = (size_t)B;
    // segments, in doubles from the start of the block (packed for this call's B)
    const size_t o_x0 = 0, o_wp = o_x0 + 7 * b, o_p = o_wp + 4 * n1 * b, o_dh = o_p + n1 * b, o_lx = o_dh + b, o_lu = o_lx + 6 * n1 * b,
                 o_U = o_lu + (size_t)N * b, o_tr = o_U + (size_t)N * b, o_int = o_tr + 7 * n1 * b;
    double *hs = h->host_stage, *ds = h->host_stage_dev;
    memcpy(hs + o_x0, x0, 7 * b * sizeof(double));
    memcpy(hs + o_wp, wp, 4 * n1 * b * sizeof(double));
    memcpy(hs + o_p, p, n1 * b * sizeof(double));
    if (d_hat) memcpy(hs + o_dh, d_hat, b * sizeof(double));
    if (lin_x) memcpy(hs + o_lx, lin_x, 6 * n1 * b * sizeof(double));
    if (lin_u) memcpy(hs + o_lu, lin_u, (size_t)N * b * sizeof(double));
    mpc::SolveArgs a = {};
    a.mode = mpc::MODE_SOLVE; a.B = B; a.nsteps = 1; a.nx_plant = 8;
    a.x0 = ds + o_x0; a.wp = ds + o_wp; a.p = ds + o_p; a.d_hat = d_hat ? ds + o_dh : nullptr;
    a.lin_x = lin_x ? ds + o_lx : nullptr; a.lin_u = lin_u ? ds + o_lu : nullptr;
    a.mp_scen = h->mp_scen; a.cost_scen = h->cost_scen; a.Hws = h->Hws; a.ipm_ws = h->ipm_ws;
    a.U_out = ds + o_U; a.traj_out = traj_out ? ds + o_tr : nullptr;
    a.status_out = reinterpret_cast<int *>(ds + o_int); a.iters_out = a.status_out + B;
    a.status_own = h->status; a.iters_own = h->iters; a.term_m = h->term_m; a.term_C = h->term_C; a.px_free = h->px_free;
    a.stat_out = h->stat;
    attach_cache(h, a);
    h->last_B = B; h->cl_valid = false;
    HIP_TRY(h, launch_for_kind(h, a, (hipStream_t)stream));
    HIP_TRY(h, hipStreamSynchronize((hipStream_t)stream));
    memcpy(U_out, hs + o_U, (size_t)N * b * sizeof(double));
    if (traj_out) memcpy(traj_out, hs + o_tr, 7 * n1 * b * sizeof(double));
    if (lin_x) memcpy(lin_x, hs + o_lx, 6 * n1 * b * sizeof(double));
    if (lin_u) memcpy(lin_u, hs + o_lu, (size_t)N * b * sizeof(double));
    const int *ints = reinterpret_cast<const int *>(hs + o_int);
    if (status_out) memcpy(status_out, ints, b * sizeof(int));
    if (iters_out) memcpy(iters_out, ints + B, b * sizeof(int));
    return MPC_OK;
}

int mpc_set_lpv_lin_mode(mpc_handle *h, int lin_prev)
{
    if (!h) return MPC_EINVAL;
    if (h->cfg.kind != MPC_KIND_LPV) return fail(h, MPC_EINVAL, "mpc_set_lpv_lin_mode: handle is not an LPV controller");
    h->cfg.lpv_lin_prev = lin_prev != 0; h->dev.lpv_lin_prev = h->cfg.lpv_lin_prev;
    return MPC_OK;
}

int mpc_closed_loop_run(mpc_handle *h, void *stream, int B, int planner, int nsteps, double *state, int *lap,
                        double *prev_progress, const double *target_vel, double *lin_x, double *lin_u,
                        double *u0_out, double *wp_out, double *traj_out, double *U_out, double *u_hist,
                        double *x_hist, int *status_out, int *iters_out);

int mpc_closed_loop_step(mpc_handle *h, void *stream, int B, int planner, double *state, int *lap,
                         double *prev_progress, const double *target_vel, double *lin_x, double *lin_u,
                         double *u0_out, double *wp_out, double *traj_out, double *U_out, int *status_out,
                         int *iters_out)
{
    return mpc_closed_loop_run(h, stream, B, planner, 1, state, lap, prev_progress, target_vel, lin_x, lin_u, u0_out,
                               wp_out, traj_out, U_out, nullptr, nullptr, status_out, iters_out);
}

int mpc_closed_loop_run(mpc_handle *h, void *stream, int B, int planner, int nsteps, double *state, int *lap,
                        double *prev_progress, const double *target_vel, double *lin_x, double *lin_u,
                        double *u0_out, double *wp_out, double *traj_out, double *U_out, double *u_hist,
                        double *x_hist, int *status_out, int *iters_out)
{
    if (int rc = check_batch(h, B, "mpc_closed_loop_step")) return rc;
    ON_HANDLE_DEVICE(h);
    if (nsteps < 1) return fail(h, MPC_EINVAL, "mpc_closed_loop_run: nsteps must be >= 1");
    if (planner != MPC_PLANNER_SKIDPAD && planner != MPC_PLANNER_STEP) return fail(h, MPC_EINVAL, "mpc_closed_loop_step: unknown planner");
    if (!state || !target_vel || !u0_out) return fail(h, MPC_EINVAL, "mpc_closed_loop_step: state, target_vel and u0_out are required");
    if (planner == MPC_PLANNER_SKIDPAD && (!lap || !prev_progress)) return fail(h, MPC_EINVAL, "mpc_closed_loop_step: skidpad planner needs lap and prev_progress");
    if (h->cfg.kind == MPC_KIND_LPV && (!lin_x || !lin_u)) return fail(h, MPC_EINVAL, "mpc_closed_loop_step: LPV needs lin_x and lin_u");
    if (h->cfg.kind == MPC_KIND_NL && !lin_u) return fail(h, MPC_EINVAL, "mpc_closed_loop_step: NL needs lin_u (the SQP iterate)");
    if (h->cfg.kind == MPC_KIND_OFL) return fail(h, MPC_EINVAL, "mpc_closed_loop_step: the OFL loop runs on the estimate; use mpc_solve_only with the EKF entry points");
    mpc::SolveArgs a = {};
    a.mode = planner == MPC_PLANNER_SKIDPAD ? mpc::MODE_SKIDPAD : mpc::MODE_STEP; a.B = B;
    a.nsteps = nsteps; a.u_hist = u_hist; a.x_hist = x_hist; a.nx_plant = 8;
    a.state = state; a.lap = lap; a.prev_progress = prev_progress; a.target_vel = target_vel; a.step_ramp = h->cfg.step_ramp_length;
    a.lin_x = lin_x; a.lin_u = lin_u; a.mp_scen = h->mp_scen; a.cost_scen = h->cost_scen; a.Hws = h->Hws; a.ipm_ws = h->ipm_ws;
    a.U_out = U_out; a.traj_out = traj_out; a.u0_out = u0_out; a.wp_out = wp_out;
    a.status_out = status_out; a.iters_out = iters_out; a.status_own = h->status; a.iters_own = h->iters; a.term_m = h->term_m; a.term_C = h->term_C; a.px_free = h->px_free;
    a.stat_out = h->stat;
    attach_cache(h, a);
    h->last_B = B; h->cl_valid = true;
    HIP_TRY(h, launch_chain(h, a, (hipStream_t)stream));
    return MPC_OK;
}

int mpc_ofl_closed_loop_run(mpc_handle *h, void *stream, int B, int nsteps, double *state10, double *ekf_x, double *ekf_P,
                            const double *target_vel, const double *q_diag_host, const double *r_diag_host,
                            const double *meas_noise, double *u0_out, double *u_hist, double *x_hist, double *est_hist,
                            int *status_out, int *iters_out)
{
    if (int rc = check_batch(h, B, "mpc_ofl_closed_loop_run")) return rc;
    ON_HANDLE_DEVICE(h);
    if (h->cfg.kind != MPC_KIND_OFL) return fail(h, MPC_EINVAL, "mpc_ofl_closed_loop_run: handle is not an OFL controller");
    if (h->lanes) return fail(h, MPC_EINVAL, "mpc_ofl_closed_loop_run: not available in the lane-per-scenario layout");
    if (nsteps < 1 || !state10 || !ekf_x || !ekf_P || !target_vel || !q_diag_host || !r_diag_host || !u0_out)
        return fail(h, MPC_EINVAL, "mpc_ofl_closed_loop_run: missing argument");
    mpc::SolveArgs a = {};
    a.mode = mpc::MODE_OFL; a.B = B; a.nsteps = nsteps; a.nx_plant = 10;
    a.state = state10; a.target_vel = target_vel; a.step_ramp = h->cfg.step_ramp_length; a.ekf_x = ekf_x; a.ekf_P = ekf_P; a.meas_noise = meas_noise;
    for (int i = 0; i < 10; i++) a.ekf_q[i] = q_diag_host[i];
    for (int i = 0; i < 7; i++) a.ekf_r[i] = r_diag_host[i];
    a.mp_scen = h->mp_scen; a.cost_scen = h->cost_scen; a.Hws = h->Hws; a.ipm_ws = h->ipm_ws; a.u0_out = u0_out; a.u_hist = u_hist; a.x_hist = x_hist; a.est_hist = est_hist;
    a.status_out = status_out; a.iters_out = iters_out; a.status_own = h->status; a.iters_own = h->iters; a.term_m = h->term_m; a.term_C = h->term_C; a.px_free = h->px_free;
    a.stat_out = h->stat;
    attach_cache(h, a);
    // The active-set solve runs here too (MPC_QP_AUTO): on noisy estimates the active sets are large and jump from step to
    // step, a third of the attempts are handed on and the gain is small (8192 scenarios x 300 steps with the EKF's
    // measurement noise: 8.3 M scenario-steps/s against 7.9 M with the interior point alone, with the kernel's smaller
    // iteration cap for this loop); without noise the sets move one stage per step and it is 2.3 times faster
    // (256 scenarios x 80 steps: 9.1 ms against 20.5 ms).
    h->last_B = B; h->cl_valid = true;
    HIP_TRY(h, launch_chain(h, a, (hipStream_t)stream));
    return MPC_OK;
}

int mpc_reset_warm_start(mpc_handle *h, void *stream)
{
    if (!h) return MPC_EINVAL;
    ON_HANDLE_DEVICE(h);
    // the validity flags of the per-scenario records: one per record (wave layout, at its end) or B packed flags (lane
    // layout); clearing the whole block covers both
    if (h->ipm_ws) HIP_TRY(h, hipMemsetAsync(h->ipm_ws, 0, (size_t)h->cfg.B * mpc::IPM_WARM_DOUBLES * sizeof(double), (hipStream_t)stream));
    h->order_B = 0;
    if (h->queue) HIP_TRY(h, hipMemsetAsync(h->queue, 0, 2 * sizeof(int), (hipStream_t)stream));
    if (h->sqp_x) HIP_TRY(h, hipMemsetAsync(h->sqp_x, 0, (size_t)h->cfg.B * 6 * ((size_t)h->cfg.N + 1) * sizeof(double), (hipStream_t)stream));
    return MPC_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// Handle-free solve that allocates nothing: the condensed Hessians and the solver records, which the handle-based entry
// points keep in the workspace mpc_create allocates, live in a file-scope device array.  Every GPU the library runs on
// holds its own copy of the array until the process ends; its address on a device is looked up once, with that device
// current, and only that address (never the array's name, which in host code is a host-side stand-in) goes to kernels.
constexpr int NOALLOC_MAX_B = MPC_NOALLOC_MAX_B, NOALLOC_MAX_N = MPC_NOALLOC_MAX_N;
constexpr size_t NOALLOC_H_DOUBLES = (size_t)NOALLOC_MAX_B * NOALLOC_MAX_N * NOALLOC_MAX_N;      // condensed Hessians
constexpr size_t NOALLOC_STAT_DOUBLES = (size_t)NOALLOC_MAX_B * mpc::STAT_W;                             // solver records
constexpr size_t NOALLOC_INT_DOUBLES = (size_t)NOALLOC_MAX_B;                                    // 2 * MAX_B ints: iterations, status
constexpr size_t NOALLOC_DOUBLES = NOALLOC_H_DOUBLES + NOALLOC_STAT_DOUBLES + NOALLOC_INT_DOUBLES;
static __device__ __align__(256) double scratch[NOALLOC_DOUBLES];
static double *g_scratch_ptr[mpc::MPC_MAX_DEVICES] = {};

int mpc_solve_only_noalloc(const mpc_config *cfg, void *stream, int B, const double *x0, const double *wp, const double *p,
                           const double *d_hat, double *U_out, double *traj_out, int *status_out, int *iters_out)
{
    if (!cfg) return fail(nullptr, MPC_EINVAL, "mpc_solve_only_noalloc: null configuration");
    if (const char *why = check_config(cfg, false)) return fail(nullptr, MPC_EINVAL, std::string("mpc_solve_only_noalloc: ") + why);   // cfg->B sizes a handle's workspace: not read here
    if (cfg->emulate_acados) return fail(nullptr, MPC_EINVAL, "mpc_solve_only_noalloc: emulate_acados keeps an iterate between calls: use a handle");
    if (cfg->kind != MPC_KIND_L && cfg->kind != MPC_KIND_OFL)
        return fail(nullptr, MPC_EINVAL, "mpc_solve_only_noalloc: kinds L and OFL only (LPV and NL carry state between calls: use a handle)");
    if (!mpc::horizon_templated(cfg->N) || cfg->N > NOALLOC_MAX_N)
        return fail(nullptr, MPC_EINVAL, "mpc_solve_only_noalloc: horizon must be 10, 20, 30, 40, 50 or 64");
    if (B < 1 || B > NOALLOC_MAX_B) {
        char b[160];
        snprintf(b, sizeof b, "mpc_solve_only_noalloc: B=%d is outside 1..%d (the static scratch array holds %d condensed problems)", B,
                 NOALLOC_MAX_B, NOALLOC_MAX_B);
        return fail(nullptr, MPC_EINVAL, b);
    }
    if (!x0 || !wp || !p || !U_out) return fail(nullptr, MPC_EINVAL, "mpc_solve_only_noalloc: x0, wp, p and U_out are required");
    if (cfg->device < 0 || cfg->device >= mpc::MPC_MAX_DEVICES) return fail(nullptr, MPC_EINVAL, "mpc_solve_only_noalloc: device ordinal out of range");
    DeviceScope scope(cfg->device);
    if (scope.err != hipSuccess) return hip_fail(nullptr, scope.err, "mpc_solve_only_noalloc: selecting the device");
    double *base = g_scratch_ptr[cfg->device];
    if (!base) {                                  // first call on this device: the array's address there
        hipError_t e = hipGetSymbolAddress((void **)&base, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !base) return hip_fail(nullptr, e, "mpc_solve_only_noalloc: hipGetSymbolAddress(scratch)");
        g_scratch_ptr[cfg->device] = base;
    }
    mpc::DevCfg dev;
    fill_devcfg(cfg, dev);
    int *ints = reinterpret_cast<int *>(base + NOALLOC_H_DOUBLES + NOALLOC_STAT_DOUBLES);
    mpc::SolveArgs a = {};
    a.mode = mpc::MODE_SOLVE; a.B = B; a.nsteps = 1; a.nx_plant = 8; a.x0 = x0; a.wp = wp; a.p = p; a.d_hat = d_hat;
    a.Hws = base;                                 // every scenario's Hessian is written before it is read: nothing to clear
    a.stat_out = base + NOALLOC_H_DOUBLES;
    a.iters_own = ints; a.status_own = ints + NOALLOC_MAX_B;
    a.U_out = U_out; a.traj_out = traj_out; a.status_out = status_out; a.iters_out = iters_out;
    hipError_t e = mpc::launch_mpc(dev, a, (hipStream_t)stream);
    if (e != hipSuccess) return hip_fail(nullptr, e, "mpc_solve_only_noalloc: launch");
    return MPC_OK;
}

int mpc_get_stats(mpc_handle *h, void *stream, double *host_out)
{
    if (!h || !host_out) return MPC_EINVAL;
    if (h->last_B < 1) return fail(h, MPC_EINVAL, "mpc_get_stats: no solve has been issued");
    ON_HANDLE_DEVICE(h);
    HIP_TRY(h, mpc::launch_stats((hipStream_t)stream, h->last_B, h->stat, h->iters, h->status, (h->cfg.kind == MPC_KIND_NL || h->sqp_x) ? h->sqp : nullptr, h->as2,
                                 h->cl_valid, h->dev, h->cost_scen, h->stats_out + MPC_STATS_LEN, h->stats_out));
    HIP_TRY(h, hipMemcpyAsync(host_out, h->stats_out, MPC_STATS_LEN * sizeof(double), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(h, hipStreamSynchronize((hipStream_t)stream));
    host_out[14] = h->sh_on && h->sh_T ? h->sh_T_resid : 0.0;
    host_out[15] = h->sh_on && h->sh_T && !h->sh_T_ok ? 1.0 : 0.0;
    return MPC_OK;
}

#define AUX_TRY(call)                                                         \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) return hip_fail(nullptr, e_, #call);            \
    } while (0)

int mpc_plant_rk4(void *stream, int B, int nx, double dt, const double *mp6_host, double *x, const double *u)
{
    if (B < 1 || (nx != 8 && nx != 10) || !mp6_host || !x || !u) return fail(nullptr, MPC_EINVAL, "mpc_plant_rk4: bad argument");
    AUX_TRY(mpc::launch_plant_rk4((hipStream_t)stream, B, nx, dt, mp6_host, x, u));
    return MPC_OK;
}

int mpc_plant_f(void *stream, int B, int nx, const double *mp6_host, const double *x, const double *u, double *xdot)
{
    if (B < 1 || (nx != 8 && nx != 10) || !mp6_host || !x || !u || !xdot) return fail(nullptr, MPC_EINVAL, "mpc_plant_f: bad argument");
    AUX_TRY(mpc::launch_plant_f((hipStream_t)stream, B, nx, mp6_host, x, u, xdot));
    return MPC_OK;
}

int mpc_ekf_time_update(void *stream, int B, double dt, const double *mp6_host, const double *q_diag_host, double *x,
                        double *P, const double *u)
{
    if (B < 1 || !mp6_host || !q_diag_host || !x || !P || !u) return fail(nullptr, MPC_EINVAL, "mpc_ekf_time_update: bad argument");
    AUX_TRY(mpc::launch_ekf_time_update((hipStream_t)stream, B, dt, mp6_host, q_diag_host, x, P, u));
    return MPC_OK;
}

int mpc_ekf_measurement_update(void *stream, int B, const double *r_diag_host, double *x, double *P, const double *y,
                               double *K_out)
{
    if (B < 1 || !r_diag_host || !x || !P || !y) return fail(nullptr, MPC_EINVAL, "mpc_ekf_measurement_update: bad argument");
    AUX_TRY(mpc::launch_ekf_meas_update((hipStream_t)stream, B, r_diag_host, x, P, y, K_out, 0));
    return MPC_OK;
}

int mpc_ekf_kalman_gain(void *stream, int B, const double *r_diag_host, const double *P, double *K_out)
{
    if (B < 1 || !r_diag_host || !P || !K_out) return fail(nullptr, MPC_EINVAL, "mpc_ekf_kalman_gain: bad argument");
    AUX_TRY(mpc::launch_ekf_meas_update((hipStream_t)stream, B, r_diag_host, nullptr, const_cast<double *>(P), nullptr, K_out, 1));
    return MPC_OK;
}

int mpc_ekf_jacobian(void *stream, int B, double dt, const double *mp6_host, const double *x, double *F)
{
    if (B < 1 || !mp6_host || !x || !F) return fail(nullptr, MPC_EINVAL, "mpc_ekf_jacobian: bad argument");
    AUX_TRY(mpc::launch_ekf_jacobian((hipStream_t)stream, B, dt, mp6_host, x, F));
    return MPC_OK;
}

int mpc_dlqr_run(void *stream, int B, int nx, int nsteps, double dt, const double *mp6_host, const double *K_host,
                 const double *K_per_scenario, double rate_limit, double angle_limit, double *x, double *u_hist,
                 double *x_hist)
{
    if (B < 1 || (nx != 8 && nx != 10) || nsteps < 1 || !(dt > 0.0) || !mp6_host || (!K_host && !K_per_scenario) || !x)
        return fail(nullptr, MPC_EINVAL, "mpc_dlqr_run: bad argument");
    AUX_TRY(mpc::launch_dlqr_run((hipStream_t)stream, B, nx, nsteps, dt, mp6_host, K_host, K_per_scenario, rate_limit,
                                 angle_limit, x, u_hist, x_hist));
    return MPC_OK;
}

int mpc_lsim_run(void *stream, int B, int nsteps, double dt, const double *mp6_host, const double *q_diag_host,
                 const double *r_diag_host, const double *u, int u_per_scenario, const double *meas_noise, double *state10,
                 double *ekf_x, double *ekf_P, double *x_hist, double *est_hist)
{
    if (B < 1 || nsteps < 1 || !(dt > 0.0) || !mp6_host || !q_diag_host || !r_diag_host || !u || !state10 || !ekf_x || !ekf_P)
        return fail(nullptr, MPC_EINVAL, "mpc_lsim_run: bad argument");
    AUX_TRY(mpc::launch_lsim_run((hipStream_t)stream, B, nsteps, dt, mp6_host, q_diag_host, r_diag_host, u, u_per_scenario,
                                 meas_noise, state10, ekf_x, ekf_P, x_hist, est_hist));
    return MPC_OK;
}

int mpc_control_buffer_interp(void *stream, int B, int N, double Tf, const double *traj, int ncol, int col,
                              const double *t, double scale, double *steer_out)
{
    if (B < 1 || N < 1 || !(Tf > 0.0) || !traj || ncol < 1 || col < 0 || col >= ncol || !t || !steer_out)
        return fail(nullptr, MPC_EINVAL, "mpc_control_buffer_interp: bad argument");
    AUX_TRY(mpc::launch_control_interp((hipStream_t)stream, B, N, Tf, traj, ncol, col, t, scale, steer_out));
    return MPC_OK;
}

int mpc_invariance_check(void *stream, int nc, int d, const double *Q_host, const double *eig_host, const double *A_host,
                         const double *B_host, const double *K_host, double umax, const double *c, int *ok_out)
{
    if (nc < 1 || d < 1 || d > 6 || !Q_host || !eig_host || !A_host || !B_host || !K_host || !(umax > 0.0) || !c || !ok_out)
        return fail(nullptr, MPC_EINVAL, "mpc_invariance_check: bad argument (1 <= d <= 6, umax > 0)");
    for (int i = 0; i < d; i++)
        if (!(eig_host[i] > 0.0)) return fail(nullptr, MPC_EINVAL, "mpc_invariance_check: P must be positive definite");
    AUX_TRY(mpc::launch_invariance((hipStream_t)stream, nc, d, Q_host, eig_host, A_host, B_host, K_host, umax, c, ok_out));
    return MPC_OK;
}

int mpc_lp_maximize(void *stream, int nlp, int d, int m, const double *A, const double *b, const double *c, double *opt_out,
                    double *x_out, int *status_out)
{
    if (nlp < 1 || d < 1 || d > 6 || m < 1 || m > MPC_LP_MAX_ROWS || !A || !b || !c || !opt_out || !status_out)
        return fail(nullptr, MPC_EINVAL, "mpc_lp_maximize: bad argument (1 <= d <= 6, 1 <= m <= MPC_LP_MAX_ROWS)");
    AUX_TRY(mpc::launch_lp_maximize((hipStream_t)stream, nlp, d, m, A, b, c, opt_out, x_out, status_out));
    return MPC_OK;
}

int mpc_plan_skidpad(void *stream, int B, int N, double dt, const double *pose, const int *lap, double *prev_progress,
                     const double *target_vel, double *wp_out, double *progress_out)
{
    if (B < 1 || N < 1 || !pose || !lap || !prev_progress || !target_vel || !wp_out) return fail(nullptr, MPC_EINVAL, "mpc_plan_skidpad: bad argument");
    AUX_TRY(mpc::launch_plan_skidpad((hipStream_t)stream, B, N, dt, pose, lap, prev_progress, target_vel, nullptr, wp_out,
                                     progress_out, nullptr));
    return MPC_OK;
}

int mpc_plan_skidpad_profile(void *stream, int B, int N, double dt, const double *pose, const int *lap,
                             double *prev_progress, const double *target_vel, const double *zones, double *wp_out,
                             double *progress_out, double *speeds_out)
{
    if (B < 1 || N < 1 || !pose || !lap || !prev_progress || !target_vel || !zones || !wp_out)
        return fail(nullptr, MPC_EINVAL, "mpc_plan_skidpad_profile: bad argument");
    AUX_TRY(mpc::launch_plan_skidpad((hipStream_t)stream, B, N, dt, pose, lap, prev_progress, target_vel, zones, wp_out,
                                     progress_out, speeds_out));
    return MPC_OK;
}

int mpc_plan_step(void *stream, int B, int N, double dt, double ramp_length, const double *pose, const double *target_vel,
                  double *wp_out, double *progress_out, double *abs_out)
{
    if (B < 1 || N < 1 || !(ramp_length >= 0.0) || !pose || !target_vel || !wp_out)
        return fail(nullptr, MPC_EINVAL, "mpc_plan_step: bad argument");
    AUX_TRY(mpc::launch_plan_step((hipStream_t)stream, B, N, dt, ramp_length, pose, target_vel, wp_out, progress_out, abs_out));
    return MPC_OK;
}

}  // extern "C"
