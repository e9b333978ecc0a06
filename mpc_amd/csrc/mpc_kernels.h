// Internal declarations shared by the translation units of libmpc_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include "../../include/mpc_amd.h"

// Horizons with a register-resident solver (the Newton-matrix row of a lane lives in N statically indexed registers);
// every other horizon up to MPC_MAX_HORIZON runs on the any-horizon kernel, whose Newton matrix lives in LDS.
#ifndef MPC_HORIZONS
#define MPC_HORIZONS(X) X(10) X(20) X(30) X(40) X(50) X(64)
#endif

namespace mpc {

constexpr int STAT_W = 20;                 // doubles per scenario in SolveArgs::stat_out (4 solver values, CL_STAT_W closed-loop statistics, spare)
constexpr int STAT_CL = 4;                 // where the closed-loop statistics start in that record
constexpr int IPM_WARM_DOUBLES = 7 * 128;  // interior-point warm-start record per scenario (see IpmParams); the any-horizon
                                           // kernel keeps [5][128] and its flag at 5 * 128, the register kernels [5][64] and 5 * 64

enum { MODE_SOLVE = 0, MODE_SKIDPAD = 1, MODE_STEP = 2, MODE_OFL = 3 };   // OFL: step planner on the EKF estimate

struct DevCfg {
    int kind, N, lpv_lin_prev, max_iter;
    double dt, eps, max_steering, max_rate, tol_mu, tol_res;
    double mp[6];
    double w[7];
    double We[36];
    // nonlinear controller (MPC_KIND_NL) only
    double w_nl[8], we_nl[7], eps_nl, sqp_tol;
    int sqp_max_iter;
    // mpc_config.emulate_acados: regularisation and stop tolerance of the emulated SQP iteration
    double reg_eps, kkt_tol;
};

struct SolveArgs {
    int mode, B;
    int nsteps;                                           // closed-loop modes: steps advanced by one launch
    double *u_hist, *x_hist;                              // optional [nsteps][B] inputs / [nsteps][B][nx_plant] states
    int nx_plant;                                         // 8, or 10 with the disturbance states (MODE_OFL)
    double *ekf_x, *ekf_P, *est_hist;                     // MODE_OFL: estimate [B][10], covariance [B][10][10], [nsteps][B][10]
    const double *meas_noise;                             // MODE_OFL: optional additive noise [nsteps][B][7]
    double ekf_q[10], ekf_r[7];                           // MODE_OFL: diag(Q), diag(R)
    const double *x0, *wp, *p, *d_hat;                    // MODE_SOLVE inputs
    double *state; int *lap; double *prev_progress;       // closed-loop state (updated)
    const double *target_vel;
    double step_ramp;                                     // MODE_STEP / MODE_OFL: ramp length of the step reference (0 = pure step)
    double *lin_x, *lin_u;                                // LPV linearisation memory (updated)
    const double *mp_scen;                                // optional per-scenario model parameters
    const double *cost_scen;                              // optional per-scenario cost [B][43]: diag W (7), then W_e (6x6)
    double *Hws;                                          // workspace: B * N * N condensed Hessians
    double *ipm_ws;                                       // workspace: B * IPM_WARM_DOUBLES interior-point warm-start records (or NULL)
    double *U_out, *traj_out, *u0_out, *wp_out;
    int *status_out, *iters_out;                          // caller's arrays (optional)
    int *status_own, *iters_own;                          // the handle's own copies (always written; mpc_get_stats reads them)
    int *sqp_out;                                         // NL: SQP iterations per scenario (summed over the steps)
    double *stat_out;                                     // B * STAT_W: mu, residual, constrained, |u0|, then (closed-loop modes) the
                                                          // CL_STAT_W closed-loop statistics of the run (mpc_device.hpp); one record, so
                                                          // that the kernels address both through one pointer
    // A long closed-loop run is issued as a chain of launches: `order` (a permutation of 0..B-1, or NULL) maps the
    // wave index to the scenario it advances, so that scenarios that needed many solver iterations in the previous
    // launch start first and share a workgroup; work_out [B] receives this launch's iterations per scenario; with
    // `accumulate` the per-scenario totals (iterations, worst status, constrained solves, SQP iterations) add to what
    // the previous launch of the chain left instead of replacing it.
    const int *order;
    int *work_out;
    int accumulate;
    // Optional work queue of the fixed-model kernel: two ints, {next position in `order`, waves that have left}, both zero
    // between launches.  With it the launch holds only as many workgroups as the GPU keeps resident and every wave takes
    // the next scenario when it has finished one, instead of a workgroup holding its slot until its slowest wave is done.
    int *queue;
    // Fixed-model cache (every scenario at speed sh_vx with the handle's model parameters); NULL = off.
    const double *sh_H;      // [N][N] condensed Hessian
    const double *sh_Kref;   // [3][N][N]: inv(H) * d g / d ref_c(stage k+1), stored [c][k][i]
    const double *sh_Kx;     // [5][N]: -inv(H) * g for unit vy, r, delta, the constant term, unit d_hat
    const double *sh_T;      // [2N][2N]: [I; P] inv(H) [I P'], P = the steering rows (active-set solve; NULL: interior point only)
    int *as_out;             // [B][6]: active-set solves that ended at the minimiser, active-set iterations, solves handed to the
                             // interior point because the set outgrew the capacity / did not settle or certify, the method's
                             // flops in units of 64, one spare (or NULL)
    double sh_vx;
    int sh_exclusive;        // caller promises every scenario runs at sh_vx: launch the fixed-model-only kernel
    int sh_mask;             // bit c set: reference component c (x, y, sin psi) carries weight
    // Terminal set C (x_N[1:] - x_ref) <= 1 (lane-per-scenario layout only): rows [term_m][5] in device memory, 0 = none.
    int term_m;
    const double *term_C;
    int px_free;             // no cost on the longitudinal position anywhere (lane layout: the solver build without px)
    // mpc_config.emulate_acados: the SQP iterate (predicted states of the previous solve) per scenario, [B][6][N+1], read as
    // the anchor of the regularisation and updated; NULL = off
    double *sqp_x;
    // Kernels whose Hessian differs per scenario and solve (NL): 1 = try the active-set solve in its free-input form (it
    // needs H alone) before the interior point; as_out receives its counters as on the fixed-model path
    int general_aset;
};

// Raises a kernel's dynamic-LDS limit to the CU's 160 KB once per device and kernel (`done` is that kernel's flag table).
constexpr int MPC_MAX_DEVICES = 64;
inline hipError_t allow_big_lds(const void *kernel, bool (&done)[MPC_MAX_DEVICES])
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= MPC_MAX_DEVICES) return hipErrorInvalidDevice;
    if (done[dev]) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) done[dev] = true;
    return e;
}

hipError_t launch_mpc(const DevCfg &cfg, const SolveArgs &a, hipStream_t st);
hipError_t launch_mpc_nl(const DevCfg &cfg, const SolveArgs &a, hipStream_t st);    // MPC_KIND_NL, any horizon 2..64: runs on the instantiation nl_horizon_capacity(N)
int nl_horizon_capacity(int N);    // smallest horizon of MPC_HORIZONS >= N (0: none); the Hessian scratch of an NL handle is B * capacity^2
hipError_t launch_mpc_any(const DevCfg &cfg, const SolveArgs &a, hipStream_t st);   // run-time horizon, no cache
// one lane per scenario, stage-wise interior point (kinds L / OFL / LPV, any horizon); ws holds
// lanes_workspace_doubles(N, Bcap) doubles laid out for a capacity of Bcap scenarios, and keeps the warm-start record between calls
hipError_t launch_mpc_lanes(const DevCfg &cfg, const SolveArgs &a, double *ws, size_t Bcap, hipStream_t st);
size_t lanes_workspace_doubles(int N, int B, int term_m);
// resid (one double, device; may be NULL): receives the largest entry of |H inv(H) - I| of the table T
hipError_t launch_condense_fixed(const DevCfg &cfg, double vx, double *H, double *Kref, double *Kx, double *T, double *resid, hipStream_t st);
bool horizon_supported(int N);    // 2..MPC_MAX_HORIZON
bool horizon_templated(int N);    // has a register-resident instantiation (and with it the fixed-model cache)

hipError_t launch_plant_rk4(hipStream_t st, int B, int nx, double dt, const double *mp6, double *x, const double *u);
hipError_t launch_plant_f(hipStream_t st, int B, int nx, const double *mp6, const double *x, const double *u, double *xd);
hipError_t launch_ekf_time_update(hipStream_t st, int B, double dt, const double *mp6, const double *qd, double *x,
                                  double *P, const double *u);
hipError_t launch_ekf_meas_update(hipStream_t st, int B, const double *rd, double *x, double *P, const double *y,
                                  double *K_out, int gain_only);
hipError_t launch_ekf_jacobian(hipStream_t st, int B, double dt, const double *mp6, const double *x, double *F);
hipError_t launch_plan_skidpad(hipStream_t st, int B, int N, double dt, const double *pose, const int *lap,
                               double *prev_progress, const double *target_vel, const double *zones, double *wp_out,
                               double *progress_out, double *speeds_out);
hipError_t launch_plan_step(hipStream_t st, int B, int N, double dt, double ramp, const double *pose, const double *target_vel,
                            double *wp_out, double *progress_out, double *abs_out);
hipError_t launch_dlqr_run(hipStream_t st, int B, int nx, int nsteps, double dt, const double *mp6, const double *K5,
                           const double *K_scen, double rate_limit, double angle_limit, double *x, double *u_hist,
                           double *x_hist);
hipError_t launch_lsim_run(hipStream_t st, int B, int nsteps, double dt, const double *mp6, const double *qd,
                           const double *rd, const double *u, int u_per_scenario, const double *meas_noise, double *x,
                           double *ekf_x, double *ekf_P, double *x_hist, double *est_hist);
hipError_t launch_control_interp(hipStream_t st, int B, int N, double Tf, const double *traj, int ncol, int col,
                                 const double *t, double scale, double *out);
hipError_t launch_balance_order(hipStream_t st, int B, const int *work, int shift, int *order);
hipError_t launch_invariance(hipStream_t st, int nc, int d, const double *Q, const double *eig, const double *A,
                             const double *B, const double *K, double umax, const double *c, int *ok_out);
// reduction of the per-scenario records into MPC_STATS_LEN doubles; part: STATS_BLOCKS * MPC_STATS_LEN doubles of scratch
constexpr int STATS_BLOCKS = 256;
hipError_t launch_lp_maximize(hipStream_t st, int nlp, int d, int m, const double *A, const double *b, const double *c,
                              double *opt_out, double *x_out, int *status_out);
hipError_t launch_stats(hipStream_t st, int B, const double *stat, const int *iters, const int *status, const int *sqp,
                        const int *as2, bool with_closed_loop, const DevCfg &cfg, const double *cost_scen, double *part, double *out);

}  // namespace mpc
