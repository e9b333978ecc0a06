/*
 * mpc_amd.h -- C ABI of libmpc_amd.so: the MI355X (gfx950) implementation of the skidpad MPC
 * per-step control path (plan -> linearise/condense -> QP solve -> RK4 plant -> EKF).
 *
 * All buffers are caller-owned DEVICE pointers unless a parameter says "host".  Scenario records are
 * contiguous ("scenario-major"): state[B][8], wp[B][N+1][4], U[B][N] ... so that the wavefront that
 * owns a scenario reads its record with one coalesced access.  Every entry point is asynchronous on
 * the hipStream_t passed as `void *stream` (NULL = default stream) and returns 0 on success or a
 * negative MPC_E* code; mpc_last_error() gives the text.  No exceptions cross this boundary and the
 * library owns nothing but the handle and its workspace.
 *
 * Concurrency: a handle carries workspace that every launch of it uses (condensed Hessians, interior-point warm-start
 * records, the scenario order of chained launches, the solver records mpc_get_stats reads).  Issue the calls of one
 * handle one after the other on ONE stream; two launches of the same handle in flight at once race on that workspace.
 * Different handles are independent.  The warm-start records tie a handle to the closed loops it advances: after
 * replacing the state arrays with unrelated scenarios call mpc_reset_warm_start (a stale record is otherwise only
 * rejected by its residual test, and the results stop being reproducible run to run).
 *
 * Devices: a handle belongs to mpc_config.device.  Entry points that take a handle make that device
 * current while they run and give the caller's current device back on return; the stream and the
 * buffers passed must live on the handle's device.  Entry points without a handle run on the calling
 * thread's current device.
 *
 * The reference has no FFI of its own for this path: its controllers hand the problem to
 * acados_template's ctypes binding (AcadosOcpSolver.cost_set/set/solve_for_x0/get, LMPC2.py:301-329,
 * LPVMPC.py:260-312, OFLMPC2.py:320-349).  The entry points below are what that binding is
 * replaced by; each cites the reference interface it stands in for.
 */
#ifndef MPC_AMD_H
#define MPC_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define MPC_OK 0
#define MPC_EINVAL (-1)   /* bad argument / unsupported horizon */
#define MPC_EHIP (-2)     /* a HIP runtime call failed */
#define MPC_ENOMEM (-3)

#define MPC_KIND_L 0      /* LMPC2.py   LOcp   */
#define MPC_KIND_OFL 1    /* OFLMPC2.py OFLOcp */
#define MPC_KIND_LPV 2    /* LPVMPC.py  LPVOcp */
#define MPC_KIND_NL 3     /* NLMPC.py   NLOcp  (any horizon 2..64) */

#define MPC_PLANNER_SKIDPAD 0 /* utils/path_planning.py SkidpadPlanner */
#define MPC_PLANNER_STEP 1    /* utils/step_planning.py StepPlanner    */

#define MPC_MAX_HORIZON 128

#define MPC_MAX_TERMINAL_ROWS 256

#define MPC_LAYOUT_AUTO 0
#define MPC_LAYOUT_WAVE 1
#define MPC_LAYOUT_LANE 2

#define MPC_QP_AUTO 0            /* active-set solve first where the fixed-model cache is on, interior point otherwise */
#define MPC_QP_INTERIOR_POINT 1  /* interior point for every constrained solve */
#define MPC_QP_ACTIVE_SET 2      /* the same as AUTO */

/* Controller definition; what the reference's constructors derive from parameters_*.yaml
 * (LMPC2.py:14-96, LPVMPC.py:10-92, OFLMPC2.py:16-100). */
typedef struct mpc_config {
    int kind;            /* MPC_KIND_*                                                        */
    int N;               /* horizon, 2..MPC_MAX_HORIZON                                       */
    int B;               /* number of scenarios the workspace is sized for                    */
    int device;          /* HIP device ordinal                                                */
    double dt;           /* Tf / N                                                            */
    double mp[6];        /* m, I_z, lf, lr, Cf, Cr                                            */
    double max_steering; /* |delta| bound, stages 1..N-1                                      */
    double max_rate;     /* |u| bound, stages 0..N-1                                          */
    double w[7];         /* diag of W = blkdiag(Q*q, R*r); stage cost is dt * 1/2 |y-yref|^2_W */
    double We[36];       /* terminal weight on the 6 states (row-major)                       */
    double eps;          /* tyre-model denominator offset (1e-3)                              */
    double tol_mu;       /* interior-point stop: complementarity (default 1e-15)                                          */
    double tol_res;      /* interior-point stop: stationarity / feasibility residual, relative (default 1e-12).  A solve also
                          * ends with status 0 at an iterate inside the acceptance band mu <= 10 tol_mu and residual <=
                          * 1e3 tol_res when the residual has stopped halving, when mu has fallen below 1e-3 tol_mu, at the
                          * iteration cap, or when the next iterate is worse or not finite (rounding sets the floor there:
                          * inputs within 1e-9 of the minimiser); mpc_get_stats [6], [7] report what was reached. */
    int max_iter;        /* interior-point iteration cap                                      */
    int lpv_lin_prev;    /* LPV only: 0 = lin_mode "reference", 1 = "prev_iter" (LPVMPC.py:254) */
    /* MPC_KIND_NL only (w, We, eps above are not read for it); NLMPC.py:159-186 hard-codes
     * W = diag(0,1e5,1e-10,1,0,1,1 | 10), W_e = 100 * diag of the state part, and +0.1 in the tyre denominators */
    double w_nl[8];      /* diag of W over [px,py,cos,sin,vy,r,delta | u]; stage cost is dt * 1/2 |y-yref|^2_W */
    double we_nl[7];     /* diag of the terminal weight on the 7 states                        */
    double eps_nl;       /* tyre-model denominator offset of the NL model (0.1)               */
    double sqp_tol;      /* SQP stop: largest change of an input in one iteration             */
    int sqp_max_iter;    /* SQP iteration cap per solve (status 1 when reached)               */
    int ipm_cold_start;  /* 0 (default): closed-loop steps, and the SQP iterations of one NL solve, start the interior
                          * point from the previous solve when that one hit a bound too; 1: always cold start */
    double step_ramp_length; /* MPC_PLANNER_STEP loops: progress over which the step reference rises (StepPlanner's
                              * ramp_length, utils/step_planning.py:51,75-79); 0 = the pure step the simulators use */
    int solver_layout;   /* how scenarios map to the GPU (results agree to the solver tolerances, not bit for bit):
                          * MPC_LAYOUT_AUTO; MPC_LAYOUT_WAVE: one wavefront per scenario, horizon condensed into a dense
                          * N x N problem; MPC_LAYOUT_LANE: one lane per scenario, stage-wise Riccati interior point
                          * (kinds L / OFL / LPV; not for mpc_ofl_closed_loop_run).  AUTO = LANE where it is the faster one:
                          * kinds L and LPV at horizons other than 10, 20, 30, 40, 50, 64 with B >= 2048; else WAVE */
    int qp_method;       /* how a solve whose unconstrained minimiser violates a bound is finished.  MPC_QP_AUTO: the
                          * active bounds are found by an active-set iteration started from a prediction of the set (the
                          * previous closed-loop steps' sets; for the NL kind also the previous SQP iteration's), each
                          * iteration solving the smaller of two small systems -- in the multipliers, built from the
                          * cached inv(H) (fixed-model cache on, mpc_condense_fixed), or in the inputs the set leaves free,
                          * built from H itself (also without the cache: wave layout at the horizons 10, 20, 30, 40, 50,
                          * 64, kinds L / OFL / LPV in closed loops and NL); the result is refined through that system and
                          * certified by the size of the refinement, and whatever does not end that way (large or cycling
                          * sets) goes to the interior point.  Both give the unique minimiser (they agree to the solver
                          * tolerances, not bit for bit).  MPC_QP_INTERIOR_POINT: interior point only.
                          * MPC_QP_ACTIVE_SET: the same as AUTO (kept for callers written when mpc_ofl_closed_loop_run
                          * needed it to use the method).  The lane layout and horizons without a register
                          * instantiation always use the interior point. */
    int emulate_acados;  /* 0 (default): every solve returns the unique minimiser of the problem the reference poses.
                          * 1 (kinds L, OFL, LPV at the horizons 10, 20, 30, 40, 50, 64; wave layout): the solve goes
                          * through what the reference's solver options (LMPC2.py:262-288: SQP, EXACT Hessian,
                          * regularize_method MIRROR, tol 1e-4, iterate kept between calls) do to that problem instead:
                          * every cost weight below reg_epsilon (the unweighted states px, vy, r) is lifted to
                          * reg_epsilon, which makes one SQP step the minimiser of the problem plus
                          * 1/2 reg_epsilon |x_i - x_prev,i|^2 on those states about the iterate kept from the previous
                          * step; full steps are repeated until reg_epsilon * max |x - x_prev| <= sqp_kkt_tol (the
                          * stationarity residual the regularisation leaves; at most sqp_max_iter steps, status 1 beyond),
                          * and the iterate stays in the handle for the next call (mpc_reset_warm_start zeroes it, as a
                          * new solver object starts).  A restatement of published behaviour, as unpinned as the rest of
                          * the QP parity: acados itself cannot run beside it.  The fixed-model cache is not used. */
    double reg_epsilon;  /* emulate_acados: acados' reg_epsilon (default 1e-4)                                   */
    double sqp_kkt_tol;  /* emulate_acados: the reference's tolerance on the KKT residual (default 1e-4)           */
} mpc_config;

typedef struct mpc_handle mpc_handle;

/* Replaces AcadosOcpSolver(ocp) construction (LMPC2.py:94). */
int mpc_create(const mpc_config *cfg, mpc_handle **out);
void mpc_destroy(mpc_handle *h);
const char *mpc_last_error(const mpc_handle *h); /* h may be NULL: last error of mpc_create */

/* Optional per-scenario model parameters mp[B][6] (tyre / mass sweeps); NULL restores cfg.mp.  This call and
 * mpc_set_cost_weights copy with a blocking hipMemcpy into the handle: the source must be complete, and no launch
 * of this handle may be in flight on another stream. */
int mpc_set_model_params(mpc_handle *h, const double *mp_per_scenario);

/* Optional per-scenario cost weights cost[B][43] (device): diag of W (7 entries, as cfg.w) followed by W_e (6x6
 * row-major, as cfg.We), for tuning sweeps where every scenario is one point of the sweep (plot_generator.py:168-515
 * rewrites parameters_*.yaml and rebuilds the controller per point); NULL restores cfg.w / cfg.We.  Scenarios are then
 * condensed individually (the fixed-model cache is bypassed).  Not for MPC_KIND_NL, whose weights are hard-coded. */
int mpc_set_cost_weights(mpc_handle *h, const double *cost_per_scenario);

/* Fixed-model cache for the linear controllers (kind L / OFL): condenses the horizon once for "every stage at
 * speed vx" (the skidpad planner hands the controller a constant speed profile, path_planning.py:238-240) and
 * caches H and inv(H) times every way the gradient depends on the step's data, and (unless qp_method is
 * MPC_QP_INTERIOR_POINT) the table [I; P] inv(H) [I P'] of the active-set solve (P = the steering rows; (2N)^2 doubles).  Scenarios whose speed profile
 * is not exactly vx still take the per-scenario condensing path inside the same launch, unless `exclusive` is
 * set: then the caller promises that every scenario runs at vx, a leaner kernel without the per-scenario path is
 * launched, and a scenario that breaks the promise is not solved but reported with status 3.  vx = NaN switches
 * the cache off.  Horizons without a register-resident kernel (anything but 10, 20, 30, 40, 50, 64) have no cache:
 * the call succeeds and changes nothing, and those horizons keep condensing per scenario.  With the table it waits for
 * the set-up kernel (the table is checked against H: mpc_get_stats slots 14, 15); without it the call is asynchronous.  Plays the role of the one-time QP data set-up inside AcadosOcpSolver(ocp) (LMPC2.py:94). */
int mpc_condense_fixed(mpc_handle *h, void *stream, double vx, int exclusive);

/* optimize(): the yref/p set loop + solve_for_x0 + x/u get loop of LMPC2.py:296-347,
 * OFLMPC2.py:315-367, LPVMPC.py:254-341, for B scenarios.
 *   x0      [B][7]  reduced state [pose(4), vy, r, delta]
 *   wp      [B][N+1][4] car-frame waypoints, p [B][N+1] speeds, d_hat [B] (OFL) or NULL
 *   lin_x   [B][6][N+1], lin_u [B][N]  LPV prev_x / prev_u, read and updated (NULL otherwise)
 *   U_out   [B][N]; traj_out [B][N+1][7] or NULL; status_out [B], iters_out [B]
 *   status: 0 ok; 2 the interior point ended without a minimiser (iteration cap or breakdown of the arithmetic, also after
 *   its cold restart); 3 speed differs from the exclusive fixed-model cache; 5 (closed-loop skidpad entry points) the lap
 *   counter left 0..7, where the reference's planner switches to its speed profile (utils/path_planning.py:241-248) and
 *   the fused loops keep the constant target speed: such steps differ from the reference (mpc_plan_skidpad_profile has
 *   that branch); 4 the start
 *   steering angle lies so far outside its bound that the reference's QP is infeasible (acados raises there): the
 *   bound was widened along the fastest way back inside and that problem solved instead
 * MPC_KIND_NL (NLMPC.py:214-252): lin_u [B][N] is required and holds the SQP iterate, read as the starting point and
 * updated with the solution, the way the reference's solver object keeps its iterate from call to call (zeros before
 * the first call); lin_x is not used; traj_out rows are the 7 NL states [x, y, cos, sin, vy, r, delta]; status 1 =
 * SQP iteration cap.  The closed-loop entry points below take lin_u the same way. */
int mpc_solve_only(mpc_handle *h, void *stream, int B, const double *x0, const double *wp, const double *p,
                   const double *d_hat, double *lin_x, double *lin_u, double *U_out, double *traj_out,
                   int *status_out, int *iters_out);

/* optimize() with HOST arrays in and out and a blocking return: the calling convention of the reference's own
 * LOcp / LPVOcp / OFLOcp / NLOcp.optimize (numpy in, numpy out; LMPC2.py:296-347), for one controller or a handful
 * (B <= min(cfg.B, MPC_HOST_MAX_B)).  Arguments as mpc_solve_only, but every pointer is host memory; lin_x / lin_u are
 * read and updated in the caller's arrays.  The library packs the arguments into a block of pinned host memory that the
 * GPU addresses directly (allocated by the first call on a handle, none later): one kernel launch and one stream
 * synchronisation per call, no copy commands -- the latency a controller that solves one problem per control period sees. */
#define MPC_HOST_MAX_B 64
int mpc_optimize_host(mpc_handle *h, void *stream, int B, const double *x0, const double *wp, const double *p,
                      const double *d_hat, double *lin_x, double *lin_u, double *U_out, double *traj_out, int *status_out,
                      int *iters_out);

/* LPVOcp.optimize(..., lin_mode) (LPVMPC.py:254-297): 0 = "reference", 1 = "prev_iter", for the following solves of an
 * LPV handle (the creation-time value is mpc_config.lpv_lin_prev). */
int mpc_set_lpv_lin_mode(mpc_handle *h, int lin_prev);

/* mpc_solve_only for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc): no handle, no workspace argument, no hipMalloc / hipFree inside.  It has no counterpart in
 * the reference.  The condensed problems live in a static device array of the library (2.1 MB per GPU, held until the
 * process ends), which limits a call to MPC_NOALLOC_MAX_B scenarios and the horizons 10, 20, 30, 40, 50, 64; more is refused
 * with MPC_EINVAL.  Kinds L and OFL (cfg->B, the cache and the warm start do not apply: every solve condenses its own
 * problem and starts cold, so the results equal mpc_solve_only's on a handle without fixed-model cache bit for bit).
 * All calls on one GPU share the array: calls that overlap in time on one GPU (two streams, two threads) race; issue them
 * one after the other.  The first call on each GPU looks the array's address up (hipGetSymbolAddress, which may also load
 * the library's code object on that GPU): make one call per GPU outside graph capture before capturing.  Arguments as
 * mpc_solve_only; errors are read with mpc_last_error(NULL). */
#define MPC_NOALLOC_MAX_B 64
#define MPC_NOALLOC_MAX_N 64
int mpc_solve_only_noalloc(const mpc_config *cfg, void *stream, int B, const double *x0, const double *wp, const double *p,
                           const double *d_hat, double *U_out, double *traj_out, int *status_out, int *iters_out);

/* One fused closed-loop step per scenario: plan -> optimize -> RK4 plant -> lap count
 * (SkidpadSimulator.step, skidpad_simulator.py:109-135; StepSimulator.simulate body, simulator.py:178-202).
 *   state [B][8] plant state (updated), lap [B] (updated), prev_progress [B] planner memory (updated),
 *   target_vel [B]; u0_out [B]; wp_out [B][N+1][4] / traj_out [B][N+1][7] / U_out [B][N] optional (NULL). */
int mpc_closed_loop_step(mpc_handle *h, void *stream, int B, int planner, double *state, int *lap,
                         double *prev_progress, const double *target_vel, double *lin_x, double *lin_u,
                         double *u0_out, double *wp_out, double *traj_out, double *U_out, int *status_out,
                         int *iters_out);

/* nsteps fused closed-loop steps in ONE launch (StepSimulator.simulate(n_steps), simulator.py:172-204, and a
 * for-loop over SkidpadSimulator.step): scenarios are independent, so each wavefront advances its scenario
 * through all steps with the plant state in registers and no host round trip.  iters_out receives the
 * interior-point iterations summed over the steps, status_out the worst status; u_hist [nsteps][B] and
 * x_hist [nsteps][B][8] (plant state after each step) are optional.  wp/traj/U outputs hold the last step. */
int mpc_closed_loop_run(mpc_handle *h, void *stream, int B, int planner, int nsteps, double *state, int *lap,
                        double *prev_progress, const double *target_vel, double *lin_x, double *lin_u,
                        double *u0_out, double *wp_out, double *traj_out, double *U_out, double *u_hist,
                        double *x_hist, int *status_out, int *iters_out);

/* Output-feedback closed loop, nsteps fused steps per launch (StepSimulator.simulate_of, simulator.py:206-281):
 * step planner and OFL controller on the EKF estimate (d_hat = estimate[9]), 10-state plant with disturbances,
 * EKF predict (EKF.py:38-43) and update (EKF.py:45-49) on y = H x_true (+ meas_noise[nsteps][B][7] if given).
 *   state10 [B][10] true plant state, ekf_x [B][10], ekf_P [B][10][10] (all updated); q_diag / r_diag host arrays
 *   of 10 / 7; u_hist [nsteps][B], x_hist [nsteps][B][10], est_hist [nsteps][B][10] optional. */
int mpc_ofl_closed_loop_run(mpc_handle *h, void *stream, int B, int nsteps, double *state10, double *ekf_x, double *ekf_P,
                            const double *target_vel, const double *q_diag_host, const double *r_diag_host,
                            const double *meas_noise, double *u0_out, double *u_hist, double *x_hist, double *est_hist,
                            int *status_out, int *iters_out);

/* Dynamics.single_track_model / rk4_integraton (continuous_dynamics.py:90-133): x [B][nx], nx = 8 or 10. */
int mpc_plant_rk4(void *stream, int B, int nx, double dt, const double *mp6_host, double *x, const double *u);
int mpc_plant_f(void *stream, int B, int nx, const double *mp6_host, const double *x, const double *u, double *xdot);

/* CarEKF (EKF.py:38-49,65-74) with Dynamics.jacobian_forward_euler (continuous_dynamics.py:167-210):
 * x [B][10], P [B][10][10]; q_diag / r_diag are host arrays of 10 / 7. */
int mpc_ekf_time_update(void *stream, int B, double dt, const double *mp6_host, const double *q_diag_host, double *x,
                        double *P, const double *u);
int mpc_ekf_measurement_update(void *stream, int B, const double *r_diag_host, double *x, double *P, const double *y,
                               double *K_out /* [B][10][7] or NULL */);
/* CarEKF.kalman_gain (EKF.py:65-74) on its own: K_out [B][10][7] from P [B][10][10]. */
int mpc_ekf_kalman_gain(void *stream, int B, const double *r_diag_host, const double *P, double *K_out);
int mpc_ekf_jacobian(void *stream, int B, double dt, const double *mp6_host, const double *x, double *F);

/* StepSimulator.dlqr_sim (simulator.py:378-428), nsteps per launch: the saturated LQR baseline u = K (ref - x[[1,3,5,6,7]])
 * with ref = (1/500, 0, 0, 0, 0), clipped to +-rate_limit and then to what keeps the last state entry within
 * +-angle_limit over one dt, followed by the RK4 plant step.  x [B][nx] (nx = 8 or 10, updated); K_host is a host
 * array of 5 (the DARE gain of the L controller, LMPC2.py:433-440), or K_per_scenario [B][5] on the device for gain
 * sweeps (takes precedence); u_hist [nsteps][B] and x_hist [nsteps][B][nx] optional. */
int mpc_dlqr_run(void *stream, int B, int nx, int nsteps, double dt, const double *mp6_host, const double *K_host,
                 const double *K_per_scenario, double rate_limit, double angle_limit, double *x, double *u_hist,
                 double *x_hist);

/* StepSimulator.lsim (simulator.py:334-376), nsteps per launch: an open-loop input sequence through the 10-state plant
 * and the EKF (time update with the same input, measurement update on the measured entries of the new plant state,
 * + meas_noise[nsteps][B][7] if given).  u is [nsteps] (device) shared by all scenarios, or [nsteps][B] when
 * u_per_scenario != 0; state10 [B][10], ekf_x [B][10], ekf_P [B][10][10] are updated; x_hist / est_hist
 * [nsteps][B][10] optional. */
int mpc_lsim_run(void *stream, int B, int nsteps, double dt, const double *mp6_host, const double *q_diag_host,
                 const double *r_diag_host, const double *u, int u_per_scenario, const double *meas_noise, double *state10,
                 double *ekf_x, double *ekf_P, double *x_hist, double *est_hist);

/* SkidpadPlanner.request_waypoints (utils/path_planning.py:220-261, constant-speed branch whatever the lap) and
 * StepPlanner.request_waypoints (utils/step_planning.py:88-117): pose [B][3] = x, y, heading;
 * wp_out [B][N+1][4]; progress_out [B]; abs_out [B][N+1][4] (step planner, may be NULL); ramp_length >= 0 is the
 * StepPlanner attribute of that name (step_planning.py:51,75-79), 0 for the pure step. */
int mpc_plan_skidpad(void *stream, int B, int N, double dt, const double *pose, const int *lap, double *prev_progress,
                     const double *target_vel, double *wp_out, double *progress_out);
int mpc_plan_step(void *stream, int B, int N, double dt, double ramp_length, const double *pose, const double *target_vel,
                  double *wp_out, double *progress_out, double *abs_out);
/* SkidpadPlanner.request_waypoints with its speed profile (utils/path_planning.py:220-261 incl. the branch for laps
 * outside 0..7, :241-248, and progress2speed, :93-126): as mpc_plan_skidpad, plus zones [B][7] = accel_zone_1_start,
 * accel_zone_1_end, braking_zone_1_start, braking_zone_1_end, braking_zone_2_start, braking_zone_2_end,
 * slowdown_speed_factor (the planner's attributes, :63-84), and speeds_out [B][N+1] (optional): the target speed on laps
 * 0..7, the profile (last entry 0) otherwise. */
int mpc_plan_skidpad_profile(void *stream, int B, int N, double dt, const double *pose, const int *lap,
                             double *prev_progress, const double *target_vel, const double *zones, double *wp_out,
                             double *progress_out, double *speeds_out);

/* Control-buffer read-out of the vehicle node (archive/mpc_node.py:421-427): between two solves the steering command is
 * the predicted steering angle, column `col` of traj [B][N+1][ncol] (6 of the 7-column traj_out above), interpolated
 * like np.interp at t [B] seconds since the solve over the node times linspace(0, Tf, N+1) (clamped outside), times
 * `scale` (pi / 2 / 0.4 there): steer_out [B]. */
int mpc_control_buffer_interp(void *stream, int B, int N, double Tf, const double *traj, int ncol, int col,
                              const double *t, double scale, double *steer_out);

/* check_control_amissible_invariance (stability.py:155-189) for nc level values c [nc] (device): the box around the
 * ellipsoid x' P x <= c in the eigenbasis of P (eigenvectors Q [d][d] as columns, eigenvalues eig [d]; host arrays, as
 * numpy.linalg.eigh returns them) has 2^d vertices (d <= 6); each is mapped through the closed loop
 * x+ = A x + B sat(-K x, +-umax) (A [d][d], B [d], K [d] host arrays; umax = HUGE_VAL for the unsaturated law), and
 * ok_out [nc] (device) receives 1 where no image leaves the box, else 0. */
int mpc_invariance_check(void *stream, int nc, int d, const double *Q_host, const double *eig_host, const double *A_host,
                         const double *B_host, const double *K_host, double umax, const double *c, int *ok_out);

/* The linear programmes of estimate_control_amissible_invariant_set (stability.py:26-78; cvxpy / ECOS there), batched:
 * programme q maximises c[q]' x subject to A x <= b, x free -- nlp objectives c [nlp][d] over one set of m rows A [m][d],
 * b [m] (all device memory; d <= 6, m <= MPC_LP_MAX_ROWS), one wavefront per programme (primal-dual interior point on the
 * d x d normal equations).  opt_out [nlp] receives the optima, x_out [nlp][d] (optional) the maximisers, status_out [nlp]
 * 0 = optimum (gap and residuals 1e-10 relative, plus the rounding floor of rows as large as the data's), 1 = unbounded,
 * 2 = iteration cap.  x = 0 need not be feasible. */
#define MPC_LP_MAX_ROWS 4096
int mpc_lp_maximize(void *stream, int nlp, int d, int m, const double *A, const double *b, const double *c, double *opt_out,
                    double *x_out, int *status_out);

/* Terminal set of the linear controllers (L, OFL, LPV):  C (x_N[1:] - x_ref) <= 1  on the last predicted state, over
 * (py, psi, vy, r, delta), x_ref = (y_ref, sin psi_ref, 0, 0, 0) of the last reference point (zero for the LPV controller,
 * like its terminal cost reference).  This is the constraint the reference prepares and leaves commented out
 * (LMPC2.py:212-218: con_h_expr_e = C (x[1:] - x_ref), uh_e = 1, lh_e = -1e7; C from
 * stability.py:93-112 reconstruct_control_amissible_invariant_set).  C_host: m rows of 5 doubles, host memory, copied;
 * m = 0 removes the set.  General rows are handled by the stage-wise solver: a handle created with MPC_LAYOUT_AUTO moves
 * to the lane-per-scenario layout with the first set it is given (and stays there; the fixed-model cache and
 * mpc_ofl_closed_loop_run belong to the wave layout and are then off), one created with MPC_LAYOUT_LANE is there already,
 * one created with MPC_LAYOUT_WAVE refuses the rows (MPC_EINVAL).  Synchronises the device, resizes the handle's workspace
 * and forgets the warm-start records.  A problem whose terminal set cannot be reached within the input bounds has no
 * minimiser: such solves end with status 2. */
int mpc_set_terminal_set(mpc_handle *h, int m, const double *C_host);

/* Forgets the interior-point warm-start records and the scenario order learnt from earlier launches: the next solve of
 * every scenario starts cold.  No counterpart in the reference (acados keeps its iterate inside the solver object and the
 * reference never resets it); for callers that reuse a handle for new scenarios. */
int mpc_reset_warm_start(mpc_handle *h, void *stream);

/* Sum / max statistics over the scenarios of the last mpc_solve_only / closed-loop call,
 * copied to a host array of MPC_STATS_LEN doubles after synchronising the stream:
 *   [0] scenarios, [1] sum iterations, [2] count status != 0, [3] count constrained solves,
 *   [4] sum |u0|, [5] max |u0|, [6] max final complementarity, [7] max final residual, [8] sum of SQP iterations (MPC_KIND_NL),
 *   [9] constrained solves finished by the active-set method, [10] sum of its iterations, [11] solves it handed to the
 *   interior point because the active set outgrew its capacity or was rank-deficient, [12] because its iteration did not
 *   settle or its result did not certify, [13] floating-point operations of the active-set solves (the method's own
 *   count); [14] largest entry of |H inv(H) - I| of the fixed-model cache's table (checked when mpc_condense_fixed builds
 *   it), [15] 1 when that exceeded 1e-9 and the active-set method was therefore left off for this cache (every
 *   constrained solve then goes to the interior point, which works from H itself), else 0.
 *   [1] counts interior-point iterations only.
 * Closed-loop statistics, over every scenario and every step of the last mpc_closed_loop_step / mpc_closed_loop_run /
 * mpc_ofl_closed_loop_run call (zero after mpc_solve_only): what the reference's experiments judge a run by
 * (skidpad_simulator.py:134-135 prints the heading pair's violation per step; plot_generator.py:1057-1072 measures the
 * tracking of the lateral reference; the cost is the controller's own LINEAR_LS stage cost, LMPC2.py:220-253):
 *   [16] sum of the stage cost at the start of the horizon with the applied input, dt * 1/2 |[x0; u0] - yref_0|^2_W
 *        (per-scenario weights where mpc_set_cost_weights gave them; the NL controller's W for MPC_KIND_NL),
 *   [17] sum of the squared lateral offset of the first reference point in the car frame (tracking error),
 *   [18] sum and [21] max of | |(cos, sin)| - 1 | of the plant's heading pair at the start of each step,
 *   [19] sum of the squared applied inputs,
 *   [20] max excess of |u0| over the rate bound or of the plant's steering angle (at the start of a step) over its
 *        bound (0: never exceeded),
 *   [22] max of | |(cos, sin)| - 1 | over the predicted trajectories (MPC_KIND_NL, where the pair is a model state; the
 *        linear controllers rebuild the pair from the angle: 0),
 *   [23] scenario-steps counted; rest reserved.
 * In a multi-GPU run [0]-[4], [8]-[13], [16]-[19], [23] add and [5]-[7], [20]-[22] take the maximum over the ranks. */
#define MPC_STATS_LEN 32
int mpc_get_stats(mpc_handle *h, void *stream, double *host_out);

#ifdef __cplusplus
}
#endif
#endif /* MPC_AMD_H */
