// Thread-per-scenario kernels for the pieces of the path that are called on their own through the
// reference's class surface: plant (continuous_dynamics.py), EKF (EKF.py), planners (utils/*.py),
// and the statistics reduction.  gfx950 only.
#include "mpc_device.hpp"
#include "mpc_kernels.h"

namespace mpc {

struct MP6 { double v[6]; };
struct V10 { double v[10]; };
struct V7 { double v[7]; };

static inline MP6 mp6_from_host(const double *p) { MP6 m; for (int i = 0; i < 6; i++) m.v[i] = p[i]; return m; }
__device__ __forceinline__ ModelParams to_mp(const MP6 &m) { return {m.v[0], m.v[1], m.v[2], m.v[3], m.v[4], m.v[5]}; }

// ------------------------------------------------------------------------------------------ plant
template <int NX>
__global__ __launch_bounds__(256) void k_plant_rk4(int B, double dt, MP6 m, double *x, const double *u)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B) return;
    double xs[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) xs[i] = x[(size_t)s * NX + i];
    plant_rk4<NX>(xs, u[s], dt, to_mp(m));
#pragma unroll
    for (int i = 0; i < NX; i++) x[(size_t)s * NX + i] = xs[i];
}

template <int NX>
__global__ __launch_bounds__(256) void k_plant_f(int B, MP6 m, const double *x, const double *u, double *xd)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B) return;
    double xs[NX], o[NX];
#pragma unroll
    for (int i = 0; i < NX; i++) xs[i] = x[(size_t)s * NX + i];
    plant_f<NX>(xs, u[s], to_mp(m), o);
#pragma unroll
    for (int i = 0; i < NX; i++) xd[(size_t)s * NX + i] = o[i];
}

static inline dim3 grid_for(int B) { return dim3((B + 255) / 256); }

hipError_t launch_plant_rk4(hipStream_t st, int B, int nx, double dt, const double *mp6, double *x, const double *u)
{
    if (nx == 8) hipLaunchKernelGGL(k_plant_rk4<8>, grid_for(B), dim3(256), 0, st, B, dt, mp6_from_host(mp6), x, u);
    else if (nx == 10) hipLaunchKernelGGL(k_plant_rk4<10>, grid_for(B), dim3(256), 0, st, B, dt, mp6_from_host(mp6), x, u);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t launch_plant_f(hipStream_t st, int B, int nx, const double *mp6, const double *x, const double *u, double *xd)
{
    if (nx == 8) hipLaunchKernelGGL(k_plant_f<8>, grid_for(B), dim3(256), 0, st, B, mp6_from_host(mp6), x, u, xd);
    else if (nx == 10) hipLaunchKernelGGL(k_plant_f<10>, grid_for(B), dim3(256), 0, st, B, mp6_from_host(mp6), x, u, xd);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------- EKF
__device__ __forceinline__ void ekf_jacobian(const double *x, double dt, const ModelParams &mp, double *F)
{
#pragma unroll
    for (int i = 0; i < 100; i++) F[i] = (i / 10 == i % 10) ? 1.0 : 0.0;
    ekf_jacobian_entries(x, dt, mp, F);
}

__global__ __launch_bounds__(64) void k_ekf_jacobian(int B, double dt, MP6 m, const double *x, double *Fout)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B) return;
    double xs[10], F[100];
#pragma unroll
    for (int i = 0; i < 10; i++) xs[i] = x[(size_t)s * 10 + i];
    ekf_jacobian(xs, dt, to_mp(m), F);
    for (int i = 0; i < 100; i++) Fout[(size_t)s * 100 + i] = F[i];
}

// EKF.py:38-43: x <- x + dt f(x,u) (forward Euler), F at the pre-update estimate, P <- F P F' + Q.
// F is sparse (rows 0-3, 5, 6 carry off-diagonal terms); the products below only visit its pattern.
__device__ __forceinline__ void ekf_time_update_one(double dt, const ModelParams &mp, const V10 &qd, double *xg, double *Pg,
                                                    double u)
{
    double xs[10], xd[10], F[100];
#pragma unroll
    for (int i = 0; i < 10; i++) xs[i] = xg[i];
    plant_f<10>(xs, u, mp, xd);
    ekf_jacobian(xs, dt, mp, F);
#pragma unroll
    for (int i = 0; i < 10; i++) xg[i] = xd[i] * dt + xs[i];
    double T[100];
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 10; j++) {
            double acc = 0.0;
            for (int k = 0; k < 10; k++) acc += F[i * 10 + k] * Pg[k * 10 + j];
            T[i * 10 + j] = acc;
        }
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 10; j++) {
            double acc = 0.0;
            for (int k = 0; k < 10; k++) acc += T[i * 10 + k] * F[j * 10 + k];
            Pg[i * 10 + j] = acc + (i == j ? qd.v[i] : 0.0);
        }
}

__global__ __launch_bounds__(64) void k_ekf_time_update(int B, double dt, MP6 m, V10 qd, double *x, double *P,
                                                        const double *u)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B) return;
    ekf_time_update_one(dt, to_mp(m), qd, x + (size_t)s * 10, P + (size_t)s * 100, u[s]);
}

__constant__ int c_meas_idx[7] = {0, 1, 2, 3, 4, 6, 7};   // continuous_dynamics.py:35-45

// EKF.py:45-49,65-74: K = solve(S', (P H')')' with S = H P H' + R, x += K (y - H x), P -= K H P.
// The 7x7 system is solved by Gaussian elimination with partial pivoting (what numpy.linalg.solve does).
__device__ __forceinline__ void ekf_meas_update_one(const V7 &rd, double *xg, double *Pg, const double *y, double *K_out,
                                                    int gain_only)
{
    double M[49], X[70];
    for (int a = 0; a < 7; a++)
        for (int b = 0; b < 7; b++)   // M = S'
            M[a * 7 + b] = Pg[c_meas_idx[b] * 10 + c_meas_idx[a]] + (a == b ? rd.v[a] : 0.0);
    for (int a = 0; a < 7; a++)
        for (int i = 0; i < 10; i++) X[a * 10 + i] = Pg[i * 10 + c_meas_idx[a]];
    for (int c = 0; c < 7; c++) {
        int piv = c;
        for (int r = c + 1; r < 7; r++)
            if (fabs(M[r * 7 + c]) > fabs(M[piv * 7 + c])) piv = r;
        if (piv != c) {
            for (int j = 0; j < 7; j++) { double t = M[c * 7 + j]; M[c * 7 + j] = M[piv * 7 + j]; M[piv * 7 + j] = t; }
            for (int j = 0; j < 10; j++) { double t = X[c * 10 + j]; X[c * 10 + j] = X[piv * 10 + j]; X[piv * 10 + j] = t; }
        }
        for (int r = c + 1; r < 7; r++) {
            const double f = M[r * 7 + c] / M[c * 7 + c];
            for (int j = c; j < 7; j++) M[r * 7 + j] -= f * M[c * 7 + j];
            for (int j = 0; j < 10; j++) X[r * 10 + j] -= f * X[c * 10 + j];
        }
    }
    for (int c = 6; c >= 0; c--)
        for (int j = 0; j < 10; j++) {
            double acc = X[c * 10 + j];
            for (int k = c + 1; k < 7; k++) acc -= M[c * 7 + k] * X[k * 10 + j];
            X[c * 10 + j] = acc / M[c * 7 + c];
        }
    // K[i][a] = X[a][i]
    if (K_out)
        for (int i = 0; i < 10; i++)
            for (int a = 0; a < 7; a++) K_out[i * 7 + a] = X[a * 10 + i];
    if (gain_only) return;
    double innov[7];
    for (int a = 0; a < 7; a++) innov[a] = y[a] - xg[c_meas_idx[a]];
    for (int i = 0; i < 10; i++) {
        double acc = 0.0;
        for (int a = 0; a < 7; a++) acc += X[a * 10 + i] * innov[a];
        xg[i] += acc;
    }
    double HP[70];
    for (int a = 0; a < 7; a++)
        for (int j = 0; j < 10; j++) HP[a * 10 + j] = Pg[c_meas_idx[a] * 10 + j];
    for (int i = 0; i < 10; i++)
        for (int j = 0; j < 10; j++) {
            double acc = 0.0;
            for (int a = 0; a < 7; a++) acc += X[a * 10 + i] * HP[a * 10 + j];
            Pg[i * 10 + j] -= acc;
        }
}

__global__ __launch_bounds__(64) void k_ekf_meas_update(int B, V7 rd, double *x, double *P, const double *y, double *K_out,
                                                        int gain_only)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B) return;
    ekf_meas_update_one(rd, x ? x + (size_t)s * 10 : nullptr, P + (size_t)s * 100, y ? y + (size_t)s * 7 : nullptr,
                        K_out ? K_out + (size_t)s * 70 : nullptr, gain_only);
}

static inline V10 v10_from_host(const double *p) { V10 m; for (int i = 0; i < 10; i++) m.v[i] = p[i]; return m; }
static inline V7 v7_from_host(const double *p) { V7 m; for (int i = 0; i < 7; i++) m.v[i] = p[i]; return m; }

hipError_t launch_ekf_time_update(hipStream_t st, int B, double dt, const double *mp6, const double *qd, double *x,
                                  double *P, const double *u)
{
    hipLaunchKernelGGL(k_ekf_time_update, dim3((B + 63) / 64), dim3(64), 0, st, B, dt, mp6_from_host(mp6),
                       v10_from_host(qd), x, P, u);
    return hipGetLastError();
}

hipError_t launch_ekf_meas_update(hipStream_t st, int B, const double *rd, double *x, double *P, const double *y,
                                  double *K_out, int gain_only)
{
    hipLaunchKernelGGL(k_ekf_meas_update, dim3((B + 63) / 64), dim3(64), 0, st, B, v7_from_host(rd), x, P, y, K_out,
                       gain_only);
    return hipGetLastError();
}

hipError_t launch_ekf_jacobian(hipStream_t st, int B, double dt, const double *mp6, const double *x, double *F)
{
    hipLaunchKernelGGL(k_ekf_jacobian, dim3((B + 63) / 64), dim3(64), 0, st, B, dt, mp6_from_host(mp6), x, F);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------- baselines
struct V5 { double v[5]; };

// StepSimulator.dlqr_sim (simulator.py:378-428): nsteps of u = K (ref - x[[1,3,5,6,7]]) with ref = (1/500, 0, 0, 0, 0),
// clipped to the rate limit and then to what keeps the LAST state entry inside the angle limit over one dt (the
// reference reads full_state[-1]: the steering angle of the 8-state plant), followed by the RK4 plant step.
template <int NX>
__global__ __launch_bounds__(64) void k_dlqr_run(int B, int nsteps, double dt, MP6 m, V5 K, const double *K_scen,
                                                 double rate_limit, double angle_limit, double *x, double *u_hist,
                                                 double *x_hist)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B) return;
    const ModelParams mp = to_mp(m);
    const PlantCoef pc = plant_coef(mp, x[(size_t)s * NX + 4]);
    double k[5], xs[NX];
#pragma unroll
    for (int i = 0; i < 5; i++) k[i] = K_scen ? K_scen[(size_t)s * 5 + i] : K.v[i];
#pragma unroll
    for (int i = 0; i < NX; i++) xs[i] = x[(size_t)s * NX + i];
    for (int t = 0; t < nsteps; t++) {
        double steer = k[0] * (1.0 / 500.0 - xs[1]) + k[1] * (0.0 - xs[3]) + k[2] * (0.0 - xs[5]) + k[3] * (0.0 - xs[6])
                       + k[4] * (0.0 - xs[7]);
        steer = fmin(fmax(steer, -rate_limit), rate_limit);
        const double cur = xs[NX - 1];
        steer = fmin(fmax(steer, (-angle_limit - cur) / dt), (angle_limit - cur) / dt);
        plant_rk4_c<NX>(xs, steer, dt, pc);
        if (u_hist) u_hist[(size_t)t * B + s] = steer;
        if (x_hist)
#pragma unroll
            for (int i = 0; i < NX; i++) x_hist[((size_t)t * B + s) * NX + i] = xs[i];
    }
#pragma unroll
    for (int i = 0; i < NX; i++) x[(size_t)s * NX + i] = xs[i];
}

hipError_t launch_dlqr_run(hipStream_t st, int B, int nx, int nsteps, double dt, const double *mp6, const double *K5,
                           const double *K_scen, double rate_limit, double angle_limit, double *x, double *u_hist,
                           double *x_hist)
{
    V5 K; for (int i = 0; i < 5; i++) K.v[i] = K5 ? K5[i] : 0.0;
    if (nx == 8)
        hipLaunchKernelGGL(k_dlqr_run<8>, dim3((B + 63) / 64), dim3(64), 0, st, B, nsteps, dt, mp6_from_host(mp6), K, K_scen,
                           rate_limit, angle_limit, x, u_hist, x_hist);
    else if (nx == 10)
        hipLaunchKernelGGL(k_dlqr_run<10>, dim3((B + 63) / 64), dim3(64), 0, st, B, nsteps, dt, mp6_from_host(mp6), K, K_scen,
                           rate_limit, angle_limit, x, u_hist, x_hist);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// StepSimulator.lsim (simulator.py:334-376): an open-loop input sequence through the 10-state plant and the EKF
// (time update with the same input, measurement update on the noiseless measurement of the new plant state, plus
// meas_noise when given).  u is [nsteps] shared by all scenarios, or [nsteps][B] when u_per_scenario is set.
__global__ __launch_bounds__(64) void k_lsim_run(int B, int nsteps, double dt, MP6 m, V10 qd, V7 rd, const double *u,
                                                 int u_per_scenario, const double *meas_noise, double *x, double *ekf_x,
                                                 double *ekf_P, double *x_hist, double *est_hist)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B) return;
    const ModelParams mp = to_mp(m);
    const PlantCoef pc = plant_coef(mp, x[(size_t)s * 10 + 4]);
    double xs[10], y[7];
#pragma unroll
    for (int i = 0; i < 10; i++) xs[i] = x[(size_t)s * 10 + i];
    double *xe = ekf_x + (size_t)s * 10, *Pe = ekf_P + (size_t)s * 100;
    for (int t = 0; t < nsteps; t++) {
        const double ut = u_per_scenario ? u[(size_t)t * B + s] : u[t];
        plant_rk4_c<10>(xs, ut, dt, pc);
        ekf_time_update_one(dt, mp, qd, xe, Pe, ut);
#pragma unroll
        for (int a = 0; a < 7; a++) {
            const int idx = a < 5 ? a : a + 1;                       // measured entries 0, 1, 2, 3, 4, 6, 7
            y[a] = xs[idx] + (meas_noise ? meas_noise[((size_t)t * B + s) * 7 + a] : 0.0);
        }
        ekf_meas_update_one(rd, xe, Pe, y, nullptr, 0);
        if (x_hist)
#pragma unroll
            for (int i = 0; i < 10; i++) x_hist[((size_t)t * B + s) * 10 + i] = xs[i];
        if (est_hist)
            for (int i = 0; i < 10; i++) est_hist[((size_t)t * B + s) * 10 + i] = xe[i];
    }
#pragma unroll
    for (int i = 0; i < 10; i++) x[(size_t)s * 10 + i] = xs[i];
}

hipError_t launch_lsim_run(hipStream_t st, int B, int nsteps, double dt, const double *mp6, const double *qd,
                           const double *rd, const double *u, int u_per_scenario, const double *meas_noise, double *x,
                           double *ekf_x, double *ekf_P, double *x_hist, double *est_hist)
{
    hipLaunchKernelGGL(k_lsim_run, dim3((B + 63) / 64), dim3(64), 0, st, B, nsteps, dt, mp6_from_host(mp6),
                       v10_from_host(qd), v7_from_host(rd), u, u_per_scenario, meas_noise, x, ekf_x, ekf_P, x_hist, est_hist);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ planners
// One wavefront per scenario, lane k -> waypoint k (and k + 64, ...).  With `zones` ([B][7]: the six zone edges and the
// slow-down factor) a skidpad request for a lap outside 0..7 follows the planner's speed profile instead of the
// constant target speed (utils/path_planning.py:238-248): that recurrence is sequential, so lane 0 walks it and parks
// the progresses in the x slot of wp_out, from where the lanes turn them into poses.
__global__ __launch_bounds__(WAVE) void k_plan(int mode, int N, double dt, double ramp, const double *pose, const int *lap,
                                               double *prev_progress, const double *target_vel, const double *zones,
                                               double *wp_out, double *progress_out, double *abs_out, double *speeds_out)
{
    const int s = blockIdx.x, lane = threadIdx.x;
    const double x = pose[3 * (size_t)s], y = pose[3 * (size_t)s + 1], heading = pose[3 * (size_t)s + 2];
    const double v = target_vel[s];
    double p0;
    bool profile = false;
    if (mode == MODE_SKIDPAD) {
        int lp = lap[s];
        p0 = sk_pos2progress(x, y, lp);
        if (prev_progress[s] > p0 + 2 * SK_R) { lp += 1; p0 = sk_pos2progress(x, y, lp); }   // path_planning.py:227-230
        profile = zones != nullptr && (lp < 0 || lp > 7);
    } else
        p0 = x;
    double *wps = wp_out + (size_t)s * (N + 1) * 4;
    if (profile) {
        if (lane == 0) {
            const double *zg = zones + 7 * (size_t)s;
            const SpeedZones z = {zg[0], zg[1], zg[2], zg[3], zg[4], zg[5], zg[6]};
            double q = p0;
            for (int i = 0; i < N; i++) {
                const double sp = sk_stage_speed(z, v, q, dt);
                wps[4 * i] = q;
                if (speeds_out) speeds_out[(size_t)s * (N + 1) + i] = sp;
                q = q + sp * dt;
            }
            wps[4 * N] = q;
            if (speeds_out) speeds_out[(size_t)s * (N + 1) + N] = 0.0;      // the last entry keeps its initial zero
        }
        __syncthreads();
    }
    const double ch = cos(heading), sh = sin(heading);
    for (int k = lane; k <= N; k += WAVE) {
        double o[4];
        const double q = profile ? wps[4 * k] : waypoint_progress(p0, v, dt, k);
        if (mode == MODE_SKIDPAD) sk_pose(q, o);
        else step_pose(q, ramp, o);
        if (abs_out) {
            double *w = abs_out + ((size_t)s * (N + 1) + k) * 4;
            w[0] = o[0]; w[1] = o[1]; w[2] = o[2]; w[3] = o[3];
        }
        to_car_frame(o, x, y, ch, sh);
        double *w = wps + (size_t)k * 4;
        w[0] = o[0]; w[1] = o[1]; w[2] = o[2]; w[3] = o[3];
        if (speeds_out && !profile) speeds_out[(size_t)s * (N + 1) + k] = v;
    }
    __syncthreads();
    if (lane == 0) {
        if (progress_out) progress_out[s] = p0;
        if (mode == MODE_SKIDPAD) prev_progress[s] = p0;                                       // path_planning.py:251
    }
}

hipError_t launch_plan_skidpad(hipStream_t st, int B, int N, double dt, const double *pose, const int *lap,
                               double *prev_progress, const double *target_vel, const double *zones, double *wp_out,
                               double *progress_out, double *speeds_out)
{
    hipLaunchKernelGGL(k_plan, dim3(B), dim3(WAVE), 0, st, (int)MODE_SKIDPAD, N, dt, 0.0, pose, lap, prev_progress,
                       target_vel, zones, wp_out, progress_out, (double *)nullptr, speeds_out);
    return hipGetLastError();
}

hipError_t launch_plan_step(hipStream_t st, int B, int N, double dt, double ramp, const double *pose, const double *target_vel,
                            double *wp_out, double *progress_out, double *abs_out)
{
    hipLaunchKernelGGL(k_plan, dim3(B), dim3(WAVE), 0, st, (int)MODE_STEP, N, dt, ramp, pose, (const int *)nullptr,
                       (double *)nullptr, target_vel, (const double *)nullptr, wp_out, progress_out, abs_out,
                       (double *)nullptr);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------- control buffer
// archive/mpc_node.py:421-427: between two solves the vehicle node steers with the predicted steering-angle trajectory
// (column `col` of the solver's state trajectory) interpolated at the time since the solve, np.interp over the node
// times linspace(0, Tf, N + 1), times `scale` (pi / 2 / 0.4 there).  np.interp clamps outside the table and evaluates
// slope * (t - t_j) + f_j inside it.
__global__ __launch_bounds__(256) void k_control_interp(int B, int N, double Tf, const double *traj, int ncol, int col,
                                                        const double *t, double scale, double *out)
{
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= B) return;
    const double *f = traj + (size_t)s * (N + 1) * ncol + col;
    const double step = Tf / N, x = t[s];
    double r;
    if (x != x) r = x;
    else if (x <= 0.0) r = f[0];                         // left of (or on) the first node
    else if (x >= Tf) r = f[(size_t)N * ncol];
    else {
        int j = (int)(x / step);
        if (j > N - 1) j = N - 1;
        auto node = [&](int i) { return i == N ? Tf : i * step; };      // linspace: i * step, the end point exact
        while (j > 0 && node(j) > x) j--;
        while (j < N - 1 && node(j + 1) <= x) j++;
        const double f0 = f[(size_t)j * ncol], f1 = f[(size_t)(j + 1) * ncol];
        const double slope = (f1 - f0) / (node(j + 1) - node(j));
        r = slope * (x - node(j)) + f0;
    }
    out[s] = r * scale;
}

hipError_t launch_control_interp(hipStream_t st, int B, int N, double Tf, const double *traj, int ncol, int col,
                                 const double *t, double scale, double *out)
{
    hipLaunchKernelGGL(k_control_interp, dim3((B + 255) / 256), dim3(256), 0, st, B, N, Tf, traj, ncol, col, t, scale, out);
    return hipGetLastError();
}

// ----------------------------------------------------------------------------------- load balance
// Counting sort of the scenarios by the work of their last launch, heaviest first: order[rank] = scenario.  One
// workgroup; keys are work >> shift clamped to 1023.  Scenarios are independent, so the order among equal keys
// (decided by the LDS atomics) changes no result, only which wave advances which scenario.
__global__ __launch_bounds__(1024) void k_balance_order(int B, const int *work, int shift, int *order)
{
    __shared__ int hist[1024];
    __shared__ int part[1024];
    const int t = threadIdx.x;
    hist[t] = 0;
    __syncthreads();
    for (int s = t; s < B; s += 1024) {
        const int key = min(max(work[s], 0) >> shift, 1023);
        atomicAdd(&hist[1023 - key], 1);                 // bin 0 = heaviest
    }
    __syncthreads();
    // exclusive prefix over the 1024 bins (Hillis-Steele in LDS)
    part[t] = hist[t];
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int v = (t >= d) ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    hist[t] = part[t] - hist[t];                         // start offset of bin t
    __syncthreads();
    for (int s = t; s < B; s += 1024) {
        const int key = min(max(work[s], 0) >> shift, 1023);
        const int pos = atomicAdd(&hist[1023 - key], 1);
        order[pos] = s;
    }
}

hipError_t launch_balance_order(hipStream_t st, int B, const int *work, int shift, int *order)
{
    hipLaunchKernelGGL(k_balance_order, dim3(1), dim3(1024), 0, st, B, work, shift, order);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------- invariance
// stability.py:155-189 (check_control_amissible_invariance) for a batch of level values c: the ellipsoid x' P x <= c is
// boxed in the eigenbasis of P (bounds sqrt(c / eig_i)); every vertex of that box is mapped through the closed loop
// x+ = A x + B sat(-K x, umax) and back into the eigenbasis, and the level passes when no image leaves the box.
// One wavefront per level, one lane per vertex (d <= 6).  Q holds the eigenvectors as columns (numpy.linalg.eigh).
struct InvArgs { double Q[36], eig[6], A[36], B[6], K[6]; double umax; int d; };

__global__ __launch_bounds__(WAVE) void k_invariance(InvArgs p, int nc, const double *c, int *ok_out)
{
    const int lane = threadIdx.x, d = p.d;
    const int ci = blockIdx.x;
    if (ci >= nc) return;
    const double cv = c[ci];
    bool bad = false;
    if (lane < (1 << d)) {
        double bnd[6], vtx[6], x[6], xn[6];
        for (int i = 0; i < d; i++) { bnd[i] = sqrt(cv / p.eig[i]); vtx[i] = ((lane >> i) & 1) ? bnd[i] : -bnd[i]; }
        double u = 0.0;
        for (int i = 0; i < d; i++) {
            double acc = 0.0;
            for (int j = 0; j < d; j++) acc += p.Q[i * d + j] * vtx[j];          // x = Q v
            x[i] = acc;
        }
        for (int j = 0; j < d; j++) u -= p.K[j] * x[j];
        u = fmin(fmax(u, -p.umax), p.umax);
        for (int i = 0; i < d; i++) {
            double acc = p.B[i] * u;
            for (int j = 0; j < d; j++) acc += p.A[i * d + j] * x[j];
            xn[i] = acc;
        }
        for (int i = 0; i < d; i++) {
            double acc = 0.0;
            for (int j = 0; j < d; j++) acc += p.Q[j * d + i] * xn[j];           // v+ = Q' x+
            bad = bad || (fabs(acc) > bnd[i]);
        }
    }
    const bool any_bad = __any(bad);
    if (lane == 0) ok_out[ci] = any_bad ? 0 : 1;
}

hipError_t launch_invariance(hipStream_t st, int nc, int d, const double *Q, const double *eig, const double *A,
                             const double *B, const double *K, double umax, const double *c, int *ok_out)
{
    InvArgs p;
    for (int i = 0; i < d * d; i++) { p.Q[i] = Q[i]; p.A[i] = A[i]; }
    for (int i = 0; i < d; i++) { p.eig[i] = eig[i]; p.B[i] = B[i]; p.K[i] = K[i]; }
    p.umax = umax; p.d = d;
    hipLaunchKernelGGL(k_invariance, dim3(nc), dim3(WAVE), 0, st, p, nc, c, ok_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------- linear programmes
// maximise c'x subject to A x <= b (x free, d <= 6 unknowns, m rows), one wavefront per programme, all programmes over the
// same rows: the loop of estimate_control_amissible_invariant_set (stability.py:26-78) asks 2 (nx + nu) such questions
// about the set it has built so far, per power of the closed-loop matrix (cvxpy / ECOS there).  Mehrotra
// predictor-corrector on  A x + s = b,  A'z = c,  s, z >= 0,  s_i z_i = 0:  the Newton step reduces to the d x d system
// (A' diag(z / s) A) dx = rhs, accumulated row by row (lane l owns rows l, l + 64, ...) and reduced over the wave; every
// lane then factors and solves it for itself.  Slacks, multipliers and their affine directions live in LDS (4 m doubles).
// status: 0 optimum (relative gap and residuals <= 1e-10), 1 unbounded (the iterates run off), 2 iteration cap.
constexpr int LP_D = 6;

__device__ __forceinline__ void lp_solve_spd(const double (&M)[21], const double (&r)[LP_D], int d, double (&x)[LP_D])
{
    // Cholesky of the packed lower triangle (entries beyond d: unit diagonal), then the two substitutions
    double L[LP_D][LP_D];
#pragma unroll
    for (int i = 0; i < LP_D; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) L[i][j] = (i < d && j < d) ? M[i * (i + 1) / 2 + j] : (i == j ? 1.0 : 0.0);
#pragma unroll
    for (int j = 0; j < LP_D; j++) {
        double dj = L[j][j];
#pragma unroll
        for (int k = 0; k < j; k++) dj -= L[j][k] * L[j][k];
        dj = sqrt(fmax(dj, 1e-300));
        L[j][j] = dj;
#pragma unroll
        for (int i = j + 1; i < LP_D; i++) {
            double v = L[i][j];
#pragma unroll
            for (int k = 0; k < j; k++) v -= L[i][k] * L[j][k];
            L[i][j] = v / dj;
        }
    }
    double y[LP_D];
#pragma unroll
    for (int i = 0; i < LP_D; i++) {
        double v = i < d ? r[i] : 0.0;
#pragma unroll
        for (int k = 0; k < i; k++) v -= L[i][k] * y[k];
        y[i] = v / L[i][i];
    }
#pragma unroll
    for (int i = LP_D - 1; i >= 0; i--) {
        double v = y[i];
#pragma unroll
        for (int k = i + 1; k < LP_D; k++) v -= L[k][i] * x[k];
        x[i] = v / L[i][i];
    }
}

__global__ __launch_bounds__(WAVE) void k_lp_maximize(int nlp, int d, int m, const double *A, const double *b, const double *c,
                                                      double *opt_out, double *x_out, int *status_out)
{
    extern __shared__ double lp_lds[];
    double *S = lp_lds, *Z = lp_lds + m, *DS = lp_lds + 2 * (size_t)m, *DZ = lp_lds + 3 * (size_t)m;
    const int lane = threadIdx.x, q = blockIdx.x;
    if (q >= nlp) return;
    double cv[LP_D], x[LP_D];
#pragma unroll
    for (int j = 0; j < LP_D; j++) { cv[j] = j < d ? c[(size_t)q * d + j] : 0.0; x[j] = 0.0; }
    double bscale = 0.0;
    for (int i = lane; i < m; i += WAVE) { S[i] = fmax(b[i], 1.0); Z[i] = 1.0; bscale = fmax(bscale, fabs(b[i])); }
    bscale = fmax(1.0, wave_max(bscale));
    double cscale = 0.0;
#pragma unroll
    for (int j = 0; j < LP_D; j++) cscale = fmax(cscale, fabs(cv[j]));
    cscale = fmax(cscale, 1e-300);
    int status = 2;
#pragma unroll 1
    for (int it = 0; it < 80; it++) {
        // pass 1: residuals, normal matrix, right-hand side of the affine step
        double M[21], rd[LP_D], v[LP_D];
#pragma unroll
        for (int e = 0; e < 21; e++) M[e] = 0.0;
#pragma unroll
        for (int j = 0; j < LP_D; j++) { rd[j] = 0.0; v[j] = 0.0; }
        double gap = 0.0, rpmax = 0.0, amax = 0.0, zsum = 0.0;
        for (int i = lane; i < m; i += WAVE) {
            double a[LP_D], ax = 0.0;
#pragma unroll
            for (int j = 0; j < LP_D; j++) { a[j] = j < d ? A[(size_t)i * d + j] : 0.0; ax += a[j] * x[j]; amax = fmax(amax, fabs(a[j])); }
            const double s = S[i], z = Z[i], rp = ax + s - b[i], w = z / s;
            gap += s * z; rpmax = fmax(rpmax, fabs(rp)); zsum += z;
            const double t = (s * z - z * rp) / s;               // S^-1 (r_c - Z r_p), r_c = s z
#pragma unroll
            for (int j = 0; j < LP_D; j++) {
                rd[j] += a[j] * z; v[j] += a[j] * t;
#pragma unroll
                for (int k = 0; k <= j; k++) M[j * (j + 1) / 2 + k] += w * a[j] * a[k];
            }
        }
#pragma unroll
        for (int e = 0; e < 21; e++) M[e] = wave_sum(M[e]);
        double rdmax = 0.0, obj = 0.0, xmax = 0.0;
#pragma unroll
        for (int j = 0; j < LP_D; j++) {
            rd[j] = wave_sum(rd[j]) - cv[j]; v[j] = wave_sum(v[j]);
            rdmax = fmax(rdmax, fabs(rd[j])); obj += cv[j] * x[j]; xmax = fmax(xmax, fabs(x[j]));
        }
        gap = wave_sum(gap); rpmax = wave_max(rpmax); amax = wave_max(amax); zsum = wave_sum(zsum);
        const double mu = gap / m;
        // (the residuals cannot fall below the rounding of the sums they are: rows of size amax against x and z)
        if (gap <= 1e-10 * (1.0 + fabs(obj)) && rpmax <= 1e-10 * bscale + 1e-13 * amax * xmax &&
            rdmax <= 1e-10 * cscale + 1e-13 * amax * zsum) { status = 0; break; }
        if (!(xmax < 1e12 * bscale) || !(fabs(obj) < 1e12 * bscale * cscale)) { status = 1; break; }
        // a whiff of regularisation keeps the factorisation alive when multipliers vanish on all rows that see a direction
        double tr = 0.0;
#pragma unroll
        for (int j = 0; j < LP_D; j++) tr += M[j * (j + 1) / 2 + j];
#pragma unroll
        for (int j = 0; j < LP_D; j++) M[j * (j + 1) / 2 + j] += 1e-15 * tr + 1e-300;
        double rhs[LP_D], dx[LP_D];
#pragma unroll
        for (int j = 0; j < LP_D; j++) rhs[j] = -rd[j] + v[j];
        lp_solve_spd(M, rhs, d, dx);
        // pass 2: affine directions, step to the boundary, complementarity after it
        double al = 1.0;
        for (int i = lane; i < m; i += WAVE) {
            double adx = 0.0, ax = 0.0;
#pragma unroll
            for (int j = 0; j < LP_D; j++) { const double aj = j < d ? A[(size_t)i * d + j] : 0.0; adx += aj * dx[j]; ax += aj * x[j]; }
            const double s = S[i], z = Z[i], rp = ax + s - b[i];
            const double ds = -rp - adx, dz = (-s * z - z * ds) / s;
            DS[i] = ds; DZ[i] = dz;
            if (ds < 0.0) al = fmin(al, -s / ds);
            if (dz < 0.0) al = fmin(al, -z / dz);
        }
        al = wave_min(al);
        double gaff = 0.0;
        for (int i = lane; i < m; i += WAVE) gaff += (S[i] + al * DS[i]) * (Z[i] + al * DZ[i]);
        gaff = wave_sum(gaff);
        const double sg = gaff / gap, sigmu = sg * sg * sg * mu;
        // pass 3: corrector right-hand side
#pragma unroll
        for (int j = 0; j < LP_D; j++) v[j] = 0.0;
        for (int i = lane; i < m; i += WAVE) {
            double a[LP_D], ax = 0.0;
#pragma unroll
            for (int j = 0; j < LP_D; j++) { a[j] = j < d ? A[(size_t)i * d + j] : 0.0; ax += a[j] * x[j]; }
            const double s = S[i], z = Z[i], rp = ax + s - b[i];
            const double t = (s * z + DS[i] * DZ[i] - sigmu - z * rp) / s;
#pragma unroll
            for (int j = 0; j < LP_D; j++) v[j] += a[j] * t;
        }
#pragma unroll
        for (int j = 0; j < LP_D; j++) rhs[j] = -rd[j] + wave_sum(v[j]);
        lp_solve_spd(M, rhs, d, dx);
        // pass 4: directions and the step
        al = 1.0;
        for (int i = lane; i < m; i += WAVE) {
            double adx = 0.0, ax = 0.0;
#pragma unroll
            for (int j = 0; j < LP_D; j++) { const double aj = j < d ? A[(size_t)i * d + j] : 0.0; adx += aj * dx[j]; ax += aj * x[j]; }
            const double s = S[i], z = Z[i], rp = ax + s - b[i];
            const double ds = -rp - adx, dz = (-(s * z + DS[i] * DZ[i] - sigmu) - z * ds) / s;
            DS[i] = ds; DZ[i] = dz;
            if (ds < 0.0) al = fmin(al, -s / ds);
            if (dz < 0.0) al = fmin(al, -z / dz);
        }
        al = fmin(1.0, 0.995 * wave_min(al));
        for (int i = lane; i < m; i += WAVE) { S[i] += al * DS[i]; Z[i] += al * DZ[i]; }
#pragma unroll
        for (int j = 0; j < LP_D; j++) x[j] += al * dx[j];       // (every row is always visited by the same lane: no hand-off through LDS)
    }
    if (lane == 0) {
        double obj = 0.0;
#pragma unroll
        for (int j = 0; j < LP_D; j++) obj += cv[j] * x[j];
        opt_out[q] = obj; status_out[q] = status;
    }
    if (x_out && lane < d) {
        double xv = x[0];
#pragma unroll
        for (int j = 1; j < LP_D; j++) xv = lane == j ? x[j] : xv;
        x_out[(size_t)q * d + lane] = xv;
    }
}

hipError_t launch_lp_maximize(hipStream_t st, int nlp, int d, int m, const double *A, const double *b, const double *c,
                              double *opt_out, double *x_out, int *status_out)
{
    const size_t lds = sizeof(double) * 4 * (size_t)m;
    static bool lds_ok[MPC_MAX_DEVICES] = {};
    if (hipError_t e = allow_big_lds(reinterpret_cast<const void *>(&k_lp_maximize), lds_ok)) return e;
    hipLaunchKernelGGL(k_lp_maximize, dim3(nlp), dim3(WAVE), lds, st, nlp, d, m, A, b, c, opt_out, x_out, status_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------ statistics
// Reduction of the per-scenario records into MPC_STATS_LEN doubles (sums and maxima; the slots are listed at
// mpc_get_stats in mpc_amd.h).  Two launches: up to STATS_BLOCKS workgroups write a partial vector each, one workgroup
// combines them in a fixed order, so the sums do not depend on the timing of atomics.
constexpr int ST_N = MPC_STATS_LEN;
__device__ __forceinline__ bool stat_is_max(int c) { return (c >= 5 && c <= 7) || (c >= 20 && c <= 22); }

// diag of W as the stage cost of the closed-loop statistics needs it: per scenario from cost_scen, else the controller's
struct StatWeights { int nl; double half_dt, umax, dmax; double w[8]; };

__global__ __launch_bounds__(256) void k_stats_partial(int B, const double *stat, const int *iters, const int *status,
                                                        const int *sqp, const int *as2, bool with_cl, StatWeights sw,
                                                        const double *cost_scen, double *part)
{
    __shared__ double sh[ST_N][256 + 1];
    double acc[ST_N];
#pragma unroll
    for (int c = 0; c < ST_N; c++) acc[c] = 0.0;
    for (int s = blockIdx.x * 256 + threadIdx.x; s < B; s += gridDim.x * 256) {
        const double *q = stat + STAT_W * (size_t)s;
        acc[0] += 1.0; acc[1] += iters[s]; acc[2] += (status[s] != 0); acc[3] += q[2]; acc[4] += q[3];
        acc[5] = fmax(acc[5], q[3]);
        if (q[2] != 0.0) { acc[6] = fmax(acc[6], q[0]); acc[7] = fmax(acc[7], q[1]); }
        if (sqp) acc[8] += sqp[s];
        if (as2) { acc[9] += as2[6 * s]; acc[10] += as2[6 * s + 1]; acc[11] += as2[6 * s + 2]; acc[12] += as2[6 * s + 3]; acc[13] += 64.0 * as2[6 * s + 4]; }
        if (with_cl) {
            const double *r = q + STAT_CL;
            double cost = 0.0;                            // sum over the steps of dt/2 |[x0; u0] - yref_0|^2_W
#pragma unroll
            for (int c = 0; c < 8; c++) cost += ((cost_scen && c < 7) ? cost_scen[43 * (size_t)s + c] : sw.w[c]) * r[CL_ERR + c];
            acc[16] += sw.half_dt * cost; acc[17] += r[CL_ERR + 1]; acc[18] += r[CL_TRIG_SUM]; acc[19] += r[CL_ERR + (sw.nl ? 7 : 6)];
            acc[20] = fmax(acc[20], fmax(r[CL_U_MAX] - sw.umax, r[CL_STEER_MAX] - sw.dmax)); acc[21] = fmax(acc[21], r[CL_TRIG_MAX]); acc[22] = fmax(acc[22], r[CL_PRED_TRIG_MAX]);
            acc[23] += r[CL_STEPS];
        }
    }
#pragma unroll
    for (int c = 0; c < ST_N; c++) sh[c][threadIdx.x] = acc[c];
    __syncthreads();
    for (int w = 128; w; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int c = 0; c < ST_N; c++)
                sh[c][threadIdx.x] = stat_is_max(c) ? fmax(sh[c][threadIdx.x], sh[c][threadIdx.x + w]) : sh[c][threadIdx.x] + sh[c][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x < ST_N) part[blockIdx.x * ST_N + threadIdx.x] = sh[threadIdx.x][0];
}

__global__ __launch_bounds__(64) void k_stats_final(int nblocks, const double *part, double *out)
{
    const int c = threadIdx.x;
    if (c >= ST_N) return;
    double v = 0.0;
    for (int b = 0; b < nblocks; b++) v = stat_is_max(c) ? fmax(v, part[b * ST_N + c]) : v + part[b * ST_N + c];
    out[c] = v;
}

hipError_t launch_stats(hipStream_t st, int B, const double *stat, const int *iters, const int *status, const int *sqp,
                        const int *as2, bool with_closed_loop, const DevCfg &cfg, const double *cost_scen, double *part, double *out)
{
    int nblocks = (B + 255) / 256;
    if (nblocks > STATS_BLOCKS) nblocks = STATS_BLOCKS;
    StatWeights sw;
    sw.nl = cfg.kind == MPC_KIND_NL; sw.half_dt = 0.5 * cfg.dt; sw.umax = cfg.max_rate; sw.dmax = cfg.max_steering;
    for (int c = 0; c < 8; c++) sw.w[c] = sw.nl ? cfg.w_nl[c] : (c < 7 ? cfg.w[c] : 0.0);
    hipLaunchKernelGGL(k_stats_partial, dim3(nblocks), dim3(256), 0, st, B, stat, iters, status, sqp, as2, with_closed_loop, sw,
                       sw.nl ? nullptr : cost_scen, part);
    hipLaunchKernelGGL(k_stats_final, dim3(1), dim3(64), 0, st, nblocks, part, out);
    return hipGetLastError();
}

}  // namespace mpc
