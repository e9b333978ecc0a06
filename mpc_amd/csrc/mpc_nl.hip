// Nonlinear MPC step (NLMPC.py) as a wave-per-scenario Gauss-Newton SQP for gfx950.
//
// States [px, py, cos, sin, vy, r, delta], input u = steering rate, stage parameter vx.  One wavefront owns one
// scenario and lane k owns stage k.  The decision variables are the inputs alone (single shooting): per SQP iteration
//   1. the state trajectory is rolled out with the explicit RK4 map (NLMPC.py:196, one step per interval),
//   2. lane k differentiates that map at (x_k, u_k): A_k = d x_{k+1} / d x_k, b_k = d x_{k+1} / d u_k,
//   3. the LINEAR_LS cost (NLMPC.py:159-186) is condensed into H = Gamma' W Gamma + R and the gradient g,
//   4. the QP in the new inputs, min 1/2 U'HU + (g - H U_cur)'U subject to |u_k| <= max_rate and
//      |delta_0 + dt sum_{j<=k} u_j| <= max_steering (the steering bound stays exactly linear in U), is solved like the
//      linear controllers' QP: unconstrained minimiser first, interior point if a bound is violated,
//   5. the full step is taken; the iteration stops when the step falls below the tolerance.
// The input iterate is kept between calls, as the reference's solver object keeps its iterate (NLMPC.py:232).
#include "mpc_ipm.hpp"
#include "mpc_aset.hpp"

#pragma clang fp contract(fast)

namespace mpc {


// The stage record: A[7][5] (columns cos, sin, vy, r, delta; the px and py columns are unit vectors) + b[7], 42 entries
// of which 33 are kept -- d (vy, r) / d (cos, sin) and the steering row of A are zero whatever the point (NLMPC.py:108-134).
// nl_slot: place of entry q = r * 5 + c (A) or 35 + r (b) in the record, -1 for the entries that are not kept.  At 33
// doubles per stage a wave's LDS is 22.5 KB for N = 50: seven waves per compute unit instead of six.
constexpr int NL_STAGE_W = 33;
constexpr int nl_slot(int q)
{
    return q < 20 ? q : q < 22 ? -1 : q < 25 ? q - 2 : q < 27 ? -1 : q < 30 ? q - 4 : q < 35 ? -1 : q - 9;
}

struct NlCoef { double vx, a11, a12, a21, a22, b1, b2; };

// A constrained QP of the SQP iteration: the active-set solve in its free-input form (H of this scenario and iteration
// is all it needs), started from the previous QP's set -- the previous closed-loop step's moved one stage ahead
// (prm.warm_mode 2) or the previous SQP iteration's as it is (1) -- and the interior point for what it hands on, with the
// back-off of the fixed-model kernels' constrained step (off by default).  Called at two places of the SQP loop:
//   * before the unconstrained minimiser exists, when the previous QP was constrained (try_aset, !allow_ipm): the solve
//     works about the current iterate Uref with the objective's gradient g0 there, and a QP that ends here never pays
//     for the factorisation of H;
//   * after the unconstrained minimiser Uref (g0 = 0) turned out to violate a bound (allow_ipm; try_aset unless the
//     first call already tried).
// g: the QP's linear term (the interior point's argument).  A function of its own: its registers stay out of the SQP loop's.
struct NlQpResult {
    double U, mu, res;
    int iters, status;           // interior-point iterations and status
    bool warm_valid, solved, tried;
};

template <int N>
static __device__ __noinline__ __attribute__((cold)) MPC_NOTAIL NlQpResult nl_solve_constrained(IpmParams prm, int lane, const double *Hm, double g,
                                                                                   double Uref, double vdref, double g0, double up_d,
                                                                                   double lo_d, double *lds, int *as_o, bool try_aset,
                                                                                   bool allow_ipm)
{
    NlQpResult out = {0.0, 0.0, 0.0, 0, 0, false, false, false};
    if (try_aset) {
        // (the record is read as the previous solve's set, never extrapolated: the QPs of one step's SQP iterations are
        // not consecutive steps)
        const AsetTry r = aset_try_general<N, 3, false>(prm, lane, Hm, Uref, vdref, g0, up_d, lo_d, lds, as_o);
        out.tried = r.tried;
        if (r.solved) {
            out.U = r.U; out.warm_valid = true; out.solved = true;
            return out;
        }
    }
    if (!allow_ipm) return out;
    out.solved = true;
    const IpmResult r = interior_point<N, 3>(prm, lane, Hm, g, up_d, lo_d, (ldouble *)lds);
    out.U = r.U; out.mu = r.mu; out.res = r.res; out.iters = r.iters; out.status = r.status; out.warm_valid = r.warm_valid;
    return out;
}

// NLMPC.py:108-134
__device__ __forceinline__ void nl_f(const double *x, double u, const NlCoef &c, double *xd)
{
    xd[0] = c.vx * x[2] - x[4] * x[3];
    xd[1] = c.vx * x[3] + x[4] * x[2];
    xd[2] = -x[3] * x[5];
    xd[3] = x[2] * x[5];
    xd[4] = c.a11 * x[4] + c.a12 * x[5] + c.b1 * x[6];
    xd[5] = c.a21 * x[4] + c.a22 * x[5] + c.b2 * x[6];
    xd[6] = u;
}

__device__ __forceinline__ void nl_rk4(double *x, double u, double h, const NlCoef &c)
{
    double k1[7], k2[7], k3[7], k4[7], t[7];
    nl_f(x, u, c, k1);
#pragma unroll
    for (int i = 0; i < 7; i++) t[i] = fma(0.5 * h, k1[i], x[i]);
    nl_f(t, u, c, k2);
#pragma unroll
    for (int i = 0; i < 7; i++) t[i] = fma(0.5 * h, k2[i], x[i]);
    nl_f(t, u, c, k3);
#pragma unroll
    for (int i = 0; i < 7; i++) t[i] = fma(h, k3[i], x[i]);
    nl_f(t, u, c, k4);
#pragma unroll
    for (int i = 0; i < 7; i++) x[i] = fma(h / 6.0, k1[i] + 2.0 * (k2[i] + k3[i]) + k4[i], x[i]);
}

// r = (d f / d x)(xs) v, rows 0..5 (row 6 of the Jacobian is zero); xs4 = [cos, sin, vy, r] of the evaluation point
__device__ __forceinline__ void nl_jv(const double *xs4, const NlCoef &c, const double *v, double *r)
{
    const double co = xs4[0], si = xs4[1], vy = xs4[2], rr = xs4[3];
    r[0] = c.vx * v[2] - vy * v[3] - si * v[4];
    r[1] = vy * v[2] + c.vx * v[3] + co * v[4];
    r[2] = -rr * v[3] - si * v[5];
    r[3] = rr * v[2] + co * v[5];
    r[4] = c.a11 * v[4] + c.a12 * v[5] + c.b1 * v[6];
    r[5] = c.a21 * v[4] + c.a22 * v[5] + c.b2 * v[6];
}

// Exact derivative of the RK4 map at (x, u): st[r*5 + (c-2)] = d xn[r] / d x[c] for c = 2..6, st[35 + r] = d xn[r] / d u.
__device__ __forceinline__ void nl_linearise(const double *x, double u, double h, const NlCoef &c, double *st)
{
    double xs[4][4];
    {
        double k[7], t[7];
#pragma unroll
        for (int i = 0; i < 4; i++) xs[0][i] = x[2 + i];
        nl_f(x, u, c, k);
#pragma unroll
        for (int i = 0; i < 7; i++) t[i] = fma(0.5 * h, k[i], x[i]);
#pragma unroll
        for (int i = 0; i < 4; i++) xs[1][i] = t[2 + i];
        nl_f(t, u, c, k);
#pragma unroll
        for (int i = 0; i < 7; i++) t[i] = fma(0.5 * h, k[i], x[i]);
#pragma unroll
        for (int i = 0; i < 4; i++) xs[2][i] = t[2 + i];
        nl_f(t, u, c, k);
#pragma unroll
        for (int i = 0; i < 7; i++) t[i] = fma(h, k[i], x[i]);
#pragma unroll
        for (int i = 0; i < 4; i++) xs[3][i] = t[2 + i];
    }
#pragma unroll
    for (int sd = 0; sd < 6; sd++) {        // seeds: states 2..6, then the input
        const double du = (sd == 5) ? 1.0 : 0.0;
        double e[7] = {0, 0, 0, 0, 0, 0, 0}, K[6], acc[6], t[7];
        if (sd < 5) e[2 + sd] = 1.0;
        nl_jv(xs[0], c, e, K);
#pragma unroll
        for (int i = 0; i < 6; i++) { acc[i] = K[i]; t[i] = fma(0.5 * h, K[i], e[i]); }
        t[6] = fma(0.5 * h, du, e[6]);
        nl_jv(xs[1], c, t, K);
#pragma unroll
        for (int i = 0; i < 6; i++) { acc[i] = fma(2.0, K[i], acc[i]); t[i] = fma(0.5 * h, K[i], e[i]); }
        nl_jv(xs[2], c, t, K);
#pragma unroll
        for (int i = 0; i < 6; i++) { acc[i] = fma(2.0, K[i], acc[i]); t[i] = fma(h, K[i], e[i]); }
        t[6] = fma(h, du, e[6]);
        nl_jv(xs[3], c, t, K);
#pragma unroll
        for (int i = 0; i < 6; i++) {
            const double col = fma(h / 6.0, acc[i] + K[i], e[i]);
            const int q = sd < 5 ? i * 5 + sd : 35 + i;
            if (nl_slot(q) >= 0) st[nl_slot(q)] = col;
        }
        if (sd == 5) st[nl_slot(35 + 6)] = h;                // (the steering row of A is the unit row: not kept)
    }
}

template <int N>
__global__ __launch_bounds__(WAVE, 2) void k_mpc_nl(const DevCfg cfg, const SolveArgs a)
{
    // N is the capacity of this instantiation (registers, LDS and the Hessian scratch are laid out for it); the horizon
    // of the problem is Ne = cfg.N <= N.  Inputs Ne..N-1 do not exist in the problem: their rows of the condensed
    // Hessian are wr on the diagonal with a zero gradient, so they stay zero and touch nothing.
    const int lane = threadIdx.x;
    const int Ne = cfg.N;
    const int s = a.order ? a.order[blockIdx.x] : blockIdx.x;      // scenario this wave advances (grid = B workgroups)
    const bool valid = lane < Ne, validd = lane < Ne - 1;
    extern __shared__ double lds[];
    double *s_ref = lds;                              // [4][N+1]: x, y, cos, sin of the car-frame waypoints
    double *s_p = s_ref + 4 * (N + 1);                // [N+1] speeds
    double *s_coef = s_p + (N + 1);                   // [N][4]: a11, a12, a21, a22 per stage
    double *s_x = s_coef + 4 * N;                     // [N+1][7] rolled-out states
    double *s_stage = s_x + 7 * (N + 1);              // [N][NL_STAGE_W]
    double *s_wv = s_stage + N * NL_STAGE_W;          // [7][N] weighted prediction columns

    ModelParams mp;
    if (a.mp_scen) {
        const double *q = a.mp_scen + 6 * (size_t)s;
        mp = {q[0], q[1], q[2], q[3], q[4], q[5]};
    } else
        mp = {cfg.mp[0], cfg.mp[1], cfg.mp[2], cfg.mp[3], cfg.mp[4], cfg.mp[5]};
    const double dt = cfg.dt;
    const double umax = cfg.max_rate, dmax = cfg.max_steering;
    const double b1 = -mp.Cf / mp.m, b2 = -mp.lf * mp.Cf / mp.Iz;
    double wq[7], wN[7];
#pragma unroll
    for (int c = 0; c < 7; c++) { wq[c] = dt * cfg.w_nl[c]; wN[c] = cfg.we_nl[c]; }   // stage costs carry the time step
    const double wr = dt * cfg.w_nl[7];

    double plant[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    double prevp = 0.0, tvel = 0.0;
    int lapv = 0;
    if (a.mode != MODE_SOLVE) {
#pragma unroll
        for (int i = 0; i < 8; i++) plant[i] = a.state[(size_t)8 * s + i];
        tvel = a.target_vel[s];
        if (a.mode == MODE_SKIDPAD) { lapv = a.lap[s]; prevp = a.prev_progress[s]; }
    }
    const PlantCoef pc = plant_coef(mp, plant[4]);             // constant over the launch: x[4] has zero derivative
    const double cs0 = speed_cumsum(tvel, lane), cs1 = (lane + WAVE <= Ne) ? speed_cumsum(tvel, lane + WAVE) : 0.0;
    double U = valid ? a.lin_u[(size_t)s * Ne + lane] : 0.0;   // SQP iterate, carried from call to call
    double *warm_rec = a.ipm_ws ? a.ipm_ws + (size_t)s * IPM_WARM_DOUBLES : nullptr;
    bool warm_ok = warm_rec && a.mode != MODE_SOLVE && warm_rec[5 * WAVE] == 1.0;
    const int nsteps = (a.mode == MODE_SOLVE) ? 1 : a.nsteps;
    int iters_tot = 0, status_max = 0, nconstr = 0, sqp_tot = 0, lap_flag = 0;
    double mu = 0.0, res = 0.0, u0 = 0.0;
    double *Hw = a.Hws + (size_t)s * N * N;
    if (lane == 0 && a.as_out && !a.accumulate) { int *o = a.as_out + 6 * (size_t)s; o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = 0; }
    // closed-loop statistics of the scenario (lane i carries statistic i).  The stage cost is the NL controller's:
    // y = [px, py, cos, sin, vy, r, delta | u] against [x, y, cos, sin, 0, 0, 0 | 0] (NLMPC.py:159-186, 208-212).
    const bool cl_on = a.stat_out != nullptr && a.mode != MODE_SOLVE;
    double cl_acc = (cl_on && a.accumulate && lane < CL_STAT_W) ? a.stat_out[STAT_W * (size_t)s + STAT_CL + lane] : 0.0;

#pragma unroll 1
    for (int t = 0; t < nsteps; t++) {
        // ------------------------------------------------------------ 1. start state and references
        double xh[7] = {0, 0, 1, 0, 0, 0, 0};                           // NLMPC.py:216
        if (a.mode == MODE_SOLVE) {
            const double *x0 = a.x0 + 7 * (size_t)s;
            xh[4] = x0[4]; xh[5] = x0[5]; xh[6] = x0[6];
            for (int k = lane; k <= Ne; k += WAVE) {
                const double *w = a.wp + ((size_t)s * (Ne + 1) + k) * 4;
#pragma unroll
                for (int c = 0; c < 4; c++) s_ref[c * (N + 1) + k] = w[c];   // NLMPC.py:208-212
                s_p[k] = a.p[(size_t)s * (Ne + 1) + k];
            }
        } else {
            const double px = plant[0], py = plant[1];
            xh[4] = plant[5]; xh[5] = plant[6]; xh[6] = plant[7];
            double p0;
            if (a.mode == MODE_SKIDPAD) {
                int lp = lapv;
                p0 = sk_pos2progress(px, py, lp);
                if (prevp > p0 + 2 * SK_R) { lp += 1; p0 = sk_pos2progress(px, py, lp); }
                if (lp < 0 || lp > 7) lap_flag = 5;      // the reference switches to its speed profile there (path_planning.py:241-248)
                prevp = p0;
            } else
                p0 = px;
            double ch, sh;
            unit_heading(plant[2], plant[3], ch, sh);
            for (int k = lane; k <= Ne; k += WAVE) {
                double o[4];
                const double q = (k == 0) ? p0 : p0 + (k < WAVE ? cs0 : cs1) * dt;
                if (a.mode == MODE_SKIDPAD) sk_pose(q, o);
                else step_pose(q, a.step_ramp, o);
                to_car_frame(o, px, py, ch, sh);
#pragma unroll
                for (int c = 0; c < 4; c++) s_ref[c * (N + 1) + k] = o[c];
                s_p[k] = tvel;
                if (a.wp_out) {
                    double *w = a.wp_out + ((size_t)s * (Ne + 1) + k) * 4;
                    w[0] = o[0]; w[1] = o[1]; w[2] = o[2]; w[3] = o[3];
                }
            }
        }
        wave_sync();
        double cl_err = 0.0;               // lane i < 7: component i of [x0; u0] - yref_0 (the input joins after the solve)
        if (cl_on) {
            cl_err = lane < 4 ? s_ref[lane * (N + 1)] : 0.0;
            cl_err = lane == 2 ? 1.0 - cl_err : cl_err;
            cl_err = lane == 4 ? xh[4] : lane == 5 ? xh[5] : lane == 6 ? xh[6] : cl_err;
        }
        NlCoef cf;          // this lane's stage
        cf.vx = s_p[valid ? lane : 0];
        {
            double a11, a12, a21, a22;
            lat_coeffs(mp, cf.vx, cfg.eps_nl, a11, a12, a21, a22);      // NLMPC.py:116-128
            cf.a11 = a11; cf.a12 = a12; cf.a21 = a21; cf.a22 = a22; cf.b1 = b1; cf.b2 = b2;
            if (valid) { s_coef[lane * 4] = a11; s_coef[lane * 4 + 1] = a12; s_coef[lane * 4 + 2] = a21; s_coef[lane * 4 + 3] = a22; }
        }
        wave_sync();
        double up_d, lo_d;
        bool relaxed;
        steer_bounds(xh[6], dmax, umax, dt, lane, up_d, lo_d, relaxed);

        // ------------------------------------------------------------ 2. SQP iterations
        int iters = 0, status = 1, sqp = 0;
        double keep[7], step_prev = 1e300;
#pragma unroll 1
        for (;;) {
            const bool last = (status == 0) || (sqp >= cfg.sqp_max_iter);
            if (last && !a.traj_out) break;
            // 2a. roll the trajectory out; lane k keeps x_k, every lane ends with x_N
            double x[7];
#pragma unroll
            for (int i = 0; i < 7; i++) { x[i] = xh[i]; keep[i] = xh[i]; }
#pragma unroll 1
            for (int k = 0; k < Ne; k++) {
                NlCoef ck;
                ck.vx = s_p[k]; ck.a11 = s_coef[k * 4]; ck.a12 = s_coef[k * 4 + 1]; ck.a21 = s_coef[k * 4 + 2];
                ck.a22 = s_coef[k * 4 + 3]; ck.b1 = b1; ck.b2 = b2;
                nl_rk4(x, bcast(U, k), dt, ck);
                if (lane == k + 1) {
#pragma unroll
                    for (int i = 0; i < 7; i++) keep[i] = x[i];
                }
            }
            if (last) {                       // the pass after the final step only produces the prediction
                if (valid) {
                    double *o = a.traj_out + ((size_t)s * (Ne + 1) + lane) * 7;
#pragma unroll
                    for (int i = 0; i < 7; i++) o[i] = keep[i];
                }
                if (lane == 0) {
                    double *o = a.traj_out + ((size_t)s * (Ne + 1) + Ne) * 7;
#pragma unroll
                    for (int i = 0; i < 7; i++) o[i] = x[i];
                }
                break;
            }
            if (valid) {
#pragma unroll
                for (int i = 0; i < 7; i++) s_x[lane * 7 + i] = keep[i];
            }
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 7; i++) s_x[Ne * 7 + i] = x[i];
            }
            // 2b. derivative of the RK4 map per stage
            if (valid) nl_linearise(keep, U, dt, cf, &s_stage[lane * NL_STAGE_W]);
            wave_sync();

            // 2c. condense: lane j carries v = d x_k / d u_j
            Newton<N> nt;
            double g = 0.0;
            {
#pragma unroll
                for (int i = 0; i < N; i++) nt.row[i] = 0.0;
                double v[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
                for (int k = 1; k <= Ne; k++) {
                    const double *stl = &s_stage[(k - 1) * NL_STAGE_W];
                    double st[42];                        // the stage record in one batch of LDS reads (otherwise the
#pragma unroll                                             // scheduler waits for each value where it is used)
                    for (int q = 0; q < 42; q++) st[q] = nl_slot(q) >= 0 ? stl[nl_slot(q) >= 0 ? nl_slot(q) : 0] : 0.0;
                    double errk[7];                       // x_k - reference of stage k, read in the same batch
#pragma unroll
                    for (int c = 0; c < 7; c++) errk[c] = s_x[k * 7 + c];
                    double refk[4];
#pragma unroll
                    for (int c = 0; c < 4; c++) refk[c] = s_ref[c * (N + 1) + k];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < 4; c++) errk[c] -= refk[c];
                    {
                        double n[6];
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            double acc = (r < 2) ? v[r] : 0.0;
#pragma unroll
                            for (int c = 0; c < 5; c++) acc = fma(st[r * 5 + c], v[2 + c], acc);
                            n[r] = acc;
                        }
#pragma unroll
                        for (int r = 4; r < 6; r++) {      // vy and r depend on vy, r, delta only
                            double acc = 0.0;
#pragma unroll
                            for (int c = 2; c < 5; c++) acc = fma(st[r * 5 + c], v[2 + c], acc);
                            n[r] = acc;
                        }
#pragma unroll
                        for (int r = 0; r < 6; r++) v[r] = n[r];
                        if (lane == k - 1) {
#pragma unroll
                            for (int r = 0; r < 7; r++) v[r] = st[35 + r];
                        }
                    }
                    double wv[7];
#pragma unroll
                    for (int c = 0; c < 7; c++) wv[c] = (k < Ne ? wq[c] : wN[c]) * v[c];
#pragma unroll
                    for (int c = 0; c < 7; c++) g = fma(wv[c], errk[c], g);
                    if (lane < N) {             // the columns of the inputs beyond Ne are zero (v never leaves zero there)
#pragma unroll
                        for (int c = 0; c < 7; c++) s_wv[c * N + lane] = wv[c];
                    }
                    wave_sync();
#pragma unroll
                    for (int c = 0; c < 7; c++) {
                        if ((k < Ne ? wq[c] : wN[c]) != 0.0) {
                            axpy_from_lds<N>(nt.row, &s_wv[c * N], v[c]);
                        }
                    }
                    wave_sync();
                }
            }
            // 2d. QP in the new inputs: gradient g + wr U - (H U) with H = Gamma' W Gamma + wr I
            double hu = 0.0;
#pragma unroll
            for (int j = 0; j < N; j++) {
                hu = fma(nt.row[j], bcast(U, j), hu);
                if ((j & 7) == 7) __builtin_amdgcn_sched_barrier(0);
            }
            double gq = valid ? g - hu : 0.0;        // (g + wr U) - (Gamma'W Gamma U + wr U)
            const double wrd = opaque_value(wr);      // keeps the N one-hot addends (and their N compare masks)
            const int lane_d = opaque_int(lane);      // out of hoisted, spilled registers
#pragma unroll
            for (int i = 0; i < N; i++) nt.row[i] += (i == lane_d) ? wrd : 0.0;
            if (lane < N) {                         // one divergent region for all N stores, not one per entry
                store_column<N>(Hw + lane, nt.row);
            }
            // start from the previous QP's solve: the previous SQP iteration's, or the previous closed-loop step's moved one
            // stage ahead
            const int wm = warm_ok ? (sqp == 0 ? 2 : 1) : 0;
            const IpmParams prm = {dt, umax, dmax, cfg.tol_mu, cfg.tol_res, cfg.max_iter, warm_rec, wm, Ne};
            int *as_o = a.as_out ? a.as_out + 6 * (size_t)s : nullptr;
            const bool aset_on = a.general_aset != 0 && Ne == N;
            double Un = 0.0;
            bool done = false, tried = false;
            // (warm_ok is wave-uniform but arrives in a vector register from the solver functions: read through
            // v_readfirstlane the branch around the call is a scalar one, and the call's register saves run with every
            // lane enabled)
            if (aset_on && __builtin_amdgcn_readfirstlane((int)warm_ok) != 0) {
                // the previous QP was constrained: this one most likely is, with nearly the same set
                const double g0 = valid ? g + wr * U : 0.0;                       // gradient of the QP at the current iterate
                const NlQpResult r = nl_solve_constrained<N>(prm, lane, Hw, gq, U, dt * incl_prefix(U, lane), g0, up_d, lo_d, s_wv, as_o, true, false);
                // (the returned flags stay in their vector registers: with these two made scalar as well the kernel was
                // measured 10% slower)
                tried = r.tried;
                if (r.solved) { done = true; nconstr++; Un = r.U; warm_ok = r.warm_valid; }
            }
            if (!done) {
                nt.template factor<false>(lane, (ldouble *)s_wv);
                Un = nt.solve(gq == 0.0 ? 0.0 : -gq, lane);   // every lane takes part: the rows beyond Ne are unit rows with a zero right-hand side
                if (!valid) Un = 0.0;
                const double vd = dt * incl_prefix(Un, lane);
                bool ok = fabs(Un) <= umax;
                if (validd) ok = ok && (vd <= up_d) && (vd >= lo_d);
                if (!__all(ok)) {
                    nconstr++;
                    const NlQpResult r = nl_solve_constrained<N>(prm, lane, Hw, gq, Un, vd, 0.0, up_d, lo_d, s_wv, as_o, aset_on && !tried, true);
                    Un = r.U; mu = r.mu; res = r.res; iters += r.iters;
                    if (r.status) status_max = max(status_max, r.status);
                    warm_ok = (warm_rec != nullptr) && r.warm_valid;
                } else
                    warm_ok = false;
            }
            const double step = wave_max(valid ? fabs(Un - U) : 0.0);
            U = valid ? Un : 0.0;
            sqp++;
            // converged: step below the tolerance, or within 1000x of it and no longer shrinking (the floor set by the
            // accuracy of the QP solutions has been reached)
            if (step <= cfg.sqp_tol || (step <= 1e3 * cfg.sqp_tol && step >= 0.5 * step_prev)) status = 0;
            step_prev = step;
        }
        if (relaxed) status_max = max(status_max, 4);
        status_max = max(status_max, lap_flag);
        sqp_tot += sqp;
        iters_tot += iters;
        status_max = max(status_max, status);

        // ------------------------------------------------------------ 3. outputs and plant
        if (valid && a.U_out) a.U_out[(size_t)s * Ne + lane] = U;
        u0 = bcast(U, 0);
        if (a.mode != MODE_SOLVE) {
            const double oldx = plant[0];
            if (cl_on) {
                // the heading pair over the last roll-out of the solve (lane k: stage k; stage N is one more RK4 step of
                // the same map): the pair is a free state of this model (skidpad_simulator.py:134)
                const double pv = wave_max(valid ? cl_pair_violation(keep[2], keep[3]) : 0.0);
                cl_acc = cl_step(cl_acc, lane, cl_value(lane, lane == 7 ? u0 : cl_err, cl_pair_violation(plant[2], plant[3]), u0, plant[7], pv));
            }
            plant_rk4_c<10>(plant, u0, dt, pc);
            if (a.mode == MODE_SKIDPAD && plant[0] > SK_CENTER && oldx <= SK_CENTER) lapv += 1;
            if (a.u_hist && lane == 0) a.u_hist[(size_t)t * a.B + s] = u0;
            if (a.x_hist && lane < 8) {
                double val = plant[0];
#pragma unroll
                for (int i = 1; i < 8; i++) val = (lane == i) ? plant[i] : val;
                a.x_hist[((size_t)t * a.B + s) * 8 + lane] = val;
            }
        }
        wave_sync();
    }

    if (valid) a.lin_u[(size_t)s * Ne + lane] = U;
    if (warm_rec && lane == 0) warm_rec[5 * WAVE] = (warm_ok && a.mode != MODE_SOLVE) ? 1.0 : 0.0;
    if (a.mode != MODE_SOLVE) {
        if (lane < 8) {
            double val = plant[0];
#pragma unroll
            for (int i = 1; i < 8; i++) val = (lane == i) ? plant[i] : val;
            a.state[(size_t)8 * s + lane] = val;
        }
        if (a.mode == MODE_SKIDPAD && lane == 0) { a.lap[s] = lapv; a.prev_progress[s] = prevp; }
    }
    if (lane == 0) {
        if (a.u0_out) a.u0_out[s] = u0;
        if (a.work_out) a.work_out[s] = iters_tot;
        const int st_all = a.accumulate ? max(a.status_own[s], status_max) : status_max;
        const int it_all = a.accumulate ? a.iters_own[s] + iters_tot : iters_tot;
        a.status_own[s] = st_all; a.iters_own[s] = it_all;
        if (a.status_out) a.status_out[s] = st_all;
        if (a.iters_out) a.iters_out[s] = it_all;
        if (a.sqp_out) a.sqp_out[s] = a.accumulate ? a.sqp_out[s] + sqp_tot : sqp_tot;
        if (a.stat_out) {
            double *so = a.stat_out + STAT_W * (size_t)s;
            so[0] = mu; so[1] = res; so[2] = (a.accumulate ? so[2] : 0.0) + (double)nconstr; so[3] = fabs(u0);
        }
    }
    if (cl_on && lane < CL_STAT_W) a.stat_out[STAT_W * (size_t)s + STAT_CL + lane] = cl_acc;
}

template <int N>
static hipError_t launch_nl_n(const DevCfg &cfg, const SolveArgs &a, hipStream_t st)
{
    const size_t lds = sizeof(double) * (size_t)(4 * (N + 1) + (N + 1) + 4 * N + 7 * (N + 1) + N * NL_STAGE_W + 7 * N);
    hipLaunchKernelGGL(k_mpc_nl<N>, dim3(a.B), dim3(WAVE), lds, st, cfg, a);
    return hipGetLastError();
}

// smallest instantiated horizon that holds N stages (0: none)
int nl_horizon_capacity(int N)
{
#define MPC_CASE(n) if (N >= 2 && N <= n) return n;
    MPC_HORIZONS(MPC_CASE)
#undef MPC_CASE
    return 0;
}

hipError_t launch_mpc_nl(const DevCfg &cfg, const SolveArgs &a, hipStream_t st)
{
    switch (nl_horizon_capacity(cfg.N)) {
#define MPC_CASE(n) case n: return launch_nl_n<n>(cfg, a, st);
        MPC_HORIZONS(MPC_CASE)
#undef MPC_CASE
    default: return hipErrorInvalidValue;
    }
}

}  // namespace mpc
