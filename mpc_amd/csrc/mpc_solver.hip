// Wave-per-scenario condensed MPC step for gfx950.
//
// One 64-lane wavefront owns one scenario; lane k owns stage k of the horizon (decision variable u_k,
// column k of the prediction matrix, row k of the condensed Hessian).  The kernel
//   1. plans the car-frame reference (closed-loop modes) or reads it (optimize() mode),
//   2. linearises/discretises the controller model per stage,
//   3. condenses the OCP into  min 1/2 U'HU + g'U,  |u_k| <= umax,  |delta_0 + dt*sum_{j<=k} u_j| <= dmax,
//   4. solves it exactly: unconstrained solve first, Mehrotra predictor-corrector interior point only if a
//      bound is violated; every Newton system K = H + A'DA is factorised K = L D L' with row k of K held in
//      the registers of lane k and pivots broadcast through v_readlane,
//   5. rolls the prediction out, advances the plant with RK4 and does the lap bookkeeping.
//
// Reference behaviour restated (file:line of the reference checkout):
//   LMPC2.py:114-168 / OFLMPC2.py:118-181 / LPVMPC.py:110-171  stage model      -> stage_model()
//   LMPC2.py:220-260, LPVMPC.py:191-226, OFLMPC2.py:239-279    LINEAR_LS cost   -> condense()
//   LMPC2.py:196-210                                           bounds           -> ipm
//   LMPC2.py:296-347, LPVMPC.py:254-341, OFLMPC2.py:315-367    optimize()       -> k_mpc mode 0
//   skidpad_simulator.py:101-135, simulator.py:178-202         closed loop      -> k_mpc mode 1/2
#include "mpc_device.hpp"
#include "mpc_kernels.h"

namespace mpc {

// ------------------------------------------------------------------------------------------------
template <int N>
struct Newton {
    // Row `lane` of the symmetric Newton matrix.  After factor(): entries k < lane hold l(lane,k),
    // entry lane holds d(lane), entries k > lane hold the frozen d(lane)*l(k,lane) (the transposed
    // factor), so both triangular solves index registers statically.
    double row[N];
    double dinv;

    __device__ __forceinline__ void factor(int lane)
    {
        dinv = 1.0;
#pragma unroll
        for (int k = 0; k < N; k++) {
            const double c = row[k];                 // column k of the trailing matrix, one entry per lane
            const double dk = bcast(c, k);
            const double inv = 1.0 / dk;
            const bool below = lane > k;
            const double l = below ? c * inv : 0.0;
            if (lane == k) dinv = inv;
#pragma unroll
            for (int j = k + 1; j < N; j++) row[j] = fma(-l, bcast(c, j), row[j]);
            if (below) row[k] = l;
        }
    }

    // solves (L D L') x = b; b, x one entry per lane
    __device__ __forceinline__ double solve(double b, int lane) const
    {
        double w = b;
#pragma unroll
        for (int k = 0; k < N - 1; k++) {
            const double wk = bcast(w, k);
            w = fma(-(lane > k ? row[k] : 0.0), wk, w);
        }
#pragma unroll
        for (int k = N - 1; k > 0; k--) {
            const double xk = bcast(w * dinv, k);
            w = fma(-(lane < k ? row[k] : 0.0), xk, w);
        }
        return w * dinv;
    }
};

struct StageRow { double e02, e03, e12, e13, f33, f34, f43, f44, c0, c1; };
constexpr int STAGE_W = 10;

// ------------------------------------------------------------------------------------------------
template <int N>
__global__ __launch_bounds__(WAVE) void k_mpc(const DevCfg cfg, const SolveArgs a)
{
    const int s = blockIdx.x;
    const int lane = threadIdx.x;
    const bool valid = lane < N;        // lane owns a decision variable
    const bool validd = lane < N - 1;   // lane owns the bound on delta_{lane+1}

    __shared__ double s_ref[(N + 1) * 3];
    __shared__ double s_p[N + 1];
    __shared__ double s_stage[N * STAGE_W];
    __shared__ double s_wv[6 * N];

    ModelParams mp;
    if (a.mp_scen) {
        const double *q = a.mp_scen + 6 * (size_t)s;
        mp = {q[0], q[1], q[2], q[3], q[4], q[5]};
    } else
        mp = {cfg.mp[0], cfg.mp[1], cfg.mp[2], cfg.mp[3], cfg.mp[4], cfg.mp[5]};
    const double dt = cfg.dt;

    // ---------------------------------------------------------------- 1. initial state and references
    double xh3, xh4, xh5;               // controller start [0,0,0,vy,r,delta] (LMPC2.py:298)
    double plant[8];
    double dhat = 0.0;
    int lapv = 0;
    if (a.mode == MODE_SOLVE) {
        const double *x0 = a.x0 + 7 * (size_t)s;
        xh3 = x0[4]; xh4 = x0[5]; xh5 = x0[6];
        if (a.d_hat) dhat = a.d_hat[s];
        for (int k = lane; k <= N; k += WAVE) {
            const double *w = a.wp + ((size_t)s * (N + 1) + k) * 4;
            s_ref[k * 3 + 0] = w[0]; s_ref[k * 3 + 1] = w[1]; s_ref[k * 3 + 2] = w[3];   // LMPC2.py:290-294
            s_p[k] = a.p[(size_t)s * (N + 1) + k];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) plant[i] = a.state[8 * (size_t)s + i];
        xh3 = plant[5]; xh4 = plant[6]; xh5 = plant[7];
        const double heading = atan2(plant[3], plant[2]);              // skidpad_simulator.py:104
        const double v = a.target_vel[s];
        double p0;
        if (a.mode == MODE_SKIDPAD) {
            lapv = a.lap[s];
            int lp = lapv;
            p0 = sk_pos2progress(plant[0], plant[1], lp);
            if (a.prev_progress[s] > p0 + 2 * SK_R) { lp += 1; p0 = sk_pos2progress(plant[0], plant[1], lp); }
        } else
            p0 = plant[0];                                             // step_planning.py:83-86
        const double ch = cos(heading), sh = sin(heading);
        for (int k = lane; k <= N; k += WAVE) {
            double o[4];
            const double q = waypoint_progress(p0, v, dt, k);
            if (a.mode == MODE_SKIDPAD) sk_pose(q, o);
            else { o[0] = q; o[1] = (q < 0.0) ? 0.0 : 1.0; o[2] = 1.0; o[3] = 0.0; }   // step_planning.py:63-80
            to_car_frame(o, plant[0], plant[1], ch, sh);
            s_ref[k * 3 + 0] = o[0]; s_ref[k * 3 + 1] = o[1]; s_ref[k * 3 + 2] = o[3];
            s_p[k] = v;
            if (a.wp_out) {
                double *w = a.wp_out + ((size_t)s * (N + 1) + k) * 4;
                w[0] = o[0]; w[1] = o[1]; w[2] = o[2]; w[3] = o[3];
            }
        }
        if (lane == 0) a.prev_progress[s] = p0;
    }
    __syncthreads();

    // ---------------------------------------------------------------- 2. per-stage model (lane k -> stage k)
    const double b1 = -mp.Cf / mp.m, b2 = -mp.lf * mp.Cf / mp.Iz;
    const double f35 = dt * b1, f45 = dt * b2;
    const double c3 = (cfg.kind == MPC_KIND_OFL) ? dt * dhat : 0.0;    // OFLMPC2.py:145
    if (valid) {
        const double vx = s_p[lane];
        double a11, a12, a21, a22;
        lat_coeffs(mp, vx, cfg.eps, a11, a12, a21, a22);
        StageRow r;
        r.f33 = 1.0 + dt * a11; r.f34 = dt * a12; r.f43 = dt * a21; r.f44 = 1.0 + dt * a22;
        if (cfg.kind == MPC_KIND_LPV) {
            // x+ = x + dt (J(xb)(x - xb) + B (u - ub) + f(xb, vx, ub)), LPVMPC.py:125-171.  Linearisation
            // point per LPVMPC.py:260-297: "reference" takes ref_points[i,3], a column that is always 0,
            // as heading; "prev_iter" takes the previous solution shifted by one stage.
            const double *lx = a.lin_x + (size_t)s * 6 * (N + 1);
            const double psib = cfg.lpv_lin_prev ? lx[2 * (N + 1) + lane + 1] : 0.0;
            const double vyb = lx[3 * (N + 1) + lane + 1];
            const double sn = sin(psib), co = cos(psib);
            const double J02 = -vx * sn - vyb * co, J03 = -sn, J12 = vx * co - vyb * sn, J13 = co;
            r.e02 = dt * J02; r.e03 = dt * J03; r.e12 = dt * J12; r.e13 = dt * J13;
            r.c0 = dt * ((vx * co - vyb * sn) - (J02 * psib + J03 * vyb));
            r.c1 = dt * ((vx * sn + vyb * co) - (J12 * psib + J13 * vyb));
        } else {
            r.e02 = 0.0; r.e03 = 0.0; r.e12 = dt * vx; r.e13 = dt;      // LMPC2.py:127-129
            r.c0 = dt * vx; r.c1 = 0.0;
        }
        double *st = &s_stage[lane * STAGE_W];
        st[0] = r.e02; st[1] = r.e03; st[2] = r.e12; st[3] = r.e13; st[4] = r.f33; st[5] = r.f34; st[6] = r.f43;
        st[7] = r.f44; st[8] = r.c0; st[9] = r.c1;
    }
    __syncthreads();

    // ---------------------------------------------------------------- 3. condense
    double wq[6];
#pragma unroll
    for (int c = 0; c < 6; c++) wq[c] = dt * cfg.w[c];   // acados scales stage costs by the time step
    const double wr = dt * cfg.w[6];
    Newton<N> nt;
    double g = 0.0;
    {
        double *Hrow = nt.row;
#pragma unroll
        for (int i = 0; i < N; i++) Hrow[i] = 0.0;
        double v[6] = {0, 0, 0, 0, 0, 0};
        double xf[6] = {0, 0, 0, xh3, xh4, xh5};
#pragma unroll 1
        for (int k = 1; k <= N; k++) {
            const double *st = &s_stage[(k - 1) * STAGE_W];
            const double e02 = st[0], e03 = st[1], e12 = st[2], e13 = st[3], f33 = st[4], f34 = st[5], f43 = st[6],
                         f44 = st[7], c0 = st[8], c1 = st[9];
            {   // free response x_k = A_{k-1} x_{k-1} + c_{k-1}
                const double n0 = xf[0] + e02 * xf[2] + e03 * xf[3] + c0;
                const double n1 = xf[1] + e12 * xf[2] + e13 * xf[3] + c1;
                const double n2 = xf[2] + dt * xf[4];
                const double n3 = f33 * xf[3] + f34 * xf[4] + f35 * xf[5] + c3;
                const double n4 = f43 * xf[3] + f44 * xf[4] + f45 * xf[5];
                xf[0] = n0; xf[1] = n1; xf[2] = n2; xf[3] = n3; xf[4] = n4;
            }
            {   // column `lane` of the prediction matrix at stage k
                const double n0 = v[0] + e02 * v[2] + e03 * v[3];
                const double n1 = v[1] + e12 * v[2] + e13 * v[3];
                const double n2 = v[2] + dt * v[4];
                const double n3 = f33 * v[3] + f34 * v[4] + f35 * v[5];
                const double n4 = f43 * v[3] + f44 * v[4] + f45 * v[5];
                v[0] = n0; v[1] = n1; v[2] = n2; v[3] = n3; v[4] = n4;
                if (lane == k - 1) { v[0] = 0; v[1] = 0; v[2] = 0; v[3] = 0; v[4] = 0; v[5] = dt; }
            }
            double err[6] = {xf[0] - s_ref[k * 3 + 0], xf[1] - s_ref[k * 3 + 1], xf[2] - s_ref[k * 3 + 2], xf[3], xf[4],
                             xf[5]};
            double wv[6];
            if (k < N) {
#pragma unroll
                for (int c = 0; c < 6; c++) { wv[c] = wq[c] * v[c]; g = fma(wv[c], err[c], g); }
            } else {
                // terminal weight; the LPV terminal reference stays at the constructor's zeros (LPVMPC.py:220-221)
                if (cfg.kind == MPC_KIND_LPV) { err[0] = xf[0]; err[1] = xf[1]; err[2] = xf[2]; }
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double t = 0.0;
#pragma unroll
                    for (int d = 0; d < 6; d++) t = fma(cfg.We[c * 6 + d], v[d], t);
                    wv[c] = t;
                    g = fma(t, err[c], g);
                }
            }
            if (valid) {
#pragma unroll
                for (int c = 0; c < 6; c++) s_wv[c * N + lane] = wv[c];
            }
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 6; c++) {
                if (k == N || wq[c] != 0.0) {
                    const double vc = v[c];
#pragma unroll
                    for (int i = 0; i < N; i++) Hrow[i] = fma(s_wv[c * N + i], vc, Hrow[i]);
                }
            }
            __syncthreads();
        }
        double *Hg = a.Hws + (size_t)s * N * N;
#pragma unroll
        for (int i = 0; i < N; i++) {
            if (i == lane) Hrow[i] += wr;
            if (valid) Hg[i * N + lane] = Hrow[i];
        }
    }
    if (!valid) g = 0.0;

    // ---------------------------------------------------------------- 4. solve
    const double umax = cfg.max_rate, dmax = cfg.max_steering;
    const double up_d = dmax - xh5, lo_d = -dmax - xh5;     // bounds on dt * prefix sum
    const double dt2 = dt * dt;
    const double *Hg = a.Hws + (size_t)s * N * N;
    double U = 0.0;
    int iters = 0, status = 0, constrained = 0;
    double mu = 0.0, res = 0.0;

    // unconstrained minimiser: H is already in nt.row
    nt.factor(lane);
    {
        const double Uu = valid ? nt.solve(-g, lane) : 0.0;
        const double vd = dt * incl_prefix(Uu, lane);
        bool ok = fabs(Uu) <= umax;
        if (validd) ok = ok && (vd <= up_d) && (vd >= lo_d);
        if (__all(ok)) U = Uu;
        else constrained = 1;
    }
    if (constrained) {
        // Mehrotra predictor-corrector on  G U + t = h,  G = [A; -A], A = [I; dt * prefix-sum]
        const double gscale = fmax(1.0, wave_max(fabs(g)));
        double tpu, tmu, lpu = 1.0, lmu = 1.0, tpd, tmd, lpd = 1.0, lmd = 1.0;
        tpu = fmax(umax - U, 0.1 * umax); tmu = fmax(U + umax, 0.1 * umax);
        tpd = fmax(up_d, 0.1 * dmax); tmd = fmax(-lo_d, 0.1 * dmax);
        if (!valid) { tpu = 1.0; tmu = 1.0; lpu = 0.0; lmu = 0.0; }
        if (!validd) { tpd = 1.0; tmd = 1.0; lpd = 0.0; lmd = 0.0; }
        const double inv_nc = 1.0 / (2.0 * (2 * N - 1));
        double res_prev = 1e300;
#pragma unroll 1
        for (;;) {
            // residuals
            double hu = 0.0;
#pragma unroll
            for (int j = 0; j < N; j++) hu = fma(valid ? Hg[j * N + lane] : 0.0, bcast(U, j), hu);
            const double vd = dt * incl_prefix(U, lane);
            const double lamd = validd ? (lpd - lmd) : 0.0;
            double rd = hu + g + (lpu - lmu) + dt * incl_suffix(lamd, lane);
            if (!valid) rd = 0.0;
            const double rppu = valid ? U + tpu - umax : 0.0, rpmu = valid ? -U + tmu - umax : 0.0;
            const double rppd = validd ? vd + tpd - up_d : 0.0, rpmd = validd ? -vd + tmd + lo_d : 0.0;
            mu = wave_sum(tpu * lpu + tmu * lmu + tpd * lpd + tmd * lmd) * inv_nc;
            const double resd = wave_max(fabs(rd));
            const double resp = wave_max(fmax(fmax(fabs(rppu), fabs(rpmu)), fmax(fabs(rppd), fabs(rpmd))));
            res = fmax(resd / gscale, resp);
            if (mu <= cfg.tol_mu && (res <= cfg.tol_res || res >= 0.5 * res_prev)) break;
            if (iters >= cfg.max_iter) { status = 2; break; }
            res_prev = res;
            // Newton matrix  K = H + diag(Du) + dt^2 * Sfx[max(i,j)],  Sfx = suffix sums of Dd
            const double Du = lpu / tpu + lmu / tmu;
            const double Dd = validd ? lpd / tpd + lmd / tmd : 0.0;
            const double sfx = incl_suffix(Dd, lane);
#pragma unroll
            for (int j = 0; j < N; j++) {
                const double sj = bcast(sfx, j);
                double kij = valid ? Hg[j * N + lane] : 0.0;
                kij = fma(dt2, (j > lane) ? sj : sfx, kij);
                if (j == lane) kij += Du;
                nt.row[j] = kij;
            }
            nt.factor(lane);
            double dU = 0, dtpu = 0, dtmu = 0, dlpu = 0, dlmu = 0, dtpd = 0, dtmd = 0, dlpd = 0, dlmd = 0;
            double sigmu = 0.0, alpha = 1.0;
#pragma unroll 1
            for (int pass = 0; pass < 2; pass++) {
                double rcpu = tpu * lpu, rcmu = tmu * lmu, rcpd = tpd * lpd, rcmd = tmd * lmd;
                if (pass) {
                    rcpu += dtpu * dlpu - sigmu; rcmu += dtmu * dlmu - sigmu;
                    rcpd += dtpd * dlpd - sigmu; rcmd += dtmd * dlmd - sigmu;
                }
                const double wu = (lpu * rppu - rcpu) / tpu - (lmu * rpmu - rcmu) / tmu;
                const double wd = validd ? (lpd * rppd - rcpd) / tpd - (lmd * rpmd - rcmd) / tmd : 0.0;
                double rhs = -rd - wu - dt * incl_suffix(wd, lane);
                if (!valid) rhs = 0.0;
                dU = nt.solve(rhs, lane);
                if (!valid) dU = 0.0;
                const double dvd = dt * incl_prefix(dU, lane);
                dtpu = -rppu - dU; dtmu = -rpmu + dU;
                dlpu = -(rcpu + lpu * dtpu) / tpu; dlmu = -(rcmu + lmu * dtmu) / tmu;
                dtpd = -rppd - dvd; dtmd = -rpmd + dvd;
                dlpd = -(rcpd + lpd * dtpd) / tpd; dlmd = -(rcmd + lmd * dtmd) / tmd;
                if (!valid) { dtpu = 0; dtmu = 0; dlpu = 0; dlmu = 0; }
                if (!validd) { dtpd = 0; dtmd = 0; dlpd = 0; dlmd = 0; }
                double al = 1.0;
                if (dtpu < 0) al = fmin(al, -tpu / dtpu);
                if (dtmu < 0) al = fmin(al, -tmu / dtmu);
                if (dlpu < 0) al = fmin(al, -lpu / dlpu);
                if (dlmu < 0) al = fmin(al, -lmu / dlmu);
                if (dtpd < 0) al = fmin(al, -tpd / dtpd);
                if (dtmd < 0) al = fmin(al, -tmd / dtmd);
                if (dlpd < 0) al = fmin(al, -lpd / dlpd);
                if (dlmd < 0) al = fmin(al, -lmd / dlmd);
                alpha = wave_min(al);
                if (pass == 0) {
                    const double mua = wave_sum((tpu + alpha * dtpu) * (lpu + alpha * dlpu) +
                                                (tmu + alpha * dtmu) * (lmu + alpha * dlmu) +
                                                (tpd + alpha * dtpd) * (lpd + alpha * dlpd) +
                                                (tmd + alpha * dtmd) * (lmd + alpha * dlmd)) * inv_nc;
                    const double sg = mua / mu;
                    sigmu = sg * sg * sg * mu;
                }
            }
            alpha *= 0.995;
            U = fma(alpha, dU, U);
            tpu = fma(alpha, dtpu, tpu); tmu = fma(alpha, dtmu, tmu);
            lpu = fma(alpha, dlpu, lpu); lmu = fma(alpha, dlmu, lmu);
            tpd = fma(alpha, dtpd, tpd); tmd = fma(alpha, dtmd, tmd);
            lpd = fma(alpha, dlpd, lpd); lmd = fma(alpha, dlmd, lmd);
            iters++;
        }
    }

    // ---------------------------------------------------------------- 5. outputs, prediction, plant
    if (valid && a.U_out) a.U_out[(size_t)s * N + lane] = U;
    const double u0 = bcast(U, 0);
    if (lane == 0) {
        if (a.u0_out) a.u0_out[s] = u0;
        if (a.status_out) a.status_out[s] = status;
        if (a.iters_out) a.iters_out[s] = iters;
        if (a.stat_out) {
            double *so = a.stat_out + 4 * (size_t)s;
            so[0] = mu; so[1] = res; so[2] = (double)constrained; so[3] = fabs(u0);
        }
    }
    if (a.traj_out || cfg.kind == MPC_KIND_LPV) {
        // predicted states x_k (lane k keeps stage k; stage N falls to lane N % 64 on a second sweep)
        double x[6] = {0, 0, 0, xh3, xh4, xh5};
        double keep[6] = {0, 0, 0, xh3, xh4, xh5};
        double last[6];
#pragma unroll 1
        for (int k = 0; k < N; k++) {
            const double *st = &s_stage[k * STAGE_W];
            const double uk = __shfl(U, k);
            const double n0 = x[0] + st[0] * x[2] + st[1] * x[3] + st[8];
            const double n1 = x[1] + st[2] * x[2] + st[3] * x[3] + st[9];
            const double n2 = x[2] + dt * x[4];
            const double n3 = st[4] * x[3] + st[5] * x[4] + f35 * x[5] + c3;
            const double n4 = st[6] * x[3] + st[7] * x[4] + f45 * x[5];
            const double n5 = x[5] + dt * uk;
            x[0] = n0; x[1] = n1; x[2] = n2; x[3] = n3; x[4] = n4; x[5] = n5;
            if (lane == k + 1) {
#pragma unroll
                for (int c = 0; c < 6; c++) keep[c] = x[c];
            }
        }
#pragma unroll
        for (int c = 0; c < 6; c++) last[c] = x[c];
        if (a.traj_out) {
            // 7-column form [x, y, cos(psi), sin(psi), vy, r, delta] (LMPC2.py:337-345)
            if (lane <= N && lane < WAVE) {
                const double *q = (lane == N) ? last : keep;
                double *o = a.traj_out + ((size_t)s * (N + 1) + lane) * 7;
                o[0] = q[0]; o[1] = q[1]; o[2] = cos(q[2]); o[3] = sin(q[2]); o[4] = q[3]; o[5] = q[4]; o[6] = q[5];
            }
            if (N == WAVE && lane == 0) {
                double *o = a.traj_out + ((size_t)s * (N + 1) + N) * 7;
                o[0] = last[0]; o[1] = last[1]; o[2] = cos(last[2]); o[3] = sin(last[2]); o[4] = last[3];
                o[5] = last[4]; o[6] = last[5];
            }
        }
        if (cfg.kind == MPC_KIND_LPV) {
            // prev_x <- solver states (6 x (N+1)), prev_u <- inputs (LPVMPC.py:313-327)
            double *lx = a.lin_x + (size_t)s * 6 * (N + 1);
#pragma unroll
            for (int c = 0; c < 6; c++) {
                if (valid) lx[c * (N + 1) + lane] = keep[c];
                if (lane == 0) lx[c * (N + 1) + N] = last[c];
            }
            if (valid) a.lin_u[(size_t)s * N + lane] = U;
        }
    }
    if (a.mode != MODE_SOLVE) {
        const double oldx = plant[0];
        plant_rk4<8>(plant, u0, dt, mp);                                // skidpad_simulator.py:128
        if (lane < 8) {
            double val = plant[0];
#pragma unroll
            for (int i = 1; i < 8; i++) val = (lane == i) ? plant[i] : val;
            a.state[8 * (size_t)s + lane] = val;
        }
        if (a.mode == MODE_SKIDPAD && lane == 0 && plant[0] > SK_CENTER && oldx <= SK_CENTER)
            a.lap[s] = lapv + 1;                                        // skidpad_simulator.py:91-93
    }
}

// ------------------------------------------------------------------------------------------------
template <int N>
static hipError_t launch_n(const DevCfg &cfg, const SolveArgs &a, hipStream_t st)
{
    hipLaunchKernelGGL(k_mpc<N>, dim3(a.B), dim3(WAVE), 0, st, cfg, a);
    return hipGetLastError();
}

hipError_t launch_mpc(const DevCfg &cfg, const SolveArgs &a, hipStream_t st)
{
    switch (cfg.N) {
#define MPC_CASE(n) case n: return launch_n<n>(cfg, a, st);
        MPC_HORIZONS(MPC_CASE)
#undef MPC_CASE
    default: return hipErrorInvalidValue;
    }
}

bool horizon_supported(int N)
{
    switch (N) {
#define MPC_CASE(n) case n: return true;
        MPC_HORIZONS(MPC_CASE)
#undef MPC_CASE
    default: return false;
    }
}

}  // namespace mpc
