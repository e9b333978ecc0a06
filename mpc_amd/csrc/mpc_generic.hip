// Any-horizon variant of the wave-per-scenario MPC step (N <= 128) for gfx950.
//
// Same algorithm and semantics as k_mpc in mpc_solver.hip (plan -> stage model -> condense -> unconstrained solve ->
// interior point -> rollout -> plant/EKF), but the horizon is a run-time value: one wavefront owns a scenario, lane l
// owns stages l and l + 64 (R = 2; R = 1 for N <= 64), and the Newton matrix lives in LDS instead of registers
// (column-major, thread-contiguous, so the factorisation runs without bank conflicts and without unrolled code).
// It serves horizons that have no register-resident instantiation, e.g. the LPV controller's N = 80
// (plot_generator.py:604) or the N = 50..90 sweep of plot_generator.py:46-75.
#include "mpc_ipm.hpp"

#pragma clang fp contract(fast)

namespace mpc {

__device__ __forceinline__ double rl(double v, int src)   // readlane with a run-time (wave-uniform) lane index
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

template <int R>
struct Vec {
    double v[R];
    __device__ __forceinline__ double at(int i) const   // value of logical entry i (wave-uniform i)
    {
        if (R == 1) return rl(v[0], i);
        const double lo = rl(v[0], i & (WAVE - 1)), hi = rl(v[R - 1], i & (WAVE - 1));   // both halves, then a scalar select:
        return i < WAVE ? lo : hi;                                                        // no branch in the callers' loops
    }
};

template <int R>
__device__ __forceinline__ double rsel(const double (&x)[R], int i)   // logical entry i (wave-uniform) of a per-lane pair
{
    if (R == 1) return rl(x[0], i);
    const double lo = rl(x[0], i & (WAVE - 1)), hi = rl(x[R - 1], i & (WAVE - 1));
    return i < WAVE ? lo : hi;
}

template <int R>
__device__ __forceinline__ Vec<R> vprefix(const Vec<R> &x, int lane)   // inclusive prefix sums in logical order
{
    Vec<R> o;
    double carry = 0.0;
#pragma unroll
    for (int q = 0; q < R; q++) {
        o.v[q] = incl_prefix(x.v[q], lane) + carry;
        carry = rl(o.v[q], WAVE - 1);
    }
    return o;
}
template <int R>
__device__ __forceinline__ Vec<R> vsuffix(const Vec<R> &x, int lane)
{
    Vec<R> o;
    double carry = 0.0;
#pragma unroll
    for (int q = R - 1; q >= 0; q--) {
        o.v[q] = incl_suffix(x.v[q], lane) + carry;
        carry = rl(o.v[q], 0);
    }
    return o;
}
template <int R> __device__ __forceinline__ double vmaxabs(const Vec<R> &x)
{
    double m = 0.0;
#pragma unroll
    for (int q = 0; q < R; q++) m = fmax(m, fabs(x.v[q]));
    return wave_max(m);
}

// Packed lower triangle, column-major: entry (row i, column j), i >= j, sits at M[colp(j, N) + i]; column j holds its
// N - j entries contiguously, so the lanes of a wave read a column without bank conflicts, and the whole matrix takes
// N (N + 1) / 2 doubles (26 KB at N = 80: four waves per CU where a full square allows two).
__device__ __forceinline__ int colp(int j, int N) { return (j * (2 * N - j - 1)) >> 1; }

// LDL' of the symmetric matrix whose lower triangle sits in M (see colp).  Column k keeps c(i,k) = d(k) l(i,k) with d(k)
// on the diagonal; the backward substitution gathers row k of it (one entry per column).
// Left-looking in blocks of LDLW columns held in registers: every earlier pivot column is read from LDS once per block
// (one coalesced read per lane plus LDLW broadcast reads for LDLW * R updates) instead of one read-modify-write of LDS
// per updated entry.  s_dinv receives the reciprocal pivots.
constexpr int LDLW = 8;

template <int R>
__device__ __forceinline__ void lds_factor(double *M, double *s_dinv, int N, int lane, double (&dinv)[R])
{

    int ic[R];
#pragma unroll
    for (int q = 0; q < R; q++) { dinv[q] = 1.0; ic[q] = min(lane + WAVE * q, N - 1); }
#pragma unroll 1
    for (int jb = 0; jb < N; jb += LDLW) {
        double a[LDLW][R];
#pragma unroll
        for (int w = 0; w < LDLW; w++) {
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int i = lane + WAVE * q, j = jb + w;
                const double m = M[colp(min(j, N - 1), N) + min(i, N - 1)];      // always read (clamped), then select: no branch
                a[w][q] = (j < N && i >= j && i < N) ? m : 0.0;
            }
        }
        // Earlier pivots, KU at a time: all their LDS reads are issued first (the scheduler otherwise waits for every
        // broadcast read right where it is used), then the updates.  jb is a multiple of LDLW, hence of KU.
        constexpr int KU = 4;
#pragma unroll 1
        for (int k0 = 0; k0 < jb; k0 += KU) {
            // Entries that belong to no row or column of the matrix (i < jb, i >= N, jb + w >= N) are computed along with
            // the rest and never used: a lane's accumulators depend on its own reads only, and the reads stay inside the
            // LDS block (the pivot reciprocals follow the matrix).
            double ikv[KU], mi[KU][R], cm[KU][LDLW];
#pragma unroll
            for (int u = 0; u < KU; u++) {
                ikv[u] = s_dinv[k0 + u];
#pragma unroll
                for (int q = 0; q < R; q++) mi[u][q] = M[colp(k0 + u, N) + ic[q]];
#pragma unroll
                for (int w = 0; w < LDLW; w++) cm[u][w] = M[colp(k0 + u, N) + jb + w];   // wave-uniform address: LDS broadcast
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < KU; u++) {
#pragma unroll
                for (int q = 0; q < R; q++) {
                    const double li = mi[u][q] * ikv[u];
#pragma unroll
                    for (int w = 0; w < LDLW; w++) a[w][q] = fma(-li, cm[u][w], a[w][q]);
                }
            }
        }
        // The LDLW pivots of the block, without branches: a column or row beyond N computes on clamped reads into
        // accumulators nobody uses, and only the stores are guarded.
#pragma unroll
        for (int w = 0; w < LDLW; w++) {
            const int k = jb + w;
            const double pk = rsel<R>(a[w], min(k, N - 1));
            const double pivot = (k < N) ? pk : 1.0;
            const double inv = fast_rcp(pivot);
            double c[R], l[R];
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int i = lane + WAVE * q;
                c[q] = (i > k && i < N) ? a[w][q] : 0.0;
                l[q] = c[q] * inv;
                dinv[q] = (i == k && k < N) ? inv : dinv[q];
            }
#pragma unroll
            for (int w2 = w + 1; w2 < LDLW; w2++) {
                const double cjk = rsel<R>(c, min(jb + w2, N - 1));
#pragma unroll
                for (int q = 0; q < R; q++) a[w2][q] = fma(-l[q], cjk, a[w2][q]);
            }
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int i = lane + WAVE * q;
                if (i == k && k < N) { M[colp(k, N) + k] = pivot; s_dinv[k] = inv; }
                if (i > k && i < N) M[colp(min(k, N - 1), N) + i] = c[q];
            }
        }
        wave_sync();
    }
}

template <int R>
__device__ __forceinline__ Vec<R> lds_solve(const double *M, const double *s_dinv, int N, int lane,
                                            const double (&dinv)[R], Vec<R> b)
{
    Vec<R> w = b;
    constexpr int KU = 4;
    int ic[R], cpl[R];
#pragma unroll
    for (int q = 0; q < R; q++) { ic[q] = min(lane + WAVE * q, N - 1); cpl[q] = colp(ic[q], N); }
    // Forward.  The dependent chain per pivot is readlane -> multiply -> fma; column k and 1 / d(k) do not depend on
    // it, so the LDS reads of KU pivots are issued together ahead of their chain steps.
#pragma unroll 1
    for (int k0 = 0; k0 < N - 1; k0 += KU) {
        double dk[KU], m[KU][R];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int k = min(k0 + u, N - 1);
            dk[u] = s_dinv[k];
#pragma unroll
            for (int q = 0; q < R; q++) m[u][q] = M[colp(k, N) + ic[q]];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int k = k0 + u;
            const double wk = (k < N - 1) ? w.at(min(k, N - 1)) * dk[u] : 0.0;      // l(i,k) w_k = c(i,k) * (w_k / d_k)
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int i = lane + WAVE * q;
                w.v[q] = fma(-((i > k && i < N) ? m[u][q] : 0.0), wk, w.v[q]);
            }
        }
    }
    // Backward: x_i = (w_i - sum_{k>i} c(k,i) x_k) / d_i.
#pragma unroll 1
    for (int k0 = N - 1; k0 > 0; k0 -= KU) {
        double dk[KU], m[KU][R];
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int k = max(k0 - u, 0);
            dk[u] = s_dinv[k];
#pragma unroll
            for (int q = 0; q < R; q++) m[u][q] = M[cpl[q] + k];      // c(k,i), i = this lane's column: a gather along row k
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < KU; u++) {
            const int k = k0 - u;
            const double xk = (k > 0) ? w.at(max(k, 0)) * dk[u] : 0.0;
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int i = lane + WAVE * q;
                w.v[q] = fma(-((i < k) ? m[u][q] : 0.0), xk, w.v[q]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < R; q++) w.v[q] *= dinv[q];
    return w;
}

// One stage of the Hessian accumulation: M(i,j) += sum_c wv_c(j) * v_c(i) for j <= i < k, over NC components whose
// weighted columns start at s_wv + off[c].  Four columns per trip: their reads are issued together, then the updates,
// then the writes.  The columns j >= k of s_wv are zero (their lanes have not started), so everything is read and
// accumulated unconditionally; only the writes are guarded.  An unused slot (off = 0 with a zero v) adds exact zeros.
template <int R, int NC>
__device__ __forceinline__ void condense_columns(double *M, const double *s_wv, const int (&off)[NC], const double (&v)[R][NC],
                                                 int k, int N, int lane)
{
    constexpr int CW = 4;
#pragma unroll 1
    for (int j0 = 0; j0 < k; j0 += CW) {
        double wj[CW][NC], acc[CW][R];
#pragma unroll
        for (int u = 0; u < CW; u++) {
            const int j = min(j0 + u, N - 1);
#pragma unroll
            for (int c = 0; c < NC; c++) wj[u][c] = s_wv[off[c] + j];
#pragma unroll
            for (int q = 0; q < R; q++) acc[u][q] = M[colp(j, N) + min(lane + WAVE * q, N - 1)];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CW; u++) {
#pragma unroll
            for (int q = 0; q < R; q++) {
#pragma unroll
                for (int c = 0; c < NC; c++) acc[u][q] = fma(wj[u][c], v[q][c], acc[u][q]);
            }
        }
#pragma unroll
        for (int u = 0; u < CW; u++) {
            const int j = j0 + u;
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int i = lane + WAVE * q;
                if (j < k && i >= j && i < k) M[colp(j, N) + i] = acc[u][q];
            }
        }
    }
}

template <int R>
__global__ __launch_bounds__(WAVE) void k_mpc_any(const DevCfg cfg, const SolveArgs a)
{
    const int s = a.order ? a.order[blockIdx.x] : blockIdx.x;      // scenario this wave advances (grid = B workgroups)
    const int lane = threadIdx.x;
    const int N = cfg.N;
    const int NM = (N * (N + 1)) >> 1;    // doubles of the packed lower triangle
    extern __shared__ double lds[];
    double *s_ref = lds, *s_p = s_ref + 3 * (N + 1), *s_stage = s_p + (N + 1), *s_wv = s_stage + N * STAGE_W;
    double *s_We = s_wv + 6 * N, *M = s_We + 36, *s_dinv = M + NM, *s_ekf = s_dinv + N;
    if (lane < 36) s_We[lane] = a.cost_scen ? a.cost_scen[43 * (size_t)s + 7 + lane] : cfg.We[lane];

    ModelParams mp;
    if (a.mp_scen) {
        const double *q = a.mp_scen + 6 * (size_t)s;
        mp = {q[0], q[1], q[2], q[3], q[4], q[5]};
    } else
        mp = {cfg.mp[0], cfg.mp[1], cfg.mp[2], cfg.mp[3], cfg.mp[4], cfg.mp[5]};
    const double dt = cfg.dt, dt2 = dt * dt;
    const double umax = cfg.max_rate, dmax = cfg.max_steering;
    const double f35 = dt * (-mp.Cf / mp.m), f45 = dt * (-mp.lf * mp.Cf / mp.Iz);
    double wq[6];
#pragma unroll
    for (int c = 0; c < 6; c++) wq[c] = dt * (a.cost_scen ? a.cost_scen[43 * (size_t)s + c] : cfg.w[c]);
    const double wr = dt * (a.cost_scen ? a.cost_scen[43 * (size_t)s + 6] : cfg.w[6]);
    // components of the weighted prediction columns that can be non-zero: stage weights are diagonal, the terminal
    // weight is a full matrix (its column c matters)
    unsigned nz_stage = 0, nz_term = 0;
#pragma unroll
    for (int c = 0; c < 6; c++) {
        if (wq[c] != 0.0) nz_stage |= 1u << c;
        bool any = false;
        for (int d = 0; d < 6; d++) any = any || (a.cost_scen ? a.cost_scen[43 * (size_t)s + 7 + d * 6 + c] : cfg.We[d * 6 + c]) != 0.0;
        if (any) nz_term |= 1u << c;
    }
    nz_stage = __builtin_amdgcn_readfirstlane(nz_stage); nz_term = __builtin_amdgcn_readfirstlane(nz_term);
    const int n_stage = __popc(nz_stage), n_term = __popc(nz_term);
    auto nsel_k = [&](bool terminal) { return terminal ? n_term : n_stage; };
    bool valid[R], validd[R];
#pragma unroll
    for (int q = 0; q < R; q++) { valid[q] = lane + WAVE * q < N; validd[q] = lane + WAVE * q < N - 1; }

    double plant[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    double prevp = 0.0, tvel = 0.0;
    int lapv = 0;
    const int nxp = a.nx_plant;
    double *s_xe = s_ekf, *s_P = s_ekf + 10, *s_F = s_ekf + 110, *s_T = s_ekf + 210, *s_aug = s_ekf + 310;
    if (a.mode != MODE_SOLVE) {
#pragma unroll
        for (int i = 0; i < 10; i++) plant[i] = (i < nxp) ? a.state[(size_t)nxp * s + i] : 0.0;
        tvel = a.target_vel[s];
        if (a.mode == MODE_SKIDPAD) { lapv = a.lap[s]; prevp = a.prev_progress[s]; }
        if (a.mode == MODE_OFL) {
            for (int e = lane; e < 100; e += WAVE) s_P[e] = a.ekf_P[(size_t)s * 100 + e];
            if (lane < 10) s_xe[lane] = a.ekf_x[(size_t)s * 10 + lane];
        }
    }
    wave_sync();
    const PlantCoef pc = plant_coef(mp, plant[4]);             // constant over the launch: x[4] has zero derivative
    double cs[3];                                              // progress increments of waypoints lane, lane+64, lane+128
#pragma unroll
    for (int q = 0; q < 3; q++) cs[q] = (lane + WAVE * q <= N) ? speed_cumsum(tvel, lane + WAVE * q) : 0.0;
    const int nsteps = (a.mode == MODE_SOLVE) ? 1 : a.nsteps;
    int iters_tot = 0, status_max = 0, nconstr = 0, lap_flag = 0;
    double mu = 0.0, res = 0.0, u0 = 0.0;
    double *Hg = a.Hws + (size_t)s * N * N;
    // closed loops start the interior point from the previous step's solve when that step needed one too (record:
    // inputs and the four multiplier vectors, [5][WS], then a validity flag; see interior_point())
    constexpr int WS = 2 * WAVE;
    double *warm_rec = (a.ipm_ws && a.mode != MODE_SOLVE) ? a.ipm_ws + (size_t)s * IPM_WARM_DOUBLES : nullptr;
    bool warm_ok = warm_rec && warm_rec[5 * WS] == 1.0;
    // closed-loop statistics of the scenario
    const bool cl_on = a.stat_out != nullptr && a.mode != MODE_SOLVE;
    double cl_acc = (cl_on && a.accumulate && lane < CL_STAT_W) ? a.stat_out[STAT_W * (size_t)s + STAT_CL + lane] : 0.0;   // lane i: statistic i

#pragma unroll 1
    for (int t = 0; t < nsteps; t++) {
        // ------------------------------------------------------------ 1. initial state and references
        double xh3, xh4, xh5, dhat = 0.0;
        if (a.mode == MODE_SOLVE) {
            const double *x0 = a.x0 + 7 * (size_t)s;
            xh3 = x0[4]; xh4 = x0[5]; xh5 = x0[6];
            if (a.d_hat) dhat = a.d_hat[s];
            for (int k = lane; k <= N; k += WAVE) {
                const double *w = a.wp + ((size_t)s * (N + 1) + k) * 4;
                s_ref[k] = w[0]; s_ref[(N + 1) + k] = w[1]; s_ref[2 * (N + 1) + k] = w[3];
                s_p[k] = a.p[(size_t)s * (N + 1) + k];
            }
        } else {
            double px = plant[0], py = plant[1], hc = plant[2], hs = plant[3];
            xh3 = plant[5]; xh4 = plant[6]; xh5 = plant[7];
            if (a.mode == MODE_OFL) {
                px = s_xe[0]; py = s_xe[1]; hc = s_xe[2]; hs = s_xe[3];
                xh3 = s_xe[5]; xh4 = s_xe[6]; xh5 = s_xe[7]; dhat = s_xe[9];
            }
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
            unit_heading(hc, hs, ch, sh);
#pragma unroll
            for (int qi = 0; qi < 3; qi++) {
                const int k = lane + WAVE * qi;
                if (k > N) continue;
                double o[4];
                const double q = (k == 0) ? p0 : p0 + cs[qi] * dt;
                if (a.mode == MODE_SKIDPAD) sk_pose(q, o);
                else step_pose(q, a.step_ramp, o);
                to_car_frame(o, px, py, ch, sh);
                s_ref[k] = o[0]; s_ref[(N + 1) + k] = o[1]; s_ref[2 * (N + 1) + k] = o[3];
                s_p[k] = tvel;
                if (a.wp_out) {
                    double *w = a.wp_out + ((size_t)s * (N + 1) + k) * 4;
                    w[0] = o[0]; w[1] = o[1]; w[2] = o[2]; w[3] = o[3];
                }
            }
        }
        wave_sync();
        double cl_err = 0.0;               // lane i < 6: component i of [x0; u0] - yref_0 (the input joins after the solve)
        if (cl_on) {
            cl_err = lane < 3 ? s_ref[lane * (N + 1)] : 0.0;
            cl_err = lane == 3 ? xh3 : lane == 4 ? xh4 : lane == 5 ? xh5 : cl_err;
        }

        // ------------------------------------------------------------ 2. per-stage model
        const double c3 = (cfg.kind == MPC_KIND_OFL) ? dt * dhat : 0.0;
        for (int k = lane; k < N; k += WAVE)
            stage_model(cfg, mp, s_p[k], a.lin_x ? a.lin_x + (size_t)s * 6 * (N + 1) : nullptr, N, k, &s_stage[k * STAGE_W]);
        for (int e = lane; e < NM; e += WAVE) M[e] = 0.0;
        wave_sync();
        // ------------------------------------------------------------ 3. condense into M (lower triangle) and g
        Vec<R> g;
        {
            double v[R][6], xf[6] = {0, 0, 0, xh3, xh4, xh5};
#pragma unroll
            for (int q = 0; q < R; q++) {
                g.v[q] = 0.0;
#pragma unroll
                for (int c = 0; c < 6; c++) v[q][c] = 0.0;
            }
#pragma unroll 1
            for (int k = 1; k <= N; k++) {
                const double *st = &s_stage[(k - 1) * STAGE_W];
                const double e02 = st[0], e03 = st[1], e12 = st[2], e13 = st[3], f33 = st[4], f34 = st[5], f43 = st[6],
                             f44 = st[7], c0 = st[8], c1 = st[9];
                {
                    const double n0 = xf[0] + e02 * xf[2] + e03 * xf[3] + c0;
                    const double n1 = xf[1] + e12 * xf[2] + e13 * xf[3] + c1;
                    const double n2 = xf[2] + dt * xf[4];
                    const double n3 = f33 * xf[3] + f34 * xf[4] + f35 * xf[5] + c3;
                    const double n4 = f43 * xf[3] + f44 * xf[4] + f45 * xf[5];
                    xf[0] = n0; xf[1] = n1; xf[2] = n2; xf[3] = n3; xf[4] = n4;
                }
                double err[6] = {xf[0], xf[1], xf[2], xf[3], xf[4], xf[5]};
                if (!(k == N && cfg.kind == MPC_KIND_LPV)) {
                    err[0] -= s_ref[k]; err[1] -= s_ref[(N + 1) + k]; err[2] -= s_ref[2 * (N + 1) + k];
                }
#pragma unroll
                for (int q = 0; q < R; q++) {
                    const int i = lane + WAVE * q;
                    double *vv = v[q];
                    const double n0 = vv[0] + e02 * vv[2] + e03 * vv[3];
                    const double n1 = vv[1] + e12 * vv[2] + e13 * vv[3];
                    const double n2 = vv[2] + dt * vv[4];
                    const double n3 = f33 * vv[3] + f34 * vv[4] + f35 * vv[5];
                    const double n4 = f43 * vv[3] + f44 * vv[4] + f45 * vv[5];
                    vv[0] = n0; vv[1] = n1; vv[2] = n2; vv[3] = n3; vv[4] = n4;
                    if (i == k - 1) { vv[0] = 0; vv[1] = 0; vv[2] = 0; vv[3] = 0; vv[4] = 0; vv[5] = dt; }
                    double wv[6];
                    if (k < N) {
#pragma unroll
                        for (int c = 0; c < 6; c++) wv[c] = wq[c] * vv[c];
                    } else {
#pragma unroll
                        for (int c = 0; c < 6; c++) {
                            double tt = 0.0;
#pragma unroll
                            for (int d = 0; d < 6; d++) tt = fma(s_We[d * 6 + c], vv[d], tt);
                            wv[c] = tt;
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 6; c++) g.v[q] = fma(wv[c], err[c], g.v[q]);
                    if (i < N) {
#pragma unroll
                        for (int c = 0; c < 6; c++) s_wv[c * N + i] = wv[c];
                    }
                }
                wave_sync();
                // M(i,j) += (W v_j)' v_i for j <= i < k (entries with i or j >= k are still zero).  Only the components
                // with a non-zero weight contribute; when there are at most three of them (e.g. the LPV controller's
                // weights on y, psi and delta) they are gathered into the first three slots and the short loop runs.
                if (nsel_k(k == N) <= 3) {
                    const unsigned nz = (k == N) ? nz_term : nz_stage;
                    int off[3] = {0, 0, 0};
                    double vs[R][3];
#pragma unroll
                    for (int q = 0; q < R; q++) { vs[q][0] = 0.0; vs[q][1] = 0.0; vs[q][2] = 0.0; }
                    int t = 0;
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        const bool on = (nz >> c) & 1u;                  // wave-uniform
#pragma unroll
                        for (int slot = 0; slot < 3; slot++) {
                            const bool here = on && t == slot;
                            off[slot] = here ? c * N : off[slot];
#pragma unroll
                            for (int q = 0; q < R; q++) vs[q][slot] = here ? v[q][c] : vs[q][slot];
                        }
                        t += on ? 1 : 0;
                    }
                    condense_columns<R, 3>(M, s_wv, off, vs, k, N, lane);
                } else {
                    const int off[6] = {0, N, 2 * N, 3 * N, 4 * N, 5 * N};
                    condense_columns<R, 6>(M, s_wv, off, v, k, N, lane);
                }
                wave_sync();
            }
        }
#pragma unroll
        for (int q = 0; q < R; q++) {
            const int i = lane + WAVE * q;
            if (i < N) M[colp(i, N) + i] += wr;
            else g.v[q] = 0.0;
        }
        wave_sync();
        // keep H in the workspace (full symmetric, row stride N) for the interior-point iterations
        for (int j = 0; j < N; j++) {
#pragma unroll
            for (int q = 0; q < R; q++) {
                const int i = lane + WAVE * q;
                if (i >= j && i < N) { const double h = M[colp(j, N) + i]; Hg[j * N + i] = h; Hg[i * N + j] = h; }
            }
        }
        __threadfence_block();
        wave_sync();
        // ------------------------------------------------------------ 4. solve
        double up_d[R], lo_d[R];
        bool relaxed = false;
#pragma unroll
        for (int q = 0; q < R; q++) steer_bounds(xh5, dmax, umax, dt, lane + WAVE * q, up_d[q], lo_d[q], relaxed);
        double dinv[R];
        lds_factor<R>(M, s_dinv, N, lane, dinv);
        Vec<R> U;
        int iters = 0, status = 0, constrained = 0;
        {
            Vec<R> rhs;
#pragma unroll
            for (int q = 0; q < R; q++) rhs.v[q] = -g.v[q];
            Vec<R> Uu = lds_solve<R>(M, s_dinv, N, lane, dinv, rhs);
#pragma unroll
            for (int q = 0; q < R; q++) if (!valid[q]) Uu.v[q] = 0.0;
            const Vec<R> vd = vprefix<R>(Uu, lane);
            bool ok = true;
#pragma unroll
            for (int q = 0; q < R; q++) {
                ok = ok && fabs(Uu.v[q]) <= umax;
                if (validd[q]) ok = ok && (dt * vd.v[q] <= up_d[q]) && (dt * vd.v[q] >= lo_d[q]);
            }
            if (__all(ok)) U = Uu;
            else constrained = 1;
        }
        if (constrained) {
            const double gscale = fmax(1.0, vmaxabs<R>(g));
            const double lam0 = fmax(1.0, 0.01 * gscale);
            Vec<R> tpu, tmu, lpu, lmu, tpd, tmd, lpd, lmd;
            auto cold_start = [&]() {
#pragma unroll
                for (int q = 0; q < R; q++) {
                    U.v[q] = 0.0;
                    tpu.v[q] = valid[q] ? umax : 1.0; tmu.v[q] = tpu.v[q];
                    lpu.v[q] = valid[q] ? lam0 : 0.0; lmu.v[q] = lpu.v[q];
                    tpd.v[q] = validd[q] ? fmax(up_d[q], 0.1 * dmax) : 1.0; tmd.v[q] = validd[q] ? fmax(-lo_d[q], 0.1 * dmax) : 1.0;
                    lpd.v[q] = validd[q] ? lam0 : 0.0; lmd.v[q] = lpd.v[q];
                }
            };
            cold_start();
            bool warm = warm_ok;
            if (warm) {
                // the previous solve moved one stage ahead, slacks pushed back inside, multipliers lifted to a common
                // small complementarity target (same rule as interior_point())
                const double delta = 3e-3;
                double a0[R], a1[R], a2[R], a3[R];
#pragma unroll
                for (int q = 0; q < R; q++) {
                    const int i = lane + WAVE * q, ii = min(i, N - 1);
                    const int iu = (i < N - 1) ? ii + 1 : ii, id = (i < N - 2) ? ii + 1 : ii;
                    const double u0 = warm_rec[iu];
                    a0[q] = warm_rec[WS + iu]; a1[q] = warm_rec[2 * WS + iu];
                    a2[q] = warm_rec[3 * WS + id]; a3[q] = warm_rec[4 * WS + id];
                    U.v[q] = valid[q] ? fmin(fmax(u0, -umax), umax) : 0.0;
                }
                const Vec<R> vd0 = vprefix<R>(U, lane);
                const double mu0 = delta * 0.01 * gscale;
#pragma unroll
                for (int q = 0; q < R; q++) {
                    if (valid[q]) {
                        tpu.v[q] = fmax(umax - U.v[q], delta * 2.0 * umax); tmu.v[q] = fmax(umax + U.v[q], delta * 2.0 * umax);
                        lpu.v[q] = fmax(a0[q], mu0 / tpu.v[q]); lmu.v[q] = fmax(a1[q], mu0 / tmu.v[q]);
                    }
                    if (validd[q]) {
                        tpd.v[q] = fmax(up_d[q] - dt * vd0.v[q], delta * 2.0 * dmax); tmd.v[q] = fmax(dt * vd0.v[q] - lo_d[q], delta * 2.0 * dmax);
                        lpd.v[q] = fmax(a2[q], mu0 / tpd.v[q]); lmd.v[q] = fmax(a3[q], mu0 / tmd.v[q]);
                    }
                }
            }
            const double inv_nc = 1.0 / (2.0 * (2 * N - 1));
            double res_prev = 1e300, mu_last = 0.0, res_last = 0.0;
            bool acc_last = false, rec_ok = true;
            Vec<R> U_last = U;
#pragma unroll 1
            for (;;) {
                // One pass over H (workspace, row j = column j) serves both the residual's H U and the Newton matrix
                // K = H + diag(Du) + dt^2 sfx[max(i,j)] of this iterate (lower triangle into M; on the lower triangle
                // max(i,j) = i).  Eight rows are requested together so that their memory latencies overlap.
                Vec<R> Du, Dd;
#pragma unroll
                for (int q = 0; q < R; q++) {
                    Du.v[q] = lpu.v[q] / tpu.v[q] + lmu.v[q] / tmu.v[q];
                    Dd.v[q] = validd[q] ? lpd.v[q] / tpd.v[q] + lmd.v[q] / tmd.v[q] : 0.0;
                }
                const Vec<R> sfx = vsuffix<R>(Dd, lane);
                Vec<R> hu;
#pragma unroll
                for (int q = 0; q < R; q++) hu.v[q] = 0.0;
                constexpr int JU = 8;
#pragma unroll 1
                for (int j0 = 0; j0 < N; j0 += JU) {
                    double h[JU][R];
#pragma unroll
                    for (int u = 0; u < JU; u++) {
#pragma unroll
                        for (int q = 0; q < R; q++) h[u][q] = Hg[min(j0 + u, N - 1) * N + min(lane + WAVE * q, N - 1)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < JU; u++) {
                        const int j = j0 + u;
                        const double uj = (j < N) ? U.at(min(j, N - 1)) : 0.0;
#pragma unroll
                        for (int q = 0; q < R; q++) {
                            const int i = lane + WAVE * q;
                            hu.v[q] = fma(valid[q] ? h[u][q] : 0.0, uj, hu.v[q]);
                            if (j < N && i >= j && i < N) M[colp(j, N) + i] = fma(dt2, sfx.v[q], h[u][q]) + (i == j ? Du.v[q] : 0.0);
                        }
                    }
                }
                const Vec<R> vd = vprefix<R>(U, lane);
                Vec<R> lamd;
#pragma unroll
                for (int q = 0; q < R; q++) lamd.v[q] = validd[q] ? lpd.v[q] - lmd.v[q] : 0.0;
                const Vec<R> sl = vsuffix<R>(lamd, lane);
                Vec<R> rd, rppu, rpmu, rppd, rpmd;
                double musum = 0.0, rpmax = 0.0;
#pragma unroll
                for (int q = 0; q < R; q++) {
                    rd.v[q] = valid[q] ? hu.v[q] + g.v[q] + (lpu.v[q] - lmu.v[q]) + dt * sl.v[q] : 0.0;
                    rppu.v[q] = valid[q] ? U.v[q] + tpu.v[q] - umax : 0.0; rpmu.v[q] = valid[q] ? -U.v[q] + tmu.v[q] - umax : 0.0;
                    rppd.v[q] = validd[q] ? dt * vd.v[q] + tpd.v[q] - up_d[q] : 0.0;
                    rpmd.v[q] = validd[q] ? -dt * vd.v[q] + tmd.v[q] + lo_d[q] : 0.0;
                    musum += tpu.v[q] * lpu.v[q] + tmu.v[q] * lmu.v[q] + tpd.v[q] * lpd.v[q] + tmd.v[q] * lmd.v[q];
                    rpmax = fmax(rpmax, fmax(fmax(fabs(rppu.v[q]), fabs(rpmu.v[q])), fmax(fabs(rppd.v[q]), fabs(rpmd.v[q]))));
                }
                mu = wave_sum(musum) * inv_nc;
                res = fmax(vmaxabs<R>(rd) / gscale, wave_max(rpmax));
                if (warm && iters == 0 && res > IPM_WARM_RES_MAX) {
                    // the problem moved too far from the previous one: fall back to the cold start
                    warm = false;
                    cold_start();
                    U_last = U;
                    continue;
                }
                // exit rules: see interior_point()
                const bool finite = fabs(mu) < 1e300;
                const bool acc = finite && mu <= 10.0 * cfg.tol_mu && res <= 1e3 * cfg.tol_res;
                if (acc_last && !acc) { U = U_last; mu = mu_last; res = res_last; rec_ok = false; break; }
                if (finite && mu <= cfg.tol_mu && res <= cfg.tol_res) break;
                if (acc && (res >= 0.5 * res_prev || mu <= 1e-3 * cfg.tol_mu)) break;
                if (!finite || mu <= 1e-3 * cfg.tol_mu || iters >= cfg.max_iter) {
                    if (acc) break;
                    if (warm && !(iters >= cfg.max_iter)) {      // once more from the cold start
                        warm = false;
                        cold_start();
                        U_last = U;
                        res_prev = 1e300; acc_last = false;
                        iters++;
                        continue;
                    }
                    if (!finite) { U = U_last; mu = mu_last; res = res_last; }
                    rec_ok = false; status = 2;
                    break;
                }
                U_last = U; mu_last = mu; res_last = res; acc_last = acc;
                res_prev = res;
                wave_sync();
                lds_factor<R>(M, s_dinv, N, lane, dinv);
                Vec<R> dU, dtpu, dtmu, dlpu, dlmu, dtpd, dtmd, dlpd, dlmd;
#pragma unroll
                for (int q = 0; q < R; q++) {
                    dU.v[q] = 0; dtpu.v[q] = 0; dtmu.v[q] = 0; dlpu.v[q] = 0; dlmu.v[q] = 0;
                    dtpd.v[q] = 0; dtmd.v[q] = 0; dlpd.v[q] = 0; dlmd.v[q] = 0;
                }
                const bool safe = iters >= IPM_SAFE_AFTER;      // long-step path following, see interior_point()
                double sigmu = safe ? 0.25 * mu : 0.0, alpha = 1.0;
#pragma unroll 1
                for (int pass = 0; pass < 2; pass++) {
                    Vec<R> rcpu, rcmu, rcpd, rcmd, wd, rhs;
#pragma unroll
                    for (int q = 0; q < R; q++) {
                        rcpu.v[q] = tpu.v[q] * lpu.v[q]; rcmu.v[q] = tmu.v[q] * lmu.v[q];
                        rcpd.v[q] = tpd.v[q] * lpd.v[q]; rcmd.v[q] = tmd.v[q] * lmd.v[q];
                        if (safe) {
                            rcpu.v[q] -= sigmu; rcmu.v[q] -= sigmu; rcpd.v[q] -= sigmu; rcmd.v[q] -= sigmu;
                        } else if (pass) {
                            rcpu.v[q] += dtpu.v[q] * dlpu.v[q] - sigmu; rcmu.v[q] += dtmu.v[q] * dlmu.v[q] - sigmu;
                            rcpd.v[q] += dtpd.v[q] * dlpd.v[q] - sigmu; rcmd.v[q] += dtmd.v[q] * dlmd.v[q] - sigmu;
                        }
                        wd.v[q] = validd[q] ? (lpd.v[q] * rppd.v[q] - rcpd.v[q]) / tpd.v[q] - (lmd.v[q] * rpmd.v[q] - rcmd.v[q]) / tmd.v[q] : 0.0;
                    }
                    const Vec<R> sw = vsuffix<R>(wd, lane);
#pragma unroll
                    for (int q = 0; q < R; q++) {
                        const double wu = (lpu.v[q] * rppu.v[q] - rcpu.v[q]) / tpu.v[q] - (lmu.v[q] * rpmu.v[q] - rcmu.v[q]) / tmu.v[q];
                        rhs.v[q] = valid[q] ? -rd.v[q] - wu - dt * sw.v[q] : 0.0;
                    }
                    dU = lds_solve<R>(M, s_dinv, N, lane, dinv, rhs);
#pragma unroll
                    for (int q = 0; q < R; q++) if (!valid[q]) dU.v[q] = 0.0;
                    const Vec<R> dvd = vprefix<R>(dU, lane);
                    double al = 1.0;
#pragma unroll
                    for (int q = 0; q < R; q++) {
                        dtpu.v[q] = -rppu.v[q] - dU.v[q]; dtmu.v[q] = -rpmu.v[q] + dU.v[q];
                        dlpu.v[q] = -(rcpu.v[q] + lpu.v[q] * dtpu.v[q]) / tpu.v[q]; dlmu.v[q] = -(rcmu.v[q] + lmu.v[q] * dtmu.v[q]) / tmu.v[q];
                        dtpd.v[q] = -rppd.v[q] - dt * dvd.v[q]; dtmd.v[q] = -rpmd.v[q] + dt * dvd.v[q];
                        dlpd.v[q] = -(rcpd.v[q] + lpd.v[q] * dtpd.v[q]) / tpd.v[q]; dlmd.v[q] = -(rcmd.v[q] + lmd.v[q] * dtmd.v[q]) / tmd.v[q];
                        if (!valid[q]) { dtpu.v[q] = 0; dtmu.v[q] = 0; dlpu.v[q] = 0; dlmu.v[q] = 0; }
                        if (!validd[q]) { dtpd.v[q] = 0; dtmd.v[q] = 0; dlpd.v[q] = 0; dlmd.v[q] = 0; }
                        if (dtpu.v[q] < 0) al = fmin(al, -tpu.v[q] / dtpu.v[q]);
                        if (dtmu.v[q] < 0) al = fmin(al, -tmu.v[q] / dtmu.v[q]);
                        if (dlpu.v[q] < 0) al = fmin(al, -lpu.v[q] / dlpu.v[q]);
                        if (dlmu.v[q] < 0) al = fmin(al, -lmu.v[q] / dlmu.v[q]);
                        if (dtpd.v[q] < 0) al = fmin(al, -tpd.v[q] / dtpd.v[q]);
                        if (dtmd.v[q] < 0) al = fmin(al, -tmd.v[q] / dtmd.v[q]);
                        if (dlpd.v[q] < 0) al = fmin(al, -lpd.v[q] / dlpd.v[q]);
                        if (dlmd.v[q] < 0) al = fmin(al, -lmd.v[q] / dlmd.v[q]);
                    }
                    alpha = wave_min(al);
                    if (safe) break;
                    if (pass == 0) {
                        double ms = 0.0;
#pragma unroll
                        for (int q = 0; q < R; q++)
                            ms += (tpu.v[q] + alpha * dtpu.v[q]) * (lpu.v[q] + alpha * dlpu.v[q]) + (tmu.v[q] + alpha * dtmu.v[q]) * (lmu.v[q] + alpha * dlmu.v[q]) +
                                  (tpd.v[q] + alpha * dtpd.v[q]) * (lpd.v[q] + alpha * dlpd.v[q]) + (tmd.v[q] + alpha * dtmd.v[q]) * (lmd.v[q] + alpha * dlmd.v[q]);
                        const double sg = wave_sum(ms) * inv_nc / mu;
                        sigmu = fmax(sg * sg * sg * mu, 1e-3 * fmin(mu, res));   // the gap must not outrun the residuals
                    }
                }
                if (!safe) {
                    alpha *= fmin(fmax(0.995, 1.0 - mu), 0.999999);   // never onto the boundary itself
                } else {
                    const double big = 1e300;
                    double mp = big;
#pragma unroll
                    for (int q = 0; q < R; q++) {
                        if (valid[q]) mp = fmin(mp, fmin(tpu.v[q] * lpu.v[q], tmu.v[q] * lmu.v[q]));
                        if (validd[q]) mp = fmin(mp, fmin(tpd.v[q] * lpd.v[q], tmd.v[q] * lmd.v[q]));
                    }
                    const double gam = fmin(0.01, 0.5 * wave_min(mp) / mu);
                    alpha *= 0.99;
#pragma unroll 1
                    for (int bt = 0; bt < 20; bt++) {
                        double sum = 0.0, mn = big;
#pragma unroll
                        for (int q = 0; q < R; q++) {
                            const double p1 = (tpu.v[q] + alpha * dtpu.v[q]) * (lpu.v[q] + alpha * dlpu.v[q]), p2 = (tmu.v[q] + alpha * dtmu.v[q]) * (lmu.v[q] + alpha * dlmu.v[q]);
                            const double p3 = (tpd.v[q] + alpha * dtpd.v[q]) * (lpd.v[q] + alpha * dlpd.v[q]), p4 = (tmd.v[q] + alpha * dtmd.v[q]) * (lmd.v[q] + alpha * dlmd.v[q]);
                            sum += p1 + p2 + p3 + p4;
                            if (valid[q]) mn = fmin(mn, fmin(p1, p2));
                            if (validd[q]) mn = fmin(mn, fmin(p3, p4));
                        }
                        if (wave_min(mn) >= gam * wave_sum(sum) * inv_nc) break;
                        alpha *= 0.6;
                    }
                }
#pragma unroll
                for (int q = 0; q < R; q++) {
                    U.v[q] = fma(alpha, dU.v[q], U.v[q]);
                    tpu.v[q] = fma(alpha, dtpu.v[q], tpu.v[q]); tmu.v[q] = fma(alpha, dtmu.v[q], tmu.v[q]);
                    lpu.v[q] = fma(alpha, dlpu.v[q], lpu.v[q]); lmu.v[q] = fma(alpha, dlmu.v[q], lmu.v[q]);
                    tpd.v[q] = fma(alpha, dtpd.v[q], tpd.v[q]); tmd.v[q] = fma(alpha, dtmd.v[q], tmd.v[q]);
                    lpd.v[q] = fma(alpha, dlpd.v[q], lpd.v[q]); lmd.v[q] = fma(alpha, dlmd.v[q], lmd.v[q]);
                }
                iters++;
            }
            if (warm_rec && rec_ok) {
#pragma unroll
                for (int q = 0; q < R; q++) {
                    const int i = lane + WAVE * q;
                    if (valid[q]) {
                        warm_rec[i] = U.v[q]; warm_rec[WS + i] = lpu.v[q]; warm_rec[2 * WS + i] = lmu.v[q];
                        warm_rec[3 * WS + i] = lpd.v[q]; warm_rec[4 * WS + i] = lmd.v[q];
                    }
                }
            }
            warm_ok = warm_rec != nullptr && status == 0 && rec_ok;
        } else
            warm_ok = false;

        if (relaxed) status = max(status, 4);
        status = max(status, lap_flag);
        // ------------------------------------------------------------ 5. outputs, prediction, plant, estimator
#pragma unroll
        for (int q = 0; q < R; q++)
            if (valid[q] && a.U_out) a.U_out[(size_t)s * N + lane + WAVE * q] = U.v[q];
        u0 = rl(U.v[0], 0);
        iters_tot += iters; status_max = max(status_max, status); nconstr += constrained;
        if (a.traj_out || cfg.kind == MPC_KIND_LPV) {
            double x[6] = {0, 0, 0, xh3, xh4, xh5};
            double *lx = cfg.kind == MPC_KIND_LPV ? a.lin_x + (size_t)s * 6 * (N + 1) : nullptr;
            for (int k = 0; k <= N; k++) {
                if (lane == 0) {          // one lane writes stage k; the rollout itself is wave-uniform
                    if (a.traj_out) {
                        double *o = a.traj_out + ((size_t)s * (N + 1) + k) * 7;
                        o[0] = x[0]; o[1] = x[1]; o[2] = cos(x[2]); o[3] = sin(x[2]); o[4] = x[3]; o[5] = x[4]; o[6] = x[5];
                    }
                    if (lx) {
#pragma unroll
                        for (int c = 0; c < 6; c++) lx[c * (N + 1) + k] = x[c];
                    }
                }
                if (k == N) break;
                const double *st = &s_stage[k * STAGE_W];
                const double uk = U.at(k);
                const double n0 = x[0] + st[0] * x[2] + st[1] * x[3] + st[8];
                const double n1 = x[1] + st[2] * x[2] + st[3] * x[3] + st[9];
                const double n2 = x[2] + dt * x[4];
                const double n3 = st[4] * x[3] + st[5] * x[4] + f35 * x[5] + c3;
                const double n4 = st[6] * x[3] + st[7] * x[4] + f45 * x[5];
                const double n5 = x[5] + dt * uk;
                x[0] = n0; x[1] = n1; x[2] = n2; x[3] = n3; x[4] = n4; x[5] = n5;
            }
            if (lx) {
#pragma unroll
                for (int q = 0; q < R; q++)
                    if (valid[q]) a.lin_u[(size_t)s * N + lane + WAVE * q] = U.v[q];
            }
        }
        if (a.mode != MODE_SOLVE) {
            const double oldx = plant[0];
            if (cl_on) cl_acc = cl_step(cl_acc, lane, cl_value(lane, lane == 6 ? u0 : cl_err, cl_pair_violation(plant[2], plant[3]), u0, plant[7], 0.0));
            plant_rk4_c<10>(plant, u0, dt, pc);
            if (a.mode == MODE_SKIDPAD && plant[0] > SK_CENTER && oldx <= SK_CENTER) lapv += 1;
            if (a.u_hist && lane == 0) a.u_hist[(size_t)t * a.B + s] = u0;
            if (a.x_hist && lane < nxp) {
                double val = plant[0];
#pragma unroll
                for (int i = 1; i < 10; i++) val = (lane == i) ? plant[i] : val;
                a.x_hist[((size_t)t * a.B + s) * nxp + lane] = val;
            }
            if (a.mode == MODE_OFL) {
                ekf_time_update_wave(lane, s_xe, s_P, s_F, s_T, u0, dt, mp, a.ekf_q);
                double y[7];
#pragma unroll
                for (int c = 0; c < 7; c++) {
                    y[c] = plant[c < 5 ? c : c + 1];
                    if (a.meas_noise) y[c] += a.meas_noise[((size_t)t * a.B + s) * 7 + c];
                }
                ekf_measurement_update_wave(lane, s_xe, s_P, s_T, s_aug, y, a.ekf_r);
                if (a.est_hist && lane < 10) a.est_hist[((size_t)t * a.B + s) * 10 + lane] = s_xe[lane];
            }
        }
        __threadfence_block();
        wave_sync();
    }

    if (a.mode != MODE_SOLVE) {
        if (lane < nxp) {
            double val = plant[0];
#pragma unroll
            for (int i = 1; i < 10; i++) val = (lane == i) ? plant[i] : val;
            a.state[(size_t)nxp * s + lane] = val;
        }
        if (a.mode == MODE_SKIDPAD && lane == 0) { a.lap[s] = lapv; a.prev_progress[s] = prevp; }
        if (a.mode == MODE_OFL) {
            for (int e = lane; e < 100; e += WAVE) a.ekf_P[(size_t)s * 100 + e] = s_P[e];
            if (lane < 10) a.ekf_x[(size_t)s * 10 + lane] = s_xe[lane];
        }
    }
    if (lane == 0) {
        if (warm_rec) warm_rec[5 * WS] = warm_ok ? 1.0 : 0.0;
        if (a.u0_out) a.u0_out[s] = u0;
        if (a.work_out) a.work_out[s] = iters_tot;
        const int st_all = a.accumulate ? max(a.status_own[s], status_max) : status_max;
        const int it_all = a.accumulate ? a.iters_own[s] + iters_tot : iters_tot;
        a.status_own[s] = st_all; a.iters_own[s] = it_all;
        if (a.status_out) a.status_out[s] = st_all;
        if (a.iters_out) a.iters_out[s] = it_all;
        if (a.stat_out) {
            double *so = a.stat_out + STAT_W * (size_t)s;
            so[0] = mu; so[1] = res; so[2] = (a.accumulate ? so[2] : 0.0) + (double)nconstr; so[3] = fabs(u0);
        }
    }
    if (cl_on && lane < CL_STAT_W) a.stat_out[STAT_W * (size_t)s + STAT_CL + lane] = cl_acc;
}

static size_t any_lds_bytes(int N, int mode)
{
    return sizeof(double) * ((size_t)4 * (N + 1) + (size_t)N * STAGE_W + 6 * (size_t)N + 36 + (size_t)(N * (N + 1) / 2) + (size_t)N +
                             (mode == MODE_OFL ? EKF_LDS : 0));
}

hipError_t launch_mpc_any(const DevCfg &cfg, const SolveArgs &a, hipStream_t st)
{
    if (cfg.N < 2 || cfg.N > MPC_MAX_HORIZON) return hipErrorInvalidValue;
    const size_t lds = any_lds_bytes(cfg.N, a.mode);
    static bool lds_ok1[MPC_MAX_DEVICES] = {}, lds_ok2[MPC_MAX_DEVICES] = {};
    if (cfg.N <= WAVE) {
        if (hipError_t e = allow_big_lds(reinterpret_cast<const void *>(&k_mpc_any<1>), lds_ok1)) return e;
        hipLaunchKernelGGL(k_mpc_any<1>, dim3(a.B), dim3(WAVE), lds, st, cfg, a);
    } else {
        if (hipError_t e = allow_big_lds(reinterpret_cast<const void *>(&k_mpc_any<2>), lds_ok2)) return e;
        hipLaunchKernelGGL(k_mpc_any<2>, dim3(a.B), dim3(WAVE), lds, st, cfg, a);
    }
    return hipGetLastError();
}

}  // namespace mpc
