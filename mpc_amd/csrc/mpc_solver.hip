// Wave-per-scenario condensed MPC step for gfx950.
//
// One 64-lane wavefront owns one scenario; lane k owns stage k of the horizon (decision variable u_k,
// column k of the prediction matrix, row k of the condensed Hessian).  Per closed-loop step the wave
//   1. plans the car-frame reference (closed-loop modes) or reads it (optimize() mode),
//   2. linearises/discretises the controller model per stage,
//   3. condenses the OCP into  min 1/2 U'HU + g'U,  |u_k| <= umax,  |delta_0 + dt*sum_{j<=k} u_j| <= dmax
//      (per scenario, or through the fixed-model cache when every stage runs at one cached speed),
//   4. solves it exactly: unconstrained minimiser first; if a bound is violated, on the fixed-model path an active-set
//      solve in the space of the multipliers (mpc_aset.hpp), and where that does not end at the minimiser, or where the
//      problem is condensed per scenario, a Mehrotra predictor-corrector interior point; every Newton system
//      K = H + A'DA is factorised K = L D L' with row k of K held in the registers of lane k and the pivot column
//      broadcast through v_readlane,
//   5. rolls the prediction out, advances the plant with RK4 and does the lap bookkeeping,
// and repeats for `nsteps` steps inside one launch (scenarios are independent, so no grid-wide dependency).
//
// Two kernel variants: GENERAL (one wave per workgroup, per-scenario condensing available) and the
// fixed-model-only variant (four waves per workgroup sharing the cached matrices in LDS, 2 workgroups per CU).
//
// Reference behaviour restated (file:line of the reference checkout):
//   LMPC2.py:114-168 / OFLMPC2.py:118-181 / LPVMPC.py:110-171  stage model      -> stage_model()
//   LMPC2.py:220-260, LPVMPC.py:191-226, OFLMPC2.py:239-279    LINEAR_LS cost   -> condense()
//   LMPC2.py:196-210                                           bounds           -> interior_point()
//   LMPC2.py:296-347, LPVMPC.py:254-341, OFLMPC2.py:315-367    optimize()       -> k_mpc MODE_SOLVE
//   skidpad_simulator.py:101-135, simulator.py:178-202         closed loop      -> k_mpc MODE_SKIDPAD / MODE_STEP
#include "mpc_device.hpp"
#include "mpc_kernels.h"
#include "mpc_wave.hpp"
#include "mpc_ipm.hpp"
#include "mpc_aset.hpp"

// Linear algebra below may contract a*b+c; the plant and planner code in mpc_device.hpp is compiled without
// contraction (command line) so that it rounds like the numpy expressions it restates.
#pragma clang fp contract(fast)

namespace mpc {

// Condensing sweep over the horizon.  Lane j carries column j of the prediction matrix, v = d x_k / d u_j, and
// NF free responses (uniform over the wave).  Per stage k = 1..N it accumulates
//   g[f]   += v' W_k (xf[f] - ref_k)            gradient contribution of free response f
//   Hrow[i] += (W_k v_i)' v_j  for every i      row/column j of H = Gamma' W Gamma
// with W_k = dt*diag(w) for k < N and the dense terminal weight at k = N.  The weighted columns W_k v_i are
// exchanged through LDS (s_wv).  gref_out (optional) receives (W_k v)[c], c = 0..2: minus d g / d ref_c(k).
// s_ref is laid out [c][k] with row length N + 1.
// NR = 6: s_ref holds a reference row for every state ([6][N+1], terminal row included as it is to be used).
template <int N, int NF, int NR = 3>
__device__ __forceinline__ void condense(const DevCfg &cfg, int lane, const double *s_stage, const double *s_ref,
                                         double *s_wv, const double *s_We, const double (&wq)[6], double f35, double f45,
                                         const double (&c01s)[NF], const double (&c3)[NF], double (&xf)[NF][6],
                                         double (&Hrow)[N], double (&g)[NF], double *gref_out)
{
    const double dt = cfg.dt;
    const bool valid = lane < N;
#pragma unroll
    for (int i = 0; i < N; i++) Hrow[i] = 0.0;
    double v[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (int k = 1; k <= N; k++) {
        const double *st = &s_stage[(k - 1) * STAGE_W];
        const double e02 = st[0], e03 = st[1], e12 = st[2], e13 = st[3], f33 = st[4], f34 = st[5], f43 = st[6],
                     f44 = st[7], c0 = st[8], c1 = st[9];
#pragma unroll
        for (int f = 0; f < NF; f++) {   // free responses x_k = A_{k-1} x_{k-1} + c_{k-1}
            double *x = xf[f];
            const double n0 = x[0] + e02 * x[2] + e03 * x[3] + c01s[f] * c0;
            const double n1 = x[1] + e12 * x[2] + e13 * x[3] + c01s[f] * c1;
            const double n2 = x[2] + dt * x[4];
            const double n3 = f33 * x[3] + f34 * x[4] + f35 * x[5] + c3[f];
            const double n4 = f43 * x[3] + f44 * x[4] + f45 * x[5];
            x[0] = n0; x[1] = n1; x[2] = n2; x[3] = n3; x[4] = n4;
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
        double wv[6];
        if (k < N) {
#pragma unroll
            for (int c = 0; c < 6; c++) wv[c] = wq[c] * v[c];
        } else {
#pragma unroll
            for (int c = 0; c < 6; c++) {
                double t = 0.0;
#pragma unroll
                for (int d = 0; d < 6; d++) t = fma(s_We[d * 6 + c], v[d], t);   // (We' v)[c]; We is symmetric
                wv[c] = t;
            }
        }
#pragma unroll
        for (int f = 0; f < NF; f++) {
            double err[6] = {xf[f][0], xf[f][1], xf[f][2], xf[f][3], xf[f][4], xf[f][5]};
            // the LPV terminal reference stays at the constructor's zeros (LPVMPC.py:220-221)
            if (NR == 6) {
                if (s_ref && f == 0) {
#pragma unroll
                    for (int c = 0; c < 6; c++) err[c] -= s_ref[c * (N + 1) + k];
                }
            } else if (s_ref && f == 0 && !(k == N && cfg.kind == MPC_KIND_LPV)) {
                err[0] -= s_ref[k]; err[1] -= s_ref[(N + 1) + k]; err[2] -= s_ref[2 * (N + 1) + k];
            }
#pragma unroll
            for (int c = 0; c < 6; c++) g[f] = fma(wv[c], err[c], g[f]);
        }
        if (valid) {
#pragma unroll
            for (int c = 0; c < 6; c++) s_wv[c * N + lane] = wv[c];
            if (gref_out) {
#pragma unroll
                for (int c = 0; c < 3; c++) gref_out[((size_t)c * N + (k - 1)) * N + lane] = wv[c];
            }
        }
        wave_sync();
#pragma unroll
        for (int c = 0; c < 6; c++) {
            if (k == N || wq[c] != 0.0) {
                axpy_from_lds<N>(Hrow, &s_wv[c * N], v[c]);
            }
        }
        wave_sync();
    }
}

// ------------------------------------------------------------------------------------------------
// A solve of the fixed-model path whose unconstrained minimiser Uu violates a bound: the active-set method where the table
// T is at hand, the interior point for what it hands on.  One call site in the per-step loop (a function of its own:
// what the step loop keeps in registers is saved around it only on the steps that get here).
#ifndef MPC_CONSTRAINED_LINKAGE
#define MPC_CONSTRAINED_LINKAGE __noinline__ __attribute__((cold)) MPC_NOTAIL
#endif
#ifdef ASET_TIMING
extern "C" int mpc_debug_aset_timing(unsigned long long *out24, int reset)
{
    if (hipMemcpyFromSymbol(out24, HIP_SYMBOL(g_aset_t), sizeof(unsigned long long) * 24) != hipSuccess) return 1;
    if (reset) { unsigned long long z[24] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_aset_t), z, sizeof z) != hipSuccess) return 1; }
    return 0;
}
#endif
struct ConstrainedResult {
    double U, mu, res;
    int iters, status;           // interior-point iterations and status
    int work;                    // both methods' iterations in units of a quarter interior-point iteration (launch ordering)
    bool warm_valid;             // the scenario's record holds this solve's result
};

template <int N, int TAG>
static __device__ MPC_CONSTRAINED_LINKAGE ConstrainedResult solve_constrained(IpmParams prm, int lane, const double *Hm, const double *T,
                                                                   double Uu, double vdu, double up_d, double lo_d, double *lds,
                                                                   int *as_o, int as_cap_large)
{
    ConstrainedResult out = {0.0, 0.0, 0.0, 0, 0, 0, false};
    AT_BEGIN();
    AT_COUNT(18);
    // Back-off (ASET_BACKOFF_LOG2 in mpc_aset.hpp; off by default): a scenario whose active-set solves keep being handed
    // on would try the method again only after 1, 3, 7, ... steps of interior point alone.  The two counters (steps to
    // wait, failures in a row) travel in the scenario's record behind its validity flag.
    gdouble *bo = prm.warm ? (gdouble *)prm.warm + 5 * WAVE + 1 : nullptr;
    double bo0 = 0.0, bo1 = 0.0, bo2 = 0.0;
    if (bo) { bo0 = bo[0]; bo1 = bo[1]; bo2 = bo[2]; }                          // one round trip for all (bo[2]: the history word of the active-set solve)
    // (the output-feedback loop's sets follow the estimate's noise, not the plan: its solves start from the previous set
    // moved one stage ahead, measured quicker there than the prediction from two steps)
    const int hist_raw = __builtin_amdgcn_readfirstlane((int)bo2);
    const int hist = as_cap_large == ASET_CAP_LARGE_NOISY ? ((hist_raw & ~3) | min(hist_raw & 3, 1)) : hist_raw;
    const int wait = __builtin_amdgcn_readfirstlane((int)bo0);                  // wave-uniform
    AT(9);
    if (T && wait > 0) {
        if (lane == 0) bo[0] = (double)(wait - 1);
    } else if (T) {
        const AsetResult r = active_set_solve<N, TAG>(lane, T, Hm, Uu, vdu, up_d, lo_d, prm.umax, prm.dmax, prm.dt, prm.warm,
                                                      prm.warm_mode, lds, as_cap_large, hist);
        out.work = r.iters;
        AT(10);
        // the scenario's active-set counters live in memory (SolveArgs::as_out): they change on these steps only, and as
        // registers of the step loop they would crowd the unconstrained steps.  (Atomics without a return value: nothing
        // waits for them.)
        if (as_o && lane == 0) {
            atomicAdd(&as_o[r.why == ASET_SOLVED ? 0 : r.why == ASET_GROW ? 2 : 3], 1);
            atomicAdd(&as_o[1], r.iters);
            atomicAdd(&as_o[4], (r.flops + 32) >> 6);
        }
        if (bo && lane == 0) {
            const int streak = r.why == ASET_SOLVED ? 0 : min((int)bo1 + 1, 4);
            if ((double)streak != bo1) bo[1] = (double)streak;
            if (streak) bo[0] = (double)((1 << min(streak, ASET_BACKOFF_LOG2)) - 1);
            if (r.why != ASET_SOLVED && hist != 0) bo[2] = 0.0;          // the record will be the interior point's: no history
        }
        if (r.why == ASET_SOLVED) {
            out.U = r.U; out.warm_valid = prm.warm != nullptr;
            AT(11);
            return out;
        }
    }
    const bool valid = lane < N;
    double hu = 0.0;                       // g = -H U_unc
#pragma unroll 10
    for (int j = 0; j < N; j++) hu = fma(valid ? Hm[j * N + lane] : 0.0, bcast(Uu, j), hu);
    const IpmResult r = interior_point<N, TAG>(prm, lane, Hm, -hu, up_d, lo_d, (ldouble *)lds);
    out.U = r.U; out.mu = r.mu; out.res = r.res; out.iters = r.iters; out.status = r.status; out.warm_valid = r.warm_valid;
    out.work += 4 * r.iters;
    return out;
}

// The same for the kernel that condenses per scenario (no table: H is the scenario's and the step's own): one attempt of
// the active-set solve in its free-input form about the unconstrained minimiser Uu; the caller's interior point takes
// what is handed on.  A function of its own, like solve_constrained.
template <int N>
static __device__ MPC_CONSTRAINED_LINKAGE AsetTry general_constrained_aset(IpmParams prm, int lane, const double *Hm, double Uu, double vdu,
                                                                     double up_d, double lo_d, double *lds, int *as_o)
{
    return aset_try_general<N, 1, true>(prm, lane, Hm, Uu, vdu, 0.0, up_d, lo_d, lds, as_o);
}

// ------------------------------------------------------------------------------------------------
// LDS per wave: references [3][N+1], speeds [N+1], stage table [N][STAGE_W], parked plant state and coefficients, and
// for the closed-loop statistics the step's value for each of them (PARK_CL + i: see cl_step)
constexpr int PARK_CL = 20;
constexpr int PARK_W = PARK_CL + CL_STAT_W;
// and, for the skidpad planner, cosine and sine of the angle each waypoint lies ahead of the first on the circles
// ([2][N+1], fixed per scenario: the target speed and the time step are)
template <int N> constexpr int wave_rot_doubles() { return 2 * (N + 1); }
template <int N> constexpr int wave_lds_doubles() { return 4 * (N + 1) + N * STAGE_W + PARK_W; }
// fixed-model-only variant, shared by the 4 waves of a workgroup: ncomp reference maps + 5 state/offset maps
constexpr int FIXED_WPB = 4;

// PULL: the work-queue form of the fixed-model kernel (fused launches); its own instantiation, so that the one-step
// launches keep the straight-line body
// EMUL: the per-scenario condensing kernel with the SQP iteration of mpc_config.emulate_acados around its solve (what
// acados' MIRROR regularisation and stop rule do to the linear-quadratic problem, LMPC2.py:262-288); its own instantiation.
template <int N, bool GENERAL, bool PULL = false, bool EMUL = false>
__global__ __launch_bounds__(GENERAL ? WAVE : FIXED_WPB * WAVE, 2) void k_mpc(const DevCfg cfg,
                                                                                          const SolveArgs a)
{
    static_assert(!(GENERAL && PULL), "the work queue belongs to the fixed-model kernel");
    static_assert(GENERAL || !EMUL, "the regularised problem is condensed per scenario");
    constexpr int WPB = GENERAL ? 1 : FIXED_WPB;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform: keeps scenario addressing scalar
    const int lane = threadIdx.x & 63;
    int widx = blockIdx.x * WPB + wave;
    int s = (a.order && widx < a.B) ? a.order[widx] : widx;             // scenario this wave advances
    const bool valid = lane < N;        // lane owns a decision variable
    const bool validd = lane < N - 1;   // lane owns the bound on delta_{lane+1}

    extern __shared__ double lds[];
    // GENERAL: [wave area][s_wv 6N][We 36]      fixed: [Kref ncomp*N*N][Kx 5N][wave areas x4]
    const int ncomp = __popc(a.sh_mask & 7);
    double *s_K = lds;
    double *s_Kx = lds + (GENERAL ? 0 : ncomp * N * N);
    const int wave_area = wave_lds_doubles<N>() + (a.mode == MODE_OFL ? EKF_LDS : wave_rot_doubles<N>());
    double *wbase = GENERAL ? lds : s_Kx + 5 * N + wave * wave_area;
    double *s_ref = wbase, *s_p = wbase + 3 * (N + 1), *s_stage = wbase + 4 * (N + 1);
    double *s_park = wbase + 4 * (N + 1) + N * STAGE_W;   // plant state [10], plant coefficients [7], planner memory and speed while a step's solve runs
    double *s_ekf = wbase + wave_lds_doubles<N>();     // MODE_OFL only: xe[10] P[100] F[100] T[100] Aug[119]
    double *s_rot = wbase + wave_lds_doubles<N>();     // every other mode: the skidpad planner's cos / sin of the waypoints' angles ahead (the two loops never run together)
    double *s_wv = GENERAL ? wbase + wave_area : nullptr;
    double *s_We = GENERAL ? s_wv + 6 * N : nullptr;
    double *s_ref6 = EMUL ? s_We + 36 : nullptr;          // EMUL: the references the regularised problem is condensed with, [6][N+1]
    unsigned reg_term = 0;                                 // EMUL: states the terminal weight leaves alone (bit c)
    if (GENERAL) {
        if (lane < 36) s_We[lane] = (a.cost_scen && widx < a.B) ? a.cost_scen[43 * (size_t)s + 7 + lane] : cfg.We[lane];
        if (EMUL) {
            // acados mirrors the eigenvalues of every Hessian block below reg_epsilon up to it: in the terminal block
            // that touches the states whose row and column are zero
            wave_sync();
            bool zero = lane < 6;
            if (lane < 6) {
                for (int j = 0; j < 6; j++) zero = zero && s_We[lane * 6 + j] == 0.0 && s_We[j * 6 + lane] == 0.0;
            }
            reg_term = (unsigned)__ballot(zero);
            wave_sync();
            if (zero) s_We[lane * 7] = cfg.reg_eps;
            wave_sync();
        }
    } else {
        int ci = 0;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (a.sh_mask & (1 << c)) {
                for (int i = threadIdx.x; i < N * N; i += WPB * WAVE) s_K[ci * N * N + i] = a.sh_Kref[(size_t)c * N * N + i];
                ci++;
            }
        }
        for (int i = threadIdx.x; i < 5 * N; i += WPB * WAVE) s_Kx[i] = a.sh_Kx[i];
        __syncthreads();                 // the only workgroup-wide barrier: from here on the waves are independent
    }
    // Work queue (fixed-model kernel, fused launches): the wave takes scenarios from `order` until none is left.  The
    // last wave to leave puts the two counters back to zero for the next launch.
    constexpr bool pull = PULL;
#pragma unroll 1
    for (;;) {
    if (pull) {
        int w = 0;
        if (lane == 0) w = atomicAdd(a.queue, 1);
        widx = __builtin_amdgcn_readfirstlane(w);
        if (widx >= a.B) {
            if (lane == 0 && atomicAdd(a.queue + 1, 1) == (int)(gridDim.x * WPB) - 1) {
                __atomic_store_n(a.queue, 0, __ATOMIC_RELAXED); __atomic_store_n(a.queue + 1, 0, __ATOMIC_RELAXED);
            }
            return;
        }
        s = a.order ? a.order[widx] : widx;
    } else if (widx >= a.B)
        return;

    ModelParams mp;
    if (a.mp_scen) {
        const double *q = a.mp_scen + 6 * (size_t)s;
        mp = {q[0], q[1], q[2], q[3], q[4], q[5]};
    } else
        mp = {cfg.mp[0], cfg.mp[1], cfg.mp[2], cfg.mp[3], cfg.mp[4], cfg.mp[5]};
    const double dt = cfg.dt;
    const double umax = cfg.max_rate, dmax = cfg.max_steering;
    const double f35 = dt * (-mp.Cf / mp.m), f45 = dt * (-mp.lf * mp.Cf / mp.Iz);

    // closed-loop state lives in registers across the steps of one launch
    double plant[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // the 8-state plant is the 10-state one with zero disturbances
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
            wave_sync();
        }
    }
    // constant over the launch: the plant's speed-dependent coefficients (x[4] has zero derivative) and the progress
    // increments of the waypoints (sequentially accumulated like np.cumsum)
    // The plant's state and coefficients are needed at the two ends of a step only (reference planning, RK4).  In between
    // they wait in LDS: held in registers across the solve they push the per-step path into scratch spills.
    {
        const PlantCoef pc = plant_coef(mp, plant[4]);
        if (lane == 0) {
            s_park[10] = pc.vx; s_park[11] = pc.c11; s_park[12] = pc.c12a; s_park[13] = pc.b1; s_park[14] = pc.c21;
            s_park[15] = pc.c22; s_park[16] = pc.b2;
        }
    }
    // Closed-loop statistics (they live in the scenario's stat_out record: no pointer or flag of their own, which would
    // be more wave-uniform values alive across the step loop, where the scalar registers are already short).  Lane
    // i < CL_STAT_W carries statistic i of the scenario: one register for all of them.
    // Per step the value of every statistic is staged in LDS (the first reference point and the controller's start state
    // by lane 0 when it parks the plant state, the input and the violation by the whole wave after the solve) and each
    // lane reads its own: selecting among a dozen wave-uniform values per lane costs several times the instructions.
    // (The step loop's registers are all taken, scalar ones included: whatever is added here moves the allocation and
    // shows up as a few more scratch accesses per step -- measured +7% on unconstrained steps, nothing on constrained
    // ones, for every arrangement tried, running values in LDS instead of a register included.)
    double cl_acc = 0.0;
    if (a.mode != MODE_SOLVE && a.accumulate && lane < CL_STAT_W) cl_acc = a.stat_out[STAT_W * (size_t)s + STAT_CL + lane];
    if (lane < CL_STAT_W) s_park[PARK_CL + lane] = lane == CL_STEPS ? 1.0 : 0.0;
    const double cs0 = speed_cumsum(tvel, lane), cs1 = (lane + WAVE <= N) ? speed_cumsum(tvel, lane + WAVE) : 0.0;
    if (a.mode == MODE_SKIDPAD) {
        for (int k = lane; k <= N; k += WAVE) {
            double sk, ck;
            sincos(((k < WAVE ? cs0 : cs1) * dt) / SK_R, &sk, &ck);
            s_rot[k] = ck; s_rot[(N + 1) + k] = sk;
        }
    }
    double *warm_rec = (a.ipm_ws && a.mode != MODE_SOLVE) ? a.ipm_ws + (size_t)s * IPM_WARM_DOUBLES : nullptr;
    bool warm_ok = warm_rec && warm_rec[5 * WAVE] == 1.0;
    const int nsteps = (a.mode == MODE_SOLVE) ? 1 : a.nsteps;
    int iters_tot = 0, status_max = 0, nconstr = 0, lap_flag = 0, work_tot = 0;
    int sqp_tot = 0;                    // EMUL: SQP steps of the scenario's solves
    if (lane == 0 && a.as_out && !a.accumulate) { int *o = a.as_out + 6 * (size_t)s; o[0] = 0; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = 0; }
    double mu = 0.0, res = 0.0, u0 = 0.0;

#pragma unroll 1
    for (int t = 0; t < nsteps; t++) {
#ifdef ASET_TIMING
        const long long step_t0 = clock64();
#endif
        // ------------------------------------------------------------ 1. initial state and references
        double xh3, xh4, xh5;               // controller start [0,0,0,vy,r,delta] (LMPC2.py:298)
        double dhat = 0.0;
        if (a.mode == MODE_SOLVE) {
            const double *x0 = a.x0 + 7 * (size_t)s;
            xh3 = x0[4]; xh4 = x0[5]; xh5 = x0[6];
            if (a.d_hat) dhat = a.d_hat[s];
            for (int k = lane; k <= N; k += WAVE) {
                const double *w = a.wp + ((size_t)s * (N + 1) + k) * 4;
                s_ref[k] = w[0]; s_ref[(N + 1) + k] = w[1]; s_ref[2 * (N + 1) + k] = w[3];   // LMPC2.py:290-294
                s_p[k] = a.p[(size_t)s * (N + 1) + k];
            }
        } else {
            // pose and controller start come from the plant state, or from the EKF estimate in the output-feedback
            // loop (simulator.py:230-251: estimated_red_state, d_est = x_est[9])
            double px = plant[0], py = plant[1], hc = plant[2], hs = plant[3];
            xh3 = plant[5]; xh4 = plant[6]; xh5 = plant[7];
            if (a.mode == MODE_OFL) {
                px = s_xe[0]; py = s_xe[1]; hc = s_xe[2]; hs = s_xe[3];
                xh3 = s_xe[5]; xh4 = s_xe[6]; xh5 = s_xe[7]; dhat = s_xe[9];
            }
            double p0;
            double c0 = 1.0, s0 = 0.0;                                     // cos / sin of the first waypoint's angle (skidpad)
            if (a.mode == MODE_SKIDPAD) {
                int lp = lapv;
                p0 = sk_pos2progress(px, py, lp);
                if (prevp > p0 + 2 * SK_R) { lp += 1; p0 = sk_pos2progress(px, py, lp); }
                if (lp < 0 || lp > 7) lap_flag = 5;      // the reference switches to its speed profile there (path_planning.py:241-248)
                prevp = p0;                                                // path_planning.py:251
                sk_progress_sincos(px, py, lp, p0, c0, s0);
            } else
                p0 = px;                                                   // step_planning.py:83-86
            double ch, sh;                                                 // skidpad_simulator.py:104, path_planning.py:253
            unit_heading(hc, hs, ch, sh);
            for (int k = lane; k <= N; k += WAVE) {
                double o[4];
                const double q = (k == 0) ? p0 : p0 + (k < WAVE ? cs0 : cs1) * dt;    // path_planning.py:236-240
                if (a.mode == MODE_SKIDPAD) {
                    const double ck = s_rot[k], sk = s_rot[(N + 1) + k];
                    sk_pose_sc(q, s0 * ck + c0 * sk, c0 * ck - s0 * sk, o);
                }
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
        if (lane == 0) {                   // (in every mode: a value that may stay in its register on one path stays live on all)
#pragma unroll
            for (int i = 0; i < 10; i++) s_park[i] = plant[i];
            s_park[17] = prevp; s_park[18] = tvel;
            // [x0; u0] - yref_0 without the input, for the closed-loop statistics (nothing reads it in MODE_SOLVE)
            s_park[PARK_CL] = s_ref[0]; s_park[PARK_CL + 1] = s_ref[N + 1]; s_park[PARK_CL + 2] = s_ref[2 * (N + 1)];   // (lane 0's own writes)
            s_park[PARK_CL + 3] = xh3; s_park[PARK_CL + 4] = xh4; s_park[PARK_CL + 5] = xh5;

        }
        wave_sync();

        // ------------------------------------------------------------ 2. per-stage model (lane k -> stage k)
        const double c3 = (cfg.kind == MPC_KIND_OFL) ? dt * dhat : 0.0;    // OFLMPC2.py:145
        const bool need_stage = GENERAL || a.traj_out;
        if (need_stage && valid)
            stage_model(cfg, mp, s_p[lane], a.lin_x ? a.lin_x + (size_t)s * 6 * (N + 1) : nullptr, N, lane,
                        &s_stage[lane * STAGE_W]);

        // ------------------------------------------------------------ 3. condense, 4. solve
        double up_d, lo_d;                                       // bounds on dt * prefix sum, per lane
        bool relaxed;
        steer_bounds(xh5, dmax, umax, dt, lane, up_d, lo_d, relaxed);
        double g = 0.0, U = 0.0;
        int iters = 0, status = 0, constrained = 0;
        bool need_ipm = false, solved_warm = false;
        const double *Hm;

        if constexpr (EMUL) {
            // What acados' solver options do to this linear-quadratic problem (LMPC2.py:262-288), per SQP step: the stage
            // weights below reg_epsilon become reg_epsilon (MIRROR) with the previous iterate as their reference, the
            // full step is taken, and steps repeat until the stationarity residual that leaves behind,
            // reg_epsilon * max |x - x_prev| over the regularised states, is below the KKT tolerance.
            double *Hw = a.Hws + (size_t)s * N * N;
            Hm = Hw;
            double wq[6];
            unsigned reg_stage = 0;
#pragma unroll
            for (int c = 0; c < 6; c++) {
                wq[c] = dt * (a.cost_scen ? a.cost_scen[43 * (size_t)s + c] : cfg.w[c]);
                if (fabs(wq[c]) < cfg.reg_eps) { wq[c] = cfg.reg_eps; reg_stage |= 1u << c; }
            }
            const double wr = dt * (a.cost_scen ? a.cost_scen[43 * (size_t)s + 6] : cfg.w[6]);
            double *anc = a.sqp_x + (size_t)s * 6 * (N + 1);      // the iterate: lane k reads and writes stage k, lane 0 stage N too
            int sq = 0;
#pragma unroll 1
            for (;;) {
                if (valid) {
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        const double base = c < 3 ? s_ref[c * (N + 1) + lane] : 0.0;
                        s_ref6[c * (N + 1) + lane] = ((reg_stage >> c) & 1) ? anc[c * (N + 1) + lane] : base;
                    }
                }
                if (lane == 0) {
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        // (the LPV controller never sets its terminal reference: LPVMPC.py:220-221)
                        const double base = (c < 3 && cfg.kind != MPC_KIND_LPV) ? s_ref[c * (N + 1) + N] : 0.0;
                        s_ref6[c * (N + 1) + N] = ((reg_term >> c) & 1) ? anc[c * (N + 1) + N] : base;
                    }
                }
                wave_sync();
                Newton<N> nt;
                double xf[1][6] = {{0, 0, 0, xh3, xh4, xh5}};
                const double c01s[1] = {1.0}, c3s[1] = {c3};
                double gs[1] = {0.0};
                condense<N, 1, 6>(cfg, lane, s_stage, s_ref6, s_wv, s_We, wq, f35, f45, c01s, c3s, xf, nt.row, gs, nullptr);
                g = gs[0];
                const double wrd = opaque_value(wr);
                const int lane_d = opaque_int(lane);
#pragma unroll
                for (int i = 0; i < N; i++) nt.row[i] += (i == lane_d) ? wrd : 0.0;
                if (valid) {
                    store_column<N>(Hw + lane, nt.row);
                }
                if (!valid) g = 0.0;
                nt.template factor<false>(lane, (ldouble *)s_wv);
                const double Uu = valid ? nt.solve(-g, lane) : 0.0;
                const double vd = dt * incl_prefix(Uu, lane);
                bool ok = fabs(Uu) <= umax;
                if (validd) ok = ok && (vd <= up_d) && (vd >= lo_d);
                if (__all(ok)) { U = Uu; warm_ok = false; }
                else {
                    constrained = 1;
                    // the first step of a call starts from the previous closed-loop step's solve moved one stage ahead,
                    // the following ones from the step before them
                    const IpmParams prm = {dt, umax, dmax, cfg.tol_mu, cfg.tol_res, cfg.max_iter, warm_rec, warm_ok ? (sq == 0 ? 2 : 1) : 0, N};
                    const IpmResult r = interior_point<N, 1>(prm, lane, Hm, g, up_d, lo_d, (ldouble *)s_wv);
                    U = r.U; mu = r.mu; res = r.res; iters += r.iters; status = max(status, r.status);
                    work_tot += 4 * r.iters;
                    warm_ok = (warm_rec != nullptr) && r.warm_valid;
                }
                // the step's states, against the iterate
                double x[6] = {0, 0, 0, xh3, xh4, xh5};
                double keep[6] = {0, 0, 0, xh3, xh4, xh5};
#pragma unroll 1
                for (int k = 0; k < N; k++) {
                    const double *st = &s_stage[k * STAGE_W];
                    const double uk = bcast(U, k);
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
                double dmaxv = 0.0;
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    if (valid) {
                        if ((reg_stage >> c) & 1) dmaxv = fmax(dmaxv, fabs(keep[c] - anc[c * (N + 1) + lane]));
                        anc[c * (N + 1) + lane] = keep[c];
                    }
                    if (lane == 0) {
                        if ((reg_term >> c) & 1) dmaxv = fmax(dmaxv, fabs(x[c] - anc[c * (N + 1) + N]));
                        anc[c * (N + 1) + N] = x[c];
                    }
                }
                const double kkt = cfg.reg_eps * wave_max(dmaxv);
                sq++;
                if (kkt <= cfg.kkt_tol) break;
                if (sq >= cfg.sqp_max_iter) { status = max(status, 1); break; }
                wave_sync();
            }
            sqp_tot += sq;
        } else {
            bool fast = !GENERAL || a.sh_H != nullptr;
            if (fast) {
                bool same = true;
                for (int k = lane; k <= N; k += WAVE) same = same && (s_p[k] == a.sh_vx);
                fast = __all(same);
            }
            wave_sync();
            if (fast) {
                // Every stage runs at the cached speed: U_unc = -inv(H) g is affine in (vy, r, delta, d_hat, references)
                // with coefficient matrices condensed once by k_condense_fixed.
                Hm = a.sh_H;
                const double *Kx = GENERAL ? a.sh_Kx : s_Kx;
                double acc = 0.0;
                if (valid)
                    acc = Kx[lane] * xh3 + Kx[N + lane] * xh4 + Kx[2 * N + lane] * xh5 + Kx[3 * N + lane] + Kx[4 * N + lane] * dhat;
                int ci = 0;
    #pragma unroll
                for (int c = 0; c < 3; c++) {
                    if (a.sh_mask & (1 << c)) {
                        const double *Kc = GENERAL ? a.sh_Kref + (size_t)c * N * N : s_K + ci * N * N;
                        const double *rc = s_ref + c * (N + 1) + 1;      // reference of stage kk + 1, read as an LDS broadcast
                        acc += dot_column<N>(Kc, rc, valid ? lane : 0);        // lanes beyond N repeat column 0 and are zeroed below
                        ci++;
                    }
                }
                const double Uu = valid ? acc : 0.0;
                const double vd = dt * incl_prefix(Uu, lane);
                bool ok = fabs(Uu) <= umax;
                if (validd) ok = ok && (vd <= up_d) && (vd >= lo_d);
                // (the expectation is for the register allocator: what it has to move out of the way of the call below belongs
                // on that branch, not on the path every step takes)
                if (__builtin_expect(__all(ok), 1)) U = Uu;
                else {
                    constrained = 1;
                    // (s_ref is dead once U_unc is formed: the solvers borrow it)
                    const IpmParams prm = {dt, umax, dmax, cfg.tol_mu, cfg.tol_res, cfg.max_iter, warm_rec, warm_ok ? 2 : 0, N};
                    const ConstrainedResult r = solve_constrained<N, GENERAL ? 1 : 2>(prm, lane, Hm, a.sh_T, Uu, vd, up_d, lo_d, s_ref,
                                                                                      a.as_out ? a.as_out + 6 * (size_t)s : nullptr,
                                                                                      a.mode == MODE_OFL ? ASET_CAP_LARGE_NOISY : ASET_CAP_LARGE);
                    U = r.U; mu = r.mu; res = r.res; iters = r.iters; status = r.status;
                    solved_warm = (warm_rec != nullptr) && r.warm_valid;
                    work_tot += r.work;
                }
            } else if (!GENERAL) {
                Hm = a.sh_H;
                status = 3;
            } else {
                double *Hw = a.Hws + (size_t)s * N * N;
                Hm = Hw;
                Newton<N> nt;
                double wq[6];
    #pragma unroll
                for (int c = 0; c < 6; c++) wq[c] = dt * (a.cost_scen ? a.cost_scen[43 * (size_t)s + c] : cfg.w[c]);   // acados scales stage costs by the time step
                double xf[1][6] = {{0, 0, 0, xh3, xh4, xh5}};
                const double c01s[1] = {1.0}, c3s[1] = {c3};
                double gs[1] = {0.0};
                condense<N, 1>(cfg, lane, s_stage, s_ref, s_wv, s_We, wq, f35, f45, c01s, c3s, xf, nt.row, gs, nullptr);
                g = gs[0];
                const double wr = dt * (a.cost_scen ? a.cost_scen[43 * (size_t)s + 6] : cfg.w[6]);
                const double wrd = opaque_value(wr);      // keeps the N one-hot addends (and their N compare masks)
                const int lane_d = opaque_int(lane);      // out of hoisted, spilled registers
    #pragma unroll
                for (int i = 0; i < N; i++) nt.row[i] += (i == lane_d) ? wrd : 0.0;
                if (valid) {                            // one divergent region for all N stores, not one per entry
                    store_column<N>(Hw + lane, nt.row);
                }
                if (!valid) g = 0.0;
                nt.template factor<false>(lane, (ldouble *)s_wv);
                const double Uu = valid ? nt.solve(-g, lane) : 0.0;
                const double vd = dt * incl_prefix(Uu, lane);
                bool ok = fabs(Uu) <= umax;
                if (validd) ok = ok && (vd <= up_d) && (vd >= lo_d);
                if (__all(ok)) U = Uu;
                else {
                    constrained = 1; need_ipm = true;
                    if (a.general_aset) {
                        const IpmParams prm = {dt, umax, dmax, cfg.tol_mu, cfg.tol_res, cfg.max_iter, warm_rec, warm_ok ? 2 : 0, N};
                        const AsetTry r = general_constrained_aset<N>(prm, lane, Hm, Uu, vd, up_d, lo_d, s_wv,
                                                                      a.as_out ? a.as_out + 6 * (size_t)s : nullptr);
                        // (the flag comes back in a vector register: made uniform here, the branch is a scalar one)
                        if (__builtin_amdgcn_readfirstlane((int)r.solved)) { U = r.U; need_ipm = false; solved_warm = true; work_tot += 4; }
                    }
                }
            }
            if (need_ipm) {
                // closed loops start the interior point from the previous step's solve when that step needed one too
                const IpmParams prm = {dt, umax, dmax, cfg.tol_mu, cfg.tol_res, cfg.max_iter, warm_rec, warm_ok ? 2 : 0, N};
                const IpmResult r = interior_point<N, GENERAL ? 1 : 2>(prm, lane, Hm, g, up_d, lo_d, (ldouble *)(GENERAL ? s_wv : s_ref));   // both areas are dead by now
                U = r.U; mu = r.mu; res = r.res; iters = r.iters; status = r.status;
                work_tot += 4 * r.iters;
                warm_ok = (warm_rec != nullptr) && r.warm_valid;
            } else
                warm_ok = solved_warm;                 // the fixed-model solve left its result in the record, or no solve was needed
        }
        if (relaxed) status = max(status, 4);    // solved with the steering bound widened: the posed QP was infeasible
        status = max(status, lap_flag);

        // ------------------------------------------------------------ 5. outputs, prediction, plant
        if (valid && a.U_out) a.U_out[(size_t)s * N + lane] = U;
        u0 = bcast(U, 0);
        iters_tot += iters; status_max = max(status_max, status); nconstr += constrained;
        if (a.traj_out || cfg.kind == MPC_KIND_LPV) {
            // predicted states x_k: lane k keeps stage k, stage N is kept by every lane as `x`
            double x[6] = {0, 0, 0, xh3, xh4, xh5};
            double keep[6] = {0, 0, 0, xh3, xh4, xh5};
#pragma unroll 1
            for (int k = 0; k < N; k++) {
                const double *st = &s_stage[k * STAGE_W];
                const double uk = bcast(U, k);                 // k is wave-uniform: a v_readlane pair, no LDS round trip
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
            if (a.traj_out) {
                // 7-column form [x, y, cos(psi), sin(psi), vy, r, delta] (LMPC2.py:337-345).  The rows go through LDS (the
                // stage table is dead once the roll-out is done) and leave as consecutive doubles: a lane writing its own
                // row touches seven cache lines per store instruction, which over the bus (mpc_optimize_host writes
                // straight into host memory) is seven times the transactions.
                static_assert(N * STAGE_W >= 7 * (N + 1), "the stage table holds the trajectory");
                wave_sync();
                if (valid) {
                    double *o = s_stage + lane * 7;
                    o[0] = keep[0]; o[1] = keep[1]; o[2] = cos(keep[2]); o[3] = sin(keep[2]); o[4] = keep[3];
                    o[5] = keep[4]; o[6] = keep[5];
                }
                if (lane == 0) {
                    double *o = s_stage + N * 7;
                    o[0] = x[0]; o[1] = x[1]; o[2] = cos(x[2]); o[3] = sin(x[2]); o[4] = x[3]; o[5] = x[4]; o[6] = x[5];
                }
                wave_sync();
                double *to = a.traj_out + (size_t)s * (N + 1) * 7;
                for (int e = lane; e < 7 * (N + 1); e += WAVE) to[e] = s_stage[e];
            }
            if (cfg.kind == MPC_KIND_LPV) {
                // prev_x <- solver states (6 x (N+1)), prev_u <- inputs (LPVMPC.py:313-327)
                double *lx = a.lin_x + (size_t)s * 6 * (N + 1);
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    if (valid) lx[c * (N + 1) + lane] = keep[c];
                    if (lane == 0) lx[c * (N + 1) + N] = x[c];
                }
                if (valid) a.lin_u[(size_t)s * N + lane] = U;
            }
        }
        wave_sync();
#pragma unroll
        for (int i = 0; i < 10; i++) plant[i] = s_park[i];
        prevp = s_park[17]; tvel = s_park[18];
        if (a.mode != MODE_SOLVE) {
            const PlantCoef pc = {s_park[10], s_park[11], s_park[12], s_park[13], s_park[14], s_park[15], s_park[16]};
            const double oldx = plant[0];
            {   // closed-loop statistics of this step, from the state the step starts in
                const double tv = cl_pair_violation(plant[2], plant[3]);
                s_park[PARK_CL + 6] = u0; s_park[PARK_CL + CL_TRIG_SUM] = tv; s_park[PARK_CL + CL_TRIG_MAX] = tv;
                s_park[PARK_CL + CL_U_MAX] = fabs(u0); s_park[PARK_CL + CL_STEER_MAX] = fabs(plant[7]);
                cl_acc = cl_step(cl_acc, lane, s_park[PARK_CL + (lane < CL_STAT_W ? lane : 0)]);
            }
            plant_rk4_c<10>(plant, u0, dt, pc);                             // skidpad_simulator.py:128
            if (a.mode == MODE_SKIDPAD && plant[0] > SK_CENTER && oldx <= SK_CENTER) lapv += 1;   // :91-93
            if (a.u_hist && lane == 0) a.u_hist[(size_t)t * a.B + s] = u0;
            if (a.x_hist && lane < nxp) {
                double val = plant[0];
#pragma unroll
                for (int i = 1; i < 10; i++) val = (lane == i) ? plant[i] : val;
                a.x_hist[((size_t)t * a.B + s) * nxp + lane] = val;
            }
            if (a.mode == MODE_OFL) {                                       // simulator.py:262-272
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
        wave_sync();
#ifdef ASET_TIMING
        if (lane == 0) {
            atomicAdd(&g_aset_t[constrained ? 12 : 13], (unsigned long long)(clock64() - step_t0));
            atomicAdd(&g_aset_t[constrained ? 19 : 20], 1ull);
        }
#endif
    }   // steps of this launch

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
        if (warm_rec) warm_rec[5 * WAVE] = warm_ok ? 1.0 : 0.0;
        if (a.u0_out) a.u0_out[s] = u0;
        if (a.work_out) a.work_out[s] = (work_tot + 3) >> 2;     // in interior-point iterations
        const int st_all = a.accumulate ? max(a.status_own[s], status_max) : status_max;
        const int it_all = a.accumulate ? a.iters_own[s] + iters_tot : iters_tot;
        a.status_own[s] = st_all; a.iters_own[s] = it_all;
        if (a.status_out) a.status_out[s] = st_all;
        if (a.iters_out) a.iters_out[s] = it_all;
        if (EMUL && a.sqp_out) a.sqp_out[s] = a.accumulate ? a.sqp_out[s] + sqp_tot : sqp_tot;
        if (a.stat_out) {
            double *so = a.stat_out + STAT_W * (size_t)s;
            so[0] = mu; so[1] = res; so[2] = (a.accumulate ? so[2] : 0.0) + (double)nconstr; so[3] = fabs(u0);
        }
    }
    if (a.mode != MODE_SOLVE && lane < CL_STAT_W) a.stat_out[STAT_W * (size_t)s + STAT_CL + lane] = cl_acc;
    if (!pull) return;
    wave_sync();            // the wave's LDS areas are reused by its next scenario
    }   // work queue
}

// ------------------------------------------------------------------------------------------------
// Fixed-model cache: condense once for "every stage at speed vx", factor H and apply inv(H) to each way the
// gradient depends on the step's data.  One wavefront; a set-up cost of about 3N+5 triangular solve pairs.
template <int N>
__global__ __launch_bounds__(WAVE) void k_condense_fixed(const DevCfg cfg, double vx, double *H, double *Kref, double *Kx,
                                                         double *T, double *resid_out)
{
    const int lane = threadIdx.x;
    const bool valid = lane < N;
    __shared__ double s_stage[N * STAGE_W];
    __shared__ double s_wv[6 * N];
    __shared__ double s_We[36];
    if (lane < 36) s_We[lane] = cfg.We[lane];
    const ModelParams mp = {cfg.mp[0], cfg.mp[1], cfg.mp[2], cfg.mp[3], cfg.mp[4], cfg.mp[5]};
    const double dt = cfg.dt;
    if (valid) stage_model(cfg, mp, vx, nullptr, N, lane, &s_stage[lane * STAGE_W]);
    wave_sync();
    double wq[6];
#pragma unroll
    for (int c = 0; c < 6; c++) wq[c] = dt * cfg.w[c];
    const double f35 = dt * (-mp.Cf / mp.m), f45 = dt * (-mp.lf * mp.Cf / mp.Iz);
    // free responses: unit vy, unit r, unit delta, the model's constant term, unit d_hat
    double xf[5][6] = {{0, 0, 0, 1, 0, 0}, {0, 0, 0, 0, 1, 0}, {0, 0, 0, 0, 0, 1}, {0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0}};
    const double c01s[5] = {0, 0, 0, 1, 0}, c3s[5] = {0, 0, 0, 0, dt};
    double g[5] = {0, 0, 0, 0, 0};
    Newton<N> nt;
    condense<N, 5>(cfg, lane, s_stage, nullptr, s_wv, s_We, wq, f35, f45, c01s, c3s, xf, nt.row, g, Kref);
#pragma unroll
    for (int i = 0; i < N; i++) {
        if (i == lane) nt.row[i] += dt * cfg.w[6];
        if (valid) H[i * N + lane] = nt.row[i];
    }
    nt.template factor<false>(lane, (ldouble *)s_wv);
#pragma unroll 1
    for (int f = 0; f < 5; f++) {
        double gf = g[0];
#pragma unroll
        for (int q = 1; q < 5; q++) gf = (f == q) ? g[q] : gf;
        const double x = nt.solve(valid ? -gf : 0.0, lane);
        if (valid) Kx[f * N + lane] = x;
    }
    __syncthreads();
#pragma unroll 1
    for (int ck = 0; ck < 3 * N; ck++) {
        const double rhs = valid ? Kref[(size_t)ck * N + lane] : 0.0;
        const double x = nt.solve(rhs, lane);
        if (valid) Kref[(size_t)ck * N + lane] = x;
    }
    // table of the active-set solve (mpc_aset.hpp): G = inv(H), G P' and P G P' with P = dt * (prefix-sum rows), as one
    // symmetric 2N x 2N matrix [[G, G P'], [P G, P G P']]; column c of G is one solve with a unit vector
    // The active-set solve certifies its points through this table alone, so the table is checked against H here:
    // resid_out[0] = the largest entry of |H G - I| (the launcher switches the method off when it is too large).
    double rmax = 0.0;
    if (T) {
        constexpr int LD = 2 * N;
        double run = 0.0;
#pragma unroll 1
        for (int c = 0; c < N; c++) {
            const double x = nt.solve(lane == c ? 1.0 : 0.0, lane);      // lane i: G[i][c]
            {
                double hx0 = 0.0, hx1 = 0.0;                              // row `lane` of H (= its column: symmetric) against x
#pragma unroll 1
                for (int j = 0; j < N; j += 2) {
                    hx0 = fma(valid ? H[j * N + lane] : 0.0, bcast(x, j), hx0);
                    if (j + 1 < N) hx1 = fma(valid ? H[(j + 1) * N + lane] : 0.0, bcast(x, j + 1), hx1);
                }
                if (valid) rmax = fmax(rmax, fabs(hx0 + hx1 - (lane == c ? 1.0 : 0.0)));
            }
            run += valid ? x : 0.0;
            const double gp = dt * run;                                    // lane i: (G P')[i][c]
            const double pgp = dt * incl_prefix(valid ? gp : 0.0, lane);   // lane k: (P G P')[k][c]
            if (valid) {
                T[lane * LD + c] = x;
                T[lane * LD + N + c] = gp;
                T[(N + c) * LD + lane] = gp;
                T[(N + lane) * LD + N + c] = pgp;
            }
        }
    }
    rmax = wave_max(rmax);
    if (resid_out && lane == 0) resid_out[0] = rmax;
}

template <int N>
static hipError_t launch_fixed_n(const DevCfg &cfg, double vx, double *H, double *Kref, double *Kx, double *T, double *resid, hipStream_t st)
{
    hipLaunchKernelGGL(k_condense_fixed<N>, dim3(1), dim3(WAVE), 0, st, cfg, vx, H, Kref, Kx, T, resid);
    return hipGetLastError();
}

hipError_t launch_condense_fixed(const DevCfg &cfg, double vx, double *H, double *Kref, double *Kx, double *T, double *resid, hipStream_t st)
{
    switch (cfg.N) {
#define MPC_CASE(n) case n: return launch_fixed_n<n>(cfg, vx, H, Kref, Kx, T, resid, st);
        MPC_HORIZONS(MPC_CASE)
#undef MPC_CASE
    default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------
template <int N>
static hipError_t launch_n(const DevCfg &cfg, const SolveArgs &a, hipStream_t st)
{
    if (a.sh_H && a.sh_exclusive) {
        const int ncomp = __builtin_popcount(a.sh_mask & 7);
        const int wave_area = wave_lds_doubles<N>() + (a.mode == MODE_OFL ? EKF_LDS : wave_rot_doubles<N>());
        const size_t lds = sizeof(double) * (size_t)(ncomp * N * N + 5 * N + FIXED_WPB * wave_area);
        static bool lds_ok[MPC_MAX_DEVICES] = {}, lds_ok_pull[MPC_MAX_DEVICES] = {};
        if (hipError_t e = a.queue ? allow_big_lds(reinterpret_cast<const void *>(&k_mpc<N, false, true>), lds_ok_pull)
                                   : allow_big_lds(reinterpret_cast<const void *>(&k_mpc<N, false, false>), lds_ok)) return e;
        int blocks = (a.B + FIXED_WPB - 1) / FIXED_WPB;
        if (a.queue) {
            // as many workgroups as stay resident (fewer when the batch is smaller): their waves share the whole batch
            // (looked up once per device and LDS size: the property query is slow)
            static int cached_resident[MPC_MAX_DEVICES] = {};
            static size_t cached_lds[MPC_MAX_DEVICES] = {};
            int dev = 0;
            if (hipError_t e = hipGetDevice(&dev)) return e;
            const int slot = dev >= 0 && dev < MPC_MAX_DEVICES ? dev : 0;
            if (!cached_resident[slot] || cached_lds[slot] != lds) {
                int per_cu = 0, cus = 0;
                if (hipError_t e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev)) return e;
                if (hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(&k_mpc<N, false, true>), FIXED_WPB * WAVE, lds)) return e;
                cached_resident[slot] = per_cu * cus; cached_lds[slot] = lds;
            }
            const int resident = cached_resident[slot];
            if (resident >= 1 && resident < blocks) blocks = resident;
        }
        if (a.queue) hipLaunchKernelGGL((k_mpc<N, false, true>), dim3(blocks), dim3(FIXED_WPB * WAVE), lds, st, cfg, a);
        else hipLaunchKernelGGL((k_mpc<N, false, false>), dim3(blocks), dim3(FIXED_WPB * WAVE), lds, st, cfg, a);
    } else {
        const int wave_area = wave_lds_doubles<N>() + (a.mode == MODE_OFL ? EKF_LDS : wave_rot_doubles<N>());
        const size_t lds = sizeof(double) * (size_t)(wave_area + 6 * N + 36 + (a.sqp_x ? 6 * (N + 1) : 0));
        if (a.sqp_x) hipLaunchKernelGGL((k_mpc<N, true, false, true>), dim3(a.B), dim3(WAVE), lds, st, cfg, a);
        else hipLaunchKernelGGL((k_mpc<N, true>), dim3(a.B), dim3(WAVE), lds, st, cfg, a);
    }
    return hipGetLastError();
}

hipError_t launch_mpc(const DevCfg &cfg, const SolveArgs &a, hipStream_t st)
{
    switch (cfg.N) {
#define MPC_CASE(n) case n: return launch_n<n>(cfg, a, st);
        MPC_HORIZONS(MPC_CASE)
#undef MPC_CASE
    default: return launch_mpc_any(cfg, a, st);
    }
}

bool horizon_supported(int N) { return N >= 2 && N <= MPC_MAX_HORIZON; }

bool horizon_templated(int N)
{
    switch (N) {
#define MPC_CASE(n) case n: return true;
        MPC_HORIZONS(MPC_CASE)
#undef MPC_CASE
    default: return false;
    }
}

}  // namespace mpc
