// Register-resident Newton systems and the interior-point iteration of the condensed MPC problem (gfx950, wave64):
// shared by the linear-model kernels (mpc_solver.hip) and the nonlinear SQP kernel (mpc_nl.hip).
#pragma once
#include <utility>
#include "mpc_wave.hpp"

#pragma clang fp contract(fast)

namespace mpc {

typedef __attribute__((address_space(1))) double gdouble;   // a double in global memory
typedef __attribute__((address_space(3))) double ldouble;   // a double in LDS

// ------------------------------------------------------------------------------------------------
template <int N>
struct Newton {
    // Row `lane` of the symmetric Newton matrix.  After factor(): entries k < lane hold l(lane,k),
    // entry lane holds d(lane), entries k > lane hold the frozen d(lane)*l(k,lane) (the transposed
    // factor), so both triangular solves index registers statically.
    double row[N];
    double dinv;

    // Elimination step K of the LDL' factorisation.  The update row[j] -= l * c(j) needs entry j of the pivot column in
    // every lane: from the owning lane's register through v_readlane (two VALU issues per value, on a VALU that is the
    // bottleneck of the interior point), or read back from a 64-double LDS slot `bc` (any per-wave LDS area that is dead
    // during the factorisation) that every lane has written its entry to; the LDS reads are issued a group ahead of
    // their multiply-adds.  With USE_LDS all but the first few columns of a step (a multiple of the group size goes
    // through LDS) take the second way; measured on the fixed-model kernel: 9% more throughput with every wave in the
    // interior point (c4 workload), 3% on c2.
    template <int K, bool USE_LDS>
    __device__ __forceinline__ void eliminate(int lane, int lane_in, ldouble *bc)
    {
        constexpr int LG = 8;                                             // LDS group
        constexpr int M = N - 1 - K;                                      // entries to update
        constexpr int NL = (USE_LDS && M >= 2 * LG) ? (M / LG) * LG : 0;             // how many of them (the last ones) go through LDS
        constexpr int JL = N - NL;                                        // first LDS column
        const double c = row[K];                 // column K of the trailing matrix, one entry per lane
        if (NL) bc[lane_in] = c;
        const double inv = fast_rcp(bcast(c, K));
        const bool below = lane > K;
        const double l = below ? c * inv : 0.0;
        if (lane == K) dinv = inv;
        double t[2][LG];
        if (NL) {
#pragma unroll
            for (int u = 0; u < LG; u++) t[0][u] = bc[JL + u];
        }
        // groups of 4: broadcast four pivot-column entries, then the four updates that use them
#pragma unroll
        for (int j0 = K + 1; j0 < JL; j0 += 4) {
            double cj[4];
#pragma unroll
            for (int g = 0; g < 4; g++)
                if (j0 + g < JL) cj[g] = bcast_pinned(c, j0 + g);
#pragma unroll
            for (int g = 0; g < 4; g++)
                if (j0 + g < JL) row[j0 + g] = fma(-l, cj[g], row[j0 + g]);
            __builtin_amdgcn_sched_barrier(0);                           // keep readlane results (SGPRs) short-lived
        }
#pragma unroll
        for (int g0 = 0; g0 < NL; g0 += LG) {
            const int cur = (g0 / LG) & 1;
            if (g0 + LG < NL) {
#pragma unroll
                for (int u = 0; u < LG; u++) t[cur ^ 1][u] = bc[JL + g0 + LG + u];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < LG; u++) row[JL + g0 + u] = fma(-l, t[cur][u], row[JL + g0 + u]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (below) row[K] = l;
        __builtin_amdgcn_sched_barrier(0);
    }

    template <bool USE_LDS, int... Ks>
    __device__ __forceinline__ void eliminate_all(int lane, int lane_in, ldouble *bc, std::integer_sequence<int, Ks...>)
    {
        (eliminate<Ks, USE_LDS>(lane, lane_in, bc), ...);
    }

    // USE_LDS = false keeps every broadcast on v_readlane (for kernels whose LDS pipe is the busier one).
    template <bool USE_LDS = true>
    __device__ __forceinline__ void factor(int lane_in, ldouble *bc)
    {
        dinv = 1.0;
        eliminate_all<USE_LDS>(opaque_int(lane_in), lane_in, bc, std::make_integer_sequence<int, N>{});
    }

    // solves (L D L') x = b; b, x one entry per lane
    __device__ __forceinline__ double solve(double b, int lane_in) const
    {
        const int lane = opaque_int(lane_in);
        double w = b;
#pragma unroll
        for (int k = 0; k < N - 1; k++) {
            const double t = fma(-row[k], bcast(w, k), w);     // the broadcast is the fma's scalar operand; the select is
            w = lane > k ? t : w;                              // on the result (a masked copy of the row would be hoisted)
            if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int k = N - 1; k > 0; k--) {
            const double t = fma(-row[k], bcast(w * dinv, k), w);
            w = lane < k ? t : w;
            if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
        return w * dinv;
    }
};

// ------------------------------------------------------------------------------------------------
// Bounds on delta_{lane+1} - delta_0 = dt * (prefix sum of U)[lane].  |delta_k| <= dmax holds for stages 1..N-1
// (LMPC2.py:196-210).  When the start value delta_0 lies so far outside that even the full steering rate cannot bring
// stage 1 back inside, the reference's QP has no feasible point (acados raises); here the bound is then widened along
// the way back in at 98% of the full rate, so that a solution exists, and `relaxed` tells the caller to flag it.
__device__ __forceinline__ void steer_bounds(double d0, double dmax, double umax, double dt, int lane, double &up,
                                             double &lo, bool &relaxed)
{
    const double reach = 0.98 * umax * dt * (double)(lane + 1);
    up = fmax(dmax, d0 - reach) - d0;
    lo = fmin(-dmax, d0 + reach) - d0;
    relaxed = fabs(d0) - 0.98 * umax * dt > dmax;
}

// ------------------------------------------------------------------------------------------------
// Mehrotra predictor-corrector on  G U + t = h,  G = [A; -A],  A = [I; dt * prefix-sum], with the multipliers started
// at 1% of the gradient's size and a fraction-to-the-boundary of min(max(0.995, 1 - mu), 0.999999).  The predictor-
// corrector has no centrality control and, rarely (order 1e-5 of the problems met in closed loop), settles into a cycle
// of short steps; after IPM_SAFE_AFTER iterations the loop therefore continues as a long-step path-following method
// (fixed centering 0.25, no second-order term, step cut back until every slack-multiplier product stays above a fixed
// fraction of their mean), which converges from any interior point.
// H is read from memory (Hm[j*N + lane], symmetric); up_d / lo_d are this lane's steer_bounds().
// Returns the solver status (0 converged, 2 iteration cap).  prm.warm / prm.warm_mode: see IpmParams.
//
// Deliberately NOT inlined: it runs on the minority of steps whose unconstrained minimiser violates a bound, and
// as a real call its ~250 registers (the Newton-matrix row alone is 2N) stay out of the register allocation of
// the per-step hot path; the caller's live state is saved around the call only when the call happens.
constexpr int IPM_SAFE_AFTER = 25;
constexpr double IPM_WARM_RES_MAX = 0.05;   // largest relative residual a warm start is accepted with

#ifndef MPC_NOTAIL
#define MPC_NOTAIL __attribute__((not_tail_called))
#endif
// TAG gives every calling kernel its own copy (internal linkage), so that the copy inherits that kernel's occupancy
// target as its register budget instead of the loosest budget among all callers.
template <int N, int TAG>
static __device__ __noinline__ MPC_NOTAIL IpmResult interior_point(IpmParams prm, int lane, const double *Hm, double g, double up_d,
                                                 double lo_d, ldouble *bc)
{
    Newton<N> nt;
    double U, mu, res;
    int iters;
    // number of inputs of the problem: N, except for the NL kernels, which run any horizon up to N on the instantiation
    // for N (the rows beyond nu are unconstrained unit rows of the Newton matrix with a zero right-hand side)
    const int nu = (TAG == 3) ? prm.nu : N;
    const bool valid = lane < nu, validd = lane < nu - 1;
    const int lc = valid ? lane : 0;
    const int lr = (TAG == 3) ? (lane < N ? lane : 0) : lc;      // row of H this lane holds (rows nu..N-1 exist in the NL kernels)
    const double dt = prm.dt, dt2 = dt * dt;
    const double umax = prm.umax, dmax = prm.dmax;
    const double gscale = fmax(1.0, wave_max(fabs(g)));
    const double lam0 = fmax(1.0, 0.01 * gscale);
    double tpu, tmu, lpu = lam0, lmu = lam0, tpd, tmd, lpd = lam0, lmd = lam0;
    U = 0.0;
    tpu = umax; tmu = umax;
    tpd = fmax(up_d, 0.1 * dmax); tmd = fmax(-lo_d, 0.1 * dmax);
    auto cold_start = [&]() {
        U = 0.0;
        tpu = umax; tmu = umax; lpu = lam0; lmu = lam0;
        tpd = fmax(up_d, 0.1 * dmax); tmd = fmax(-lo_d, 0.1 * dmax); lpd = lam0; lmd = lam0;
        if (!valid) { tpu = 1.0; tmu = 1.0; lpu = 0.0; lmu = 0.0; }
        if (!validd) { tpd = 1.0; tmd = 1.0; lpd = 0.0; lmd = 0.0; }
    };
    bool warm = prm.warm_mode != 0;
    if (warm) {
        // Start from the previous solve: its inputs (moved one stage ahead for the next closed-loop step), slacks pushed
        // back to at least 0.3% of the bound's width, multipliers lifted so that no product starts below a common small
        // target.  Typically a third fewer iterations than the cold start.
        const gdouble *w = (const gdouble *)prm.warm;
        double u0 = w[lc], a0 = w[WAVE + lc], a1 = w[2 * WAVE + lc], a2 = w[3 * WAVE + lc], a3 = w[4 * WAVE + lc];
        if (prm.warm_mode == 2) {
            // the record of the next stage, straight from memory (no cross-lane shuffle of loaded values)
            const int ln = min(lc + 1, nu - 1);
            const double un = w[ln], b0 = w[WAVE + ln], b1 = w[2 * WAVE + ln], b2 = w[3 * WAVE + ln], b3 = w[4 * WAVE + ln];
            if (lane < nu - 1) { u0 = un; a0 = b0; a1 = b1; }
            if (lane < nu - 2) { a2 = b2; a3 = b3; }
        }
        const double delta = 3e-3;
        U = valid ? fmin(fmax(u0, -umax), umax) : 0.0;
        const double vd0 = dt * incl_prefix(U, lane);
        tpu = fmax(umax - U, delta * 2.0 * umax); tmu = fmax(umax + U, delta * 2.0 * umax);
        tpd = fmax(up_d - vd0, delta * 2.0 * dmax); tmd = fmax(vd0 - lo_d, delta * 2.0 * dmax);
        const double mu0 = delta * 0.01 * gscale;
        lpu = fmax(a0, mu0 / tpu); lmu = fmax(a1, mu0 / tmu);
        lpd = fmax(a2, mu0 / tpd); lmd = fmax(a3, mu0 / tmd);
    }
    if (!valid) { tpu = 1.0; tmu = 1.0; lpu = 0.0; lmu = 0.0; }
    if (!validd) { tpd = 1.0; tmd = 1.0; lpd = 0.0; lmd = 0.0; }
    const double inv_nc = 1.0 / (2.0 * (2 * nu - 1));
    double res_prev = 1e300, U_last = 0.0, mu_last = 0.0, res_last = 0.0;
    bool acc_last = false, rec_ok = true;      // rec_ok: the multipliers belong to the returned inputs (warm-start record)
    int status = 0;
    iters = 0;
#pragma unroll 1
    for (;;) {
        // H is re-read from memory (L2-resident) wherever it is needed: laundering the pointer keeps the compiler
        // from hoisting the N loads out of the iteration loop into 2N registers per lane.
        // (loads are unconditional from a clamped row index and global-addressed: a predicated generic load costs a
        // branch and a flat instruction per entry)
        // This lane's row of H goes into the Newton row once per iteration, all loads in flight together: the residual
        // takes H U from it, and the Newton matrix is then built in place.
        const gdouble *Hr = (const gdouble *)opaque_uniform(Hm);
#pragma unroll
        for (int j = 0; j < N; j++) nt.row[j] = Hr[j * N + lr];
        __builtin_amdgcn_sched_barrier(0);
        double hu = 0.0;
#pragma unroll
        for (int j = 0; j < N; j++) hu = fma(nt.row[j], bcast(U, j), hu);
        if (!valid) hu = 0.0;
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
        if (warm && iters == 0 && res > IPM_WARM_RES_MAX) {
            // the problem moved too far from the previous one (e.g. a noisy state estimate): a start this close to the
            // boundary with residuals this large only blocks the steps.  Fall back to the cold start.
            warm = false;
            cold_start();
            continue;
        }
        // Exit rules.  An iterate is `acceptable` when its complementarity is within 10x of the tolerance and its residual
        // within 1000x of its own: close to the solution the Newton matrix (the steering rows add dt^2 * sum(lambda/t) to every
        // entry) loses its small eigenvalues to rounding, the residual stops at a floor of a few 1e-12 of the gradient's
        // size, and one step further the factorisation can break down altogether.  So: stop at the tolerances; stop at
        // an acceptable iterate whose residual no longer halves or whose complementarity is already 1000x below its
        // tolerance; and when the step taken from an acceptable iterate produced something worse or not finite, return
        // that acceptable iterate.  A solve that breaks down otherwise is repeated from the cold start if it began from the
        // previous solve's record; status 2 is left for what still fails.
        const bool finite = fabs(mu) < 1e300;                        // false for NaN and infinity
        const bool acc = finite && mu <= 10.0 * prm.tol_mu && res <= 1e3 * prm.tol_res;
        if (acc_last && !acc) { U = U_last; mu = mu_last; res = res_last; rec_ok = false; break; }
        if (finite && mu <= prm.tol_mu && res <= prm.tol_res) break;
        if (acc && (res >= 0.5 * res_prev || mu <= 1e-3 * prm.tol_mu)) break;
        if (!finite || mu <= 1e-3 * prm.tol_mu || iters >= prm.max_iter) {
            if (acc) break;                                          // (the iteration cap at an acceptable iterate)
            if (warm && !(iters >= prm.max_iter)) {
                // a solve that started from the previous one's record broke down: once more from the cold start
                warm = false;
                cold_start();
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
        // reciprocals of the slacks and multipliers, shared by everything below (a quotient costs ~3x a reciprocal)
        const double itpu = fast_rcp(tpu), itmu = fast_rcp(tmu), itpd = fast_rcp(tpd), itmd = fast_rcp(tmd);
        const double ilpu = valid ? fast_rcp(lpu) : 0.0, ilmu = valid ? fast_rcp(lmu) : 0.0;
        const double ilpd = validd ? fast_rcp(lpd) : 0.0, ilmd = validd ? fast_rcp(lmd) : 0.0;
        // Newton matrix  K = H + diag(Du) + dt^2 * Sfx[max(i,j)],  Sfx = suffix sums of Dd
        const double Du = lpu * itpu + lmu * itmu;
        const double Dd = validd ? lpd * itpd + lmd * itmd : 0.0;
        const double sfx = incl_suffix(Dd, lane);
        const int lane_k = opaque_int(lane);
        // Dd >= 0, so the suffix sums do not increase with the stage and Sfx[max(i,j)] = min(Sfx[i], Sfx[j]): one v_min
        // with the broadcast value instead of a compare-and-select pair per entry (and symmetric whatever the rounding
        // of the scan).  Lanes beyond N carry a copy of row 0: finite values nobody reads.
#pragma unroll
        for (int j = 0; j < N; j++) {
            double kij = fma(dt2, fmin(sfx, bcast(sfx, j)), nt.row[j]);
            kij += (j == lane_k) ? Du : 0.0;
            nt.row[j] = kij;
        }
        nt.template factor<TAG == 2>(lane, bc);       // TAG 2 is the fixed-model kernel; the others condense per scenario,
                                                      // which loads the LDS pipe already
        double dU = 0, dtpu = 0, dtmu = 0, dlpu = 0, dlmu = 0, dtpd = 0, dtmd = 0, dlpd = 0, dlmd = 0;
        const bool safe = iters >= IPM_SAFE_AFTER;
        double sigmu = safe ? 0.25 * mu : 0.0, alpha = 1.0;
#pragma unroll 1
        for (int pass = 0; pass < 2; pass++) {
            double rcpu = tpu * lpu, rcmu = tmu * lmu, rcpd = tpd * lpd, rcmd = tmd * lmd;
            if (safe) {
                rcpu -= sigmu; rcmu -= sigmu; rcpd -= sigmu; rcmd -= sigmu;
            } else if (pass) {
                rcpu += dtpu * dlpu - sigmu; rcmu += dtmu * dlmu - sigmu;
                rcpd += dtpd * dlpd - sigmu; rcmd += dtmd * dlmd - sigmu;
            }
            const double wu = (lpu * rppu - rcpu) * itpu - (lmu * rpmu - rcmu) * itmu;
            const double wd = validd ? (lpd * rppd - rcpd) * itpd - (lmd * rpmd - rcmd) * itmd : 0.0;
            double rhs = -rd - wu - dt * incl_suffix(wd, lane);
            if (!valid) rhs = 0.0;
            dU = nt.solve(rhs, lane);
            if (!valid) dU = 0.0;
            const double dvd = dt * incl_prefix(dU, lane);
            dtpu = -rppu - dU; dtmu = -rpmu + dU;
            dlpu = -(rcpu + lpu * dtpu) * itpu; dlmu = -(rcmu + lmu * dtmu) * itmu;
            dtpd = -rppd - dvd; dtmd = -rpmd + dvd;
            dlpd = -(rcpd + lpd * dtpd) * itpd; dlmd = -(rcmd + lmd * dtmd) * itmd;
            if (!valid) { dtpu = 0; dtmu = 0; dlpu = 0; dlmu = 0; }
            if (!validd) { dtpd = 0; dtmd = 0; dlpd = 0; dlmd = 0; }
            // largest step keeping every slack and multiplier positive: 1 / max(1, max_i(-d_i / v_i))
            double rmax = fmax(fmax(-dtpu * itpu, -dtmu * itmu), fmax(-dlpu * ilpu, -dlmu * ilmu));
            rmax = fmax(rmax, fmax(fmax(-dtpd * itpd, -dtmd * itmd), fmax(-dlpd * ilpd, -dlmd * ilmd)));
            rmax = wave_max(rmax);
            alpha = rmax > 1.0 ? fast_rcp(rmax) : 1.0;
            if (safe) break;
            if (pass == 0) {
                const double mua = wave_sum((tpu + alpha * dtpu) * (lpu + alpha * dlpu) +
                                            (tmu + alpha * dtmu) * (lmu + alpha * dlmu) +
                                            (tpd + alpha * dtpd) * (lpd + alpha * dlpd) +
                                            (tmd + alpha * dtmd) * (lmd + alpha * dlmd)) * inv_nc;
                const double sg = mua / mu;
                sigmu = fmax(sg * sg * sg * mu, 1e-3 * fmin(mu, res));   // the gap must not outrun the residuals
            }
        }
        if (!safe) {
            alpha *= fmin(fmax(0.995, 1.0 - mu), 0.999999);   // never onto the boundary itself
        } else {
            // neighbourhood of the central path: no product may fall below gam times the mean (gam from the current
            // point if that is already worse than 0.01)
            const double big = 1e300;
            const double minp0 = wave_min(fmin(fmin(valid ? tpu * lpu : big, valid ? tmu * lmu : big),
                                               fmin(validd ? tpd * lpd : big, validd ? tmd * lmd : big)));
            const double gam = fmin(0.01, 0.5 * minp0 / mu);
            alpha *= 0.99;
#pragma unroll 1
            for (int bt = 0; bt < 20; bt++) {
                const double p1 = (tpu + alpha * dtpu) * (lpu + alpha * dlpu), p2 = (tmu + alpha * dtmu) * (lmu + alpha * dlmu);
                const double p3 = (tpd + alpha * dtpd) * (lpd + alpha * dlpd), p4 = (tmd + alpha * dtmd) * (lmd + alpha * dlmd);
                const double mun = wave_sum(p1 + p2 + p3 + p4) * inv_nc;
                const double minp = wave_min(fmin(fmin(valid ? p1 : big, valid ? p2 : big), fmin(validd ? p3 : big, validd ? p4 : big)));
                if (minp >= gam * mun) break;
                alpha *= 0.6;
            }
        }
        U = fma(alpha, dU, U);
        tpu = fma(alpha, dtpu, tpu); tmu = fma(alpha, dtmu, tmu);
        lpu = fma(alpha, dlpu, lpu); lmu = fma(alpha, dlmu, lmu);
        tpd = fma(alpha, dtpd, tpd); tmd = fma(alpha, dtmd, tmd);
        lpd = fma(alpha, dlpd, lpd); lmd = fma(alpha, dlmd, lmd);
        iters++;
    }
    if (prm.warm && rec_ok) {
        gdouble *w = (gdouble *)prm.warm;
        // the multiplier of a bound that is not active (slack above multiplier at the end of the path) goes into the
        // record as an exact zero: the next warm start lifts it to its floor either way, and the active-set solves read
        // the record as "active where positive"
        lpu = lpu > tpu ? lpu : 0.0; lmu = lmu > tmu ? lmu : 0.0;
        lpd = lpd > tpd ? lpd : 0.0; lmd = lmd > tmd ? lmd : 0.0;
        if (valid) { w[lane] = U; w[WAVE + lane] = lpu; w[2 * WAVE + lane] = lmu; w[3 * WAVE + lane] = lpd; w[4 * WAVE + lane] = lmd; }
    }
    IpmResult out = {U, mu, res, iters, status, rec_ok && status == 0};
    return out;
}

}  // namespace mpc
