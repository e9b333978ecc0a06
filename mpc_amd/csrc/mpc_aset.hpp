// Active-set solve of the condensed MPC problem in the space of the multipliers (gfx950, wave64), for the fixed-model
// kernels: the problem's Hessian H is the same for every scenario and step, so inv(H) and its products with the
// steering rows are condensed once (k_condense_fixed) and a solve works on the few active bounds only.
//
//   min 1/2 U'HU + g'U   s.t.  |u_i| <= umax,  lo_k <= dt * sum_{j<=k} u_j <= up_k        (LMPC2.py:196-260)
//
// With the unconstrained minimiser Uu = -inv(H) g and an active set of m bounds  E U = b  (rows of E: a unit row for an
// input bound, dt * prefix-sum row for a steering bound), the minimiser on that set is
//   U = Uu - inv(H) E' nu,     (E inv(H) E') nu = E Uu - b,
// an m x m positive definite system whose entries are read from the table T = [I; P] inv(H) [I P'] (P = the steering
// rows), 2N x 2N, symmetric.
// Most rows of a large set fix one input each (an input at its rate bound; inside a run of active steering bounds, the
// input between two of them): with F the fixed inputs, R the free ones and P the remaining steering rows, the same
// point solves the bordered system in the free inputs
//   [H_RR  P_R'] [x_R]   [-H_RF x_F          ]
//   [P_R   0   ] [mu ] = [b_P - P Uu - P_F x_F],      x = U - Uu,  x_F = (fixed value) - Uu_F,
// and the multipliers of the fixed inputs follow from  H x + E' nu = 0.  An iteration takes whichever system is
// smaller (|R| + |P| against |F| + |P| rows): with N inputs that is at most N / 2 + |P|.
// The set is found by exchanging bounds (per iteration the bound whose multiplier has the
// wrong sign by the most leaves, the bound the new point violates by the most enters) started from a prediction of
// the set out of the previous two closed-loop steps' solves (aset_predict; the previous step's set moved one stage
// ahead when only one is at hand); when nothing is left to exchange the KKT conditions hold, i.e. U is the unique
// minimiser.  The result is then refined through the same factorisation (what it misses its active bounds
// by) and certified by the size of that correction; optionally the correction is the Newton step computed from H
// itself (ASET_CERT_FROM_H).  Whatever does not end that way within a few iterations (sets larger than the capacity,
// a cycling iteration) is left to the interior point.
#pragma once
#include <utility>
#include "mpc_wave.hpp"

#pragma clang fp contract(fast)

namespace mpc {

typedef __attribute__((address_space(1))) double aset_gdouble;
typedef __attribute__((address_space(3))) double aset_ldouble;
typedef __attribute__((address_space(3))) int aset_lint;

// iteration caps (counted over the whole solve): an iteration of the largest size class costs about as much as one of
// the interior point, so a set of that size that has not settled after a few exchanges is handed on
#ifndef ASET_MAX_ITER_N
#define ASET_MAX_ITER_N 10
#endif
constexpr int ASET_MAX_ITER = ASET_MAX_ITER_N;       // size classes up to 32
constexpr int ASET_CAP_LARGE = 6;       // the larger classes; the caller passes its own (`cap_large`): the output-feedback
constexpr int ASET_CAP_LARGE_NOISY = 4; // loop, whose sets jump with the estimate's noise, hands on earlier (measured)
// After a solve that was handed on the next 2^min(failures in a row, ASET_BACKOFF_LOG2) - 1 constrained steps of the
// scenario skip the method.  0 = never wait: with the free-input form an attempt costs a few small systems, far less than
// the interior-point solve it may save (measured on the noisy output-feedback workload, where 4 in 10 attempts are
// handed on: 9.6 -> 11.8 M scenario-steps/s without the wait; 62 -> 67 M on the skidpad driver window).
#ifndef ASET_BACKOFF_LOG2
#define ASET_BACKOFF_LOG2 0
#endif
constexpr int ASET_COLD_MAX = 16;      // largest violated set a solve without a previous set starts from
// sizes of the certificate's corrections relative to the input bound: a first correction below EARLY is taken as final
// (what is left after it is smaller by the accuracy of the cached inverse), a later one has to come out below TOL
#ifndef ASET_DU_EARLY
#define ASET_DU_EARLY 1e-6
#endif
constexpr double ASET_DU_TOL = 1e-10;
// what a point may miss its active bounds by (relative to the bound) without being refined
constexpr double ASET_MISS_TOL = 1e-12;
// 1: the certificate's correction is the Newton step computed from H itself instead of the refinement through the table
#ifndef ASET_CERT_FROM_H
#define ASET_CERT_FROM_H 0
#endif
// entries of a column of H (or of its inverse) requested at a time in the certificate: one memory round trip per chunk, two
// registers per entry while it waits
// rows of the table requested at a time when the inputs are updated (smallest size class / the others)
#ifndef ASET_ROWS_SMALL
#define ASET_ROWS_SMALL 8
#endif
#ifndef ASET_ROWS_LARGE
#define ASET_ROWS_LARGE 8
#endif
#ifndef ASET_GATHER_CHUNK
#define ASET_GATHER_CHUNK 16
#endif
#ifndef ASET_COL_CHUNK
#define ASET_COL_CHUNK 10
#endif
// A pivot below this fraction of the row's own diagonal entry: the bound's row depends on the rows before it
constexpr double ASET_PIV_MIN = 1e-9;

// The system is solved in one of five sizes, 8 .. 32 and 48 rows (straight-line code each, rows beyond the system are unit
// rows).  With the smaller of the two forms per iteration a system has at most N / 2 rows plus its steering rows.
constexpr int aset_class_rows(int i) { return i < 4 ? 8 * (i + 1) : 48; }
template <int N> constexpr int aset_class(int i) { return 2 * N < aset_class_rows(i) ? 2 * N : aset_class_rows(i); }
constexpr int ASET_CLASSES = 5;
// entries of the lists the solve keeps in LDS: the rows of the largest system, or every input (the fixed inputs of the
// free-input form)
template <int N> constexpr int aset_list_rows() { return aset_class<N>(ASET_CLASSES - 1) > N + (N & 1) ? aset_class<N>(ASET_CLASSES - 1) : N + (N & 1); }
// LDS the solve borrows from the wave's area (4 (N + 1) doubles: references and speeds, dead once the unconstrained
// minimiser is formed): the list of active bounds (ints), one list of doubles and, where it fits, the 64-double slot the
// larger classes broadcast their pivot columns through
template <int N> constexpr int aset_list_doubles() { return aset_list_rows<N>() / 2 + aset_list_rows<N>(); }
template <int N> constexpr bool aset_lds_broadcast() { return aset_list_doubles<N>() + 64 <= 4 * (N + 1); }

// LDL' of an MM x MM matrix, row r in the registers of lane r (statically indexed), the pivot column broadcast through
// v_readlane.  A step whose pivot shows that the row depends on the rows before it is skipped: the multiplier of that
// row comes out as zero and the lane is left with dinv = 0 as the mark.  Pivots may be negative (a positive definite
// block bordered by constraint rows, eliminated in that order, needs no pivoting).
template <int MM>
struct SmallLdl {
    double row[MM];
    double dinv;

    // Entry j of the pivot column reaches every lane through v_readlane (two VALU issues per value) or, with USE_LDS, is
    // read back from a 64-double LDS slot `bc` that every lane has written its entry to: all but the first few columns
    // of a step (a multiple of the group size goes through LDS), the reads issued a group ahead of their multiply-adds.
    template <int K, bool USE_LDS>
    __device__ __forceinline__ void eliminate(int lane, int lane_in, double diag, aset_ldouble *bc)
    {
        constexpr int LG = 8;
        constexpr int M = MM - 1 - K;
        constexpr int NL = (USE_LDS && M >= 2 * LG) ? (M / LG) * LG : 0;
        constexpr int JL = MM - NL;
        const double c = row[K];
        if (NL) bc[lane_in] = c;
        const double piv = bcast(c, K);
        const double inv = fabs(piv) > ASET_PIV_MIN * bcast(diag, K) ? fast_rcp(piv) : 0.0;   // (negative: the bordered system of the free-input form)
        const bool below = lane > K;
        const double l = below ? c * inv : 0.0;
        if (lane == K) dinv = inv;
        double t[2][LG];
        if (NL) {
#pragma unroll
            for (int u = 0; u < LG; u++) t[0][u] = bc[JL + u];
        }
#pragma unroll
        for (int j0 = K + 1; j0 < JL; j0 += 4) {
            double cj[4];
#pragma unroll
            for (int g = 0; g < 4; g++)
                if (j0 + g < JL) cj[g] = bcast_pinned(c, j0 + g);
#pragma unroll
            for (int g = 0; g < 4; g++)
                if (j0 + g < JL) row[j0 + g] = fma(-l, cj[g], row[j0 + g]);
            __builtin_amdgcn_sched_barrier(0);
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
    __device__ __forceinline__ void eliminate_all(int lane, int lane_in, double diag, aset_ldouble *bc, std::integer_sequence<int, Ks...>)
    {
        (eliminate<Ks, USE_LDS>(lane, lane_in, diag, bc), ...);
    }

    template <bool USE_LDS>
    __device__ __forceinline__ void factor(int lane_in, double diag, aset_ldouble *bc)
    {
        dinv = 1.0;
        eliminate_all<USE_LDS>(opaque_int(lane_in), lane_in, diag, bc, std::make_integer_sequence<int, MM>{});
    }

    __device__ __forceinline__ double solve(double b, int lane_in) const
    {
        const int lane = opaque_int(lane_in);
        double w = b;
#pragma unroll
        for (int k = 0; k < MM - 1; k++) {
            const double t = fma(-row[k], bcast(w, k), w);
            w = lane > k ? t : w;
        }
#pragma unroll
        for (int k = MM - 1; k > 0; k--) {
            const double t = fma(-row[k], bcast(w * dinv, k), w);
            w = lane < k ? t : w;
        }
        return w * dinv;
    }
};

// acc + sign * sum_{s < m} w_s * T[bound of lane s][col]: the columns of inv(H) [I P'] of the set's bounds against their
// multipliers (w in lanes 0..m-1, the bounds' rows in my_cid).  CH rows are requested before the first is used: one
// memory round trip per CH bounds instead of one per bound.
template <unsigned LD, int CH>
__device__ __forceinline__ double aset_combine(const aset_gdouble *Tg, int my_cid, double w, int m, unsigned col, double acc, double sign)
{
    double acc1 = 0.0;
#pragma unroll 1
    for (int s0 = 0; s0 < m; s0 += CH) {
        double t[CH], n[CH];
#pragma unroll
        for (int g = 0; g < CH; g++) {
            const int sl = min(s0 + g, m - 1);
            const int cs = __builtin_amdgcn_readlane(my_cid, sl);
            t[g] = (Tg + (unsigned)cs * LD)[col];
            n[g] = (s0 + g < m) ? bcast(w, sl) : 0.0;
        }
#pragma unroll
        for (int g = 0; g < CH; g += 2) {
            acc = fma(sign * n[g], t[g], acc);
            acc1 = fma(sign * n[g + 1], t[g + 1], acc1);
        }
    }
    return acc + acc1;
}

// value of `v` in lane `src` (per-lane index, 0..63) through the LDS crossbar
__device__ __forceinline__ int lane_read(int v, int src) { return __builtin_amdgcn_ds_bpermute(src << 2, v); }
__device__ __forceinline__ double lane_read(double v, int src)
{
    const int lo = __builtin_amdgcn_ds_bpermute(src << 2, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(src << 2, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

// The multiplier of this lane's steering bound from the multipliers of the rows the system was solved with.  A run of
// active steering bounds k1..k2 (same side) enters the system as the prefix-sum row of k1 (multiplier mu) and the unit rows
// of the inputs k1+1..k2 (multipliers eta_j): the same constraints, but rows that do not grow ever more parallel along
// the run.  sum_k nu_k prefix_k = mu prefix_k1 + sum_j eta_j e_j  gives  nu_j = (eta_j - eta_{j+1}) / dt  inside the run
// (eta beyond its end = 0) and  nu_k1 = mu - eta_{k1+1} / dt.
__device__ __forceinline__ double aset_steer_multiplier(int lane, int ad, bool in_run, double tm, double dt)
{
    const int nxt = min(lane + 1, WAVE - 1);
    const bool next_in_run = lane < WAVE - 1 && lane_read((int)in_run, nxt) != 0;
    const double tn = lane_read(tm, nxt);
    const double eta_next = next_in_run ? tn : 0.0;
    if (ad == 0) return 0.0;
    return in_run ? (tm - eta_next) / dt : tm - eta_next / dt;
}

enum { ASET_SOLVED = 0, ASET_GROW = 1, ASET_UNSETTLED = 3, ASET_UNCERTIFIED = 4 };

// Developer builds with -DASET_TIMING: shader-clock cycles per phase of the solve, summed over the waves of a launch
// (g_aset_t, read through mpc_debug_aset_timing; run few scenarios so that a wave has its SIMD to itself).
#ifdef ASET_TIMING
static __device__ unsigned long long g_aset_t[24];      // (one per translation unit; the reader sits in mpc_solver.hip)
#define AT_BEGIN() long long at0_ = clock64()
#define AT(i) do { const long long at1_ = clock64(); if ((threadIdx.x & 63) == 0) { atomicAdd(&g_aset_t[i], (unsigned long long)(at1_ - at0_)); } at0_ = clock64(); } while (0)
#define AT_COUNT(i) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_aset_t[i], 1ull); } while (0)
#else
#define AT_BEGIN() do {} while (0)
#define AT(i) do {} while (0)
#define AT_COUNT(i) do {} while (0)
#endif

// History of a scenario's solves behind its warm-start record (the record itself: [5][WAVE] inputs and multipliers, a
// validity flag and two back-off counters): two slots of [4][WAVE] doubles -- signed multipliers of the input and of
// the steering bounds (positive at the upper bound, zero where the bound is not active), and the distance of every
// input and steering angle from its nearer bound relative to the bound -- for the last two steps this method solved
// in a row, and one word `hist` = depth | newest slot << 2 (depth 0: the slots say nothing, 1: the newest describes the
// previous step, 2: the other one the step before it as well).
constexpr int ASET_HIST_WORD = 5 * WAVE + 3;
constexpr int ASET_HIST_BASE = 5 * WAVE + 8;
constexpr int ASET_HIST_SLOT = 4 * WAVE;

// What one row (a bound of stage i) does at the next step, judged in one frame of reference from its state at the last
// two steps there: +-1 stays active on that side, 0 inactive, 2 becomes active.  nu: signed multiplier (0 = not
// active), s: distance from the bound; _t the previous step, _p the one before.  An active row's multiplier and an
// inactive row's distance are continued linearly; a row that has just changed keeps its new state.
__device__ __forceinline__ int aset_continue(double nu_t, double s_t, double nu_p, double s_p)
{
    const int a_t = nu_t > 0.0 ? 1 : (nu_t < 0.0 ? -1 : 0), a_p = nu_p > 0.0 ? 1 : (nu_p < 0.0 ? -1 : 0);
    if (a_t != 0) return (a_p != a_t || (2.0 * nu_t - nu_p) * (double)a_t > 0.0) ? a_t : 0;
    return (a_p == 0 && !(2.0 * s_t - s_p > 0.0)) ? 2 : 0;
}

// The set the next solve starts from, for one kind of bound (input or steering; `last` = its last stage), out of the
// two history slots `cur` (previous step) and `prv` (the step before): nu / s arrays of one kind, WAVE apart as laid
// out above.  Between two closed-loop steps the solution moves one stage ahead where it follows the plan, and stays
// where it is tied to the end of the horizon; most ends of a run of active bounds do something in between.  A row
// inside a run or far from one is the same in both frames; at the end of a run the row is taken as active when both
// frames -- row i continued from (i+1, i+2) of the last two steps, and from (i, i) -- say so.  A wrong guess costs an
// exchange iteration, nothing else.
__device__ __forceinline__ int aset_predict(const aset_gdouble *cur_nu, const aset_gdouble *cur_s, const aset_gdouble *prv_nu,
                                            const aset_gdouble *prv_s, int lane, int last)
{
    const int i0 = min(lane, last), i1 = min(lane + 1, last), i2 = min(lane + 2, last);
    const bool in1 = lane + 1 <= last, in2 = lane + 2 <= last;
    // (all twelve requests go out before the first is used)
    const double nt_x = cur_nu[i0], st_x = cur_s[i0], np_x = prv_nu[i0], sp_x = prv_s[i0];
    double nt_t = cur_nu[i1], st_t = cur_s[i1], np_t = prv_nu[i2], sp_t = prv_s[i2];
    if (!in1) { nt_t = 0.0; st_t = 1.0; }
    if (!in2) { np_t = 0.0; sp_t = 1.0; }
    const int A = nt_t > 0.0 ? 1 : (nt_t < 0.0 ? -1 : 0), B = nt_x > 0.0 ? 1 : (nt_x < 0.0 ? -1 : 0);
    if (A == B) return A;
    const int pa = aset_continue(nt_t, st_t, np_t, sp_t), pb = aset_continue(nt_x, st_x, np_x, sp_x);
    if (pa == 0 || pb == 0) return 0;
    return pa != 2 ? pa : (A != 0 ? A : B);
}

struct AsetResult {
    double U;          // the minimiser (this lane's input), valid when why == ASET_SOLVED
    int au, ad;        // this lane's input / steering bound in the set: +1 upper, -1 lower, 0 none
    int iters;         // active-set iterations spent so far
    int flops;         // floating-point operations spent so far (the method's own count: LDL' m^3/3, two solves 2m^2,
                       // the update of the inputs 2mN, scans ~20N per iteration; 2m^2 + 2mN per refinement pass, 4N^2 more from H)
    int why;           // ASET_SOLVED, or handed on: GROW the set outgrew this size class, UNSETTLED the iteration did not
                       // settle within ASET_MAX_ITER, UNCERTIFIED the corrections of the certificate stayed too large
};

#ifndef ASET_FREE_FORM
#define ASET_FREE_FORM 1
#endif

// Rows of the system an iteration solves for the set (au, ad): the smaller of the two forms above; without the table
// (DUAL = false: kernels whose H differs per scenario and solve) the free-input form, which needs H alone.
template <int N, bool DUAL>
__device__ __forceinline__ int aset_system_rows(int lane, int au, int ad)
{
    const int ad_prev = lane > 0 ? lane_read(ad, max(lane - 1, 0)) : 0;
    const bool in_run = ad != 0 && ad_prev == ad;
    const int n_fix = __popcll(__ballot(au != 0 || in_run)), n_pre = __popcll(__ballot(ad != 0 && !in_run));
    if (!DUAL) return N - n_fix + n_pre;
    return ASET_FREE_FORM ? min(n_fix, N - n_fix) + n_pre : n_fix + n_pre;
}

// The iteration in one size class, from the set (au, ad) and iteration count `it0`.  Uu: this lane's unconstrained
// minimiser (0 beyond the horizon), vdu = dt * prefix sum of Uu; up_d / lo_d: this lane's steering bounds
// (steer_bounds()); T: the table above, row length 2N (not read when DUAL is false); Hm: H, [N][N] symmetric; rec: the scenario's warm-start record
// ([5][WAVE]: inputs, then the multipliers of the upper / lower input bounds and upper / lower steering bounds),
// written with the result when the solve succeeds, together with the history slot and word `hist` names (or NULL); lds:
// the wave's 4 (N + 1) doubles of references and speeds.  g0: this lane's entry of the objective's gradient at Uu -- zero
// when Uu is the unconstrained minimiser; the free-input form works about any point (DUAL = false: e.g. the current
// iterate of an SQP method, which saves the factorisation of H that the unconstrained minimiser costs).
// Inlined into the caller's one non-inlined function of the constrained step (solve_constrained in mpc_solver.hip), whose
// registers stay out of the per-step path's allocation.
template <int N, int MM, bool DUAL = true>
__device__ __forceinline__ AsetResult aset_body(int lane, const double *T, const double *Hm, double Uu, double vdu,
                                                double up_d, double lo_d, double umax, double dmax, double dt,
                                                double *rec, int au, int ad, int it0, int fl0, double *lds, int cap_large, int hist,
                                                double g0)
{
    constexpr unsigned LD = 2 * N;                       // row length of T
    const bool valid = lane < N, validd = lane < N - 1;
    const unsigned lc = valid ? lane : 0;
    aset_lint *cid_l = (aset_lint *)lds;
    aset_ldouble *val_l = (aset_ldouble *)(lds + aset_list_rows<N>() / 2);
    aset_ldouble *bc = (aset_ldouble *)(lds + aset_list_doubles<N>());
    constexpr bool USE_LDS = MM >= 24 && aset_lds_broadcast<N>();
    const aset_gdouble *Tg = (const aset_gdouble *)T;
    const aset_gdouble *Hg = (const aset_gdouble *)Hm;
    const double tol_u = 1e-12 * umax, tol_d = 1e-12 * dmax;

    SmallLdl<MM> ldl;
    AsetResult out = {0.0, au, ad, it0, fl0, ASET_UNSETTLED};
    double U = Uu, nu_u = 0.0, nu_d = 0.0, nu = 0.0, tm = 0.0, b_run = 0.0;
    int my_cid = 0, m = 0, rank_u = 0, rank_d = 0;
    bool converged = false, in_run = false;
    // the free-input form's state of the last iteration
    unsigned long long mask_fix = 0ull, mask_pre = 0ull;
    int n_free = 0, msys = 0, rank_f = 0, rank_p = 0;
    bool free_form = false;
    double res_free = 0.0;
    const int it_cap = MM <= 32 ? ASET_MAX_ITER : cap_large;
    AT_BEGIN();
#pragma unroll 1
    for (int it = it0; it < it_cap; it++) {
        AT_COUNT(16);
        // a steering bound whose predecessor stage is active on the same side enters as "this input = the step between
        // the two bounds" (0 unless the bound was widened): see aset_steer_multiplier.  Such an input cannot sit at its
        // rate bound as well (the step between two steering bounds is never the full rate): that bound leaves the set.
        {
            const int prv = max(lane - 1, 0);
            const double b_d = ad > 0 ? up_d : lo_d;
            const int ad_prev = lane > 0 ? lane_read(ad, prv) : 0;
            const double b_prev = lane_read(b_d, prv);
            in_run = ad != 0 && ad_prev == ad;
            b_run = in_run ? (b_d - b_prev) / dt : b_d;
            if (in_run) au = 0;
        }
        const unsigned long long mask_u = __ballot(au != 0), mask_d = __ballot(ad != 0);
        const int m_u = __popcll(mask_u);
        m = m_u + __popcll(mask_d);
        const unsigned long long mask_run = __ballot(in_run);
        mask_fix = mask_u | mask_run;                    // inputs the set fixes (an in_run lane has au = 0)
        mask_pre = mask_d & ~mask_run;                   // steering rows that stay rows
        n_free = N - __popcll(mask_fix);
        free_form = !DUAL || (ASET_FREE_FORM && n_free + __popcll(mask_pre) < m);
        msys = free_form ? n_free + __popcll(mask_pre) : m;
        if (msys > MM) { out.au = au; out.ad = ad; out.iters = it; out.why = ASET_GROW; return out; }
        out.iters = it + 1;
        out.flops += msys * msys * msys / 3 + 2 * msys * msys + 2 * m * N + 20 * N;
        const unsigned long long below = (1ull << lane) - 1ull;
        if (free_form) {
            // ---- the free-input form
            const bool is_fix = (mask_fix >> lane) & 1ull, is_pre = (mask_pre >> lane) & 1ull, is_free = valid && !is_fix;
            const int rank_x = __popcll(mask_fix & below);
            rank_f = __popcll(~mask_fix & below);                             // (lanes below a valid one are valid)
            rank_p = n_free + __popcll(mask_pre & below);
            const double xF = is_fix ? (au != 0 ? (au > 0 ? umax : -umax) : b_run) - Uu : 0.0;
            // y = H[:, F] x_F for every stage: the fixed inputs and their steps in lanes 0..|F|-1
            if (is_fix) { cid_l[rank_x] = lane; val_l[rank_x] = xF; }
            wave_sync();
            const int n_fix = N - n_free;
            const int cidF = lane < n_fix ? cid_l[lane] : 0;
            const double wF = lane < n_fix ? val_l[lane] : 0.0;
            wave_sync();
            const double y = valid ? aset_combine<(unsigned)N, ASET_ROWS_LARGE>(Hg, cidF, wF, n_fix, lc, 0.0, 1.0) : 0.0;
            const double pxF = dt * incl_prefix(xF, lane);
            // the system's rows in lanes 0..msys-1: free inputs (their stage), then steering rows (N + stage)
            if (is_free) { cid_l[rank_f] = lane; val_l[rank_f] = -(y + g0); }
            if (is_pre) { cid_l[rank_p] = N + lane; val_l[rank_p] = b_run - vdu - pxF; }
            wave_sync();
            const bool act = lane < msys;
            my_cid = act ? cid_l[lane] : 0;
            const double my_d = act ? val_l[lane] : 0.0;
            wave_sync();
            AT(0);
            const bool row_free = my_cid < N;
            const int my_stage = row_free ? my_cid : my_cid - N;
            // (the steering row's own entry of the table bounds its pivot: inv(H_RR) <= inv(H)_RR; without the table, the
            // share of one free input of that stage's size)
            const double hdiag = Hg[(unsigned)my_stage * N + (unsigned)my_stage];
            double diag0 = 1.0;
            if (DUAL) { if (act) diag0 = row_free ? hdiag : Tg[(unsigned)my_cid * LD + (unsigned)my_cid]; }
            else if (act) diag0 = row_free ? hdiag : dt * dt / hdiag;
            const int lane_o = opaque_int(lane);
#pragma unroll
            for (int s0 = 0; s0 < MM; s0 += ASET_GATHER_CHUNK) {
                double t[ASET_GATHER_CHUNK];
#pragma unroll
                for (int g = 0; g < ASET_GATHER_CHUNK; g++) {
                    if (s0 + g < MM) {
                        const int cs = __builtin_amdgcn_readlane(my_cid, s0 + g);
                        t[g] = (Hg + (cs < N ? cs : 0) * N)[row_free ? my_stage : 0];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < ASET_GATHER_CHUNK; g++) {
                    if (s0 + g < MM) {
                        const int cs = __builtin_amdgcn_readlane(my_cid, s0 + g);
                        const bool col_free = cs < N;
                        const int cstage = col_free ? cs : cs - N;
                        const double border = (row_free != col_free && (row_free ? my_stage <= cstage : cstage <= my_stage)) ? dt : 0.0;
                        const double v = (row_free && col_free) ? t[g] : border;
                        ldl.row[s0 + g] = (act && s0 + g < msys) ? v : (lane_o == s0 + g ? 1.0 : 0.0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            AT(1);
            ldl.template factor<USE_LDS>(lane, diag0, bc);
            AT(2);
            const double sol = ldl.solve(my_d, lane);
            // (a steering row no free input stands before -- the fixed ones decide that angle -- was left out: it leaves)
            if (act) val_l[lane] = ldl.dinv == 0.0 ? __builtin_nan("") : sol;
            wave_sync();
            const double xR = is_free ? val_l[rank_f] : 0.0;
            const double muP = is_pre ? val_l[rank_p] : 0.0;
            wave_sync();
            U = valid ? Uu + (is_fix ? xF : xR) : 0.0;
            AT(3);
            // H x for every stage, and with it the multipliers of the fixed inputs:  nu_F = -(H x + P' mu)_F
            const double solc = (act && row_free && ldl.dinv != 0.0) ? sol : 0.0;
            const double hx = valid ? aset_combine<(unsigned)N, ASET_ROWS_LARGE>(Hg, my_cid, solc, n_free, lc, y, 1.0) : 0.0;
            const double force = hx + g0 + dt * incl_suffix((is_pre && muP == muP) ? muP : 0.0, lane);
            res_free = is_free ? force : 0.0;
            nu_u = au != 0 ? -force : 0.0;
            tm = ad != 0 ? (in_run ? -force : muP) : 0.0;
            nu_d = aset_steer_multiplier(lane, ad, in_run, tm, dt);
            AT(4);
        } else if (DUAL) {
        rank_u = __popcll(mask_u & below);
        rank_d = m_u + __popcll(mask_d & below);
        // the active bounds in lanes 0..m-1: which bound, and by how much the unconstrained minimiser misses it
        if (au != 0) { cid_l[rank_u] = lane; val_l[rank_u] = Uu - (au > 0 ? umax : -umax); }
        if (ad != 0) { cid_l[rank_d] = in_run ? lane : N + lane; val_l[rank_d] = (in_run ? Uu : vdu) - b_run; }
        wave_sync();
        const bool act = lane < m;
        my_cid = act ? cid_l[lane] : 0;
        const double my_d = act ? val_l[lane] : 0.0;
        wave_sync();
        AT(0);
        // row `lane` of E inv(H) E', gathered from the table; lanes and columns beyond the set form a unit matrix
        const unsigned rowoff = (unsigned)my_cid * LD;
        const double diag0 = act ? Tg[rowoff + (unsigned)my_cid] : 1.0;
        const int lane_o = opaque_int(lane);
        // (entries requested ASET_GATHER_CHUNK at a time: with all of a large class in flight next to the row itself the
        // registers run out)
#pragma unroll
        for (int s0 = 0; s0 < MM; s0 += ASET_GATHER_CHUNK) {
            double t[ASET_GATHER_CHUNK];
#pragma unroll
            for (int g = 0; g < ASET_GATHER_CHUNK; g++) {
                if (s0 + g < MM) {
                    const int cs = __builtin_amdgcn_readlane(my_cid, s0 + g);
                    t[g] = (Tg + cs)[rowoff];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < ASET_GATHER_CHUNK; g++)
                if (s0 + g < MM) ldl.row[s0 + g] = (act && s0 + g < m) ? t[g] : (lane_o == s0 + g ? 1.0 : 0.0);
            __builtin_amdgcn_sched_barrier(0);
        }
        AT(1);
        ldl.template factor<USE_LDS>(lane, diag0, bc);
        AT(2);
        nu = ldl.solve(my_d, lane);
        // a bound whose row depends on the rows before it (an input run that already fixes a steering row, ...) was left
        // out of the system: it leaves the set here (not-a-number fails the sign test below) and comes back if violated
        if (act) val_l[lane] = ldl.dinv == 0.0 ? __builtin_nan("") : nu;
        wave_sync();
        nu_u = au != 0 ? val_l[rank_u] : 0.0;
        tm = ad != 0 ? val_l[rank_d] : 0.0;
        wave_sync();
        nu_d = aset_steer_multiplier(lane, ad, in_run, tm, dt);
        AT(3);
        // U = Uu - sum_s nu_s * (column of inv(H) [I P'] of bound s)
        U = valid ? aset_combine<LD, (MM <= 16 ? ASET_ROWS_SMALL : ASET_ROWS_LARGE)>(Tg, my_cid, nu, m, lc, Uu, -1.0) : 0.0;
        AT(4);
        }
        const double vd = dt * incl_prefix(U, lane);
        // next set: the one bound whose multiplier has the wrong sign by the most (or whose row was left out) goes, the
        // one bound the new point violates by the most comes in.  (Exchanging every such bound at once is quicker when
        // it settles, but from a set that is off by a stage at the end of a run it throws out half the run and cycles.)
        const bool bad_u = au != 0 && !(nu_u * (double)au >= 0.0), bad_d = ad != 0 && !(nu_d * (double)ad >= 0.0);
        const double sc_u = bad_u ? (nu_u != nu_u ? 1e300 : fabs(nu_u) * umax) : -1.0;
        const double sc_d = bad_d ? (nu_d != nu_d ? 1e300 : fabs(nu_d) * dmax) : -1.0;
        const double worst = wave_max(fmax(sc_u, sc_d));
        const double vi_u = (au == 0 && valid) ? (fabs(U) - umax - tol_u) / umax : -1.0;
        const double vi_d = (ad == 0 && validd) ? (fmax(vd - up_d, lo_d - vd) - tol_d) / dmax : -1.0;
        const double most = wave_max(fmax(vi_u, vi_d));
        if (!(worst >= 0.0) && !(most > 0.0)) { converged = true; AT(5); break; }
        if (worst >= 0.0) {
            const unsigned long long cu = __ballot(sc_u == worst), cd = __ballot(sc_d == worst);
            if (cu) { if (lane == __ffsll((long long)cu) - 1) au = 0; }
            else if (lane == __ffsll((long long)cd) - 1) ad = 0;
        }
        if (most > 0.0) {
            const unsigned long long cu = __ballot(vi_u == most), cd = __ballot(vi_d == most);
            if (cu) { if (lane == __ffsll((long long)cu) - 1) au = U > 0.0 ? 1 : -1; }
            else if (lane == __ffsll((long long)cd) - 1) ad = vd > up_d ? 1 : -1;
        }
        AT(5);
    }
    out.au = au; out.ad = ad;
    if (!converged) { AT_COUNT(21); return out; }
    out.why = ASET_UNCERTIFIED;

    // Refinement and certificate.  What the point misses its active bounds by (the rounding of the factorisation and of
    // the triangular solves, large where rows of the set are close to dependent) is put through the factorisation at
    // hand once more:  w = inv(S) (-(E U - b)),  nu -= w,  U += inv(H) E' w;  the size of that correction is the error
    // of the point it is applied to.  This trusts the table as the unconstrained steps trust the cached maps that
    // U_unc comes from (both are written by one launch of k_condense_fixed from one factorisation of H).
    // With ASET_CERT_FROM_H the correction is the full Newton step computed from H itself:
    //   r = H (U - Uu) + E' nu,  z = inv(H) r,  w = inv(S) (E z - (E U - b)),  nu -= w,  U += -z + inv(H) E' w
    // (two more passes over N x N matrices per solve, about a fifth of the solve's time, for an answer that differs
    // from the 80-bit checker's by the same 1e-10 of the input bound either way).
    bool certified = false;
    if (free_form) {
        // The free-input form: the residual of the free rows (at hand from the multipliers' H x) and what the point
        // misses its steering rows by go through the factorisation once more; the fixed inputs are exact by construction.
        const bool is_fix = (mask_fix >> lane) & 1ull, is_pre = (mask_pre >> lane) & 1ull, is_free = valid && !is_fix;
        const double vdc = dt * incl_prefix(valid ? U : 0.0, lane);
        if (is_free) val_l[rank_f] = -res_free;
        if (is_pre) val_l[rank_p] = b_run - vdc;
        wave_sync();
        const double rz = lane < msys ? val_l[lane] : 0.0;
        wave_sync();
        const double dsol = ldl.solve(rz, lane);
        if (lane < msys) val_l[lane] = dsol;
        wave_sync();
        const double dx = is_free ? val_l[rank_f] : 0.0;
        const double dmu = is_pre ? val_l[rank_p] : 0.0;
        wave_sync();
        U += dx;
        tm += dmu;
        nu_d = aset_steer_multiplier(lane, ad, in_run, tm, dt);
        out.flops += 2 * msys * msys;
        certified = wave_max(fabs(dx)) <= ASET_DU_EARLY * umax;
        AT_COUNT(17);
    }
#pragma unroll 1
    for (int pass = 0; DUAL && pass < 3 && !certified && !free_form; pass++) {
        double z = 0.0;
        if (ASET_CERT_FROM_H) {
        const double dU = valid ? U - Uu : 0.0;
        // (a column of H, then one of inv(H), in chunks of ASET_COL_CHUNK entries requested before the first is used)
        double hu0 = 0.0, hu1 = 0.0;
#pragma unroll
        for (int c0 = 0; c0 < N; c0 += ASET_COL_CHUNK) {
            double hc[ASET_COL_CHUNK];
#pragma unroll
            for (int j = 0; j < ASET_COL_CHUNK; j++)
                if (c0 + j < N) hc[j] = Hg[(c0 + j) * N + lc];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < ASET_COL_CHUNK; j += 2) {
                if (c0 + j < N) hu0 = fma(hc[j], bcast(dU, c0 + j), hu0);
                if (c0 + j + 1 < N) hu1 = fma(hc[j + 1], bcast(dU, c0 + j + 1), hu1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const double frc = nu_u + dt * incl_suffix(validd ? nu_d : 0.0, lane);      // (E' nu)[lane]
        const double r = valid ? hu0 + hu1 + frc : 0.0;
        double z0 = 0.0, z1 = 0.0;
#pragma unroll
        for (int c0 = 0; c0 < N; c0 += ASET_COL_CHUNK) {
            double gc[ASET_COL_CHUNK];
#pragma unroll
            for (int j = 0; j < ASET_COL_CHUNK; j++)
                if (c0 + j < N) gc[j] = Tg[(c0 + j) * LD + lc];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < ASET_COL_CHUNK; j += 2) {
                if (c0 + j < N) z0 = fma(gc[j], bcast(r, c0 + j), z0);
                if (c0 + j + 1 < N) z1 = fma(gc[j + 1], bcast(r, c0 + j + 1), z1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        z = valid ? z0 + z1 : 0.0;
        }
        const double pz = ASET_CERT_FROM_H ? dt * incl_prefix(z, lane) : 0.0;
        const double vdc = dt * incl_prefix(valid ? U : 0.0, lane);
        if (!ASET_CERT_FROM_H) {
            double miss = 0.0;
            if (au != 0) miss = fabs(U - (au > 0 ? umax : -umax)) / umax;
            if (ad != 0) miss = fmax(miss, in_run ? fabs(U - b_run) / umax : fabs(vdc - b_run) / dmax);
            if (wave_max(miss) <= ASET_MISS_TOL) { certified = true; break; }      // at rounding level: nothing to refine
        }
        out.flops += (ASET_CERT_FROM_H ? 4 * N * N : 0) + 2 * m * m + 2 * m * N;
        // right-hand side E z - (E U - b): the correction also closes what the point misses its active bounds by
        if (au != 0) val_l[rank_u] = z - (U - (au > 0 ? umax : -umax));
        if (ad != 0) val_l[rank_d] = in_run ? z - (U - b_run) : pz - (vdc - b_run);
        wave_sync();
        const double ez = lane < m ? val_l[lane] : 0.0;
        wave_sync();
        const double dn = ldl.solve(ez, lane);
        if (lane < m) val_l[lane] = dn;
        wave_sync();
        if (au != 0) nu_u -= val_l[rank_u];
        if (ad != 0) tm -= val_l[rank_d];
        wave_sync();
        nu_d = aset_steer_multiplier(lane, ad, in_run, tm, dt);
        const double corr = valid ? aset_combine<LD, (MM <= 16 ? ASET_ROWS_SMALL : ASET_ROWS_LARGE)>(Tg, my_cid, dn, m, lc, -z, 1.0) : 0.0;
        U += corr;
        const double cmax = wave_max(fabs(corr));
        certified = cmax <= (pass == 0 ? ASET_DU_EARLY : ASET_DU_TOL) * umax;
        AT_COUNT(17);
    }
    AT(6);
    if (!certified) { AT_COUNT(22); return out; }
    // the refined point against the bounds and the signs once more, at the accuracy of the certificate
    {
        const double force = nu_u + dt * incl_suffix(validd ? nu_d : 0.0, lane);
        const double scale = fmax(1.0, wave_max(fabs(valid ? force : 0.0)));
        const double vd = dt * incl_prefix(valid ? U : 0.0, lane);
        bool good = true;
        if (valid) good = fabs(U) <= umax * (1.0 + 1e-11) && nu_u * (double)au >= -1e-11 * scale;
        if (validd) good = good && vd <= up_d + 1e-11 * dmax && vd >= lo_d - 1e-11 * dmax && nu_d * (double)ad >= -1e-11 * scale;
        if (!__all(good)) { AT_COUNT(23); return out; }
    }
    U = valid ? fmin(fmax(U, -umax), umax) : 0.0;
    if (rec && valid) {
        aset_gdouble *w = (aset_gdouble *)rec;
        const double pu = au > 0 ? fmax(nu_u, 0.0) : 0.0, mu = au < 0 ? fmax(-nu_u, 0.0) : 0.0;
        const double pd = ad > 0 ? fmax(nu_d, 0.0) : 0.0, md = ad < 0 ? fmax(-nu_d, 0.0) : 0.0;
        w[lane] = U;
        w[WAVE + lane] = pu;
        w[2 * WAVE + lane] = mu;
        w[3 * WAVE + lane] = pd;
        w[4 * WAVE + lane] = md;
        aset_gdouble *hs = w + ASET_HIST_BASE + ((hist >> 2) & 1) * ASET_HIST_SLOT;
        const double vdf = dt * incl_prefix(U, lane);
        hs[lane] = pu - mu;
        hs[WAVE + lane] = pd - md;
        hs[2 * WAVE + lane] = (umax - fabs(U)) / umax;
        hs[3 * WAVE + lane] = fmin(up_d - vdf, vdf - lo_d) / dmax;
        if (lane == 0) w[ASET_HIST_WORD] = (double)hist;
    }
    AT(7);
    out.U = U; out.why = ASET_SOLVED;
    return out;
}

// The size classes by name.  They are part of the caller's one non-inlined function of the constrained step: a call level
// of their own costs another save and restore of the callee-saved registers per solve (measured: 2% of the c2 window).
#ifndef ASET_CLASS_LINKAGE
#define ASET_CLASS_LINKAGE __forceinline__
#endif
#define MPC_ASET_ARGS int lane, const double *T, const double *Hm, double Uu, double vdu, double up_d, double lo_d, double umax, \
                      double dmax, double dt, double *rec, int au, int ad, int it0, int fl0, double *lds, int cap_large, int hist, double g0
#define MPC_ASET_PASS lane, T, Hm, Uu, vdu, up_d, lo_d, umax, dmax, dt, rec, au, ad, it0, fl0, lds, cap_large, hist, g0
template <int N, int MM, int TAG, bool DUAL>
static __device__ ASET_CLASS_LINKAGE AsetResult aset_run_small(MPC_ASET_ARGS)
{
    return aset_body<N, MM, DUAL>(MPC_ASET_PASS);
}
template <int N, int MM, int TAG, bool DUAL>
static __device__ ASET_CLASS_LINKAGE AsetResult aset_run_medium(MPC_ASET_ARGS)
{
    return aset_body<N, MM, DUAL>(MPC_ASET_PASS);
}
template <int N, int MM, int TAG, bool DUAL>
static __device__ ASET_CLASS_LINKAGE AsetResult aset_run_large(MPC_ASET_ARGS)
{
    return aset_body<N, MM, DUAL>(MPC_ASET_PASS);
}
template <int N, int MM, int TAG, bool DUAL>
static __device__ ASET_CLASS_LINKAGE AsetResult aset_run_full(MPC_ASET_ARGS)
{
    return aset_body<N, MM, DUAL>(MPC_ASET_PASS);
}
template <int N, int MM, int TAG, bool DUAL>
static __device__ ASET_CLASS_LINKAGE AsetResult aset_run_huge(MPC_ASET_ARGS)
{
    return aset_body<N, MM, DUAL>(MPC_ASET_PASS);
}
#undef MPC_ASET_ARGS
#undef MPC_ASET_PASS

// One solve: the starting set, then the iteration in the smallest size class that holds the set, moving up a class when
// the set outgrows it.  warm_mode 0: start from what the unconstrained minimiser violates; 1: from the set in the record
// (the previous solve of the same closed-loop step); 2: the record is the previous closed-loop step's -- aset_predict
// when its history `hist_in` holds two steps, else its set one stage ahead.  When Uu is not the unconstrained minimiser
// (g0 != 0) only the starts from a record make sense.  PREDICT = false compiles the prediction out (callers that never
// have two consecutive steps in the record).  Other arguments as aset_body.
template <int N, int TAG, bool DUAL = true, bool PREDICT = true>
__device__ __forceinline__ AsetResult active_set_solve(int lane, const double *T, const double *Hm, double Uu, double vdu,
                                                       double up_d, double lo_d, double umax, double dmax, double dt,
                                                       double *rec, int warm_mode, double *lds, int cap_large, int hist_in,
                                                       double g0 = 0.0)
{
    const bool rec_valid = warm_mode != 0;
    constexpr int C0 = aset_class<N>(0), C1 = aset_class<N>(1), C2 = aset_class<N>(2), C3 = aset_class<N>(3), C4 = aset_class<N>(4);
    const bool valid = lane < N, validd = lane < N - 1;
    int au = 0, ad = 0;
    AT_BEGIN();
    // the history after this solve: one step more when the previous step was solved here too, else this step alone
    const bool chain = warm_mode == 2 && (hist_in & 3) != 0;
    const int cur = (hist_in >> 2) & 1;
    const int hist = chain ? (min((hist_in & 3) + 1, 2) | ((cur ^ 1) << 2)) : 1;
    if (PREDICT && chain && (hist_in & 3) == 2) {
        const aset_gdouble *hc = (const aset_gdouble *)rec + ASET_HIST_BASE + cur * ASET_HIST_SLOT;
        const aset_gdouble *hp = (const aset_gdouble *)rec + ASET_HIST_BASE + (cur ^ 1) * ASET_HIST_SLOT;
        au = aset_predict(hc, hc + 2 * WAVE, hp, hp + 2 * WAVE, lane, N - 1);
        ad = aset_predict(hc + WAVE, hc + 3 * WAVE, hp + WAVE, hp + 3 * WAVE, lane, N - 2);
    } else if (rec_valid) {
        const aset_gdouble *w = (const aset_gdouble *)rec;
        const int lc = valid ? lane : 0;
        const int sh = warm_mode == 2 ? 1 : 0;
        const int ln = min(lc + sh, N - 1), lnd = min(lc + sh, N - 2);
        const double a0 = w[WAVE + ln], a1 = w[2 * WAVE + ln], a2 = w[3 * WAVE + lnd], a3 = w[4 * WAVE + lnd];
        au = a0 > 0.0 ? 1 : (a1 > 0.0 ? -1 : 0);
        ad = a2 > 0.0 ? 1 : (a3 > 0.0 ? -1 : 0);
    } else {
        au = Uu > umax ? 1 : (Uu < -umax ? -1 : 0);
        ad = vdu > up_d ? 1 : (vdu < lo_d ? -1 : 0);
    }
    if (!valid) au = 0;
    if (!validd) ad = 0;
    wave_sync();                                         // the borrowed LDS area is still being read as something else
    AT(8);
    AsetResult r = {0.0, au, ad, 0, 0, ASET_GROW};
    if (!rec_valid && __popcll(__ballot(au != 0)) + __popcll(__ballot(ad != 0)) > ASET_COLD_MAX) return r;   // a large violated set says little about the active one
    int m = aset_system_rows<N, DUAL>(lane, au, ad);
    if (m <= C0) r = aset_run_small<N, C0, TAG, DUAL>(lane, T, Hm, Uu, vdu, up_d, lo_d, umax, dmax, dt, rec, r.au, r.ad, r.iters, r.flops, lds, cap_large, hist, g0);
    if (C1 > C0 && r.why == ASET_GROW) {
        m = aset_system_rows<N, DUAL>(lane, r.au, r.ad);
        if (m <= C1) r = aset_run_medium<N, C1, TAG, DUAL>(lane, T, Hm, Uu, vdu, up_d, lo_d, umax, dmax, dt, rec, r.au, r.ad, r.iters, r.flops, lds, cap_large, hist, g0);
    }
    if (C2 > C1 && r.why == ASET_GROW) {
        m = aset_system_rows<N, DUAL>(lane, r.au, r.ad);
        if (m <= C2) r = aset_run_large<N, C2, TAG, DUAL>(lane, T, Hm, Uu, vdu, up_d, lo_d, umax, dmax, dt, rec, r.au, r.ad, r.iters, r.flops, lds, cap_large, hist, g0);
    }
    if (C3 > C2 && r.why == ASET_GROW) {
        m = aset_system_rows<N, DUAL>(lane, r.au, r.ad);
        if (m <= C3) r = aset_run_full<N, C3, TAG, DUAL>(lane, T, Hm, Uu, vdu, up_d, lo_d, umax, dmax, dt, rec, r.au, r.ad, r.iters, r.flops, lds, cap_large, hist, g0);
    }
    if (C4 > C3 && r.why == ASET_GROW) {
        m = aset_system_rows<N, DUAL>(lane, r.au, r.ad);
        if (m <= C4) r = aset_run_huge<N, C4, TAG, DUAL>(lane, T, Hm, Uu, vdu, up_d, lo_d, umax, dmax, dt, rec, r.au, r.ad, r.iters, r.flops, lds, cap_large, hist, g0);
    }
    return r;
}

// One attempt at a QP whose H belongs to this scenario and solve alone (no table): the free-input form about the point
// Uref (vdref = dt * prefix sum of it) where the objective's gradient is g0 -- the unconstrained minimiser with g0 = 0, or
// any other point -- started as prm.warm_mode says from the scenario's record prm.warm, with the back-off of the
// fixed-model kernels' constrained step kept in that record (ASET_BACKOFF_LOG2; off by default).  PREDICT: the record's
// history may be extrapolated (one QP per closed-loop step), else the record is only read as the previous solve's set.
// as_o: the scenario's counters (SolveArgs::as_out) or NULL.
struct AsetTry {
    double U;
    bool solved, tried;
};

template <int N, int TAG, bool PREDICT>
__device__ __forceinline__ AsetTry aset_try_general(const IpmParams &prm, int lane, const double *Hm, double Uref, double vdref, double g0,
                                                    double up_d, double lo_d, double *lds, int *as_o)
{
    AsetTry out = {0.0, false, false};
    aset_gdouble *bo = prm.warm ? (aset_gdouble *)prm.warm + 5 * WAVE + 1 : nullptr;
    if (!bo) return out;
    out.tried = true;
    const double bo0 = bo[0], bo1 = bo[1], bo2 = bo[2];                         // one round trip for all
    const int wait = __builtin_amdgcn_readfirstlane((int)bo0);
    if (wait > 0) {
        if (lane == 0) bo[0] = (double)(wait - 1);
        return out;
    }
    const int hist_raw = __builtin_amdgcn_readfirstlane((int)bo2);
    const int hist = PREDICT ? hist_raw : ((hist_raw & ~3) | min(hist_raw & 3, 1));
    const AsetResult r = active_set_solve<N, TAG, false, PREDICT>(lane, nullptr, Hm, Uref, vdref, up_d, lo_d, prm.umax, prm.dmax, prm.dt, prm.warm,
                                                         prm.warm_mode, lds, ASET_CAP_LARGE, hist, g0);
    if (as_o && lane == 0) {
        atomicAdd(&as_o[r.why == ASET_SOLVED ? 0 : r.why == ASET_GROW ? 2 : 3], 1);
        atomicAdd(&as_o[1], r.iters);
        atomicAdd(&as_o[4], (r.flops + 32) >> 6);
    }
    if (lane == 0) {
        const int streak = r.why == ASET_SOLVED ? 0 : min((int)bo1 + 1, 4);
        if ((double)streak != bo1) bo[1] = (double)streak;
        if (streak) bo[0] = (double)((1 << min(streak, ASET_BACKOFF_LOG2)) - 1);
        if (r.why != ASET_SOLVED && hist_raw != 0) bo[2] = 0.0;                 // the record will be another method's: no history
    }
    if (r.why == ASET_SOLVED) { out.U = r.U; out.solved = true; }
    return out;
}

}  // namespace mpc
