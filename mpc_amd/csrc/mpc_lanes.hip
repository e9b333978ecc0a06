// Lane-per-scenario MPC step for gfx950: a stage-wise (Riccati) primal-dual interior point.
//
// One LANE owns one scenario.  Where the wave-per-scenario kernels condense the horizon into a dense N x N problem and
// spend most of their instructions on cross-lane traffic (pivot broadcasts, scans), this kernel keeps the problem in its
// stage-wise form
//     min  sum_k 1/2 (x_k - r_k)' W_k (x_k - r_k) + 1/2 wr u_k^2,   x_{k+1} = A_k x_k + b u_k + c_k,
//          |u_k| <= umax (k < N),  lo_k <= delta_k <= hi_k (k = 1..N-1)
// and solves every Newton system of the interior point with a Riccati recursion over the stages: O(N) work per
// iteration instead of O(N^3), no condensing at all, no communication between lanes.  A wavefront therefore advances 64
// scenarios with the instruction count the other kernels spend on one.  The price is memory: the per-stage data of a
// scenario (inputs, states, slacks, multipliers, Riccati gains; LN_FIELDS doubles per stage) lives in a workspace in HBM,
// laid out [field][stage][scenario] so that the lanes of a wave touch consecutive addresses.
//
// Serves the linear-model controllers with a model that changes per stage and step (the LPV controller; any horizon).
//
// Reference behaviour restated (file:line of the reference checkout):
//   LPVMPC.py:110-171 stage model, :191-226 cost, :254-341 optimize()      -> stage_model() (mpc_wave.hpp), this kernel
//   LMPC2.py:196-210 bounds (stages 1..N-1 on delta, all stages on u)       -> the two constraint families below
//   skidpad_simulator.py:101-135, simulator.py:178-202 closed loop          -> MODE_SKIDPAD / MODE_STEP
#include <cstdlib>
#include "mpc_device.hpp"
#include "mpc_kernels.h"
#include "mpc_wave.hpp"
#include "mpc_ipm.hpp"

#pragma clang fp contract(fast)

// This file is compiled twice.  LANE_X0 = 1 builds the solver for problems that put no weight on the longitudinal
// position px (stage weight and terminal row/column zero: the controllers' parameter files): px drives nothing, so its
// adjoint and its row and column of the Riccati matrix are exactly zero and everything that would compute or move those
// zeros is left out.  LANE_X0 = 0 is the general solver (per-scenario cost tables, other weights).  Results are
// identical: the skipped operations only ever add exact zeros.
#ifndef LANE_X0
#define LANE_X0 0
#endif
#if LANE_X0
#define LANE_FN(name) name##_nopx
#else
#define LANE_FN(name) name
#endif

namespace mpc {
namespace LANE_FN(lanes) {
constexpr int X0 = LANE_X0;          // first state that takes part in the adjoint and matrix recursions

// workspace fields, each [N + 1][stride] doubles
enum {
    LF_U = 0, LF_UPREV,                                  // inputs, inputs of the previous iterate
    LF_X = 2,                                            // 6 fields: predicted states
    LF_TPU = 8, LF_TMU, LF_LPU, LF_LMU,                  // slacks / multipliers of  u <= umax,  -u <= umax
    LF_TPX, LF_TMX, LF_LPX, LF_LMX,                      // slacks / multipliers of  delta <= hi,  -delta <= -lo
    LF_DTPU = 16, LF_DTMU, LF_DLPU, LF_DLMU, LF_DTPX, LF_DTMX, LF_DLPX, LF_DLMX,   // their directions
    LF_ATPB = 24,                                        // 6 fields: A' P b of the Riccati recursion
    LF_KG = 30, LF_IRE, LF_DU,                           // feed-forward, 1 / (wr + Du + b'Pb), input direction
    LF_ST = 33,                                          // 10 fields: stage model
    LF_REF = 43,                                         // 3 fields: reference x, y, sin(psi)
    LN_FIELDS = 46
};

enum { TF_T = 0, TF_L, TF_DT, TF_DL };      // per terminal row: slack, multiplier, their directions

// Workspace layout: scenarios in blocks of LPW (the lanes of one wave that carry a scenario), and inside a block
// [stage][field][lane]: every access of a wave is one contiguous run of LPW doubles, and the fields of one stage sit at
// compile-time offsets from the stage's base address (immediate offsets, no address arithmetic per access).  After the
// stages of a block: [terminal row][TF_*][lane] when a terminal set is in use.
// The workspace is addressed through a buffer descriptor per wave (the wave's block of the workspace): an access is
// descriptor + scalar offset (the stage and a 4 KB window of its fields; the same for every lane, computed by the scalar
// unit once per stage) + the lane's own byte offset (one vector register for every access of the kernel) + a 12-bit
// immediate (the field inside the window).  With 64-bit flat addresses the compiler keeps a vector register pair per group
// of fields and spills them.
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int LPW>
struct LaneCtxT {
    __amdgpu_buffer_rsrc_t rsrc;      // wave-uniform
    mutable unsigned lane8;           // lane * 8
    unsigned tbase;                   // byte offset of the terminal-row block (after the N + 1 stages)
    static constexpr unsigned WINDOW = 4096;
    struct Cell {
        const LaneCtxT &c; unsigned soff, imm;
        __device__ __forceinline__ operator double() const { return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(c.rsrc, c.lane8 + imm, soff, 0)); }
        __device__ __forceinline__ const Cell &operator=(double v) const { __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, v), c.rsrc, c.lane8 + imm, soff, 0); return *this; }
        __device__ __forceinline__ const Cell &operator=(const Cell &o) const { return *this = (double)o; }
    };
    __device__ __forceinline__ Cell at(int field, int k) const
    {
        pin();
        const unsigned fo = (unsigned)field * (LPW * 8);
        return Cell{*this, (unsigned)k * (LN_FIELDS * LPW * 8) + fo / WINDOW * WINDOW, fo % WINDOW};
    }
    // slack, multiplier and their directions (f = TF_T..TF_DL) of terminal row i
    __device__ __forceinline__ Cell term(int f, int i) const
    {
        pin();
        return Cell{*this, tbase + (unsigned)(i * 4 + f) * (LPW * 8), 0u};
    }
    // The same workspace with this lane switched off unless `on`: its offset lies outside the descriptor's range, so the
    // hardware drops its stores and its loads return zero, with no change of EXEC.  The solver's control flow is uniform
    // per wave; lanes that are done, or take another path, run along switched off and their results are discarded by
    // selects.  (Divergent regions around whole sweeps made the compiler park live registers in AGPRs inside the region,
    // ahead of the EXEC restore of the join: lanes that skipped the region got stale values back.)
    __device__ __forceinline__ LaneCtxT masked(bool on) const
    {
        LaneCtxT r = *this;
        r.lane8 = on ? lane8 : 0x80000000u;
        return r;
    }
    // Makes the lane offset opaque to the optimiser at every access.  "lane offset + field constant" is loop invariant and
    // common to many accesses: hoisted or shared, it becomes one more vector register (and addition) per field instead of
    // the immediate of the memory instruction; opaque, the addition stays next to its access and folds.
    __device__ __forceinline__ void pin() const { asm volatile("" : "+v"(lane8)); }
};

// Reciprocal for the barrier terms: the hardware's approximation (4.6e-8 relative, measured) and one Newton step,
// 2e-15 relative (tools/rcp_accuracy.hip).  It only scales Newton directions, which the next iteration corrects; the
// residuals that decide convergence do not go through it.
__device__ __forceinline__ double lane_rcp(double x)
{
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}

__device__ __forceinline__ bool wave_any(bool x) { return __builtin_amdgcn_ballot_w64(x) != 0ull; }

struct StageM { double e02, e03, e12, e13, f33, f34, f43, f44, c0, c1; };
struct ScenM { double dt, f35, f45, c3; };     // parts of the model that do not change over the horizon

template <int LPW>
__device__ __forceinline__ StageM load_stage(const LaneCtxT<LPW> &c, int k)
{
    StageM s;
    s.e02 = c.at(LF_ST + 0, k); s.e03 = c.at(LF_ST + 1, k); s.e12 = c.at(LF_ST + 2, k); s.e13 = c.at(LF_ST + 3, k);
    s.f33 = c.at(LF_ST + 4, k); s.f34 = c.at(LF_ST + 5, k); s.f43 = c.at(LF_ST + 6, k); s.f44 = c.at(LF_ST + 7, k);
    s.c0 = c.at(LF_ST + 8, k); s.c1 = c.at(LF_ST + 9, k);
    return s;
}

// symmetric 6 x 6 matrix, upper triangle in 21 registers (indices fold at compile time in unrolled code)
struct Sym6 {
    double v[21];
    static __device__ __forceinline__ constexpr int idx(int i, int j) { return i <= j ? i * 6 - (i * (i - 1)) / 2 + (j - i) : j * 6 - (j * (j - 1)) / 2 + (i - j); }
    __device__ __forceinline__ double operator()(int i, int j) const { return v[idx(i, j)]; }
    __device__ __forceinline__ double &operator()(int i, int j) { return v[idx(i, j)]; }
};

// Riccati matrix step  P <- A'PA - a a' ire + diag(wq) + e5 e5' dx5  (a = A'Pb), column by column: column j of P A is
// formed, multiplied by A' and the upper part stored; P itself stays intact until every column is done.
__device__ __forceinline__ void riccati_matrix_step(Sym6 &P, const StageM &s, const ScenM &m, const double (&atpb)[6], double ire,
                                                    const double (&wq)[6], double dx5)
{
    Sym6 Q;
#pragma unroll
    for (int j = X0; j < 6; j++) {
        double col[6], o[6];
#pragma unroll
        for (int i = X0; i < 6; i++) {
            double t;
            if (j == 0) t = P(i, 0);
            else if (j == 1) t = P(i, 1);
            else if (j == 2) { t = s.e12 * P(i, 1) + P(i, 2); if (X0 == 0) t += s.e02 * P(i, 0); }
            else if (j == 3) { t = s.e13 * P(i, 1) + s.f33 * P(i, 3) + s.f43 * P(i, 4); if (X0 == 0) t += s.e03 * P(i, 0); }
            else if (j == 4) t = m.dt * P(i, 2) + s.f34 * P(i, 3) + s.f44 * P(i, 4);
            else t = m.f35 * P(i, 3) + m.f45 * P(i, 4) + P(i, 5);
            col[i] = t;
        }
        // rows X0..j of A' col
        if (X0 == 0) o[0] = col[0];
        o[1] = col[1];
        o[2] = s.e12 * col[1] + col[2];
        o[3] = s.e13 * col[1] + s.f33 * col[3] + s.f43 * col[4];
        if (X0 == 0) { o[2] += s.e02 * col[0]; o[3] += s.e03 * col[0]; }
        o[4] = m.dt * col[2] + s.f34 * col[3] + s.f44 * col[4];
        o[5] = m.f35 * col[3] + m.f45 * col[4] + col[5];
#pragma unroll
        for (int i = X0; i <= j; i++)
            Q(i, j) = o[i] - atpb[i] * atpb[j] * ire + ((i == j) ? wq[i] + ((i == 5) ? dx5 : 0.0) : 0.0);
    }
#pragma unroll
    for (int i = X0; i < 6; i++) {
#pragma unroll
        for (int j = i; j < 6; j++) P(i, j) = Q(i, j);
    }
}

// y = A x (the linear part of the stage map; b u and c are added by the caller)
__device__ __forceinline__ void mulA(const StageM &s, const ScenM &m, const double (&x)[6], double (&y)[6])
{
    y[0] = x[0] + s.e02 * x[2] + s.e03 * x[3];
    y[1] = x[1] + s.e12 * x[2] + s.e13 * x[3];
    y[2] = x[2] + m.dt * x[4];
    y[3] = s.f33 * x[3] + s.f34 * x[4] + m.f35 * x[5];
    y[4] = s.f43 * x[3] + s.f44 * x[4] + m.f45 * x[5];
    y[5] = x[5];
}
// y = A x for a direction of the states: component 0 is left out when px takes no part (nothing reads it)
__device__ __forceinline__ void mulA_dir(const StageM &s, const ScenM &m, const double (&x)[6], double (&y)[6])
{
    if (X0 == 0) y[0] = x[0] + s.e02 * x[2] + s.e03 * x[3];
    y[1] = x[1] + s.e12 * x[2] + s.e13 * x[3];
    y[2] = x[2] + m.dt * x[4];
    y[3] = s.f33 * x[3] + s.f34 * x[4] + m.f35 * x[5];
    y[4] = s.f43 * x[3] + s.f44 * x[4] + m.f45 * x[5];
    y[5] = x[5];
}
// y = A' v (components X0..5; v[0] is zero when px takes no part)
__device__ __forceinline__ void mulAt(const StageM &s, const ScenM &m, const double (&v)[6], double (&y)[6])
{
    if (X0 == 0) y[0] = v[0];
    y[1] = v[1];
    y[2] = s.e12 * v[1] + v[2];
    y[3] = s.e13 * v[1] + s.f33 * v[3] + s.f43 * v[4];
    if (X0 == 0) { y[2] += s.e02 * v[0]; y[3] += s.e03 * v[0]; }
    y[4] = m.dt * v[2] + s.f34 * v[3] + s.f44 * v[4];
    y[5] = m.f35 * v[3] + m.f45 * v[4] + v[5];
}

// Largest ratio -d / v over the slacks and multipliers v > 0 with directions d: the step to the boundary is its
// reciprocal when it exceeds one.  The reciprocals of the slacks are at hand (the directions need them); those of the
// multipliers come from the hardware's approximate reciprocal alone (about 1e-8 relative), which is enough here: the step
// is shortened by at least 1e-6 afterwards, so an iterate never reaches the boundary because of it.
struct Ratio {
    double rmax = 0.0;
    __device__ __forceinline__ void add(double d, double inv_v) { rmax = fmax(rmax, -d * inv_v); }
    __device__ __forceinline__ double step() const { return rmax > 1.0 ? lane_rcp(rmax) : 1.0; }
};

struct LaneProb {
    double wq[6], wr;
    const double *QN;              // terminal weight, 6 x 6 row-major (the controller's, or this scenario's row of the cost table)
    double umax, dmax, d0;         // d0: steering angle at the start of the horizon
    double xr_term_zero;           // 1: the terminal reference is zero (LPV), 0: the planner's
    int N;
    int mT;                        // rows of the terminal set  CT (x_N[1:] - x_ref) <= 1  (0: none)
    const double *CT;              // [mT][5], the same for every scenario (wave-uniform: scalar loads)
};

// x_N[1:] - x_ref of the terminal set, x_ref = (y_ref, sin psi_ref, 0, 0, 0) of the last stage (zero where the terminal
// cost reference is), from the states in the workspace or from `xN` when given
template <int LPW>
__device__ __forceinline__ void term_error(const LaneCtxT<LPW> &c, const LaneProb &p, const double *xN, double (&e)[5])
{
#pragma unroll
    for (int j = 0; j < 5; j++) e[j] = xN ? xN[1 + j] : (double)c.at(LF_X + 1 + j, p.N);
    if (p.xr_term_zero == 0.0) { e[0] -= c.at(LF_REF + 1, p.N); e[1] -= c.at(LF_REF + 2, p.N); }
}
__device__ __forceinline__ double term_dot(const double *cr, const double (&e)[5])
{
    return cr[0] * e[0] + cr[1] * e[1] + cr[2] * e[2] + cr[3] * e[3] + cr[4] * e[4];
}

__device__ __forceinline__ void delta_bounds(const LaneProb &p, double dt, int k, double &lo, double &hi)
{
    // |delta_k| <= dmax on stages 1..N-1, widened along the fastest way back inside when the start value lies outside
    // (see steer_bounds() in mpc_ipm.hpp)
    const double reach = 0.98 * p.umax * dt * (double)k;
    hi = fmax(p.dmax, p.d0 - reach);
    lo = fmin(-p.dmax, p.d0 + reach);
}

// Every sweep below walks the stages with the same shape: everything a stage needs is requested from memory first, then
// the stage is computed (sched_barrier keeps the compiler from sinking the loads back to their uses), so that a wave
// waits for memory once per stage, not once per value: left alone the scheduler issues the loads in pairs with a full
// wait behind each pair.
#define LANE_LOADS_DONE() __builtin_amdgcn_sched_barrier(0)

// Walks the stages 0..N-1 (FWD) or N-1..0 with the values of a stage requested R - 1 stages before they are used: a
// round trip to the workspace takes longer than the arithmetic of a light stage, so one stage of lookahead leaves the
// wave waiting most of the time.  `load(k)` returns what stage k needs, `body(in, k)` computes the stage.  The R buffers
// The R buffers
// are used in turn and the loop is unrolled R times, so every buffer has a fixed set of registers and the values just
// consumed are overwritten by the request for stage k +- (R - 1): no copies.
template <int R, bool FWD, class In, class Load, class Body>
__device__ __forceinline__ void lane_sweep(int N, Load load, Body body)
{
    In ring[R];
#pragma unroll
    for (int d = 0; d < R - 1; d++) ring[d] = load(FWD ? min(d, N - 1) : max(N - 1 - d, 0));
#pragma unroll 1
    for (int ib = 0; ib < N; ib += R) {
#pragma unroll
        for (int d = 0; d < R; d++) {
            const int i = ib + d;
            if (i < N) {
                const int k = FWD ? i : N - 1 - i;
                ring[(d + R - 1) % R] = load(FWD ? min(k + R - 1, N - 1) : max(k - (R - 1), 0));
                LANE_LOADS_DONE();
                body(ring[d], k);
                LANE_LOADS_DONE();
            }
        }
    }
}

#ifndef LN_RING
#define LN_RING 2
#endif
#ifndef LN_RING_HEAVY
#define LN_RING_HEAVY 2
#endif

struct Con4 { double tp, tm, lp, lm; };       // slacks and multipliers of one two-sided bound, or their directions

template <int LPW> __device__ __forceinline__ Con4 load4(const LaneCtxT<LPW> &c, int field, int k)
{
    Con4 r; r.tp = c.at(field, k); r.tm = c.at(field + 1, k); r.lp = c.at(field + 2, k); r.lm = c.at(field + 3, k);
    return r;
}
template <int LPW> __device__ __forceinline__ void store4(const LaneCtxT<LPW> &c, int field, int k, const Con4 &v)
{
    c.at(field, k) = v.tp; c.at(field + 1, k) = v.tm; c.at(field + 2, k) = v.lp; c.at(field + 3, k) = v.lm;
}
template <int LPW> __device__ __forceinline__ void load6(const LaneCtxT<LPW> &c, int field, int k, double (&v)[6])
{
#pragma unroll
    for (int i = X0; i < 6; i++) v[i] = c.at(field + i, k);      // component 0 is not needed when px takes no part
}

// Roll the states out for the inputs in the workspace: x_0 = xh, x_{k+1} = A_k x_k + b u_k + c_k, stored.  With
// `zero_u` the inputs are taken as zero (free response).
template <int LPW>
__device__ __forceinline__ void lane_rollout(const LaneCtxT<LPW> &c, const LaneProb &p, const ScenM &m, const double (&xh)[6], bool zero_u)
{
    double x[6];
#pragma unroll
    for (int i = 0; i < 6; i++) { x[i] = xh[i]; c.at(LF_X + i, 0) = xh[i]; }
    struct In { StageM s; double u; };
    lane_sweep<LN_RING, true, In>(p.N,
        [&](int k) { In v; v.s = load_stage(c, k); v.u = zero_u ? 0.0 : c.at(LF_U, k); return v; },
        [&](const In &v, int k) {
            double y[6];
            mulA(v.s, m, x, y);
            y[0] += v.s.c0; y[1] += v.s.c1; y[3] += m.c3; y[5] += m.dt * v.u;
#pragma unroll
            for (int i = 0; i < 6; i++) { x[i] = y[i]; c.at(LF_X + i, k + 1) = y[i]; }
        });
}

// gradient of the terminal cost at x_N
template <int LPW>
__device__ __forceinline__ void terminal_grad(const LaneCtxT<LPW> &c, const LaneProb &p, double (&g)[6])
{
    double e[6];
    load6(c, LF_X, p.N, e);
    if (p.xr_term_zero == 0.0) {
        if (X0 == 0) e[0] -= c.at(LF_REF, p.N);
        e[1] -= c.at(LF_REF + 1, p.N); e[2] -= c.at(LF_REF + 2, p.N);
    }
#pragma unroll
    for (int i = X0; i < 6; i++) {
        double t = 0.0;
#pragma unroll
        for (int j = X0; j < 6; j++) t = fma(p.QN[i * 6 + j], e[j], t);
        g[i] = t;
    }
}

// state-cost gradient wq .* (x_k - r_k) from loaded values
__device__ __forceinline__ void stage_grad(const LaneProb &p, const double (&x)[6], const double (&r)[3], double (&g)[6])
{
    if (X0 == 0) g[0] = p.wq[0] * (x[0] - r[0]);
    g[1] = p.wq[1] * (x[1] - r[1]); g[2] = p.wq[2] * (x[2] - r[2]);
    g[3] = p.wq[3] * x[3]; g[4] = p.wq[4] * x[4]; g[5] = p.wq[5] * x[5];
}

// Largest entry of the reduced gradient at u = 0 (states of the free response must be in the workspace).
template <int LPW>
__device__ __forceinline__ double lane_grad_scale(const LaneCtxT<LPW> &c, const LaneProb &p, const ScenM &m)
{
    double pi[6], g0 = 0.0;
    terminal_grad(c, p, pi);
    struct In { StageM s; double x[6], r[3]; };
    lane_sweep<LN_RING, false, In>(p.N,
        [&](int k) {
            In v; v.s = load_stage(c, k); load6(c, LF_X, k, v.x);
            if (X0 == 0) v.r[0] = c.at(LF_REF, k);
            v.r[1] = c.at(LF_REF + 1, k); v.r[2] = c.at(LF_REF + 2, k);
            return v;
        },
        [&](const In &v, int k) {
            g0 = fmax(g0, fabs(m.dt * pi[5]));
            double y[6], g[6];
            mulAt(v.s, m, pi, y);
            stage_grad(p, v.x, v.r, g);
#pragma unroll
            for (int i = X0; i < 6; i++) pi[i] = g[i] + y[i];
        });
    return g0;
}

// What one backward step reads from the workspace.
struct BackIn {
    StageM s;
    double u, x[6], r[3];
    Con4 cu, cx, du, dx;          // bounds on the input and on the steering angle of stage k; their (affine) directions
    double atpb[6], ire;
};
template <int LPW, int mode, bool factor>
__device__ __forceinline__ BackIn load_back(const LaneCtxT<LPW> &c, int k)
{
    BackIn b;
    b.s = load_stage(c, k);
    b.u = c.at(LF_U, k);
    if (mode != 3) { b.cu = load4(c, LF_TPU, k); b.cx = load4(c, LF_TPX, k); }
    load6(c, LF_X, k, b.x);
    if (X0 == 0) b.r[0] = c.at(LF_REF, k);
    b.r[1] = c.at(LF_REF + 1, k); b.r[2] = c.at(LF_REF + 2, k);
    if (mode == 1) { b.du = load4(c, LF_DTPU, k); b.dx = load4(c, LF_DTPX, k); }
    if (!factor) { load6(c, LF_ATPB, k, b.atpb); b.ire = c.at(LF_IRE, k); }
    return b;
}

// Backward sweep.  Computes the dual residual (largest entry of the reduced gradient of the Lagrangian) and, for the
// Newton system of the current iterate with complementarity targets given by `mode`, the Riccati gains:
//   mode 0  predictor        t * lambda -> 0
//   mode 1  corrector        t * lambda -> sigmu - dt_aff * dlambda_aff   (affine directions are in the direction fields)
//   mode 2  plain centring   t * lambda -> sigmu
//   mode 3  no bounds at all (multipliers taken as zero): the gains of the unconstrained minimiser
// With `factor` the matrix recursion runs (A'Pb, 1/Re stored); without it the stored factorisation is reused and only
// the vector recursion runs.
template <int LPW, int mode, bool factor>
__device__ __forceinline__ double lane_backward(const LaneCtxT<LPW> &c, const LaneProb &p, const ScenM &m, double sigmu)
{
    const int N = p.N;
    Sym6 P;
    double pv[6], pi[6], resd = 0.0;
    terminal_grad(c, p, pi);
#pragma unroll
    for (int i = X0; i < 6; i++) {
        pv[i] = pi[i];
#pragma unroll
        for (int j = i; j < 6; j++) P(i, j) = p.QN[i * 6 + j];
    }
    if (mode != 3 && p.mT > 0) {
        // terminal rows: multipliers into both adjoints, the Newton right-hand side and (factor) the barrier Hessian
        double e[5];
        term_error(c, p, nullptr, e);
#pragma unroll 1
        for (int i = 0; i < p.mT; i++) {
            const double *cr = p.CT + 5 * i;
            const double t = c.term(TF_T, i), l = c.term(TF_L, i);
            double rc = t * l;
            if (mode == 1) rc += (double)c.term(TF_DT, i) * (double)c.term(TF_DL, i) - sigmu; else if (mode == 2) rc -= sigmu;
            const double rp = term_dot(cr, e) + t - 1.0;
            const double it = lane_rcp(t);
            const double w = (l * rp - rc) * it, D = l * it;
#pragma unroll
            for (int a = 0; a < 5; a++) {
                pi[1 + a] += l * cr[a];
                pv[1 + a] += (l + w) * cr[a];
                if (factor) {
#pragma unroll
                    for (int b = a; b < 5; b++) P(1 + a, 1 + b) += D * cr[a] * cr[b];
                }
            }
        }
    }
    lane_sweep<factor ? LN_RING_HEAVY : LN_RING, false, BackIn>(N, [&](int k) { return load_back<LPW, mode, factor>(c, k); }, [&](const BackIn &b, int k) {
        // ---- input constraints of stage k
        double Du = 0.0, wu = 0.0, gu = (mode == 3) ? 0.0 : p.wr * b.u;      // mode 3 starts from u = 0
        if (mode != 3) {
            const double itp = lane_rcp(b.cu.tp), itm = lane_rcp(b.cu.tm);
            const double rpp = b.u + b.cu.tp - p.umax, rpm = -b.u + b.cu.tm - p.umax;
            double rcp = b.cu.tp * b.cu.lp, rcm = b.cu.tm * b.cu.lm;
            if (mode == 1) { rcp += b.du.tp * b.du.lp - sigmu; rcm += b.du.tm * b.du.lm - sigmu; }
            else if (mode == 2) { rcp -= sigmu; rcm -= sigmu; }
            wu = (b.cu.lp * rpp - rcp) * itp - (b.cu.lm * rpm - rcm) * itm;
            Du = b.cu.lp * itp + b.cu.lm * itm;
            gu += b.cu.lp - b.cu.lm;
        }
        resd = fmax(resd, fabs(gu + m.dt * pi[5]));
        double atpb[6], ire;
        if (factor) {
            double pb[6];
#pragma unroll
            for (int i = X0; i < 6; i++) pb[i] = m.dt * P(i, 5);
            ire = lane_rcp(p.wr + Du + m.dt * pb[5]);
            mulAt(b.s, m, pb, atpb);
#pragma unroll
            for (int i = X0; i < 6; i++) c.at(LF_ATPB + i, k) = atpb[i];
            c.at(LF_IRE, k) = ire;
        } else {
#pragma unroll
            for (int i = X0; i < 6; i++) atpb[i] = b.atpb[i];
            ire = b.ire;
        }
        const double kg = -(gu + wu + m.dt * pv[5]) * ire;
        c.at(LF_KG, k) = kg;
        if (k > 0) {
            // ---- state cost and steering-angle constraints of stage k
            double gx[6], Dx = 0.0, wx = 0.0;
            stage_grad(p, b.x, b.r, gx);
            if (mode != 3) {
                const double itp = lane_rcp(b.cx.tp), itm = lane_rcp(b.cx.tm);
                double lo, hi;
                delta_bounds(p, m.dt, k, lo, hi);
                const double rpp = b.x[5] + b.cx.tp - hi, rpm = -b.x[5] + b.cx.tm + lo;
                double rcp = b.cx.tp * b.cx.lp, rcm = b.cx.tm * b.cx.lm;
                if (mode == 1) { rcp += b.dx.tp * b.dx.lp - sigmu; rcm += b.dx.tm * b.dx.lm - sigmu; }
                else if (mode == 2) { rcp -= sigmu; rcm -= sigmu; }
                wx = (b.cx.lp * rpp - rcp) * itp - (b.cx.lm * rpm - rcm) * itm;
                Dx = b.cx.lp * itp + b.cx.lm * itm;
                gx[5] += b.cx.lp - b.cx.lm;
            }
            // vector recursions: p <- (gx + e5 wx) + A'p + A'Pb kg,  pi <- gx + A'pi
            double y[6];
            mulAt(b.s, m, pv, y);
#pragma unroll
            for (int i = X0; i < 6; i++) pv[i] = gx[i] + y[i] + atpb[i] * kg;
            pv[5] += wx;
            mulAt(b.s, m, pi, y);
#pragma unroll
            for (int i = X0; i < 6; i++) pi[i] = gx[i] + y[i];
            if (factor) riccati_matrix_step(P, b.s, m, atpb, ire, p.wq, Dx);
        }
    });
    return resd;
}

// What one forward step reads: the gains of stage k, the input block of stage k and the steering block of stage k + 1.
struct FwdIn {
    StageM s;
    double atpb[6], kg, ire, u, v;
    Con4 cu, cx, du, dx;
};
template <int LPW, int mode>
__device__ __forceinline__ FwdIn load_fwd(const LaneCtxT<LPW> &c, int k, int N)
{
    const int q = min(k + 1, N - 1);
    FwdIn f;
    f.s = load_stage(c, k);
    load6(c, LF_ATPB, k, f.atpb);
    f.kg = c.at(LF_KG, k); f.ire = c.at(LF_IRE, k);
    if (mode != 3) {
        f.u = c.at(LF_U, k);
        f.cu = load4(c, LF_TPU, k);
        f.v = c.at(LF_X + 5, q);
        f.cx = load4(c, LF_TPX, q);
        if (mode == 1) { f.du = load4(c, LF_DTPU, k); f.dx = load4(c, LF_DTPX, q); }
    }
    return f;
}

// one two-sided bound: directions of slacks and multipliers for a move `dv` of the bounded quantity
struct ProdSums { double s0 = 0.0, s1 = 0.0, s2 = 0.0; };      // sum of (t + a dt)(l + a dl) = s0 + a s1 + a^2 s2, for any a
__device__ __forceinline__ Con4 bound_dirs(const Con4 &t, const Con4 &aff, double rpp, double rpm, double dv, int mode, double sigmu,
                                           Ratio &r, ProdSums &ps)
{
    double rcp = t.tp * t.lp, rcm = t.tm * t.lm;
    if (mode == 1) { rcp += aff.tp * aff.lp - sigmu; rcm += aff.tm * aff.lm - sigmu; }
    else if (mode == 2) { rcp -= sigmu; rcm -= sigmu; }
    Con4 d;
    d.tp = -rpp - dv; d.tm = -rpm + dv;
    const double itp = lane_rcp(t.tp), itm = lane_rcp(t.tm);
    d.lp = -(rcp + t.lp * d.tp) * itp; d.lm = -(rcm + t.lm * d.tm) * itm;
    r.add(d.tp, itp); r.add(d.tm, itm); r.add(d.lp, __builtin_amdgcn_rcp(t.lp)); r.add(d.lm, __builtin_amdgcn_rcp(t.lm));
    ps.s0 += t.tp * t.lp + t.tm * t.lm;
    ps.s1 += t.tp * d.lp + t.lp * d.tp + t.tm * d.lm + t.lm * d.tm;
    ps.s2 += d.tp * d.lp + d.tm * d.lm;
    return d;
}

// Forward sweep: Newton direction from the stored gains, directions of slacks and multipliers (stored), the largest step
// that keeps them positive, and the three sums from which the complementarity after a step of any length follows.  `mode` as in lane_backward (in mode 1 the affine directions are read before they are
// overwritten); mode 3 writes the inputs of the unconstrained minimiser to LF_U and nothing else.
template <int LPW, int mode>
__device__ __forceinline__ double lane_forward(const LaneCtxT<LPW> &c, const LaneProb &p, const ScenM &m, double sigmu, ProdSums &ps)
{
    const int N = p.N;
    double dx[6] = {0, 0, 0, 0, 0, 0};
    Ratio r;
    lane_sweep<LN_RING, true, FwdIn>(N, [&](int k) { return load_fwd<LPW, mode>(c, k, N); }, [&](const FwdIn &f, int k) {
        const int q = k + 1;
        double acc = 0.0;
#pragma unroll
        for (int i = X0; i < 6; i++) acc = fma(f.atpb[i], dx[i], acc);
        const double du = f.kg - f.ire * acc;
        double y[6];
        mulA_dir(f.s, m, dx, y);
        y[5] += m.dt * du;
#pragma unroll
        for (int i = X0; i < 6; i++) dx[i] = y[i];
        if (mode == 3)
            c.at(LF_U, k) = du;
        else {
            c.at(LF_DU, k) = du;
            const Con4 d = bound_dirs(f.cu, f.du, f.u + f.cu.tp - p.umax, -f.u + f.cu.tm - p.umax, du, mode, sigmu, r, ps);
            store4(c, LF_DTPU, k, d);
            if (q < N) {
                double lo, hi;
                delta_bounds(p, m.dt, q, lo, hi);
                const Con4 e = bound_dirs(f.cx, f.dx, f.v + f.cx.tp - hi, -f.v + f.cx.tm + lo, dx[5], mode, sigmu, r, ps);
                store4(c, LF_DTPX, q, e);
            }
        }
    });
    if (mode != 3 && p.mT > 0) {
        double e[5];
        term_error(c, p, nullptr, e);
        const double dxe[5] = {dx[1], dx[2], dx[3], dx[4], dx[5]};
#pragma unroll 1
        for (int i = 0; i < p.mT; i++) {
            const double *cr = p.CT + 5 * i;
            const double t = c.term(TF_T, i), l = c.term(TF_L, i);
            double rc = t * l;
            if (mode == 1) rc += (double)c.term(TF_DT, i) * (double)c.term(TF_DL, i) - sigmu; else if (mode == 2) rc -= sigmu;
            const double d_t = -(term_dot(cr, e) + t - 1.0) - term_dot(cr, dxe);
            const double it = lane_rcp(t);
            const double d_l = -(rc + l * d_t) * it;
            r.add(d_t, it); r.add(d_l, __builtin_amdgcn_rcp(l));
            ps.s0 += t * l; ps.s1 += t * d_l + l * d_t; ps.s2 += d_t * d_l;
            c.term(TF_DT, i) = d_t; c.term(TF_DL, i) = d_l;
        }
    }
    return r.step();
}

// Sum and smallest of the slack-multiplier products after a step of length alpha along the stored directions.
template <int LPW>
__device__ __forceinline__ void lane_products(const LaneCtxT<LPW> &c, const LaneProb &p, double alpha, double &sum, double &mn)
{
    sum = 0.0; mn = 1e300;
    struct In { Con4 tu, du, tx, dx; };
    lane_sweep<LN_RING, true, In>(p.N,
        [&](int k) {
            const int q = min(k + 1, p.N - 1);
            In v; v.tu = load4(c, LF_TPU, k); v.du = load4(c, LF_DTPU, k); v.tx = load4(c, LF_TPX, q); v.dx = load4(c, LF_DTPX, q);
            return v;
        },
        [&](const In &v, int k) {
        const Con4 &tu = v.tu, &du = v.du, &tx = v.tx, &dx = v.dx;
        const double p1 = (tu.tp + alpha * du.tp) * (tu.lp + alpha * du.lp), p2 = (tu.tm + alpha * du.tm) * (tu.lm + alpha * du.lm);
        sum += p1 + p2; mn = fmin(mn, fmin(p1, p2));
        if (k + 1 < p.N) {
            const double p3 = (tx.tp + alpha * dx.tp) * (tx.lp + alpha * dx.lp), p4 = (tx.tm + alpha * dx.tm) * (tx.lm + alpha * dx.lm);
            sum += p3 + p4; mn = fmin(mn, fmin(p3, p4));
        }
    });
#pragma unroll 1
    for (int i = 0; i < p.mT; i++) {
        const double pr = ((double)c.term(TF_T, i) + alpha * (double)c.term(TF_DT, i)) * ((double)c.term(TF_L, i) + alpha * (double)c.term(TF_DL, i));
        sum += pr; mn = fmin(mn, pr);
    }
}

// Takes the step of length alpha along the stored directions (a fresh iterate has zero directions), rolls the states out for the new inputs
// and returns the complementarity (mean product) and the primal residual of the new iterate.  The inputs before the
// step are kept in LF_UPREV.
struct StepIn { StageM s; double u, du; Con4 cu, cx, dcu, dcx; };
template <int LPW>
__device__ __forceinline__ StepIn load_step(const LaneCtxT<LPW> &c, int k, int N)
{
    constexpr bool step = true;
    const int q = min(k + 1, N - 1);
    StepIn v;
    v.s = load_stage(c, k);
    v.u = c.at(LF_U, k);
    v.cu = load4(c, LF_TPU, k); v.cx = load4(c, LF_TPX, q);
    if (step) { v.du = c.at(LF_DU, k); v.dcu = load4(c, LF_DTPU, k); v.dcx = load4(c, LF_DTPX, q); }
    return v;
}
template <int LPW>
__device__ __forceinline__ void lane_step_rollout(const LaneCtxT<LPW> &c, const LaneProb &p, const ScenM &m, const double (&xh)[6],
                                                  double alpha, double &mu, double &resp)
{
    const int N = p.N;
    constexpr bool step = true;
    double x[6], musum = 0.0;
    resp = 0.0;
#pragma unroll
    for (int i = 0; i < 6; i++) x[i] = xh[i];
    lane_sweep<LN_RING, true, StepIn>(N, [&](int k) { return load_step(c, k, N); }, [&](const StepIn &v, int k) {
        const int q = k + 1;
        double u = v.u;
        Con4 cu = v.cu, cx = v.cx;
        if (step) {
            c.at(LF_UPREV, k) = u;
            u = fma(alpha, v.du, u);
            cu.tp = fma(alpha, v.dcu.tp, cu.tp); cu.tm = fma(alpha, v.dcu.tm, cu.tm);
            cu.lp = fma(alpha, v.dcu.lp, cu.lp); cu.lm = fma(alpha, v.dcu.lm, cu.lm);
            c.at(LF_U, k) = u;
            store4(c, LF_TPU, k, cu);
        }
        musum += cu.tp * cu.lp + cu.tm * cu.lm;
        resp = fmax(resp, fmax(fabs(u + cu.tp - p.umax), fabs(-u + cu.tm - p.umax)));
        double y[6];
        mulA(v.s, m, x, y);
        y[0] += v.s.c0; y[1] += v.s.c1; y[3] += m.c3; y[5] += m.dt * u;
#pragma unroll
        for (int i = 0; i < 6; i++) { x[i] = y[i]; c.at(LF_X + i, q) = y[i]; }
        if (q < N) {
            if (step) {
                cx.tp = fma(alpha, v.dcx.tp, cx.tp); cx.tm = fma(alpha, v.dcx.tm, cx.tm);
                cx.lp = fma(alpha, v.dcx.lp, cx.lp); cx.lm = fma(alpha, v.dcx.lm, cx.lm);
                store4(c, LF_TPX, q, cx);
            }
            double lo, hi;
            delta_bounds(p, m.dt, q, lo, hi);
            musum += cx.tp * cx.lp + cx.tm * cx.lm;
            resp = fmax(resp, fmax(fabs(x[5] + cx.tp - hi), fabs(-x[5] + cx.tm + lo)));
        }
    });
    if (p.mT > 0) {
        double e[5];
        term_error(c, p, x, e);
#pragma unroll 1
        for (int i = 0; i < p.mT; i++) {
            double t = c.term(TF_T, i), l = c.term(TF_L, i);
            if (step) {
                t = fma(alpha, (double)c.term(TF_DT, i), t); l = fma(alpha, (double)c.term(TF_DL, i), l);
                c.term(TF_T, i) = t; c.term(TF_L, i) = l;
            }
            musum += t * l;
            resp = fmax(resp, fabs(term_dot(p.CT + 5 * i, e) + t - 1.0));
        }
    }
    mu = musum / (2.0 * (2 * N - 1) + p.mT);
}

// true when the inputs in the workspace and their predicted steering angles satisfy every bound
template <int LPW>
__device__ __forceinline__ bool lane_feasible(const LaneCtxT<LPW> &c, const LaneProb &p, double dt)
{
    bool ok = true;
    struct In { double u, v; };
    lane_sweep<LN_RING, true, In>(p.N, [&](int k) { In v; v.u = c.at(LF_U, k); v.v = c.at(LF_X + 5, k); return v; },
        [&](const In &v, int k) {
            ok = ok && fabs(v.u) <= p.umax;
            if (k >= 1) {
                double lo, hi;
                delta_bounds(p, dt, k, lo, hi);
                ok = ok && v.v <= hi && v.v >= lo;
            }
        });
    if (p.mT > 0) {
        double e[5];
        term_error(c, p, nullptr, e);
#pragma unroll 1
        for (int i = 0; i < p.mT; i++) ok = ok && term_dot(p.CT + 5 * i, e) <= 1.0;
    }
    return ok;
}

struct LaneResult { int iters, status, constrained; double mu, res; bool warm_valid; };

#ifdef MPC_LANE_TIMING
// development build: shader-clock cycles per part of the solve, printed by the first wave
__device__ long long g_lane_t[8];
#define LT_BEGIN() long long lt0_ = clock64()
#define LT(i) do { const long long lt1_ = clock64(); if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)&g_lane_t[i], (unsigned long long)(lt1_ - lt0_)); lt0_ = clock64(); } while (0)
#else
#define LT_BEGIN() do { } while (0)
#define LT(i) do { } while (0)
#endif

constexpr int LN_SAFE_AFTER = 25;

// Starting point of the interior point on the lanes of `c` that are switched on: with `warm` (per lane) the inputs and
// multipliers that the previous closed-loop step left in the workspace, moved one stage ahead, slacks pushed back to a
// small distance from the bounds (as interior_point() does); otherwise u = 0, slacks at the distance of the bounds,
// multipliers lam0.  The directions are zeroed, so that the first pass of the step sweep leaves the iterate as it is.
template <int LPW>
__device__ __forceinline__ void lane_init_iterate(const LaneCtxT<LPW> &c, const LaneProb &p, const ScenM &m, bool warm, double lam0, double gscale)
{
    const int N = p.N;
    const double delta = 3e-3, mu0 = delta * 0.01 * gscale;
    double dl = p.d0;
#pragma unroll 1
    for (int k = 0; k < N; k++) {
        const int kn = (k < N - 1) ? k + 1 : k;                              // the record one stage ahead
        const double uw = fmin(fmax((double)c.at(LF_U, kn), -p.umax), p.umax);
        const double a0 = c.at(LF_LPU, kn), a1 = c.at(LF_LMU, kn), a2 = c.at(LF_LPX, kn), a3 = c.at(LF_LMX, kn);
        LANE_LOADS_DONE();
        const double u = warm ? uw : 0.0;
        const double tpu = warm ? fmax(p.umax - u, delta * 2.0 * p.umax) : p.umax, tmu = warm ? fmax(p.umax + u, delta * 2.0 * p.umax) : p.umax;
        double lo, hi;
        delta_bounds(p, m.dt, k, lo, hi);
        const double tpx = warm ? fmax(hi - dl, delta * 2.0 * p.dmax) : fmax(hi - p.d0, 0.1 * p.dmax);
        const double tmx = warm ? fmax(dl - lo, delta * 2.0 * p.dmax) : fmax(p.d0 - lo, 0.1 * p.dmax);
        c.at(LF_U, k) = u;
        c.at(LF_TPU, k) = tpu; c.at(LF_TMU, k) = tmu;
        c.at(LF_LPU, k) = warm ? fmax(a0, mu0 / tpu) : lam0; c.at(LF_LMU, k) = warm ? fmax(a1, mu0 / tmu) : lam0;
        c.at(LF_TPX, k) = tpx; c.at(LF_TMX, k) = tmx;
        c.at(LF_LPX, k) = warm ? fmax(a2, mu0 / tpx) : lam0; c.at(LF_LMX, k) = warm ? fmax(a3, mu0 / tmx) : lam0;
        c.at(LF_DU, k) = 0.0;
        store4(c, LF_DTPU, k, Con4{0.0, 0.0, 0.0, 0.0});
        store4(c, LF_DTPX, k, Con4{0.0, 0.0, 0.0, 0.0});
        dl = fma(m.dt, u, dl);                                               // steering angle of stage k + 1
        LANE_LOADS_DONE();
    }
    if (p.mT > 0) {
        double e[5];
        term_error(c, p, nullptr, e);           // of whatever roll-out the workspace holds: only a starting size
#pragma unroll 1
        for (int i = 0; i < p.mT; i++) {        // terminal rows stay with the last stage: nothing to move
            const double tw = fmax((double)c.term(TF_T, i), delta * 2.0), lw = c.term(TF_L, i);
            const double t = warm ? tw : fmax(1.0 - term_dot(p.CT + 5 * i, e), 0.1);
            c.term(TF_T, i) = t; c.term(TF_L, i) = warm ? fmax(lw, mu0 / t) : lam0;
            c.term(TF_DT, i) = 0.0; c.term(TF_DL, i) = 0.0;
        }
    }
}

// One solve.  Stage model and references of the step must be in the workspace; `warm` says that the workspace still
// holds inputs and multipliers of a constrained solve of the previous closed-loop step.
// Control flow is uniform over the wave: every branch that contains a sweep is taken when any lane needs it, with the
// other lanes switched off through their workspace context, and the per-lane decisions are selects.
template <int LPW>
__device__ __forceinline__ LaneResult lane_solve(const LaneCtxT<LPW> &c, const LaneProb &p, const ScenM &m, const double (&xh)[6],
                                                 bool warm, double tol_mu, double tol_res, int max_iter)
{
    const int N = p.N;
    LaneResult out = {0, 0, 0, 0.0, 0.0, false};
    LT_BEGIN();
    // free response and the size of the gradient at u = 0 (scale of the dual residual and of the first multipliers)
    lane_rollout(c, p, m, xh, true);
    const double gscale = fmax(1.0, lane_grad_scale(c, p, m));
    bool active = true;                     // the lane still iterates
    if (wave_any(!warm)) {
        // unconstrained minimiser first: it is the minimiser whenever it satisfies every bound.  (After a constrained
        // step the next one is almost always constrained too; the interior point then starts right away.)
        const LaneCtxT<LPW> cc = c.masked(!warm);
        (void)lane_backward<LPW, 3, true>(cc, p, m, 0.0);
        ProdSums unused;
        (void)lane_forward<LPW, 3>(cc, p, m, 0.0, unused);
        lane_rollout(cc, p, m, xh, false);
        const bool feasible = lane_feasible(cc, p, m.dt);
        active = warm || !feasible;
    }
    out.constrained = active ? 1 : 0;
    const double lam0 = fmax(1.0, 0.01 * gscale);
    if (wave_any(active)) lane_init_iterate(c.masked(active), p, m, warm, lam0, gscale);
    LT(0);

    const double nc2 = 2.0 * (2 * N - 1) + p.mT;
    double alpha = 0.0, mu = 0.0, res = 0.0, res_prev = 1e300, mu_last = 0.0, res_last = 0.0;
    bool acc_last = false, rec_ok = true;
    int iters = 0, status = 0;
    while (wave_any(active)) {
        const LaneCtxT<LPW> ca = c.masked(active);
        double mu_n, resp;
        lane_step_rollout(ca, p, m, xh, alpha, mu_n, resp);
        LT(1);
        const bool safe = iters >= LN_SAFE_AFTER;
        const double sigmu_s = 0.25 * mu_n;
        double resd = 0.0;
        if (wave_any(active && !safe)) resd = lane_backward<LPW, 0, true>(c.masked(active && !safe), p, m, 0.0);
        if (wave_any(active && safe)) {
            const double r2 = lane_backward<LPW, 2, true>(c.masked(active && safe), p, m, sigmu_s);
            resd = safe ? r2 : resd;
        }
        LT(2);
        const double res_n = fmax(resd / gscale, resp);
        mu = active ? mu_n : mu; res = active ? res_n : res;
        // exit rules: see interior_point() in mpc_ipm.hpp
        const bool reject = active && warm && iters == 0 && res > IPM_WARM_RES_MAX;      // the problem moved too far from the previous one
        const bool live = active && !reject;
        const bool finite = fabs(mu) < 1e300 && fabs(res) < 1e300;
        const bool acc = finite && mu <= 10.0 * tol_mu && res <= 1e3 * tol_res;
        const bool worse = live && acc_last && !acc;                                      // back to the last acceptable iterate
        const bool rest = live && !worse;
        const bool done = rest && ((finite && mu <= tol_mu && res <= tol_res) || (acc && (res >= 0.5 * res_prev || mu <= 1e-3 * tol_mu)));
        const bool stuck = rest && !done && (!finite || mu <= 1e-3 * tol_mu || iters >= max_iter);
        const bool done2 = stuck && acc;
        const bool restart = stuck && !acc && warm && iters < max_iter;                   // once more from the cold start
        const bool failed = stuck && !acc && !restart;
        const bool restore = worse || (failed && !finite && iters > 0);
        rec_ok = rec_ok && !worse && !failed;
        status = failed ? 2 : status;
        if (wave_any(restore)) {
            const LaneCtxT<LPW> cr = c.masked(restore);
#pragma unroll 1
            for (int k = 0; k < N; k++) cr.at(LF_U, k) = cr.at(LF_UPREV, k);
            lane_rollout(cr, p, m, xh, false);
        }
        mu = restore ? mu_last : mu; res = restore ? res_last : res;
        const bool again = reject || restart;                                             // cold start, no iteration counted for a rejected warm start
        if (wave_any(again)) lane_init_iterate(c.masked(again), p, m, false, lam0, gscale);
        warm = warm && !again;
        iters += restart ? 1 : 0;
        res_prev = restart ? 1e300 : res_prev; acc_last = restart ? false : acc_last;
        active = active && !(worse || done || done2 || failed);
        const bool go = active && !again;                                                 // lanes that compute a step in this pass
        mu_last = go ? mu : mu_last; res_last = go ? res : res_last; acc_last = go ? acc : acc_last; res_prev = go ? res : res_prev;
        double alpha_n = 0.0;
        if (wave_any(go && !safe)) {
            const LaneCtxT<LPW> cn = c.masked(go && !safe);
            ProdSums ps, ps2;
            const double a_aff = lane_forward<LPW, 0>(cn, p, m, 0.0, ps);
            LT(3);
            const double sum = fmax(ps.s0 + a_aff * (ps.s1 + a_aff * ps.s2), 0.0);       // complementarity after the affine step
            const double sg = (sum / nc2) / mu;
            const double sigmu = fmax(sg * sg * sg * mu, 1e-3 * fmin(mu, res));           // the gap must not outrun the residuals
            (void)lane_backward<LPW, 1, false>(cn, p, m, sigmu);
            LT(4);
            alpha_n = lane_forward<LPW, 1>(cn, p, m, sigmu, ps2);
            LT(5);
            alpha_n *= fmin(fmax(0.995, 1.0 - mu), 0.999999);                             // never onto the boundary itself
        }
        if (wave_any(go && safe)) {
            // long-step path following with a neighbourhood of the central path (see interior_point())
            const bool on = go && safe;
            const LaneCtxT<LPW> cs = c.masked(on);
            ProdSums ps;
            double a_s = lane_forward<LPW, 2>(cs, p, m, sigmu_s, ps);
            double sum, mn;
            lane_products(cs, p, 0.0, sum, mn);
            const double gam = fmin(0.01, 0.5 * mn / mu);
            a_s *= 0.99;
            bool cut = on;
            for (int bt = 0; bt < 20 && wave_any(cut); bt++) {
                lane_products(c.masked(cut), p, a_s, sum, mn);
                cut = cut && !(mn >= gam * sum / nc2);
                a_s = cut ? 0.6 * a_s : a_s;
            }
            alpha_n = safe ? a_s : alpha_n;
        }
        alpha = go ? alpha_n : 0.0;
        iters += go ? 1 : 0;
    }
    out.iters = iters; out.status = status; out.mu = mu; out.res = res; out.warm_valid = rec_ok && status == 0;
    return out;
}

// ------------------------------------------------------------------------------------------------
// LPW lanes of every wave carry a scenario (the other lanes idle); WPB waves per workgroup (1 in every launch shape in
// use: workgroups of one wave spread over the compute units first).
template <int LPW, int WPB>
__global__ __launch_bounds__(WPB * WAVE) void k_mpc_lanes(const DevCfg cfg, const SolveArgs a, double *ws)
{
    const int lane = threadIdx.x & (WAVE - 1);
    const int blk = __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + (threadIdx.x >> 6));   // block of LPW scenarios this wave advances
    const int s = blk * LPW + lane;
    if (lane >= LPW || s >= a.B) return;
    const int N = cfg.N;
    const size_t stage_doubles = (size_t)(N + 1) * LN_FIELDS * LPW;
    const size_t blk_doubles = stage_doubles + (size_t)4 * a.term_m * LPW;      // < 2^28: checked by the launcher
    LaneCtxT<LPW> c = {__builtin_amdgcn_make_buffer_rsrc(ws + (size_t)blk * blk_doubles, 0, (int)(blk_doubles * 8), 0x00020000), (unsigned)lane * 8u,
                       (unsigned)(stage_doubles * 8)};
    ModelParams mp;
    if (a.mp_scen) {
        const double *q = a.mp_scen + 6 * (size_t)s;
        mp = {q[0], q[1], q[2], q[3], q[4], q[5]};
    } else
        mp = {cfg.mp[0], cfg.mp[1], cfg.mp[2], cfg.mp[3], cfg.mp[4], cfg.mp[5]};
    const double dt = cfg.dt;
    LaneProb p;
    p.N = N; p.umax = cfg.max_rate; p.dmax = cfg.max_steering;
    p.xr_term_zero = (cfg.kind == MPC_KIND_LPV) ? 1.0 : 0.0;     // LPVMPC.py:220-221: the terminal reference stays zero
    p.mT = a.term_m; p.CT = a.term_C;
#pragma unroll
    for (int i = 0; i < 6; i++) p.wq[i] = dt * (a.cost_scen ? a.cost_scen[43 * (size_t)s + i] : cfg.w[i]);   // acados scales stage costs by dt
    p.wr = dt * (a.cost_scen ? a.cost_scen[43 * (size_t)s + 6] : cfg.w[6]);
    p.QN = a.cost_scen ? a.cost_scen + 43 * (size_t)s + 7 : cfg.We;
    ScenM m;
    m.dt = dt; m.f35 = dt * (-mp.Cf / mp.m); m.f45 = dt * (-mp.lf * mp.Cf / mp.Iz); m.c3 = 0.0;

    double plant[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    double prevp = 0.0, tvel = 0.0;
    int lapv = 0;
    if (a.mode != MODE_SOLVE) {
#pragma unroll
        for (int i = 0; i < 8; i++) plant[i] = a.state[(size_t)8 * s + i];
        tvel = a.target_vel[s];
        if (a.mode == MODE_SKIDPAD) { lapv = a.lap[s]; prevp = a.prev_progress[s]; }
    }
    const PlantCoef pc = plant_coef(mp, plant[4]);
    double *warm_flag = a.ipm_ws ? a.ipm_ws + s : nullptr;            // one flag per scenario: the record is in the workspace
    bool warm_ok = warm_flag && a.mode != MODE_SOLVE && *warm_flag == 1.0;
    const int nsteps = (a.mode == MODE_SOLVE) ? 1 : a.nsteps;
    int iters_tot = 0, status_max = 0, nconstr = 0, lap_flag = 0;
    double mu = 0.0, res = 0.0, u0 = 0.0;
    const double *lx = a.lin_x ? a.lin_x + (size_t)s * 6 * (N + 1) : nullptr;
    // closed-loop statistics: kept in the scenario's record in memory and updated once per step (eight more values per
    // lane would not fit the registers of this kernel)
    const bool cl_on = a.stat_out != nullptr && a.mode != MODE_SOLVE;

    for (int t = 0; t < nsteps; t++) {
        // ---- 1. start state, references, stage model
        double xh[6] = {0, 0, 0, 0, 0, 0};
        double dhat = 0.0;
        double r0x = 0.0, r0y = 0.0, r0s = 0.0;       // first reference point (closed-loop statistics)
        if (a.mode == MODE_SOLVE) {
            const double *x0 = a.x0 + 7 * (size_t)s;
            xh[3] = x0[4]; xh[4] = x0[5]; xh[5] = x0[6];
            if (a.d_hat) dhat = a.d_hat[s];
            for (int k = 0; k <= N; k++) {
                const double *w = a.wp + ((size_t)s * (N + 1) + k) * 4;
                c.at(LF_REF, k) = w[0]; c.at(LF_REF + 1, k) = w[1]; c.at(LF_REF + 2, k) = w[3];      // LMPC2.py:290-294
                if (k < N) {
                    double st[STAGE_W];
                    stage_model(cfg, mp, a.p[(size_t)s * (N + 1) + k], lx, N, k, st);
#pragma unroll
                    for (int q = 0; q < STAGE_W; q++) c.at(LF_ST + q, k) = st[q];
                }
            }
        } else {
            const double px = plant[0], py = plant[1];
            xh[3] = plant[5]; xh[4] = plant[6]; xh[5] = plant[7];
            double p0;
            if (a.mode == MODE_SKIDPAD) {
                int lp = lapv;
                p0 = sk_pos2progress(px, py, lp);
                if (prevp > p0 + 2 * SK_R) { lp += 1; p0 = sk_pos2progress(px, py, lp); }
                if (lp < 0 || lp > 7) lap_flag = 5;      // the reference switches to its speed profile there (path_planning.py:241-248)
                prevp = p0;                                                // path_planning.py:251
            } else
                p0 = px;                                                   // step_planning.py:83-86
            double ch, sh;
            unit_heading(plant[2], plant[3], ch, sh);
            double cs = 0.0;                                               // np.cumsum of the speeds, sequentially
            for (int k = 0; k <= N; k++) {
                double o[4];
                const double q = (k == 0) ? p0 : p0 + cs * dt;            // path_planning.py:236-240
                if (a.mode == MODE_SKIDPAD) sk_pose(q, o);
                else step_pose(q, a.step_ramp, o);
                to_car_frame(o, px, py, ch, sh);
                c.at(LF_REF, k) = o[0]; c.at(LF_REF + 1, k) = o[1]; c.at(LF_REF + 2, k) = o[3];
                if (k == 0) { r0x = o[0]; r0y = o[1]; r0s = o[3]; }
                if (a.wp_out) {
                    double *w = a.wp_out + ((size_t)s * (N + 1) + k) * 4;
                    w[0] = o[0]; w[1] = o[1]; w[2] = o[2]; w[3] = o[3];
                }
                if (k < N) {
                    double st[STAGE_W];
                    stage_model(cfg, mp, tvel, lx, N, k, st);
#pragma unroll
                    for (int qq = 0; qq < STAGE_W; qq++) c.at(LF_ST + qq, k) = st[qq];
                }
                cs = (k == 0) ? tvel : cs + tvel;
            }
        }
        m.c3 = (cfg.kind == MPC_KIND_OFL) ? dt * dhat : 0.0;              // OFLMPC2.py:145
        p.d0 = xh[5];
        const bool relaxed = fabs(p.d0) - 0.98 * p.umax * dt > p.dmax;

        // ---- 2. solve
        const LaneResult r = lane_solve(c, p, m, xh, warm_ok, cfg.tol_mu, cfg.tol_res, cfg.max_iter);
        warm_ok = warm_flag != nullptr && a.mode != MODE_SOLVE && r.constrained && r.warm_valid;
        int status = r.status;
        if (relaxed) status = max(status, 4);
        status = max(status, lap_flag);
        if (r.constrained) { mu = r.mu; res = r.res; }
        iters_tot += r.iters; status_max = max(status_max, status); nconstr += r.constrained;

        // ---- 3. outputs, plant
        u0 = c.at(LF_U, 0);
        if (a.U_out) for (int k = 0; k < N; k++) a.U_out[(size_t)s * N + k] = c.at(LF_U, k);
        if (a.traj_out) {
            for (int k = 0; k <= N; k++) {                                 // LMPC2.py:337-345
                double *o = a.traj_out + ((size_t)s * (N + 1) + k) * 7;
                const double psi = c.at(LF_X + 2, k);
                o[0] = c.at(LF_X, k); o[1] = c.at(LF_X + 1, k); o[2] = cos(psi); o[3] = sin(psi);
                o[4] = c.at(LF_X + 3, k); o[5] = c.at(LF_X + 4, k); o[6] = c.at(LF_X + 5, k);
            }
        }
        if (cfg.kind == MPC_KIND_LPV) {                                    // LPVMPC.py:313-327
            double *lxo = a.lin_x + (size_t)s * 6 * (N + 1);
            for (int k = 0; k <= N; k++) {
#pragma unroll
                for (int i = 0; i < 6; i++) lxo[i * (N + 1) + k] = c.at(LF_X + i, k);
            }
            for (int k = 0; k < N; k++) a.lin_u[(size_t)s * N + k] = c.at(LF_U, k);
        }
        if (a.mode != MODE_SOLVE) {
            const double oldx = plant[0];
            if (cl_on) {
                double *rec = a.stat_out + STAT_W * (size_t)s + STAT_CL;
                const double err[8] = {r0x, r0y, r0s, xh[3], xh[4], xh[5], u0, 0.0};
                const double tv = cl_pair_violation(plant[2], plant[3]);
#pragma unroll
                for (int i = 0; i < CL_STAT_W; i++)
                    rec[i] = cl_step((t > 0 || a.accumulate != 0) ? rec[i] : 0.0, i, cl_value(i, i < 8 ? err[i] : 0.0, tv, u0, plant[7], 0.0));
            }
            plant_rk4_c<10>(plant, u0, dt, pc);                            // skidpad_simulator.py:128
            if (a.mode == MODE_SKIDPAD && plant[0] > SK_CENTER && oldx <= SK_CENTER) lapv += 1;   // :91-93
            if (a.u_hist) a.u_hist[(size_t)t * a.B + s] = u0;
            if (a.x_hist) {
#pragma unroll
                for (int i = 0; i < 8; i++) a.x_hist[((size_t)t * a.B + s) * 8 + i] = plant[i];
            }
        }
    }

    if (a.mode != MODE_SOLVE) {
#pragma unroll
        for (int i = 0; i < 8; i++) a.state[(size_t)8 * s + i] = plant[i];
        if (a.mode == MODE_SKIDPAD) { a.lap[s] = lapv; a.prev_progress[s] = prevp; }
    }
#ifdef MPC_LANE_TIMING
    if (s == 0)
        printf("lane timing (sum over waves so far, cycles): setup %lld step %lld back0 %lld fwd0 %lld back1 %lld fwd1 %lld\n", g_lane_t[0], g_lane_t[1],
               g_lane_t[2], g_lane_t[3], g_lane_t[4], g_lane_t[5]);
#endif
    if (warm_flag) *warm_flag = warm_ok ? 1.0 : 0.0;
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

template <int LPW, int WPB>
static hipError_t launch_lanes_t(const DevCfg &cfg, const SolveArgs &a, double *ws, hipStream_t st)
{
    const int nblk = (a.B + LPW - 1) / LPW;
    hipLaunchKernelGGL((k_mpc_lanes<LPW, WPB>), dim3((nblk + WPB - 1) / WPB), dim3(WPB * WAVE), 0, st, cfg, a, ws);
    return hipGetLastError();
}

// this build of the solver (with or without px) for `lpw` lanes per wave
hipError_t launch_variant(const DevCfg &cfg, const SolveArgs &a, double *ws, int lpw, hipStream_t st)
{
    switch (lpw) {
    case 64: return launch_lanes_t<64, 1>(cfg, a, ws, st);
    case 32: return launch_lanes_t<32, 1>(cfg, a, ws, st);
    case 16: return launch_lanes_t<16, 1>(cfg, a, ws, st);
    default: return launch_lanes_t<8, 1>(cfg, a, ws, st);
    }
}

}  // namespace lanes / lanes_nopx

#if !LANE_X0
namespace lanes_nopx { hipError_t launch_variant(const DevCfg &cfg, const SolveArgs &a, double *ws, int lpw, hipStream_t st); }

// Lanes per wave that carry a scenario, by batch size.  A wave runs alone on its compute unit at the rate its own
// instruction stream issues, whatever the number of active lanes, so full waves are the cheapest as soon as there are
// scenarios for them; below that, narrower waves lose fewer iterations to the slowest lane of the wave.  (Measured on the
// LPV workload at N = 80: 64 lanes win from 4096 scenarios up, and two waves on one compute unit already contend for its
// memory pipeline.)  MPC_LANES_LPW overrides the choice for experiments.
static int lanes_per_wave(int B)
{
    static int forced = -1;
    if (forced < 0) { const char *e = getenv("MPC_LANES_LPW"); forced = e ? atoi(e) : 0; }
    if (forced == 64 || forced == 32 || forced == 16 || forced == 8) return forced;
    return B >= 4096 ? 64 : B >= 2048 ? 16 : 8;
}

size_t lanes_workspace_doubles(int N, int B, int term_m)
{
    const size_t lpw = (size_t)lanes_per_wave(B);
    return ((size_t)B + lpw - 1) / lpw * lpw * ((size_t)lanes::LN_FIELDS * (size_t)(N + 1) + (size_t)4 * term_m);
}

// `Bcap` is the batch size the workspace was laid out for (the handle's capacity): it fixes the lanes per wave, so that
// a scenario keeps its place in the workspace (its warm-start record) whatever the batch size of a call.
hipError_t launch_mpc_lanes(const DevCfg &cfg, const SolveArgs &a, double *ws, size_t Bcap, hipStream_t st)
{
    if (cfg.kind == MPC_KIND_NL || a.mode == MODE_OFL || !ws) return hipErrorInvalidValue;
    if (((size_t)(cfg.N + 1) * lanes::LN_FIELDS + (size_t)4 * a.term_m) * 64 * 8 >= ((size_t)1 << 31)) return hipErrorInvalidValue;   // a wave's block must fit a 32-bit byte offset
    if (a.term_m < 0 || (a.term_m > 0 && !a.term_C)) return hipErrorInvalidValue;
    // px takes no part when nothing weighs it (a.px_free: stage weight zero and a zero first row and column of the
    // terminal weight, in the controller's weights or in every row of the per-scenario cost table)
    const bool nopx = a.px_free != 0;
    const int lpw = lanes_per_wave((int)Bcap);
    return nopx ? lanes_nopx::launch_variant(cfg, a, ws, lpw, st) : lanes::launch_variant(cfg, a, ws, lpw, st);
}
#endif

}  // namespace mpc
