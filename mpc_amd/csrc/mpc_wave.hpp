// Wave-level helpers shared by the templated (mpc_solver.hip) and the any-horizon (mpc_generic.hip) kernels.
#pragma once
#include "mpc_device.hpp"
#include "mpc_kernels.h"

#pragma clang fp contract(fast)

namespace mpc {

// row[i] += src[i] * vc for i < N, with src in LDS at wave-uniform addresses (broadcast reads).  The reads go out in
// groups of CH ahead of their multiply-adds: left to itself the scheduler waits for every read right where it is used,
// and the accumulation becomes a chain of LDS round trips.
template <int N>
__device__ __forceinline__ void axpy_from_lds(double (&row)[N], const double *src, double vc)
{
    constexpr int CH = (N % 10 == 0) ? 10 : 8, G = N / CH;
    static_assert(N % CH == 0, "horizon must be a multiple of 10 or of 8");
    double t[2][CH];                       // the reads of group g + 1 are in flight while group g is accumulated
#pragma unroll
    for (int u = 0; u < CH; u++) t[0][u] = src[u];
#pragma unroll
    for (int g = 0; g < G; g++) {
        if (g + 1 < G) {
#pragma unroll
            for (int u = 0; u < CH; u++) t[(g + 1) & 1][u] = src[(g + 1) * CH + u];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CH; u++) row[g * CH + u] = fma(t[g & 1][u], vc, row[g * CH + u]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// sum_k K[k * N + col] * r[k], k < N: column `col` of K against a wave-uniform vector (r in LDS: broadcast reads).  The
// reads go out one group ahead of the multiply-adds, which run on two accumulators: issued one by one they form a
// chain of N memory round trips and N dependent fma latencies.
template <int N>
__device__ __forceinline__ double dot_column(const double *K, const double *r, int col)
{
#ifndef MPC_DOT_CH
#define MPC_DOT_CH 10
#endif
    constexpr int CH = (N % 10 == 0) ? MPC_DOT_CH : 8, G = N / CH;
    static_assert(N % CH == 0, "horizon must be a multiple of 10 or of 8");
    double kv[2][CH], rv[2][CH];
#pragma unroll
    for (int u = 0; u < CH; u++) { kv[0][u] = K[u * N + col]; rv[0][u] = r[u]; }
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
    for (int g = 0; g < G; g++) {
        if (g + 1 < G) {
#pragma unroll
            for (int u = 0; u < CH; u++) { kv[(g + 1) & 1][u] = K[((g + 1) * CH + u) * N + col]; rv[(g + 1) & 1][u] = r[(g + 1) * CH + u]; }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < CH; u += 2) {
            acc0 = fma(kv[g & 1][u], rv[g & 1][u], acc0);
            if (u + 1 < CH) acc1 = fma(kv[g & 1][u + 1], rv[g & 1][u + 1], acc1);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    return acc0 + acc1;
}

// LDS hand-off between the lanes of ONE wave: DS operations of a wave execute in order, so only the compiler
// has to be kept from moving accesses across this point.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Wave-uniform copy of a pointer that the compiler cannot see through: the loads behind it are neither treated
// as per-lane addressed nor hoisted out of the loop the call sits in.
__device__ __forceinline__ const double *opaque_uniform(const double *p)
{
    const unsigned long long v = (unsigned long long)p;
    unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return (const double *)(((unsigned long long)hi << 32) | lo);
}

// A copy of a per-lane value that the compiler cannot see through: what is computed from it stays where it is
// written (no hoisting of a whole unrolled loop's worth of loop-invariant temporaries into spilled registers).
__device__ __forceinline__ double opaque_value(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    asm volatile("" : "+v"(lo), "+v"(hi));
    return __hiloint2double(hi, lo);
}

// Stores row[0..N-1] of this lane to dst[i * N] (column `lane` of a row-major N x N matrix in global memory).  The
// address is rebuilt every ten rows from a value the compiler cannot see through, so the rows in between are reached
// by immediate offsets and no table of N hoisted addresses is kept (and spilled).
template <int N>
__device__ __forceinline__ void store_column(double *dst, const double (&row)[N])
{
    typedef __attribute__((address_space(1))) double gd;
#pragma unroll
    for (int c0 = 0; c0 < N; c0 += 10) {
        unsigned long long a = (unsigned long long)(dst + c0 * N);
        unsigned lo = (unsigned)a, hi = (unsigned)(a >> 32);
        asm volatile("" : "+v"(lo), "+v"(hi));
        gd *p = (gd *)(((unsigned long long)hi << 32) | lo);
#pragma unroll
        for (int u = 0; u < 10; u++)
            if (c0 + u < N) p[u * N] = row[c0 + u];
    }
}

// 1/x to ~1 ulp: hardware estimate plus two Newton steps (the pivots of an SPD factorisation are never 0/inf)
__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

constexpr int STAGE_W = 10;

// Forward-Euler stage matrices of the controller model, as 10 scalars (the rest of A_k is structural):
//   px+ = px + e02 psi + e03 vy + c0      py+ = py + e12 psi + e13 vy + c1      psi+ = psi + dt r
//   vy+ = f33 vy + f34 r + f35 delta (+ dt d_hat)      r+ = f43 vy + f44 r + f45 delta      delta+ = delta + dt u
__device__ __forceinline__ void stage_model(const DevCfg &cfg, const ModelParams &mp, double vx, const double *lx, int N,
                                            int k, double *st)
{
    const double dt = cfg.dt;
    double a11, a12, a21, a22;
    lat_coeffs(mp, vx, cfg.eps, a11, a12, a21, a22);
    st[4] = 1.0 + dt * a11; st[5] = dt * a12; st[6] = dt * a21; st[7] = 1.0 + dt * a22;
    if (cfg.kind == MPC_KIND_LPV) {
        // x+ = x + dt (J(xb)(x - xb) + B (u - ub) + f(xb, vx, ub)), LPVMPC.py:125-171.  Linearisation point per
        // LPVMPC.py:260-297: "reference" takes ref_points[i,3], a column that is always 0, as heading;
        // "prev_iter" takes the previous solution shifted by one stage.
        const double psib = cfg.lpv_lin_prev ? lx[2 * (N + 1) + k + 1] : 0.0;
        const double vyb = lx[3 * (N + 1) + k + 1];
        const double sn = sin(psib), co = cos(psib);
        const double J02 = -vx * sn - vyb * co, J03 = -sn, J12 = vx * co - vyb * sn, J13 = co;
        st[0] = dt * J02; st[1] = dt * J03; st[2] = dt * J12; st[3] = dt * J13;
        st[8] = dt * ((vx * co - vyb * sn) - (J02 * psib + J03 * vyb));
        st[9] = dt * ((vx * sn + vyb * co) - (J12 * psib + J13 * vyb));
    } else {
        st[0] = 0.0; st[1] = 0.0; st[2] = dt * vx; st[3] = dt;          // LMPC2.py:127-129
        st[8] = dt * vx; st[9] = 0.0;
    }
}

// ------------------------------------------------------------------------------------------------
// Wave-cooperative EKF on LDS-resident data (10 states, 7 measured channels).  Lane e handles covariance entries
// e and e + 64; uniform scalars are recomputed by every lane.
constexpr int EKF_LDS = 10 + 100 + 100 + 100 + 7 * 17;   // xe, P, F, T, augmented system
// measured channels [0,1,2,3,4,6,7] (continuous_dynamics.py:35-45)
__device__ __forceinline__ int meas_idx(int c) { return c < 5 ? c : c + 1; }

// EKF.py:38-43: x <- x + dt f(x,u) (forward Euler), F at the pre-update estimate, P <- F P F' + Q
__device__ __forceinline__ void ekf_time_update_wave(int lane, double *xe, double *P, double *F, double *T, double u,
                                                     double dt, const ModelParams &mp, const double *qd)
{
    double x[10], xd[10];
#pragma unroll
    for (int i = 0; i < 10; i++) x[i] = xe[i];
    plant_f<10>(x, u, mp, xd);
    for (int e = lane; e < 100; e += WAVE) F[e] = (e / 10 == e % 10) ? 1.0 : 0.0;
    wave_sync();
    if (lane == 0) ekf_jacobian_entries(x, dt, mp, F);
    wave_sync();
    for (int e = lane; e < 100; e += WAVE) {
        const int i = e / 10, j = e % 10;
        double acc = 0.0;
        for (int k = 0; k < 10; k++) acc += F[i * 10 + k] * P[k * 10 + j];
        T[e] = acc;
    }
    wave_sync();
    for (int e = lane; e < 100; e += WAVE) {
        const int i = e / 10, j = e % 10;
        double acc = 0.0;
        for (int k = 0; k < 10; k++) acc += T[i * 10 + k] * F[j * 10 + k];
        P[e] = acc + (i == j ? qd[i] : 0.0);
    }
    if (lane < 10) {
        double xv = x[0], dv = xd[0];
#pragma unroll
        for (int i = 1; i < 10; i++) { xv = (lane == i) ? x[i] : xv; dv = (lane == i) ? xd[i] : dv; }
        xe[lane] = dv * dt + xv;
    }
    wave_sync();
}

// EKF.py:45-49,65-74: K = solve(S', (P H')')' with S = H P H' + R, x += K (y - H x), P -= K H P.  The 7x7 system with
// its 10 right-hand sides is eliminated in LDS with partial pivoting (what numpy.linalg.solve does).
__device__ __forceinline__ void ekf_measurement_update_wave(int lane, double *xe, double *P, double *T, double *Aug,
                                                            const double *y, const double *rd)
{
    // Aug[a][0..6] = S'[a][:], Aug[a][7 + i] = (P H')[i][a]
    for (int e = lane; e < 7 * 17; e += WAVE) {
        const int a = e / 17, c = e % 17;
        Aug[e] = (c < 7) ? P[meas_idx(c) * 10 + meas_idx(a)] + (a == c ? rd[a] : 0.0) : P[(c - 7) * 10 + meas_idx(a)];
    }
    wave_sync();
    for (int c = 0; c < 7; c++) {
        int piv = c;
        double best = fabs(Aug[c * 17 + c]);
        for (int r = c + 1; r < 7; r++) {
            const double v = fabs(Aug[r * 17 + c]);
            if (v > best) { best = v; piv = r; }
        }
        if (piv != c && lane < 17) {
            const double t = Aug[c * 17 + lane];
            Aug[c * 17 + lane] = Aug[piv * 17 + lane];
            Aug[piv * 17 + lane] = t;
        }
        wave_sync();
        const double pc = Aug[c * 17 + c];
        double f[7];
#pragma unroll
        for (int r = 0; r < 7; r++) f[r] = (r > c) ? Aug[r * 17 + c] / pc : 0.0;
        wave_sync();
        if (lane < 17 && lane >= c) {
            const double top = Aug[c * 17 + lane];
#pragma unroll
            for (int r = 0; r < 7; r++)
                if (r > c) Aug[r * 17 + lane] -= f[r] * top;
        }
        wave_sync();
    }
    if (lane < 10) {   // back substitution of right-hand side `lane`; K[lane][a] ends up in Aug[a][7 + lane]
        for (int c = 6; c >= 0; c--) {
            double acc = Aug[c * 17 + 7 + lane];
            for (int k = c + 1; k < 7; k++) acc -= Aug[c * 17 + k] * Aug[k * 17 + 7 + lane];
            Aug[c * 17 + 7 + lane] = acc / Aug[c * 17 + c];
        }
    }
    wave_sync();
    for (int e = lane; e < 100; e += WAVE) {
        const int i = e / 10, j = e % 10;
        double acc = 0.0;
        for (int a = 0; a < 7; a++) acc += Aug[a * 17 + 7 + i] * P[meas_idx(a) * 10 + j];
        T[e] = P[e] - acc;
    }
    double dx = 0.0;
    if (lane < 10)
        for (int a = 0; a < 7; a++) dx += Aug[a * 17 + 7 + lane] * (y[a] - xe[meas_idx(a)]);
    wave_sync();
    for (int e = lane; e < 100; e += WAVE) P[e] = T[e];
    if (lane < 10) xe[lane] += dx;
    wave_sync();
}

// warm: per-scenario record of IPM_WARM_DOUBLES doubles in global memory (inputs and the four multiplier vectors of the
// last solve, [5][64], then a validity flag; the any-horizon kernel uses the same layout with rows of 128);
// warm_mode 0 = cold start, 1 = start from the record, 2 = start from the record moved one stage ahead (the next
// closed-loop step).  The record is rewritten at exit whenever warm != nullptr.
struct IpmParams { double dt, umax, dmax, tol_mu, tol_res; int max_iter; double *warm; int warm_mode; int nu; };   // nu: see interior_point()
struct IpmResult { double U, mu, res; int iters, status; bool warm_valid; };   // warm_valid: the record may start the next solve


}  // namespace mpc
