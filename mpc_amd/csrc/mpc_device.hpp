// Device-side building blocks shared by the kernels of libmpc_amd (gfx950 only, wave64).
#pragma once
#include <hip/hip_runtime.h>

namespace mpc {

constexpr int WAVE = 64;

// Uniform broadcast of lane `src` (wave-uniform index) through v_readlane: the result lives in SGPRs.
__device__ __forceinline__ double bcast(double v, int src)
{
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// The same broadcast with the two v_readlane pinned in program order (volatile asm; `src` must fold to a constant).
// Issued in groups ahead of the arithmetic that consumes them, the reads overlap instead of forming
// readlane-readlane-wait-fma chains on one SGPR pair.
__device__ __forceinline__ double bcast_pinned(double v, int src)
{
    int lo, hi;
    asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(lo) : "v"(__double2loint(v)), "n"(src));
    asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(hi) : "v"(__double2hiint(v)), "n"(src));
    return __hiloint2double(hi, lo);
}

// Cross-lane arithmetic through DPP (data-parallel primitives: the operand of a VALU move comes from another lane of
// the same 16-lane row) and v_readlane, not through ds_bpermute: an LDS round trip per step of a scan or reduction is a
// chain of ~100-cycle latencies, a DPP step is one VALU issue.  dpp<CTRL>(v, fill) = v of the lane CTRL selects, or
// `fill` where that lane lies outside the row.
constexpr int DPP_ROW_SHL = 0x100, DPP_ROW_SHR = 0x110;     // + n: lane i reads lane i + n / i - n of its row
template <int CTRL>
__device__ __forceinline__ double dpp(double v, double fill)
{
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(fill), __double2loint(v), CTRL, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(fill), __double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

struct OpSum { static __device__ __forceinline__ double id() { return 0.0; } static __device__ __forceinline__ double f(double a, double b) { return a + b; } };
struct OpMax { static __device__ __forceinline__ double id() { return -__builtin_huge_val(); } static __device__ __forceinline__ double f(double a, double b) { return fmax(a, b); } };
struct OpMin { static __device__ __forceinline__ double id() { return __builtin_huge_val(); } static __device__ __forceinline__ double f(double a, double b) { return fmin(a, b); } };

// Reduction over the 64 lanes, the same value in every lane: four DPP steps leave each row's result in its last lane,
// the four row results are read out and combined.
template <class Op>
__device__ __forceinline__ double wave_reduce(double v)
{
    v = Op::f(v, dpp<DPP_ROW_SHR + 1>(v, Op::id()));
    v = Op::f(v, dpp<DPP_ROW_SHR + 2>(v, Op::id()));
    v = Op::f(v, dpp<DPP_ROW_SHR + 4>(v, Op::id()));
    v = Op::f(v, dpp<DPP_ROW_SHR + 8>(v, Op::id()));
    return Op::f(Op::f(bcast(v, 15), bcast(v, 31)), Op::f(bcast(v, 47), bcast(v, 63)));
}
// A copy of a per-lane integer that the compiler cannot see through.  Comparisons of the lane index with unrolled loop
// counters are loop-invariant for the caller's iteration loop; hoisted, their N mask pairs overflow the SGPR file and
// come back through v_writelane / v_readlane spills, which costs more than recomputing each compare where it is used.
__device__ __forceinline__ int opaque_int(int x)
{
    asm volatile("" : "+v"(x));
    return x;
}

__device__ __forceinline__ double wave_max(double v) { return wave_reduce<OpMax>(v); }
__device__ __forceinline__ double wave_min(double v) { return wave_reduce<OpMin>(v); }
__device__ __forceinline__ double wave_sum(double v) { return wave_reduce<OpSum>(v); }

// Inclusive scans over the 64 lanes (callers zero the lanes that do not take part): a scan inside each row of 16, then
// the totals of the rows before (after) a lane's row are added.
__device__ __forceinline__ double incl_prefix(double v, int lane)
{
    v += dpp<DPP_ROW_SHR + 1>(v, 0.0);
    v += dpp<DPP_ROW_SHR + 2>(v, 0.0);
    v += dpp<DPP_ROW_SHR + 4>(v, 0.0);
    v += dpp<DPP_ROW_SHR + 8>(v, 0.0);
    const double r0 = bcast(v, 15), r1 = bcast(v, 31), r2 = bcast(v, 47);
    const double r01 = r0 + r1, r012 = r01 + r2;
    return v + (lane < 16 ? 0.0 : lane < 32 ? r0 : lane < 48 ? r01 : r012);
}
__device__ __forceinline__ double incl_suffix(double v, int lane)
{
    v += dpp<DPP_ROW_SHL + 1>(v, 0.0);
    v += dpp<DPP_ROW_SHL + 2>(v, 0.0);
    v += dpp<DPP_ROW_SHL + 4>(v, 0.0);
    v += dpp<DPP_ROW_SHL + 8>(v, 0.0);
    const double r3 = bcast(v, 48), r2 = bcast(v, 32), r1 = bcast(v, 16);
    const double r32 = r3 + r2, r321 = r32 + r1;
    return v + (lane >= 48 ? 0.0 : lane >= 32 ? r3 : lane >= 16 ? r32 : r321);
}

// ---------------------------------------------------------------- model pieces (uniform or per lane)
struct ModelParams { double m, Iz, lf, lr, Cf, Cr; };

// Lateral coefficients of the controller model, with the whole bracket of a12 divided
// (LMPC2.py:133-149, LPVMPC.py:131-147, OFLMPC2.py:138-155).
__device__ __forceinline__ void lat_coeffs(const ModelParams &mp, double vx, double eps, double &a11, double &a12,
                                           double &a21, double &a22)
{
    a11 = -(mp.Cf + mp.Cr) / (mp.m * vx + eps);
    a12 = (-vx + (mp.Cr * mp.lr - mp.Cf * mp.lf)) / (mp.m * vx + eps);
    a21 = (mp.lr * mp.Cr - mp.lf * mp.Cf) / (mp.Iz * vx + eps);
    a22 = -(mp.lf * mp.lf * mp.Cf + mp.lr * mp.lr * mp.Cr) / (mp.Iz * vx + eps);
}

// Speed-dependent coefficients of the plant's lateral rows.  The longitudinal speed x[4] has zero derivative, so every
// stage of an RK4 step (and every step of a run) divides by the same values: computing the quotients once gives the
// same bits as evaluating continuous_dynamics.py:104-118 term by term.
struct PlantCoef { double vx, c11, c12a, b1, c21, c22, b2; };

__device__ __forceinline__ PlantCoef plant_coef(const ModelParams &mp, double vx)
{
    PlantCoef c;
    c.vx = vx;
    c.c11 = -(mp.Cf + mp.Cr) / (mp.m * vx);
    c.c12a = (mp.Cr * mp.lr - mp.Cf * mp.lf) / (mp.m * vx);
    c.b1 = mp.Cf / mp.m;
    c.c21 = (mp.lr * mp.Cr - mp.lf * mp.Cf) / (mp.Iz * vx);
    c.c22 = (mp.lf * mp.lf * mp.Cf + mp.lr * mp.lr * mp.Cr) / (mp.Iz * vx);
    c.b2 = (mp.Cf * mp.lf) / mp.Iz;
    return c;
}

// Plant right-hand side (continuous_dynamics.py:90-124); x has NX = 8 or 10 entries; c = plant_coef(mp, x[4]).
template <int NX>
__device__ __forceinline__ void plant_f_c(const double *x, double u, const PlantCoef &c, double *xd)
{
    double sd = 0.0, fd = 0.0;
    if (NX == 10) { sd = x[8]; fd = x[9]; }
    xd[0] = x[2] * x[4] - x[3] * x[5];
    xd[1] = x[3] * x[4] + x[2] * x[5];
    xd[2] = -x[6] * x[3];
    xd[3] = x[6] * x[2];
    xd[4] = 0.0;
    xd[5] = ((c.c11 * x[5] + (-x[4] + c.c12a) * x[6]) - c.b1 * (x[7] + sd) + fd);
    xd[6] = (c.c21 * x[5] - c.c22 * x[6] - c.b2 * (x[7] + sd));
    xd[7] = u;
    if (NX == 10) { xd[8] = 0.0; xd[9] = 0.0; }
}

template <int NX>
__device__ __forceinline__ void plant_f(const double *x, double u, const ModelParams &mp, double *xd)
{
    plant_f_c<NX>(x, u, plant_coef(mp, x[4]), xd);
}

// Classic RK4 with the input held (continuous_dynamics.py:126-133); c = plant_coef(mp, x[4]).
template <int NX>
__device__ __forceinline__ void plant_rk4_c(double *x, double u, double dt, const PlantCoef &c)
{
    double k1[NX], k2[NX], k3[NX], k4[NX], t[NX];
    plant_f_c<NX>(x, u, c, k1);
#pragma unroll
    for (int i = 0; i < NX; i++) t[i] = x[i] + dt / 2 * k1[i];
    plant_f_c<NX>(t, u, c, k2);
#pragma unroll
    for (int i = 0; i < NX; i++) t[i] = x[i] + dt / 2 * k2[i];
    plant_f_c<NX>(t, u, c, k3);
#pragma unroll
    for (int i = 0; i < NX; i++) t[i] = x[i] + dt * k3[i];
    plant_f_c<NX>(t, u, c, k4);
#pragma unroll
    // (the reference divides by 6; the product with 1/6 differs from that by at most one unit in the last place of the
    // increment, and a double-precision division is a dozen instructions per state on a path that is issue-bound)
    for (int i = 0; i < NX; i++) x[i] = x[i] + dt * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]) * (1.0 / 6.0);
}

template <int NX>
__device__ __forceinline__ void plant_rk4(double *x, double u, double dt, const ModelParams &mp)
{
    plant_rk4_c<NX>(x, u, dt, plant_coef(mp, x[4]));
}

// ---------------------------------------------------------------- closed-loop statistics
// What a batch of closed loops is judged by, per scenario, over the steps of one entry-point call.  The kernels keep
// CL_STAT_W running values per scenario and write them once per launch; the statistics kernel turns them into the
// quantities mpc_get_stats reports (the stage cost needs the weights, which the step loops therefore never touch):
//   CL_ERR .. CL_ERR + 7   sums of the squared components of [x0; u0] - yref_0 at the start of the horizon, x0 the
//                car-frame start state, u0 the applied input, yref_0 the first reference point.  Linear controllers
//                (LMPC2.py:220-253, 290-298): (x_ref, y_ref, sin psi_ref, vy, r, delta, u0, -); NL controller
//                (NLMPC.py:159-186, 208-216): (x_ref, y_ref, 1 - cos psi_ref, sin psi_ref, vy, r, delta, u0).
//                The stage cost of a step is dt * 1/2 * sum_c w_c e_c^2; e_1 is the lateral tracking error.
//   CL_TRIG_SUM / CL_TRIG_MAX   | |(cos, sin)| - 1 | of the plant's heading pair at the start of the step (the plant does
//                not renormalise it; skidpad_simulator.py:134 prints the same quantity for the predicted trajectory)
//   CL_PRED_TRIG_MAX   the same over the predicted trajectory, where the pair is a state of the model (NL controller;
//                the linear controllers rebuild the pair from the angle, so it lies on the unit circle to rounding)
//   CL_U_MAX, CL_STEER_MAX   largest |u0| and largest |steering angle| of the plant (at the start of a step); the
//                statistics kernel reports their excess over the rate and the angle bound (the bounds, like the weights,
//                stay out of the step loops: two more wave-uniform values alive across the solve are scalar-register
//                spills on every step)
// Everything a step contributes is known once its input is: nothing here waits for the plant's RK4 step, whose chain of
// dependent operations ends the closed-loop step (a wave per scenario is latency-bound there).
//   CL_STEPS     steps counted
constexpr int CL_STAT_W = 14;
enum { CL_ERR = 0, CL_TRIG_SUM = 8, CL_STEPS = 9, CL_U_MAX = 10, CL_TRIG_MAX = 11, CL_PRED_TRIG_MAX = 12, CL_STEER_MAX = 13 };

// | |(hc, hs)| - 1 | of a heading pair.  Near the unit circle (always, in practice: the pair drifts by rounding only)
// sqrt(1 + d) - 1 is summed as a series in d = hc^2 + hs^2 - 1, exact to 4e-10 of the value for |d| < 0.01; the closed
// loops are issue-bound in their steady state and a double-precision square root is ~25 instructions.
__device__ __forceinline__ double cl_pair_violation(double hc, double hs)
{
    const double d = fma(hc, hc, fma(hs, hs, -1.0));
    if (fabs(d) < 0.01) return fabs(d * (0.5 - d * (0.125 - d * 0.0625)));
    return fabs(sqrt(hc * hc + hs * hs) - 1.0);
}

// One step's contribution to statistic `i` (running value `acc`) from the step's value `v` for that statistic:
// i < 8 component i of [x0; u0] - yref_0 (squared and added); CL_TRIG_SUM the pair's violation (added); CL_STEPS 1.0;
// CL_U_MAX |u0|, CL_STEER_MAX |steering angle|, CL_TRIG_MAX the pair's violation, CL_PRED_TRIG_MAX that of the
// predicted trajectory (maxima).  In the wave-per-scenario kernels lane i < CL_STAT_W carries statistic i in one register.
__device__ __forceinline__ double cl_step(double acc, int i, double v)
{
    const double add = i == CL_TRIG_SUM ? v : v * v;      // (1.0 squared counts the step)
    return i >= CL_U_MAX ? fmax(acc, v) : acc + add;
}

// The step's value for statistic `i` from wave-uniform quantities (kernels that do not stage them in LDS): err = this
// lane's component of [x0; u0] - yref_0 (i < 8), tv the pair's violation, u0 the input, steer the steering angle, pred see above.
__device__ __forceinline__ double cl_value(int i, double err, double tv, double u0, double steer, double pred)
{
    double v = err;
    v = (i == CL_TRIG_SUM || i == CL_TRIG_MAX) ? tv : v;
    v = i == CL_STEPS ? 1.0 : v;
    v = i == CL_U_MAX ? fabs(u0) : v;
    v = i == CL_STEER_MAX ? fabs(steer) : v;
    v = i == CL_PRED_TRIG_MAX ? pred : v;
    return v;
}

// ---------------------------------------------------------------- EKF linearisation
// Off-diagonal and non-unit entries of F = I + dt*A of Dynamics.jacobian_forward_euler (continuous_dynamics.py:167-210),
// including its log(vx) terms, the positive tf22dr and the 1/m side-force column (:151-165) the estimator depends on.
// F (row-major 10x10) must already hold the identity.
__device__ __forceinline__ void ekf_jacobian_entries(const double *x, double dt, const ModelParams &mp, double *F)
{
    const double lg = log(x[4]);
    const double cs = mp.Cf + mp.Cr, cd = mp.Cr * mp.lr - mp.Cf * mp.lf, ci = mp.lf * mp.lf * mp.Cf + mp.lr * mp.lr * mp.Cr;
    const double tf11dvx = -cs / mp.m * x[5] * lg;
    const double tf11dvy = -cs / (mp.m * x[4]);
    const double tf12dvx = -x[6] + cd / mp.m * lg;
    const double tf12dr = -x[4] + cd / (mp.m * x[4]);
    const double tf1dd = -mp.Cf / mp.m;
    const double tf21dvx = (mp.lr * mp.Cr - mp.lf * mp.Cf) / mp.Iz * x[5] * lg;
    const double tf21dvy = (mp.lr * mp.Cr - mp.lf * mp.Cf) / (mp.Iz * x[4]);
    const double tf22dvx = (ci / mp.Iz * x[6]) * lg;
    const double tf22dr = ci / (mp.Iz * x[4]);
    const double tf2dd = -(mp.Cf * mp.lf) / mp.Iz;
#define FF(r, c) F[(r) * 10 + (c)]
    FF(0, 2) = dt * x[4]; FF(0, 3) = dt * -x[5]; FF(0, 4) = dt * x[2]; FF(0, 5) = dt * -x[3];
    FF(1, 2) = dt * x[5]; FF(1, 3) = dt * x[4];  FF(1, 4) = dt * x[3]; FF(1, 5) = dt * x[2];
    FF(2, 3) = dt * -x[6]; FF(2, 6) = dt * -x[3];
    FF(3, 2) = dt * x[6];  FF(3, 6) = dt * x[2];
    FF(5, 4) = dt * (tf11dvx + tf12dvx); FF(5, 5) = 1.0 + dt * tf11dvy; FF(5, 6) = dt * tf12dr; FF(5, 7) = dt * tf1dd;
    FF(6, 4) = dt * (tf21dvx + tf22dvx); FF(6, 5) = dt * tf21dvy; FF(6, 6) = 1.0 + dt * tf22dr; FF(6, 7) = dt * tf2dd;
    FF(5, 8) = dt * (-mp.Cf / mp.m); FF(6, 8) = dt * (-(mp.Cf * mp.lf) / mp.Iz);
    FF(5, 9) = dt * (1.0 / mp.m);
#undef FF
}

// ---------------------------------------------------------------- planners
constexpr double SK_R = 9.125, SK_CENTER = 16.75;
constexpr double PI = 3.14159265358979323846;
constexpr double SK_LAP = 2 * SK_R * PI;

// utils/path_planning.py:196-209
__device__ __forceinline__ double sk_pos2progress(double x, double y, int lap)
{
    if (lap == 0) return x;
    if (lap > 0 && lap < 3) return (-atan2(x - SK_CENTER, -SK_R - y) + PI) * SK_R + (lap - 1) * SK_LAP + SK_CENTER;
    if (lap > 2 && lap < 5) return (atan2(SK_CENTER - x, y - SK_R) + PI) * SK_R + (lap - 1) * SK_LAP + SK_CENTER;
    return x + 4 * SK_LAP;
}

// utils/path_planning.py:131-167; "a // b > 0" with b > 0 is "a >= b".  One sincos per pose: the second circle's
// position angle a = b - 4 pi has the sine and cosine of b (the reference evaluates both angles; the two agree to the
// rounding of b, a few 1e-15).
__device__ __forceinline__ void sk_pose(double q, double *o)
{
    const double b = (q - SK_CENTER) / SK_R;
    if (q - SK_CENTER >= 4 * SK_LAP) { o[0] = q - 4 * SK_LAP; o[1] = 0.0; o[2] = 1.0; o[3] = 0.0; }
    else if (q - SK_CENTER > 0) {
        double sb, cb;
        sincos(b, &sb, &cb);
        if (q - SK_CENTER >= 2 * SK_LAP) { o[0] = SK_CENTER + sb * SK_R; o[1] = SK_R - cb * SK_R; o[2] = cb; o[3] = sb; }
        else { o[0] = SK_CENTER + sb * SK_R; o[1] = -SK_R + cb * SK_R; o[2] = cb; o[3] = -sb; }
    } else { o[0] = q; o[1] = 0.0; o[2] = 1.0; o[3] = 0.0; }
}

// sk_pose with the sine and cosine of b = (q - centre) / r handed in.
__device__ __forceinline__ void sk_pose_sc(double q, double sb, double cb, double *o)
{
    if (q - SK_CENTER >= 4 * SK_LAP) { o[0] = q - 4 * SK_LAP; o[1] = 0.0; o[2] = 1.0; o[3] = 0.0; }
    else if (q - SK_CENTER > 0) {
        if (q - SK_CENTER >= 2 * SK_LAP) { o[0] = SK_CENTER + sb * SK_R; o[1] = SK_R - cb * SK_R; o[2] = cb; o[3] = sb; }
        else { o[0] = SK_CENTER + sb * SK_R; o[1] = -SK_R + cb * SK_R; o[2] = cb; o[3] = -sb; }
    } else { o[0] = q; o[1] = 0.0; o[2] = 1.0; o[3] = 0.0; }
}

// Cosine and sine of b0 = (p0 - centre) / r for the progress p0 = sk_pos2progress(x, y, lap).  On the circles p0 is an
// arctangent of the car's position about the circle's centre, so the pair is that position's direction (with the signs
// of the lap's formula) and needs no trigonometric function; elsewhere sincos.  The waypoints ahead are this pair turned
// by the fixed angles of their distances (the fused step loops keep those turns in a table per scenario), which agrees
// with evaluating sincos per waypoint to a few units in the last place.
__device__ __forceinline__ void sk_progress_sincos(double x, double y, int lap, double p0, double &c0, double &s0)
{
    double X = 0.0, Y = 0.0, sgn = 1.0;
    if (lap > 0 && lap < 3) { X = x - SK_CENTER; Y = -SK_R - y; }                   // b0 = pi - atan2(X, Y) + 2 pi (lap - 1)
    else if (lap > 2 && lap < 5) { X = SK_CENTER - x; Y = y - SK_R; sgn = -1.0; }   // b0 = pi + atan2(X, Y) + 2 pi (lap - 1)
    const double n2 = X * X + Y * Y;
    if (n2 > 0.0) {
        const double rn = 1.0 / sqrt(n2);
        c0 = -Y * rn; s0 = sgn * X * rn;
    } else
        sincos((p0 - SK_CENTER) / SK_R, &s0, &c0);
}

// utils/step_planning.py:63-80: a unit step in y at progress 0, reached over `ramp` metres of progress (ramp_length;
// 0 = the pure step of the reference's simulators), heading along x.
__device__ __forceinline__ void step_pose(double q, double ramp, double *o)
{
    o[0] = q; o[1] = (q < 0.0) ? 0.0 : (q < ramp) ? q / ramp : 1.0; o[2] = 1.0; o[3] = 0.0;
}

// cos and sin of atan2(hs, hc) without the angle: the heading vector normalised (skidpad_simulator.py:104 takes the
// angle and path_planning.py:253-260 its cosine and sine again)
__device__ __forceinline__ void unit_heading(double hc, double hs, double &ch, double &sh)
{
    const double n2 = hc * hc + hs * hs;
    if (n2 > 0.0) {
        const double rn = 1.0 / sqrt(n2);
        ch = hc * rn; sh = hs * rn;
    } else { ch = 1.0; sh = 0.0; }
}

// translate by the car position and right-multiply by [[c,-s],[s,c]] (path_planning.py:253-260)
__device__ __forceinline__ void to_car_frame(double *o, double x, double y, double ch, double sh)
{
    double px = o[0] - x, py = o[1] - y, hx = o[2], hy = o[3];
    o[0] = px * ch + py * sh;
    o[1] = -px * sh + py * ch;
    o[2] = hx * ch + hy * sh;
    o[3] = -hx * sh + hy * ch;
}

// cumsum(speeds[:-1])[k-1] for a constant speed v with np.cumsum's sequential accumulation (0 for k = 0)
__device__ __forceinline__ double speed_cumsum(double v, int k)
{
    if (k == 0) return 0.0;
    double cs = v;
    for (int i = 1; i < k; i++) cs += v;
    return cs;
}

// progress of waypoint k: p0 + cumsum(speeds[:-1])[k-1] * dt
__device__ __forceinline__ double waypoint_progress(double p0, double v, double dt, int k)
{
    if (k == 0) return p0;
    return p0 + speed_cumsum(v, k) * dt;
}

// Speed zones of the skidpad planner (utils/path_planning.py:63-84): acceleration zone, first braking zone (to the
// slow-down speed), second braking zone (to standstill); f is the slow-down speed factor.
struct SpeedZones { double a1s, a1e, b1s, b1e, b2s, b2e, f; };

// utils/path_planning.py:93-126; a progress exactly on a zone edge falls through to the 0.1 of the last branch.
__device__ __forceinline__ double sk_progress2speed(const SpeedZones &z, double v, double q)
{
    if (q > z.b2e) return 0.0;
    if (z.b2e > q && q > z.b2s) return v * z.f - (q - z.b2s) * v * z.f;
    if (z.b2s > q && q > z.b1e) return v * z.f;
    if (z.b1e > q && q > z.b1s) return v - (q - z.b1s) * v * (1.0 - z.f);
    if (z.b1s > q && q > z.a1e) return v;
    if (z.a1e > q && q > z.a1s) return (q - z.a1s) * (v - 0.1) + 0.1;
    return 0.1;
}

// One stage of the speed profile (utils/path_planning.py:242-248): the speed the planner keeps is the one at the
// midpoint of a step taken with the speed at the stage's progress, both capped at the target speed.
__device__ __forceinline__ double sk_stage_speed(const SpeedZones &z, double v, double q, double dt)
{
    const double s0 = fmin(sk_progress2speed(z, v, q), v);
    return fmin(sk_progress2speed(z, v, q + s0 * dt / 2), v);
}

}  // namespace mpc
