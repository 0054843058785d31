// Device-side building blocks of the batched iterative linear MPC engine (gfx950).
//
// Shared by every solve kernel: the models, the bounds of the free controls, numeric helpers;
// and the one-instance-per-lane dense solver over a storage policy
//   RegStore<T>  every per-instance array is a local array with compile-time
//                indices after full unrolling (VGPR/AGPR resident); the
//                condensed Hessian and gradient sit in LDS as [element][lane].
//
// Algorithm per instance (float64 throughout):
//   rollout yaw -> T linearisations -> eliminate states (condensed QP in the 2
//   free controls per stage) -> primal-dual interior point (Mehrotra
//   predictor-corrector, strictly feasible start, slacks recomputed from the
//   iterate) with one Cholesky per iteration -> controls, predicted states,
//   objective.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "mpcx.h"

namespace mpcx {

#ifdef MPCX_MARK_PHASES
#define MPCX_PHASE(name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; PHASE " name ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define MPCX_PHASE(name)
#endif

constexpr int kWave = 64;
constexpr double kSlackMin = 1e-16;   // floor on a recomputed slack
constexpr double kMu0 = 0.1;          // initial complementarity s*lambda
#ifndef MPCX_TOL_D
#define MPCX_TOL_D 1e-8
#endif
#ifndef MPCX_TOL_MU
#define MPCX_TOL_MU 1e-11
#endif
constexpr double kTolD = MPCX_TOL_D;    // relative stationarity
constexpr double kTolMu = MPCX_TOL_MU;  // mean complementarity
constexpr double kMuFloor = 1e-13;    // below this the recomputed slacks are rounding noise: stop
constexpr double kLooseD = 1e-6;      // "inaccurate but accepted" thresholds
constexpr double kLooseMu = 1e-7;
constexpr double kEtaCap = 0.999;
#ifndef MPCX_SWITCH_MU
#define MPCX_SWITCH_MU 1e-6
#endif
constexpr double kSwitchMu = MPCX_SWITCH_MU;   // stage-wise kernel: complementarity at which the active set is read off the interior point
constexpr double kSwitchD = 1e-3;     // ... and the relative dual residual it asks for
constexpr double kMinWidth = 2e-9;    // a first-stage window narrower than this is widened to it
constexpr double kBig = 1e300;
#ifndef MPCX_RHO_CAP
#define MPCX_RHO_CAP 1e10
#endif
constexpr double kRhoCap = MPCX_RHO_CAP;      // largest penalty parameter of the active-set rounds (long-horizon kernels)

__device__ __forceinline__ double rcp_raw(double x) { return __builtin_amdgcn_rcp(x); }
__device__ __forceinline__ double rcp_nr(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double rsqrt_nr(double x)
{
    double r = __builtin_amdgcn_rsq(x);
    double e = fma(-(x * r), r, 1.0);
    r = fma(0.5 * r, e, r);
    e = fma(-(x * r), r, 1.0);
    r = fma(0.5 * r, e, r);
    return r;
}
// Returns x, but the compiler may not assume so: used where a cheap value is deliberately
// recomputed in a later phase instead of being kept in registers across a factorisation
// (common-subexpression elimination would otherwise undo that and spill).
__device__ __forceinline__ double opaque(double x)
{
    asm volatile("" : "+v"(x));
    return x;
}
__device__ __forceinline__ bool finite_d(double x) { return fabs(x) <= 1.79769313486231570e308; }

// sin and cos together: three-constant Cody-Waite reduction by pi/2 with fused
// multiply-adds (each product is exact inside its fma and the running remainder stays
// below 1, so the reduced argument carries an absolute error near 1e-16 for any |x| whose
// quadrant count is still an exact double, far beyond any yaw or steer angle), then the
// classic degree-13/14 minimax kernels on [-pi/4, pi/4] (fdlibm's __kernel_sin /
// __kernel_cos coefficients): about one ulp.  Branch-free; NaN/Inf in gives NaN out.
__device__ __forceinline__ void sincos_fast(double x, double &sn, double &cs)
{
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632679489655800e+00, x);
    r = fma(-k, 6.12323399573676603587e-17, r);
    r = fma(-k, -1.49738490485919873423e-33, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double s0 = fma(z * r, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double c0 = fma(z * z, pc, fma(-0.5, z, 1.0));
    const int q = (int)(long long)k;
    const double sa = (q & 1) ? c0 : s0, ca = (q & 1) ? s0 : c0;
    sn = (q & 2) ? -sa : sa;
    cs = ((q + 1) & 2) ? -ca : ca;
}

// ------------------------------------------------------------------------------------
// models
// ------------------------------------------------------------------------------------

// A = I + (a02 at [0][2], a12 at [1][2]); Bm 3x4 (NU columns used); C[3]; omega = yaw
// rate of the plant at these controls (what predict_motion integrates).
// Diff-drive: mpc_diff_drive.py:220-238 and :92-100.
// Dual steer: mpc_dsr.py:397-426 entry by entry (B[2][0] uses sin(delta_r), B[1][3] has
// the (sin dr cos th - cos dr sin th) pattern, C multiplies by (delta + theta): these are
// the reference's formulas, not the analytic Jacobian) and :141-167 for the twist.
// The part of the linearisation that depends on the controls only: sines and cosines of the
// two steer angles (dual steer) and the yaw rate of the plant.
struct SteerTerms {
    double sf, cf, sr, cr, omega;
};

template <int MODEL>
__device__ __forceinline__ void steer_terms(const mpcx_variant &v, const double (&u)[4], SteerTerms &k)
{
    if (MODEL == MPCX_MODEL_DIFF_DRIVE) {
        k.sf = k.sr = 0.0;
        k.cf = k.cr = 1.0;
        k.omega = u[1];
    } else {
        sincos_fast(u[1], k.sf, k.cf);
        sincos_fast(u[3], k.sr, k.cr);
        k.omega = (k.sf * u[0] - k.sr * u[2]) / v.wb;
    }
}

template <int MODEL>
__device__ __forceinline__ void linearize_at(const mpcx_variant &v, double th, const double (&u)[4],
                                             const SteerTerms &k, double &a02, double &a12,
                                             double (&Bm)[3][4], double (&C)[3])
{
    const double dt = v.dt;
    double st, ct;
    sincos_fast(th, st, ct);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) Bm[r][c] = 0.0;
    if (MODEL == MPCX_MODEL_DIFF_DRIVE) {
        const double vr = u[0];
        a02 = -vr * st * dt;
        a12 = vr * ct * dt;
        Bm[0][0] = ct * dt;
        Bm[1][0] = st * dt;
        Bm[2][1] = dt;
        C[0] = vr * st * th * dt;
        C[1] = -vr * ct * th * dt;
        C[2] = 0.0;
    } else {
        const double v_f = u[0], d_f = u[1], v_r = u[2], d_r = u[3];
        const double sf = k.sf, cf = k.cf, sr = k.sr, cr = k.cr;
        const double h = 0.5 * dt, kk = 1.0 / v.wb * dt;
        const double vc = v_f * cf + v_r * cr, vs = v_f * sf + v_r * sr;
        a02 = -0.5 * dt * vc * st - 0.5 * dt * vs * ct;
        a12 = 0.5 * dt * vc * ct - 0.5 * dt * vs * st;
        Bm[0][0] = h * (cf * ct - sf * st);
        Bm[0][1] = -h * (sf * ct + cf * st) * v_f;
        Bm[0][2] = h * (cr * ct - sr * st);
        Bm[0][3] = -h * (sr * ct + cr * st) * v_r;
        Bm[1][0] = h * (cf * st + sf * ct);
        Bm[1][1] = -h * (sf * st - cf * ct) * v_f;
        Bm[1][2] = h * (cr * st + sr * ct);
        Bm[1][3] = -h * (sr * ct - cr * st) * v_r;
        Bm[2][0] = kk * sr;
        Bm[2][1] = kk * cf * v_f;
        Bm[2][2] = -kk * sr;
        Bm[2][3] = -kk * cr * v_r;
        const double af = d_f + th, ar = d_r + th;
        C[0] = h * (v_f * cf * st * af + v_r * cr * st * ar + v_f * sf * ct * af + v_r * sr * ct * ar);
        C[1] = h * (-v_f * cf * ct * af - v_r * cr * ct * ar + v_f * sf * st * af + v_r * sr * st * ar);
        C[2] = kk * (-cf * d_f * v_f + cr * d_r * v_r);
    }
}

template <int MODEL>
__device__ __forceinline__ void linearize(const mpcx_variant &v, double th, const double (&u)[4],
                                          double &a02, double &a12, double (&Bm)[3][4],
                                          double (&C)[3], double &omega)
{
    SteerTerms k;
    steer_terms<MODEL>(v, u, k);
    linearize_at<MODEL>(v, th, u, k, a02, a12, Bm, C);
    omega = k.omega;
}

// Reduced input matrix and offset of one stage under the variant's control equalities:
// u = E w + f  =>  bt = Bm E,  ct = C + Bm f.
__device__ __forceinline__ void reduce_controls(const mpcx_variant &v, const double (&Bm)[3][4],
                                                const double (&C)[3], double (&bt)[3][2], double (&ct)[3])
{
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if (v.eq_mode == MPCX_EQ_NONE) {
            bt[r][0] = Bm[r][0]; bt[r][1] = Bm[r][1]; ct[r] = C[r];
        } else if (v.eq_mode == MPCX_EQ_ACKERMANN) {
            bt[r][0] = Bm[r][0] + Bm[r][2]; bt[r][1] = Bm[r][1] - Bm[r][3]; ct[r] = C[r];
        } else if (v.eq_mode == MPCX_EQ_CRAB) {
            bt[r][0] = Bm[r][0] + Bm[r][2]; bt[r][1] = Bm[r][1] + Bm[r][3]; ct[r] = C[r];
        } else {
            bt[r][0] = Bm[r][0]; bt[r][1] = Bm[r][2];
            ct[r] = C[r] + (Bm[r][1] + Bm[r][3]) * v.steer_fixed;
        }
    }
}

// One plant step of predict_motion (mpc_diff_drive.py:92-100; mpc_dsr.py:114-123,141-167).
template <int MODEL>
__device__ __forceinline__ void plant_step(const mpcx_variant &v, double &x, double &y, double &yaw,
                                           const double (&u)[4])
{
    double s, c;
    sincos_fast(yaw, s, c);
    const double dt = v.dt;
    if (MODEL == MPCX_MODEL_DIFF_DRIVE) {
        x = x + u[0] * c * dt;
        y = y + u[0] * s * dt;
        yaw = yaw + u[1] * dt;
    } else {
        double sf, cf, sr, cr;
        sincos_fast(u[1], sf, cf);
        sincos_fast(u[3], sr, cr);
        const double v1x = cf * u[0], v1y = sf * u[0], v2x = cr * u[2], v2y = sr * u[2];
        const double vx = 0.5 * (v1x + v2x), vy = 0.5 * (v1y + v2y), w = (v1y - v2y) / v.wb;
        x = x + vx * c * dt - vy * s * dt;
        y = y + vx * s * dt + vy * c * dt;
        yaw = yaw + w * dt;
    }
}

// Yaw rate only (the QP needs xbar[2][t] and nothing else of the rollout).
template <int MODEL>
__device__ __forceinline__ double yaw_rate(const mpcx_variant &v, const double (&u)[4])
{
    if (MODEL == MPCX_MODEL_DIFF_DRIVE) return u[1];
    double sf, cf, sr, cr;
    sincos_fast(u[1], sf, cf);
    sincos_fast(u[3], sr, cr);
    return (sf * u[0] - sr * u[2]) / v.wb;
}

// ------------------------------------------------------------------------------------
// bounds of the two free controls per stage (uniform parts + per-instance first stage)
// ------------------------------------------------------------------------------------
struct FreeBounds {
    double b[2];     // |w_c| <= b[c] for stages >= 1 (and stage 0 before the window)
    double rr[2];    // |w_c[t+1] - w_c[t]| <= rr[c]
    double lo0[2], hi0[2];
    bool box[2], box0[2], rate[2];   // which of them are finite (uniform over the batch)
    bool feasible;
};

__device__ __forceinline__ double dmin(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ double dmax(double a, double b) { return __builtin_fmax(a, b); }

// uref0: first column of uref (all NU rows; unused rows ignored).
__device__ __forceinline__ void free_bounds(const mpcx_variant &v, const double (&u0)[4], FreeBounds &fb)
{
    const double *um = v.umax, *dm = v.dumax, *fs = v.first_step;
    double wl[2], wh[2];
    bool fsfin[2];
    fb.feasible = true;
    if (v.eq_mode == MPCX_EQ_NONE) {
        fb.b[0] = um[0]; fb.b[1] = um[1];
        fb.rr[0] = dm[0]; fb.rr[1] = dm[1];
        wl[0] = u0[0] - fs[0]; wh[0] = u0[0] + fs[0];
        wl[1] = u0[1] - fs[1]; wh[1] = u0[1] + fs[1];
        fsfin[0] = finite_d(fs[0]); fsfin[1] = finite_d(fs[1]);
    } else if (v.eq_mode == MPCX_EQ_FIXED_STEER) {
        fb.b[0] = um[0]; fb.b[1] = um[2];
        fb.rr[0] = dm[0]; fb.rr[1] = dm[2];
        wl[0] = u0[0] - fs[0]; wh[0] = u0[0] + fs[0];
        wl[1] = u0[2] - fs[2]; wh[1] = u0[2] + fs[2];
        fsfin[0] = finite_d(fs[0]); fsfin[1] = finite_d(fs[2]);
        if (fabs(v.steer_fixed) > um[1] || fabs(u0[1] - v.steer_fixed) > fs[1]) fb.feasible = false;
        if (fabs(v.steer_fixed) > um[3] || fabs(u0[3] - v.steer_fixed) > fs[3]) fb.feasible = false;
    } else {
        const double sg = v.eq_mode == MPCX_EQ_ACKERMANN ? -1.0 : 1.0;
        fb.b[0] = dmin(um[0], um[2]); fb.b[1] = dmin(um[1], um[3]);
        fb.rr[0] = dmin(dm[0], dm[2]); fb.rr[1] = dmin(dm[1], dm[3]);
        wl[0] = dmax(u0[0] - fs[0], u0[2] - fs[2]); wh[0] = dmin(u0[0] + fs[0], u0[2] + fs[2]);
        wl[1] = dmax(u0[1] - fs[1], sg * u0[3] - fs[3]); wh[1] = dmin(u0[1] + fs[1], sg * u0[3] + fs[3]);
        fsfin[0] = finite_d(fs[0]) || finite_d(fs[2]);
        fsfin[1] = finite_d(fs[1]) || finite_d(fs[3]);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        fb.box[c] = finite_d(fb.b[c]);
        fb.rate[c] = finite_d(fb.rr[c]);
        fb.box0[c] = fb.box[c] || fsfin[c];
        fb.lo0[c] = dmax(-fb.b[c], wl[c]);
        fb.hi0[c] = dmin(fb.b[c], wh[c]);
        if (!(fb.lo0[c] <= fb.hi0[c])) fb.feasible = false;
        // a (near) zero-width window has no interior: widen it to kMinWidth, far below the
        // 1e-3 the controls are compared at
        if (fb.box0[c] && fb.hi0[c] - fb.lo0[c] < kMinWidth) {
            const double mid = 0.5 * (fb.lo0[c] + fb.hi0[c]);
            fb.lo0[c] = mid - 0.5 * kMinWidth;
            fb.hi0[c] = mid + 0.5 * kMinWidth;
        }
    }
}

// ------------------------------------------------------------------------------------
// storage policies
// ------------------------------------------------------------------------------------
template <int T_>
struct RegStore {
    static constexpr int T = T_, N = 2 * T_, NT = N * (N + 1) / 2;
    static constexpr int kLdsDoubles = NT + N;   // per lane: H (packed lower) then g
    double K_[NT];
    double G_[3 * N], g_[N], w_[N], lamU_[N], lamL_[N], nuU_[N], nuL_[N];
    double dwa_[N], dw_[N];
    // reciprocal slacks, later dual steps, in the polish 0/1 flags: single precision is enough
    // (every use is consistent with the rounded value, so it only perturbs the Newton step)
    // and it halves their registers
    typedef float aux_t;
    aux_t isu_[N], isl_[N], iru_[N], irl_[N];
    double theta_[T], uref_[4 * T];   // only live in the chained (mpc_sim) instantiation
    typedef __attribute__((address_space(3))) double lds_double;
    lds_double *Hs;                   // LDS, [NT + N][64], already offset by the lane
    __device__ __forceinline__ int horizon() const { return T; }
    __device__ __forceinline__ static int tri(int i, int j) { return i * (i + 1) / 2 + j; }
    // assembly accumulates H in the registers K will use, g in registers, then parks both in LDS
    __device__ __forceinline__ double &Hacc(int i, int j) { return K_[tri(i, j)]; }
    __device__ __forceinline__ double &gacc(int i) { return g_[i]; }
    __device__ __forceinline__ void finish_assembly()
    {
#pragma unroll
        for (int e = 0; e < NT; ++e) Hs[e * kWave] = K_[e];
#pragma unroll
        for (int i = 0; i < N; ++i) Hs[(NT + i) * kWave] = g_[i];
    }
    // volatile: H and g must be re-read from LDS where they are used; a plain load is loop
    // invariant, gets hoisted out of the solver loop and takes 130 registers with it
    __device__ __forceinline__ double H(int i, int j) const { return ((const volatile lds_double *)Hs)[tri(i, j) * kWave]; }
    __device__ __forceinline__ double g(int i) const { return ((const volatile lds_double *)Hs)[(NT + i) * kWave]; }
    __device__ __forceinline__ double &K(int i, int j) { return K_[tri(i, j)]; }
    __device__ __forceinline__ double &G(int r, int j) { return G_[r * N + j]; }
    __device__ __forceinline__ double &w(int i) { return w_[i]; }
    __device__ __forceinline__ double &lamU(int i) { return lamU_[i]; }
    __device__ __forceinline__ double &lamL(int i) { return lamL_[i]; }
    __device__ __forceinline__ double &nuU(int i) { return nuU_[i]; }
    __device__ __forceinline__ double &nuL(int i) { return nuL_[i]; }
    // hw (= H w + g) and dwa (predictor step) share storage: the predictor solve runs in place
    __device__ __forceinline__ double &hw(int i) { return dwa_[i]; }
    __device__ __forceinline__ double &dwa(int i) { return dwa_[i]; }
    __device__ __forceinline__ double &dw(int i) { return dw_[i]; }
    __device__ __forceinline__ aux_t &isu(int i) { return isu_[i]; }
    __device__ __forceinline__ aux_t &isl(int i) { return isl_[i]; }
    __device__ __forceinline__ aux_t &iru(int i) { return iru_[i]; }
    __device__ __forceinline__ aux_t &irl(int i) { return irl_[i]; }
    __device__ __forceinline__ double &theta(int t) { return theta_[t]; }
    __device__ __forceinline__ double &uref(int r, int t) { return uref_[r * T + t]; }
};

// ------------------------------------------------------------------------------------
// linearisation point: where the controls uref[:, t] and the yaw xbar[2][t] come from.
//   STORED = false  straight from the caller's buffers (or the cold-start constants), yaw
//                   either read from xbar or rolled forward from x0 with predict_motion's
//                   yaw update; nothing is kept in registers, the epilogue re-reads it;
//   STORED = true   from the store's uref/theta arrays (chained solves rewrite uref between
//                   passes; the memory-resident path always uses this).
// ------------------------------------------------------------------------------------
struct LinPoint {
    const double *u;      // [NU][T] of this instance, or nullptr -> cold start
    const double *yaw;    // xbar[2][0..T-1] of this instance, or nullptr -> roll from theta0
    const double *cold;   // v.cold_start
    double theta0;
};

// One stage: controls and yaw -> reduced dynamics  x+ = A x + bt w_t + ct  with
// A = I + a02 e0 e2' + a12 e1 e2'.  `th` carries the rolled yaw from stage to stage.
template <int MODEL, bool STORED, class S>
__device__ __forceinline__ void stage_dynamics(S &s, const mpcx_variant &v, const LinPoint &lp, int t,
                                               double &th, double &a02, double &a12, double (&bt)[3][2],
                                               double (&ct)[3])
{
    constexpr int NU = MODEL == MPCX_MODEL_DIFF_DRIVE ? 2 : 4;
    const int T = s.horizon();
    double u[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < NU; ++r) u[r] = STORED ? s.uref(r, t) : (lp.u ? lp.u[r * T + t] : lp.cold[r]);
    if (STORED) th = s.theta(t);
    else if (lp.yaw) th = lp.yaw[t];
    double Bm[3][4], C[3], omega;
    linearize<MODEL>(v, th, u, a02, a12, Bm, C, omega);
    th = th + omega * v.dt;   // used only when the yaw is rolled
    reduce_controls(v, Bm, C, bt, ct);
}

// ------------------------------------------------------------------------------------
// condensed QP assembly
//   x_t = p_t + G_t w,  p_{t+1} = A_t p_t + c_t,  G_{t+1} = A_t G_t with block t = B_t E
//   H = 2 (Rw + D^T Rdw D + sum_t G_t^T Q_t G_t),  g = 2 sum_t G_t^T Q_t (p_t - xref_t)
// Stage weights: none at t = 0 (`if t != 0`, mpc_diff_drive.py:181), Q * growth^t for
// t = 1..T-1 (mpc_dsr.py:349), Qf at t = T.
// ------------------------------------------------------------------------------------
template <int MODEL, bool STORED, class S>
__device__ __forceinline__ void assemble(S &s, const mpcx_variant &v, const LinPoint &lp,
                                         const double (&x0)[3], const double *__restrict__ xref)
{
    const int T = s.horizon(), N = 2 * T, W = T + 1;
    double rw[2], rdw[2];
    if (v.eq_mode == MPCX_EQ_NONE) {
        rw[0] = v.R[0]; rw[1] = v.R[1]; rdw[0] = v.Rd[0]; rdw[1] = v.Rd[1];
    } else if (v.eq_mode == MPCX_EQ_FIXED_STEER) {
        rw[0] = v.R[0]; rw[1] = v.R[2]; rdw[0] = v.Rd[0]; rdw[1] = v.Rd[2];
    } else {
        rw[0] = v.R[0] + v.R[2]; rw[1] = v.R[1] + v.R[3];
        rdw[0] = v.Rd[0] + v.Rd[2]; rdw[1] = v.Rd[1] + v.Rd[3];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) s.Hacc(i, j) = 0.0;
        s.gacc(i) = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) s.G(r, i) = 0.0;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int c = i & 1, t = i >> 1;
        double d = 2.0 * rw[c];
        if (t > 0) d += 2.0 * rdw[c];
        if (t + 1 < T) d += 2.0 * rdw[c];
        s.Hacc(i, i) = d;
        if (t > 0) s.Hacc(i, i - 2) = -2.0 * rdw[c];
    }
    double p[3] = {x0[0], x0[1], x0[2]};
    double qg = 1.0;
    double th = lp.theta0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        double a02, a12, bt[3][2], ct[3];
        stage_dynamics<MODEL, STORED>(s, v, lp, t, th, a02, a12, bt, ct);
        const double pn0 = p[0] + a02 * p[2] + ct[0];
        const double pn1 = p[1] + a12 * p[2] + ct[1];
        const double pn2 = p[2] + ct[2];
        p[0] = pn0; p[1] = pn1; p[2] = pn2;
#pragma unroll
        for (int j = 0; j < 2 * t; ++j) {
            const double g2 = s.G(2, j);
            s.G(0, j) = fma(a02, g2, s.G(0, j));
            s.G(1, j) = fma(a12, g2, s.G(1, j));
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) { s.G(r, 2 * t) = bt[r][0]; s.G(r, 2 * t + 1) = bt[r][1]; }
        const int ts = t + 1;
        double Qs[3];
        if (ts < T) {
            qg *= v.q_growth;
#pragma unroll
            for (int r = 0; r < 3; ++r) Qs[r] = v.Q[r] * qg;
        } else {
#pragma unroll
            for (int r = 0; r < 3; ++r) Qs[r] = v.Qf[r];
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double e = p[r] - xref[r * W + ts];
            const double q2 = 2.0 * Qs[r];
#pragma unroll
            for (int i = 0; i < 2 * ts; ++i) {
                const double gi = q2 * s.G(r, i);
                s.gacc(i) = fma(gi, e, s.gacc(i));
#pragma unroll
                for (int j = 0; j <= i; ++j) s.Hacc(i, j) = fma(gi, s.G(r, j), s.Hacc(i, j));
            }
        }
    }
    s.finish_assembly();
}

// Reciprocal of a slack, already rounded to the precision it is stored in, so that the
// barrier Hessian and every later use see exactly the same number.
template <class S>
__device__ __forceinline__ double slack_recip(double x)
{
    if (sizeof(typename S::aux_t) == sizeof(float)) return (double)(float)rcp_raw(x);
    return rcp_nr(x);
}

// Cholesky K = L L' in place (right-looking); the diagonal is left holding 1 / L_ii.
template <class S>
__device__ __forceinline__ void chol_factor(S &s)
{
    const int N = 2 * s.horizon();
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double r = rsqrt_nr(s.K(j, j));
        s.K(j, j) = r;
#pragma unroll
        for (int i = j + 1; i < N; ++i) s.K(i, j) *= r;
#pragma unroll
        for (int i = j + 1; i < N; ++i) {
            const double lij = s.K(i, j);
#pragma unroll
            for (int k = j + 1; k <= i; ++k) s.K(i, k) = fma(-lij, s.K(k, j), s.K(i, k));
        }
    }
}

// vec <- K^-1 vec with the factor above; vec(i) returns a reference.
template <class S, class V>
__device__ __forceinline__ void chol_apply(S &s, V &&vec)
{
    const int N = 2 * s.horizon();
#pragma unroll
    for (int i = 0; i < N; ++i) {
        double a = vec(i);
#pragma unroll
        for (int k = 0; k < i; ++k) a = fma(-s.K(i, k), vec(k), a);
        vec(i) = a * s.K(i, i);
    }
#pragma unroll
    for (int ii = 0; ii < N; ++ii) {
        const int i = N - 1 - ii;
        double a = vec(i);
#pragma unroll
        for (int k = i + 1; k < N; ++k) a = fma(-s.K(k, i), vec(k), a);
        vec(i) = a * s.K(i, i);
    }
}

// hw <- H w + g (symmetric packed H, one pass); optionally K <- H on the way.
template <bool COPY_K, class S>
__device__ __forceinline__ void hess_times_w(S &s)
{
    const int N = 2 * s.horizon();
#pragma unroll
    for (int i = 0; i < N; ++i) s.hw(i) = s.g(i);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double wi = s.w(i);
        double acc = s.hw(i);
#pragma unroll
        for (int j = 0; j < i; ++j) {
            const double h = s.H(i, j);
            if (COPY_K) s.K(i, j) = h;
            acc = fma(h, s.w(j), acc);
            s.hw(j) = fma(h, wi, s.hw(j));
        }
        const double hd = s.H(i, i);
        if (COPY_K) s.K(i, i) = hd;
        s.hw(i) = fma(hd, wi, acc);
    }
}

// ------------------------------------------------------------------------------------
// interior point solve of  min 1/2 w'Hw + g'w,  lo <= w <= hi,  |w[i+2]-w[i]| <= rr[i&1]
// ------------------------------------------------------------------------------------
template <bool ALL, class S>
__device__ __forceinline__ void ipm_solve(S &s, const FreeBounds &fb, int max_iter, bool skip,
                                          int &status, int &iters)
{
    const int T = s.horizon(), N = 2 * T, NR = N - 2;
    const bool bx[2] = {ALL || fb.box[0], ALL || fb.box[1]};
    const bool bx0[2] = {ALL || fb.box0[0], ALL || fb.box0[1]};
    const bool rt[2] = {ALL || fb.rate[0], ALL || fb.rate[1]};
    auto hasbox = [&](int i) { return (i >> 1) == 0 ? bx0[i & 1] : bx[i & 1]; };
    auto hi_of = [&](int i) { return (i >> 1) == 0 ? fb.hi0[i & 1] : fb.b[i & 1]; };
    auto lo_of = [&](int i) { return (i >> 1) == 0 ? fb.lo0[i & 1] : -fb.b[i & 1]; };

    // strictly feasible start: every stage at the middle of the first-stage window
    double wc[2];
    int m = 0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        wc[c] = bx0[c] ? 0.5 * (fb.lo0[c] + fb.hi0[c]) : 0.0;
        m += (bx0[c] ? 2 : 0) + (bx[c] ? 2 * (T - 1) : 0) + (rt[c] ? 2 * (T - 1) : 0);
    }
    const double inv_m = m > 0 ? 1.0 / (double)m : 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double wi = wc[i & 1];
        s.w(i) = wi;
        if (hasbox(i)) {
            s.lamU(i) = kMu0 * rcp_nr(dmax(hi_of(i) - wi, kSlackMin));
            s.lamL(i) = kMu0 * rcp_nr(dmax(wi - lo_of(i), kSlackMin));
        } else {
            s.lamU(i) = 0.0; s.lamL(i) = 0.0;
        }
    }
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const double l0 = rt[i & 1] ? kMu0 * rcp_nr(fb.rr[i & 1]) : 0.0;
        s.nuU(i) = l0; s.nuL(i) = l0;
    }

    bool done = skip;
    int my_iters = 0;
    status = MPCX_ST_FAILED;
    for (int it = 0;; ++it) {
        hess_times_w<true>(s);   // K <- H and hw <- H w + g in one pass over the Hessian
        MPCX_PHASE("SLACKS");
        // slacks, barrier Hessian terms, C'lambda (kept in dw), complementarity
        double musum = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) s.dw(i) = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (hasbox(i)) {
                const double wi = s.w(i);
                const double su = dmax(hi_of(i) - wi, kSlackMin), sl = dmax(wi - lo_of(i), kSlackMin);
                const double iu = slack_recip<S>(su), il = slack_recip<S>(sl);
                const double lu = s.lamU(i), ll = s.lamL(i);
                s.isu(i) = (typename S::aux_t)(iu); s.isl(i) = (typename S::aux_t)(il);
                s.K(i, i) += fma(lu, iu, ll * il);
                s.dw(i) += lu - ll;
                musum = fma(su, lu, fma(sl, ll, musum));
            } else {
                s.isu(i) = (typename S::aux_t)(0.0); s.isl(i) = (typename S::aux_t)(0.0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (rt[i & 1]) {
                const double rr = fb.rr[i & 1];
                const double dv = s.w(i + 2) - s.w(i);
                const double ru = dmax(rr - dv, kSlackMin), rl = dmax(rr + dv, kSlackMin);
                const double iu = slack_recip<S>(ru), il = slack_recip<S>(rl);
                const double nu = s.nuU(i), nl = s.nuL(i);
                s.iru(i) = (typename S::aux_t)(iu); s.irl(i) = (typename S::aux_t)(il);
                const double d = fma(nu, iu, nl * il);
                s.K(i, i) += d;
                s.K(i + 2, i + 2) += d;
                s.K(i + 2, i) -= d;
                const double dn = nu - nl;
                s.dw(i + 2) += dn;
                s.dw(i) -= dn;
                musum = fma(ru, nu, fma(rl, nl, musum));
            } else {
                s.iru(i) = (typename S::aux_t)(0.0); s.irl(i) = (typename S::aux_t)(0.0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const double mu = musum * inv_m;
        double rdmax = 0.0, scale = 1.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double hwi = s.hw(i), gi = s.g(i), ci = s.dw(i);
            rdmax = dmax(rdmax, fabs(hwi + ci));
            scale = dmax(scale, dmax(fabs(hwi - gi), dmax(fabs(gi), fabs(ci))));
        }
        scale += 1.0;
        const bool conv = rdmax <= kTolD * scale && mu <= kTolMu;
        if (!done && conv) { done = true; status = MPCX_ST_OK; my_iters = it; }
        // (without any constraint the complementarity is zero by definition, not by rounding:
        // the iteration is then a plain Newton step)
        if (!done && ((inv_m > 0.0 && mu < kMuFloor) || it >= max_iter)) {
            // weakly active constraints make the tail converge linearly; once the barrier
            // parameter is at the rounding floor further steps only add noise
            done = true;
            my_iters = it;
            status = (rdmax <= kLooseD * scale && mu <= kLooseMu) ? MPCX_ST_INACCURATE : MPCX_ST_FAILED;
        }
        if (__all(done)) break;

        MPCX_PHASE("CHOL");
        chol_factor(s);
        auto chol_solve = [&](auto &&vec) { chol_apply(s, vec); };

        MPCX_PHASE("PRED");
        // predictor: (H + C'DC) dwa = -(Hw + g)
#pragma unroll
        for (int i = 0; i < N; ++i) s.dwa(i) = -s.hw(i);   // in place: hw and dwa share storage
        chol_solve([&](int i) -> double & { return s.dwa(i); });

        MPCX_PHASE("STATS");
        // x = (C dwa)_j / s_j ; primal ratio max x, dual ratio max (1 - x)
        double rp = 0.0, rdm = 1.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (hasbox(i)) {
                const double d = s.dwa(i);
                const double xu = d * s.isu(i), xl = -d * s.isl(i);
                rp = dmax(rp, dmax(xu, xl));
                rdm = dmax(rdm, dmax(1.0 - xu, 1.0 - xl));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (rt[i & 1]) {
                const double d = s.dwa(i + 2) - s.dwa(i);
                const double yu = d * s.iru(i), yl = -d * s.irl(i);
                rp = dmax(rp, dmax(yu, yl));
                rdm = dmax(rdm, dmax(1.0 - yu, 1.0 - yl));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const double apa = rp > 1.0 ? rcp_nr(rp) : 1.0;
        const double ada = rcp_nr(rdm);
        double muaff = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (hasbox(i)) {
                const double wi = opaque(s.w(i)), d = opaque(s.dwa(i));
                const double su = dmax(hi_of(i) - wi, kSlackMin), sl = dmax(wi - lo_of(i), kSlackMin);
                const double xu = d * s.isu(i), xl = -d * s.isl(i);
                muaff += (su * fma(-apa, xu, 1.0)) * (s.lamU(i) * fma(ada, xu - 1.0, 1.0));
                muaff += (sl * fma(-apa, xl, 1.0)) * (s.lamL(i) * fma(ada, xl - 1.0, 1.0));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (rt[i & 1]) {
                const double rr = fb.rr[i & 1];
                const double dv = opaque(s.w(i + 2) - s.w(i)), d = opaque(s.dwa(i + 2) - s.dwa(i));
                const double ru = dmax(rr - dv, kSlackMin), rl = dmax(rr + dv, kSlackMin);
                const double yu = d * s.iru(i), yl = -d * s.irl(i);
                muaff += (ru * fma(-apa, yu, 1.0)) * (s.nuU(i) * fma(ada, yu - 1.0, 1.0));
                muaff += (rl * fma(-apa, yl, 1.0)) * (s.nuL(i) * fma(ada, yl - 1.0, 1.0));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        muaff *= inv_m;
        const double ratio = muaff * rcp_nr(dmax(mu, 1e-300));
        const double sm = ratio * ratio * ratio * mu;   // sigma * mu

        MPCX_PHASE("CORR");
        // corrector as a correction to the predictor: K (dw - dwa) = -C' q,
        // q_j = (sigma mu - ds_aff dl_aff)/s_j = sigma mu / s_j + lambda_j x_j (x_j - 1)
#pragma unroll
        for (int i = 0; i < N; ++i) s.dw(i) = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (hasbox(i)) {
                const double d = opaque(s.dwa(i));
                const double iu = s.isu(i), il = s.isl(i);
                const double xu = d * iu, xl = -d * il;
                const double qu = fma(sm, iu, s.lamU(i) * xu * (xu - 1.0));
                const double ql = fma(sm, il, s.lamL(i) * xl * (xl - 1.0));
                s.dw(i) -= qu - ql;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (rt[i & 1]) {
                const double d = opaque(s.dwa(i + 2) - s.dwa(i));
                const double iu = s.iru(i), il = s.irl(i);
                const double yu = d * iu, yl = -d * il;
                const double qu = fma(sm, iu, s.nuU(i) * yu * (yu - 1.0));
                const double ql = fma(sm, il, s.nuL(i) * yl * (yl - 1.0));
                const double q = qu - ql;
                s.dw(i + 2) -= q;
                s.dw(i) += q;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        chol_solve([&](int i) -> double & { return s.dw(i); });
#pragma unroll
        for (int i = 0; i < N; ++i) s.dw(i) += s.dwa(i);

        MPCX_PHASE("FINAL");
        // dual steps (stored over the reciprocal slacks) and the two ratios
        double rp2 = 0.0, rd2 = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (hasbox(i)) {
                const double da = opaque(s.dwa(i)), d = s.dw(i);
                const double iu = s.isu(i), il = s.isl(i), lu = s.lamU(i), ll = s.lamL(i);
                const double xu = da * iu, xl = -da * il;
                const double qu = fma(sm, iu, lu * xu * (xu - 1.0));
                const double ql = fma(sm, il, ll * xl * (xl - 1.0));
                const double zu = d * iu, zl = -d * il;
                const double dlu = fma(lu, zu, qu - lu), dll = fma(ll, zl, ql - ll);
                rp2 = dmax(rp2, dmax(zu, zl));
                rd2 = dmax(rd2, dmax(-dlu * rcp_raw(lu), -dll * rcp_raw(ll)));
                s.isu(i) = (typename S::aux_t)(dlu); s.isl(i) = (typename S::aux_t)(dll);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (rt[i & 1]) {
                const double da = opaque(s.dwa(i + 2) - s.dwa(i)), d = s.dw(i + 2) - s.dw(i);
                const double iu = s.iru(i), il = s.irl(i), nu = s.nuU(i), nl = s.nuL(i);
                const double yu = da * iu, yl = -da * il;
                const double qu = fma(sm, iu, nu * yu * (yu - 1.0));
                const double ql = fma(sm, il, nl * yl * (yl - 1.0));
                const double zu = d * iu, zl = -d * il;
                const double dnu = fma(nu, zu, qu - nu), dnl = fma(nl, zl, ql - nl);
                rp2 = dmax(rp2, dmax(zu, zl));
                rd2 = dmax(rd2, dmax(-dnu * rcp_raw(nu), -dnl * rcp_raw(nl)));
                s.iru(i) = (typename S::aux_t)(dnu); s.irl(i) = (typename S::aux_t)(dnl);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const double eta = dmin(kEtaCap, dmax(0.99, 1.0 - mu));
        const double ap = rp2 > eta ? eta * rcp_nr(rp2) : 1.0;
        const double ad = rd2 > eta ? eta * rcp_nr(rd2) : 1.0;
        if (!done) {   // converged lanes keep their iterate while the rest of the wave finishes
#pragma unroll
            for (int i = 0; i < N; ++i) {
                s.w(i) = fma(ap, s.dw(i), s.w(i));
                if (hasbox(i)) {
                    s.lamU(i) = fma(ad, s.isu(i), s.lamU(i));
                    s.lamL(i) = fma(ad, s.isl(i), s.lamL(i));
                }
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                if (rt[i & 1]) {
                    s.nuU(i) = fma(ad, s.iru(i), s.nuU(i));
                    s.nuL(i) = fma(ad, s.irl(i), s.nuL(i));
                }
            }
        }
    }
    iters = my_iters;
}

// ------------------------------------------------------------------------------------
// Active-set polish.  The interior-point iterate is only ~sqrt(mu) accurate on weakly
// active constraints; here the constraints it marks active (lambda_j > s_j) are imposed as
// equalities and the equality-constrained QP is solved by two augmented-Lagrangian Newton
// steps (each shrinks the constraint residual by ~H/rho = 1e-5) that share one Cholesky factor of K = H + rho * A_act' A_act.  Residuals are formed
// with the true H, so the fixed point is the exact KKT point whatever the accuracy of the
// factor.  If the result violates a constraint that was left out, or gives a kept one a
// negative multiplier, the set is corrected and the round repeated (at most kPolishRounds;
// the wave leaves as soon as every lane is settled).  A lane takes the polished point only
// if it is feasible for every constraint, its multipliers are non-negative and it is
// stationary to rounding; otherwise it keeps the interior-point iterate.
// Returns true when accepted.
// ------------------------------------------------------------------------------------
constexpr int kPolishRounds = 3;

template <bool ALL, class S>
__device__ __forceinline__ bool polish(S &s, const FreeBounds &fb, bool skip)
{
    const int T = s.horizon(), N = 2 * T, NR = N - 2;
    const bool bx[2] = {ALL || fb.box[0], ALL || fb.box[1]};
    const bool bx0[2] = {ALL || fb.box0[0], ALL || fb.box0[1]};
    const bool rt[2] = {ALL || fb.rate[0], ALL || fb.rate[1]};
    auto hasbox = [&](int i) { return (i >> 1) == 0 ? bx0[i & 1] : bx[i & 1]; };
    auto hi_of = [&](int i) { return (i >> 1) == 0 ? fb.hi0[i & 1] : fb.b[i & 1]; };
    auto lo_of = [&](int i) { return (i >> 1) == 0 ? fb.lo0[i & 1] : -fb.b[i & 1]; };

    double hmax = 0.0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        hmax = dmax(hmax, s.H(i, i));
        s.G(0, i) = s.w(i);   // what the lane falls back to: the interior-point iterate
    }
    const double rho = 1e5 * hmax;
    bool accepted = false, settled = skip;
    double scale = 1.0;

#pragma unroll 1
    for (int round = 0; round < kPolishRounds; ++round) {
        // active flags, kept as 0/1 in the reciprocal-slack arrays.  First round: lambda_j > s_j.
        // Later rounds: keep a constraint unless its multiplier went negative, add every
        // violated one.
        const double ntol = -1e-9 * scale;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            double au = 0.0, al = 0.0;
            if (hasbox(i)) {
                const double wi = s.w(i), su = hi_of(i) - wi, sl = wi - lo_of(i);
                if (round == 0) {
                    au = s.lamU(i) > su ? 1.0 : 0.0;
                    al = s.lamL(i) > sl ? 1.0 : 0.0;
                } else {
                    au = ((s.isu(i) != 0.0 && s.lamU(i) >= ntol) || su < -1e-12) ? 1.0 : 0.0;
                    al = ((s.isl(i) != 0.0 && s.lamL(i) >= ntol) || sl < -1e-12) ? 1.0 : 0.0;
                }
            }
            s.isu(i) = (typename S::aux_t)(au); s.isl(i) = (typename S::aux_t)(al);
            s.lamU(i) = au * dmax(s.lamU(i), 0.0);
            s.lamL(i) = al * dmax(s.lamL(i), 0.0);
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            double au = 0.0, al = 0.0;
            if (rt[i & 1]) {
                const double rr = fb.rr[i & 1], dv = s.w(i + 2) - s.w(i);
                if (round == 0) {
                    au = s.nuU(i) > rr - dv ? 1.0 : 0.0;
                    al = s.nuL(i) > rr + dv ? 1.0 : 0.0;
                } else {
                    au = ((s.iru(i) != 0.0 && s.nuU(i) >= ntol) || rr - dv < -1e-12) ? 1.0 : 0.0;
                    al = ((s.irl(i) != 0.0 && s.nuL(i) >= ntol) || rr + dv < -1e-12) ? 1.0 : 0.0;
                }
            }
            s.iru(i) = (typename S::aux_t)(au); s.irl(i) = (typename S::aux_t)(al);
            s.nuU(i) = au * dmax(s.nuU(i), 0.0);
            s.nuL(i) = al * dmax(s.nuL(i), 0.0);
        }

        // K = H + rho A'A on the active rows, one factorisation per round
#pragma unroll
        for (int i = 0; i < N; ++i) {
#pragma unroll
            for (int j = 0; j < i; ++j) s.K(i, j) = s.H(i, j);
            s.K(i, i) = fma(rho, s.isu(i) + s.isl(i), s.H(i, i));
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const double d = rho * (s.iru(i) + s.irl(i));
            s.K(i, i) += d;
            s.K(i + 2, i + 2) += d;
            s.K(i + 2, i) -= d;
        }
        chol_factor(s);

        double stat = 0.0;
#pragma unroll 1
        for (int pass = 0; pass < 3; ++pass) {
            hess_times_w<false>(s);
            // multiplier update with the constraint values at the current point (not on the
            // first pass), then gradient of the Lagrangian (dw) and of the augmented one (hw)
#pragma unroll
            for (int i = 0; i < N; ++i) s.dw(i) = s.hw(i);
#pragma unroll
            for (int i = 0; i < N; ++i) {
                if (hasbox(i)) {
                    const double wi = s.w(i);
                    const double cu = s.isu(i) * (wi - hi_of(i)), cl = s.isl(i) * (lo_of(i) - wi);
                    if (pass > 0) { s.lamU(i) = fma(rho, cu, s.lamU(i)); s.lamL(i) = fma(rho, cl, s.lamL(i)); }
                    const double t = s.lamU(i) - s.lamL(i);
                    s.dw(i) += t;
                    s.hw(i) += t + rho * (cu - cl);
                }
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                if (rt[i & 1]) {
                    const double rr = fb.rr[i & 1], dv = s.w(i + 2) - s.w(i);
                    const double cu = s.iru(i) * (dv - rr), cl = s.irl(i) * (-dv - rr);
                    if (pass > 0) { s.nuU(i) = fma(rho, cu, s.nuU(i)); s.nuL(i) = fma(rho, cl, s.nuL(i)); }
                    const double t = s.nuU(i) - s.nuL(i);
                    s.dw(i + 2) += t; s.dw(i) -= t;
                    const double ta = t + rho * (cu - cl);
                    s.hw(i + 2) += ta; s.hw(i) -= ta;
                }
            }
            stat = 0.0; scale = 1.0;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                stat = dmax(stat, fabs(s.dw(i)));
                scale = dmax(scale, dmax(fabs(s.g(i)), fabs(s.dw(i) - s.hw(i))));
            }
            scale += 1.0;
            if (pass == 2) break;
            // Newton step on the augmented Lagrangian
#pragma unroll
            for (int i = 0; i < N; ++i) s.hw(i) = -s.hw(i);
            chol_apply(s, [&](int i) -> double & { return s.hw(i); });
            if (!settled) {
#pragma unroll
                for (int i = 0; i < N; ++i) s.w(i) += s.hw(i);
            }
        }
        // acceptance: stationary, every constraint satisfied, multipliers non-negative: the KKT
        // conditions of a strictly convex QP, so an accepted point is its minimiser wherever
        // the interior-point iterate stood.  max() drops NaN operands; the running sum does not.
        double viol = 0.0, neg = 0.0, chk = 0.0, cact = 0.0;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double wi = s.w(i);
            chk += wi + s.dw(i);
            if (hasbox(i)) {
                viol = dmax(viol, dmax(wi - hi_of(i), lo_of(i) - wi));
                cact = dmax(cact, dmax(s.isu(i) * fabs(wi - hi_of(i)), s.isl(i) * fabs(lo_of(i) - wi)));
                neg = dmax(neg, dmax(-s.lamU(i), -s.lamL(i)));
            }
        }
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (rt[i & 1]) {
                const double rr = fb.rr[i & 1], dv = s.w(i + 2) - s.w(i);
                viol = dmax(viol, dmax(dv - rr, -dv - rr));
                cact = dmax(cact, dmax(s.iru(i) * fabs(dv - rr), s.irl(i) * fabs(-dv - rr)));
                neg = dmax(neg, dmax(-s.nuU(i), -s.nuL(i)));
            }
        }
        if (!settled) {
            const bool sane = finite_d(chk) && stat <= 1e-10 * scale;
            if (sane && viol <= 1e-12 && cact <= 1e-12 && neg <= 1e-9 * scale) { accepted = true; settled = true; }
            else if (!sane) settled = true;   // not an active-set problem: keep the iterate
        }
        if (__all(settled)) break;
    }
    if (!accepted) {
#pragma unroll
        for (int i = 0; i < N; ++i) s.w(i) = s.G(0, i);
    }
    return accepted;
}

}  // namespace mpcx
