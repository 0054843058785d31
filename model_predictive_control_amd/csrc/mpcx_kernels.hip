// Kernels and C ABI of libmpcx.so (gfx950 only).  See include/mpcx.h for the contract.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>

#include "mpcx.h"
#include "mpcx_device.h"
#include "mpcx_stage.h"

namespace mpcx {
// Diagnostic builds (-DMPCX_STAMPS) record the shader clock at phase boundaries of the pair
// kernel into a buffer nothing else reads; product builds contain no stamp.
#ifdef MPCX_STAMPS
constexpr int kStampWaves = 4096, kStampSlots = 16;
__device__ unsigned long long g_stamps[kStampWaves * kStampSlots];
#define MPCX_STAMP(k)                                                                         \
    do {                                                                                      \
        if (threadIdx.x == 0 && blockIdx.x < kStampWaves)                                     \
            g_stamps[blockIdx.x * kStampSlots + (k)] = __builtin_amdgcn_s_memtime();          \
    } while (0)
#define MPCX_STAMP_CLEAR()                                                                    \
    do {                                                                                      \
        if (threadIdx.x < kStampSlots && blockIdx.x < kStampWaves)                            \
            g_stamps[blockIdx.x * kStampSlots + threadIdx.x] = 0;                             \
    } while (0)
#else
#define MPCX_STAMP(k)
#define MPCX_STAMP_CLEAR()
#endif


// ------------------------------------------------------------------------------------
// solve: one instance per lane
// ------------------------------------------------------------------------------------
struct SolveArgs {
    mpcx_variant v;
    int64_t B;
    const double *x0, *xref, *xbar, *uin;
    const uint8_t *has_prev;
    double *u_out, *x_out, *obj_out;
    int32_t *status_out, *iters_out;
    int32_t linear_mode;   // 1: (xbar, uref) given -> one QP;  0: cold start + rollout (+ chain)
    int32_t max_iter;
    const int32_t *range;  // device memory, {first, end}: only these instances are solved (nullptr: all);
                           // with vtab: n_modes + 1 bounds of the runs of a batch sorted by mode
    const mpcx_variant *vtab;   // device memory: the variant of every run (nullptr: one variant, v)
    int32_t n_modes;
};

// STORED: the linearisation point lives in the store (chained mpc_sim solves, and the
// memory-resident path); otherwise it is read from the caller's buffers when needed.
template <int MODEL, bool ALL, bool STORED, class S>
__device__ __forceinline__ void run_instance(S &s, const SolveArgs &a, int64_t b, bool live)
{
    const mpcx_variant &v = a.v;
    constexpr int NU = MODEL == MPCX_MODEL_DIFF_DRIVE ? 2 : 4;
    const int T = s.horizon(), W = T + 1;
    const double *__restrict__ xr = a.xref + b * 3 * W;
    double x0[3] = {a.x0[b * 3], a.x0[b * 3 + 1], a.x0[b * 3 + 2]};
    bool finite = finite_d(x0[0]) && finite_d(x0[1]) && finite_d(x0[2]);
#pragma unroll
    for (int i = 0; i < 3 * W; ++i) finite = finite && finite_d(xr[i]);

    const double *uin = a.uin ? a.uin + b * NU * T : nullptr;
    const bool warm = a.linear_mode || (uin && (!a.has_prev || a.has_prev[b]));
    const int chain = a.linear_mode ? 1 : (v.chain > 1 ? (v.chain < T ? v.chain : T) : 1);
    LinPoint lp;
    lp.u = warm ? uin : nullptr;
    lp.cold = v.cold_start;
    lp.yaw = nullptr;
    lp.theta0 = x0[2];
    if (a.linear_mode) {
        const double *xb = a.xbar + b * 3 * W;
#pragma unroll
        for (int i = 0; i < 3 * W; ++i) finite = finite && finite_d(xb[i]);
        lp.yaw = xb + 2 * W;
    }
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < NU; ++r) finite = finite && finite_d(lp.u ? lp.u[r * T + t] : lp.cold[r]);

    if (STORED) {
        // xbar[2][t] of the incoming controls (predict_motion), kept fixed over chained solves
        // (mpc_sim.py:294); uref starts as the incoming controls, or, when chaining, as column 0
        // of them with the rest zero (mpc_sim.py:292,295-299)
        double th = x0[2];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            double u[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < NU; ++r) u[r] = lp.u ? lp.u[r * T + t] : lp.cold[r];
            s.theta(t) = lp.yaw ? lp.yaw[t] : th;
            th = th + yaw_rate<MODEL>(v, u) * v.dt;
#pragma unroll
            for (int r = 0; r < 4; ++r) s.uref(r, t) = (chain > 1 && t > 0) ? 0.0 : u[r];
        }
    }

    int status = MPCX_ST_OK, iters = 0;
    bool skip = !finite;
    if (!finite) {
        status = MPCX_ST_NONFINITE;
        x0[0] = x0[1] = x0[2] = 0.0;
        lp.theta0 = 0.0;
    }
    const double sg = v.eq_mode == MPCX_EQ_ACKERMANN ? -1.0 : 1.0;
    auto controls = [&](int t, double (&u)[4]) {
        const double w0 = s.w(2 * t), w1 = s.w(2 * t + 1);
        if (v.eq_mode == MPCX_EQ_NONE) { u[0] = w0; u[1] = w1; u[2] = 0.0; u[3] = 0.0; }
        else if (v.eq_mode == MPCX_EQ_FIXED_STEER) { u[0] = w0; u[1] = v.steer_fixed; u[2] = w1; u[3] = v.steer_fixed; }
        else { u[0] = w0; u[1] = w1; u[2] = w0; u[3] = sg * w1; }
    };

    bool inaccurate = false;   // an inaccurate link makes the whole chain's answer inaccurate
    for (int pass = 0; pass < (STORED ? chain : 1); ++pass) {
        if (STORED && pass > 0) {   // column `pass` of uref <- previous answer (mpc_sim.py:296-299 after :302)
#pragma unroll
            for (int t = 1; t < T; ++t) {   // static indices under a runtime predicate: arrays stay in registers
                if (t == pass) {
                    double u[4];
                    controls(t, u);
#pragma unroll
                    for (int r = 0; r < 4; ++r) s.uref(r, t) = u[r];
                }
            }
        }
        assemble<MODEL, STORED>(s, v, lp, x0, xr);
        double u0[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int r = 0; r < NU; ++r) u0[r] = STORED ? s.uref(r, 0) : (lp.u ? lp.u[r * T] : lp.cold[r]);
        FreeBounds fb;
        free_bounds(v, u0, fb);
        if (!fb.feasible && !skip) { skip = true; status = MPCX_ST_INFEASIBLE; }
        int st = MPCX_ST_OK, it = 0;
        ipm_solve<ALL>(s, fb, a.max_iter, skip, st, it);
        if (!skip) {
            iters += it;
            status = st;
        }
        // the polish runs even after an interior-point solve that stalled short of its
        // tolerances: it certifies its own answer, and a stalled iterate usually has the set
        const bool polished = polish<ALL>(s, fb, skip);
        if (!skip && polished) status = MPCX_ST_OK;
        if (!skip && status == MPCX_ST_INACCURATE) inaccurate = true;
        if (!skip && status == MPCX_ST_FAILED) skip = true;
    }
    if (status == MPCX_ST_OK && inaccurate) status = MPCX_ST_INACCURATE;

    // controls, predicted states (the linearised dynamics re-evaluated stage by stage),
    // objective = the cost expression of mpc_diff_drive.py:178-193 plus the difference term
    // of mpc.py:152-153 at the returned point
    const bool bad = status == MPCX_ST_NONFINITE || status == MPCX_ST_INFEASIBLE;
    const double nan = __builtin_nan("");
    double J = 0.0, qg = 1.0;
    double x[3] = {x0[0], x0[1], x0[2]};
    double up[4] = {0, 0, 0, 0};
    double *uo = a.u_out + b * NU * T, *xo = a.x_out + b * 3 * W;
    if (live) {
#pragma unroll
        for (int r = 0; r < 3; ++r) xo[r * W] = bad ? nan : x[r];
    }
    double th = lp.theta0;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        double u[4];
        controls(t, u);
        const double w0 = s.w(2 * t), w1 = s.w(2 * t + 1);
#pragma unroll
        for (int r = 0; r < NU; ++r) {
            J = fma(v.R[r] * u[r], u[r], J);
            if (t > 0) { const double d = u[r] - up[r]; J = fma(v.Rd[r] * d, d, J); }
            up[r] = u[r];
            if (live) uo[r * T + t] = bad ? nan : u[r];
        }
        double a02, a12, bt[3][2], ct[3];
        stage_dynamics<MODEL, STORED>(s, v, lp, t, th, a02, a12, bt, ct);
        const double n0 = x[0] + a02 * x[2] + bt[0][0] * w0 + bt[0][1] * w1 + ct[0];
        const double n1 = x[1] + a12 * x[2] + bt[1][0] * w0 + bt[1][1] * w1 + ct[1];
        const double n2 = x[2] + bt[2][0] * w0 + bt[2][1] * w1 + ct[2];
        x[0] = n0; x[1] = n1; x[2] = n2;
        const int ts = t + 1;
        if (ts < T) qg *= v.q_growth;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double e = xr[r * W + ts] - x[r];
            const double q = ts < T ? v.Q[r] * qg : v.Qf[r];
            J = fma(q * e, e, J);
            if (live) xo[r * W + ts] = bad ? nan : x[r];
        }
    }
    if (live) {
        a.obj_out[b] = bad ? nan : J;
        a.status_out[b] = status;
        if (a.iters_out) a.iters_out[b] = iters;
    }
}

// One lane per instance with the dense condensed QP (the first-generation kernel): built only
// into the test-hook library (-DMPCX_TEST_HOOKS), where the cross-kernel tests route to it.
#ifdef MPCX_TEST_HOOKS
template <int MODEL, int T, bool CHAIN>
__global__ __launch_bounds__(kWave) void k_solve_reg(SolveArgs a)
{
    __shared__ double Hs[RegStore<T>::kLdsDoubles * kWave];
    RegStore<T> s;
    s.Hs = (typename RegStore<T>::lds_double *)Hs + threadIdx.x;
    int64_t b = (int64_t)blockIdx.x * kWave + threadIdx.x;
    const bool live = b < a.B;
    if (!live) b = a.B - 1;   // idle lanes redo the last instance so wave votes stay uniform
    run_instance<MODEL, true, CHAIN>(s, a, b, live);
}
#endif

// Two lanes per instance (T = 5, every bound present, single QP per call).
#ifndef MPCX_PAIR_OCC
#define MPCX_PAIR_OCC 1
#endif
#ifndef MPCX_PAIR_K
#define MPCX_PAIR_K 5
#endif
constexpr int kPairIpmIters = MPCX_PAIR_K;   // interior-point iterations before the active-set rounds
#ifndef MPCX_PAIR_MORE
#define MPCX_PAIR_MORE 2
#endif
constexpr int kPairMoreIters = MPCX_PAIR_MORE;   // ... and the further ones of the second attempt
#ifndef MPCX_PAIR_K10
#define MPCX_PAIR_K10 6
#endif
constexpr int kPairIpmIters10 = MPCX_PAIR_K10;   // ... at T = 10 (measured on 65,536 mpc.py instances: 4: 1.97 ms, 5: 1.64, 6: 1.47, 7: 1.61)
#ifndef MPCX_PAIR_R0
#define MPCX_PAIR_R0 6
#endif
constexpr int kPairRounds0 = MPCX_PAIR_R0;              // active-set rounds after them
constexpr int kPairRounds1 = 4;              // ... and after a converged interior-point solve
}  // namespace mpcx

#include "mpcx_wave.h"

#define MPCX_PAIR_TT 5
#define MPCX_PAIR_NS pair5
#define MPCX_PAIR_KERNEL k_solve_pair5
#include "mpcx_pair.h"
#include "mpcx_pair_kernel.h"
#undef MPCX_PAIR_TT
#undef MPCX_PAIR_NS
#undef MPCX_PAIR_KERNEL

// the lane-pair layout at T = 10 (superseded by four lanes per instance): test-hook library only
#ifdef MPCX_TEST_HOOKS
#define MPCX_PAIR_TT 10
#define MPCX_PAIR_NS pair10
#define MPCX_PAIR_KERNEL k_solve_pair10
#include "mpcx_pair.h"
#include "mpcx_pair_kernel.h"
#undef MPCX_PAIR_TT
#undef MPCX_PAIR_NS
#undef MPCX_PAIR_KERNEL
#endif

#include "mpcx_quad.h"
#include "mpcx_quad_kernel.h"

namespace mpcx {

// ------------------------------------------------------------------------------------
// solve: one wavefront per instance (horizons beyond the lane-pair kernels, n = 2T <= 64)
// ------------------------------------------------------------------------------------
constexpr int kWaveAttempts = 3;
constexpr int kWaveIpmCap = 40;      // interior-point iterations at most per attempt but the last
// complementarity at which an attempt hands over to the active-set rounds (0: converged), and
// the rounds it may take
__device__ const double kWaveSwitchMu[kWaveAttempts] = {kSwitchMu, 1e-9, 0.0};
__device__ const int kWaveRounds[kWaveAttempts] = {5, 5, 5};

__host__ __device__ inline int wave_lds_doubles(int T)
{
    const int n = 2 * T;
    return n * (n + 1) + kWave + 11 * T + (T + 1) + 3 * (T + 1);   // H and K (packed triangles), broadcast buffer, dynamics, yaw, reference window
}

// NFIX: 2T when the kernel is built for one horizon (condensing and Cholesky then run on the matrix
// cores, the factor stays in registers), 0: any.
template <int MODEL, int NFIX = 0>
__global__ __launch_bounds__(kWave) void k_solve_wave(SolveArgs a)
{
    extern __shared__ double smem[];
    constexpr int NU = MODEL == MPCX_MODEL_DIFF_DRIVE ? 2 : 4;
    const mpcx_variant &v = a.v;
    const int T = v.T, W = T + 1, n = 2 * T;
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x;
    wave::Ctx c;
    c.n = n;
    c.H = (volatile wave::lds_double *)smem;
    c.K = c.H + c.size();
    c.vec = c.K + c.size();
    volatile wave::lds_double *dyn = c.vec + kWave, *th = dyn + 11 * T, *xw = th + W;
    c.lane = lane; c.act = lane < n; c.row = c.act ? lane : 0; c.own = c.tri(c.row);
    MPCX_STAMP_CLEAR();
    MPCX_STAMP(0);
    const int st_own = c.act ? lane >> 1 : T - 1, ch = lane & 1;   // the lane's stage and channel

    const double *__restrict__ xr = a.xref + b * 3 * W;
    double x0[3] = {a.x0[b * 3], a.x0[b * 3 + 1], a.x0[b * 3 + 2]};
    const double *uin = a.uin ? a.uin + b * NU * T : nullptr;
    const bool warm = a.linear_mode || (uin && (!a.has_prev || a.has_prev[b]));
    // inputs: every lane checks a slice, the wave votes
    bool fin = finite_d(x0[0]) && finite_d(x0[1]) && finite_d(x0[2]);
    for (int i = lane; i < 3 * W; i += kWave) {   // the window stays in LDS: the stage loops below read it
        const double xv = xr[i];
        fin = fin && finite_d(xv);
        xw[i] = xv;
    }
    if (a.linear_mode)
        for (int i = lane; i < 3 * W; i += kWave) fin = fin && finite_d(a.xbar[b * 3 * W + i]);
    double u[4] = {0.0, 0.0, 0.0, 0.0}, u0[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int r = 0; r < NU; ++r) {
        u[r] = warm ? uin[r * T + st_own] : v.cold_start[r];
        u0[r] = warm ? uin[r * T] : v.cold_start[r];
        fin = fin && finite_d(u[r]);
    }
    const bool finite = __all(fin) != 0;
    int status = MPCX_ST_OK, iters = 0;
    bool skip = !finite;
    if (!finite) {
        status = MPCX_ST_NONFINITE;
        x0[0] = x0[1] = x0[2] = 0.0;
    }

    // stage dynamics: lanes 2t and 2t + 1 evaluate stage t; the yaw is rolled in stage order
    SteerTerms kt;
    steer_terms<MODEL>(v, u, kt);
    if (a.linear_mode) {
        if (c.act && ch == 0) th[st_own] = a.xbar[b * 3 * W + 2 * W + st_own];
    } else {
        // every lane runs the roll (the rate of stage t from lane 2t's register) and keeps its own stage's
        // yaw; the lanes of channel 0 publish theirs
        double t0 = x0[2], mine = x0[2];
        for (int t = 0; t < T; ++t) {
            mine = st_own == t ? t0 : mine;
            t0 = t0 + wave::bcast(kt.omega, 2 * t) * v.dt;
        }
        if (c.act && ch == 0) th[st_own] = mine;
    }
    __builtin_amdgcn_wave_barrier();
    {
        double a02, a12, Bm[3][4], C[3], bt[3][2], ct[3];
        linearize_at<MODEL>(v, th[st_own], u, kt, a02, a12, Bm, C);
        reduce_controls(v, Bm, C, bt, ct);
        if (c.act && ch == 0) {
            volatile wave::lds_double *p = dyn + 11 * st_own;
            p[0] = a02; p[1] = a12;
#pragma unroll
            for (int r = 0; r < 3; ++r) { p[2 + 2 * r] = bt[r][0]; p[3 + 2 * r] = bt[r][1]; p[8 + r] = ct[r]; }
        }
    }
    __builtin_amdgcn_wave_barrier();

    MPCX_STAMP(1);
    // bounds of the lane's control
    {
        FreeBounds fb;
        free_bounds(v, u0, fb);
        if (!fb.feasible && !skip) { skip = true; status = MPCX_ST_INFEASIBLE; }
        const bool first = st_own == 0;
        const bool box0 = ch ? fb.box0[1] : fb.box0[0], box = ch ? fb.box[1] : fb.box[0], rate = ch ? fb.rate[1] : fb.rate[0];
        const double bb = ch ? fb.b[1] : fb.b[0];
        c.hb = c.act && (first ? box0 : box);
        c.lo = first ? (ch ? fb.lo0[1] : fb.lo0[0]) : -bb;
        c.hi = first ? (ch ? fb.hi0[1] : fb.hi0[0]) : bb;
        c.hr = c.act && lane < n - 2 && rate;
        c.rr = ch ? fb.rr[1] : fb.rr[0];
        const double m = wave::wsum((c.hb ? 2.0 : 0.0) + (c.hr ? 2.0 : 0.0));
        c.inv_m = m > 0.0 ? 1.0 / m : 0.0;
    }

    // condensed QP: row `lane` of H into LDS, g in a register
    {
        double rw[2], rdw[2];
        if (v.eq_mode == MPCX_EQ_NONE) {
            rw[0] = v.R[0]; rw[1] = v.R[1]; rdw[0] = v.Rd[0]; rdw[1] = v.Rd[1];
        } else if (v.eq_mode == MPCX_EQ_FIXED_STEER) {
            rw[0] = v.R[0]; rw[1] = v.R[2]; rdw[0] = v.Rd[0]; rdw[1] = v.Rd[2];
        } else {
            rw[0] = v.R[0] + v.R[2]; rw[1] = v.R[1] + v.R[3];
            rdw[0] = v.Rd[0] + v.Rd[2]; rdw[1] = v.Rd[1] + v.Rd[3];
        }
        const double rwc = ch ? rw[1] : rw[0], rdc = ch ? rdw[1] : rdw[0];
        if (NFIX > 0) {
            // One horizon: H = sum over stages and state components of (2 Q_r G_r)(G_r)' is a sum of
            // rank-3 updates, which the matrix cores take as 16 x 16 tiles (v_mfma_f64_16x16x4, the
            // fourth column of the operands zero): ten tiles cover the lower triangle of the 64 x 64
            // frame, four doubles per lane each.  Per stage the lanes publish their three entries of G
            // in LDS and read them back in operand order (lane l: entry of row 16 I + (l & 15), component
            // l >> 4); tiles that only lanes of later stages reach are still zero and are skipped.
            constexpr int N = NFIX > 0 ? NFIX : 2, TF = N / 2;
            typedef double tile_t __attribute__((ext_vector_type(4)));
            tile_t Ht[10];
#pragma unroll
            for (int q = 0; q < 10; ++q) Ht[q] = tile_t{0.0, 0.0, 0.0, 0.0};
            volatile wave::lds_double *gb = c.K;     // [4][64], K's area is idle until the first factorisation
            const int kq = lane >> 4, m16 = lane & 15;
            gb[3 * kWave + lane] = 0.0;
            double p[3] = {x0[0], x0[1], x0[2]}, G[3] = {0.0, 0.0, 0.0}, gacc = 0.0, qg = 1.0;
#pragma unroll 1
            for (int t = 0; t < TF; ++t) {
                volatile wave::lds_double *d = dyn + 11 * t;
                const double a02 = d[0], a12 = d[1];
                const double ct0 = d[8], ct1 = d[9], ct2 = d[10];
                const double pn0 = p[0] + a02 * p[2] + ct0;
                const double pn1 = p[1] + a12 * p[2] + ct1;
                const double pn2 = p[2] + ct2;
                p[0] = pn0; p[1] = pn1; p[2] = pn2;
                if (lane < 2 * t) {
                    G[0] = fma(a02, G[2], G[0]);
                    G[1] = fma(a12, G[2], G[1]);
                }
                if (c.act && st_own == t) {
#pragma unroll
                    for (int r = 0; r < 3; ++r) G[r] = d[2 + 2 * r + ch];
                }
                const int ts = t + 1;
                double Qs[3];
                if (ts < TF) {
                    qg *= v.q_growth;
#pragma unroll
                    for (int r = 0; r < 3; ++r) Qs[r] = v.Q[r] * qg;
                } else {
#pragma unroll
                    for (int r = 0; r < 3; ++r) Qs[r] = v.Qf[r];
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    gacc = fma(2.0 * Qs[r] * G[r], p[r] - xw[r * W + ts], gacc);
                    gb[r * kWave + lane] = G[r];
                }
                __builtin_amdgcn_wave_barrier();
                const double qk = kq == 0 ? 2.0 * Qs[0] : kq == 1 ? 2.0 * Qs[1] : kq == 2 ? 2.0 * Qs[2] : 0.0;
                double gt[4];
#pragma unroll
                for (int I = 0; I < 4; ++I) gt[I] = gb[kq * kWave + 16 * I + m16];
#pragma unroll
                for (int I = 0; I < 4; ++I) {
                    if (16 * I < 2 * ts) {
                        const double ai = gt[I] * qk;
#pragma unroll
                        for (int J = 0; J <= I; ++J)
                            Ht[I * (I + 1) / 2 + J] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, gt[J], Ht[I * (I + 1) / 2 + J], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
            // tile (I, J), register i of lane l = element (16 I + (l >> 4) + 4 i, 16 J + (l & 15))
#pragma unroll
            for (int J = 0; J < 4; ++J) {
                const int col = 16 * J + m16;
                const int tc = c.tri(col < N ? col : 0);
#pragma unroll
                for (int I = J; I < 4; ++I) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int row = 16 * I + kq + 4 * i;
                        if (row >= col && row < N) c.H[tc + row] = Ht[I * (I + 1) / 2 + J][i];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            if (c.act) {
                double dd = 2.0 * rwc;
                if (st_own > 0) { dd += 2.0 * rdc; c.H[c.tri(lane - 2) + lane] -= 2.0 * rdc; }
                if (st_own + 1 < TF) dd += 2.0 * rdc;
                c.H[c.own + lane] += dd;
            }
            __builtin_amdgcn_wave_barrier();
            c.g = c.act ? gacc : 0.0;
        } else {
        for (int e = lane; e < c.size(); e += kWave) c.H[e] = 0.0;
        __builtin_amdgcn_wave_barrier();
        if (c.act) {
            double d = 2.0 * rwc;
            if (st_own > 0) { d += 2.0 * rdc; c.H[c.tri(lane - 2) + lane] = -2.0 * rdc; }
            if (st_own + 1 < T) d += 2.0 * rdc;
            c.H[c.own + lane] = d;
        }
        __builtin_amdgcn_wave_barrier();
        double p[3] = {x0[0], x0[1], x0[2]}, G[3] = {0.0, 0.0, 0.0}, gacc = 0.0, qg = 1.0;
        for (int t = 0; t < T; ++t) {
            volatile wave::lds_double *d = dyn + 11 * t;
            const double a02 = d[0], a12 = d[1];
            const double ct0 = d[8], ct1 = d[9], ct2 = d[10];
            const double pn0 = p[0] + a02 * p[2] + ct0;
            const double pn1 = p[1] + a12 * p[2] + ct1;
            const double pn2 = p[2] + ct2;
            p[0] = pn0; p[1] = pn1; p[2] = pn2;
            if (lane < 2 * t) {
                G[0] = fma(a02, G[2], G[0]);
                G[1] = fma(a12, G[2], G[1]);
            }
            if (c.act && st_own == t) {
#pragma unroll
                for (int r = 0; r < 3; ++r) G[r] = d[2 + 2 * r + ch];
            }
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
            double gi[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                gi[r] = 2.0 * Qs[r] * G[r];
                gacc = fma(gi[r], p[r] - xw[r * W + ts], gacc);
            }
            // H(i, k) += sum_r gi_r G_r(k) for k <= i < 2 ts; G_r(k) from lane k's registers
            auto cols = [&](int k, auto qn) {
                constexpr int Q = decltype(qn)::value;
                double h[Q];
                int at[Q];
#pragma unroll
                for (int q = 0; q < Q; ++q) { at[q] = c.tri(k + q) + c.row; h[q] = c.H[at[q]]; }
#pragma unroll
                for (int q = 0; q < Q; ++q) {
                    h[q] = fma(gi[0], wave::bcast(G[0], k + q), h[q]);
                    h[q] = fma(gi[1], wave::bcast(G[1], k + q), h[q]);
                    h[q] = fma(gi[2], wave::bcast(G[2], k + q), h[q]);
                }
#pragma unroll
                for (int q = 0; q < Q; ++q)
                    if (lane >= k + q && lane < 2 * ts) c.H[at[q]] = h[q];
            };
            int k = 0;
            for (; k + 7 < 2 * ts; k += 8) cols(k, std::integral_constant<int, 8>());
            if (k + 3 < 2 * ts) { cols(k, std::integral_constant<int, 4>()); k += 4; }
            if (k + 1 < 2 * ts) cols(k, std::integral_constant<int, 2>());
            __builtin_amdgcn_wave_barrier();
        }
        c.g = c.act ? gacc : 0.0;
        }
    }
    const double hmax = wave::wmax(c.act ? c.H[c.own + lane] : 0.0);
    MPCX_STAMP(2);

    // solve
    c.w = 0.0; c.lamU = c.lamL = c.nuU = c.nuL = 0.0;
    if (!skip) {
        // Attempts: the interior point down to a complementarity at which the active set can
        // usually be read off the iterate, then the active-set rounds; if the rounds do not settle
        // the interior point goes on from where it stood (nothing is redone) and they are tried
        // again, the last time from a converged iterate, whose own verdict stands if they still fail.
        wave::Iterate keep;
        bool settled = false;
        for (int attempt = 0; attempt < kWaveAttempts && !settled; ++attempt) {
            int st = MPCX_ST_OK, it = 0;
            if (attempt > 0) keep.restore(c);
            const int cap = a.max_iter < kWaveIpmCap || attempt + 1 == kWaveAttempts ? a.max_iter : kWaveIpmCap;
            wave::ipm<NFIX>(c, cap, kWaveSwitchMu[attempt], attempt > 0, st, it);
            iters += it;
            status = st;
            keep.save(c);
            if (attempt == 0) { MPCX_STAMP(3); }
            settled = wave::pdas<NFIX>(c, kWaveRounds[attempt], hmax);
            if (attempt == 0) { MPCX_STAMP(4); }
        }
        if (settled) status = MPCX_ST_OK;
    }
    MPCX_STAMP(5);

    // epilogue: controls by the stage's lane pair, states by a roll over the stages
    const bool bad = status == MPCX_ST_NONFINITE || status == MPCX_ST_INFEASIBLE;
    const double nan = __builtin_nan("");
    const double sg = v.eq_mode == MPCX_EQ_ACKERMANN ? -1.0 : 1.0;
    c.vec[lane] = c.act ? c.w : 0.0;
    __builtin_amdgcn_wave_barrier();
    double Ju = 0.0;
    {
        auto controls = [&](double w0, double w1, double (&uu)[4]) {
            if (v.eq_mode == MPCX_EQ_NONE) { uu[0] = w0; uu[1] = w1; uu[2] = 0.0; uu[3] = 0.0; }
            else if (v.eq_mode == MPCX_EQ_FIXED_STEER) { uu[0] = w0; uu[1] = v.steer_fixed; uu[2] = w1; uu[3] = v.steer_fixed; }
            else { uu[0] = w0; uu[1] = w1; uu[2] = w0; uu[3] = sg * w1; }
        };
        double uc[4], up[4];
        controls(c.vec[2 * st_own], c.vec[2 * st_own + 1], uc);
        const int tp = st_own > 0 ? st_own - 1 : 0;
        controls(c.vec[2 * tp], c.vec[2 * tp + 1], up);
        if (c.act && ch == 0) {
#pragma unroll
            for (int r = 0; r < NU; ++r) {
                Ju = fma(v.R[r] * uc[r], uc[r], Ju);
                if (st_own > 0) { const double d = uc[r] - up[r]; Ju = fma(v.Rd[r] * d, d, Ju); }
                a.u_out[(b * NU + r) * T + st_own] = bad ? nan : uc[r];
            }
        }
    }
    // The states: every lane runs the roll over the stages (uniform work: LDS reads and ten
    // multiply-adds per stage), lane t keeps the state after stage t and its weight, and the tracking
    // cost and the stores are done side by side afterwards.
    {
        double x[3] = {x0[0], x0[1], x0[2]}, qg = 1.0;
        double mx[3] = {0.0, 0.0, 0.0}, mq = 0.0;
        double *xo = a.x_out + b * 3 * W;
        if (lane < 3) xo[lane * W] = bad ? nan : (lane == 0 ? x[0] : lane == 1 ? x[1] : x[2]);
        // (plain loads: nothing writes these areas any more, the compiler may fetch stages ahead)
        const wave::lds_double *dynr = (const wave::lds_double *)dyn, *vecr = (const wave::lds_double *)c.vec, *xwr = (const wave::lds_double *)xw;
        for (int t = 0; t < T; ++t) {
            const wave::lds_double *d = dynr + 11 * t;
            const double w0 = vecr[2 * t], w1 = vecr[2 * t + 1];
            const double n0 = x[0] + d[0] * x[2] + d[2] * w0 + d[3] * w1 + d[8];
            const double n1 = x[1] + d[1] * x[2] + d[4] * w0 + d[5] * w1 + d[9];
            const double n2 = x[2] + d[6] * w0 + d[7] * w1 + d[10];
            x[0] = n0; x[1] = n1; x[2] = n2;
            if (t + 1 < T) qg *= v.q_growth;
            const bool me = lane == t;
            mx[0] = me ? n0 : mx[0]; mx[1] = me ? n1 : mx[1]; mx[2] = me ? n2 : mx[2];
            mq = me ? qg : mq;
        }
        if (lane < T) {
            const int ts = lane + 1;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double e = xwr[r * W + ts] - mx[r];
                const double q = ts < T ? v.Q[r] * mq : v.Qf[r];
                Ju = fma(q * e, e, Ju);
                xo[r * W + ts] = bad ? nan : mx[r];
            }
        }
    }
    const double J = wave::wsum(Ju);
    if (lane == 0) {
        a.obj_out[b] = bad ? nan : J;
        a.status_out[b] = status;
        if (a.iters_out) a.iters_out[b] = iters;
    }
    MPCX_STAMP(6);
}

// ------------------------------------------------------------------------------------
// solve: one instance per lane, stage-wise (Riccati) linear algebra, arrays in a workspace
// ------------------------------------------------------------------------------------
#ifndef MPCX_STAGE_R0
#define MPCX_STAGE_R0 5
#endif
constexpr int kStageIpmCap = 40;   // interior-point iterations at most per attempt but the last
constexpr int kStageAttempts = 3;
// complementarity at which an attempt hands over to the active-set rounds (0: converged), and
// the rounds it may take
__device__ const double kStageSwitchMu[kStageAttempts] = {kSwitchMu, 1e-9, 0.0};
__device__ const int kStageRounds[kStageAttempts] = {MPCX_STAGE_R0, 6, 6};

#ifndef MPCX_STAGE_OCC
#define MPCX_STAGE_OCC 1   // wavefronts per SIMD the register budget is set for
#endif
// REDO: a second pass over a batch that another solve kernel has already answered: only the
// instances that kernel left MPCX_ST_INACCURATE or MPCX_ST_FAILED are solved again (from the same
// inputs) and rewritten; a wavefront without one returns at once.
template <int MODEL, bool REDO = false>
__global__ __launch_bounds__(kWave, MPCX_STAGE_OCC) void k_solve_stage(SolveArgs a, double *ws)
{
    constexpr int NU = MODEL == MPCX_MODEL_DIFF_DRIVE ? 2 : 4;
    const mpcx_variant &v = a.v;
    stagewise::Store s;
    const int64_t slot = (int64_t)blockIdx.x * kWave + threadIdx.x;
    s.bind(ws, v.T, blockIdx.x, threadIdx.x);
    int64_t b = slot;
    bool live = b < a.B;
    if (!live) b = a.B - 1;
    if (REDO) {
        live = live && (a.status_out[b] == MPCX_ST_INACCURATE || a.status_out[b] == MPCX_ST_FAILED);
        if (!__any(live)) return;
    }
    const int T = v.T, W = T + 1;

    stagewise::Problem p;
    const double *__restrict__ xr = a.xref + b * 3 * W;
    p.x0[0] = a.x0[b * 3]; p.x0[1] = a.x0[b * 3 + 1]; p.x0[2] = a.x0[b * 3 + 2];
    bool finite = finite_d(p.x0[0]) && finite_d(p.x0[1]) && finite_d(p.x0[2]);
    // The instance's inputs are requested eight stages at a time: a load issued after a store
    // waits behind it, so element-by-element copying would cost one memory round trip each.
    constexpr int kChunk = 8;
    for (int t0 = 0; t0 < W; t0 += kChunk) {
        double x[3][kChunk];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < kChunk; ++k) x[r][k] = xr[r * W + (t0 + k < W ? t0 + k : W - 1)];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < kChunk; ++k) {
                const int ts = t0 + k;
                finite = finite && finite_d(x[r][k]);   // past the end: the last column again
                if (ts > 0 && ts < W) s.st(s.f(s.fXr + r, ts - 1), x[r][k]);
            }
    }
    const double *uin = a.uin ? a.uin + b * NU * T : nullptr;
    const bool warm = a.linear_mode || (uin && (!a.has_prev || a.has_prev[b]));
    const int chain = a.linear_mode ? 1 : (v.chain > 1 ? (v.chain < T ? v.chain : T) : 1);
    // linearisation point: the incoming controls (or the cold-start constants) and the yaw
    // predict_motion gives them (or xbar's); chained calls keep the yaw and start from column 0
    // of the controls, the rest zero (mpc_sim.py:292-299)
    {
        double th = p.x0[2];
        const double *xb = a.linear_mode ? a.xbar + b * 3 * W : nullptr;
        for (int t0 = 0; t0 < T; t0 += kChunk) {
            double uu[4][kChunk], xs[3][kChunk];
#pragma unroll
            for (int k = 0; k < kChunk; ++k) {
                const int tt = t0 + k < T ? t0 + k : T - 1;
#pragma unroll
                for (int r = 0; r < 4; ++r) uu[r][k] = r < NU ? (warm ? uin[r * T + tt] : v.cold_start[r]) : 0.0;
#pragma unroll
                for (int r = 0; r < 3; ++r) xs[r][k] = xb ? xb[r * W + tt] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < kChunk; ++k) {
                const int t = t0 + k;
                if (t < T) {
                    double u[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { u[r] = uu[r][k]; finite = finite && finite_d(u[r]); }
                    if (xb) {
                        th = xs[2][k];
                        finite = finite && finite_d(xs[0][k]) && finite_d(xs[1][k]) && finite_d(xs[2][k]);
                    }
                    s.st(s.f(s.fTheta, t), th);
                    th = th + yaw_rate<MODEL>(v, u) * v.dt;
#pragma unroll
                    for (int r = 0; r < 4; ++r) s.st(s.f(s.fUref + r, t), (chain > 1 && t > 0) ? 0.0 : u[r]);
                }
            }
        }
        if (xb)
            for (int r = 0; r < 3; ++r) finite = finite && finite_d(xb[r * W + T]);
    }
    int status = MPCX_ST_OK, iters = 0;
    bool skip = !finite;
    if (!finite) {
        status = MPCX_ST_NONFINITE;
        p.x0[0] = p.x0[1] = p.x0[2] = 0.0;
    }
    if (v.eq_mode == MPCX_EQ_NONE) {
        p.rw[0] = v.R[0]; p.rw[1] = v.R[1]; p.rdw[0] = v.Rd[0]; p.rdw[1] = v.Rd[1];
    } else if (v.eq_mode == MPCX_EQ_FIXED_STEER) {
        p.rw[0] = v.R[0]; p.rw[1] = v.R[2]; p.rdw[0] = v.Rd[0]; p.rdw[1] = v.Rd[2];
    } else {
        p.rw[0] = v.R[0] + v.R[2]; p.rw[1] = v.R[1] + v.R[3];
        p.rdw[0] = v.Rd[0] + v.Rd[2]; p.rdw[1] = v.Rd[1] + v.Rd[3];
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) { p.q[r] = v.Q[r]; p.qf[r] = v.Qf[r]; }
    p.growth = v.q_growth;
    const double sg = v.eq_mode == MPCX_EQ_ACKERMANN ? -1.0 : 1.0;
    auto controls_of = [&](double w0, double w1, double (&u)[4]) {
        if (v.eq_mode == MPCX_EQ_NONE) { u[0] = w0; u[1] = w1; u[2] = 0.0; u[3] = 0.0; }
        else if (v.eq_mode == MPCX_EQ_FIXED_STEER) { u[0] = w0; u[1] = v.steer_fixed; u[2] = w1; u[3] = v.steer_fixed; }
        else { u[0] = w0; u[1] = w1; u[2] = w0; u[3] = sg * w1; }
    };
    auto controls = [&](int t, double (&u)[4]) { controls_of(s.ld(s.p(s.fW, t, 0)), s.ld(s.p(s.fW, t, 1)), u); };
    for (int t = 0; t < T; ++t) { s.st(s.p(s.fW, t, 0), 0.0); s.st(s.p(s.fW, t, 1), 0.0); }

    bool inaccurate = false;
    for (int pass = 0; pass < chain; ++pass) {
        if (pass > 0 && !skip) {   // column `pass` of uref <- previous answer (mpc_sim.py:296-299 after :302)
            double u[4];
            controls(pass, u);
#pragma unroll
            for (int r = 0; r < 4; ++r) s.st(s.f(s.fUref + r, pass), u[r]);
        }
        // Stage dynamics at the linearisation point and, in the same sweep, the two rollouts the
        // solve starts from: at w = 0 (for the gradient that scales the residual tests) and at the
        // interior point's starting controls, which the bounds of stage 0 determine.
        {
            struct Rec { double u[4], th, xr[3]; };
            stagewise::Roller at_zero, at_start;
            at_zero.start(p);
            at_start.start(p);
            double wst[2] = {0.0, 0.0};
            stagewise::sweep<Rec>(0, T - 1, 1,
                                  [&](int t, Rec &in) {
#pragma unroll
                                      for (int r = 0; r < 4; ++r) in.u[r] = s.ld(s.f(s.fUref + r, t));
                                      in.th = s.ld(s.f(s.fTheta, t));
#pragma unroll
                                      for (int r = 0; r < 3; ++r) in.xr[r] = s.ld(s.f(s.fXr + r, t));
                                  },
                                  [&](int t, const Rec &in) {
                                      if (t == 0) {
                                          free_bounds(v, in.u, p.fb);
                                          wst[0] = p.start(0); wst[1] = p.start(1);
                                      }
                                      double a02, a12, Bm[3][4], C[3], bt[3][2], ct[3], omega;
                                      linearize<MODEL>(v, in.th, in.u, a02, a12, Bm, C, omega);
                                      reduce_controls(v, Bm, C, bt, ct);
                                      stagewise::Dyn d;
                                      stagewise::RollIn ri;
                                      d.a02 = a02; d.a12 = a12;
                                      s.dyn_st(0, t, a02); s.dyn_st(1, t, a12);
#pragma unroll
                                      for (int r = 0; r < 3; ++r) {
                                          d.b[r][0] = bt[r][0]; d.b[r][1] = bt[r][1];
                                          ri.c[r] = ct[r]; ri.xr[r] = in.xr[r];
                                          s.dyn_st(2 + 2 * r, t, bt[r][0]); s.dyn_st(3 + 2 * r, t, bt[r][1]); s.dyn_st(8 + r, t, ct[r]);
                                      }
                                      at_zero.step(s, p, t, 0.0, 0.0, d, ri, s.fDq);
                                      at_start.step(s, p, t, wst[0], wst[1], d, ri);
                                  });
        }
        if (!p.fb.feasible && !skip) { skip = true; status = MPCX_ST_INFEASIBLE; }
        const double gmax = stagewise::gradient_at_zero(s, s.fDq);
        // initial complementarity in proportion to the gradient: with weights that grow like 2^t
        // (mpc_dsr.py:349) the gradient reaches 1e7, and from 0.1 the interior point spends 25
        // iterations just raising its multipliers to that scale
        p.mu0 = kMu0 * dmax(1.0, gmax);
        // Attempts: the interior point down to a complementarity at which the active set can
        // usually be read off the iterate, then the active-set rounds; a lane whose rounds do not
        // settle lets the interior point run further (it continues, nothing is redone) and tries
        // again, the last time from a converged iterate, whose own verdict stands if the rounds
        // still fail.
        int st = MPCX_ST_OK;
        const bool run = !skip;
        bool settled = false, todo = run;
        int spent = 0;
        for (int attempt = 0; attempt < kStageAttempts; ++attempt) {
            if (!__any(todo)) break;
            int it = 0, used = 0;
            const int cap = a.max_iter < kStageIpmCap || attempt + 1 == kStageAttempts ? a.max_iter : kStageIpmCap;
            stagewise::ipm(s, p, cap, kStageSwitchMu[attempt], attempt > 0, todo, st, it, /*rolled=*/true);
            if (todo) { iters += it; spent += it; status = st; }
            const bool ok = stagewise::pdas(s, p, kStageRounds[attempt], todo, used);
#ifdef MPCX_DIAG   // diagnostic builds report the rounds in the iteration count
            if (todo) iters += (attempt == 0 ? 100 : 10000) * (used + (attempt == 0 ? 0 : 1));
#endif
            if (todo && ok) { settled = true; todo = false; }
        }
        (void)spent;
        if (run) {
            if (settled) status = MPCX_ST_OK;
            if (status == MPCX_ST_INACCURATE) inaccurate = true;
            if (status == MPCX_ST_FAILED) skip = true;
        }
    }
    if (status == MPCX_ST_OK && inaccurate) status = MPCX_ST_INACCURATE;

    // controls, predicted states (the stage dynamics applied to the answer), objective
    const bool bad = status == MPCX_ST_NONFINITE || status == MPCX_ST_INFEASIBLE;
    const double nan = __builtin_nan("");
    double J = 0.0, qg = 1.0;
    double x[3] = {p.x0[0], p.x0[1], p.x0[2]};
    double up[4] = {0, 0, 0, 0};
    double *uo = a.u_out + b * NU * T, *xo = a.x_out + b * 3 * W;
    if (live) {
#pragma unroll
        for (int r = 0; r < 3; ++r) xo[r * W] = bad ? nan : x[r];
    }
    {
        struct Rec { double w[2]; stagewise::Dyn d; stagewise::RollIn r; };
        stagewise::sweep<Rec>(0, T - 1, 1,
                              [&](int t, Rec &in) {
                                  in.w[0] = s.ld(s.p(s.fW, t, 0)); in.w[1] = s.ld(s.p(s.fW, t, 1));
                                  stagewise::load_dyn(s, t, in.d);
                                  stagewise::load_roll(s, t, in.r);
                              },
                              [&](int t, const Rec &in) {
            const double w0 = in.w[0], w1 = in.w[1];
            double u[4];
            controls_of(w0, w1, u);
#pragma unroll
            for (int r = 0; r < NU; ++r) {
                J = fma(v.R[r] * u[r], u[r], J);
                if (t > 0) { const double d = u[r] - up[r]; J = fma(v.Rd[r] * d, d, J); }
                up[r] = u[r];
                if (live) uo[r * T + t] = bad ? nan : u[r];
            }
            const stagewise::Dyn &d = in.d;
            const double n0 = x[0] + d.a02 * x[2] + d.b[0][0] * w0 + d.b[0][1] * w1 + in.r.c[0];
            const double n1 = x[1] + d.a12 * x[2] + d.b[1][0] * w0 + d.b[1][1] * w1 + in.r.c[1];
            const double n2 = x[2] + d.b[2][0] * w0 + d.b[2][1] * w1 + in.r.c[2];
            x[0] = n0; x[1] = n1; x[2] = n2;
            const int ts = t + 1;
            if (ts < T) qg *= v.q_growth;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const double e = in.r.xr[r] - x[r];   // the workspace's copy of the reference window
                const double q = ts < T ? v.Q[r] * qg : v.Qf[r];
                J = fma(q * e, e, J);
                if (live) xo[r * W + ts] = bad ? nan : x[r];
            }
        });
    }
    if (live) {
        a.obj_out[b] = bad ? nan : J;
        a.status_out[b] = status;
        if (a.iters_out) a.iters_out[b] = iters;
    }
}

// ------------------------------------------------------------------------------------
// predict_motion / get_linear_model_matrix as stand-alone batched operators
// ------------------------------------------------------------------------------------
template <int MODEL>
__global__ __launch_bounds__(256) void k_rollout(mpcx_variant v, int64_t B, const double *__restrict__ x0,
                                                 const double *__restrict__ u, double *__restrict__ xbar)
{
    constexpr int NU = MODEL == MPCX_MODEL_DIFF_DRIVE ? 2 : 4;
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int T = v.T, W = T + 1;
    double x = x0[b * 3], y = x0[b * 3 + 1], yaw = x0[b * 3 + 2];
    double *xb = xbar + b * 3 * W;
    const double *ub = u + b * NU * T;
    xb[0] = x; xb[W] = y; xb[2 * W] = yaw;
    for (int t = 0; t < T; ++t) {
        double uu[4] = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < NU; ++r) uu[r] = ub[r * T + t];
        plant_step<MODEL>(v, x, y, yaw, uu);
        xb[t + 1] = x; xb[W + t + 1] = y; xb[2 * W + t + 1] = yaw;
    }
}

template <int MODEL>
__global__ __launch_bounds__(256) void k_linearize(mpcx_variant v, int64_t B, const double *__restrict__ theta,
                                                   const double *__restrict__ u, double *__restrict__ A,
                                                   double *__restrict__ Bo, double *__restrict__ C)
{
    constexpr int NU = MODEL == MPCX_MODEL_DIFF_DRIVE ? 2 : 4;
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double uu[4] = {0, 0, 0, 0}, Bm[3][4], Cc[3], a02, a12, om;
#pragma unroll
    for (int r = 0; r < NU; ++r) uu[r] = u[b * NU + r];
    linearize<MODEL>(v, theta[b], uu, a02, a12, Bm, Cc, om);
    double *Ab = A + b * 9;
    Ab[0] = 1; Ab[1] = 0; Ab[2] = a02; Ab[3] = 0; Ab[4] = 1; Ab[5] = a12; Ab[6] = 0; Ab[7] = 0; Ab[8] = 1;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int k = 0; k < NU; ++k) Bo[(b * 3 + r) * NU + k] = Bm[r][k];
        C[b * 3 + r] = Cc[r];
    }
}

// ------------------------------------------------------------------------------------
// nearest course point + reference window: 16 lanes per instance.
// Lane k of a group looks at course point pind + k (+16 per extra sweep), the group
// reduces (distance, index) with lexicographic min so ties go to the lowest index like
// list.index(min(d)) does (mpc_diff_drive.py:110-112), and lanes 0..T write the columns.
// Distances are dx*dx + dy*dy in float64 with no fused multiply-add: the index has to be
// bit-exact.
// ------------------------------------------------------------------------------------
// course builder: calc_spline_course + smooth_yaw, one 64-lane block per course
// ------------------------------------------------------------------------------------
constexpr int kMaxWaypoints = 64;

// Natural cubic spline coefficients through (s_i, a_i), i < n (cubic_spline_planner.py:38-66,
// 146-171): the reference solves the tridiagonal system densely; here it is eliminated
// directly.  c[0] = c[n-1] = 0.
__device__ inline void spline_coefficients(int n, const double *sk, const double *a, double *b, double *c, double *d,
                                           double *dp, double *rp)
{
    for (int i = 0; i < n; ++i) c[i] = 0.0;
    const int m = n - 2;
    if (m > 0) {
        for (int i = 0; i < m; ++i) {
            const double h0 = sk[i + 1] - sk[i], h1 = sk[i + 2] - sk[i + 1];
            dp[i] = 2.0 * (h0 + h1);
            rp[i] = 3.0 * ((a[i + 2] - a[i + 1]) / h1 - (a[i + 1] - a[i]) / h0);
        }
        for (int i = 1; i < m; ++i) {
            const double off = sk[i + 1] - sk[i];   // sub- and super-diagonal entry between rows i-1 and i
            const double f = off / dp[i - 1];
            dp[i] -= f * off;
            rp[i] -= f * rp[i - 1];
        }
        c[m] = rp[m - 1] / dp[m - 1];
        for (int i = m - 2; i >= 0; --i) c[i + 1] = (rp[i] - (sk[i + 2] - sk[i + 1]) * c[i + 2]) / dp[i];
    }
    for (int i = 0; i + 1 < n; ++i) {
        const double h = sk[i + 1] - sk[i];
        d[i] = (c[i + 1] - c[i]) / (3.0 * h);
        b[i] = (a[i + 1] - a[i]) / h - h / 3.0 * (2.0 * c[i] + c[i + 1]);
    }
}

// wx, wy [K][Nmax] waypoints (nway[k] valid, >= 2); outputs cx, cy, cyaw, ck [K][Lmax] and
// clen[k] = number of samples s = 0, ds, 2 ds, ... < total chord length
// (cubic_spline_planner.py:311-323), or -1 when they do not fit Lmax or the input is bad.
// With `smooth` the yaw column is unwrapped sequentially as smooth_yaw does
// (mpc_diff_drive.py:404-417).
__global__ __launch_bounds__(kWave) void k_build_courses(int64_t K, int Nmax, const double *__restrict__ wx,
                                                         const double *__restrict__ wy,
                                                         const int32_t *__restrict__ nway, double ds, int64_t Lmax,
                                                         int smooth, double *__restrict__ cx, double *__restrict__ cy,
                                                         double *__restrict__ cyaw, double *__restrict__ ck,
                                                         int32_t *__restrict__ clen)
{
    __shared__ double sk[kMaxWaypoints], ax[kMaxWaypoints], ay[kMaxWaypoints];
    __shared__ double bx[kMaxWaypoints], cxs[kMaxWaypoints], dx_[kMaxWaypoints];
    __shared__ double by[kMaxWaypoints], cys[kMaxWaypoints], dy_[kMaxWaypoints];
    __shared__ double w1[kMaxWaypoints], w2[kMaxWaypoints], w3[kMaxWaypoints], w4[kMaxWaypoints];
    __shared__ int s_len;
    const int64_t k = blockIdx.x;
    if (k >= K) return;   // the whole block: nothing below is reached by part of it
    const int lane = threadIdx.x;
    const int n = nway[k];
    const bool ok = n >= 2 && n <= Nmax && n <= kMaxWaypoints;
    if (ok && lane < n) { ax[lane] = wx[k * Nmax + lane]; ay[lane] = wy[k * Nmax + lane]; }
    __syncthreads();
    if (lane == 0) {
        int len = -1;
        if (ok) {
            sk[0] = 0.0;
            bool good = true;
            for (int i = 0; i + 1 < n; ++i) {   // cumulative chord length, summed in order like numpy.cumsum
                const double h = hypot(ax[i + 1] - ax[i], ay[i + 1] - ay[i]);
                sk[i + 1] = sk[i] + h;
                good = good && h > 0.0 && finite_d(h);
            }
            if (good && ds > 0.0) {
                const double cnt = ceil(sk[n - 1] / ds);   // len(numpy.arange(0, s_end, ds))
                if (cnt >= 1.0 && cnt <= (double)Lmax) len = (int)cnt;
            }
        }
        s_len = len;
    }
    __syncthreads();
    const int len = s_len;
    if (len < 0) {
        if (lane == 0) clen[k] = -1;
        return;
    }
    if (lane == 0) spline_coefficients(n, sk, ax, bx, cxs, dx_, w1, w2);
    if (lane == 1) spline_coefficients(n, sk, ay, by, cys, dy_, w3, w4);
    __syncthreads();
    double *ox = cx + k * Lmax, *oy = cy + k * Lmax, *ow = cyaw + k * Lmax, *oc = ck ? ck + k * Lmax : nullptr;
    for (int i = lane; i < len; i += kWave) {
        const double sv = (double)i * ds;                      // numpy.arange: start + i * step
        int lo = 0, hi = n;                                    // bisect.bisect(knots, sv) - 1
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (sv < sk[mid]) hi = mid; else lo = mid + 1;
        }
        int seg = lo - 1;
        seg = seg < 0 ? 0 : (seg > n - 2 ? n - 2 : seg);
        const double t = sv - sk[seg];
        const double t2 = __dmul_rn(t, t), t3 = __dmul_rn(t2, t);
        // a + b t + c t^2 + d t^3, summed left to right without contraction, as Python does
        const double px = __dadd_rn(__dadd_rn(__dadd_rn(ax[seg], __dmul_rn(bx[seg], t)), __dmul_rn(cxs[seg], t2)), __dmul_rn(dx_[seg], t3));
        const double py = __dadd_rn(__dadd_rn(__dadd_rn(ay[seg], __dmul_rn(by[seg], t)), __dmul_rn(cys[seg], t2)), __dmul_rn(dy_[seg], t3));
        const double vx = __dadd_rn(__dadd_rn(bx[seg], __dmul_rn(__dmul_rn(2.0, cxs[seg]), t)), __dmul_rn(__dmul_rn(3.0, dx_[seg]), t2));
        const double vy = __dadd_rn(__dadd_rn(by[seg], __dmul_rn(__dmul_rn(2.0, cys[seg]), t)), __dmul_rn(__dmul_rn(3.0, dy_[seg]), t2));
        const double axx = __dadd_rn(__dmul_rn(2.0, cxs[seg]), __dmul_rn(__dmul_rn(6.0, dx_[seg]), t));
        const double ayy = __dadd_rn(__dmul_rn(2.0, cys[seg]), __dmul_rn(__dmul_rn(6.0, dy_[seg]), t));
        ox[i] = px;
        oy[i] = py;
        ow[i] = atan2(vy, vx);
        if (oc) oc[i] = (ayy * vx - axx * vy) / pow(vx * vx + vy * vy, 1.5);
    }
    if (lane == 0) clen[k] = len;
    if (!smooth) return;
    __syncthreads();   // the block's own global writes are visible to it after the barrier
    if (lane == 0) {
        double prev = ow[0];
        for (int i = 0; i + 1 < len; ++i) {
            double cur = ow[i + 1];
            double dyaw = cur - prev;
            while (dyaw >= M_PI / 2.0) { cur -= M_PI * 2.0; dyaw = cur - prev; }
            while (dyaw <= -M_PI / 2.0) { cur += M_PI * 2.0; dyaw = cur - prev; }
            ow[i + 1] = cur;
            prev = cur;
        }
    }
}

// ------------------------------------------------------------------------------------
// closed loop: what do_simulation does between two solves
// ------------------------------------------------------------------------------------
// One lane per instance.  update_state with the first control column (mpc_diff_drive.py:92-100;
// mpc_dsr.py:112-121 after transfrom_gbm_command, :123-150), only when the solve succeeded
// (`if ov is not None`, mpc_diff_drive.py:368-370); the next solve is warm-started from this
// solution, or cold-started after a failure (the reference's ov stays None); check_goal
// (mpc_diff_drive.py:299-316) against the last point of the instance's course.  Instances
// that reached the goal (done = 1) or lost their reference window (done = 2, where the
// reference raises) are frozen.
template <int MODEL>
__global__ __launch_bounds__(256) void k_advance(mpcx_variant v, int64_t B, double goal_dis, double stop_speed,
                                                 const double *__restrict__ u, const int32_t *__restrict__ status,
                                                 const int32_t *__restrict__ target_ind,
                                                 const int32_t *__restrict__ course_id,
                                                 const double *__restrict__ cx, const double *__restrict__ cy,
                                                 const int32_t *__restrict__ clen, int64_t n_courses, int64_t Lmax,
                                                 double *__restrict__ state, double *__restrict__ twist,
                                                 uint8_t *__restrict__ has_prev, uint8_t *__restrict__ done,
                                                 int32_t *__restrict__ ticks, double *__restrict__ log_out)
{
    constexpr int NU = MODEL == MPCX_MODEL_DIFF_DRIVE ? 2 : 4;
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double x = state[4 * b], y = state[4 * b + 1], yaw = state[4 * b + 2], vx = state[4 * b + 3];
    if (!done[b]) {
        const int T = v.T;
        const bool solved = status[b] <= MPCX_ST_INACCURATE;
        if (solved) {
            double uc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < NU; ++r) uc[r] = u[(b * NU + r) * T];
            double s, c;
            sincos_fast(yaw, s, c);
            const double dt = v.dt;
            if (MODEL == MPCX_MODEL_DIFF_DRIVE) {
                x = x + uc[0] * c * dt;
                y = y + uc[0] * s * dt;
                yaw = yaw + uc[1] * dt;
                vx = uc[0];
                if (twist) { twist[2 * b] = 0.0; twist[2 * b + 1] = uc[1]; }
            } else {
                double sf, cf, sr, cr;
                sincos_fast(uc[1], sf, cf);
                sincos_fast(uc[3], sr, cr);
                const double v1x = cf * uc[0], v1y = sf * uc[0], v2x = cr * uc[2], v2y = sr * uc[2];
                const double tx = 0.5 * (v1x + v2x), ty = 0.5 * (v1y + v2y), tw = (v1y - v2y) / v.wb;
                x = x + tx * c * dt - ty * s * dt;
                y = y + tx * s * dt + ty * c * dt;
                yaw = yaw + tw * dt;
                vx = tx;
                if (twist) { twist[2 * b] = ty; twist[2 * b + 1] = tw; }
            }
            state[4 * b] = x; state[4 * b + 1] = y; state[4 * b + 2] = yaw; state[4 * b + 3] = vx;
        }
        has_prev[b] = solved ? 1 : 0;
        ticks[b] += 1;
        // a course id outside the table, or a course the builder rejected (clen = -1) or that
        // claims more points than a row holds, counts as a lost reference window
        const int64_t course = course_id ? course_id[b] : 0;
        const bool known = course >= 0 && course < n_courses;
        const int len = known ? clen[course] : 0;
        const int tind = target_ind[b];
        if (tind < 0 || len < 1 || (int64_t)len > Lmax) {
            done[b] = 2;
        } else {
            const double dx = x - cx[course * Lmax + len - 1], dy = y - cy[course * Lmax + len - 1];
            bool isgoal = hypot(dx, dy) <= goal_dis;
            if (abs(tind - len) >= 5) isgoal = false;
            if (isgoal && fabs(vx) <= stop_speed) done[b] = 1;
        }
    }
    if (log_out) {
        log_out[4 * b] = x; log_out[4 * b + 1] = y; log_out[4 * b + 2] = yaw; log_out[4 * b + 3] = vx;
    }
}

// What MPC.doSimulationWithAllMotionModes (mpc_sim.py:254-265) does between two ticks of a vehicle
// whose course is a list of segments, and what doSimulation does before its loop for the segment
// entered (mpc_sim.py:204-222): count the tick against the segment's budget; a lost reference
// window ends the run; a goal reached or a spent budget moves on to the next segment (state kept,
// warm start dropped, initial yaw compensation against the new segment's first yaw, nearest index
// to be searched from the segment's start: enter = 1) or, after the last one, ends the run.
__global__ __launch_bounds__(256) void k_segment_step(int64_t B, int cap, const uint8_t *__restrict__ was_active,
                                                      const int32_t *__restrict__ seg_start,
                                                      const int32_t *__restrict__ seg_count, int32_t *__restrict__ seg,
                                                      int32_t *__restrict__ seg_ticks, uint8_t *__restrict__ finished,
                                                      uint8_t *__restrict__ done, uint8_t *__restrict__ has_prev,
                                                      int32_t *__restrict__ course_id, double *__restrict__ state,
                                                      const double *__restrict__ cyaw, int64_t n_courses, int64_t Lmax,
                                                      const int32_t *__restrict__ target_ind, int32_t *__restrict__ pind_next,
                                                      uint8_t *__restrict__ enter)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    bool go = false;
    if (was_active[b]) {
        const int nt = seg_ticks[b] + 1;
        seg_ticks[b] = nt;
        const int d = done[b];
        if (d == 2) {
            finished[b] = 1;
        } else if (d == 1 || nt >= cap) {
            const int sn = seg[b] + 1;
            seg[b] = sn;
            if (sn >= seg_count[b]) {
                finished[b] = 1;
                done[b] = 1;
            } else {
                go = true;
                const int64_t course = (int64_t)seg_start[b] + sn;
                course_id[b] = (int32_t)course;
                if (course >= 0 && course < n_courses) {   // rows outside the table: an empty window, never an index
                    const double yaw = state[4 * b + 2];
                    const double dy = yaw - cyaw[course * Lmax];
                    const double two_pi = 2.0 * 3.141592653589793;
                    if (dy >= 3.141592653589793) state[4 * b + 2] = yaw - two_pi;
                    else if (dy <= -3.141592653589793) state[4 * b + 2] = yaw + two_pi;
                }
                has_prev[b] = 0;
                done[b] = 0;
                seg_ticks[b] = 0;
            }
        }
    }
    enter[b] = go ? 1 : 0;
    pind_next[b] = go ? 0 : target_ind[b];
}

// ------------------------------------------------------------------------------------
// A fleet sorted by motion mode (MPC.iterativeLMPC's dispatch on `mode`, mpc_sim.py:301-306, for
// many vehicles): a stable counting sort by key = the mode of the vehicle's current table row, or
// n_modes for a vehicle at rest, in three small kernels (per-block counts, one-block scan, ranks
// and gather), and the way back for the answers.  kSortBlock vehicles per block.
// ------------------------------------------------------------------------------------
constexpr int kSortBlock = 1024, kSortMaxKeys = 8, kSortScanMax = 12288;

__device__ __forceinline__ int sort_key(int64_t b, int64_t B, int n_modes, const uint8_t *finished, const uint8_t *done,
                                        const int32_t *course_id, const int32_t *seg_mode, int64_t n_courses)
{
    if (b >= B) return -1;
    if (finished[b] != 0 || done[b] != 0) return n_modes;
    int64_t c = course_id[b];
    c = c < 0 ? 0 : (c >= n_courses ? n_courses - 1 : c);
    const int m = seg_mode[c];
    return m < 0 || m >= n_modes ? n_modes : m;
}

__global__ __launch_bounds__(kSortBlock) void k_mode_count(int64_t B, int n_modes, const uint8_t *__restrict__ finished,
                                                           const uint8_t *__restrict__ done, const int32_t *__restrict__ course_id,
                                                           const int32_t *__restrict__ seg_mode, int64_t n_courses,
                                                           uint8_t *__restrict__ was_active, int32_t *__restrict__ counts)
{
    __shared__ int cnt[kSortMaxKeys];
    if (threadIdx.x < kSortMaxKeys) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t b = (int64_t)blockIdx.x * kSortBlock + threadIdx.x;
    const int key = sort_key(b, B, n_modes, finished, done, course_id, seg_mode, n_courses);
    if (b < B) was_active[b] = key < n_modes ? 1 : 0;
    for (int m = 0; m <= n_modes; ++m) {
        const unsigned long long mask = __ballot(key == m);
        if ((threadIdx.x & (kWave - 1)) == 0 && mask) atomicAdd(&cnt[m], __popcll(mask));
    }
    __syncthreads();
    if ((int)threadIdx.x <= n_modes) counts[(int64_t)blockIdx.x * (n_modes + 1) + threadIdx.x] = cnt[threadIdx.x];
}

// counts [nblocks][n_modes + 1] -> in place: start of (block, key) in the sorted order; bounds
// [n_modes + 2]: start of every key's run and B.
__global__ __launch_bounds__(256) void k_mode_scan(int nblocks, int n_modes, int32_t *__restrict__ counts,
                                                   int32_t *__restrict__ bounds)
{
    __shared__ int buf[kSortScanMax];
    __shared__ int total[kSortMaxKeys + 1];
    const int nk = n_modes + 1, n = nblocks * nk;
    for (int i = threadIdx.x; i < n; i += blockDim.x) buf[i] = counts[i];
    __syncthreads();
    if ((int)threadIdx.x < nk) {      // one thread per key walks the blocks
        int run = 0;
        for (int blk = 0; blk < nblocks; ++blk) {
            const int c = buf[blk * nk + threadIdx.x];
            buf[blk * nk + threadIdx.x] = run;
            run += c;
        }
        total[threadIdx.x] = run;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int m = 0; m < nk; ++m) { const int c = total[m]; total[m] = run; bounds[m] = run; run += c; }
        bounds[nk] = run;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) counts[i] = buf[i] + total[i % nk];
}

// Position of every vehicle in the sorted order (stable: by vehicle index inside a key), the
// permutation, and the solve's inputs gathered into that order, record by record.
__global__ __launch_bounds__(kSortBlock) void k_mode_gather(int64_t B, int n_modes, const uint8_t *__restrict__ finished,
                                                            const uint8_t *__restrict__ done, const int32_t *__restrict__ course_id,
                                                            const int32_t *__restrict__ seg_mode, int64_t n_courses,
                                                            const int32_t *__restrict__ starts, int32_t *__restrict__ order,
                                                            int n0, const double *__restrict__ x0, double *__restrict__ x0_s,
                                                            int n1, const double *__restrict__ xref, double *__restrict__ xref_s,
                                                            int n2, const double *__restrict__ up, double *__restrict__ up_s,
                                                            const uint8_t *__restrict__ hp, uint8_t *__restrict__ hp_s)
{
    constexpr int kWaves = kSortBlock / kWave;
    __shared__ int wcnt[kWaves][kSortMaxKeys];
    __shared__ int pos[kSortBlock];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
    const int64_t base = (int64_t)blockIdx.x * kSortBlock, b = base + threadIdx.x;
    const int key = sort_key(b, B, n_modes, finished, done, course_id, seg_mode, n_courses);
    int rank = 0;
    for (int m = 0; m <= n_modes; ++m) {
        const unsigned long long mask = __ballot(key == m);
        if (key == m) rank = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[wave][m] = __popcll(mask);
    }
    __syncthreads();
    int p = -1;
    if (key >= 0) {
        int before = 0;
        for (int w = 0; w < wave; ++w) before += wcnt[w][key];
        p = starts[(int64_t)blockIdx.x * (n_modes + 1) + key] + before + rank;
        order[p] = (int32_t)b;
        hp_s[p] = hp ? hp[b] : 1;
    }
    pos[threadIdx.x] = p;
    __syncthreads();
    const int64_t live = B - base < kSortBlock ? B - base : kSortBlock;
    auto copy = [&](int per, const double *__restrict__ src, double *__restrict__ dst) {
        if (!src) return;
        const int64_t n = live * per;
        for (int64_t e = threadIdx.x; e < n; e += kSortBlock) {
            const int r = (int)(e / per), j = (int)(e % per);
            dst[(int64_t)pos[r] * per + j] = src[(base + r) * per + j];
        }
    };
    copy(n0, x0, x0_s);
    copy(n1, xref, xref_s);
    copy(n2, up, up_s);
}

// The answers back in the fleet's order: row p of the sorted arrays belongs to vehicle order[p].
__global__ __launch_bounds__(kSortBlock) void k_mode_scatter(int64_t B, const int32_t *__restrict__ order, int per,
                                                             const double *__restrict__ u_s, double *__restrict__ u,
                                                             const int32_t *__restrict__ status_s, int32_t *__restrict__ status)
{
    __shared__ int who[kSortBlock];
    const int64_t base = (int64_t)blockIdx.x * kSortBlock, p = base + threadIdx.x;
    who[threadIdx.x] = p < B ? order[p] : -1;
    if (p < B) status[who[threadIdx.x]] = status_s[p];
    __syncthreads();
    const int64_t live = B - base < kSortBlock ? B - base : kSortBlock;
    const int64_t n = live * per;
    for (int64_t e = threadIdx.x; e < n; e += kSortBlock) {
        const int r = (int)(e / per), j = (int)(e % per);
        u[(int64_t)who[r] * per + j] = u_s[(base + r) * per + j];
    }
}

// ------------------------------------------------------------------------------------
constexpr int kGroup = 16;

__global__ __launch_bounds__(256) void k_nearest_ref(int mode, int T, double dt, double param, int n_search,
                                                     int sign, int64_t B, const double *__restrict__ state,
                                                     const int32_t *__restrict__ course_id,
                                                     const int32_t *pind_in,   // may be the same buffer as ind_out
                                                     const double *__restrict__ cx, const double *__restrict__ cy,
                                                     const double *__restrict__ cyaw,
                                                     const int32_t *__restrict__ clen, int64_t n_courses, int64_t Lmax,
                                                     int32_t *ind_out, double *__restrict__ mind_out,
                                                     double *__restrict__ xref_out, double *__restrict__ x0_out)
{
    const int64_t gid = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / kGroup;
    const int k = threadIdx.x % kGroup;
    const bool live = gid < B;
    const int64_t b = live ? gid : B - 1;
    // a course id outside the table, or a course whose length is not 1..Lmax (the builder marks
    // a rejected course with -1), is searched as an empty window
    int64_t course = course_id ? course_id[b] : 0;
    const bool known = course >= 0 && course < n_courses;
    if (!known) course = 0;
    const double *tx = cx + course * Lmax, *ty = cy + course * Lmax, *tw = cyaw + course * Lmax;
    int len = known ? clen[course] : 0;
    if (len < 0 || (int64_t)len > Lmax) len = 0;
    const int pind = pind_in[b];
    const double x = state[4 * b], y = state[4 * b + 1];
    // x0 = [state.x, state.y, state.yaw] of the solve that follows (mpc_diff_drive.py:363)
    if (live && x0_out && k < 3) x0_out[3 * b + k] = state[4 * b + k];
    // a negative previous index (a window lost earlier) is an empty window too: never a read
    // in front of the row
    const int lo = pind < 0 ? 0 : (pind < len ? pind : len);
    const int64_t end = (int64_t)pind + n_search;
    const int hi = end < len ? (int)end : len;
    double best = kBig;
    int bi = 0x7fffffff;
    for (int j = lo + k; j < hi; j += kGroup) {
        const double dx = x - tx[j], dy = y - ty[j];
        const double d = __dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy));
        if (d < best) { best = d; bi = j; }   // increasing j per lane: strict < keeps the first
    }
#pragma unroll
    for (int off = kGroup / 2; off > 0; off >>= 1) {
        const double ob = __shfl_xor(best, off, kGroup);
        const int oi = __shfl_xor(bi, off, kGroup);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    const bool empty = pind < 0 || hi <= lo;
    // every distance NaN (a non-finite position): no comparison ever held, and Python's min()
    // and list.index() then settle on the first element of the window
    if (!empty && bi == 0x7fffffff) { bi = lo; best = __builtin_nan(""); }
    const int W = T + 1;
    if (empty) {
        if (live && k == 0) {
            ind_out[b] = -1;
            if (mind_out) mind_out[b] = __builtin_nan("");
        }
        if (live && xref_out)
            for (int i = k; i < 3 * W; i += kGroup) xref_out[b * 3 * W + i] = __builtin_nan("");
        return;
    }
    if (live && k == 0 && mind_out) {
        double m = sqrt(best);
        if (sign) {
            double ang = tw[bi] - atan2(ty[bi] - y, tx[bi] - x);
            while (ang > M_PI) ang -= 2.0 * M_PI;     // pi_2_pi, mpc_diff_drive.py:66-73
            while (ang < -M_PI) ang += 2.0 * M_PI;
            if (ang < 0) m = -m;
        }
        mind_out[b] = m;
    }
    int ind = bi;
    if (pind >= ind) ind = pind;   // mpc_diff_drive.py:247-248
    if (live && k == 0) ind_out[b] = ind;
    if (!live || !xref_out) return;
    double *xr = xref_out + b * 3 * W;
    if (mode == 0) {
        // calc_ref_trajectory: travel accumulates sequentially, int(round()) is half-to-even
        const double step = fabs(state[4 * b + 3]) * dt;
        double travel = 0.0;
        for (int i = 0; i < W; ++i) {
            travel += step;
            if (i % kGroup != k) continue;
            // a speed or spacing that is not a number gives a window that is not one either (the
            // reference's int(round()) raises there); anything else is clamped onto the row
            const double q = travel / param;
            if (!(q == q)) {
                xr[i] = xr[W + i] = xr[2 * W + i] = __builtin_nan("");
                continue;
            }
            const long dind = q >= (double)len ? (long)len : (q <= -(double)len ? -(long)len : (long)rint(q));
            long kk = ind + dind < len ? ind + dind : len - 1;
            if (kk < 0) kk = 0;
            xr[i] = tx[kk]; xr[W + i] = ty[kk]; xr[2 * W + i] = tw[kk];
        }
    } else {
        int cnt = param >= 1.0 ? (param < 2147483647.0 ? (int)param : 2147483647) : 1;   // at least the point itself
        if ((int64_t)ind + cnt > len) cnt = len - ind;
        for (int i = k; i < W; i += kGroup) {
            const int kk = ind + (i < cnt ? i : cnt - 1);
            xr[i] = tx[kk]; xr[W + i] = ty[kk]; xr[2 * W + i] = tw[kk];
        }
    }
}

// ------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static int hip_fail(hipError_t e, const char *what)
{
    return fail(MPCX_E_HIP, "%s: %s", what, hipGetErrorString(e));
}

// hipGetLastError() also returns (and clears) an error left behind by an unrelated earlier HIP
// call of the process; drop such a leftover before a launch so that the check after it reports
// this launch only.
static void clear_stale_error() { (void)hipGetLastError(); }

static int check_variant(const mpcx_variant *v)
{
    if (!v) return fail(MPCX_E_ARG, "variant is NULL");
    if (v->T < 1 || v->T > MPCX_MAX_T) return fail(MPCX_E_ARG, "horizon T=%d outside 1..%d", v->T, MPCX_MAX_T);
    if (v->model == MPCX_MODEL_DIFF_DRIVE) {
        if (v->nu != 2 || v->eq_mode != MPCX_EQ_NONE)
            return fail(MPCX_E_ARG, "diff-drive needs nu=2 and eq_mode NONE");
    } else if (v->model == MPCX_MODEL_DSR) {
        if (v->nu != 4 || v->eq_mode < MPCX_EQ_ACKERMANN || v->eq_mode > MPCX_EQ_FIXED_STEER)
            return fail(MPCX_E_ARG, "dual-steer needs nu=4 and an ackermann/crab/fixed-steer equality map");
    } else {
        return fail(MPCX_E_ARG, "unknown model %d", v->model);
    }
    if (!(v->dt > 0.0) || !(v->wb > 0.0)) return fail(MPCX_E_ARG, "dt and wb must be positive");
    return 0;
}

// every free channel has a finite box and a finite rate bound -> branch-free solver
static bool all_present(const mpcx_variant *v)
{
    auto fin = [](double x) { return std::isfinite(x); };
    if (v->eq_mode == MPCX_EQ_NONE) return fin(v->umax[0]) && fin(v->umax[1]) && fin(v->dumax[0]) && fin(v->dumax[1]);
    if (v->eq_mode == MPCX_EQ_FIXED_STEER)
        return fin(v->umax[0]) && fin(v->umax[2]) && fin(v->dumax[0]) && fin(v->dumax[2]);
    return (fin(v->umax[0]) || fin(v->umax[2])) && (fin(v->umax[1]) || fin(v->umax[3])) &&
           (fin(v->dumax[0]) || fin(v->dumax[2])) && (fin(v->dumax[1]) || fin(v->dumax[3]));
}

// the register-resident kernels are built for T = 5 with every bound present
static bool uses_registers(const mpcx_variant *v) { return v->T == 5 && all_present(v); }

// the lane-pair kernels need a finite box on both free channels; a missing rate bound is
// replaced in the kernel by one the box makes redundant
static bool boxes_present(const mpcx_variant *v)
{
    auto fin = [](double x) { return std::isfinite(x); };
    if (v->eq_mode == MPCX_EQ_NONE) return fin(v->umax[0]) && fin(v->umax[1]);
    if (v->eq_mode == MPCX_EQ_FIXED_STEER) return fin(v->umax[0]) && fin(v->umax[2]);
    return (fin(v->umax[0]) || fin(v->umax[2])) && (fin(v->umax[1]) || fin(v->umax[3]));
}

// Routing hooks of the test library only (-DMPCX_TEST_HOOKS; the product library reads no
// environment variable): MPCX_SINGLE_LANE=1 routes T = 5 solves to the one-lane-per-instance
// kernel; MPCX_STAGEWISE=1 routes every horizon other than 5 to the stage-wise kernel, =0 every
// one the wavefront-per-instance kernel can take to that kernel.
#ifdef MPCX_TEST_HOOKS
static const bool g_force_single_lane = [] { const char *e = getenv("MPCX_SINGLE_LANE"); return e && e[0] == '1'; }();
static const int g_stagewise = [] { const char *e = getenv("MPCX_STAGEWISE"); return e ? atoi(e) : -1; }();
// MPCX_T10=pair routes T = 10 to the lane-pair kernel (up to its batch limit), =quad to the four-lane kernel
static const int g_t10 = [] { const char *e = getenv("MPCX_T10"); return !e ? 0 : (!strcmp(e, "pair") ? 1 : (!strcmp(e, "quad") ? 2 : 0)); }();
#else
constexpr bool g_force_single_lane = false;
constexpr int g_stagewise = -1;
constexpr int g_t10 = 0;
#endif

// grow-only per-device workspace of the stage-wise kernel
static std::mutex g_ws_mutex;
static double *g_ws[64] = {nullptr};
static size_t g_ws_bytes[64] = {0};
// the workspace has one user at a time: a call waits (on the device, not the host) for the
// previous call's last launch, whatever stream that ran on
static hipEvent_t g_ws_event[64] = {nullptr};
static bool g_ws_event_set[64] = {false};

static int64_t stage_doubles_per_instance(int T) { return stagewise::Store::doubles_per_instance(T); }
// instances per launch of the stage-wise kernel (bounds the workspace)
static int64_t stage_chunk(int64_t B)
{
    const int64_t cap = 262144;
    return B < cap ? ((B + kWave - 1) / kWave) * kWave : cap;
}

// Largest batch the wavefront-per-instance kernel takes.  In general: what the chip holds at once
// (256 CUs, per CU as many wavefronts as LDS and registers allow, in whole wavefronts per SIMD: a
// fifth wavefront on a CU shares a SIMD with another and doubles the launch's time); one such round
// takes less than the latency of a single stage-wise wavefront (measured, warm: T = 12, 3,072: 0.36 ms
// against 0.54 ms), a second round does not.  At T = 30, where the instantiation with the matrix
// instructions runs (one wavefront per SIMD, 1,024 instances in 0.21 ms), eight rounds still do:
// 8,192 instances warm 1.40 ms against 1.54 ms stage-wise (mpc_dsr.py's Q 2^t variant 2.6 against
// 4.1 ms), 12,288 2.08 against 1.56 ms (scripts/wave_crossover.py).
static int64_t wave_batch_max(int T)
{
    if (T == 30) return 8 * 1024;
    const int64_t by_lds = (160 * 1024) / ((int64_t)wave_lds_doubles(T) * (int64_t)sizeof(double));
    const int64_t by_regs = 12;   // 3 wavefronts per SIMD at the kernel's register count
    int64_t per_cu = by_lds < by_regs ? by_lds : by_regs;
    if (per_cu > 4) per_cu = per_cu / 4 * 4;
    return 256 * per_cu;
}

enum SolvePath { kPathPair5, kPathReg5, kPathPair10, kPathQuad10, kPathWave, kPathStage };

// Largest batch the T = 10 lane-pair kernel (test-hook library only) is given: it spills 1.6 KB per
// lane and two of its wavefronts fill a CU's LDS, so the chip holds 16,384 instances at once.
constexpr int64_t kPair10BatchMax = 32768;

// T = 5 with every bound present: lane-pair kernel, chained or not.  T = 10 (mpc.py) with both
// boxes present, unchained: four lanes per instance at every batch size (one wavefront of 16
// instances takes 0.10-0.12 ms, 65,536 instances 0.47 ms warm; the wavefront-per-instance kernel
// needs 0.18-0.21 ms for what the chip holds at once, the stage-wise kernel 1.16 ms for 65,536).
// Other horizons, unchained, up to T = 32: batches that the chip holds at once go one wavefront per
// instance (lowest latency: 64 lanes share one instance).  Everything else one lane per instance
// with stage-wise linear algebra (far higher throughput once the chip is filled).
static SolvePath choose_path(const mpcx_variant *v, int64_t B, bool chained)
{
    if (uses_registers(v)) return g_force_single_lane ? kPathReg5 : kPathPair5;
    const bool wave_fits = v->T <= 32 && !chained;
    if (g_stagewise == 1) return kPathStage;
    if (g_stagewise == 0 && wave_fits) return kPathWave;
    const bool t10 = v->T == 10 && boxes_present(v) && !chained && !g_force_single_lane;
    if (t10 && g_t10 == 1 && B <= kPair10BatchMax) return kPathPair10;
    if (t10) return kPathQuad10;
    if (wave_fits && B <= wave_batch_max(v->T)) return kPathWave;
    return kPathStage;
}

// The stage-wise kernel over a batch, `chunk` instances (a multiple of 64) per launch, every
// launch using the same workspace `ws` (launches on one stream run one after the other).
static int launch_stage_chunks(const SolveArgs &a, hipStream_t st, double *ws, int64_t chunk, bool redo = false)
{
    const mpcx_variant *v = &a.v;
    const int NU = v->nu, T = v->T, W = T + 1;
    const int m = v->model == MPCX_MODEL_DIFF_DRIVE ? 0 : 1;
    const dim3 block(kWave);
    for (int64_t off = 0; off < a.B; off += chunk) {
        SolveArgs c = a;
        c.B = a.B - off < chunk ? a.B - off : chunk;
        c.x0 = a.x0 + off * 3;
        c.xref = a.xref + off * 3 * W;
        if (a.xbar) c.xbar = a.xbar + off * 3 * W;
        if (a.uin) c.uin = a.uin + off * NU * T;
        if (a.has_prev) c.has_prev = a.has_prev + off;
        c.u_out = a.u_out + off * NU * T;
        c.x_out = a.x_out + off * 3 * W;
        c.obj_out = a.obj_out + off;
        c.status_out = a.status_out + off;
        if (a.iters_out) c.iters_out = a.iters_out + off;
        const dim3 grid((unsigned)((c.B + kWave - 1) / kWave));
        if (redo) {
            if (m == 0) hipLaunchKernelGGL((k_solve_stage<0, true>), grid, block, 0, st, c, ws);
            else hipLaunchKernelGGL((k_solve_stage<1, true>), grid, block, 0, st, c, ws);
        } else {
            if (m == 0) hipLaunchKernelGGL((k_solve_stage<0>), grid, block, 0, st, c, ws);
            else hipLaunchKernelGGL((k_solve_stage<1>), grid, block, 0, st, c, ws);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "k_solve_stage launch");
    }
    return 0;
}

// Workspace of the entry point that allocates nothing: a fixed array in the library's own
// device image (every GPU the library runs on holds its copy from the moment the code object is
// loaded until the process ends).  In host code the array's name is only a handle for
// hipGetSymbolAddress: the address kernels and memsets are given is looked up once per device,
// with that device current, and kept.
constexpr size_t kScratchBytes = (size_t)64 << 20;
static __device__ __align__(256) double scratch[kScratchBytes / sizeof(double)];

// 0 when the stage-wise kernel's arrays for B instances fit the fixed array of the entry point that
// allocates nothing, otherwise the error it returns.
static int fits_fixed_workspace(int64_t B, int T)
{
    const int64_t slots = ((B + kWave - 1) / kWave) * kWave;
    const size_t per = (size_t)stage_doubles_per_instance(T) * sizeof(double);
    const size_t need = per * (size_t)slots;
    if (need <= kScratchBytes) return 0;
    return fail(MPCX_E_NOMEM, "%lld instances at T=%d need %zu bytes of workspace; the entry point that "
                "allocates nothing holds %zu (at most %lld instances at this horizon)", (long long)B, T,
                need, kScratchBytes, (long long)(kScratchBytes / per / kWave * kWave));
}
static double *g_scratch_ptr[64] = {nullptr};
static std::mutex g_scratch_mutex;

static int scratch_of_device(int dev, double **out)
{
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[dev]) {
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));   // the current device's copy
        if (e != hipSuccess || !p) return hip_fail(e, "hipGetSymbolAddress(scratch)");
        g_scratch_ptr[dev] = (double *)p;
    }
    *out = g_scratch_ptr[dev];
    return 0;
}

static int launch_solve(const SolveArgs &a, hipStream_t st, bool fixed_workspace = false)
{
    const mpcx_variant *v = &a.v;
    if (a.B == 0) return 0;
    const bool chained = !a.linear_mode && v->chain > 1;
    const dim3 block(kWave);
    clear_stale_error();
    const int m = v->model == MPCX_MODEL_DIFF_DRIVE ? 0 : 1;
    bool redo = false;
    if ((a.range || a.vtab) && choose_path(v, a.B, chained) != kPathPair5)
        return fail(MPCX_E_UNSUPPORTED, "a device-resident instance range is taken by the T = 5 lane-pair kernel only");
    switch (choose_path(v, a.B, chained)) {
    case kPathPair5: {
        const dim3 pgrid((unsigned)((2 * a.B + kWave - 1) / kWave));
        if (a.vtab) {
            // every run of the mode-sorted batch starts a wavefront of its own
            const dim3 mgrid(pgrid.x + (unsigned)a.n_modes);
            if (m == 0) return fail(MPCX_E_UNSUPPORTED, "a batch sorted by mode is taken for the dual-steer model only");
            if (chained) hipLaunchKernelGGL((k_solve_pair5<1, true, 2>), mgrid, block, 0, st, a);
            else hipLaunchKernelGGL((k_solve_pair5<1, false, 2>), mgrid, block, 0, st, a);
        } else if (a.range) {
            if (m == 0) {
                if (chained) hipLaunchKernelGGL((k_solve_pair5<0, true, 1>), pgrid, block, 0, st, a);
                else hipLaunchKernelGGL((k_solve_pair5<0, false, 1>), pgrid, block, 0, st, a);
            } else {
                if (chained) hipLaunchKernelGGL((k_solve_pair5<1, true, 1>), pgrid, block, 0, st, a);
                else hipLaunchKernelGGL((k_solve_pair5<1, false, 1>), pgrid, block, 0, st, a);
            }
        } else if (m == 0) {
            if (chained) hipLaunchKernelGGL((k_solve_pair5<0, true>), pgrid, block, 0, st, a);
            else hipLaunchKernelGGL((k_solve_pair5<0, false>), pgrid, block, 0, st, a);
        } else {
            if (chained) hipLaunchKernelGGL((k_solve_pair5<1, true>), pgrid, block, 0, st, a);
            else hipLaunchKernelGGL((k_solve_pair5<1, false>), pgrid, block, 0, st, a);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "k_solve_pair launch");
        return 0;
    }
    case kPathReg5: {
#ifdef MPCX_TEST_HOOKS
        const dim3 grid((unsigned)((a.B + kWave - 1) / kWave));
        if (m == 0) {
            if (chained) hipLaunchKernelGGL((k_solve_reg<0, 5, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_solve_reg<0, 5, false>), grid, block, 0, st, a);
        } else {
            if (chained) hipLaunchKernelGGL((k_solve_reg<1, 5, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_solve_reg<1, 5, false>), grid, block, 0, st, a);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "k_solve_reg launch");
        return 0;
#else
        return fail(MPCX_E_UNSUPPORTED, "the one-lane-per-instance kernel is part of the test-hook library only");
#endif
    }
    case kPathPair10: {
#ifdef MPCX_TEST_HOOKS
        const dim3 pgrid((unsigned)((2 * a.B + kWave - 1) / kWave));
        if (m == 0) hipLaunchKernelGGL((k_solve_pair10<0, false>), pgrid, block, 0, st, a);
        else hipLaunchKernelGGL((k_solve_pair10<1, false>), pgrid, block, 0, st, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "k_solve_pair10 launch");
        return 0;
#else
        return fail(MPCX_E_UNSUPPORTED, "the T = 10 lane-pair kernel is part of the test-hook library only");
#endif
    }
    case kPathQuad10: {
        const dim3 qgrid((unsigned)((a.B + quad::kInst - 1) / quad::kInst));
        if (m == 0) hipLaunchKernelGGL((k_solve_quad<0, 10>), qgrid, block, 0, st, a);
        else hipLaunchKernelGGL((k_solve_quad<1, 10>), qgrid, block, 0, st, a);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "k_solve_quad launch");
        return 0;
    }
    case kPathWave: {
        const bool second_pass = !(g_stagewise == 0 || (a.uin && (const void *)a.uin == (const void *)a.u_out));
        // the caller that may not allocate is refused before anything is launched
        if (second_pass && fixed_workspace)
            if (int rc = fits_fixed_workspace(a.B, v->T)) return rc;
        // K and H of the instance in LDS
        const size_t lds = (size_t)wave_lds_doubles(v->T) * sizeof(double);
        const dim3 wgrid((unsigned)a.B);
        if (v->T == 30) {   // mpc_dsr.py's long horizon: the instantiation on the matrix cores
            if (m == 0) hipLaunchKernelGGL((k_solve_wave<0, 60>), wgrid, block, lds, st, a);
            else hipLaunchKernelGGL((k_solve_wave<1, 60>), wgrid, block, lds, st, a);
        } else {
            if (m == 0) hipLaunchKernelGGL((k_solve_wave<0>), wgrid, block, lds, st, a);
            else hipLaunchKernelGGL((k_solve_wave<1>), wgrid, block, lds, st, a);
        }
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "k_solve_wave launch");
        // About one instance in 10^5 to 10^6 of hard random controllers is left INACCURATE by this
        // kernel (its interior-point iterate short of tolerance and its active-set rounds not
        // settled) where the stage-wise kernel reaches the optimum: a second, stage-wise pass
        // solves those again from the same inputs, which it can only do while the incoming
        // controls have not been overwritten by the answer.  A wavefront of that pass that finds
        // nothing to do returns at once.  Forcing this kernel through the test hook shows it raw.
        if (!second_pass) return 0;
        redo = true;
        break;
    }
    case kPathStage:
        break;
    }
    // stage-wise kernel: per-lane arrays in a workspace
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return hip_fail(e, "hipGetDevice");
    if (dev < 0 || dev >= 64) return fail(MPCX_E_UNSUPPORTED, "device index %d", dev);
    if (fixed_workspace) {
        // the caller may not allocate: the library's own fixed array, one launch, no event
        const int64_t slots = ((a.B + kWave - 1) / kWave) * kWave;
        if (int rc = fits_fixed_workspace(a.B, v->T)) return rc;
        double *ws = nullptr;
        if (int rc = scratch_of_device(dev, &ws)) return rc;
        return launch_stage_chunks(a, st, ws, slots, redo);
    }
    // launches chunked so that the grow-only buffer stays bounded
    const int64_t chunk = stage_chunk(a.B);
    const size_t need = (size_t)stage_doubles_per_instance(v->T) * (size_t)chunk * sizeof(double);
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    if (g_ws_bytes[dev] < need) {
        if (g_ws[dev]) {
            e = hipDeviceSynchronize();   // earlier launches may still use the old buffer
            if (e != hipSuccess) return hip_fail(e, "hipDeviceSynchronize");
            (void)hipFree(g_ws[dev]);
            g_ws[dev] = nullptr;
            g_ws_bytes[dev] = 0;
        }
        e = hipMalloc((void **)&g_ws[dev], need);
        if (e != hipSuccess) return fail(MPCX_E_NOMEM, "workspace of %zu bytes: %s", need, hipGetErrorString(e));
        g_ws_bytes[dev] = need;
    }
    if (!g_ws_event[dev]) {
        e = hipEventCreateWithFlags(&g_ws_event[dev], hipEventDisableTiming);
        if (e != hipSuccess) return hip_fail(e, "hipEventCreate");
    }
    if (g_ws_event_set[dev]) {
        e = hipStreamWaitEvent(st, g_ws_event[dev], 0);
        if (e != hipSuccess) return hip_fail(e, "hipStreamWaitEvent");
    }
    if (int rc = launch_stage_chunks(a, st, g_ws[dev], chunk, redo)) return rc;
    e = hipEventRecord(g_ws_event[dev], st);
    if (e != hipSuccess) return hip_fail(e, "hipEventRecord");
    g_ws_event_set[dev] = true;
    return 0;
}

}  // namespace mpcx

using namespace mpcx;

extern "C" {

int mpcx_abi_version(void) { return MPCX_ABI_VERSION; }

const char *mpcx_last_error(void) { return g_err; }

int mpcx_variant_by_name(const char *name, mpcx_variant *out)
{
    if (!name || !out) return fail(MPCX_E_ARG, "NULL argument");
    const double inf = INFINITY, d45 = M_PI / 4.0, d90 = M_PI / 2.0;
    mpcx_variant v;
    memset(&v, 0, sizeof(v));
    // shared dual-steer defaults (mpc_dsr.py:20-48, mpc_sim.py:167-195)
    v.model = MPCX_MODEL_DSR; v.nu = 4; v.T = 5; v.chain = 1; v.dt = 0.2; v.wb = 1.0;
    for (int i = 0; i < 4; ++i) { v.R[i] = 0.01; v.Rd[i] = 0.0; v.umax[i] = inf; v.dumax[i] = inf; v.first_step[i] = inf; }
    v.Q[0] = 1.0; v.Q[1] = 1.0; v.Q[2] = 0.5; v.q_growth = 1.0;
    v.Qf[0] = 1.0; v.Qf[1] = 1.0; v.Qf[2] = 0.5;
    auto steer_mode = [&](int eq, double vmax, double smax, double dv, double dsteer) {
        v.eq_mode = eq;
        v.umax[0] = v.umax[2] = vmax; v.umax[1] = v.umax[3] = smax;
        v.dumax[0] = v.dumax[2] = dv; v.dumax[1] = v.dumax[3] = dsteer;
        v.first_step[1] = v.first_step[3] = dsteer;
    };
    auto fixed_mode = [&](double vmax, double dv) {
        v.eq_mode = MPCX_EQ_FIXED_STEER;
        v.steer_fixed = d90;
        v.umax[0] = v.umax[2] = vmax;
        v.dumax[0] = v.dumax[2] = dv;
    };
    if (!strcmp(name, "diff_drive")) {            // mpc_diff_drive.py:19-48,170-217
        v.model = MPCX_MODEL_DIFF_DRIVE; v.nu = 2; v.eq_mode = MPCX_EQ_NONE;
        v.umax[0] = 0.47; v.umax[1] = 3.77; v.dumax[0] = 0.2; v.dumax[1] = 1.0;
        v.cold_start[0] = 0.3;
        v.R[2] = v.R[3] = 0.0;
    } else if (!strcmp(name, "dsr_ackermann")) {  // mpc_dsr.py:201-267
        steer_mode(MPCX_EQ_ACKERMANN, 0.47, d90, 0.5 * 0.2, d45 * 0.2);
    } else if (!strcmp(name, "dsr_crab")) {       // mpc_dsr.py:269-335
        steer_mode(MPCX_EQ_CRAB, 0.47, d90, 0.5 * 0.2, d45 * 0.2);
    } else if (!strcmp(name, "dsr_diff")) {       // mpc_dsr.py:337-395
        fixed_mode(0.47, 0.5 * 0.2);
        v.q_growth = 2.0;
    } else if (!strcmp(name, "sim_ackermann")) {  // mpc_sim.py:312-380
        steer_mode(MPCX_EQ_ACKERMANN, 0.47, d45, 0.2 * 0.2, d45 * 0.2);
        v.chain = 5;
    } else if (!strcmp(name, "sim_diff")) {       // mpc_sim.py:382-440
        fixed_mode(0.47, 0.2 * 0.2);
        v.chain = 5;
    } else if (!strcmp(name, "sim_crab")) {       // mpc_sim.py:442-508
        steer_mode(MPCX_EQ_CRAB, 0.5, d45, 0.2 * 0.2, d45 * 0.2);
        v.chain = 5;
    } else if (!strcmp(name, "mpc_py")) {         // mpc.py:33-47,117-187
        v.T = 10; v.dt = 0.1; v.eq_mode = MPCX_EQ_ACKERMANN;
        for (int i = 0; i < 4; ++i) { v.R[i] = 0.001; v.Rd[i] = 0.001; }
        v.Q[0] = v.Q[1] = v.Q[2] = 0.1;
        v.Qf[0] = v.Qf[1] = 0.1; v.Qf[2] = 0.0;
        v.umax[0] = v.umax[2] = 0.5; v.umax[1] = v.umax[3] = 3.14;
        v.dumax[1] = v.dumax[3] = 3 * 0.1;
        v.first_step[1] = v.first_step[3] = 3 * 0.1;
    } else {
        return fail(MPCX_E_ARG, "unknown variant '%s'", name);
    }
    *out = v;
    return 0;
}

int64_t mpcx_workspace_bytes(const mpcx_variant *v, int64_t B)
{
    if (check_variant(v) != 0 || B < 0) return MPCX_E_ARG;
    const SolvePath path = choose_path(v, B, v->chain > 1);
    // the wavefront-per-instance kernel is followed by a stage-wise pass over what it left unsettled
    if (path != kPathStage && !(path == kPathWave && g_stagewise != 0)) return 0;   // registers / LDS only
    return stage_doubles_per_instance(v->T) * stage_chunk(B) * (int64_t)sizeof(double);
}

int mpcx_solve_batch(const mpcx_variant *v, int64_t B, const double *x0, const double *xref,
                     const double *uprev, const uint8_t *has_prev, double *u_out, double *x_out,
                     double *obj_out, int32_t *status_out, int32_t *iters_out, void *stream)
{
    int rc = check_variant(v);
    if (rc) return rc;
    if (B < 0) return fail(MPCX_E_ARG, "negative batch");
    if (B > 0 && (!x0 || !xref || !u_out || !x_out || !obj_out || !status_out))
        return fail(MPCX_E_ARG, "NULL buffer");
    SolveArgs a;
    memset(&a, 0, sizeof(a));
    a.v = *v; a.B = B; a.x0 = x0; a.xref = xref; a.xbar = nullptr; a.uin = uprev; a.has_prev = has_prev;
    a.u_out = u_out; a.x_out = x_out; a.obj_out = obj_out; a.status_out = status_out; a.iters_out = iters_out;
    a.linear_mode = 0;
    a.max_iter = 60;
    return launch_solve(a, (hipStream_t)stream);
}

// Also the set-up call of mpcx_solve_batch_noalloc for the current device: the address of that
// device's copy of the array is looked up here (loading the code object on the device if this
// is the library's first use of it), so that the solve call itself does no lookup.
int mpcx_solve_range_batch(const mpcx_variant *v, int64_t B, const int32_t *range, const double *x0, const double *xref,
                           const double *uprev, const uint8_t *has_prev, double *u_out, double *x_out,
                           double *obj_out, int32_t *status_out, int32_t *iters_out, void *stream)
{
    int rc = check_variant(v);
    if (rc) return rc;
    if (B < 0 || B > 2147483647LL) return fail(MPCX_E_ARG, "batch outside 0..2^31-1");
    if (!range) return fail(MPCX_E_ARG, "range is NULL");
    if (B > 0 && (!x0 || !xref || !u_out || !x_out || !obj_out || !status_out))
        return fail(MPCX_E_ARG, "NULL buffer");
    SolveArgs a;
    memset(&a, 0, sizeof(a));
    a.v = *v; a.B = B; a.x0 = x0; a.xref = xref; a.xbar = nullptr; a.uin = uprev; a.has_prev = has_prev;
    a.u_out = u_out; a.x_out = x_out; a.obj_out = obj_out; a.status_out = status_out; a.iters_out = iters_out;
    a.linear_mode = 0;
    a.max_iter = 60;
    a.range = range;
    return launch_solve(a, (hipStream_t)stream);
}

int mpcx_solve_modes_batch(const mpcx_variant *variants, int32_t n_modes, const mpcx_variant *variants_dev,
                           const int32_t *bounds, int64_t B, const double *x0, const double *xref, const double *uprev,
                           const uint8_t *has_prev, double *u_out, double *x_out, double *obj_out, int32_t *status_out,
                           int32_t *iters_out, void *stream)
{
    if (!variants || n_modes < 1 || n_modes > 16) return fail(MPCX_E_ARG, "1..16 modes");
    for (int32_t m = 0; m < n_modes; ++m) {
        if (int rc = check_variant(&variants[m])) return rc;
        const mpcx_variant &a = variants[0], &b = variants[m];
        if (a.model != b.model || a.nu != b.nu || a.T != b.T || (a.chain > 1) != (b.chain > 1) ||
            (a.chain > 1 && a.chain != b.chain))
            return fail(MPCX_E_ARG, "the modes of one launch share model, horizon and chain length");
        if (!uses_registers(&b))
            return fail(MPCX_E_UNSUPPORTED, "a batch sorted by mode is taken by the T = 5 lane-pair kernel only");
    }
    if (B < 0 || B > 2147483647LL) return fail(MPCX_E_ARG, "batch outside 0..2^31-1");
    if (!variants_dev || !bounds) return fail(MPCX_E_ARG, "variants_dev / bounds is NULL");
    if (B > 0 && (!x0 || !xref || !u_out || !x_out || !obj_out || !status_out))
        return fail(MPCX_E_ARG, "NULL buffer");
    SolveArgs a;
    memset(&a, 0, sizeof(a));
    a.v = variants[0]; a.B = B; a.x0 = x0; a.xref = xref; a.xbar = nullptr; a.uin = uprev; a.has_prev = has_prev;
    a.u_out = u_out; a.x_out = x_out; a.obj_out = obj_out; a.status_out = status_out; a.iters_out = iters_out;
    a.linear_mode = 0;
    a.max_iter = 60;
    a.range = bounds;
    a.vtab = variants_dev;
    a.n_modes = n_modes;
    return launch_solve(a, (hipStream_t)stream);
}

int64_t mpcx_mode_sort_scratch_ints(int64_t B, int32_t n_modes)
{
    if (B < 0 || n_modes < 1 || n_modes + 1 > kSortMaxKeys) return MPCX_E_ARG;
    const int64_t nblocks = (B + kSortBlock - 1) / kSortBlock;
    if (nblocks * (n_modes + 1) > kSortScanMax) return MPCX_E_UNSUPPORTED;
    return nblocks * (n_modes + 1);
}

int mpcx_mode_sort_batch(int64_t B, int32_t n_modes, int32_t nu, int32_t T, const uint8_t *finished, const uint8_t *done,
                         const int32_t *course_id, const int32_t *seg_mode, int64_t n_courses, const double *x0,
                         const double *xref, const double *uprev, const uint8_t *has_prev, uint8_t *was_active,
                         int32_t *order, int32_t *bounds, double *x0_sorted, double *xref_sorted, double *uprev_sorted,
                         uint8_t *has_prev_sorted, int32_t *scratch, void *stream)
{
    if (B < 0 || n_modes < 1 || n_modes + 1 > kSortMaxKeys || n_courses < 1 || nu < 1 || T < 1)
        return fail(MPCX_E_ARG, "bad sizes (at most %d modes)", kSortMaxKeys - 1);
    const int64_t nblocks = (B + kSortBlock - 1) / kSortBlock;
    if (nblocks * (n_modes + 1) > kSortScanMax)
        return fail(MPCX_E_UNSUPPORTED, "a fleet of %lld vehicles in %d modes is beyond the one-block scan", (long long)B, n_modes);
    if (!finished || !done || !course_id || !seg_mode || !was_active || !order || !bounds || !has_prev_sorted || !scratch ||
        (x0 && !x0_sorted) || (xref && !xref_sorted) || (uprev && !uprev_sorted))
        return fail(MPCX_E_ARG, "NULL buffer");
    hipStream_t st = (hipStream_t)stream;
    clear_stale_error();
    if (B > 0) {
        hipLaunchKernelGGL(k_mode_count, dim3((unsigned)nblocks), dim3(kSortBlock), 0, st, B, (int)n_modes, finished, done, course_id,
                           seg_mode, n_courses, was_active, scratch);
    }
    hipLaunchKernelGGL(k_mode_scan, dim3(1), dim3(256), 0, st, (int)nblocks, (int)n_modes, scratch, bounds);
    if (B > 0) {
        hipLaunchKernelGGL(k_mode_gather, dim3((unsigned)nblocks), dim3(kSortBlock), 0, st, B, (int)n_modes, finished, done,
                           course_id, seg_mode, n_courses, (const int32_t *)scratch, order, 3, x0, x0_sorted, 3 * (T + 1), xref,
                           xref_sorted, nu * T, uprev, uprev_sorted, has_prev, has_prev_sorted);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "mode sort launch");
}

int mpcx_mode_unsort_batch(int64_t B, int32_t nu, int32_t T, const int32_t *order, const double *u_sorted,
                           const int32_t *status_sorted, double *u, int32_t *status, void *stream)
{
    if (B < 0 || nu < 1 || T < 1) return fail(MPCX_E_ARG, "bad sizes");
    if (B == 0) return 0;
    if (!order || !u_sorted || !status_sorted || !u || !status) return fail(MPCX_E_ARG, "NULL buffer");
    clear_stale_error();
    hipLaunchKernelGGL(k_mode_scatter, dim3((unsigned)((B + kSortBlock - 1) / kSortBlock)), dim3(kSortBlock), 0, (hipStream_t)stream,
                       B, order, (int)(nu * T), u_sorted, u, status_sorted, status);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "k_mode_scatter launch");
}

int mpcx_segment_step_batch(int64_t B, int32_t max_ticks_per_segment, const uint8_t *was_active, const int32_t *seg_start,
                            const int32_t *seg_count, int32_t *seg, int32_t *seg_ticks, uint8_t *finished, uint8_t *done,
                            uint8_t *has_prev, int32_t *course_id, double *state, const double *cyaw, int64_t n_courses,
                            int64_t Lmax, const int32_t *target_ind, int32_t *pind_next, uint8_t *enter, void *stream)
{
    if (B < 0 || n_courses < 1 || Lmax < 1 || max_ticks_per_segment < 1) return fail(MPCX_E_ARG, "bad sizes");
    if (B == 0) return 0;
    if (!was_active || !seg_start || !seg_count || !seg || !seg_ticks || !finished || !done || !has_prev || !course_id ||
        !state || !cyaw || !target_ind || !pind_next || !enter)
        return fail(MPCX_E_ARG, "NULL buffer");
    clear_stale_error();
    hipLaunchKernelGGL(k_segment_step, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B,
                       (int)max_ticks_per_segment, was_active, seg_start, seg_count, seg, seg_ticks, finished, done, has_prev,
                       course_id, state, cyaw, n_courses, Lmax, target_ind, pind_next, enter);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "k_segment_step launch");
}

int64_t mpcx_static_workspace_bytes(void)
{
    int dev = 0;
    double *p = nullptr;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) (void)scratch_of_device(dev, &p);
    return (int64_t)kScratchBytes;
}

int mpcx_solve_batch_noalloc(const mpcx_variant *v, int64_t B, const double *x0, const double *xref,
                             const double *uprev, const uint8_t *has_prev, double *u_out, double *x_out,
                             double *obj_out, int32_t *status_out, int32_t *iters_out, void *stream)
{
    int rc = check_variant(v);
    if (rc) return rc;
    if (B < 0) return fail(MPCX_E_ARG, "negative batch");
    if (B > 0 && (!x0 || !xref || !u_out || !x_out || !obj_out || !status_out))
        return fail(MPCX_E_ARG, "NULL buffer");
    SolveArgs a;
    memset(&a, 0, sizeof(a));
    a.v = *v; a.B = B; a.x0 = x0; a.xref = xref; a.xbar = nullptr; a.uin = uprev; a.has_prev = has_prev;
    a.u_out = u_out; a.x_out = x_out; a.obj_out = obj_out; a.status_out = status_out; a.iters_out = iters_out;
    a.linear_mode = 0;
    a.max_iter = 60;
    return launch_solve(a, (hipStream_t)stream, true);
}

int mpcx_linear_mpc_batch(const mpcx_variant *v, int64_t B, const double *xref, const double *xbar,
                          const double *x0, const double *uref, double *u_out, double *x_out,
                          double *obj_out, int32_t *status_out, int32_t *iters_out, void *stream)
{
    int rc = check_variant(v);
    if (rc) return rc;
    if (B < 0) return fail(MPCX_E_ARG, "negative batch");
    if (B > 0 && (!x0 || !xref || !xbar || !uref || !u_out || !x_out || !obj_out || !status_out))
        return fail(MPCX_E_ARG, "NULL buffer");
    SolveArgs a;
    memset(&a, 0, sizeof(a));
    a.v = *v; a.B = B; a.x0 = x0; a.xref = xref; a.xbar = xbar; a.uin = uref; a.has_prev = nullptr;
    a.u_out = u_out; a.x_out = x_out; a.obj_out = obj_out; a.status_out = status_out; a.iters_out = iters_out;
    a.linear_mode = 1;
    a.max_iter = 60;
    return launch_solve(a, (hipStream_t)stream);
}

int mpcx_rollout_batch(const mpcx_variant *v, int64_t B, const double *x0, const double *u, double *xbar,
                       void *stream)
{
    int rc = check_variant(v);
    if (rc) return rc;
    if (B < 0) return fail(MPCX_E_ARG, "negative batch");
    if (B == 0) return 0;
    if (!x0 || !u || !xbar) return fail(MPCX_E_ARG, "NULL buffer");
    const dim3 grid((unsigned)((B + 255) / 256)), block(256);
    clear_stale_error();
    if (v->model == MPCX_MODEL_DIFF_DRIVE)
        hipLaunchKernelGGL((k_rollout<0>), grid, block, 0, (hipStream_t)stream, *v, B, x0, u, xbar);
    else
        hipLaunchKernelGGL((k_rollout<1>), grid, block, 0, (hipStream_t)stream, *v, B, x0, u, xbar);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "k_rollout launch");
}

int mpcx_linearize_batch(const mpcx_variant *v, int64_t B, const double *theta, const double *u, double *A,
                         double *Bm, double *C, void *stream)
{
    int rc = check_variant(v);
    if (rc) return rc;
    if (B < 0) return fail(MPCX_E_ARG, "negative batch");
    if (B == 0) return 0;
    if (!theta || !u || !A || !Bm || !C) return fail(MPCX_E_ARG, "NULL buffer");
    const dim3 grid((unsigned)((B + 255) / 256)), block(256);
    clear_stale_error();
    if (v->model == MPCX_MODEL_DIFF_DRIVE)
        hipLaunchKernelGGL((k_linearize<0>), grid, block, 0, (hipStream_t)stream, *v, B, theta, u, A, Bm, C);
    else
        hipLaunchKernelGGL((k_linearize<1>), grid, block, 0, (hipStream_t)stream, *v, B, theta, u, A, Bm, C);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "k_linearize launch");
}

int mpcx_nearest_ref_batch(int32_t mode, int32_t T, double dt, double param, int32_t n_search, int32_t sign,
                           int64_t B, const double *state, const int32_t *course_id, const int32_t *pind,
                           const double *cx, const double *cy, const double *cyaw, const int32_t *clen,
                           int64_t n_courses, int64_t Lmax, int32_t *ind_out, double *mind_out,
                           double *xref_out, double *x0_out, void *stream)
{
    if (mode != 0 && mode != 1) return fail(MPCX_E_ARG, "mode must be 0 or 1");
    if (T < 1 || T > MPCX_MAX_T) return fail(MPCX_E_ARG, "horizon T=%d outside 1..%d", T, MPCX_MAX_T);
    if (n_search < 1) return fail(MPCX_E_ARG, "n_search must be >= 1");
    if (mode == 0 && !(param > 0.0)) return fail(MPCX_E_ARG, "dl must be positive");
    if (mode == 1 && !(param >= 1.0)) return fail(MPCX_E_ARG, "n_search_ind must be >= 1");
    if (B < 0 || n_courses < 1 || Lmax < 1) return fail(MPCX_E_ARG, "bad sizes");
    if (B == 0) return 0;
    if (!state || !pind || !cx || !cy || !cyaw || !clen || !ind_out) return fail(MPCX_E_ARG, "NULL buffer");
    const int64_t threads = B * kGroup;
    const dim3 grid((unsigned)((threads + 255) / 256)), block(256);
    clear_stale_error();
    hipLaunchKernelGGL(k_nearest_ref, grid, block, 0, (hipStream_t)stream, mode, T, dt, param, n_search, sign, B,
                       state, course_id, pind, cx, cy, cyaw, clen, n_courses, Lmax, ind_out, mind_out, xref_out, x0_out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "k_nearest_ref launch");
}


int mpcx_advance_batch(const mpcx_variant *v, int64_t B, double goal_dis, double stop_speed, const double *u,
                       const int32_t *status, const int32_t *target_ind, const int32_t *course_id,
                       const double *cx, const double *cy, const int32_t *clen, int64_t n_courses, int64_t Lmax,
                       double *state, double *twist, uint8_t *has_prev, uint8_t *done, int32_t *ticks,
                       double *log_out, void *stream)
{
    if (int rc = check_variant(v)) return rc;
    if (B < 0 || n_courses < 1 || Lmax < 1) return fail(MPCX_E_ARG, "bad sizes");
    if (B == 0) return 0;
    if (!u || !status || !target_ind || !cx || !cy || !clen || !state || !has_prev || !done || !ticks)
        return fail(MPCX_E_ARG, "NULL buffer");
    const dim3 grid((unsigned)((B + 255) / 256)), block(256);
    clear_stale_error();
    if (v->model == MPCX_MODEL_DIFF_DRIVE)
        hipLaunchKernelGGL(k_advance<MPCX_MODEL_DIFF_DRIVE>, grid, block, 0, (hipStream_t)stream, *v, B, goal_dis,
                           stop_speed, u, status, target_ind, course_id, cx, cy, clen, n_courses, Lmax, state, twist, has_prev,
                           done, ticks, log_out);
    else
        hipLaunchKernelGGL(k_advance<MPCX_MODEL_DSR>, grid, block, 0, (hipStream_t)stream, *v, B, goal_dis,
                           stop_speed, u, status, target_ind, course_id, cx, cy, clen, n_courses, Lmax, state, twist, has_prev,
                           done, ticks, log_out);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "k_advance launch");
}


int mpcx_build_courses_batch(int64_t K, int32_t n_way_max, const double *wx, const double *wy, const int32_t *nway,
                             double ds, int64_t Lmax, int32_t smooth, double *cx, double *cy, double *cyaw,
                             double *ck, int32_t *clen, void *stream)
{
    if (K < 0 || n_way_max < 2 || n_way_max > kMaxWaypoints) return fail(MPCX_E_ARG, "waypoints per course must be 2..%d", kMaxWaypoints);
    if (!(ds > 0.0) || Lmax < 1) return fail(MPCX_E_ARG, "ds must be positive and Lmax >= 1");
    if (K == 0) return 0;
    if (!wx || !wy || !nway || !cx || !cy || !cyaw || !clen) return fail(MPCX_E_ARG, "NULL buffer");
    clear_stale_error();
    hipLaunchKernelGGL(k_build_courses, dim3((unsigned)K), dim3(kWave), 0, (hipStream_t)stream, K, (int)n_way_max, wx, wy,
                       nway, ds, Lmax, (int)smooth, cx, cy, cyaw, ck, clen);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : hip_fail(e, "k_build_courses launch");
}

}  // extern "C"

#ifdef MPCX_STAMPS
extern "C" int mpcx_debug_stamps(unsigned long long *host, int n_waves)
{
    if (n_waves > mpcx::kStampWaves) n_waves = mpcx::kStampWaves;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(mpcx::g_stamps), sizeof(unsigned long long) * n_waves * mpcx::kStampSlots)
                   == hipSuccess ? n_waves : -1;
}
#endif
