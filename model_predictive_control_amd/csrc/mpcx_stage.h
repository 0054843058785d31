// Stage-wise solver for long horizons: one instance per lane, O(T) work per Newton step.
//
// The condensed Hessian of the MPC problem (mpc_dsr.py:331-388 and siblings, states
// eliminated) is dense, n = 2T, but it is the Hessian of a linear-quadratic control problem,
// and so is every matrix K = H + (barrier or penalty terms on the box and rate constraints)
// the solver factorises: the box terms are diagonal per stage, the rate terms couple a stage
// to the one before it.  K dw = r is therefore solved exactly by a Riccati sweep over the
// stages with the state (dx_t, dw_{t-1}) in R^5 -- about 150 multiply-adds per stage instead
// of n^3 / 3 for a dense factorisation -- and H w + g by a forward rollout and an adjoint
// sweep.  Nothing of size n^2 is ever formed.
//
// Per-lane arrays (the iterate, the stage dynamics, the Riccati gains) live in a global
// workspace, one contiguous chunk per wavefront laid out [stage][field][lane]: every access of
// a wavefront is one contiguous 512-byte row and each sweep over the stages streams through the
// chunk in address order.  The solver itself is the one the other kernels use: a fixed number of predictor-corrector interior-point iterations to find
// the active set, primal-dual active-set rounds that certify the KKT conditions, and for
// whatever does not settle a converged interior-point solve and the rounds again.
#pragma once
#include "mpcx_device.h"

namespace mpcx {
namespace stagewise {

struct Store {
    // Layout: every wavefront owns one contiguous chunk of the workspace, [stage][field][lane]:
    // the record of stage t holds everything indexed by t (kFields doubles per lane), so a sweep
    // over the stages walks through the chunk front to back or back to front, each access one
    // contiguous 512-byte row.  Addressed as buffer loads/stores: the chunk's descriptor and
    // the row's byte offset are uniform (scalar registers), the lane's byte offset is the one
    // vector register shared by every access -- no 64-bit address arithmetic per access.
    enum Field {
        fDyn = 0,        // a02 a12 b00 b01 b10 b11 b20 b21 c0 c1 c2
        fXr = 11,        // reference of stage t + 1 (what the rollout of stage t is compared with)
        fQe = 14,        // 2 Q_{t+1} (x_{t+1} - xref_{t+1})
        fG = 17,         // gradient at w = 0
        fW = 19,
        fLU = 21, fLL = 23,
        fNU = 25, fNL = 27,   // rate multipliers: the record of stage t holds constraint (t, t + 1)
        fLx = 29, fLv = 35, fHi = 39,   // Riccati gains
        fKv = 42,        // feed-forward of the current right-hand side
        fDwa = 44, fDw = 46,
        fBak = 48,       // controls before the active-set rounds
        fSetB = 50,      // active-set flags of the boxes (1: upper, 2: lower)
        fSetR = 52,      // ... of the rate constraints (t, t + 1), same coding
        fUref = 54,      // linearisation point: controls and yaw (rewritten between chained solves)
        fTheta = 58,
        kFields = 59
    };
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t lane_bytes;
    int T;
    __host__ __device__ static int64_t doubles_per_instance(int T) { return (int64_t)kFields * T; }
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    // ws: the workspace; wave: index of this wavefront's chunk (uniform); lane: 0..63
    __device__ __forceinline__ void bind(double *ws, int T_, int64_t wave, int lane)
    {
        T = T_;
        const int64_t chunk = doubles_per_instance(T) * kWave;
        rsrc = __builtin_amdgcn_make_buffer_rsrc(ws + wave * chunk, 0, (int)(chunk * 8), 0x00020000);
        lane_bytes = (uint32_t)lane * 8u;
    }
    // element indices: field f of stage t; channel c of a two-channel field of stage t
    __device__ __forceinline__ int f(int field, int t) const { return t * kFields + field; }
    __device__ __forceinline__ int p(int field, int t, int c) const { return t * kFields + field + c; }
    __device__ __forceinline__ double ld(int o) const
    {
        const u32x2 r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)lane_bytes, o * (kWave * 8), 0);
        return __builtin_bit_cast(double, r);
    }
    __device__ __forceinline__ void st(int o, double x) const
    {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, x), rsrc, (int)lane_bytes, o * (kWave * 8), 0);
    }
    __device__ __forceinline__ double dyn(int fld, int t) const { return ld(f(fDyn + fld, t)); }
    __device__ __forceinline__ void dyn_st(int fld, int t, double x) const { st(f(fDyn + fld, t), x); }
};

// Everything about the instance that is not per stage.
struct Problem {
    FreeBounds fb;
    double rw[2], rdw[2];      // cost on the free controls and on their differences
    double q[3], qf[3], growth;
    double x0[3];
    double mu0;                // initial complementarity
    __device__ __forceinline__ bool hb(int t, int c) const { return t == 0 ? fb.box0[c] : fb.box[c]; }
    __device__ __forceinline__ double lo(int t, int c) const { return t == 0 ? fb.lo0[c] : -fb.b[c]; }
    __device__ __forceinline__ double hi(int t, int c) const { return t == 0 ? fb.hi0[c] : fb.b[c]; }
    __device__ __forceinline__ bool hr(int c) const { return fb.rate[c]; }
};

struct Dyn {
    double a02, a12, b[3][2];
};
__device__ __forceinline__ void load_dyn(const Store &s, int t, Dyn &d)
{
    d.a02 = s.dyn(0, t); d.a12 = s.dyn(1, t);
#pragma unroll
    for (int r = 0; r < 3; ++r) { d.b[r][0] = s.dyn(2 + 2 * r, t); d.b[r][1] = s.dyn(3 + 2 * r, t); }
}

// Forward rollout of the linearised dynamics, one stage per call; leaves
// 2 Q_{t+1} (x_{t+1} - xref_{t+1}) for the adjoint sweep.
struct Roller {
    double x[3], qg;
    __device__ __forceinline__ void start(const Problem &p) { x[0] = p.x0[0]; x[1] = p.x0[1]; x[2] = p.x0[2]; qg = 1.0; }
    __device__ __forceinline__ void step(const Store &s, const Problem &p, int t, double w0, double w1)
    {
        const int T = s.T;
        Dyn d;
        load_dyn(s, t, d);
        const double n0 = x[0] + d.a02 * x[2] + d.b[0][0] * w0 + d.b[0][1] * w1 + s.dyn(8, t);
        const double n1 = x[1] + d.a12 * x[2] + d.b[1][0] * w0 + d.b[1][1] * w1 + s.dyn(9, t);
        const double n2 = x[2] + d.b[2][0] * w0 + d.b[2][1] * w1 + s.dyn(10, t);
        x[0] = n0; x[1] = n1; x[2] = n2;
        const int ts = t + 1;
        if (ts < T) qg *= p.growth;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double q = ts < T ? p.q[r] * qg : p.qf[r];
            s.st(s.f(s.fQe + r, t), 2.0 * q * (x[r] - s.ld(s.f(s.fXr + r, ts - 1))));
        }
    }
};
__device__ __forceinline__ void roll_zero(const Store &s, const Problem &p)
{
    Roller ro;
    ro.start(p);
    _Pragma("nounroll") for (int t = 0; t < s.T; ++t) ro.step(s, p, t, 0.0, 0.0);
}

// The backward sweeps of the interior point and of the active-set rounds have one shape: from
// the last stage to the first they form hw_t = (H w + g)_t (adjoint of the rollout + the input
// cost terms), the right-hand side r_t of the phase, the Riccati gains of
// K = H + diag(box terms) + rate terms, and the feed-forward for r.
// Value function of stage t: V_t(dx, v) = 1/2 [dx v]' P [dx v] + p' [dx v], v = the previous
// stage's control step; P is kept as its blocks, the symmetric ones as upper triangles.
struct Riccati {
    double xx[6];      // Pxx: 00 01 02 11 12 22
    double xu[3][2];   // Pxu
    double uu[3];      // Puu: 00 01 11
    double px[3], pu[2];
    __device__ __forceinline__ void terminal(const double (&qf)[3])
    {
        xx[0] = 2.0 * qf[0]; xx[1] = 0.0; xx[2] = 0.0; xx[3] = 2.0 * qf[1]; xx[4] = 0.0; xx[5] = 2.0 * qf[2];
#pragma unroll
        for (int r = 0; r < 3; ++r) { xu[r][0] = xu[r][1] = 0.0; px[r] = 0.0; }
        uu[0] = uu[1] = uu[2] = 0.0;
        pu[0] = pu[1] = 0.0;
    }
    __device__ __forceinline__ double Pxx(int i, int j) const
    {
        const int a = i < j ? i : j, b = i < j ? j : i;
        return xx[a == 0 ? b : (a == 1 ? 2 + b : 5)];
    }
};

struct Gains {
    double Lx[2][3], Lv[2][2], Hi[3];
};

// One stage of the matrix recursion.  wr[c]: weight on the stage's control step (cost + box
// terms), we[c]: weight on its difference to the previous stage's (cost + rate terms; zero at
// stage 0), wq[r]: weight on dx_t (zero at stage 0).  Stores the gains and hands them back,
// updates the matrices of P (not its vector part).
__device__ __forceinline__ void riccati_matrix_step(const Store &s, int t, const Dyn &d, const double (&wr)[2],
                                                    const double (&we)[2], const double (&wq)[3], Riccati &R,
                                                    Gains &gn, bool store)
{
    const int T = s.T;
    // BP = B' Pxx + Pxu'  (2 x 3)
    double BP[2][3];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            BP[c][j] = fma(d.b[0][c], R.Pxx(0, j), fma(d.b[1][c], R.Pxx(1, j), fma(d.b[2][c], R.Pxx(2, j), R.xu[j][c])));
    // Huu = Wr + We + BP B + B' Pxu + Puu
    double h00 = wr[0] + we[0] + R.uu[0], h01 = R.uu[1], h11 = wr[1] + we[1] + R.uu[2];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        h00 = fma(BP[0][j], d.b[j][0], fma(d.b[j][0], R.xu[j][0], h00));
        h01 = fma(BP[0][j], d.b[j][1], fma(d.b[j][0], R.xu[j][1], h01));
        h11 = fma(BP[1][j], d.b[j][1], fma(d.b[j][1], R.xu[j][1], h11));
    }
    const double idet = rcp_nr(fma(h00, h11, -h01 * h01));
    const double i00 = h11 * idet, i01 = -h01 * idet, i11 = h00 * idet;
    // Hux = BP A
    double Hux[2][3];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        Hux[c][0] = BP[c][0];
        Hux[c][1] = BP[c][1];
        Hux[c][2] = fma(d.a02, BP[c][0], fma(d.a12, BP[c][1], BP[c][2]));
    }
    double (&Lx)[2][3] = gn.Lx, (&Lv)[2][2] = gn.Lv;
    gn.Hi[0] = i00; gn.Hi[1] = i01; gn.Hi[2] = i11;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        Lx[0][j] = fma(i00, Hux[0][j], i01 * Hux[1][j]);
        Lx[1][j] = fma(i01, Hux[0][j], i11 * Hux[1][j]);
    }
    Lv[0][0] = -i00 * we[0]; Lv[0][1] = -i01 * we[1];
    Lv[1][0] = -i01 * we[0]; Lv[1][1] = -i11 * we[1];
    if (store) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int j = 0; j < 3; ++j) s.st(s.f(s.fLx + c * 3 + j, t), Lx[c][j]);
            s.st(s.f(s.fLv + c * 2, t), Lv[c][0]);
            s.st(s.f(s.fLv + c * 2 + 1, t), Lv[c][1]);
        }
        s.st(s.f(s.fHi, t), i00); s.st(s.f(s.fHi + 1, t), i01); s.st(s.f(s.fHi + 2, t), i11);
    }
    if (t == 0) return;
    // A' Pxx A
    const double m00 = R.xx[0], m01 = R.xx[1], m02 = R.xx[2], m11 = R.xx[3], m12 = R.xx[4], m22 = R.xx[5];
    const double n02 = fma(d.a02, m00, fma(d.a12, m01, m02));
    const double n12 = fma(d.a02, m01, fma(d.a12, m11, m12));
    const double n22 = fma(d.a02, m02 + n02, fma(d.a12, m12 + n12, m22));
    Riccati N;
    N.xx[0] = wq[0] + m00 - fma(Hux[0][0], Lx[0][0], Hux[1][0] * Lx[1][0]);
    N.xx[1] = m01 - fma(Hux[0][0], Lx[0][1], Hux[1][0] * Lx[1][1]);
    N.xx[2] = n02 - fma(Hux[0][0], Lx[0][2], Hux[1][0] * Lx[1][2]);
    N.xx[3] = wq[1] + m11 - fma(Hux[0][1], Lx[0][1], Hux[1][1] * Lx[1][1]);
    N.xx[4] = n12 - fma(Hux[0][1], Lx[0][2], Hux[1][1] * Lx[1][2]);
    N.xx[5] = wq[2] + n22 - fma(Hux[0][2], Lx[0][2], Hux[1][2] * Lx[1][2]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 2; ++e) N.xu[i][e] = -fma(Hux[0][i], Lv[0][e], Hux[1][i] * Lv[1][e]);
    N.uu[0] = fma(we[0], Lv[0][0], we[0]);
    N.uu[1] = we[0] * Lv[0][1];
    N.uu[2] = fma(we[1], Lv[1][1], we[1]);
#pragma unroll
    for (int e = 0; e < 6; ++e) R.xx[e] = N.xx[e];
#pragma unroll
    for (int i = 0; i < 3; ++i) { R.xu[i][0] = N.xu[i][0]; R.xu[i][1] = N.xu[i][1]; }
    R.uu[0] = N.uu[0]; R.uu[1] = N.uu[1]; R.uu[2] = N.uu[2];
}

// One stage of the vector recursion for right-hand side r_t (K dw = r): hu = -r + B' px + pu,
// feed-forward k = Huu^-1 hu (stored), p <- (A' px - Lx' hu, -Lv' hu).  Gains as arguments.
__device__ __forceinline__ void riccati_vector_step(const Store &s, int t, const Dyn &d, const double (&r)[2],
                                                    const double (&Lx)[2][3], const double (&Lv)[2][2],
                                                    const double (&Hi)[3], Riccati &R, bool store)
{
    double hu[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
        hu[c] = fma(d.b[0][c], R.px[0], fma(d.b[1][c], R.px[1], fma(d.b[2][c], R.px[2], R.pu[c]))) - r[c];
    if (store) {
        s.st(s.p(s.fKv, t, 0), fma(Hi[0], hu[0], Hi[1] * hu[1]));
        s.st(s.p(s.fKv, t, 1), fma(Hi[1], hu[0], Hi[2] * hu[1]));
    }
    const double p2 = fma(d.a02, R.px[0], fma(d.a12, R.px[1], R.px[2]));
    R.px[0] = R.px[0] - fma(Lx[0][0], hu[0], Lx[1][0] * hu[1]);
    R.px[1] = R.px[1] - fma(Lx[0][1], hu[0], Lx[1][1] * hu[1]);
    R.px[2] = p2 - fma(Lx[0][2], hu[0], Lx[1][2] * hu[1]);
    R.pu[0] = -fma(Lv[0][0], hu[0], Lv[1][0] * hu[1]);
    R.pu[1] = -fma(Lv[0][1], hu[0], Lv[1][1] * hu[1]);
}

__device__ __forceinline__ void load_gains(const Store &s, int t, double (&Lx)[2][3], double (&Lv)[2][2])
{
#pragma unroll
    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int j = 0; j < 3; ++j) Lx[c][j] = s.ld(s.f(s.fLx + c * 3 + j, t));
        Lv[c][0] = s.ld(s.f(s.fLv + c * 2, t));
        Lv[c][1] = s.ld(s.f(s.fLv + c * 2 + 1, t));
    }
}

__device__ __forceinline__ void gains_of(const Store &s, int t, Gains &g)
{
    load_gains(s, t, g.Lx, g.Lv);
    g.Hi[0] = s.ld(s.f(s.fHi, t)); g.Hi[1] = s.ld(s.f(s.fHi + 1, t)); g.Hi[2] = s.ld(s.f(s.fHi + 2, t));
}

// Forward sweep: the step dw_t = -(Lx dx_t + Lv dw_{t-1} + k_t), dx_{t+1} = A dx_t + B dw_t.
// `sink(t, dw_t)` receives each stage's step in stage order.
template <class F>
__device__ __forceinline__ void forward(const Store &s, F &&sink)
{
    const int T = s.T;
    double dx[3] = {0.0, 0.0, 0.0}, vp[2] = {0.0, 0.0};
    _Pragma("nounroll") for (int t = 0; t < T; ++t) {
        double Lx[2][3], Lv[2][2];
        load_gains(s, t, Lx, Lv);
        Dyn d;
        load_dyn(s, t, d);
        double u[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
            u[c] = -(fma(Lx[c][0], dx[0], fma(Lx[c][1], dx[1], fma(Lx[c][2], dx[2],
                     fma(Lv[c][0], vp[0], fma(Lv[c][1], vp[1], s.ld(s.p(s.fKv, t, c))))))));
        const double n0 = dx[0] + d.a02 * dx[2] + d.b[0][0] * u[0] + d.b[0][1] * u[1];
        const double n1 = dx[1] + d.a12 * dx[2] + d.b[1][0] * u[0] + d.b[1][1] * u[1];
        const double n2 = dx[2] + d.b[2][0] * u[0] + d.b[2][1] * u[1];
        dx[0] = n0; dx[1] = n1; dx[2] = n2;
        vp[0] = u[0]; vp[1] = u[1];
        sink(t, u);
    }
}

// Adjoint state entering stage t's gradient: lambda_{t+1} = qe_t + A_{t+1}' lambda_{t+2}.
__device__ __forceinline__ void adjoint_step(const Store &s, int t, double a02n, double a12n, double (&lam)[3])
{
    const double l2 = fma(a02n, lam[0], fma(a12n, lam[1], lam[2]));   // a*n = 0 at the last stage
    lam[0] = lam[0] + s.ld(s.f(s.fQe, t));
    lam[1] = lam[1] + s.ld(s.f(s.fQe + 1, t));
    lam[2] = l2 + s.ld(s.f(s.fQe + 2, t));
}

// Gradient at w = 0 (the g of the condensed problem), kept for the scale of the residual tests.
__device__ __forceinline__ double gradient_at_zero(const Store &s, const Problem &p)
{
    double gmax = 0.0;
    const int T = s.T;
    roll_zero(s, p);
    double lam[3] = {0.0, 0.0, 0.0}, a02n = 0.0, a12n = 0.0;
    _Pragma("nounroll") for (int t = T - 1; t >= 0; --t) {
        Dyn d;
        load_dyn(s, t, d);
        adjoint_step(s, t, a02n, a12n, lam);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const double g = fma(d.b[0][c], lam[0], fma(d.b[1][c], lam[1], d.b[2][c] * lam[2]));
            gmax = dmax(gmax, fabs(g));
            s.st(s.p(s.fG, t, c), g);
        }
        a02n = d.a02; a12n = d.a12;
    }
    return gmax;
}

// Diagonal entries (0, 0) and (1, 1) of the condensed H, the largest ones up to the input
// weights: stage 0's controls act on every later state.  Scales the penalty of the rounds.
__device__ __forceinline__ double hessian_scale(const Store &s, const Problem &p)
{
    const int T = s.T;
    double G[2][3], h[2], qg = 1.0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        h[c] = 2.0 * p.rw[c] + (T > 1 ? 2.0 * p.rdw[c] : 0.0);
#pragma unroll
        for (int r = 0; r < 3; ++r) G[c][r] = s.dyn(2 + 2 * r + c, 0);
    }
    for (int ts = 1; ts <= T; ++ts) {
        if (ts < T) qg *= p.growth;
        if (ts > 1) {
            const double a02 = s.dyn(0, ts - 1), a12 = s.dyn(1, ts - 1);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                G[c][0] = fma(a02, G[c][2], G[c][0]);
                G[c][1] = fma(a12, G[c][2], G[c][1]);
            }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 3; ++r) h[c] = fma(2.0 * (ts < T ? p.q[r] * qg : p.qf[r]) * G[c][r], G[c][r], h[c]);
    }
    return dmax(h[0], h[1]);
}

// Stage weights going backward: Q growth^t for 0 < t < T (none at t = 0).
__device__ __forceinline__ void stage_weight(const Problem &p, int t, double qg, double (&wq)[3])
{
#pragma unroll
    for (int r = 0; r < 3; ++r) wq[r] = t > 0 ? 2.0 * p.q[r] * qg : 0.0;
}

// Input-cost part of (H w)_t from the controls of stages t - 1, t, t + 1.
__device__ __forceinline__ double input_cost_grad(const Problem &p, int c, int t, int T, double wp, double wc, double wn)
{
    double h = 2.0 * p.rw[c] * wc;
    if (t > 0) h = fma(2.0 * p.rdw[c], wc - wp, h);
    if (t + 1 < T) h = fma(-2.0 * p.rdw[c], wn - wc, h);
    return h;
}

// ------------------------------------------------------------------------------------
// interior point
// ------------------------------------------------------------------------------------
// Per constraint quantities of one stage t (box of stage t, rate constraint (t - 1, t)) at the
// stored iterate; everything the sweeps need is recomputed from w and the multipliers.
struct Slack {
    double isu[2], isl[2], iru[2], irl[2];   // reciprocal slacks (0 where the constraint is absent)
    double su[2], sl[2], ru[2], rl[2];
};
__device__ __forceinline__ void slacks(const Problem &p, int t, const double (&wc)[2], const double (&wp)[2], Slack &k)
{
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const bool hb = p.hb(t, c), hr = p.hr(c) && t > 0;
        k.su[c] = dmax(p.hi(t, c) - wc[c], kSlackMin);
        k.sl[c] = dmax(wc[c] - p.lo(t, c), kSlackMin);
        const double dv = wc[c] - wp[c];
        k.ru[c] = dmax(p.fb.rr[c] - dv, kSlackMin);
        k.rl[c] = dmax(p.fb.rr[c] + dv, kSlackMin);
        k.isu[c] = hb ? rcp_raw(k.su[c]) : 0.0;
        k.isl[c] = hb ? rcp_raw(k.sl[c]) : 0.0;
        k.iru[c] = hr ? rcp_raw(k.ru[c]) : 0.0;
        k.irl[c] = hr ? rcp_raw(k.rl[c]) : 0.0;
    }
}

// Predictor-corrector interior point on the lanes with `run` set.  `fixed`: exactly max_iter
// iterations, no convergence test.  Otherwise status/iters as in the other kernels.
__device__ __forceinline__ void ipm(const Store &s, const Problem &p, int max_iter, bool fixed, bool run,
                                    int &status, int &iters)
{
    const int T = s.T;
    const double inv_m = [&] {
        double m = 0.0;
        for (int c = 0; c < 2; ++c)
            m += (p.hb(0, c) ? 2.0 : 0.0) + (T > 1 && p.hb(1, c) ? 2.0 * (T - 1) : 0.0) + (p.hr(c) ? 2.0 * (T - 1) : 0.0);
        return m > 0.0 ? 1.0 / m : 0.0;
    }();
    if (run) {
        // strictly feasible start: every stage at the middle of its channel's first-stage window
        double wc[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) wc[c] = p.hb(0, c) ? 0.5 * (p.lo(0, c) + p.hi(0, c)) : 0.0;
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bool hb = p.hb(t, c), hr = p.hr(c) && t + 1 < T;
                s.st(s.p(s.fW, t, c), wc[c]);
                s.st(s.p(s.fLU, t, c), hb ? p.mu0 * rcp_nr(dmax(p.hi(t, c) - wc[c], kSlackMin)) : 0.0);
                s.st(s.p(s.fLL, t, c), hb ? p.mu0 * rcp_nr(dmax(wc[c] - p.lo(t, c), kSlackMin)) : 0.0);
                const double l0 = hr ? p.mu0 * rcp_nr(p.fb.rr[c]) : 0.0;
                s.st(s.p(s.fNU, t, c), l0);
                s.st(s.p(s.fNL, t, c), l0);
            }
    }
    status = MPCX_ST_FAILED;
    iters = 0;
    // step of the previous iteration, applied at the start of the next one in the same sweep
    // that rolls the dynamics forward (and once more on the way out)
    bool pend = false;
    double ap = 0.0, ad = 0.0, sm = 0.0;
    // corrector terms of one stage: q for the box of stage t, p for the rate constraint (t - 1, t)
    auto corr = [&](const Slack &k, int t, int c, double lu, double ll, double nu, double nl, double da, double dap,
                    double &qu, double &ql, double &pu, double &pl) {
        const double xu = da * k.isu[c], xl = -da * k.isl[c];
        const double dva = da - dap;
        const double yu = dva * k.iru[c], yl = -dva * k.irl[c];
        const bool hb = p.hb(t, c), hr = p.hr(c) && t > 0;
        qu = hb ? fma(sm, k.isu[c], lu * xu * (xu - 1.0)) : 0.0;
        ql = hb ? fma(sm, k.isl[c], ll * xl * (xl - 1.0)) : 0.0;
        pu = hr ? fma(sm, k.iru[c], nu * yu * (yu - 1.0)) : 0.0;
        pl = hr ? fma(sm, k.irl[c], nl * yl * (yl - 1.0)) : 0.0;
    };
    auto update_and_roll = [&](bool do_roll) {
        const bool upd = pend && run;
        Roller ro;
        ro.start(p);
        double wpv[2] = {0.0, 0.0}, dpv[2] = {0.0, 0.0}, dapv[2] = {0.0, 0.0};
        _Pragma("nounroll") for (int t = 0; t < T; ++t) {
            double wc[2] = {s.ld(s.p(s.fW, t, 0)), s.ld(s.p(s.fW, t, 1))}, wnew[2];
            if (__any(upd)) {   // old controls of the previous stage are kept for the rate terms
                Slack k;
                slacks(p, t, wc, wpv, k);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double da = s.ld(s.p(s.fDwa, t, c)), dw = s.ld(s.p(s.fDw, t, c));
                    const double lu = s.ld(s.p(s.fLU, t, c)), ll = s.ld(s.p(s.fLL, t, c));
                    const double nu = t > 0 ? s.ld(s.p(s.fNU, t - 1, c)) : 0.0, nl = t > 0 ? s.ld(s.p(s.fNL, t - 1, c)) : 0.0;
                    double qu, ql, pu, pl;
                    corr(k, t, c, lu, ll, nu, nl, da, dapv[c], qu, ql, pu, pl);
                    const double dvv = dw - dpv[c];
                    const double zu = dw * k.isu[c], zl = -dw * k.isl[c], vu = dvv * k.iru[c], vl = -dvv * k.irl[c];
                    wnew[c] = upd ? fma(ap, dw, wc[c]) : wc[c];
                    if (upd) {
                        s.st(s.p(s.fW, t, c), wnew[c]);
                        s.st(s.p(s.fLU, t, c), fma(ad, fma(lu, zu, qu - lu), lu));
                        s.st(s.p(s.fLL, t, c), fma(ad, fma(ll, zl, ql - ll), ll));
                        if (t > 0) {
                            s.st(s.p(s.fNU, t - 1, c), fma(ad, fma(nu, vu, pu - nu), nu));
                            s.st(s.p(s.fNL, t - 1, c), fma(ad, fma(nl, vl, pl - nl), nl));
                        }
                    }
                    wpv[c] = wc[c];
                    dpv[c] = dw;
                    dapv[c] = da;
                }
            } else {
                wnew[0] = wc[0]; wnew[1] = wc[1];
            }
            if (do_roll) ro.step(s, p, t, wnew[0], wnew[1]);
        }
        pend = false;
    };
    for (int it = 0;; ++it) {
        if (fixed && it >= max_iter) {
            if (run) iters = it;
            break;
        }
        if (!__any(run)) break;
        update_and_roll(true);
        // backward: gradient, residuals, Riccati gains, feed-forward of the predictor (r = -hw)
        double musum = 0.0, rdmax = 0.0, scl = 0.0;
        {
            Riccati R;
            R.terminal(p.qf);
            double lam[3] = {0.0, 0.0, 0.0}, a02n = 0.0, a12n = 0.0;
            double wn[2] = {0.0, 0.0}, dnun[2] = {0.0, 0.0};   // controls of stage t + 1, nuU - nuL of constraint (t, t + 1)
            double qg = 1.0;
            for (int t = 1; t < T; ++t) qg *= p.growth;        // growth^(T - 1)
            double wc[2] = {s.ld(s.p(s.fW, T - 1, 0)), s.ld(s.p(s.fW, T - 1, 1))};
            _Pragma("nounroll") for (int t = T - 1; t >= 0; --t) {
                Dyn d;
                load_dyn(s, t, d);
                double wp[2] = {0.0, 0.0}, nup[2] = {0.0, 0.0}, nlp[2] = {0.0, 0.0};
                if (t > 0) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        wp[c] = s.ld(s.p(s.fW, t - 1, c));
                        nup[c] = s.ld(s.p(s.fNU, t - 1, c));
                        nlp[c] = s.ld(s.p(s.fNL, t - 1, c));
                    }
                }
                Slack k;
                slacks(p, t, wc, wp, k);
                adjoint_step(s, t, a02n, a12n, lam);
                double wr[2], we[2], wq[3], r[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double lu = s.ld(s.p(s.fLU, t, c)), ll = s.ld(s.p(s.fLL, t, c));
                    const double hw = input_cost_grad(p, c, t, T, wp[c], wc[c], wn[c])
                                      + fma(d.b[0][c], lam[0], fma(d.b[1][c], lam[1], d.b[2][c] * lam[2]));
                    const double ctl = (lu - ll) + (nup[c] - nlp[c]) - dnun[c];
                    if (p.hb(t, c)) musum = fma(k.su[c], lu, fma(k.sl[c], ll, musum));
                    if (p.hr(c) && t > 0) musum = fma(k.ru[c], nup[c], fma(k.rl[c], nlp[c], musum));
                    rdmax = dmax(rdmax, fabs(hw + ctl));
                    const double g = s.ld(s.p(s.fG, t, c));
                    scl = dmax(scl, dmax(fabs(hw - g), dmax(fabs(g), fabs(ctl))));
                    wr[c] = 2.0 * p.rw[c] + fma(lu, k.isu[c], ll * k.isl[c]);
                    we[c] = t > 0 ? 2.0 * p.rdw[c] + fma(nup[c], k.iru[c], nlp[c] * k.irl[c]) : 0.0;
                    r[c] = -hw;
                    dnun[c] = nup[c] - nlp[c];
                    wn[c] = wc[c];
                    wc[c] = wp[c];
                }
                stage_weight(p, t, qg, wq);
                Gains gn;
                riccati_matrix_step(s, t, d, wr, we, wq, R, gn, run);   // touches the matrices of R only
                riccati_vector_step(s, t, d, r, gn.Lx, gn.Lv, gn.Hi, R, run);
                a02n = d.a02; a12n = d.a12;
                qg = qg / p.growth;
            }
        }
        const double mu = musum * inv_m;
        const double scale = dmax(1.0, scl) + 1.0;
        if (!fixed && run) {
            const bool conv = rdmax <= kTolD * scale && mu <= kTolMu;
            if (conv) { status = MPCX_ST_OK; iters = it; run = false; }
            else if (mu < kMuFloor || it >= max_iter || !finite_d(mu + rdmax)) {
                iters = it;
                status = (rdmax <= kLooseD * scale && mu <= kLooseMu) ? MPCX_ST_INACCURATE : MPCX_ST_FAILED;
                run = false;
            }
        }
        if (!__any(run)) break;
        // forward: predictor step and its step lengths
        double rp = 0.0, rdm = 1.0;
        {
            double wpv[2] = {0.0, 0.0}, dpv[2] = {0.0, 0.0};
            forward(s, [&](int t, const double (&u)[2]) {
                double wc[2] = {s.ld(s.p(s.fW, t, 0)), s.ld(s.p(s.fW, t, 1))};
                Slack k;
                slacks(p, t, wc, wpv, k);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (run) s.st(s.p(s.fDwa, t, c), u[c]);
                    const double xu = u[c] * k.isu[c], xl = -u[c] * k.isl[c];
                    const double dva = u[c] - dpv[c];
                    const double yu = dva * k.iru[c], yl = -dva * k.irl[c];
                    if (p.hb(t, c)) { rp = dmax(rp, dmax(xu, xl)); rdm = dmax(rdm, dmax(1.0 - xu, 1.0 - xl)); }
                    if (p.hr(c) && t > 0) { rp = dmax(rp, dmax(yu, yl)); rdm = dmax(rdm, dmax(1.0 - yu, 1.0 - yl)); }
                    wpv[c] = wc[c];
                    dpv[c] = u[c];
                }
            });
        }
        const double apa = rp > 1.0 ? rcp_nr(rp) : 1.0, ada = rcp_nr(rdm);
        const double back = 1.0 - dmin(apa, ada);
        if (run) sm = back * back * back * mu;
        // backward: feed-forward of the corrector, r_t = -(C' q)_t
        {
            Riccati R;
            R.terminal(p.qf);
            double trn[2] = {0.0, 0.0};    // (pu - pl) of constraint (t, t + 1)
            double wc[2] = {s.ld(s.p(s.fW, T - 1, 0)), s.ld(s.p(s.fW, T - 1, 1))};
            double dc[2] = {s.ld(s.p(s.fDwa, T - 1, 0)), s.ld(s.p(s.fDwa, T - 1, 1))};
            _Pragma("nounroll") for (int t = T - 1; t >= 0; --t) {
                Dyn d;
                load_dyn(s, t, d);
                double wp[2] = {0.0, 0.0}, dp[2] = {0.0, 0.0}, nup[2] = {0.0, 0.0}, nlp[2] = {0.0, 0.0};
                if (t > 0) {
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        wp[c] = s.ld(s.p(s.fW, t - 1, c));
                        dp[c] = s.ld(s.p(s.fDwa, t - 1, c));
                        nup[c] = s.ld(s.p(s.fNU, t - 1, c));
                        nlp[c] = s.ld(s.p(s.fNL, t - 1, c));
                    }
                }
                Slack k;
                slacks(p, t, wc, wp, k);
                double r[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    double qu, ql, pu, pl;
                    corr(k, t, c, s.ld(s.p(s.fLU, t, c)), s.ld(s.p(s.fLL, t, c)), nup[c], nlp[c], dc[c], dp[c], qu, ql, pu, pl);
                    r[c] = -((qu - ql) + (pu - pl) - trn[c]);
                    trn[c] = pu - pl;
                    wc[c] = wp[c];
                    dc[c] = dp[c];
                }
                Gains gn;
                gains_of(s, t, gn);
                riccati_vector_step(s, t, d, r, gn.Lx, gn.Lv, gn.Hi, R, run);
            }
        }
        // forward: full step, its step lengths
        double rp2 = 0.0, rd2 = 0.0, chk = 0.0;
        {
            double wpv[2] = {0.0, 0.0}, dpv[2] = {0.0, 0.0}, dapv[2] = {0.0, 0.0};
            forward(s, [&](int t, const double (&uc)[2]) {
                double wc[2] = {s.ld(s.p(s.fW, t, 0)), s.ld(s.p(s.fW, t, 1))};
                Slack k;
                slacks(p, t, wc, wpv, k);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double da = s.ld(s.p(s.fDwa, t, c));
                    const double dw = da + uc[c];
                    if (run) s.st(s.p(s.fDw, t, c), dw);
                    const double lu = s.ld(s.p(s.fLU, t, c)), ll = s.ld(s.p(s.fLL, t, c));
                    const double nu = t > 0 ? s.ld(s.p(s.fNU, t - 1, c)) : 0.0, nl = t > 0 ? s.ld(s.p(s.fNL, t - 1, c)) : 0.0;
                    double qu, ql, pu, pl;
                    corr(k, t, c, lu, ll, nu, nl, da, dapv[c], qu, ql, pu, pl);
                    const double dvv = dw - dpv[c];
                    const double zu = dw * k.isu[c], zl = -dw * k.isl[c], vu = dvv * k.iru[c], vl = -dvv * k.irl[c];
                    const double dlu = fma(lu, zu, qu - lu), dll = fma(ll, zl, ql - ll);
                    const double dnu = fma(nu, vu, pu - nu), dnl = fma(nl, vl, pl - nl);
                    if (p.hb(t, c)) {
                        rp2 = dmax(rp2, dmax(zu, zl));
                        rd2 = dmax(rd2, dmax(-dlu * rcp_raw(lu), -dll * rcp_raw(ll)));
                        chk += dlu + dll;
                    }
                    if (p.hr(c) && t > 0) {
                        rp2 = dmax(rp2, dmax(vu, vl));
                        rd2 = dmax(rd2, dmax(-dnu * rcp_raw(nu), -dnl * rcp_raw(nl)));
                        chk += dnu + dnl;
                    }
                    chk += dw;
                    wpv[c] = wc[c];
                    dpv[c] = dw;
                    dapv[c] = da;
                }
            });
        }
        // a factorisation that broke down leaves NaN in the step: stop at the iterate we have
        if (run && !finite_d(chk)) {
            iters = it;
            status = (rdmax <= kLooseD * scale && mu <= kLooseMu) ? MPCX_ST_INACCURATE : MPCX_ST_FAILED;
            run = false;
        }
        const double eta = dmin(kEtaCap, dmax(0.99, 1.0 - mu));
        if (run) {
            ap = rp2 > eta ? eta * rcp_nr(rp2) : 1.0;
            ad = rd2 > eta ? eta * rcp_nr(rd2) : 1.0;
            pend = true;
        }
    }
    if (__any(pend && run)) update_and_roll(false);   // whoever continues from here rolls for itself
}

// ------------------------------------------------------------------------------------
// primal-dual active-set rounds
// ------------------------------------------------------------------------------------
// Starts from the interior-point iterate on the lanes with `run`; returns true where the
// point is certified (then w holds it), otherwise w is put back as it was.
__device__ __forceinline__ bool pdas(const Store &s, const Problem &p, int max_rounds, double hmax, bool run)
{
    const int T = s.T;
    // penalty of the augmented Lagrangian: 1e5 times the curvature scale, capped where weights that
    // grow like 2^t would push it to 1e14 and the Newton step of the rounds into rounding noise
    // (T = 30 fixed-steer variant: agreement with the oracle 5e-5 -> 2e-7 in the controls)
    const double rho = dmin(1e5 * hmax, kRhoCap);
    const bool ran = run;
    // initial set: multiplier above slack; the same sweep rolls the dynamics for the first round
    if (__any(run)) {
        Roller ro;
        ro.start(p);
        double wpv[2] = {0.0, 0.0};
        _Pragma("nounroll") for (int t = 0; t < T; ++t) {
            double wst[2];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double wc = s.ld(s.p(s.fW, t, c));
                wst[c] = wc;
                if (!run) continue;
                s.st(s.p(s.fBak, t, c), wc);
                const double lu = s.ld(s.p(s.fLU, t, c)), ll = s.ld(s.p(s.fLL, t, c));
                const bool au = p.hb(t, c) && lu > p.hi(t, c) - wc, al = p.hb(t, c) && ll > wc - p.lo(t, c);
                s.st(s.p(s.fLU, t, c), au ? lu : 0.0);
                s.st(s.p(s.fLL, t, c), al ? ll : 0.0);
                s.st(s.p(s.fSetB, t, c), (double)((au ? 1 : 0) | (al ? 2 : 0)));
                if (t > 0) {
                    const double dv = wc - wpv[c];
                    const double nu = s.ld(s.p(s.fNU, t - 1, c)), nl = s.ld(s.p(s.fNL, t - 1, c));
                    const bool bu = p.hr(c) && nu > p.fb.rr[c] - dv, bl = p.hr(c) && nl > p.fb.rr[c] + dv;
                    s.st(s.p(s.fNU, t - 1, c), bu ? nu : 0.0);
                    s.st(s.p(s.fNL, t - 1, c), bl ? nl : 0.0);
                    s.st(s.p(s.fSetR, t - 1, c), (double)((bu ? 1 : 0) | (bl ? 2 : 0)));
                }
                wpv[c] = wc;
            }
            ro.step(s, p, t, wst[0], wst[1]);
        }
    }
    bool changed = true, accepted = false;
    for (int round = 0;; ++round) {
        if (!__any(run)) break;
        double stat = 0.0, scl = 0.0, viol = 0.0, cact = 0.0, neg = 0.0, fsum = 0.0;
        {
            Riccati R;
            R.terminal(p.qf);
            double lam[3] = {0.0, 0.0, 0.0}, a02n = 0.0, a12n = 0.0;
            double wn[2] = {0.0, 0.0}, dnun[2] = {0.0, 0.0}, trn[2] = {0.0, 0.0};
            double qg = 1.0;
            for (int t = 1; t < T; ++t) qg *= p.growth;
            double wc[2] = {s.ld(s.p(s.fW, T - 1, 0)), s.ld(s.p(s.fW, T - 1, 1))};
            _Pragma("nounroll") for (int t = T - 1; t >= 0; --t) {
                Dyn d;
                load_dyn(s, t, d);
                adjoint_step(s, t, a02n, a12n, lam);
                double wr[2], we[2], wq[3], r[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    double wp = 0.0, nup = 0.0, nlp = 0.0;
                    int setp = 0;
                    if (t > 0) {
                        wp = s.ld(s.p(s.fW, t - 1, c));
                        nup = s.ld(s.p(s.fNU, t - 1, c));
                        nlp = s.ld(s.p(s.fNL, t - 1, c));
                        setp = (int)s.ld(s.p(s.fSetR, t - 1, c));
                    }
                    const int setc = (int)s.ld(s.p(s.fSetB, t, c));
                    const bool au = setc & 1, al = setc & 2, bu = setp & 1, bl = setp & 2;
                    const double lu = s.ld(s.p(s.fLU, t, c)), ll = s.ld(s.p(s.fLL, t, c));
                    const double dv = wc[c] - wp;
                    const double vu = wc[c] - p.hi(t, c), vl = p.lo(t, c) - wc[c];
                    const double ru = dv - p.fb.rr[c], rl = -dv - p.fb.rr[c];
                    const double cu = au ? vu : 0.0, cl = al ? vl : 0.0, du = bu ? ru : 0.0, dl = bl ? rl : 0.0;
                    const double hw = input_cost_grad(p, c, t, T, wp, wc[c], wn[c])
                                      + fma(d.b[0][c], lam[0], fma(d.b[1][c], lam[1], d.b[2][c] * lam[2]));
                    const double gl = hw + (lu - ll) + (nup - nlp) - dnun[c];
                    const double ga = rho * ((cu - cl) + (du - dl) - trn[c]);
                    if (p.hb(t, c)) viol = dmax(viol, dmax(vu, vl));
                    if (p.hr(c) && t > 0) viol = dmax(viol, dmax(ru, rl));
                    cact = dmax(cact, dmax(dmax(fabs(cu), fabs(cl)), dmax(fabs(du), fabs(dl))));
                    neg = dmax(neg, dmax(dmax(-lu, -ll), dmax(-nup, -nlp)));
                    stat = dmax(stat, fabs(gl));
                    const double g = s.ld(s.p(s.fG, t, c));
                    scl = dmax(scl, dmax(fabs(g), fabs(hw - g)));
                    fsum += gl + ga + wc[c];
                    wr[c] = 2.0 * p.rw[c] + rho * ((au ? 1.0 : 0.0) + (al ? 1.0 : 0.0));
                    we[c] = t > 0 ? 2.0 * p.rdw[c] + rho * ((bu ? 1.0 : 0.0) + (bl ? 1.0 : 0.0)) : 0.0;
                    r[c] = -(gl + ga);
                    dnun[c] = nup - nlp;
                    trn[c] = du - dl;
                    wn[c] = wc[c];
                    wc[c] = wp;
                }
                stage_weight(p, t, qg, wq);
                Gains gn;
                riccati_matrix_step(s, t, d, wr, we, wq, R, gn, run);
                riccati_vector_step(s, t, d, r, gn.Lx, gn.Lv, gn.Hi, R, run);
                a02n = d.a02; a12n = d.a12;
                qg = qg / p.growth;
            }
        }
        const double scale = dmax(1.0, scl) + 1.0;
        const bool fin = finite_d(fsum);   // max() drops NaN operands
        if (run && !changed && fin && stat <= 1e-10 * scale && viol <= 1e-12 && cact <= 1e-12 && neg <= 1e-9 * scale) {
            accepted = true;
            run = false;
        }
        if (run && (round >= max_rounds || !fin)) run = false;
        if (!__any(run)) break;
        // forward: Newton step of the augmented Lagrangian, multiplier update, set correction
        // (box of stage t, and the rate constraint (t - 1, t) once both its controls are new)
        const double ntol = -1e-9 * scale;
        bool ch = false;
        {
            double wpn[2] = {0.0, 0.0};   // new controls of the previous stage
            Roller ro;
            ro.start(p);
            forward(s, [&](int t, const double (&u)[2]) {
                double wst[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double wnew = s.ld(s.p(s.fW, t, c)) + u[c];
                    wst[c] = wnew;
                    const int setc = (int)s.ld(s.p(s.fSetB, t, c));
                    const bool au = setc & 1, al = setc & 2;
                    const double vu2 = wnew - p.hi(t, c), vl2 = p.lo(t, c) - wnew;
                    const double lu = au ? fma(rho, vu2, s.ld(s.p(s.fLU, t, c))) : 0.0;
                    const double ll = al ? fma(rho, vl2, s.ld(s.p(s.fLL, t, c))) : 0.0;
                    const bool au2 = p.hb(t, c) && ((au && lu >= ntol) || vu2 > 1e-12);
                    const bool al2 = p.hb(t, c) && ((al && ll >= ntol) || vl2 > 1e-12);
                    ch = ch || au2 != au || al2 != al;
                    if (run) {
                        s.st(s.p(s.fW, t, c), wnew);
                        s.st(s.p(s.fLU, t, c), au2 ? dmax(lu, 0.0) : 0.0);
                        s.st(s.p(s.fLL, t, c), al2 ? dmax(ll, 0.0) : 0.0);
                        s.st(s.p(s.fSetB, t, c), (double)((au2 ? 1 : 0) | (al2 ? 2 : 0)));
                    }
                    if (t > 0) {
                        const int setp = (int)s.ld(s.p(s.fSetR, t - 1, c));
                        const bool bu = setp & 1, bl = setp & 2;
                        const double dv2 = wnew - wpn[c];
                        const double ru2 = dv2 - p.fb.rr[c], rl2 = -dv2 - p.fb.rr[c];
                        const double nu = bu ? fma(rho, ru2, s.ld(s.p(s.fNU, t - 1, c))) : 0.0;
                        const double nl = bl ? fma(rho, rl2, s.ld(s.p(s.fNL, t - 1, c))) : 0.0;
                        const bool bu2 = p.hr(c) && ((bu && nu >= ntol) || ru2 > 1e-12);
                        const bool bl2 = p.hr(c) && ((bl && nl >= ntol) || rl2 > 1e-12);
                        ch = ch || bu2 != bu || bl2 != bl;
                        if (run) {
                            s.st(s.p(s.fNU, t - 1, c), bu2 ? dmax(nu, 0.0) : 0.0);
                            s.st(s.p(s.fNL, t - 1, c), bl2 ? dmax(nl, 0.0) : 0.0);
                            s.st(s.p(s.fSetR, t - 1, c), (double)((bu2 ? 1 : 0) | (bl2 ? 2 : 0)));
                        }
                    }
                    wpn[c] = wnew;
                }
                ro.step(s, p, t, wst[0], wst[1]);   // for the next round's adjoint sweep
            });
        }
        changed = ch;
    }
    if (ran && !accepted)
        for (int t = 0; t < T; ++t) {
            s.st(s.p(s.fW, t, 0), s.ld(s.p(s.fBak, t, 0)));
            s.st(s.p(s.fW, t, 1), s.ld(s.p(s.fBak, t, 1)));
        }
    return accepted;
}

}  // namespace stagewise
}  // namespace mpcx
