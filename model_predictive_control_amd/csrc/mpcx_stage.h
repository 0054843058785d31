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
// chunk in address order.  The solver: predictor-corrector interior-point iterations, started in
// proportion to the gradient, until the complementarity is small enough to read the active set
// off the iterate; then primal-dual active-set rounds in which the working set is eliminated
// exactly (a control on its box is fixed, a control on a rate bound moves with the previous
// stage's: both are affine policies the Riccati sweep carries), multipliers are recovered from
// the true gradient by sums along the chains of tied stages, and a point is accepted on the KKT
// conditions stage by stage, each residual measured against the rounding noise of the terms it
// sums.  Whatever does not settle continues the interior point from where it stood and tries again.
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
        fLx = 29, fLv = 35,   // Riccati gains
        fDq = 39,        // 2 Q_{t+1} (change of x_{t+1} along the step): the rollout of the next iterate without rolling
        fKv = 42,        // feed-forward of the current right-hand side
        fDwa = 44, fDw = 46,
        fBak = 48,       // controls before the active-set rounds
        fUref = 50,      // linearisation point: controls and yaw (rewritten between chained solves)
        fTheta = 54,
        kFields = 55,
        // the rounds keep the gradient, the chain sum it hands to its left neighbour and the
        // noise scale where the interior point keeps its steps
        fGr = fDwa, fMag = fDw, fCur = fKv
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
    // where the interior point starts: every stage at the middle of its channel's first-stage window
    __device__ __forceinline__ double start(int c) const { return hb(0, c) ? 0.5 * (lo(0, c) + hi(0, c)) : 0.0; }
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

// Every sweep over the stages is latency bound unless it is written for the memory system: a
// load that follows a store in program order stays behind it (the compiler cannot tell that
// the two do not overlap), so a stage body that interleaves them pays one memory round trip
// per interleaving.  All sweeps therefore have the same shape: `load(t, rec)` issues every load
// of stage t back to back into a record, `body(t, rec)` computes and stores; with kPrefetch the
// record of the next stage is requested before the current stage is worked on, so that its
// round trip overlaps the arithmetic and the stores (no stage reads what its predecessor in
// the same sweep writes).  Stages first, first + step, ..., last.
#ifndef MPCX_STAGE_PREFETCH
#define MPCX_STAGE_PREFETCH 1
#endif
constexpr bool kPrefetch = MPCX_STAGE_PREFETCH != 0;

template <class Rec, class LD, class BODY>
__device__ __forceinline__ void sweep(int first, int last, int step, LD &&load, BODY &&body)
{
    if (kPrefetch) {
        Rec cur;
        load(first, cur);
        _Pragma("nounroll") for (int t = first;; t += step) {
            Rec nxt = cur;
            const bool more = t != last;
            if (more) load(t + step, nxt);
            body(t, cur);
            if (!more) break;
            cur = nxt;
        }
    } else {
        _Pragma("nounroll") for (int t = first;; t += step) {
            Rec cur;
            load(t, cur);
            body(t, cur);
            if (t == last) break;
        }
    }
}

// Forward rollout of the linearised dynamics, one stage per call; leaves
// 2 Q_{t+1} (x_{t+1} - xref_{t+1}) for the adjoint sweep.
struct RollIn {
    double c[3], xr[3];   // affine term of the stage, reference of stage t + 1
};
__device__ __forceinline__ void load_roll(const Store &s, int t, RollIn &in)
{
#pragma unroll
    for (int r = 0; r < 3; ++r) { in.c[r] = s.dyn(8 + r, t); in.xr[r] = s.ld(s.f(s.fXr + r, t)); }
}
struct Roller {
    double x[3], qg;
    __device__ __forceinline__ void start(const Problem &p) { x[0] = p.x0[0]; x[1] = p.x0[1]; x[2] = p.x0[2]; qg = 1.0; }
    __device__ __forceinline__ void step(const Store &s, const Problem &p, int t, double w0, double w1, const Dyn &d,
                                         const RollIn &in, int dst = Store::fQe)
    {
        const int T = s.T;
        const double n0 = x[0] + d.a02 * x[2] + d.b[0][0] * w0 + d.b[0][1] * w1 + in.c[0];
        const double n1 = x[1] + d.a12 * x[2] + d.b[1][0] * w0 + d.b[1][1] * w1 + in.c[1];
        const double n2 = x[2] + d.b[2][0] * w0 + d.b[2][1] * w1 + in.c[2];
        x[0] = n0; x[1] = n1; x[2] = n2;
        const int ts = t + 1;
        if (ts < T) qg *= p.growth;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double q = ts < T ? p.q[r] * qg : p.qf[r];
            s.st(s.f(dst + r, t), 2.0 * q * (x[r] - in.xr[r]));
        }
    }
};
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

// Forward sweep: the step dw_t = -(Lx dx_t + Lv dw_{t-1} + k_t), dx_{t+1} = A dx_t + B dw_t.
// `ldx(t, x)` loads whatever else the caller needs of stage t (type X) along with the gains;
// `sink(t, dw_t, dx_{t+1}, d, x)` receives each stage's step and the state step it leads to, in
// stage order.
template <class X, class LDX, class F>
__device__ __forceinline__ void forward(const Store &s, LDX &&ldx, F &&sink)
{
    struct Rec { double Lx[2][3], Lv[2][2], kv[2]; Dyn d; X x; };
    double dx[3] = {0.0, 0.0, 0.0}, vp[2] = {0.0, 0.0};
    sweep<Rec>(0, s.T - 1, 1,
               [&](int t, Rec &in) {
                   load_gains(s, t, in.Lx, in.Lv);
                   load_dyn(s, t, in.d);
                   in.kv[0] = s.ld(s.p(s.fKv, t, 0)); in.kv[1] = s.ld(s.p(s.fKv, t, 1));
                   ldx(t, in.x);
               },
               [&](int t, const Rec &in) {
                   const Dyn &d = in.d;
                   double u[2];
#pragma unroll
                   for (int c = 0; c < 2; ++c)
                       u[c] = -(fma(in.Lx[c][0], dx[0], fma(in.Lx[c][1], dx[1], fma(in.Lx[c][2], dx[2],
                                fma(in.Lv[c][0], vp[0], fma(in.Lv[c][1], vp[1], in.kv[c]))))));
                   const double n0 = dx[0] + d.a02 * dx[2] + d.b[0][0] * u[0] + d.b[0][1] * u[1];
                   const double n1 = dx[1] + d.a12 * dx[2] + d.b[1][0] * u[0] + d.b[1][1] * u[1];
                   const double n2 = dx[2] + d.b[2][0] * u[0] + d.b[2][1] * u[1];
                   dx[0] = n0; dx[1] = n1; dx[2] = n2;
                   vp[0] = u[0]; vp[1] = u[1];
                   sink(t, u, dx, d, in.x);
               });
}

// Adjoint state entering stage t's gradient: lambda_{t+1} = qe_t + A_{t+1}' lambda_{t+2}.
__device__ __forceinline__ void adjoint_step(const double (&qe)[3], double a02n, double a12n, double (&lam)[3])
{
    const double l2 = fma(a02n, lam[0], fma(a12n, lam[1], lam[2]));   // a*n = 0 at the last stage
    lam[0] = lam[0] + qe[0];
    lam[1] = lam[1] + qe[1];
    lam[2] = l2 + qe[2];
}
__device__ __forceinline__ void load_qe(const Store &s, int t, double (&qe)[3])
{
#pragma unroll
    for (int r = 0; r < 3; ++r) qe[r] = s.ld(s.f(s.fQe + r, t));
}

// Gradient at w = 0 (the g of the condensed problem), kept for the scale of the residual tests:
// the adjoint sweep over the rollout at w = 0 that the caller left in field `qe0`.
__device__ __forceinline__ double gradient_at_zero(const Store &s, int qe0)
{
    struct Rec { Dyn d; double qe[3]; };
    double gmax = 0.0;
    double lam[3] = {0.0, 0.0, 0.0}, a02n = 0.0, a12n = 0.0;
    sweep<Rec>(s.T - 1, 0, -1,
               [&](int t, Rec &in) {
                   load_dyn(s, t, in.d);
#pragma unroll
                   for (int r = 0; r < 3; ++r) in.qe[r] = s.ld(s.f(qe0 + r, t));
               },
               [&](int t, const Rec &in) {
                   const Dyn &d = in.d;
                   adjoint_step(in.qe, a02n, a12n, lam);
#pragma unroll
                   for (int c = 0; c < 2; ++c) {
                       const double g = fma(d.b[0][c], lam[0], fma(d.b[1][c], lam[1], d.b[2][c] * lam[2]));
                       gmax = dmax(gmax, fabs(g));
                       s.st(s.p(s.fG, t, c), g);
                   }
                   a02n = d.a02; a12n = d.a12;
               });
    return gmax;
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

// Predictor-corrector interior point on the lanes with `run` set.  With mu_stop > 0 a lane
// stops as soon as its complementarity is below mu_stop and its dual residual is small (the
// active set can then be read off the iterate); otherwise it runs to convergence; at most
// max_iter iterations either way.  `resume` continues from the stored iterate (controls and
// multipliers as the previous call left them) instead of starting over.  `rolled`: the rollout of
// the starting controls (Problem::start) is already in place.
__device__ __forceinline__ void ipm(const Store &s, const Problem &p, int max_iter, double mu_stop, bool resume, bool run,
                                    int &status, int &iters, bool rolled = false)
{
    const int T = s.T;
    const double inv_m = [&] {
        double m = 0.0;
        for (int c = 0; c < 2; ++c)
            m += (p.hb(0, c) ? 2.0 : 0.0) + (T > 1 && p.hb(1, c) ? 2.0 * (T - 1) : 0.0) + (p.hr(c) ? 2.0 * (T - 1) : 0.0);
        return m > 0.0 ? 1.0 / m : 0.0;
    }();
    if (run && !resume) {
        // strictly feasible start: every stage at the middle of its channel's first-stage window
        const double wc[2] = {p.start(0), p.start(1)};
        // The complementarity a constraint starts at follows the gradient of its own stage (capped
        // by the instance's): with stage weights that grow like 2^t the early stages' multipliers
        // end five orders of magnitude below the late ones', and starting them all at the scale
        // of the largest costs iterations that only bring them down.
        double gn[2] = {fabs(s.ld(s.p(s.fG, 0, 0))), fabs(s.ld(s.p(s.fG, 0, 1)))};
        for (int t = 0; t < T; ++t) {
            const double gc[2] = {gn[0], gn[1]};
            if (t + 1 < T) { gn[0] = fabs(s.ld(s.p(s.fG, t + 1, 0))); gn[1] = fabs(s.ld(s.p(s.fG, t + 1, 1))); }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const bool hb = p.hb(t, c), hr = p.hr(c) && t + 1 < T;
                const double mub = dmin(p.mu0, kMu0 * dmax(1.0, gc[c]));
                const double mur = dmin(p.mu0, kMu0 * dmax(1.0, dmax(gc[c], gn[c])));
                s.st(s.p(s.fW, t, c), wc[c]);
                s.st(s.p(s.fLU, t, c), hb ? mub * rcp_nr(dmax(p.hi(t, c) - wc[c], kSlackMin)) : 0.0);
                s.st(s.p(s.fLL, t, c), hb ? mub * rcp_nr(dmax(wc[c] - p.lo(t, c), kSlackMin)) : 0.0);
                const double l0 = hr ? mur * rcp_nr(p.fb.rr[c]) : 0.0;
                s.st(s.p(s.fNU, t, c), l0);
                s.st(s.p(s.fNL, t, c), l0);
            }
        }
    }
    status = MPCX_ST_FAILED;
    iters = 0;
    // Step of the previous iteration (primal and dual step lengths, centring term): applied at
    // the start of the next iteration, inside the backward sweep that factorises, because only
    // then are the step lengths known.  The rollout is not repeated for the new iterate: the
    // forward sweep that found the step left 2 Q dx along it, and x is linear in w.
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
    // barrier weights of one stage from the iterate
    auto weights = [&](const Slack &k, int t, const double (&lu)[2], const double (&ll)[2], const double (&nup)[2],
                       const double (&nlp)[2], double (&wr)[2], double (&we)[2]) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            wr[c] = 2.0 * p.rw[c] + fma(lu[c], k.isu[c], ll[c] * k.isl[c]);
            we[c] = t > 0 ? 2.0 * p.rdw[c] + fma(nup[c], k.iru[c], nlp[c] * k.irl[c]) : 0.0;
        }
    };
    // rollout at the iterate we start from (the rounds leave their own trial point's behind);
    // `rolled`: the caller has already rolled the starting controls
    if (__any(run) && !(rolled && !resume)) {
        struct Rec { double w[2]; Dyn d; RollIn r; };
        Roller ro;
        ro.start(p);
        sweep<Rec>(0, T - 1, 1,
                   [&](int t, Rec &in) {
                       in.w[0] = s.ld(s.p(s.fW, t, 0)); in.w[1] = s.ld(s.p(s.fW, t, 1));
                       load_dyn(s, t, in.d);
                       load_roll(s, t, in.r);
                   },
                   [&](int t, const Rec &in) { ro.step(s, p, t, in.w[0], in.w[1], in.d, in.r); });
    }
    for (int it = 0;; ++it) {
        if (!__any(run)) break;
        // backward: the pending step, then gradient, residuals, Riccati gains and the feed-forward
        // of the predictor (r = -hw) at the new iterate
        double musum = 0.0, rdmax = 0.0, scl = 0.0;
        {
            struct Rec { Dyn d; double wp[2], dwp[2], dap[2], nup[2], nlp[2], lu[2], ll[2], qe[3], dq[3], g[2]; };
            const bool upd = pend && run;
            const bool any_upd = __any(upd);
            Riccati R;
            R.terminal(p.qf);
            double lam[3] = {0.0, 0.0, 0.0}, a02n = 0.0, a12n = 0.0;
            double wn[2] = {0.0, 0.0}, dnun[2] = {0.0, 0.0};   // controls of stage t + 1, nuU - nuL of constraint (t, t + 1)
            double qg = 1.0;
            for (int t = 1; t < T; ++t) qg *= p.growth;        // growth^(T - 1)
            // stage t's controls before and after the pending step, and its steps
            double wo[2] = {s.ld(s.p(s.fW, T - 1, 0)), s.ld(s.p(s.fW, T - 1, 1))}, wc[2], dwc[2] = {0.0, 0.0}, dac[2] = {0.0, 0.0};
            if (any_upd) {
#pragma unroll
                for (int c = 0; c < 2; ++c) { dwc[c] = s.ld(s.p(s.fDw, T - 1, c)); dac[c] = s.ld(s.p(s.fDwa, T - 1, c)); }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) wc[c] = upd ? fma(ap, dwc[c], wo[c]) : wo[c];
            sweep<Rec>(T - 1, 0, -1,
                       [&](int t, Rec &in) {
                           load_dyn(s, t, in.d);
#pragma unroll
                           for (int c = 0; c < 2; ++c) {
                               in.wp[c] = t > 0 ? s.ld(s.p(s.fW, t - 1, c)) : 0.0;
                               in.nup[c] = t > 0 ? s.ld(s.p(s.fNU, t - 1, c)) : 0.0;
                               in.nlp[c] = t > 0 ? s.ld(s.p(s.fNL, t - 1, c)) : 0.0;
                               in.lu[c] = s.ld(s.p(s.fLU, t, c)); in.ll[c] = s.ld(s.p(s.fLL, t, c));
                               in.g[c] = s.ld(s.p(s.fG, t, c));
                               in.dwp[c] = any_upd && t > 0 ? s.ld(s.p(s.fDw, t - 1, c)) : 0.0;
                               in.dap[c] = any_upd && t > 0 ? s.ld(s.p(s.fDwa, t - 1, c)) : 0.0;
                           }
                           load_qe(s, t, in.qe);
                           if (any_upd) {
#pragma unroll
                               for (int r = 0; r < 3; ++r) in.dq[r] = s.ld(s.f(s.fDq + r, t));
                           }
                       },
                       [&](int t, const Rec &in) {
                           const Dyn &d = in.d;
                           double wp[2] = {in.wp[0], in.wp[1]}, nup[2] = {in.nup[0], in.nup[1]}, nlp[2] = {in.nlp[0], in.nlp[1]};
                           double lu[2] = {in.lu[0], in.lu[1]}, ll[2] = {in.ll[0], in.ll[1]};
                           double qe[3] = {in.qe[0], in.qe[1], in.qe[2]};
                           if (any_upd) {
                               Slack ko;   // at the iterate the step was computed for
                               slacks(p, t, wo, in.wp, ko);
#pragma unroll
                               for (int c = 0; c < 2; ++c) {
                                   double qu, ql, pu, pl;
                                   corr(ko, t, c, in.lu[c], in.ll[c], in.nup[c], in.nlp[c], dac[c], in.dap[c], qu, ql, pu, pl);
                                   const double dw = dwc[c], dvv = dw - in.dwp[c];
                                   const double zu = dw * ko.isu[c], zl = -dw * ko.isl[c], vu = dvv * ko.iru[c], vl = -dvv * ko.irl[c];
                                   if (upd) {
                                       lu[c] = fma(ad, fma(in.lu[c], zu, qu - in.lu[c]), in.lu[c]);
                                       ll[c] = fma(ad, fma(in.ll[c], zl, ql - in.ll[c]), in.ll[c]);
                                       s.st(s.p(s.fW, t, c), wc[c]);
                                       s.st(s.p(s.fLU, t, c), lu[c]);
                                       s.st(s.p(s.fLL, t, c), ll[c]);
                                       if (t > 0) {
                                           wp[c] = fma(ap, in.dwp[c], in.wp[c]);
                                           nup[c] = fma(ad, fma(in.nup[c], vu, pu - in.nup[c]), in.nup[c]);
                                           nlp[c] = fma(ad, fma(in.nlp[c], vl, pl - in.nlp[c]), in.nlp[c]);
                                           s.st(s.p(s.fNU, t - 1, c), nup[c]);
                                           s.st(s.p(s.fNL, t - 1, c), nlp[c]);
                                       }
                                   }
                               }
                               if (upd) {
#pragma unroll
                                   for (int r = 0; r < 3; ++r) {
                                       qe[r] = fma(ap, in.dq[r], qe[r]);
                                       s.st(s.f(s.fQe + r, t), qe[r]);
                                   }
                               }
                           }
                           Slack k;
                           slacks(p, t, wc, wp, k);
                           adjoint_step(qe, a02n, a12n, lam);
                           double wr[2], we[2], wq[3], r[2];
                           weights(k, t, lu, ll, nup, nlp, wr, we);
#pragma unroll
                           for (int c = 0; c < 2; ++c) {
                               const double hw = input_cost_grad(p, c, t, T, wp[c], wc[c], wn[c])
                                                 + fma(d.b[0][c], lam[0], fma(d.b[1][c], lam[1], d.b[2][c] * lam[2]));
                               const double ctl = (lu[c] - ll[c]) + (nup[c] - nlp[c]) - dnun[c];
                               if (p.hb(t, c)) musum = fma(k.su[c], lu[c], fma(k.sl[c], ll[c], musum));
                               if (p.hr(c) && t > 0) musum = fma(k.ru[c], nup[c], fma(k.rl[c], nlp[c], musum));
                               rdmax = dmax(rdmax, fabs(hw + ctl));
                               const double g = in.g[c];
                               scl = dmax(scl, dmax(fabs(hw - g), dmax(fabs(g), fabs(ctl))));
                               r[c] = -hw;
                               dnun[c] = nup[c] - nlp[c];
                               wn[c] = wc[c];
                               wc[c] = wp[c];
                               wo[c] = in.wp[c];
                               dwc[c] = in.dwp[c];
                               dac[c] = in.dap[c];
                           }
                           stage_weight(p, t, qg, wq);
                           Gains gn;
                           riccati_matrix_step(s, t, d, wr, we, wq, R, gn, run);   // touches the matrices of R only
                           riccati_vector_step(s, t, d, r, gn.Lx, gn.Lv, gn.Hi, R, run);
                           a02n = d.a02; a12n = d.a12;
                           qg = qg / p.growth;
                       });
            pend = false;
        }
        const double mu = musum * inv_m;
        const double scale = dmax(1.0, scl) + 1.0;
        if (run) {
            const bool conv = rdmax <= kTolD * scale && mu <= kTolMu;
            if (conv) { status = MPCX_ST_OK; iters = it; run = false; }
            else if (mu_stop > 0.0 && mu <= mu_stop && rdmax <= kSwitchD * scale) {
                iters = it;
                status = (rdmax <= kLooseD * scale && mu <= kLooseMu) ? MPCX_ST_INACCURATE : MPCX_ST_FAILED;
                run = false;
            }
            else if ((inv_m > 0.0 && mu < kMuFloor) || it >= max_iter || !finite_d(mu + rdmax)) {
                iters = it;
                status = (rdmax <= kLooseD * scale && mu <= kLooseMu) ? MPCX_ST_INACCURATE : MPCX_ST_FAILED;
                run = false;
            }
        }
        if (!__any(run)) break;
        // forward: predictor step and its step lengths
        double rp = 0.0, rdm = 1.0;
        {
            struct X { double w[2]; };
            double wpv[2] = {0.0, 0.0}, dpv[2] = {0.0, 0.0};
            forward<X>(s, [&](int t, X &x) { x.w[0] = s.ld(s.p(s.fW, t, 0)); x.w[1] = s.ld(s.p(s.fW, t, 1)); },
                       [&](int t, const double (&u)[2], const double (&)[3], const Dyn &, const X &x) {
                Slack k;
                slacks(p, t, x.w, wpv, k);
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    if (run) s.st(s.p(s.fDwa, t, c), u[c]);
                    const double xu = u[c] * k.isu[c], xl = -u[c] * k.isl[c];
                    const double dva = u[c] - dpv[c];
                    const double yu = dva * k.iru[c], yl = -dva * k.irl[c];
                    if (p.hb(t, c)) { rp = dmax(rp, dmax(xu, xl)); rdm = dmax(rdm, dmax(1.0 - xu, 1.0 - xl)); }
                    if (p.hr(c) && t > 0) { rp = dmax(rp, dmax(yu, yl)); rdm = dmax(rdm, dmax(1.0 - yu, 1.0 - yl)); }
                    wpv[c] = x.w[c];
                    dpv[c] = u[c];
                }
            });
        }
        const double apa = rp > 1.0 ? rcp_nr(rp) : 1.0, ada = rcp_nr(rdm);
        const double back = 1.0 - dmin(apa, ada);
        if (run) sm = back * back * back * mu;
        // backward: feed-forward of the corrector, r_t = -(C' q)_t; the gains are formed again from
        // the iterate (arithmetic the memory system does not see) rather than read back
        {
            struct Rec { Dyn d; double wp[2], dp[2], nup[2], nlp[2], lu[2], ll[2]; };
            Riccati R;
            R.terminal(p.qf);
            double trn[2] = {0.0, 0.0};    // (pu - pl) of constraint (t, t + 1)
            double qg = 1.0;
            for (int t = 1; t < T; ++t) qg *= p.growth;
            double wc[2] = {s.ld(s.p(s.fW, T - 1, 0)), s.ld(s.p(s.fW, T - 1, 1))};
            double dc[2] = {s.ld(s.p(s.fDwa, T - 1, 0)), s.ld(s.p(s.fDwa, T - 1, 1))};
            sweep<Rec>(T - 1, 0, -1,
                       [&](int t, Rec &in) {
                           load_dyn(s, t, in.d);
#pragma unroll
                           for (int c = 0; c < 2; ++c) {
                               in.wp[c] = t > 0 ? s.ld(s.p(s.fW, t - 1, c)) : 0.0;
                               in.dp[c] = t > 0 ? s.ld(s.p(s.fDwa, t - 1, c)) : 0.0;
                               in.nup[c] = t > 0 ? s.ld(s.p(s.fNU, t - 1, c)) : 0.0;
                               in.nlp[c] = t > 0 ? s.ld(s.p(s.fNL, t - 1, c)) : 0.0;
                               in.lu[c] = s.ld(s.p(s.fLU, t, c)); in.ll[c] = s.ld(s.p(s.fLL, t, c));
                           }
                       },
                       [&](int t, const Rec &in) {
                           Slack k;
                           slacks(p, t, wc, in.wp, k);
                           double r[2], wr[2], we[2], wq[3];
                           weights(k, t, in.lu, in.ll, in.nup, in.nlp, wr, we);
#pragma unroll
                           for (int c = 0; c < 2; ++c) {
                               double qu, ql, pu, pl;
                               corr(k, t, c, in.lu[c], in.ll[c], in.nup[c], in.nlp[c], dc[c], in.dp[c], qu, ql, pu, pl);
                               r[c] = -((qu - ql) + (pu - pl) - trn[c]);
                               trn[c] = pu - pl;
                               wc[c] = in.wp[c];
                               dc[c] = in.dp[c];
                           }
                           stage_weight(p, t, qg, wq);
                           Gains gn;
                           riccati_matrix_step(s, t, in.d, wr, we, wq, R, gn, false);
                           riccati_vector_step(s, t, in.d, r, gn.Lx, gn.Lv, gn.Hi, R, run);
                           qg = qg / p.growth;
                       });
        }
        // forward: full step, its step lengths, the change of the rollout along it
        double rp2 = 0.0, rd2 = 0.0, chk = 0.0;
        {
            struct X { double w[2], da[2], lu[2], ll[2], nu[2], nl[2]; };
            double wpv[2] = {0.0, 0.0}, dpv[2] = {0.0, 0.0}, dapv[2] = {0.0, 0.0};
            // this sweep's own state step belongs to the corrector's difference to the predictor;
            // the state step of the whole step is rolled beside it
            double dxf[3] = {0.0, 0.0, 0.0}, qgf = 1.0;
            forward<X>(s,
                       [&](int t, X &x) {
#pragma unroll
                           for (int c = 0; c < 2; ++c) {
                               x.w[c] = s.ld(s.p(s.fW, t, c)); x.da[c] = s.ld(s.p(s.fDwa, t, c));
                               x.lu[c] = s.ld(s.p(s.fLU, t, c)); x.ll[c] = s.ld(s.p(s.fLL, t, c));
                               x.nu[c] = t > 0 ? s.ld(s.p(s.fNU, t - 1, c)) : 0.0;
                               x.nl[c] = t > 0 ? s.ld(s.p(s.fNL, t - 1, c)) : 0.0;
                           }
                       },
                       [&](int t, const double (&uc)[2], const double (&)[3], const Dyn &d, const X &x) {
                Slack k;
                slacks(p, t, x.w, wpv, k);
                double dwv[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double da = x.da[c];
                    const double dw = da + uc[c];
                    dwv[c] = dw;
                    if (run) s.st(s.p(s.fDw, t, c), dw);
                    const double lu = x.lu[c], ll = x.ll[c], nu = x.nu[c], nl = x.nl[c];
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
                    wpv[c] = x.w[c];
                    dpv[c] = dw;
                    dapv[c] = da;
                }
                const double n0 = dxf[0] + d.a02 * dxf[2] + d.b[0][0] * dwv[0] + d.b[0][1] * dwv[1];
                const double n1 = dxf[1] + d.a12 * dxf[2] + d.b[1][0] * dwv[0] + d.b[1][1] * dwv[1];
                const double n2 = dxf[2] + d.b[2][0] * dwv[0] + d.b[2][1] * dwv[1];
                dxf[0] = n0; dxf[1] = n1; dxf[2] = n2;
                const int ts = t + 1;
                if (ts < T) qgf *= p.growth;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double q = ts < T ? p.q[r] * qgf : p.qf[r];
                    if (run) s.st(s.f(s.fDq + r, t), 2.0 * q * dxf[r]);
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
}

// ------------------------------------------------------------------------------------
// primal-dual active-set rounds with the working set eliminated exactly
// ------------------------------------------------------------------------------------
// The working set, one bit per stage (T <= 64): bu/bl the upper/lower box of stage t, ru/rl the
// rate constraint between stages t and t + 1 at +rr / -rr.
struct Sets {
    uint64_t bu[2], bl[2], ru[2], rl[2];
    __device__ __forceinline__ void clear()
    {
#pragma unroll
        for (int c = 0; c < 2; ++c) bu[c] = bl[c] = ru[c] = rl[c] = 0;
    }
};
__device__ __forceinline__ bool bit(uint64_t m, int t) { return t < 64 && ((m >> t) & 1ull) != 0; }   // stage T of a T = 64 horizon: no bit
__device__ __forceinline__ uint64_t one(int t) { return 1ull << t; }

// Stages whose control is determined by the working set alone: on a box, or tied to such a
// stage further on (fixedR_t = box_t or (tie_t and fixedR_{t+1})).
__device__ __forceinline__ uint64_t fixed_from_right(uint64_t box, uint64_t tie, int T)
{
    uint64_t f = 0;
    bool nxt = false;
    for (int t = T - 1; t >= 0; --t) {
        const bool cur = bit(box, t) || (bit(tie, t) && nxt);
        if (cur) f |= one(t);
        nxt = cur;
    }
    return f;
}

#ifndef MPCX_NOISE_TOL
#define MPCX_NOISE_TOL 5e-14
#endif
constexpr double kNoiseTol = MPCX_NOISE_TOL;   // KKT residuals are accepted below this multiple of (1 + the summed magnitudes)
constexpr double kFeasTol = 1e-12;    // constraint violation that enters the working set

// Backward sweep at the stored controls: the true gradient gr_t = (H w + g)_t by the adjoint of
// the rollout, the size mag_t of the terms it sums (x - xref cancels, so its rounding error is
// relative to |x| + |xref| <= |x - xref| + 2 |xref|), and for every rate constraint whose later
// stage moves with the earlier one the multiplier nu_t = nu_{t+1} - gr_{t+1} (stationarity of
// that later stage); a multiplier of the wrong sign is marked in `drop`.  Stores gr, the chain
// term cur_t = nu_t of such a constraint (else 0) and mag per stage.
__device__ __forceinline__ void rounds_adjoint(const Store &s, const Problem &p, const Sets &S, const uint64_t (&fixedR)[2],
                                               Sets &drop)
{
    struct Rec { Dyn d; double qe[3], xr[3], wp[2]; };
    const int T = s.T;
    double lam[3] = {0.0, 0.0, 0.0}, la[3] = {0.0, 0.0, 0.0}, a02n = 0.0, a12n = 0.0;
    double wn[2] = {0.0, 0.0}, curn[2] = {0.0, 0.0}, grn[2] = {0.0, 0.0}, magn[2] = {0.0, 0.0};
    double qg = 1.0;
    for (int t = 1; t < T; ++t) qg *= p.growth;        // growth^(T - 1)
    double wc[2] = {s.ld(s.p(s.fW, T - 1, 0)), s.ld(s.p(s.fW, T - 1, 1))};
    sweep<Rec>(T - 1, 0, -1,
               [&](int t, Rec &in) {
                   load_dyn(s, t, in.d);
                   load_qe(s, t, in.qe);
#pragma unroll
                   for (int r = 0; r < 3; ++r) in.xr[r] = s.ld(s.f(s.fXr + r, t));
                   in.wp[0] = t > 0 ? s.ld(s.p(s.fW, t - 1, 0)) : 0.0;
                   in.wp[1] = t > 0 ? s.ld(s.p(s.fW, t - 1, 1)) : 0.0;
               },
               [&](int t, const Rec &in) {
                   const Dyn &d = in.d;
                   const double (&wp)[2] = in.wp;
                   // adjoint and the size of its terms
                   {
                       const double l2 = fma(a02n, lam[0], fma(a12n, lam[1], lam[2]));
                       const double m2 = fma(fabs(a02n), la[0], fma(fabs(a12n), la[1], la[2]));
                       double qa[3];
#pragma unroll
                       for (int r = 0; r < 3; ++r) {
                           const double q = t + 1 < T ? p.q[r] * qg * p.growth : p.qf[r];
                           qa[r] = fabs(in.qe[r]) + 4.0 * q * fabs(in.xr[r]);
                       }
                       lam[0] += in.qe[0]; lam[1] += in.qe[1]; lam[2] = l2 + in.qe[2];
                       la[0] += qa[0]; la[1] += qa[1]; la[2] = m2 + qa[2];
                   }
#pragma unroll
                   for (int c = 0; c < 2; ++c) {
                       const double gr = input_cost_grad(p, c, t, T, wp[c], wc[c], wn[c])
                                         + fma(d.b[0][c], lam[0], fma(d.b[1][c], lam[1], d.b[2][c] * lam[2]));
                       double mag = fabs(2.0 * p.rw[c] * wc[c]) + fma(fabs(d.b[0][c]), la[0], fma(fabs(d.b[1][c]), la[1], fabs(d.b[2][c]) * la[2]));
                       if (t > 0) mag += fabs(2.0 * p.rdw[c] * (wc[c] - wp[c]));
                       if (t + 1 < T) mag += fabs(2.0 * p.rdw[c] * (wn[c] - wc[c]));
                       const bool up = bit(S.ru[c], t), dn = bit(S.rl[c], t);
                       double cur = 0.0;
                       if ((up || dn) && !bit(fixedR[c], t + 1)) {
                           cur = curn[c] - grn[c];
                           const double tol = kNoiseTol * (dmax(mag, magn[c]) + 1.0);
                           if (up && cur < -tol) drop.ru[c] |= one(t);
                           if (dn && -cur < -tol) drop.rl[c] |= one(t);
                       }
                       s.st(s.p(s.fGr, t, c), gr);
                       s.st(s.p(s.fCur, t, c), cur);
                       s.st(s.p(s.fMag, t, c), mag);
                       curn[c] = cur; grn[c] = gr; magn[c] = mag;
                       wn[c] = wc[c];
                       wc[c] = wp[c];
                   }
                   a02n = d.a02; a12n = d.a12;
                   qg = qg / p.growth;
               });
}

// Forward pass over what rounds_adjoint stored: the remaining multipliers from
// gr_t + mu_t + nu_{t-1} - nu_t = 0 (mu the box multiplier of stage t, nu_t the multiplier of
// the rate constraint (t, t + 1)).  Along a chain of stages tied to a fixed one the sums run
// forward, nu_t = gr_t + mu_t + nu_{t-1}, and a box at the end of the chain takes what is left.
// A chain that also passes over boxes (a steer that zigzags between its box and one rate step
// off it: the box is a whole number of rate steps wide) has one constraint more than controls,
// so its multipliers are not unique: what is carried forward is the interval [flo, fhi] of the
// values nu_{t-1} can take with every multiplier so far of the right sign, and the working set
// passes if the interval never empties.  The constraint at which it does is marked in `drop`,
// as is every uniquely determined multiplier of the wrong sign; returns whether every free
// degree of freedom is stationary to rounding.
__device__ __forceinline__ bool rounds_multipliers(const Store &s, const Sets &S, const uint64_t (&fixedR)[2], Sets &drop,
                                                   double &check)
{
    struct Rec { double gr[2], cur[2], mag[2]; };
    const int T = s.T;
    bool stat_ok = true;
    double flo[2] = {0.0, 0.0}, fhi[2] = {0.0, 0.0}, magp[2] = {0.0, 0.0};
    sweep<Rec>(0, T - 1, 1,
               [&](int t, Rec &in) {
#pragma unroll
                   for (int c = 0; c < 2; ++c) {
                       in.gr[c] = s.ld(s.p(s.fGr, t, c)); in.cur[c] = s.ld(s.p(s.fCur, t, c)); in.mag[c] = s.ld(s.p(s.fMag, t, c));
                   }
               },
               [&](int t, const Rec &in) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const double gr = in.gr[c], cur = in.cur[c], mag = in.mag[c];
            check += gr + cur;
            const bool fx = bit(fixedR[c], t);
            const bool tie_prev = t > 0 && (bit(S.ru[c], t - 1) || bit(S.rl[c], t - 1));
            const bool up = bit(S.ru[c], t), dn = bit(S.rl[c], t);
            const bool bu = bit(S.bu[c], t), bl = bit(S.bl[c], t);
            const bool chained = tie_prev && fx;          // nu_{t-1} comes along a fixed chain
            const double plo = chained ? flo[c] : 0.0, phi = chained ? fhi[c] : 0.0;
            const double tol = kNoiseTol * (dmax(mag, magp[c]) + 1.0);
            double nlo = 0.0, nhi = 0.0;
            if ((up || dn) && bit(fixedR[c], t + 1)) {
                // nu_t = gr + mu_t + nu_{t-1}: mu_t >= 0 on an upper box, <= 0 on a lower one
                nlo = bl ? -kBig : gr + plo;
                nhi = bu ? kBig : gr + phi;
                if (up) {
                    if (nhi < -tol) { drop.ru[c] |= one(t); nlo = nhi = 0.0; }
                    else { nlo = dmax(nlo, 0.0); nhi = dmax(nhi, nlo); }
                } else {
                    if (nlo > tol) { drop.rl[c] |= one(t); nlo = nhi = 0.0; }
                    else { nhi = dmin(nhi, 0.0); nlo = dmin(nlo, nhi); }
                }
            } else if (bu || bl) {
                // mu_t = -(gr - cur) - nu_{t-1} for some nu_{t-1} in [plo, phi]
                const double need = -(gr - cur);
                if (bu && need - plo < -tol) drop.bu[c] |= one(t);
                if (bl && need - phi > tol) drop.bl[c] |= one(t);
            } else if (!fx && !tie_prev) {
                if (!(fabs(gr - cur) <= kNoiseTol * (mag + 1.0))) stat_ok = false;
            }
            flo[c] = nlo; fhi[c] = nhi;
            magp[c] = mag;
        }
               });
    return stat_ok;
}

// Backward Riccati sweep of the equality-constrained problem on the working set.  Per channel a
// stage is FREE, FIXED (dw = target - w: on its box, or a fixed number of rate steps before a
// stage that is) or TIED (dw = dw of the previous stage + what restores the rate constraint).
// With u_n the non-free part of the policy (sigma v + o) the stage policy is
//   u = u_n - Hi (Hux dx + Huv v + Huu u_n + hu),   Hi = inverse of Huu on the free controls,
// i.e. u = -(Lx dx + Lv v + k) with Lx = Hi Hux, Lv = Hi (Huv + Huu Sigma) - Sigma,
// k = Hi (hu + Huu o) - o, and the value function follows from the general affine policy:
//   P' = Hzz - L' Huz - M' L,  p' = hz - L' hu - M' k,  M = Huz - Huu L   (z = (dx, v)).
// Right-hand side: minus the gradient, either the one rounds_adjoint stored or (ADJOINT, the
// first round, whose multipliers nobody looks at) formed on the way by the adjoint recursion.
// Stores the gains and the feed-forward.
template <bool ADJOINT>
__device__ __forceinline__ void rounds_riccati(const Store &s, const Problem &p, const Sets &S, const uint64_t (&fixedR)[2],
                                               bool run, double &check)
{
    const int T = s.T;
    Riccati R;
    R.terminal(p.qf);
    double qg = 1.0;
    for (int t = 1; t < T; ++t) qg *= p.growth;
    double tau[2] = {0.0, 0.0};
    double lam[3] = {0.0, 0.0, 0.0}, a02n = 0.0, a12n = 0.0, wn[2] = {0.0, 0.0};
    struct Rec { Dyn d; double wp[2], qe[3], gr[2]; };
    double wc[2] = {s.ld(s.p(s.fW, T - 1, 0)), s.ld(s.p(s.fW, T - 1, 1))};
    sweep<Rec>(T - 1, 0, -1,
               [&](int t, Rec &in) {
                   load_dyn(s, t, in.d);
                   in.wp[0] = t > 0 ? s.ld(s.p(s.fW, t - 1, 0)) : 0.0;
                   in.wp[1] = t > 0 ? s.ld(s.p(s.fW, t - 1, 1)) : 0.0;
                   if (ADJOINT) load_qe(s, t, in.qe);
                   else { in.gr[0] = s.ld(s.p(s.fGr, t, 0)); in.gr[1] = s.ld(s.p(s.fGr, t, 1)); }
               },
               [&](int t, const Rec &in) {
        const Dyn &d = in.d;
        const double (&wp)[2] = in.wp;
        double sig[2], o[2], wr[2], we[2], wq[3], gr[2];
        bool fr[2];
        if (ADJOINT) {
            adjoint_step(in.qe, a02n, a12n, lam);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                gr[c] = input_cost_grad(p, c, t, T, wp[c], wc[c], wn[c])
                        + fma(d.b[0][c], lam[0], fma(d.b[1][c], lam[1], d.b[2][c] * lam[2]));
                check += gr[c];
                wn[c] = wc[c];
            }
            a02n = d.a02; a12n = d.a12;
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const bool fx = bit(fixedR[c], t);
            const bool up_prev = t > 0 && bit(S.ru[c], t - 1), dn_prev = t > 0 && bit(S.rl[c], t - 1);
            sig[c] = 0.0; o[c] = 0.0;
            fr[c] = false;
            if (fx) {
                double target;
                if (bit(S.bu[c], t)) target = p.hi(t, c);
                else if (bit(S.bl[c], t)) target = p.lo(t, c);
                else target = tau[c] - (bit(S.ru[c], t) ? p.fb.rr[c] : -p.fb.rr[c]);
                tau[c] = target;
                o[c] = target - wc[c];
            } else if (up_prev || dn_prev) {
                sig[c] = 1.0;
                o[c] = (up_prev ? p.fb.rr[c] : -p.fb.rr[c]) - (wc[c] - wp[c]);
            } else {
                fr[c] = true;
            }
            wr[c] = 2.0 * p.rw[c];
            we[c] = t > 0 ? 2.0 * p.rdw[c] : 0.0;
            if (!ADJOINT) gr[c] = in.gr[c];
            wc[c] = wp[c];
        }
        stage_weight(p, t, qg, wq);
        // Q-function blocks
        double BP[2][3];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                BP[c][j] = fma(d.b[0][c], R.Pxx(0, j), fma(d.b[1][c], R.Pxx(1, j), fma(d.b[2][c], R.Pxx(2, j), R.xu[j][c])));
        double h00 = wr[0] + we[0] + R.uu[0], h01 = R.uu[1], h11 = wr[1] + we[1] + R.uu[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            h00 = fma(BP[0][j], d.b[j][0], fma(d.b[j][0], R.xu[j][0], h00));
            h01 = fma(BP[0][j], d.b[j][1], fma(d.b[j][0], R.xu[j][1], h01));
            h11 = fma(BP[1][j], d.b[j][1], fma(d.b[j][1], R.xu[j][1], h11));
        }
        const double Huu[2][2] = {{h00, h01}, {h01, h11}};
        double Hux[2][3];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            Hux[c][0] = BP[c][0];
            Hux[c][1] = BP[c][1];
            Hux[c][2] = fma(d.a02, BP[c][0], fma(d.a12, BP[c][1], BP[c][2]));
        }
        double hu[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
            hu[c] = fma(d.b[0][c], R.px[0], fma(d.b[1][c], R.px[1], fma(d.b[2][c], R.px[2], R.pu[c]))) + gr[c];
        // inverse on the free controls
        double i00 = 0.0, i01 = 0.0, i11 = 0.0;
        if (fr[0] && fr[1]) {
            const double idet = rcp_nr(fma(h00, h11, -h01 * h01));
            i00 = h11 * idet; i01 = -h01 * idet; i11 = h00 * idet;
        } else if (fr[0]) {
            i00 = rcp_nr(h00);
        } else if (fr[1]) {
            i11 = rcp_nr(h11);
        }
        // G = Huv + Huu Sigma (2 x 2), hv = hu + Huu o
        double G[2][2], ho[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
#pragma unroll
            for (int e = 0; e < 2; ++e) G[r][e] = (r == e ? -we[e] : 0.0) + Huu[r][e] * sig[e];
            ho[r] = hu[r] + Huu[r][0] * o[0] + Huu[r][1] * o[1];
        }
        double Lx[2][3], Lv[2][2], k[2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            Lx[0][j] = fma(i00, Hux[0][j], i01 * Hux[1][j]);
            Lx[1][j] = fma(i01, Hux[0][j], i11 * Hux[1][j]);
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            Lv[0][e] = fma(i00, G[0][e], i01 * G[1][e]) - (e == 0 ? sig[0] : 0.0);
            Lv[1][e] = fma(i01, G[0][e], i11 * G[1][e]) - (e == 1 ? sig[1] : 0.0);
        }
        k[0] = fma(i00, ho[0], i01 * ho[1]) - o[0];
        k[1] = fma(i01, ho[0], i11 * ho[1]) - o[1];
        if (run) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
#pragma unroll
                for (int j = 0; j < 3; ++j) s.st(s.f(s.fLx + c * 3 + j, t), Lx[c][j]);
                s.st(s.f(s.fLv + c * 2, t), Lv[c][0]);
                s.st(s.f(s.fLv + c * 2 + 1, t), Lv[c][1]);
                s.st(s.p(s.fKv, t, c), k[c]);
            }
        }
        if (t > 0) {
            // M = Huz - Huu L
            double Mx[2][3], Mv[2][2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int j = 0; j < 3; ++j) Mx[r][j] = Hux[r][j] - fma(Huu[r][0], Lx[0][j], Huu[r][1] * Lx[1][j]);
#pragma unroll
                for (int e = 0; e < 2; ++e) Mv[r][e] = (r == e ? -we[e] : 0.0) - fma(Huu[r][0], Lv[0][e], Huu[r][1] * Lv[1][e]);
            }
            // A' Pxx A
            const double m00 = R.xx[0], m01 = R.xx[1], m02 = R.xx[2], m11 = R.xx[3], m12 = R.xx[4], m22 = R.xx[5];
            const double n02 = fma(d.a02, m00, fma(d.a12, m01, m02));
            const double n12 = fma(d.a02, m01, fma(d.a12, m11, m12));
            const double n22 = fma(d.a02, m02 + n02, fma(d.a12, m12 + n12, m22));
            const double Hxx[6] = {wq[0] + m00, m01, n02, wq[1] + m11, n12, wq[2] + n22};
            Riccati N;
            int e6 = 0;
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = i; j < 3; ++j, ++e6)
                    N.xx[e6] = Hxx[e6] - fma(Lx[0][i], Hux[0][j], Lx[1][i] * Hux[1][j]) - fma(Mx[0][i], Lx[0][j], Mx[1][i] * Lx[1][j]);
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int e = 0; e < 2; ++e)   // Huv is diagonal: (Lx' Huv)[i][e] = -Lx[e][i] we[e]
                    N.xu[i][e] = Lx[e][i] * we[e] - fma(Mx[0][i], Lv[0][e], Mx[1][i] * Lv[1][e]);
            // Pvv' = We - Lv' Huv - Mv' Lv
            N.uu[0] = we[0] + Lv[0][0] * we[0] - fma(Mv[0][0], Lv[0][0], Mv[1][0] * Lv[1][0]);
            N.uu[1] = Lv[1][0] * we[1] - fma(Mv[0][0], Lv[0][1], Mv[1][0] * Lv[1][1]);
            N.uu[2] = we[1] + Lv[1][1] * we[1] - fma(Mv[0][1], Lv[0][1], Mv[1][1] * Lv[1][1]);
            const double p2 = fma(d.a02, R.px[0], fma(d.a12, R.px[1], R.px[2]));
            const double hx[3] = {R.px[0], R.px[1], p2};
#pragma unroll
            for (int i = 0; i < 3; ++i)
                N.px[i] = hx[i] - fma(Lx[0][i], hu[0], Lx[1][i] * hu[1]) - fma(Mx[0][i], k[0], Mx[1][i] * k[1]);
#pragma unroll
            for (int e = 0; e < 2; ++e)
                N.pu[e] = -fma(Lv[0][e], hu[0], Lv[1][e] * hu[1]) - fma(Mv[0][e], k[0], Mv[1][e] * k[1]);
            R = N;
        }
        qg = qg / p.growth;
               });
}

// Forward sweep of a round: the step of the working-set problem, the new controls (a control on
// its box lands on it exactly), the rollout for the next adjoint, and every constraint the new
// point violates (`add`).  A working set with more constraints than controls on a stretch (a box
// at both ends of a chain of rate constraints) cannot hold everywhere unless its values happen
// to fit: members the new point does not sit on are returned in `loose`.
__device__ __forceinline__ void rounds_forward(const Store &s, const Problem &p, const Sets &S, bool run, Sets &add,
                                               Sets &loose, double &check)
{
    struct X { double w[2]; RollIn r; };
    double wpn[2] = {0.0, 0.0};   // new controls of the previous stage
    Roller ro;
    ro.start(p);
    forward<X>(s, [&](int t, X &x) { x.w[0] = s.ld(s.p(s.fW, t, 0)); x.w[1] = s.ld(s.p(s.fW, t, 1)); load_roll(s, t, x.r); },
               [&](int t, const double (&u)[2], const double (&)[3], const Dyn &d, const X &x) {
        double wst[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            double wnew = x.w[c] + u[c];
            if (bit(S.bu[c], t)) wnew = p.hi(t, c);
            else if (bit(S.bl[c], t)) wnew = p.lo(t, c);
            wst[c] = wnew;
            check += wnew;
            if (p.hb(t, c)) {
                if (wnew - p.hi(t, c) > kFeasTol) add.bu[c] |= one(t);
                if (p.lo(t, c) - wnew > kFeasTol) add.bl[c] |= one(t);
            }
            if (t > 0 && p.hr(c)) {
                const double dv = wnew - wpn[c];
                if (dv - p.fb.rr[c] > kFeasTol) add.ru[c] |= one(t - 1);
                if (-dv - p.fb.rr[c] > kFeasTol) add.rl[c] |= one(t - 1);
                if (bit(S.ru[c], t - 1) && p.fb.rr[c] - dv > kFeasTol) loose.ru[c] |= one(t - 1);
                if (bit(S.rl[c], t - 1) && p.fb.rr[c] + dv > kFeasTol) loose.rl[c] |= one(t - 1);
            }
            if (run) s.st(s.p(s.fW, t, c), wnew);
            wpn[c] = wnew;
        }
        ro.step(s, p, t, wst[0], wst[1], d, x.r);   // for the next round's adjoint sweep
    });
}

// Starts from the interior-point iterate on the lanes with `run`; returns true where a point is
// certified (then w holds it), otherwise w is put back as it was (the interior point's
// multipliers are never touched, so it can go on from there).
__device__ __forceinline__ bool pdas(const Store &s, const Problem &p, int max_rounds, bool run, int &used)
{
    used = 0;
    const int T = s.T;
    const bool ran = run;
    Sets S;
    S.clear();
    // working set off the iterate (multiplier above slack), backup of the controls; the interior
    // point leaves the rollout of its last iterate behind
    if (__any(run)) {
        struct Rec { double w[2], lu[2], ll[2], nu[2], nl[2]; };
        double wpv[2] = {0.0, 0.0};
        sweep<Rec>(0, T - 1, 1,
                   [&](int t, Rec &in) {
#pragma unroll
                       for (int c = 0; c < 2; ++c) {
                           in.w[c] = s.ld(s.p(s.fW, t, c));
                           in.lu[c] = s.ld(s.p(s.fLU, t, c)); in.ll[c] = s.ld(s.p(s.fLL, t, c));
                           in.nu[c] = t > 0 ? s.ld(s.p(s.fNU, t - 1, c)) : 0.0;
                           in.nl[c] = t > 0 ? s.ld(s.p(s.fNL, t - 1, c)) : 0.0;
                       }
                   },
                   [&](int t, const Rec &in) {
#pragma unroll
                       for (int c = 0; c < 2; ++c) {
                           const double wc = in.w[c];
                           if (run) s.st(s.p(s.fBak, t, c), wc);
                           if (p.hb(t, c)) {
                               if (in.lu[c] > p.hi(t, c) - wc) S.bu[c] |= one(t);
                               else if (in.ll[c] > wc - p.lo(t, c)) S.bl[c] |= one(t);
                           }
                           if (t > 0 && p.hr(c)) {
                               const double dv = wc - wpv[c];
                               if (in.nu[c] > p.fb.rr[c] - dv) S.ru[c] |= one(t - 1);
                               else if (in.nl[c] > p.fb.rr[c] + dv) S.rl[c] |= one(t - 1);
                           }
                           wpv[c] = wc;
                       }
                   });
    }
    bool accepted = false;
    bool violated = true;   // no point of this working set has been checked yet
    for (int round = 0;; ++round) {
        if (!__any(run)) break;
        if (run) used = round;
        uint64_t fixedR[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) fixedR[c] = fixed_from_right(S.bu[c] | S.bl[c], S.ru[c] | S.rl[c], T);
        double check = 0.0;
        if (round == 0) {
            // the interior-point iterate does not solve its working set's problem yet: its
            // multipliers decide nothing, so the gradient is formed inside the Riccati sweep
            rounds_riccati<true>(s, p, S, fixedR, run, check);
            if (run && !finite_d(check)) run = false;
        } else {
            Sets drop;
            drop.clear();
            rounds_adjoint(s, p, S, fixedR, drop);
            const bool stat_ok = rounds_multipliers(s, S, fixedR, drop, check);
            bool dropped = false;
#pragma unroll
            for (int c = 0; c < 2; ++c) dropped = dropped || (drop.bu[c] | drop.bl[c] | drop.ru[c] | drop.rl[c]) != 0;
            const bool fin = finite_d(check);
            if (run && fin && stat_ok && !violated && !dropped) { accepted = true; run = false; }
            if (run && (round >= max_rounds || !fin)) run = false;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                S.bu[c] &= ~drop.bu[c]; S.bl[c] &= ~drop.bl[c];
                S.ru[c] &= ~drop.ru[c]; S.rl[c] &= ~drop.rl[c];
            }
            if (!__any(run)) break;
#pragma unroll
            for (int c = 0; c < 2; ++c) fixedR[c] = fixed_from_right(S.bu[c] | S.bl[c], S.ru[c] | S.rl[c], T);
            rounds_riccati<false>(s, p, S, fixedR, run, check);
        }
        if (!__any(run)) break;
        Sets add, loose;
        add.clear();
        loose.clear();
        double chk2 = 0.0;
        rounds_forward(s, p, S, run, add, loose, chk2);
        if (run && !finite_d(chk2)) run = false;
        violated = false;   // "the working set changed after the point was computed"
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            violated = violated || (add.bu[c] | add.bl[c] | add.ru[c] | add.rl[c] | loose.ru[c] | loose.rl[c]) != 0;
            S.bu[c] = (S.bu[c] | add.bu[c]) & ~add.bl[c]; S.bl[c] = (S.bl[c] | add.bl[c]) & ~add.bu[c];
            S.ru[c] = ((S.ru[c] & ~loose.ru[c]) | add.ru[c]) & ~add.rl[c];
            S.rl[c] = ((S.rl[c] & ~loose.rl[c]) | add.rl[c]) & ~add.ru[c];
        }
    }
    if (__any(ran && !accepted)) {
        struct Rec { double w[2]; };
        const bool back = ran && !accepted;
        sweep<Rec>(0, T - 1, 1, [&](int t, Rec &in) { in.w[0] = s.ld(s.p(s.fBak, t, 0)); in.w[1] = s.ld(s.p(s.fBak, t, 1)); },
                   [&](int t, const Rec &in) {
                       if (back) { s.st(s.p(s.fW, t, 0), in.w[0]); s.st(s.p(s.fW, t, 1), in.w[1]); }
                   });
    }
    return accepted;
}

}  // namespace stagewise
}  // namespace mpcx
