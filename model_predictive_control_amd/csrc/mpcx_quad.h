// Four lanes per MPC instance (horizon TT, a finite box on both free channels; built for TT = 10,
// mpc.py's controller).
//
// The condensed QP has 2 free controls per stage; variable 2t + c belongs to channel c.  The
// symmetric matrices are cut into 2x2 blocks (I, J) over the stages, and the four lanes of a quad
// own the four entries of EVERY block: lane q = c + 2e holds entry (c, e), i.e.
//     A[I][J] = K(2I + c, 2J + e),  J <= I.
// All four lanes therefore have the same shapes and run one instruction stream; at TT = 10 a lane
// keeps 55 matrix entries (the lane-pair layout needs 110 and spills), so the whole solve stays in
// registers, and a wavefront's 16 instances need under 40 KB of LDS, so four wavefronts share a CU.
//
// Vectors of length 2TT are kept per channel: lanes (c, 0) and (c, 1) both hold the TT entries of
// channel c.  The inequality constraints of channel c are split between those two lanes by kind:
// lane e = 0 owns the TT box constraints, lane e = 1 the TT - 1 rate constraints.  Both kinds are
// two-sided bounds on a linear form of the channel's controls,
//     box:  w[t]              rate:  w[t+1] - w[t]
// selected per lane, so that the two kinds still run one instruction stream; the rate lanes' slot
// TT - 1 is unused (its bounds are pushed out to +-1e30 and its contributions are zeroed).
//
// Values cross the quad by DPP quad permutes: x1 flips c (the other channel's lane with the same
// e), x2 flips e (the same channel's other lane), x3 flips both, bcast<k> reads lane k.
#include "mpcx_device.h"

namespace mpcx {
namespace quad {

typedef __attribute__((address_space(3))) double lds_double;
constexpr int kInst = kWave / 4;             // instances per wavefront
constexpr int kDynPerStage = 11;             // a02, a12, bt[3][2], ct[3]

__device__ __forceinline__ constexpr int bi(int I, int J) { return I * (I + 1) / 2 + J; }   // J <= I

template <int CTRL>
__device__ __forceinline__ double dpp_double(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double x1(double x) { return dpp_double<0xB1>(x); }   // quad_perm [1,0,3,2]
__device__ __forceinline__ double x2(double x) { return dpp_double<0x4E>(x); }   // quad_perm [2,3,0,1]
__device__ __forceinline__ double x3(double x) { return dpp_double<0x1B>(x); }   // quad_perm [3,2,1,0]
template <int K>
__device__ __forceinline__ double bcast(double x) { return dpp_double<K * 0x55>(x); }
__device__ __forceinline__ double quad_sum(double x) { x = x + x1(x); return x + x2(x); }
__device__ __forceinline__ double quad_max(double x) { x = dmax(x, x1(x)); return dmax(x, x2(x)); }
__device__ __forceinline__ bool quad_any(bool b)
{
    int v = b ? 1 : 0;
    v |= __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);
    v |= __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);
    return v != 0;
}

// 1 / sqrt(x): hardware estimate and one Newton step (backward error of the factor at rounding level)
__device__ __forceinline__ double rsqrt_1(double x)
{
    const double r = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * r), r, 1.0);
    return fma(0.5 * r, e, r);
}

template <int TT>
struct State {
    static constexpr int NBLK = TT * (TT + 1) / 2;
    static constexpr int kLdsH = (NBLK + TT) * kWave;      // H entries and g, one slot per lane
    double A[NBLK];                  // K(2I + c, 2J + e), J <= I; then its Cholesky factor (diagonal slots: 1 / L(2J, 2J))
    double l10[TT], r1[TT];          // rest of the diagonal blocks' factor: L(2J+1, 2J), 1 / L(2J+1, 2J+1) (identical on the four lanes)
    double lamU[TT], lamL[TT];       // multipliers of the lane's constraints, upper / lower side
    double isu[TT], isl[TT];         // reciprocal slacks / dual steps / active-set flags
    double ws[TT];                   // controls of channel c (the lane's rows)
    double hw[TT], dwa[TT], dw[TT];  // channel c: H w + g, predictor, corrector / step
    lds_double *lds;                 // H [NBLK][64], g [TT][64]; + lane
    lds_double *dyn;                 // [kDynPerStage TT][kInst]; + instance
    bool rowodd, colodd, diag;       // lane parities as wave masks (no vector registers)
    double bu0, buR, bl0, blR;       // bounds of the lane's constraints: stage 0, later stages
    double lo0, hi0;                 // first-stage window of channel c
    __device__ __forceinline__ void bind(lds_double *h, lds_double *d, int lane)
    {
        rowodd = (lane & 1) != 0;
        colodd = (lane & 2) != 0;
        diag = rowodd == colodd;
        lds = h + lane;
        dyn = d + (lane >> 2);
    }
    __device__ __forceinline__ double H(int I, int J) const { return ((const volatile lds_double *)lds)[bi(I, J) * kWave]; }
    __device__ __forceinline__ double g(int I) const { return ((const volatile lds_double *)lds)[(NBLK + I) * kWave]; }
    // the rate lanes' slot TT - 1 is unused: bounds far out, contributions zeroed by used()
    __device__ __forceinline__ double bu(int t) const { return t == 0 && TT > 1 ? bu0 : (t == TT - 1 ? (colodd ? 1e30 : (TT > 1 ? buR : bu0)) : buR); }
    __device__ __forceinline__ double bl(int t) const { return t == 0 && TT > 1 ? bl0 : (t == TT - 1 ? (colodd ? -1e30 : (TT > 1 ? blR : bl0)) : blR); }
    __device__ __forceinline__ double used(int t, double x) const { return t == TT - 1 ? (colodd ? 0.0 : x) : x; }
    __device__ __forceinline__ bool is_used(int t) const { return t == TT - 1 ? !colodd : true; }
    // the constraint's linear form applied to a channel vector: x[t] (box) or x[t+1] - x[t] (rate)
    __device__ __forceinline__ double form(const double (&x)[TT], int t) const
    {
        if (t + 1 < TT) {
            const double d = x[t + 1] - x[t];
            return colodd ? d : x[t];
        }
        return colodd ? -x[t] : x[t];
    }
    // the lane's part of the transpose of form(): z[t] (box) or z[t-1] - z[t] (rate)
    __device__ __forceinline__ double form_t(const double (&z)[TT], int t) const
    {
        if (t > 0) {
            const double d = z[t - 1] - z[t];
            return colodd ? d : z[t];
        }
        return colodd ? -z[t] : z[t];
    }
    // bounds of the lane's constraints from the channel's box b, rate rr and first-stage window
    __device__ __forceinline__ void set_bounds(double b, double rr, double lo0_, double hi0_)
    {
        lo0 = lo0_; hi0 = hi0_;
        bu0 = colodd ? rr : hi0_;  bl0 = colodd ? -rr : lo0_;
        buR = colodd ? rr : b;     blR = colodd ? -rr : -b;
    }
};

// channel vector C' z of per-constraint values z (the transpose of State::form), summed over the
// two kinds; z of the unused slot must be zero
template <int TT>
__device__ __forceinline__ void form_transpose(const State<TT> &s, const double (&z)[TT], double (&out)[TT])
{
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const double cv = s.form_t(z, t);
        out[t] = cv + x2(cv);
    }
}

// K += C' diag(dd) C for the channel's constraints: everything lands on the same-parity entries,
// which the diagonal lanes hold.  dd of the unused slot must be zero.
template <int TT>
__device__ __forceinline__ void add_barrier(State<TT> &s, const double (&dd)[TT])
{
    double dbox[TT], drate[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const double oth = x2(dd[t]);
        dbox[t] = s.colodd ? oth : dd[t];
        drate[t] = s.colodd ? dd[t] : oth;
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        double d = dbox[t] + drate[t];
        if (t > 0) d += drate[t - 1];
        s.A[bi(t, t)] += s.diag ? d : 0.0;
        if (t + 1 < TT) s.A[bi(t + 1, t)] -= s.diag ? drate[t] : 0.0;
    }
}

// hw <- channel c of H w + g; optionally K <- H on the way.
template <int TT, bool COPY_K>
__device__ __forceinline__ void matvec(State<TT> &s)
{
    double rowp[TT], colp[TT], wo[TT];   // wo: controls of channel e (the lane's columns)
#pragma unroll
    for (int I = 0; I < TT; ++I) {
        rowp[I] = 0.0; colp[I] = 0.0;
        const double o = x1(s.ws[I]);
        wo[I] = s.diag ? s.ws[I] : o;
    }
#pragma unroll
    for (int I = 0; I < TT; ++I) {
#pragma unroll
        for (int J = 0; J < I; ++J) {
            const double h = s.H(I, J);
            rowp[I] = fma(h, wo[J], rowp[I]);     // row 2I + c, column 2J + e
            colp[J] = fma(h, s.ws[I], colp[J]);     // by symmetry: row 2J + e, column 2I + c
            if (COPY_K) s.A[bi(I, J)] = h;
        }
        const double hd = s.H(I, I);
        rowp[I] = fma(hd, wo[I], rowp[I]);
        if (COPY_K) s.A[bi(I, I)] = hd;
    }
#pragma unroll
    for (int I = 0; I < TT; ++I) {
        const double rs = rowp[I] + x2(rowp[I]);          // both column parities of row 2I + c
        const double cp = colp[I] + x1(colp[I]);          // rows of channel e, both row parities of the source
        const double cpx = x2(cp);                        // (a lane permute must not sit inside a select: it would run
        const double cpc = s.diag ? cp : cpx;             //  with its source lanes switched off)  ... of channel c
        s.hw[I] = s.g(I) + rs + cpc;
    }
}

// In-place Cholesky of the quad-distributed matrix, by 2x2 blocks.
template <int TT>
__device__ __forceinline__ void chol(State<TT> &s)
{
#pragma unroll
    for (int J = 0; J < TT; ++J) {
        const double dj = s.A[bi(J, J)];
        const double a = bcast<0>(dj), b = bcast<1>(dj), c = bcast<3>(dj);
        const double r0 = rsqrt_1(a);
        const double l10 = b * r0;
        const double r1 = rsqrt_1(fma(-l10, l10, c));
        s.A[bi(J, J)] = r0; s.l10[J] = l10; s.r1[J] = r1;   // the diagonal block's own slot is free from here on
        double L[TT], X2[TT], CO[TT], CT[TT];
#pragma unroll
        for (int I = J + 1; I < TT; ++I) {   // panel block (I, J) times the inverse transpose of the diagonal factor
            const double own = s.A[bi(I, J)];
            const double t = own * r0;
            const double tp = x2(t);
            L[I] = s.colodd ? fma(-tp, l10, own) * r1 : t;
            s.A[bi(I, J)] = L[I];
        }
#pragma unroll
        for (int I = J + 1; I < TT; ++I) {
            const double xa = x1(L[I]), xb = x2(L[I]), xc = x3(L[I]);
            X2[I] = xb;                        // L(2I + c, 2J + 1 - e)
            CO[I] = s.diag ? L[I] : xa;        // L(2I + e, 2J + e)
            CT[I] = s.diag ? xb : xc;          // L(2I + e, 2J + 1 - e)
        }
#pragma unroll
        for (int I = J + 1; I < TT; ++I) {
#pragma unroll
            for (int K = J + 1; K <= I; ++K)
                s.A[bi(I, K)] = fma(-L[I], CO[K], fma(-X2[I], CT[K], s.A[bi(I, K)]));
        }
    }
}

// r (channel c) <- channel c of K^-1 r, using the factor left by chol().
template <int TT>
__device__ __forceinline__ void solve(const State<TT> &s, double (&r)[TT])
{
    double yE[TT], yO[TT], ys[TT];
#pragma unroll
    for (int I = 0; I < TT; ++I) {
        double p = 0.0;
#pragma unroll
        for (int J = 0; J < I; ++J) p = fma(s.A[bi(I, J)], ys[J], p);
        const double acc = I > 0 ? r[I] - (p + x2(p)) : r[I];
        const double aE = bcast<0>(acc), aO = bcast<1>(acc);
        yE[I] = aE * s.A[bi(I, I)];
        yO[I] = fma(-s.l10[I], yE[I], aO) * s.r1[I];
        ys[I] = s.colodd ? yO[I] : yE[I];
    }
#pragma unroll
    for (int kk = 0; kk < TT; ++kk) {
        const int K = TT - 1 - kk;
        double sE = 0.0, sO = 0.0;
        if (K < TT - 1) {
            double sa = 0.0;
#pragma unroll
            for (int I = K + 1; I < TT; ++I) sa = fma(s.A[bi(I, K)], r[I], sa);
            const double st = sa + x1(sa);     // column 2K + e over both row parities
            sE = bcast<0>(st);
            sO = bcast<2>(st);
        }
        const double zO = (yO[K] - sO) * s.r1[K];
        const double zE = fma(-s.l10[K], zO, yE[K] - sE) * s.A[bi(K, K)];
        r[K] = s.rowodd ? zO : zE;
    }
}

// Predictor-corrector interior point on the quad: the formulas of the lane-pair kernel
// (mpcx_pair.h, ipm), every constraint loop over the lane's own kind and the scalar reductions
// completed over the quad.
template <int TT>
__device__ __forceinline__ void ipm(State<TT> &s, int max_iter, bool fixed, bool skip, int &status, int &iters)
{
    constexpr double kInvM = 1.0 / (4.0 * TT + 4.0 * (TT - 1));
    if (!skip) {   // lanes that sit this solve out keep their controls
        const double wc = 0.5 * (s.lo0 + s.hi0);
#pragma unroll
        for (int t = 0; t < TT; ++t) s.ws[t] = wc;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const double f = s.form(s.ws, t);
            s.lamU[t] = kMu0 * rcp_nr(dmax(s.bu(t) - f, kSlackMin));
            s.lamL[t] = kMu0 * rcp_nr(dmax(f - s.bl(t), kSlackMin));
        }
    }
    bool done = skip;
    int my_iters = 0;
    status = MPCX_ST_FAILED;
    for (int it = 0;; ++it) {
        if (fixed) {
            if (!done && it >= max_iter) { done = true; my_iters = it; }
            if (__all(done)) break;
        }
        matvec<TT, true>(s);
        double musum = 0.0, dd[TT], dl[TT], ctl[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const double f = s.form(s.ws, t);
            const double su = dmax(s.bu(t) - f, kSlackMin), sl = dmax(f - s.bl(t), kSlackMin);
            const double fu = rcp_raw(su), fl = rcp_raw(sl);
            s.isu[t] = fu; s.isl[t] = fl;
            const double lu = s.lamU[t], ll = s.lamL[t];
            dd[t] = s.used(t, fma(lu, fu, ll * fl));
            dl[t] = s.used(t, lu - ll);
            musum += s.used(t, fma(su, lu, sl * ll));
        }
        add_barrier(s, dd);
        form_transpose(s, dl, ctl);
        const double mu = quad_sum(musum) * kInvM;
        double rdmax = 0.0, scale = 1.0;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const double hwi = s.hw[t], gi = s.g(t), ci = ctl[t];
            rdmax = dmax(rdmax, fabs(hwi + ci));
            scale = dmax(scale, dmax(fabs(hwi - gi), dmax(fabs(gi), fabs(ci))));
        }
        rdmax = quad_max(rdmax);
        scale = quad_max(scale) + 1.0;
        const bool conv = rdmax <= kTolD * scale && mu <= kTolMu;
#ifdef MPCX_QUAD_TRACE
        if (blockIdx.x == 0 && threadIdx.x < 4)
            printf("  ipm it %d lane %d mu %.6g rdmax %.6g scale %.6g ws0 %.9g ws1 %.9g ws9 %.9g hw0 %.9g ctl0 %.9g A00 %.9g A10 %.9g\n", it, (int)threadIdx.x, mu, rdmax, scale,
                   s.ws[0], s.ws[1], s.ws[TT - 1], s.hw[0], ctl[0], s.A[0], s.A[1]);
#endif
        if (!done && conv) { done = true; status = MPCX_ST_OK; my_iters = it; }
        if (!done && (mu < kMuFloor || it >= max_iter)) {
            done = true;
            my_iters = it;
            status = (rdmax <= kLooseD * scale && mu <= kLooseMu) ? MPCX_ST_INACCURATE : MPCX_ST_FAILED;
        }
        if (__all(done)) break;

        chol(s);
        // predictor: K dwa = -(H w + g)
#pragma unroll
        for (int t = 0; t < TT; ++t) s.dwa[t] = -s.hw[t];
        solve(s, s.dwa);

        double rp = 0.0, rdm = 1.0;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const double d = s.used(t, s.form(s.dwa, t));
            const double xu = d * s.isu[t], xl = -d * s.isl[t];
            rp = dmax(rp, dmax(xu, xl));
            rdm = dmax(rdm, dmax(1.0 - xu, 1.0 - xl));
        }
        rp = quad_max(rp);
        rdm = quad_max(rdm);
        const double apa = rp > 1.0 ? rcp_nr(rp) : 1.0;
        const double ada = rcp_nr(rdm);
        const double back = 1.0 - dmin(apa, ada);
        const double sm = back * back * back * mu;   // sigma * mu, sigma = (1 - alpha_aff)^3

        // corrector as a correction to the predictor: K (dw - dwa) = -C' q
        double z[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const double d = s.form(s.dwa, t);
            const double iu = s.isu[t], il = s.isl[t];
            const double xu = d * iu, xl = -d * il;
            const double qu = fma(sm, iu, s.lamU[t] * xu * (xu - 1.0));
            const double ql = fma(sm, il, s.lamL[t] * xl * (xl - 1.0));
            z[t] = s.used(t, ql - qu);
        }
        form_transpose(s, z, s.dw);
        solve(s, s.dw);
#pragma unroll
        for (int t = 0; t < TT; ++t) s.dw[t] += s.dwa[t];

        // dual steps (stored over the reciprocal slacks) and the two step lengths
        double rp2 = 0.0, rd2 = 0.0;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const double da = s.form(s.dwa, t), d = s.form(s.dw, t);
            const double iu = s.isu[t], il = s.isl[t], lu = s.lamU[t], ll = s.lamL[t];
            const double xu = da * iu, xl = -da * il;
            const double qu = fma(sm, iu, lu * xu * (xu - 1.0));
            const double ql = fma(sm, il, ll * xl * (xl - 1.0));
            const double zu = d * iu, zl = -d * il;
            const double dlu = fma(lu, zu, qu - lu), dll = fma(ll, zl, ql - ll);
            rp2 = dmax(rp2, s.used(t, dmax(zu, zl)));
            rd2 = dmax(rd2, s.used(t, dmax(-dlu * rcp_raw(lu), -dll * rcp_raw(ll))));
            s.isu[t] = dlu; s.isl[t] = dll;
        }
        rp2 = quad_max(rp2);
        rd2 = quad_max(rd2);
        const double eta = dmin(kEtaCap, dmax(0.99, 1.0 - mu));
        const double ap = rp2 > eta ? eta * rcp_nr(rp2) : 1.0;
        const double ad = rd2 > eta ? eta * rcp_nr(rd2) : 1.0;
        // a factorisation that broke down leaves NaN in the step; max() would hide it, a sum does not
        double chk = 0.0;
#pragma unroll
        for (int t = 0; t < TT; ++t) chk += s.dw[t] + s.isu[t] + s.isl[t];
        if (!done && !finite_d(quad_sum(chk))) {
            done = true;
            my_iters = it;
            status = (rdmax <= kLooseD * scale && mu <= kLooseMu) ? MPCX_ST_INACCURATE : MPCX_ST_FAILED;
        }
        if (!done) {
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                s.ws[t] = fma(ap, s.dw[t], s.ws[t]);
                s.lamU[t] = fma(ad, s.isu[t], s.lamU[t]);
                s.lamL[t] = fma(ad, s.isl[t], s.lamL[t]);
            }
        }
        }
    iters = my_iters;
}

#ifndef MPCX_QUAD_GUESS
#define MPCX_QUAD_GUESS 0.5
#endif
// first active set: multiplier > kQuadGuess x slack (see kPairGuess; mpc.py's controller, 32,768 / 65,536
// warm instances: 218 / 418 us with 1, 211 / 408 us with 0.5; with 0.3 or 0.15 the larger batch takes 404 us
// but one instance of the smaller one needs the second attempt, and its launch 275 us)
constexpr double kQuadGuess = MPCX_QUAD_GUESS;

// Primal-dual active-set rounds on the quad (mpcx_pair.h, pdas): one Newton step of the augmented
// Lagrangian of the equality-constrained QP on the current set per round, first-order multiplier
// update, set correction; a lane settles when its set stood still and the point passes the KKT
// test of the original QP.  Returns true where the instance settled; an instance that did not
// gets its controls back as they were on entry.
template <int TT>
__device__ __forceinline__ bool pdas(State<TT> &s, bool skip, int max_rounds, int *rounds_done = nullptr)
{
    double hmax = 0.0, wbak[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) { hmax = dmax(hmax, s.diag ? s.H(t, t) : 0.0); wbak[t] = s.ws[t]; }
    const double rho = 1e5 * quad_max(hmax);
    // initial set from the interior-point iterate
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const double f = s.form(s.ws, t);
        const bool au = s.is_used(t) && s.lamU[t] > kQuadGuess * (s.bu(t) - f), al = s.is_used(t) && s.lamL[t] > kQuadGuess * (f - s.bl(t));
        s.isu[t] = au ? 1.0 : 0.0; s.isl[t] = al ? 1.0 : 0.0;
        s.lamU[t] = au ? s.lamU[t] : 0.0; s.lamL[t] = al ? s.lamL[t] : 0.0;
    }
    bool settled = skip, accepted = false, changed = true;
    double scale = 1.0;
#pragma unroll 1
    for (int round = 0;; ++round) {
        matvec<TT, true>(s);
        double viol = 0.0, cact = 0.0, neg = 0.0, stat = 0.0;
        double t1[TT], ta[TT], dd[TT], gl[TT], ga[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const double f = s.form(s.ws, t), vu = f - s.bu(t), vl = s.bl(t) - f;
            const double au = s.isu[t], al = s.isl[t];
            const double cu = au * vu, cl = al * vl;
            t1[t] = s.lamU[t] - s.lamL[t];
            ta[t] = rho * (cu - cl);
            dd[t] = rho * (au + al);
            viol = dmax(viol, dmax(vu, vl));
            cact = dmax(cact, dmax(fabs(cu), fabs(cl)));
            neg = dmax(neg, dmax(-s.lamU[t], -s.lamL[t]));
        }
        add_barrier(s, dd);
        form_transpose(s, t1, gl);
        form_transpose(s, ta, ga);
        scale = 1.0;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            gl[t] += s.hw[t];
            stat = dmax(stat, fabs(gl[t]));
            const double gi = s.g(t);
            scale = dmax(scale, dmax(fabs(gi), fabs(s.hw[t] - gi)));
        }
        stat = quad_max(stat); viol = quad_max(viol); cact = quad_max(cact); neg = quad_max(neg);
        scale = quad_max(scale) + 1.0;
        // max() drops NaN operands, so the tests alone would wave a non-finite point through
        double chk = 0.0;
#pragma unroll
        for (int t = 0; t < TT; ++t) chk += gl[t] + ga[t] + s.ws[t];
        const bool fin = finite_d(quad_sum(chk));
        // Stationarity is asked relative to the gradient's terms or, where that is larger, to
        // |H| |w|: the multiplier update lam += rho c(w) carries the rounding of the controls (an
        // ulp, 1.1e-16 |w|) times rho = 1e5 max(diag H) into the multipliers, a floor of some
        // 5e-11 max(diag H) |w| that no further round gets below; with a large Hessian and a small
        // gradient the first scale alone would sit under that floor for ever.
        double wmax = 0.0;
#pragma unroll
        for (int t = 0; t < TT; ++t) wmax = dmax(wmax, fabs(s.ws[t]));
        const double stat_tol = 1e-10 * dmax(scale, 4e-5 * rho * quad_max(wmax));
        if (!settled && !changed && fin && stat <= stat_tol && viol <= 1e-12 && cact <= 1e-12 && neg <= 1e-9 * scale) {
            settled = true;
            accepted = true;
        }
        if (round >= max_rounds || __all(settled)) {
            if (rounds_done) *rounds_done = round;
            break;
        }

        chol(s);
#pragma unroll
        for (int t = 0; t < TT; ++t) s.dw[t] = -(gl[t] + ga[t]);
        solve(s, s.dw);
        const double ntol = -1e-9 * scale;
        bool ch = false;
        if (!settled) {
#pragma unroll
            for (int t = 0; t < TT; ++t) s.ws[t] += s.dw[t];
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const double f = s.form(s.ws, t), vu = f - s.bu(t), vl = s.bl(t) - f;
                const bool au = s.isu[t] != 0, al = s.isl[t] != 0;
                const double lu = au ? fma(rho, vu, s.lamU[t]) : 0.0, ll = al ? fma(rho, vl, s.lamL[t]) : 0.0;
                const bool nu = (au && lu >= ntol) || vu > 1e-12, nl = (al && ll >= ntol) || vl > 1e-12;
                ch = ch || nu != au || nl != al;
                s.isu[t] = nu ? 1.0 : 0.0; s.isl[t] = nl ? 1.0 : 0.0;
                s.lamU[t] = nu ? dmax(lu, 0.0) : 0.0; s.lamL[t] = nl ? dmax(ll, 0.0) : 0.0;
            }
        }
        changed = quad_any(ch);   // the four lanes hold one QP: the set moved if any part of it did
        }
    if (!accepted && !skip) {
#pragma unroll
        for (int t = 0; t < TT; ++t) s.ws[t] = wbak[t];
    }
    return accepted;
}

// Contiguous block of n doubles between global memory and LDS, 512 bytes per wave instruction; the
// records of the wave's instances lie back to back in every API array and are laid out in LDS
// with an odd stride.  A value that is not finite raises the flag of its record.
__device__ __forceinline__ constexpr int padded(int per) { return per | 1; }

template <int MAXN, int PER>
struct StageIn {
    static constexpr int K = (MAXN + kWave - 1) / kWave;
    double tmp[K];
    __device__ __forceinline__ void load(const double *__restrict__ g, int n, int lane)
    {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = lane + k * kWave;
            tmp[k] = g[i < n ? i : n - 1];
        }
    }
    __device__ __forceinline__ void put(int n, lds_double *dst, int lane, volatile __attribute__((address_space(3))) int *flag) const
    {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = lane + k * kWave;
            if (i < n) {
                const int rec = i / PER;
                ((volatile lds_double *)dst)[i + rec * (padded(PER) - PER)] = tmp[k];
                if (!finite_d(tmp[k])) flag[rec] = 1;
            }
        }
    }
};
template <int MAXN, int PER>
__device__ __forceinline__ void stage_out(double *__restrict__ g, int n, const lds_double *src, int lane)
{
    constexpr int K = (MAXN + kWave - 1) / kWave;
    double tmp[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        int i = lane + k * kWave;
        i = i < MAXN ? i : MAXN - 1;
        tmp[k] = ((const volatile lds_double *)src)[i + (i / PER) * (padded(PER) - PER)];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int i = lane + k * kWave;
        if (i < n) g[i] = tmp[k];
    }
}

struct Dyn {
    double a02, a12, bt[3][2], ct[3];
};

template <int TT>
__device__ __forceinline__ void load_dyn(const State<TT> &s, int t, Dyn &d)
{
    const volatile lds_double *p = (const volatile lds_double *)s.dyn + t * kDynPerStage * kInst;
    d.a02 = p[0];
    d.a12 = p[kInst];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        d.bt[r][0] = p[(2 + 2 * r) * kInst];
        d.bt[r][1] = p[(3 + 2 * r) * kInst];
        d.ct[r] = p[(8 + r) * kInst];
    }
}

// Linearisation point of one instance, read from the wave's LDS staging area.
template <int TT>
struct Lin {
    const volatile lds_double *u;     // [NU][TT]
    const volatile lds_double *yaw;   // xbar[2][0..TT-1] when given
    const double *cold;               // v.cold_start
    bool warm, has_yaw;
    double theta0;
    __device__ __forceinline__ double control(int r, int t) const
    {
        const double x = u[r * TT + t];
        return warm ? x : cold[r];
    }
};

// Stage dynamics of the instance into LDS: lane q of the quad evaluates stages q, q + 4, ...; the
// yaw of the linearisation point is rolled from the yaw rates exactly as predict_motion does
// (mpc_diff_drive.py:126-139), the rates passing through `om` ([TT][kInst] + instance).
template <int MODEL, int TT>
__device__ __forceinline__ void stage_eval(State<TT> &s, const mpcx_variant &v, const Lin<TT> &lp, volatile lds_double *om, int q)
{
    constexpr int NU = MODEL == MPCX_MODEL_DIFF_DRIVE ? 2 : 4;
    constexpr int NSLOT = (TT + 3) / 4;
    double u[NSLOT][4], th[NSLOT];
    SteerTerms k[NSLOT];
    int tj[NSLOT];
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        int t = 4 * j + q;
        t = t < TT ? t : TT - 1;       // a slot past the end repeats the last stage (same values, written twice)
        tj[j] = t;
#pragma unroll
        for (int r = 0; r < 4; ++r) u[j][r] = 0.0;
#pragma unroll
        for (int r = 0; r < NU; ++r) u[j][r] = lp.control(r, t);
        steer_terms<MODEL>(v, u[j], k[j]);
        om[t * kInst] = k[j].omega;
    }
    __builtin_amdgcn_wave_barrier();
    if (lp.has_yaw) {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) th[j] = lp.yaw[tj[j]];
    } else {
        double run = lp.theta0;
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) th[j] = run;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
#pragma unroll
            for (int j = 0; j < NSLOT; ++j) th[j] = tj[j] == t ? run : th[j];
            run = run + om[t * kInst] * v.dt;
        }
    }
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        double a02, a12, Bm[3][4], C[3], bt[3][2], ct[3];
        linearize_at<MODEL>(v, th[j], u[j], k[j], a02, a12, Bm, C);
        reduce_controls(v, Bm, C, bt, ct);
        volatile lds_double *p = (volatile lds_double *)s.dyn + tj[j] * kDynPerStage * kInst;
        p[0] = a02;
        p[kInst] = a12;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            p[(2 + 2 * r) * kInst] = bt[r][0];
            p[(3 + 2 * r) * kInst] = bt[r][1];
            p[(8 + r) * kInst] = ct[r];
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// Condensed Hessian entries and gradient entries of the lane, straight into LDS: the lane holds
// the sensitivities of the states to the controls of channel c (its rows) and channel e (its
// columns) and accumulates H(2I + c, 2J + e); the input-cost terms couple a channel with itself
// only, so they go to the diagonal lanes.
template <int TT>
__device__ __forceinline__ void condense(State<TT> &s, const mpcx_variant &v, const double (&x0)[3],
                                         const volatile lds_double *xref)
{
    constexpr int W = TT + 1, NBLK = State<TT>::NBLK;
    double rw[2], rdw[2];
    if (v.eq_mode == MPCX_EQ_NONE) {
        rw[0] = v.R[0]; rw[1] = v.R[1]; rdw[0] = v.Rd[0]; rdw[1] = v.Rd[1];
    } else if (v.eq_mode == MPCX_EQ_FIXED_STEER) {
        rw[0] = v.R[0]; rw[1] = v.R[2]; rdw[0] = v.Rd[0]; rdw[1] = v.Rd[2];
    } else {
        rw[0] = v.R[0] + v.R[2]; rw[1] = v.R[1] + v.R[3];
        rdw[0] = v.Rd[0] + v.Rd[2]; rdw[1] = v.Rd[1] + v.Rd[3];
    }
    const double rwc = s.rowodd ? rw[1] : rw[0], rdc = s.rowodd ? rdw[1] : rdw[0];
    double GC[3][TT], GE[3][TT], gacc[TT], acc[NBLK];
#pragma unroll
    for (int e = 0; e < NBLK; ++e) acc[e] = 0.0;
#pragma unroll
    for (int I = 0; I < TT; ++I) {
        gacc[I] = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) { GC[r][I] = 0.0; GE[r][I] = 0.0; }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        double d = 2.0 * rwc;
        if (t > 0) d += 2.0 * rdc;
        if (t + 1 < TT) d += 2.0 * rdc;
        acc[bi(t, t)] = s.diag ? d : 0.0;
        if (t > 0) acc[bi(t, t - 1)] = s.diag ? -2.0 * rdc : 0.0;
    }
    double p[3] = {x0[0], x0[1], x0[2]};
    double qg = 1.0;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        Dyn dy;
        load_dyn(s, t, dy);
        const double a02 = dy.a02, a12 = dy.a12;
        const double pn0 = p[0] + a02 * p[2] + dy.ct[0];
        const double pn1 = p[1] + a12 * p[2] + dy.ct[1];
        const double pn2 = p[2] + dy.ct[2];
        p[0] = pn0; p[1] = pn1; p[2] = pn2;
#pragma unroll
        for (int J = 0; J < t; ++J) {
            GC[0][J] = fma(a02, GC[2][J], GC[0][J]);
            GC[1][J] = fma(a12, GC[2][J], GC[1][J]);
            GE[0][J] = fma(a02, GE[2][J], GE[0][J]);
            GE[1][J] = fma(a12, GE[2][J], GE[1][J]);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            GC[r][t] = s.rowodd ? dy.bt[r][1] : dy.bt[r][0];
            GE[r][t] = s.colodd ? dy.bt[r][1] : dy.bt[r][0];
        }
        const int ts = t + 1;
        double Qs[3];
        if (ts < TT) {
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
            for (int I = 0; I < ts; ++I) {
                const double gi = q2 * GC[r][I];
                gacc[I] = fma(gi, e, gacc[I]);
#pragma unroll
                for (int J = 0; J <= I; ++J) acc[bi(I, J)] = fma(gi, GE[r][J], acc[bi(I, J)]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < NBLK; ++e) s.lds[e * kWave] = acc[e];
#pragma unroll
    for (int I = 0; I < TT; ++I) s.lds[(NBLK + I) * kWave] = gacc[I];
}

}  // namespace quad
}  // namespace mpcx
