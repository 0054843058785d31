// Two lanes per MPC instance (horizon TT = 5 or 10, a finite box on both free channels).
//
// The condensed QP has 2 free controls per stage; variable 2t + c belongs to channel c.
// Lane parity c owns channel c: its TT controls, its box and rate constraints (all of
// them couple controls of one channel only), and rows 2t + c of the symmetric matrices.
// That halves the per-lane state: at TT = 5 the whole solve stays in registers with nothing
// spilled to memory.  Partner values travel by DPP quad permutes (no LDS, no divergence).
//
// H (constant over the solve) lives in LDS by 2x2 blocks (I, J), J <= I, two entries per
// block and lane, in parity-relative order:
//     HS[I][J] = H(2I + c, 2J + c)        "same-parity column"
//     HO[I][J] = H(2I + c, 2J + 1 - c)    "opposite-parity column"
// so that the matrix-vector product runs one instruction stream on both lanes.
//
// K = H + barrier terms and its Cholesky factor live in registers in column-absolute order
// (E = even column 2J, D = odd column 2J + 1 of the lane's rows), which a lane reads from
// LDS through a parity-dependent base address.  The factorisation works by 2x2 blocks: the
// diagonal block is factored on both lanes redundantly, the panel below it is local to the
// row owner, and only the panel entries the trailing update needs cross the pair; no
// select on the lane parity is left in the factorisation or in the triangular solves.
//
// This file is a template by inclusion: define MPCX_PAIR_TT (the horizon) and MPCX_PAIR_NS (the
// namespace the instantiation lives in) before each #include; mpcx_kernels.hip includes it
// once for T = 5 and once for T = 10.
#include "mpcx_device.h"

#if !defined(MPCX_PAIR_TT) || !defined(MPCX_PAIR_NS)
#error "define MPCX_PAIR_TT and MPCX_PAIR_NS before including mpcx_pair.h"
#endif

namespace mpcx {
namespace MPCX_PAIR_NS {

#ifndef MPCX_PAIR_AUX
#define MPCX_PAIR_AUX double
#endif
typedef MPCX_PAIR_AUX aux_t;             // storage type of the reciprocal slacks
#ifndef MPCX_PAIR_MU0
#define MPCX_PAIR_MU0 0.05
#endif
#ifndef MPCX_PAIR_GUESS
#define MPCX_PAIR_GUESS 0.15
#endif
// The first active set of the rounds: constraints whose multiplier exceeds kPairGuess x their slack at
// the interior-point iterate.  Erring towards inclusion pays: measured on 65,536 instances of every
// T = 5 variant, closed-loop and random inputs, warm starts take 3-9 % less with 0.15 than with 1 (the
// general bicycle 79.6 -> 76.0 us; a numpy model of the kernel: 2.94 against 3.33 rounds for the
// slowest of 32 instances), cold starts 2 % more, the 2^t-weighted fixed-steer mode 3 % more.
constexpr double kPairGuess = MPCX_PAIR_GUESS;
constexpr double kPairMu0 = MPCX_PAIR_MU0;   // initial complementarity s*lambda at gradients up to 1
constexpr int TT = MPCX_PAIR_TT;         // horizon = number of 2x2 block rows
constexpr int NBLK = TT * (TT + 1) / 2;  // lower-triangular blocks
constexpr int NOFF = TT * (TT - 1) / 2;  // blocks strictly below the diagonal
constexpr int NRT = TT - 1;              // rate constraints per channel and side
constexpr int kLdsPerLane = 2 * NBLK + TT;   // H same, H opp, g own
constexpr double kInvM = 1.0 / (4.0 * TT + 4.0 * NRT);   // 36 constraints per instance
constexpr int kDynPerStage = 11;             // a02, a12, bt[3][2], ct[3]
constexpr int kDynDoubles = kDynPerStage * TT;
constexpr int kPairs = kWave / 2;

__device__ __forceinline__ constexpr int bi(int I, int J) { return I * (I + 1) / 2 + J; }   // J <= I
__device__ __forceinline__ constexpr int oi(int I, int J) { return I * (I - 1) / 2 + J; }   // J <  I

template <int CTRL>
__device__ __forceinline__ double dpp_double(double x)
{
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// quad_perm [1,0,3,2]: value of the partner lane
__device__ __forceinline__ double xchg(double x) { return dpp_double<0xB1>(x); }
// quad_perm [0,0,2,2] / [1,1,3,3]: value of the pair's even / odd lane on both lanes
__device__ __forceinline__ double from_even(double x) { return dpp_double<0xA0>(x); }
__device__ __forceinline__ double from_odd(double x) { return dpp_double<0xF5>(x); }
__device__ __forceinline__ double pair_sum(double x) { return x + xchg(x); }
__device__ __forceinline__ double pair_max(double x) { return dmax(x, xchg(x)); }

typedef __attribute__((address_space(3))) double lds_double;

struct State {
    // K, then its factor.  Blocks below the diagonal: E[I][J] = K(2I + c, 2J), D[I][J] =
    // K(2I + c, 2J + 1).  Diagonal block I: d = the lane's own diagonal entry, b = the entry
    // coupling rows 2I and 2I + 1 (kept identical on both lanes).
    double E[NOFF], D[NOFF], d[TT], b[TT];
    // factor of the diagonal blocks, identical on both lanes:
    // 1 / L(2I, 2I), L(2I + 1, 2I), 1 / L(2I + 1, 2I + 1)
    double r0[TT], l10[TT], r1[TT];
    double lamU[TT], lamL[TT], nuU[NRT], nuL[NRT];
    aux_t isu[TT], isl[TT], iru[NRT], irl[NRT];   // reciprocal slacks / dual steps / active-set flags
    double ws[TT], wo[TT];                    // own and partner controls
    double hw[TT], dwa[TT], dw[TT];           // own rows of H w + g (then predictor), corrector
    lds_double *lds;                          // [kLdsPerLane][64] + lane
    lds_double *ldsE, *ldsD;                  // where this lane finds its even / odd column blocks of H
    lds_double *dyn;                          // [kDynDoubles][32] + pair: stage dynamics of the instance
    bool odd;
    double mE, mO;                            // 1 on the even / odd lane, 0 on the other
    double b_, lo0, hi0, rr;                  // own channel's bounds
    __device__ __forceinline__ void bind(lds_double *base, lds_double *dynbase, int lane)
    {
        odd = (lane & 1) != 0;
        lds = base + lane;
        ldsE = lds + (odd ? NBLK * kWave : 0);
        ldsD = lds + (odd ? 0 : NBLK * kWave);
        dyn = dynbase + (lane >> 1);
        mE = odd ? 0.0 : 1.0;
        mO = odd ? 1.0 : 0.0;
    }
    __device__ __forceinline__ double HS(int I, int J) const { return ((const volatile lds_double *)lds)[bi(I, J) * kWave]; }
    __device__ __forceinline__ double HO(int I, int J) const { return ((const volatile lds_double *)lds)[(NBLK + bi(I, J)) * kWave]; }
    __device__ __forceinline__ double HE_(int I, int J) const { return ((const volatile lds_double *)ldsE)[bi(I, J) * kWave]; }
    __device__ __forceinline__ double HD_(int I, int J) const { return ((const volatile lds_double *)ldsD)[bi(I, J) * kWave]; }
    __device__ __forceinline__ double g(int I) const { return ((const volatile lds_double *)lds)[(2 * NBLK + I) * kWave]; }
    __device__ __forceinline__ double hi(int t) const { return t == 0 ? hi0 : b_; }
    __device__ __forceinline__ double lo(int t) const { return t == 0 ? lo0 : -b_; }
};

// hw <- own rows of H w + g; optionally K <- H on the way.
template <bool COPY_K>
__device__ __forceinline__ void matvec(State &s)
{
    double tS[TT], tO[TT];
#pragma unroll
    for (int I = 0; I < TT; ++I) { tS[I] = 0.0; tO[I] = 0.0; }
#pragma unroll
    for (int I = 0; I < TT; ++I) {
        double acc = s.g(I);
        const double wi = s.ws[I];
#pragma unroll
        for (int J = 0; J < I; ++J) {
            const double hs = s.HS(I, J), ho = s.HO(I, J);
            acc = fma(hs, s.ws[J], acc);
            acc = fma(ho, s.wo[J], acc);
            tS[J] = fma(hs, wi, tS[J]);   // column of my parity: one of my own rows
            tO[J] = fma(ho, wi, tO[J]);   // column of the other parity: a row of the partner
        }
        const double hd = s.HS(I, I), hx = s.HO(I, I);
        if (COPY_K) { s.d[I] = hd; s.b[I] = hx; }
        acc = fma(hd, wi, acc);
        acc = fma(hx, s.wo[I], acc);
        s.hw[I] = acc;
    }
#pragma unroll
    for (int J = 0; J < TT - 1; ++J) s.hw[J] += tS[J] + xchg(tO[J]);
    if (COPY_K) {
#pragma unroll
        for (int I = 1; I < TT; ++I) {
#pragma unroll
            for (int J = 0; J < I; ++J) { s.E[oi(I, J)] = s.HE_(I, J); s.D[oi(I, J)] = s.HD_(I, J); }
        }
    }
}

// 1 / sqrt(x): hardware estimate (relative error 5e-8 measured on gfx950) and one Newton
// step (4e-15): a backward error of the factor at rounding level.
__device__ __forceinline__ double rsqrt_1(double x)
{
    const double r = __builtin_amdgcn_rsq(x);
    const double e = fma(-(x * r), r, 1.0);
    return fma(0.5 * r, e, r);
}

// In-place Cholesky of the pair-distributed matrix, by 2x2 blocks.
__device__ __forceinline__ void chol(State &s)
{
#pragma unroll
    for (int J = 0; J < TT; ++J) {
        const double a = from_even(s.d[J]), c = from_odd(s.d[J]);
        const double r0 = rsqrt_1(a);
        const double l10 = s.b[J] * r0;
        const double r1 = rsqrt_1(fma(-l10, l10, c));
        s.r0[J] = r0; s.l10[J] = l10; s.r1[J] = r1;
        double Le[TT], Ld[TT], cEe[TT], cEd[TT], cOe[TT], cOd[TT];
#pragma unroll
        for (int I = J + 1; I < TT; ++I) {   // panel: local to the row owner
            Le[I] = s.E[oi(I, J)] * r0;
            Ld[I] = fma(-Le[I], l10, s.D[oi(I, J)]) * r1;
            s.E[oi(I, J)] = Le[I];
            s.D[oi(I, J)] = Ld[I];
        }
#pragma unroll
        for (int K = J + 1; K < TT; ++K) {   // panel rows 2K (even lane) and 2K + 1 (odd lane)
            cEe[K] = from_even(Le[K]); cEd[K] = from_even(Ld[K]);
            cOe[K] = from_odd(Le[K]);  cOd[K] = from_odd(Ld[K]);
        }
#pragma unroll
        for (int I = J + 1; I < TT; ++I) {
#pragma unroll
            for (int K = J + 1; K < I; ++K) {
                s.E[oi(I, K)] = fma(-Le[I], cEe[K], fma(-Ld[I], cEd[K], s.E[oi(I, K)]));
                s.D[oi(I, K)] = fma(-Le[I], cOe[K], fma(-Ld[I], cOd[K], s.D[oi(I, K)]));
            }
            s.d[I] = fma(-Le[I], Le[I], fma(-Ld[I], Ld[I], s.d[I]));
            s.b[I] = fma(-cEe[I], cOe[I], fma(-cEd[I], cOd[I], s.b[I]));
        }
    }
}

// r (own rows) <- own rows of K^-1 r, using the factor left by chol().
__device__ __forceinline__ void solve(const State &s, double (&r)[TT])
{
    double yE[TT], yO[TT];
#pragma unroll
    for (int I = 0; I < TT; ++I) {
        double acc = r[I];
#pragma unroll
        for (int J = 0; J < I; ++J) {
            acc = fma(-s.E[oi(I, J)], yE[J], acc);
            acc = fma(-s.D[oi(I, J)], yO[J], acc);
        }
        const double aE = from_even(acc), aO = from_odd(acc);
        yE[I] = aE * s.r0[I];
        yO[I] = fma(-s.l10[I], yE[I], aO) * s.r1[I];
    }
#pragma unroll
    for (int kk = 0; kk < TT; ++kk) {
        const int K = TT - 1 - kk;
        double sE = 0.0, sO = 0.0;
#pragma unroll
        for (int I = K + 1; I < TT; ++I) {
            sE = fma(s.E[oi(I, K)], r[I], sE);
            sO = fma(s.D[oi(I, K)], r[I], sO);
        }
        if (K < TT - 1) { sE = pair_sum(sE); sO = pair_sum(sO); }
        const double zO = (yO[K] - sO) * s.r1[K];
        const double zE = fma(-s.l10[K], zO, yE[K] - sE) * s.r0[K];
        r[K] = s.odd ? zO : zE;
    }
}

// K <- K + (diagonal and same-channel sub-diagonal barrier terms); everything is local.
__device__ __forceinline__ void add_box(State &s, int t, double dd) { s.d[t] += dd; }
__device__ __forceinline__ void add_rate(State &s, int i, double dd)
{
    s.d[i] += dd;
    s.d[i + 1] += dd;
    // K(2(i+1) + c, 2i + c): the even lane's even column, the odd lane's odd column
    s.E[oi(i + 1, i)] = fma(-s.mE, dd, s.E[oi(i + 1, i)]);
    s.D[oi(i + 1, i)] = fma(-s.mO, dd, s.D[oi(i + 1, i)]);
}

// Predictor-corrector interior point on the lane pair (formulas as in ipm_solve() of
// mpcx_device.h, except that the centring parameter comes from the predictor's step lengths,
// sigma = (1 - alpha_aff)^3, which costs no pass over the constraints); every constraint loop
// runs over the lane's own channel and the scalar reductions are completed with the partner.
// `fixed`: a lane runs exactly max_iter iterations (no pass is spent on a convergence test
// after the last one) and reports MPCX_ST_FAILED unless it converged on the way; max_iter
// may differ between lanes, the wave loops until its last lane is through.  A step that is
// not finite (a factorisation that broke down) ends the lane's solve at the iterate it has.
// The interior point's iterate (own controls and multipliers), kept across the active-set rounds,
// which overwrite it, so that a second attempt can go on from it instead of starting over.
struct Iterate {
    // Held in accumulation registers, which this kernel has to spare (its vector registers are
    // full): one v_accvgpr_write per 32 bits when saved, one v_accvgpr_read when taken back.
    static constexpr int N = 3 * TT + 2 * NRT;
    int lo[N], hi[N];
    static __device__ __forceinline__ int to_acc(int x)
    {
        int a;
        asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(x));
        return a;
    }
    static __device__ __forceinline__ int from_acc(int a)
    {
        int x;
        asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x) : "a"(a));
        return x;
    }
    __device__ __forceinline__ void put(int k, double x) { lo[k] = to_acc(__double2loint(x)); hi[k] = to_acc(__double2hiint(x)); }
    __device__ __forceinline__ double get(int k) const { return __hiloint2double(from_acc(hi[k]), from_acc(lo[k])); }
    __device__ __forceinline__ void save(const State &s)
    {
#pragma unroll
        for (int t = 0; t < TT; ++t) { put(t, s.ws[t]); put(TT + t, s.lamU[t]); put(2 * TT + t, s.lamL[t]); }
#pragma unroll
        for (int i = 0; i < NRT; ++i) { put(3 * TT + i, s.nuU[i]); put(3 * TT + NRT + i, s.nuL[i]); }
    }
    __device__ __forceinline__ void restore(State &s, bool which) const
    {
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const double w = get(t), lu = get(TT + t), ll = get(2 * TT + t);
            s.ws[t] = which ? w : s.ws[t];
            s.lamU[t] = which ? lu : s.lamU[t];
            s.lamL[t] = which ? ll : s.lamL[t];
        }
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const double nu = get(3 * TT + i), nl = get(3 * TT + NRT + i);
            s.nuU[i] = which ? nu : s.nuU[i];
            s.nuL[i] = which ? nl : s.nuL[i];
        }
    }
};

// `resume`: go on from the iterate in `s` instead of starting at the centre of the bounds.
__device__ __forceinline__ void ipm(State &s, int max_iter, bool fixed, bool skip, int &status, int &iters, bool resume = false)
{
    // Start: every stage at the middle of the first-stage window (strictly feasible: all
    // rate slacks equal rr).  Starting instead from the previous optimum pulled towards that
    // point lets the active set show after 3 iterations instead of 5 and lowers the mean wave
    // time by 8 %, but it makes a few waves per launch take the converged-solve fallback, and
    // the launch ends with its slowest wave: measured 114 us against 91 us, so it is not done.
    if (!skip && !resume) {   // lanes that sit this solve out keep their controls
        const double wc = 0.5 * (s.lo0 + s.hi0);
        // initial complementarity s*lambda of every constraint, in proportion to the gradient at
        // zero controls where that exceeds 1 (stage weights that grow like 2^t).  Measured on
        // 65,536 instances of every T = 5 variant, closed-loop and random inputs: from 0.1 the
        // five iterations leave one closed-loop diff-drive wavefront in 2,048 unsettled (the launch
        // waits for its second attempt: 94 us instead of 74), from 0.03 the rounds are fewer still
        // (78.6 against 80 us, general bicycle) but random chained inputs lose wavefronts to the
        // second attempt (497 against 325 us); 0.05 has neither.
        double gm = 0.0;
#pragma unroll
        for (int t = 0; t < TT; ++t) gm = dmax(gm, fabs(s.g(t)));
        const double mu0 = kPairMu0 * dmax(1.0, pair_max(gm));
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            s.ws[t] = wc;
            s.lamU[t] = mu0 * rcp_nr(dmax(s.hi(t) - wc, kSlackMin));
            s.lamL[t] = mu0 * rcp_nr(dmax(wc - s.lo(t), kSlackMin));
        }
        const double l0 = mu0 * rcp_nr(s.rr);
#pragma unroll
        for (int i = 0; i < NRT; ++i) { s.nuU[i] = l0; s.nuL[i] = l0; }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) s.wo[t] = xchg(s.ws[t]);
    bool done = skip;
    int my_iters = 0;
    status = MPCX_ST_FAILED;
    for (int it = 0;; ++it) {
        if (fixed) {
            if (!done && it >= max_iter) { done = true; my_iters = it; }
            if (__all(done)) break;
        }
        matvec<true>(s);
        double musum = 0.0, ctl[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const double wi = s.ws[t];
            const double su = dmax(s.hi(t) - wi, kSlackMin), sl = dmax(wi - s.lo(t), kSlackMin);
            const aux_t fu = (aux_t)rcp_raw(su), fl = (aux_t)rcp_raw(sl);
            s.isu[t] = fu; s.isl[t] = fl;
            const double lu = s.lamU[t], ll = s.lamL[t];
            add_box(s, t, fma(lu, (double)fu, ll * (double)fl));
            ctl[t] = lu - ll;
            musum = fma(su, lu, fma(sl, ll, musum));
        }
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const double dv = s.ws[i + 1] - s.ws[i];
            const double ru = dmax(s.rr - dv, kSlackMin), rl = dmax(s.rr + dv, kSlackMin);
            const aux_t fu = (aux_t)rcp_raw(ru), fl = (aux_t)rcp_raw(rl);
            s.iru[i] = fu; s.irl[i] = fl;
            const double nu = s.nuU[i], nl = s.nuL[i];
            add_rate(s, i, fma(nu, (double)fu, nl * (double)fl));
            const double dn = nu - nl;
            ctl[i + 1] += dn;
            ctl[i] -= dn;
            musum = fma(ru, nu, fma(rl, nl, musum));
        }
        const double mu = pair_sum(musum) * kInvM;
        double rdmax = 0.0, scale = 1.0;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const double hwi = s.hw[t], gi = s.g(t), ci = ctl[t];
            rdmax = dmax(rdmax, fabs(hwi + ci));
            scale = dmax(scale, dmax(fabs(hwi - gi), dmax(fabs(gi), fabs(ci))));
        }
        rdmax = pair_max(rdmax);
        scale = pair_max(scale) + 1.0;
        const bool conv = rdmax <= kTolD * scale && mu <= kTolMu;
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
            const double d = s.dwa[t];
            const double xu = d * (double)s.isu[t], xl = -d * (double)s.isl[t];
            rp = dmax(rp, dmax(xu, xl));
            rdm = dmax(rdm, dmax(1.0 - xu, 1.0 - xl));
        }
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const double d = s.dwa[i + 1] - s.dwa[i];
            const double yu = d * (double)s.iru[i], yl = -d * (double)s.irl[i];
            rp = dmax(rp, dmax(yu, yl));
            rdm = dmax(rdm, dmax(1.0 - yu, 1.0 - yl));
        }
        rp = pair_max(rp);
        rdm = pair_max(rdm);
        const double apa = rp > 1.0 ? rcp_nr(rp) : 1.0;
        const double ada = rcp_nr(rdm);
        // centring from the predictor's step lengths alone: sigma = (1 - alpha_aff)^3
        const double back = 1.0 - dmin(apa, ada);
        const double sm = back * back * back * mu;   // sigma * mu

        // corrector as a correction to the predictor: K (dw - dwa) = -C' q
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const double d = s.dwa[t];
            const double iu = (double)s.isu[t], il = (double)s.isl[t];
            const double xu = d * iu, xl = -d * il;
            const double qu = fma(sm, iu, s.lamU[t] * xu * (xu - 1.0));
            const double ql = fma(sm, il, s.lamL[t] * xl * (xl - 1.0));
            s.dw[t] = ql - qu;
        }
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const double d = s.dwa[i + 1] - s.dwa[i];
            const double iu = (double)s.iru[i], il = (double)s.irl[i];
            const double yu = d * iu, yl = -d * il;
            const double qu = fma(sm, iu, s.nuU[i] * yu * (yu - 1.0));
            const double ql = fma(sm, il, s.nuL[i] * yl * (yl - 1.0));
            const double q = qu - ql;
            s.dw[i + 1] -= q;
            s.dw[i] += q;
        }
        solve(s, s.dw);
#pragma unroll
        for (int t = 0; t < TT; ++t) s.dw[t] += s.dwa[t];

        // dual steps (stored over the reciprocal slacks) and the two step lengths
        double rp2 = 0.0, rd2 = 0.0;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const double da = s.dwa[t], d = s.dw[t];
            const double iu = (double)s.isu[t], il = (double)s.isl[t], lu = s.lamU[t], ll = s.lamL[t];
            const double xu = da * iu, xl = -da * il;
            const double qu = fma(sm, iu, lu * xu * (xu - 1.0));
            const double ql = fma(sm, il, ll * xl * (xl - 1.0));
            const double zu = d * iu, zl = -d * il;
            const double dlu = fma(lu, zu, qu - lu), dll = fma(ll, zl, ql - ll);
            rp2 = dmax(rp2, dmax(zu, zl));
            rd2 = dmax(rd2, dmax(-dlu * rcp_raw(lu), -dll * rcp_raw(ll)));
            s.isu[t] = (aux_t)dlu; s.isl[t] = (aux_t)dll;
        }
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const double da = s.dwa[i + 1] - s.dwa[i], d = s.dw[i + 1] - s.dw[i];
            const double iu = (double)s.iru[i], il = (double)s.irl[i], nu = s.nuU[i], nl = s.nuL[i];
            const double yu = da * iu, yl = -da * il;
            const double qu = fma(sm, iu, nu * yu * (yu - 1.0));
            const double ql = fma(sm, il, nl * yl * (yl - 1.0));
            const double zu = d * iu, zl = -d * il;
            const double dnu = fma(nu, zu, qu - nu), dnl = fma(nl, zl, ql - nl);
            rp2 = dmax(rp2, dmax(zu, zl));
            rd2 = dmax(rd2, dmax(-dnu * rcp_raw(nu), -dnl * rcp_raw(nl)));
            s.iru[i] = (aux_t)dnu; s.irl[i] = (aux_t)dnl;
        }
        rp2 = pair_max(rp2);
        rd2 = pair_max(rd2);
        const double eta = dmin(kEtaCap, dmax(0.99, 1.0 - mu));
        const double ap = rp2 > eta ? eta * rcp_nr(rp2) : 1.0;
        const double ad = rd2 > eta ? eta * rcp_nr(rd2) : 1.0;
        // A factorisation that broke down (K lost definiteness to rounding near a degenerate
        // vertex) leaves NaN in the step; max() would hide it, a sum does not.  The lane then
        // stops at the iterate it has, which the active-set rounds can still finish.
        double chk = 0.0;
#pragma unroll
        for (int t = 0; t < TT; ++t) chk += s.dw[t] + (double)s.isu[t] + (double)s.isl[t];
#pragma unroll
        for (int i = 0; i < NRT; ++i) chk += (double)s.iru[i] + (double)s.irl[i];
        if (!done && !finite_d(pair_sum(chk))) {
            done = true;
            my_iters = it;
            status = (rdmax <= kLooseD * scale && mu <= kLooseMu) ? MPCX_ST_INACCURATE : MPCX_ST_FAILED;
        }
        if (!done) {
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                s.ws[t] = fma(ap, s.dw[t], s.ws[t]);
                s.lamU[t] = fma(ad, (double)s.isu[t], s.lamU[t]);
                s.lamL[t] = fma(ad, (double)s.isl[t], s.lamL[t]);
            }
#pragma unroll
            for (int i = 0; i < NRT; ++i) {
                s.nuU[i] = fma(ad, (double)s.iru[i], s.nuU[i]);
                s.nuL[i] = fma(ad, (double)s.irl[i], s.nuL[i]);
            }
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) s.wo[t] = xchg(s.ws[t]);
    }
    iters = my_iters;
}

// Primal-dual active-set rounds on the lane pair, started from an interior-point iterate
// (constraints with multiplier > slack are taken as active).  Each round is one Newton step
// of the augmented Lagrangian of the equality-constrained QP on the current set (factor of
// H + rho A'A), a first-order multiplier update, and a correction of the set: multipliers
// that turned negative leave, violated constraints enter.  A lane settles when its set
// stood still over the last round and the point passes the KKT test of the original QP
// (stationarity 1e-10 relative to the larger of the gradient's terms and |H| |w|, feasibility
// and active-row residual 1e-12, multipliers >= -1e-9 relative); rounds on an unchanged set are refinement steps, so a settled point
// is exact to rounding.  Lanes in `skip` are left alone.  Returns true where the lane
// settled; a lane that did not gets its controls back as they were on entry.
__device__ __forceinline__ bool pdas(State &s, bool skip, int max_rounds, int *rounds_done = nullptr)
{
    double hmax = 0.0, wbak[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) { hmax = dmax(hmax, s.HS(t, t)); wbak[t] = s.ws[t]; }
    const double rho = 1e5 * pair_max(hmax);
    // initial set from the interior-point iterate
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const double wi = s.ws[t];
        const bool au = s.lamU[t] > kPairGuess * (s.hi(t) - wi), al = s.lamL[t] > kPairGuess * (wi - s.lo(t));
        s.isu[t] = au ? 1.0 : 0.0; s.isl[t] = al ? 1.0 : 0.0;
        s.lamU[t] = au ? s.lamU[t] : 0.0; s.lamL[t] = al ? s.lamL[t] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
        const double dv = s.ws[i + 1] - s.ws[i];
        const bool au = s.nuU[i] > kPairGuess * (s.rr - dv), al = s.nuL[i] > kPairGuess * (s.rr + dv);
        s.iru[i] = au ? 1.0 : 0.0; s.irl[i] = al ? 1.0 : 0.0;
        s.nuU[i] = au ? s.nuU[i] : 0.0; s.nuL[i] = al ? s.nuL[i] : 0.0;
    }
    bool settled = skip, accepted = false, changed = true;
    double scale = 1.0;
#pragma unroll 1
    for (int round = 0;; ++round) {
        matvec<true>(s);
        double viol = 0.0, cact = 0.0, neg = 0.0, stat = 0.0;
        double gl[TT], ga[TT];
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const double wi = s.ws[t], vu = wi - s.hi(t), vl = s.lo(t) - wi;
            const double au = (double)s.isu[t], al = (double)s.isl[t];
            const double cu = au * vu, cl = al * vl;
            const double t1 = s.lamU[t] - s.lamL[t];
            gl[t] = s.hw[t] + t1;
            ga[t] = rho * (cu - cl);
            viol = dmax(viol, dmax(vu, vl));
            cact = dmax(cact, dmax(fabs(cu), fabs(cl)));
            neg = dmax(neg, dmax(-s.lamU[t], -s.lamL[t]));
            add_box(s, t, rho * (au + al));
        }
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const double dv = s.ws[i + 1] - s.ws[i], vu = dv - s.rr, vl = -dv - s.rr;
            const double au = (double)s.iru[i], al = (double)s.irl[i];
            const double cu = au * vu, cl = al * vl;
            const double t1 = s.nuU[i] - s.nuL[i];
            gl[i + 1] += t1; gl[i] -= t1;
            const double ta = rho * (cu - cl);
            ga[i + 1] += ta; ga[i] -= ta;
            viol = dmax(viol, dmax(vu, vl));
            cact = dmax(cact, dmax(fabs(cu), fabs(cl)));
            neg = dmax(neg, dmax(-s.nuU[i], -s.nuL[i]));
            add_rate(s, i, rho * (au + al));
        }
        scale = 1.0;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            stat = dmax(stat, fabs(gl[t]));
            const double gi = s.g(t);
            scale = dmax(scale, dmax(fabs(gi), fabs(s.hw[t] - gi)));
        }
        stat = pair_max(stat); viol = pair_max(viol); cact = pair_max(cact); neg = pair_max(neg);
        scale = pair_max(scale) + 1.0;
        // max() drops NaN operands, so the tests alone would wave a non-finite point through
        double chk = 0.0;
#pragma unroll
        for (int t = 0; t < TT; ++t) chk += gl[t] + ga[t] + s.ws[t];
        const bool fin = finite_d(pair_sum(chk));
        // Stationarity is asked relative to the gradient's terms or, where that is larger, to
        // |H| |w|: the multiplier update lam += rho c(w) carries the rounding of the controls (an
        // ulp, 1.1e-16 |w|) times rho = 1e5 max(diag H) into the multipliers, a floor of some
        // 5e-11 max(diag H) |w| that no further round gets below; with a large Hessian and a small
        // gradient the first scale alone would sit under that floor for ever.
        // (|w| <= the box bound of the channel: the first stage's window lies inside it.)
        const double stat_tol = 1e-10 * dmax(scale, 4e-5 * rho * pair_max(s.b_));
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
                const double wi = s.ws[t], vu = wi - s.hi(t), vl = s.lo(t) - wi;
                const bool au = s.isu[t] != 0, al = s.isl[t] != 0;
                const double lu = au ? fma(rho, vu, s.lamU[t]) : 0.0, ll = al ? fma(rho, vl, s.lamL[t]) : 0.0;
                const bool nu = (au && lu >= ntol) || vu > 1e-12, nl = (al && ll >= ntol) || vl > 1e-12;
                ch = ch || nu != au || nl != al;
                s.isu[t] = nu ? 1.0 : 0.0; s.isl[t] = nl ? 1.0 : 0.0;
                s.lamU[t] = nu ? dmax(lu, 0.0) : 0.0; s.lamL[t] = nl ? dmax(ll, 0.0) : 0.0;
            }
#pragma unroll
            for (int i = 0; i < NRT; ++i) {
                const double dv = s.ws[i + 1] - s.ws[i], vu = dv - s.rr, vl = -dv - s.rr;
                const bool au = s.iru[i] != 0, al = s.irl[i] != 0;
                const double lu = au ? fma(rho, vu, s.nuU[i]) : 0.0, ll = al ? fma(rho, vl, s.nuL[i]) : 0.0;
                const bool nu = (au && lu >= ntol) || vu > 1e-12, nl = (al && ll >= ntol) || vl > 1e-12;
                ch = ch || nu != au || nl != al;
                s.iru[i] = nu ? 1.0 : 0.0; s.irl[i] = nl ? 1.0 : 0.0;
                s.nuU[i] = nu ? dmax(lu, 0.0) : 0.0; s.nuL[i] = nl ? dmax(ll, 0.0) : 0.0;
            }
        }
        // the partner's channel belongs to the same QP: the set moved if either half did
        changed = __builtin_amdgcn_mov_dpp((int)ch, 0xB1, 0xF, 0xF, true) != 0 || ch;
#pragma unroll
        for (int t = 0; t < TT; ++t) s.wo[t] = xchg(s.ws[t]);
    }
    if (!accepted && !skip) {
#pragma unroll
        for (int t = 0; t < TT; ++t) s.ws[t] = wbak[t];
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) s.wo[t] = xchg(s.ws[t]);
    return accepted;
}

// Contiguous block of n doubles between global memory and LDS, 512 bytes per wave
// instruction (the wave's 32 instance records lie back to back in every API array).
// load(): every lane fetches its doubles of the block (indices past the end are clamped, so
// all lanes load from valid addresses and nothing waits on a branch); put(): into LDS.  The
// caller issues the loads of all its blocks before the first put, so the wave pays one memory
// round trip.  A value that is not finite raises the flag of the instance record it belongs
// to (PER doubles each).
// Records are laid out in LDS with an odd stride (in doubles), so that the 32 pairs of a
// wave reading the same field of their records hit 32 different bank pairs.
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
    __device__ __forceinline__ void put(int n, lds_double *dst, int lane,
                                        volatile __attribute__((address_space(3))) int *flag) const
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
__device__ __forceinline__ void stage_out(double *__restrict__ g, int n0, int n, const lds_double *src, int lane)
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
        if (i >= n0 && i < n) g[i] = tmp[k];
    }
}

// Linearisation point of one instance, read from the wave's LDS staging area.
struct Lin {
    const volatile lds_double *u;     // [NU][T] (always a readable address)
    const volatile lds_double *yaw;   // xbar[2][0..T-1] when given
    const double *cold;               // v.cold_start
    bool warm, has_yaw;
    double theta0;
    __device__ __forceinline__ double control(int r, int t) const
    {
        const double x = u[r * TT + t];
        return warm ? x : cold[r];
    }
};

// Reduced dynamics of one stage:  x+ = (I + a02 e0 e2' + a12 e1 e2') x + bt w_t + ct.
struct Dyn {
    double a02, a12, bt[3][2], ct[3];
};

__device__ __forceinline__ void load_dyn(const State &s, int t, Dyn &d)
{
    const volatile lds_double *p = (const volatile lds_double *)s.dyn + t * kDynPerStage * kPairs;
    d.a02 = p[0];
    d.a12 = p[kPairs];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        d.bt[r][0] = p[(2 + 2 * r) * kPairs];
        d.bt[r][1] = p[(3 + 2 * r) * kPairs];
        d.ct[r] = p[(8 + r) * kPairs];
    }
}

// Linearisation point of the lane's own stages (even lane: stages 0, 2, 4, ...; odd lane: 1, 3,
// ... and, for an odd horizon, the last stage again), kept in registers while it is needed.
constexpr int NSLOT = (TT + 1) / 2;
struct LinRegs {
    double u[NSLOT][4];    // controls at the linearisation point
    double th[NSLOT];      // yaw at the linearisation point
    SteerTerms k[NSLOT];   // sines / cosines of the steer angles of u
    int tj[NSLOT];         // which stage each slot is
};

// Controls and yaw of the lane's stages from the staged inputs.  The yaw is rolled from the
// yaw rates exactly as predict_motion does (mpc_diff_drive.py:126-139), the rates of the
// partner's stages arriving by DPP.  With `chained` (MPC.iterativeLMPC, mpc_sim.py:292-299)
// the yaw still follows the incoming controls, but the controls of the linearisation point
// start as column 0 of them with the rest zero.
template <int MODEL>
__device__ __forceinline__ void prepare_lin(const State &s, const mpcx_variant &v, const Lin &lp, bool chained,
                                            LinRegs &L)
{
    constexpr int NU = MODEL == MPCX_MODEL_DIFF_DRIVE ? 2 : 4;
    const bool odd = s.odd;
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        int t = 2 * j + (odd ? 1 : 0);
        t = t < TT ? t : TT - 1;
        L.tj[j] = t;
#pragma unroll
        for (int r = 0; r < 4; ++r) L.u[j][r] = 0.0;
#pragma unroll
        for (int r = 0; r < NU; ++r) L.u[j][r] = lp.control(r, t);
        steer_terms<MODEL>(v, L.u[j], L.k[j]);
    }
    if (lp.has_yaw) {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) L.th[j] = lp.yaw[L.tj[j]];
    } else {
        double th = lp.theta0;   // yaw at stage 2j (even lane's slot j), then at stage 2j + 1
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            const double we = from_even(L.k[j].omega), wo = from_odd(L.k[j].omega);
            const double th_even = th;
            const double th_odd = th_even + we * v.dt;
            // for an odd horizon the odd lane's last slot repeats the last (even) stage
            const bool repeat = 2 * j + 1 >= TT;
            L.th[j] = (odd && !repeat) ? th_odd : th_even;
            th = th_odd + wo * v.dt;
        }
    }
    if (chained) {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) {
            if (L.tj[j] > 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) L.u[j][r] = 0.0;
            }
        }
    }
}

// Stage dynamics of the lane's stages into LDS (both lanes then read all five).  `fresh`:
// the controls changed since the steer terms were last evaluated.
template <int MODEL>
__device__ __forceinline__ void stage_eval(State &s, const mpcx_variant &v, LinRegs &L, bool fresh)
{
    if (fresh) {
#pragma unroll
        for (int j = 0; j < NSLOT; ++j) steer_terms<MODEL>(v, L.u[j], L.k[j]);
    }
#pragma unroll
    for (int j = 0; j < NSLOT; ++j) {
        double a02, a12, Bm[3][4], C[3], bt[3][2], ct[3];
        linearize_at<MODEL>(v, L.th[j], L.u[j], L.k[j], a02, a12, Bm, C);
        reduce_controls(v, Bm, C, bt, ct);
        volatile lds_double *p = (volatile lds_double *)s.dyn + L.tj[j] * kDynPerStage * kPairs;
        p[0] = a02;
        p[kPairs] = a12;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            p[(2 + 2 * r) * kPairs] = bt[r][0];
            p[(3 + 2 * r) * kPairs] = bt[r][1];
            p[(8 + r) * kPairs] = ct[r];
        }
    }
    __builtin_amdgcn_wave_barrier();   // one wave: LDS keeps program order, the compiler must too
}

// Condensed Hessian rows and gradient entries of the lane's channel, straight into LDS.
// The stage dynamics come from LDS (stage_eval); each lane accumulates its own rows of H.
__device__ __forceinline__ void condense(State &s, const mpcx_variant &v, const double (&x0)[3],
                                         const volatile lds_double *xref)
{
    constexpr int W = TT + 1;
    const bool odd = s.odd;
    double rw[2], rdw[2];
    if (v.eq_mode == MPCX_EQ_NONE) {
        rw[0] = v.R[0]; rw[1] = v.R[1]; rdw[0] = v.Rd[0]; rdw[1] = v.Rd[1];
    } else if (v.eq_mode == MPCX_EQ_FIXED_STEER) {
        rw[0] = v.R[0]; rw[1] = v.R[2]; rdw[0] = v.Rd[0]; rdw[1] = v.Rd[2];
    } else {
        rw[0] = v.R[0] + v.R[2]; rw[1] = v.R[1] + v.R[3];
        rdw[0] = v.Rd[0] + v.Rd[2]; rdw[1] = v.Rd[1] + v.Rd[3];
    }
    const double rwc = odd ? rw[1] : rw[0], rdc = odd ? rdw[1] : rdw[0];
    double GS[3][TT], GO[3][TT], gacc[TT], HSa[NBLK], HOa[NBLK];
#pragma unroll
    for (int I = 0; I < TT; ++I) {
#pragma unroll
        for (int J = 0; J <= I; ++J) { HSa[bi(I, J)] = 0.0; HOa[bi(I, J)] = 0.0; }
        gacc[I] = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) { GS[r][I] = 0.0; GO[r][I] = 0.0; }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        double d = 2.0 * rwc;
        if (t > 0) d += 2.0 * rdc;
        if (t + 1 < TT) d += 2.0 * rdc;
        HSa[bi(t, t)] = d;
        if (t > 0) HSa[bi(t, t - 1)] = -2.0 * rdc;
    }
    double p[3] = {x0[0], x0[1], x0[2]};
    double qg = 1.0;
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        Dyn dy;
        load_dyn(s, t, dy);
        const double a02 = dy.a02, a12 = dy.a12;
        const double (&bt)[3][2] = dy.bt;
        const double (&ct)[3] = dy.ct;
        const double pn0 = p[0] + a02 * p[2] + ct[0];
        const double pn1 = p[1] + a12 * p[2] + ct[1];
        const double pn2 = p[2] + ct[2];
        p[0] = pn0; p[1] = pn1; p[2] = pn2;
#pragma unroll
        for (int J = 0; J < t; ++J) {
            GS[0][J] = fma(a02, GS[2][J], GS[0][J]);
            GS[1][J] = fma(a12, GS[2][J], GS[1][J]);
            GO[0][J] = fma(a02, GO[2][J], GO[0][J]);
            GO[1][J] = fma(a12, GO[2][J], GO[1][J]);
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            GS[r][t] = odd ? bt[r][1] : bt[r][0];
            GO[r][t] = odd ? bt[r][0] : bt[r][1];
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
                const double gi = q2 * GS[r][I];
                gacc[I] = fma(gi, e, gacc[I]);
#pragma unroll
                for (int J = 0; J <= I; ++J) {
                    HSa[bi(I, J)] = fma(gi, GS[r][J], HSa[bi(I, J)]);
                    HOa[bi(I, J)] = fma(gi, GO[r][J], HOa[bi(I, J)]);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < NBLK; ++e) { s.lds[e * kWave] = HSa[e]; s.lds[(NBLK + e) * kWave] = HOa[e]; }
#pragma unroll
    for (int I = 0; I < TT; ++I) s.lds[(2 * NBLK + I) * kWave] = gacc[I];
}

}  // namespace MPCX_PAIR_NS
}  // namespace mpcx
