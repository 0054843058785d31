// Two lanes per MPC instance (T = 5, every bound present).
//
// The condensed QP has 2 free controls per stage; variable 2t + c belongs to channel c.
// Lane parity c owns channel c: its five controls, its box and rate constraints (all of
// them couple controls of one channel only), and rows 2t + c of the symmetric matrices.
// That halves the per-lane state (~150 registers instead of > 500), so two waves fit on
// a SIMD and nothing spills.  Matrices are stored by 2x2 blocks (I, J), J <= I, two
// entries per block and lane, in parity-relative order:
//     S[I][J] = M(2I + c, 2J + c)        "same-parity column"
//     O[I][J] = M(2I + c, 2J + 1 - c)    "opposite-parity column"
// With this order the trailing update of the Cholesky factorisation, the triangular solves
// and the matrix-vector product run the same instruction stream on both lanes; partner
// values travel by DPP quad permutes (no LDS, no divergence).  For the even lane O[I][I]
// is the mirror of the odd lane's sub-diagonal entry: it is kept consistent but never read
// as part of the factor.
#pragma once
#include "mpcx_device.h"

namespace mpcx {
namespace pair {

#ifndef MPCX_PAIR_AUX
#define MPCX_PAIR_AUX double
#endif
typedef MPCX_PAIR_AUX aux_t;             // storage type of the reciprocal slacks
constexpr int TT = 5;                    // horizon = number of 2x2 block rows
constexpr int NBLK = TT * (TT + 1) / 2;  // lower-triangular blocks
constexpr int NRT = TT - 1;              // rate constraints per channel and side
constexpr int kLdsPerLane = 2 * NBLK + TT;   // H same, H opp, g own
constexpr double kInvM = 1.0 / (4.0 * TT + 4.0 * NRT);   // 36 constraints per instance
constexpr int kDynPerStage = 11;             // a02, a12, bt[3][2], ct[3]
constexpr int kDynDoubles = kDynPerStage * TT;
constexpr int kPairs = kWave / 2;

__device__ __forceinline__ constexpr int bi(int I, int J) { return I * (I + 1) / 2 + J; }

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
    double S[NBLK], O[NBLK];                  // K, then its Cholesky factor (diagonal = 1/L_ii)
    double lamU[TT], lamL[TT], nuU[NRT], nuL[NRT];
    aux_t isu[TT], isl[TT], iru[NRT], irl[NRT];   // reciprocal slacks / dual steps / polish flags
    double ws[TT], wo[TT];                    // own and partner controls
    double hw[TT], dwa[TT], dw[TT];           // own rows of H w + g (then predictor), corrector
    lds_double *lds;                          // [kLdsPerLane][64] + lane
    lds_double *dyn;                          // [kDynDoubles][32] + pair: stage dynamics of the instance
    bool odd;
    double b, lo0, hi0, rr;                   // own channel's bounds
    __device__ __forceinline__ double HS(int I, int J) const { return ((const volatile lds_double *)lds)[bi(I, J) * kWave]; }
    __device__ __forceinline__ double HO(int I, int J) const { return ((const volatile lds_double *)lds)[(NBLK + bi(I, J)) * kWave]; }
    __device__ __forceinline__ double g(int I) const { return ((const volatile lds_double *)lds)[(2 * NBLK + I) * kWave]; }
    __device__ __forceinline__ double hi(int t) const { return t == 0 ? hi0 : b; }
    __device__ __forceinline__ double lo(int t) const { return t == 0 ? lo0 : -b; }
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
            if (COPY_K) { s.S[bi(I, J)] = hs; s.O[bi(I, J)] = ho; }
            acc = fma(hs, s.ws[J], acc);
            acc = fma(ho, s.wo[J], acc);
            tS[J] = fma(hs, wi, tS[J]);   // column of my parity: one of my own rows
            tO[J] = fma(ho, wi, tO[J]);   // column of the other parity: a row of the partner
        }
        const double hd = s.HS(I, I), hx = s.HO(I, I);
        if (COPY_K) { s.S[bi(I, I)] = hd; s.O[bi(I, I)] = hx; }
        acc = fma(hd, wi, acc);
        acc = fma(hx, s.wo[I], acc);
        s.hw[I] = acc;
    }
#pragma unroll
    for (int J = 0; J < TT - 1; ++J) s.hw[J] += tS[J] + xchg(tO[J]);
}

// In-place Cholesky of the pair-distributed matrix; S[I][I] ends up as 1 / L_ii of the
// lane's own diagonal entry.
__device__ __forceinline__ void chol(State &s)
{
    const bool odd = s.odd;
#pragma unroll
    for (int J = 0; J < TT; ++J) {
        {   // pivot column 2J (the even lane's diagonal)
            const double r = rsqrt_nr(from_even(s.S[bi(J, J)]));
            const double rs = odd ? 1.0 : r, ro = odd ? r : 1.0;   // which of my slots is that column
            double own[TT], oth[TT];
#pragma unroll
            for (int I = J + 1; I < TT; ++I) {
                s.S[bi(I, J)] *= rs;
                s.O[bi(I, J)] *= ro;
                own[I] = odd ? s.O[bi(I, J)] : s.S[bi(I, J)];
            }
            const double lJ = s.O[bi(J, J)] * r;   // odd lane: L(2J+1, 2J)
            s.O[bi(J, J)] = lJ;
            s.S[bi(J, J)] = odd ? s.S[bi(J, J)] : r;
            own[J] = odd ? lJ : 0.0;               // the pivot row itself takes no update
#pragma unroll
            for (int I = J; I < TT; ++I) oth[I] = xchg(own[I]);
#pragma unroll
            for (int I = J; I < TT; ++I) {
                const double li = own[I];
#pragma unroll
                for (int K = J; K <= I; ++K) {
                    s.S[bi(I, K)] = fma(-li, own[K], s.S[bi(I, K)]);
                    s.O[bi(I, K)] = fma(-li, oth[K], s.O[bi(I, K)]);
                }
            }
        }
        {   // pivot column 2J + 1 (the odd lane's diagonal)
            const double r = rsqrt_nr(from_odd(s.S[bi(J, J)]));
            const double rs = odd ? r : 1.0, ro = odd ? 1.0 : r;
            double own[TT], oth[TT];
#pragma unroll
            for (int I = J + 1; I < TT; ++I) {
                s.S[bi(I, J)] *= rs;
                s.O[bi(I, J)] *= ro;
                own[I] = odd ? s.S[bi(I, J)] : s.O[bi(I, J)];
            }
            s.S[bi(J, J)] = odd ? r : s.S[bi(J, J)];
#pragma unroll
            for (int I = J + 1; I < TT; ++I) oth[I] = xchg(own[I]);
#pragma unroll
            for (int I = J + 1; I < TT; ++I) {
                const double li = own[I];
#pragma unroll
                for (int K = J + 1; K <= I; ++K) {
                    s.S[bi(I, K)] = fma(-li, own[K], s.S[bi(I, K)]);
                    s.O[bi(I, K)] = fma(-li, oth[K], s.O[bi(I, K)]);
                }
            }
        }
    }
}

// r (own rows) <- own rows of K^-1 r, using the factor left by chol().
__device__ __forceinline__ void solve(const State &s, double (&r)[TT])
{
    const bool odd = s.odd;
    double ys[TT], yo[TT];
#pragma unroll
    for (int I = 0; I < TT; ++I) {
        double acc = r[I];
#pragma unroll
        for (int J = 0; J < I; ++J) {
            acc = fma(-s.S[bi(I, J)], ys[J], acc);
            acc = fma(-s.O[bi(I, J)], yo[J], acc);
        }
        const double d = s.S[bi(I, I)];
        const double yE = from_even(acc * d);                       // row 2I
        const double yO = from_odd(fma(-s.O[bi(I, I)], yE, acc) * d);   // row 2I + 1
        ys[I] = odd ? yO : yE;
        yo[I] = odd ? yE : yO;
    }
    double zs[TT];
#pragma unroll
    for (int kk = 0; kk < TT; ++kk) {
        const int K = TT - 1 - kk;
        double sumS = 0.0, sumO = 0.0;
#pragma unroll
        for (int I = K + 1; I < TT; ++I) {
            sumS = fma(s.S[bi(I, K)], zs[I], sumS);
            sumO = fma(s.O[bi(I, K)], zs[I], sumO);
        }
        const double d = s.S[bi(K, K)];
        // column 2K + 1: the odd lane's same-parity entries, the even lane's opposite ones
        const double zO = from_odd((ys[K] - pair_sum(odd ? sumS : sumO)) * d);
        // column 2K: plus the odd lane's sub-diagonal entry of this block
        const double zE = from_even((ys[K] - pair_sum(odd ? fma(s.O[bi(K, K)], zO, sumO) : sumS)) * d);
        zs[K] = odd ? zO : zE;
    }
#pragma unroll
    for (int I = 0; I < TT; ++I) r[I] = zs[I];
}

// K <- K + (diagonal and same-channel sub-diagonal barrier terms); everything is local.
__device__ __forceinline__ void add_box(State &s, int t, double d) { s.S[bi(t, t)] += d; }
__device__ __forceinline__ void add_rate(State &s, int i, double d)
{
    s.S[bi(i, i)] += d;
    s.S[bi(i + 1, i + 1)] += d;
    s.S[bi(i + 1, i)] -= d;
}

// Mehrotra predictor-corrector on the lane pair.  See ipm_solve() in mpcx_device.h for the
// formulas; here every constraint loop runs over the lane's own channel and the scalar
// reductions are completed with the partner.
__device__ __forceinline__ void ipm(State &s, int max_iter, bool skip, int &status, int &iters)
{
    if (!skip) {   // lanes that sit this solve out keep their controls
        const double wc = 0.5 * (s.lo0 + s.hi0);
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            s.ws[t] = wc;
            s.lamU[t] = kMu0 * rcp_nr(dmax(s.hi(t) - wc, kSlackMin));
            s.lamL[t] = kMu0 * rcp_nr(dmax(wc - s.lo(t), kSlackMin));
        }
        const double l0 = kMu0 * rcp_nr(s.rr);
#pragma unroll
        for (int i = 0; i < NRT; ++i) { s.nuU[i] = l0; s.nuL[i] = l0; }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) s.wo[t] = xchg(s.ws[t]);
    bool done = skip;
    int my_iters = 0;
    status = MPCX_ST_FAILED;
    for (int it = 0;; ++it) {
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
        double muaff = 0.0;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const double wi = s.ws[t], d = s.dwa[t];
            const double su = dmax(s.hi(t) - wi, kSlackMin), sl = dmax(wi - s.lo(t), kSlackMin);
            const double xu = d * (double)s.isu[t], xl = -d * (double)s.isl[t];
            muaff += (su * fma(-apa, xu, 1.0)) * (s.lamU[t] * fma(ada, xu - 1.0, 1.0));
            muaff += (sl * fma(-apa, xl, 1.0)) * (s.lamL[t] * fma(ada, xl - 1.0, 1.0));
        }
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const double dv = s.ws[i + 1] - s.ws[i], d = s.dwa[i + 1] - s.dwa[i];
            const double ru = dmax(s.rr - dv, kSlackMin), rl = dmax(s.rr + dv, kSlackMin);
            const double yu = d * (double)s.iru[i], yl = -d * (double)s.irl[i];
            muaff += (ru * fma(-apa, yu, 1.0)) * (s.nuU[i] * fma(ada, yu - 1.0, 1.0));
            muaff += (rl * fma(-apa, yl, 1.0)) * (s.nuL[i] * fma(ada, yl - 1.0, 1.0));
        }
        muaff = pair_sum(muaff) * kInvM;
        const double ratio = muaff * rcp_nr(dmax(mu, 1e-300));
        const double sm = ratio * ratio * ratio * mu;   // sigma * mu

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

// Active-set polish on the lane pair (see polish() in mpcx_device.h).
__device__ __forceinline__ bool polish(State &s, bool skip)
{
    double hmax = 0.0, wbak[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) { hmax = dmax(hmax, s.HS(t, t)); wbak[t] = s.ws[t]; }
    const double rho = 1e5 * pair_max(hmax);
    bool accepted = false, settled = skip;
    double scale = 1.0;
#pragma unroll 1
    for (int round = 0; round < kPolishRounds; ++round) {
        const double ntol = -1e-9 * scale;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const double wi = s.ws[t], su = s.hi(t) - wi, sl = wi - s.lo(t);
            aux_t au, al;
            if (round == 0) {
                au = s.lamU[t] > su ? 1.f : 0.f;
                al = s.lamL[t] > sl ? 1.f : 0.f;
            } else {
                au = ((s.isu[t] != 0.f && s.lamU[t] >= ntol) || su < -1e-12) ? 1.f : 0.f;
                al = ((s.isl[t] != 0.f && s.lamL[t] >= ntol) || sl < -1e-12) ? 1.f : 0.f;
            }
            s.isu[t] = au; s.isl[t] = al;
            s.lamU[t] = (double)au * dmax(s.lamU[t], 0.0);
            s.lamL[t] = (double)al * dmax(s.lamL[t], 0.0);
        }
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const double dv = s.ws[i + 1] - s.ws[i];
            aux_t au, al;
            if (round == 0) {
                au = s.nuU[i] > s.rr - dv ? 1.f : 0.f;
                al = s.nuL[i] > s.rr + dv ? 1.f : 0.f;
            } else {
                au = ((s.iru[i] != 0.f && s.nuU[i] >= ntol) || s.rr - dv < -1e-12) ? 1.f : 0.f;
                al = ((s.irl[i] != 0.f && s.nuL[i] >= ntol) || s.rr + dv < -1e-12) ? 1.f : 0.f;
            }
            s.iru[i] = au; s.irl[i] = al;
            s.nuU[i] = (double)au * dmax(s.nuU[i], 0.0);
            s.nuL[i] = (double)al * dmax(s.nuL[i], 0.0);
        }
        // K = H + rho A'A on the active rows
#pragma unroll
        for (int I = 0; I < TT; ++I) {
#pragma unroll
            for (int J = 0; J <= I; ++J) { s.S[bi(I, J)] = s.HS(I, J); s.O[bi(I, J)] = s.HO(I, J); }
        }
#pragma unroll
        for (int t = 0; t < TT; ++t) add_box(s, t, rho * (double)(s.isu[t] + s.isl[t]));
#pragma unroll
        for (int i = 0; i < NRT; ++i) add_rate(s, i, rho * (double)(s.iru[i] + s.irl[i]));
        chol(s);

        double stat = 0.0;
#pragma unroll 1
        for (int pass = 0; pass < 3; ++pass) {
            matvec<false>(s);
            double gl[TT];   // gradient of the Lagrangian; hw becomes that of the augmented one
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                const double wi = s.ws[t];
                const double cu = (double)s.isu[t] * (wi - s.hi(t)), cl = (double)s.isl[t] * (s.lo(t) - wi);
                if (pass > 0) { s.lamU[t] = fma(rho, cu, s.lamU[t]); s.lamL[t] = fma(rho, cl, s.lamL[t]); }
                const double t1 = s.lamU[t] - s.lamL[t];
                gl[t] = s.hw[t] + t1;
                s.hw[t] += t1 + rho * (cu - cl);
            }
#pragma unroll
            for (int i = 0; i < NRT; ++i) {
                const double dv = s.ws[i + 1] - s.ws[i];
                const double cu = (double)s.iru[i] * (dv - s.rr), cl = (double)s.irl[i] * (-dv - s.rr);
                if (pass > 0) { s.nuU[i] = fma(rho, cu, s.nuU[i]); s.nuL[i] = fma(rho, cl, s.nuL[i]); }
                const double t1 = s.nuU[i] - s.nuL[i];
                gl[i + 1] += t1; gl[i] -= t1;
                const double ta = t1 + rho * (cu - cl);
                s.hw[i + 1] += ta; s.hw[i] -= ta;
            }
            stat = 0.0; scale = 1.0;
#pragma unroll
            for (int t = 0; t < TT; ++t) {
                stat = dmax(stat, fabs(gl[t]));
                scale = dmax(scale, dmax(fabs(s.g(t)), fabs(gl[t] - s.hw[t])));
            }
            stat = pair_max(stat);
            scale = pair_max(scale) + 1.0;
            if (pass == 2) break;
#pragma unroll
            for (int t = 0; t < TT; ++t) s.hw[t] = -s.hw[t];
            solve(s, s.hw);
            if (!settled) {
#pragma unroll
                for (int t = 0; t < TT; ++t) s.ws[t] += s.hw[t];
            }
#pragma unroll
            for (int t = 0; t < TT; ++t) s.wo[t] = xchg(s.ws[t]);
        }
        double viol = 0.0, neg = 0.0, far = 0.0, cact = 0.0;
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const double wi = s.ws[t];
            far = dmax(far, fabs(wi - wbak[t]));
            viol = dmax(viol, dmax(wi - s.hi(t), s.lo(t) - wi));
            cact = dmax(cact, dmax((double)s.isu[t] * fabs(wi - s.hi(t)), (double)s.isl[t] * fabs(s.lo(t) - wi)));
            neg = dmax(neg, dmax(-s.lamU[t], -s.lamL[t]));
        }
#pragma unroll
        for (int i = 0; i < NRT; ++i) {
            const double dv = s.ws[i + 1] - s.ws[i];
            viol = dmax(viol, dmax(dv - s.rr, -dv - s.rr));
            cact = dmax(cact, dmax((double)s.iru[i] * fabs(dv - s.rr), (double)s.irl[i] * fabs(-dv - s.rr)));
            neg = dmax(neg, dmax(-s.nuU[i], -s.nuL[i]));
        }
        viol = pair_max(viol); neg = pair_max(neg); far = pair_max(far); cact = pair_max(cact);
        if (!settled) {
            const bool sane = stat <= 1e-10 * scale && far <= 1e-3;
            if (sane && viol <= 1e-12 && cact <= 1e-12 && neg <= 1e-9 * scale) { accepted = true; settled = true; }
            else if (!sane) settled = true;
        }
        if (__all(settled)) break;
    }
    if (!accepted) {
#pragma unroll
        for (int t = 0; t < TT; ++t) s.ws[t] = wbak[t];
#pragma unroll
        for (int t = 0; t < TT; ++t) s.wo[t] = xchg(s.ws[t]);
    }
    return accepted;
}

// Primal-dual active-set rounds on the lane pair, started from an interior-point iterate
// (constraints with multiplier > slack are taken as active).  Each round is one Newton step
// of the augmented Lagrangian of the equality-constrained QP on the current set (factor of
// H + rho A'A), a first-order multiplier update, and a correction of the set: multipliers
// that turned negative leave, violated constraints enter.  A lane settles when its set
// stood still over the last round and the point passes the KKT test of the original QP
// (stationarity 1e-10 relative, feasibility and active-row residual 1e-12, multipliers
// >= -1e-9 relative); rounds on an unchanged set are refinement steps, so a settled point
// is exact to rounding.  Lanes in `skip` are left alone.  Returns true where the lane
// settled; a lane that did not gets its controls back as they were on entry.
__device__ __forceinline__ bool pdas(State &s, bool skip, int max_rounds)
{
    double hmax = 0.0, wbak[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) { hmax = dmax(hmax, s.HS(t, t)); wbak[t] = s.ws[t]; }
    const double rho = 1e5 * pair_max(hmax);
    // initial set from the interior-point iterate
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const double wi = s.ws[t];
        const bool au = s.lamU[t] > s.hi(t) - wi, al = s.lamL[t] > wi - s.lo(t);
        s.isu[t] = au ? 1.0 : 0.0; s.isl[t] = al ? 1.0 : 0.0;
        s.lamU[t] = au ? s.lamU[t] : 0.0; s.lamL[t] = al ? s.lamL[t] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
        const double dv = s.ws[i + 1] - s.ws[i];
        const bool au = s.nuU[i] > s.rr - dv, al = s.nuL[i] > s.rr + dv;
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
        if (!settled && !changed && stat <= 1e-10 * scale && viol <= 1e-12 && cact <= 1e-12 && neg <= 1e-9 * scale) {
            settled = true;
            accepted = true;
        }
        if (round >= max_rounds || __all(settled)) break;

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

// Stage dynamics of the instance into LDS, the five stages split between the two lanes
// (even lane: stages 0, 2, 4; odd lane: 1, 3 and 4 again, writing the same numbers).  The
// yaw of the linearisation points is rolled from the yaw rates exactly as predict_motion
// does (mpc_diff_drive.py:126-139), the rates of the partner's stages arriving by DPP.
template <int MODEL>
__device__ __forceinline__ void stage_split(State &s, const mpcx_variant &v, const LinPoint &lp)
{
    constexpr int NU = MODEL == MPCX_MODEL_DIFF_DRIVE ? 2 : 4;
    const bool odd = s.odd;
    double u[3][4], th[3];
    SteerTerms k[3];
    int tj[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        int t = 2 * j + (odd ? 1 : 0);
        t = t < TT ? t : TT - 1;
        tj[j] = t;
#pragma unroll
        for (int r = 0; r < 4; ++r) u[j][r] = 0.0;
#pragma unroll
        for (int r = 0; r < NU; ++r) u[j][r] = lp.u ? lp.u[r * TT + t] : lp.cold[r];
        steer_terms<MODEL>(v, u[j], k[j]);
    }
    if (lp.yaw) {
#pragma unroll
        for (int j = 0; j < 3; ++j) th[j] = lp.yaw[tj[j]];
    } else {
        const double w0 = from_even(k[0].omega), w1 = from_odd(k[0].omega);
        const double w2 = from_even(k[1].omega), w3 = from_odd(k[1].omega);
        const double t0 = lp.theta0;
        const double t1 = t0 + w0 * v.dt;
        const double t2 = t1 + w1 * v.dt;
        const double t3 = t2 + w2 * v.dt;
        const double t4 = t3 + w3 * v.dt;
        th[0] = odd ? t1 : t0;
        th[1] = odd ? t3 : t2;
        th[2] = t4;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        double a02, a12, Bm[3][4], C[3], bt[3][2], ct[3];
        linearize_at<MODEL>(v, th[j], u[j], k[j], a02, a12, Bm, C);
        reduce_controls(v, Bm, C, bt, ct);
        volatile lds_double *p = (volatile lds_double *)s.dyn + tj[j] * kDynPerStage * kPairs;
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
// The stage dynamics come from LDS (stage_split); each lane accumulates its own rows of H.
template <int MODEL>
__device__ __forceinline__ void assemble(State &s, const mpcx_variant &v, const LinPoint &lp,
                                         const double (&x0)[3], const double *__restrict__ xref)
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
    double GS[3][TT], GO[3][TT], gacc[TT];
#pragma unroll
    for (int I = 0; I < TT; ++I) {
#pragma unroll
        for (int J = 0; J <= I; ++J) { s.S[bi(I, J)] = 0.0; s.O[bi(I, J)] = 0.0; }
        gacc[I] = 0.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) { GS[r][I] = 0.0; GO[r][I] = 0.0; }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        double d = 2.0 * rwc;
        if (t > 0) d += 2.0 * rdc;
        if (t + 1 < TT) d += 2.0 * rdc;
        s.S[bi(t, t)] = d;
        if (t > 0) s.S[bi(t, t - 1)] = -2.0 * rdc;
    }
    stage_split<MODEL>(s, v, lp);
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
                    s.S[bi(I, J)] = fma(gi, GS[r][J], s.S[bi(I, J)]);
                    s.O[bi(I, J)] = fma(gi, GO[r][J], s.O[bi(I, J)]);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < NBLK; ++e) { s.lds[e * kWave] = s.S[e]; s.lds[(NBLK + e) * kWave] = s.O[e]; }
#pragma unroll
    for (int I = 0; I < TT; ++I) s.lds[(2 * NBLK + I) * kWave] = gacc[I];
}

}  // namespace pair
}  // namespace mpcx
