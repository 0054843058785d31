// One wavefront per MPC instance, for horizons the lane-pair kernels do not cover
// (n = 2T free controls, n <= 64, i.e. T <= 32; the reference's T = 30 runs are the case
// in point).
//
// Lane i owns free control i = (stage i >> 1, channel i & 1): its value, its two box
// constraints, the two rate constraints that tie it to control i + 2 (same channel, next
// stage) and row i of every matrix.  Per-lane state is a handful of scalars; the matrices
// (both symmetric) live in LDS as packed lower triangles, column after column, so that two of
// them fit in under 30 KB at n = 60 and five instances share a compute unit: the kernel issues
// one instruction stream per instance, and what it needs most is enough of them per SIMD.
//
// Solver = the scheme of the stage-wise kernel on the condensed problem: interior-point
// iterations (predictor-corrector, centring from the predictor's step lengths) until the
// complementarity is small enough to read the active set off the iterate, then primal-dual
// active-set rounds that certify the KKT conditions of the QP; if they do not settle the
// interior point goes on from where it stood and the rounds are tried again.  One instance per
// wavefront: an instance takes the iterations it needs and waits for no other.
#pragma once
#include "mpcx_device.h"

namespace mpcx {
namespace wave {

typedef __attribute__((address_space(3))) double lds_double;

__device__ __forceinline__ double wsum(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, kWave);
    return x;
}
__device__ __forceinline__ double wmax(double x)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = dmax(x, __shfl_xor(x, off, kWave));
    return x;
}
// value of lane j on every lane, j wave-uniform: two v_readlane, no trip through LDS
__device__ __forceinline__ double bcast(double x, int j)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), j), hi = __builtin_amdgcn_readlane(__double2hiint(x), j);
    return __hiloint2double(hi, lo);
}
// value of lane `lane - 2` / `lane + 2` (0 where there is none)
__device__ __forceinline__ double from_prev(double x, int lane) { const double y = __shfl_up(x, 2, kWave); return lane >= 2 ? y : 0.0; }
__device__ __forceinline__ double from_next(double x, int lane) { const double y = __shfl_down(x, 2, kWave); return lane < kWave - 2 ? y : 0.0; }

struct Ctx {
    volatile lds_double *H;     // packed lower triangle of H (see tri())
    volatile lds_double *K;     // same layout: K, then its Cholesky factor (diagonal = 1 / L_ii)
    volatile lds_double *vec;   // [64] broadcast buffer
    int n, lane;
    int row;                    // lane, or 0 on lanes beyond n (a valid address; never written)
    int own;                    // tri(row): element (j, row) of the lane's own column sits at own + j
    bool act;                   // lane < n
    bool hb, hr;                // this lane has a (finite) box / a (finite) rate constraint to lane + 2
    double lo, hi, rr;          // its bounds
    double g;                   // its gradient entry
    double inv_m;               // 1 / number of inequality constraints of the instance
    // iterate
    double w, lamU, lamL, nuU, nuL;
    // Element (i, k), k <= i, of a packed matrix sits at tri(k) + i: column k starts at
    // tri(k) + k and the columns follow one another without gaps.  A lane walking along its
    // row (k uniform) reads consecutive addresses across the wave; walking down its own
    // column (own + j) it meets bank conflicts of a few ways at most.
    __device__ __forceinline__ int tri(int k) const { return (k * (2 * n - 1 - k)) >> 1; }
    __device__ __forceinline__ int size() const { return (n * (n + 1)) >> 1; }
};

// hw_i = g_i + sum_k H(i, k) v_k from the packed triangle (row walk left of the diagonal,
// column walk below it); v_k comes from lane k's register (v_readlane).
// COPY_K: K <- H on the way (same layout: a straight copy).
template <bool COPY_K>
__device__ __forceinline__ double matvec(const Ctx &c, double v)
{
    v = c.act ? v : 0.0;
    if (COPY_K) {
        const int sz = c.size();
        int e = c.lane;
        for (; e + 3 * kWave < sz; e += 4 * kWave) {
            double t[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) t[q] = c.H[e + q * kWave];
#pragma unroll
            for (int q = 0; q < 4; ++q) c.K[e + q * kWave] = t[q];
        }
        for (; e < sz; e += kWave) c.K[e] = c.H[e];
    }
    double acc0 = c.g, acc1 = 0.0;
    int k = 0, tk = 0;          // tk = tri(k), advanced by n - 1 - k per column
    for (; k + 7 < c.n; k += 8) {   // loads of eight columns in flight together
        double h[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            h[q] = c.H[k + q <= c.row ? tk + c.row : c.own + k + q];
            tk += c.n - 1 - (k + q);
        }
#pragma unroll
        for (int q = 0; q < 8; q += 2) {
            acc0 = fma(h[q], bcast(v, k + q), acc0);
            acc1 = fma(h[q + 1], bcast(v, k + q + 1), acc1);
        }
    }
    for (; k + 1 < c.n; k += 2) {
        const double h0 = c.H[k <= c.row ? tk + c.row : c.own + k];
        tk += c.n - 1 - k;
        const double h1 = c.H[k + 1 <= c.row ? tk + c.row : c.own + k + 1];
        tk += c.n - 2 - k;
        acc0 = fma(h0, bcast(v, k), acc0);
        acc1 = fma(h1, bcast(v, k + 1), acc1);
    }
    __builtin_amdgcn_wave_barrier();
    return acc0 + acc1;   // n is even
}

// Q columns k .. k + Q - 1 of the trailing update K(i, c) -= le_i le_c + ld_i ld_c (c <= i):
// the lane's own entries are read unconditionally (so the loads go out together; a lane
// left of the diagonal reads a neighbouring column's entry and drops it), le_c / ld_c come
// from lane c's registers.  `tk` = tri(k).
template <int Q>
__device__ __forceinline__ int chol_trail(const Ctx &c, int k, int tk, double le, double ld)
{
    double kv[Q];
    int at[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        at[q] = tk + c.row;
        kv[q] = c.K[at[q]];
        tk += c.n - 1 - (k + q);
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        const double lek = bcast(le, k + q), ldk = bcast(ld, k + q);
        kv[q] = fma(-le, lek, fma(-ld, ldk, kv[q]));
    }
#pragma unroll
    for (int q = 0; q < Q; ++q)
        if (c.lane >= k + q && c.act) c.K[at[q]] = kv[q];
    return tk;
}

// In-place right-looking Cholesky of the packed K (rows of inactive lanes are ignored).
// K(j, j) ends up as 1 / L_jj.  Two columns per sweep of the trailing matrix (rank-2 update).
__device__ __forceinline__ void chol(const Ctx &c)
{
    const int n = c.n, i = c.lane;
    int tj = 0;                              // tri(j)
    for (int j = 0; j < n; j += 2) {
        const int tj1 = tj + n - 1 - j;      // tri(j + 1)
        // 2x2 diagonal block, on every lane; the lane's panel entries are fetched alongside
        const double a = c.K[tj + j], b = c.K[tj + j + 1], d = c.K[tj1 + j + 1];
        const double p0 = c.K[tj + c.row], p1 = c.K[tj1 + c.row];
        const double r0 = rsqrt_nr(a);
        const double l10 = b * r0;
        const double r1 = rsqrt_nr(fma(-l10, l10, d));
        // panel: rows below the block
        const bool below = i > j + 1 && c.act;
        double le = p0 * r0;
        double ld = fma(-le, l10, p1) * r1;
        le = below ? le : 0.0;
        ld = below ? ld : 0.0;
        if (below) {
            c.K[tj + i] = le;
            c.K[tj1 + i] = ld;
        }
        if (i == j) c.K[tj + j] = r0;
        if (i == j + 1) { c.K[tj + j + 1] = l10; c.K[tj1 + j + 1] = r1; }
        // trailing update of columns j + 2 .. n - 1 (an even count)
        int k = j + 2, tk = tj1 + n - 2 - j;  // tri(j + 2)
        tj = tk;
        for (; k + 7 < n; k += 8) tk = chol_trail<8>(c, k, tk, le, ld);
        if (k + 3 < n) { tk = chol_trail<4>(c, k, tk, le, ld); k += 4; }
        if (k + 1 < n) chol_trail<2>(c, k, tk, le, ld);
        __builtin_amdgcn_wave_barrier();
    }
}

// The same factorisation for a horizon known at compile time (N = 2T free controls): lane i takes
// row i of the packed K into registers, the columns are eliminated one by one there (the column's
// entry of lane k reaches the other lanes through v_readlane, N (N - 1) / 2 of them), and the factor
// goes back to the packed triangle for the solves.  The version above walks the matrix in LDS and is
// bound by the round trips (157 k cycles at N = 60, three quarters of the kernel's time; this one
// 46 k; sending the column through LDS and reading it back as broadcasts instead: 78 k).
// Entries right of the diagonal in a lane's registers are whatever the load found; they stay in
// that lane (only entries at or below the diagonal are ever broadcast or stored).
template <int N>
__device__ __forceinline__ void chol_regs(const Ctx &c, double (&R)[N])
{
    {
        int tk = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) { R[k] = c.K[tk + c.row]; tk += N - 1 - k; }
    }
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double r = rsqrt_nr(bcast(R[j], j));
        const double lj = R[j] * r;              // lanes below the diagonal: L(i, j)
        R[j] = c.lane == j ? r : lj;             // the diagonal slot keeps 1 / L_jj
#pragma unroll
        for (int k = j + 1; k < N; ++k) R[k] = fma(-lj, bcast(lj, k), R[k]);
    }
    {
        int tk = 0;
#pragma unroll
        for (int k = 0; k < N; ++k) {
            if (c.lane >= k && c.act) c.K[tk + c.lane] = R[k];
            tk += N - 1 - k;
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// NFIX: the number of free controls when the kernel is built for one horizon (0: any).  The
// factor's rows then stay in registers for the forward substitutions that follow.
template <int NFIX>
struct Factor {
    double R[NFIX > 0 ? NFIX : 1];
};
template <int NFIX>
__device__ __forceinline__ void factor(const Ctx &c, Factor<NFIX> &F)
{
    if constexpr (NFIX > 0) chol_regs<NFIX>(c, F.R);
    else chol(c);
}

// x <- K^-1 r with the factor left by chol(); every lane passes its entry and gets its entry.
// The factor entries of the next four columns are fetched while the chain of broadcasts works
// through the current four.
__device__ __forceinline__ double solve_forward(const Ctx &c, double acc)
{
    const int n = c.n, i = c.lane;
    const int row = c.row;
    const int n4 = n & ~3;
    double dj[4], lij[4], djn[4], lijn[4];
    // forward: L y = r, column by column
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = q < n ? q : 0, tc = c.tri(col);
        dj[q] = c.K[tc + col]; lij[q] = c.K[tc + row];
    }
    int j = 0;
    for (; j < n4; j += 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = j + 4 + q < n ? j + 4 + q : 0, tc = c.tri(col);
            djn[q] = c.K[tc + col]; lijn[q] = c.K[tc + row];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double yj = bcast(acc, j + q) * dj[q];
            if (i == j + q) acc = yj;
            else if (i > j + q && i < n) acc = fma(-lij[q], yj, acc);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { dj[q] = djn[q]; lij[q] = lijn[q]; }
    }
    for (int q = 0; j < n; ++j, ++q) {   // n % 4 == 2: the two columns already fetched
        const double yj = bcast(acc, j) * (q == 0 ? dj[0] : dj[1]);
        if (i == j) acc = yj;
        else if (i > j && i < n) acc = fma(-(q == 0 ? lij[0] : lij[1]), yj, acc);
    }
    return acc;
}

__device__ __forceinline__ double solve_backward(const Ctx &c, double acc)
{
    const int n = c.n, i = c.lane;
    const int row = c.row;
    double dj[4], lij[4], djn[4], lijn[4];
    int j;
    // backward: L' z = y, row by row of L (the lane walks down its own column)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int col = n - 1 - q >= 0 ? n - 1 - q : 0;
        dj[q] = c.K[c.tri(col) + col]; lij[q] = c.K[col >= row ? c.own + col : 0];
    }
    j = n - 1;
    for (; j >= 3; j -= 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = j - 4 - q >= 0 ? j - 4 - q : 0;
            djn[q] = c.K[c.tri(col) + col]; lijn[q] = c.K[col >= row ? c.own + col : 0];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double zj = bcast(acc, j - q) * dj[q];
            if (i == j - q) acc = zj;
            else if (i < j - q) acc = fma(-lij[q], zj, acc);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) { dj[q] = djn[q]; lij[q] = lijn[q]; }
    }
    for (int q = 0; j >= 0; --j, ++q) {
        const double zj = bcast(acc, j) * (q == 0 ? dj[0] : dj[1]);
        if (i == j) acc = zj;
        else if (i < j) acc = fma(-(q == 0 ? lij[0] : lij[1]), zj, acc);
    }
    return acc;
}

// x <- K^-1 r.  With the factor's rows in registers the forward substitution needs no LDS: lane j
// scales its entry by 1 / L_jj (its diagonal slot), one v_readlane pair hands it to the lanes below.
// The backward substitution walks columns, which only the packed triangle in LDS offers.
template <int NFIX>
__device__ __forceinline__ double solve(const Ctx &c, const Factor<NFIX> &F, double r)
{
    double acc = c.act ? r : 0.0;
    if constexpr (NFIX > 0) {
#pragma unroll
        for (int j = 0; j < NFIX; ++j) {
            const double yj = bcast(acc * F.R[j], j);
            acc = c.lane == j ? yj : (c.lane > j && c.act ? fma(-F.R[j], yj, acc) : acc);
        }
    } else {
        acc = solve_forward(c, acc);
    }
    return solve_backward(c, acc);
}

// K diagonal and same-channel sub-diagonal terms: `db` on the lane's own diagonal entry, `dr`
// for its rate constraint (touches (i, i), (i + 2, i + 2) and (i + 2, i)).
__device__ __forceinline__ void add_terms(const Ctx &c, double db, double dr)
{
    const int i = c.lane;
    const double dprev = from_prev(dr, i);
    if (c.act) {
        c.K[c.own + i] += db + dr + dprev;
        if (i >= 2) c.K[c.tri(i - 2) + i] -= dprev;
    }
    __builtin_amdgcn_wave_barrier();
}

// C' applied to per-constraint values: box part `tb` (upper minus lower) and rate part `tr`.
__device__ __forceinline__ double ct_apply(const Ctx &c, double tb, double tr) { return tb + from_prev(tr, c.lane) - tr; }

// The interior point's iterate, kept across the active-set rounds (which overwrite it) so that
// a later attempt can go on from it.
struct Iterate {
    double w, lamU, lamL, nuU, nuL;
    __device__ __forceinline__ void save(const Ctx &c) { w = c.w; lamU = c.lamU; lamL = c.lamL; nuU = c.nuU; nuL = c.nuL; }
    __device__ __forceinline__ void restore(Ctx &c) const { c.w = w; c.lamU = lamU; c.lamL = lamL; c.nuU = nuU; c.nuL = nuL; }
};

// Predictor-corrector interior point (see pair kernels for the formulas).  With mu_stop > 0 it
// stops once the complementarity is below mu_stop and the dual residual small (the active set
// can then be read off the iterate); otherwise it runs to convergence; max_iter iterations at
// most either way.  `resume` goes on from the iterate in `c`
// instead of starting over.  Status / iteration count are wave-uniform: one instance per wave.
template <int NFIX>
__device__ __forceinline__ void ipm(Ctx &c, int max_iter, double mu_stop, bool resume, int &status, int &iters)
{
    const int i = c.lane;
    if (!resume) {
        // strictly feasible start: every stage at the middle of its channel's first-stage window
        double wc = c.hb ? 0.5 * (c.lo + c.hi) : 0.0;     // valid on lanes 0, 1
        wc = __shfl(wc, i & 1, kWave);
        c.w = c.act ? wc : 0.0;
        // The complementarity a constraint starts at follows the gradient (at zero controls) of
        // its own control, capped by the instance's largest: with stage weights that grow like 2^t
        // (mpc_dsr.py:349) the gradient reaches 1e7, and from 0.1 everywhere the iteration spends
        // 25 steps raising the late stages' multipliers to that scale.
        const double gi = c.act ? fabs(c.g) : 0.0;
        const double mu0 = kMu0 * dmax(1.0, wmax(gi));
        const double mub = dmin(mu0, kMu0 * dmax(1.0, gi));
        const double mur = dmin(mu0, kMu0 * dmax(1.0, dmax(gi, from_next(gi, i))));
        c.lamU = c.hb ? mub * rcp_nr(dmax(c.hi - wc, kSlackMin)) : 0.0;
        c.lamL = c.hb ? mub * rcp_nr(dmax(wc - c.lo, kSlackMin)) : 0.0;
        const double l0 = c.hr ? mur * rcp_nr(c.rr) : 0.0;
        c.nuU = l0; c.nuL = l0;
    }
    status = MPCX_ST_FAILED;
    iters = 0;
    Factor<NFIX> F;
    for (int it = 0;; ++it) {
        const double hw = matvec<true>(c, c.w);
        // slacks, barrier terms
        const double wn = from_next(c.w, i);
        const double su = dmax(c.hi - c.w, kSlackMin), sl = dmax(c.w - c.lo, kSlackMin);
        const double dv = wn - c.w;
        const double ru = dmax(c.rr - dv, kSlackMin), rl = dmax(c.rr + dv, kSlackMin);
        const double isu = c.hb ? rcp_raw(su) : 0.0, isl = c.hb ? rcp_raw(sl) : 0.0;
        const double iru = c.hr ? rcp_raw(ru) : 0.0, irl = c.hr ? rcp_raw(rl) : 0.0;
        add_terms(c, fma(c.lamU, isu, c.lamL * isl), fma(c.nuU, iru, c.nuL * irl));
        const double ctl = ct_apply(c, c.lamU - c.lamL, c.nuU - c.nuL);
        double musum = 0.0;
        if (c.hb) musum = fma(su, c.lamU, sl * c.lamL);
        if (c.hr) musum = fma(ru, c.nuU, fma(rl, c.nuL, musum));
        const double mu = wsum(musum) * c.inv_m;
        const double rdmax = wmax(c.act ? fabs(hw + ctl) : 0.0);
        const double scale = dmax(1.0, wmax(c.act ? dmax(fabs(hw - c.g), dmax(fabs(c.g), fabs(ctl))) : 0.0)) + 1.0;
        const bool conv = rdmax <= kTolD * scale && mu <= kTolMu;
        if (conv) { status = MPCX_ST_OK; iters = it; return; }
        const bool handover = mu_stop > 0.0 && mu <= mu_stop && rdmax <= kSwitchD * scale;
        // no constraint at all: zero complementarity by definition, the iteration is a Newton step
        if (handover || (c.inv_m > 0.0 && mu < kMuFloor) || it >= max_iter || !finite_d(mu + rdmax)) {
            iters = it;
            status = (rdmax <= kLooseD * scale && mu <= kLooseMu) ? MPCX_ST_INACCURATE : MPCX_ST_FAILED;
            return;
        }
        factor<NFIX>(c, F);
        // predictor
        const double dwa = solve<NFIX>(c, F, -hw);
        const double dva = from_next(dwa, i) - dwa;
        const double xu = dwa * isu, xl = -dwa * isl, yu = dva * iru, yl = -dva * irl;
        double rp = 0.0, rdm = 1.0;
        if (c.hb) { rp = dmax(xu, xl); rdm = dmax(1.0, dmax(1.0 - xu, 1.0 - xl)); }
        if (c.hr) { rp = dmax(rp, dmax(yu, yl)); rdm = dmax(rdm, dmax(1.0 - yu, 1.0 - yl)); }
        rp = wmax(rp); rdm = wmax(rdm);
        const double apa = rp > 1.0 ? rcp_nr(rp) : 1.0, ada = rcp_nr(rdm);
        const double back = 1.0 - dmin(apa, ada);
        const double sm = back * back * back * mu;
        // corrector as a correction to the predictor: K (dw - dwa) = -C' q
        const double qu = c.hb ? fma(sm, isu, c.lamU * xu * (xu - 1.0)) : 0.0;
        const double ql = c.hb ? fma(sm, isl, c.lamL * xl * (xl - 1.0)) : 0.0;
        const double pu = c.hr ? fma(sm, iru, c.nuU * yu * (yu - 1.0)) : 0.0;
        const double pl = c.hr ? fma(sm, irl, c.nuL * yl * (yl - 1.0)) : 0.0;
        const double dw = dwa + solve<NFIX>(c, F, -ct_apply(c, qu - ql, pu - pl));
        const double dvv = from_next(dw, i) - dw;
        const double zu = dw * isu, zl = -dw * isl, vu = dvv * iru, vl = -dvv * irl;
        const double dlu = fma(c.lamU, zu, qu - c.lamU), dll = fma(c.lamL, zl, ql - c.lamL);
        const double dnu = fma(c.nuU, vu, pu - c.nuU), dnl = fma(c.nuL, vl, pl - c.nuL);
        double rp2 = 0.0, rd2 = 0.0;
        if (c.hb) { rp2 = dmax(zu, zl); rd2 = dmax(-dlu * rcp_raw(c.lamU), -dll * rcp_raw(c.lamL)); }
        if (c.hr) { rp2 = dmax(rp2, dmax(vu, vl)); rd2 = dmax(rd2, dmax(-dnu * rcp_raw(c.nuU), -dnl * rcp_raw(c.nuL))); }
        rp2 = wmax(rp2); rd2 = wmax(rd2);
        // a factorisation that broke down leaves NaN in the step: stop at the iterate we have
        const double chk = wsum(c.act ? dw + dlu + dll + dnu + dnl : 0.0);
        if (!finite_d(chk)) {
            iters = it;
            status = (rdmax <= kLooseD * scale && mu <= kLooseMu) ? MPCX_ST_INACCURATE : MPCX_ST_FAILED;
            return;
        }
        const double eta = dmin(kEtaCap, dmax(0.99, 1.0 - mu));
        const double ap = rp2 > eta ? eta * rcp_nr(rp2) : 1.0;
        const double ad = rd2 > eta ? eta * rcp_nr(rd2) : 1.0;
        if (c.act) c.w = fma(ap, dw, c.w);
        c.lamU = fma(ad, dlu, c.lamU); c.lamL = fma(ad, dll, c.lamL);
        c.nuU = fma(ad, dnu, c.nuU); c.nuL = fma(ad, dnl, c.nuL);
    }
}

#ifndef MPCX_WAVE_GUESS
#define MPCX_WAVE_GUESS 1.0
#endif
// first active set: multiplier > kWaveGuess x slack (see kPairGuess; here 0.3 changes nothing and 0.15
// costs the 2^t-weighted T = 30 variant 17 %: an instance has its wavefront to itself, nobody waits)
constexpr double kWaveGuess = MPCX_WAVE_GUESS;

// Primal-dual active-set rounds (see pdas() of the pair kernels).  Returns true when the
// point is certified; otherwise the controls are put back as they were.
template <int NFIX>
__device__ __forceinline__ bool pdas(Ctx &c, int max_rounds, double hmax)
{
    const int i = c.lane;
    const double wbak = c.w;
    double rho = dmin(1e5 * hmax, kRhoCap);
    bool au, al, bu, bl;
    {
        const double dv = from_next(c.w, i) - c.w;
        au = c.hb && c.lamU > kWaveGuess * (c.hi - c.w); al = c.hb && c.lamL > kWaveGuess * (c.w - c.lo);
        bu = c.hr && c.nuU > kWaveGuess * (c.rr - dv); bl = c.hr && c.nuL > kWaveGuess * (c.rr + dv);
        c.lamU = au ? c.lamU : 0.0; c.lamL = al ? c.lamL : 0.0;
        c.nuU = bu ? c.nuU : 0.0; c.nuL = bl ? c.nuL : 0.0;
    }
    bool changed = true, accepted = false, eased = false;
    Factor<NFIX> F;
    for (int round = 0;; ++round) {
        const double hw = matvec<true>(c, c.w);
        const double dv = from_next(c.w, i) - c.w;
        const double vu = c.w - c.hi, vl = c.lo - c.w, ru = dv - c.rr, rl = -dv - c.rr;
        const double cu = au ? vu : 0.0, cl = al ? vl : 0.0, du = bu ? ru : 0.0, dl = bl ? rl : 0.0;
        const double gl = hw + ct_apply(c, c.lamU - c.lamL, c.nuU - c.nuL);
        const double ga = rho * ct_apply(c, cu - cl, du - dl);
        double viol = 0.0, cact = dmax(dmax(fabs(cu), fabs(cl)), dmax(fabs(du), fabs(dl)));
        double neg = dmax(dmax(-c.lamU, -c.lamL), dmax(-c.nuU, -c.nuL));
        if (c.hb) viol = dmax(vu, vl);
        if (c.hr) viol = dmax(viol, dmax(ru, rl));
        add_terms(c, rho * ((au ? 1.0 : 0.0) + (al ? 1.0 : 0.0)), rho * ((bu ? 1.0 : 0.0) + (bl ? 1.0 : 0.0)));
        const double stat = wmax(c.act ? fabs(gl) : 0.0);
        const double scale = dmax(1.0, wmax(c.act ? dmax(fabs(c.g), fabs(hw - c.g)) : 0.0)) + 1.0;
        viol = wmax(viol); cact = wmax(cact); neg = wmax(neg);
        const bool fin = finite_d(wsum(c.act ? gl + ga + c.w : 0.0));   // max() drops NaN operands
        const bool on_set = !changed && fin && viol <= 1e-12 && cact <= 1e-12 && neg <= 1e-9 * scale;
        if (on_set && stat <= 1e-10 * scale) {
            accepted = true;
            break;
        }
        // On its active set with the set at rest, yet not stationary: the multiplier update
        // lam += rho c(w) carries the rounding of the controls (an ulp of w, 1e-16) times rho into
        // the multipliers, which for a large Hessian and a small gradient is above the tolerance.
        // The remaining rounds (and two more) use a hundredth of the penalty: the point is already
        // on the constraints, so the slower contraction costs little, and the floor drops with it.
        if (on_set && !eased) {
            eased = true;
            rho *= 1e-2;
            max_rounds += 2;
        }
        if (round >= max_rounds || !fin) break;
        MPCX_STAMP(8);
        factor<NFIX>(c, F);
        MPCX_STAMP(9);
        const double step = solve<NFIX>(c, F, -(gl + ga));
        MPCX_STAMP(10);
#ifdef MPCX_STAMPS
        if (threadIdx.x == 0 && blockIdx.x < kStampWaves) g_stamps[blockIdx.x * kStampSlots + 7] += 1;
#endif
        if (c.act) c.w += step;
        const double ntol = -1e-9 * scale;
        const double dv2 = from_next(c.w, i) - c.w;
        const double vu2 = c.w - c.hi, vl2 = c.lo - c.w, ru2 = dv2 - c.rr, rl2 = -dv2 - c.rr;
        const double lu = au ? fma(rho, vu2, c.lamU) : 0.0, ll = al ? fma(rho, vl2, c.lamL) : 0.0;
        const double nu = bu ? fma(rho, ru2, c.nuU) : 0.0, nl = bl ? fma(rho, rl2, c.nuL) : 0.0;
        const bool au2 = c.hb && ((au && lu >= ntol) || vu2 > 1e-12), al2 = c.hb && ((al && ll >= ntol) || vl2 > 1e-12);
        const bool bu2 = c.hr && ((bu && nu >= ntol) || ru2 > 1e-12), bl2 = c.hr && ((bl && nl >= ntol) || rl2 > 1e-12);
        const bool ch = au2 != au || al2 != al || bu2 != bu || bl2 != bl;
        changed = __any(ch) != 0;
#ifdef MPCX_WAVE_TRACE
        if (blockIdx.x == MPCX_WAVE_TRACE) {
            if (i == 0) printf("round %d stat %.3e scale %.3e viol %.3e cact %.3e neg %.3e\n", round, stat, scale, viol, cact, neg);
            if (ch) printf("   lane %d: au %d->%d al %d->%d bu %d->%d bl %d->%d | lu %.3e ll %.3e nu %.3e nl %.3e | vu2 %.3e vl2 %.3e ru2 %.3e rl2 %.3e ntol %.3e\n",
                           i, au, au2, al, al2, bu, bu2, bl, bl2, lu, ll, nu, nl, vu2, vl2, ru2, rl2, ntol);
        }
#endif
        au = au2; al = al2; bu = bu2; bl = bl2;
        c.lamU = au ? dmax(lu, 0.0) : 0.0; c.lamL = al ? dmax(ll, 0.0) : 0.0;
        c.nuU = bu ? dmax(nu, 0.0) : 0.0; c.nuL = bl ? dmax(nl, 0.0) : 0.0;
    }
    if (!accepted) c.w = wbak;
    return accepted;
}

}  // namespace wave
}  // namespace mpcx
