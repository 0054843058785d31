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
// Built for one horizon (NFIX = 2T, used for the reference's T = 30) the kernel keeps only H in
// LDS: K = H + barrier terms is formed and factorised in registers as 16 x 16 tiles of the f64
// matrix instruction (chol_mfma), the factor stays in registers as rows and as columns for the two
// substitutions (solve), H w is an unrolled walk with the vector broadcast from LDS (matvec_fixed).
// That instantiation holds one wavefront per SIMD (the whole register file).
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

// value of lane j on every lane, j wave-uniform: two v_readlane, no trip through LDS
__device__ __forceinline__ double bcast(double x, int j)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), j), hi = __builtin_amdgcn_readlane(__double2hiint(x), j);
    return __hiloint2double(hi, lo);
}
// Lane exchanges by DPP (a few cycles each; __shfl compiles to ds_bpermute, an LDS round trip per
// 32 bits, and a reduction is a chain of six of them).
template <int CTRL>
__device__ __forceinline__ double dpp(double x)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
constexpr int kQuadSwap1 = 0xB1, kQuadSwap2 = 0x4E, kRowHalfMirror = 0x141, kRowMirror = 0x140, kWaveShr1 = 0x138, kWaveShl1 = 0x130;
// Reductions over the wavefront: a butterfly inside every row of 16 lanes (partner lanes 1 and 2 apart,
// then mirrored in the half row and in the row: every lane ends with its row's result), and the four
// rows' results from lanes 0, 16, 32, 48.
__device__ __forceinline__ double wsum(double x)
{
    x += dpp<kQuadSwap1>(x);
    x += dpp<kQuadSwap2>(x);
    x += dpp<kRowHalfMirror>(x);
    x += dpp<kRowMirror>(x);
    return (bcast(x, 0) + bcast(x, 16)) + (bcast(x, 32) + bcast(x, 48));
}
__device__ __forceinline__ double wmax(double x)
{
    x = dmax(x, dpp<kQuadSwap1>(x));
    x = dmax(x, dpp<kQuadSwap2>(x));
    x = dmax(x, dpp<kRowHalfMirror>(x));
    x = dmax(x, dpp<kRowMirror>(x));
    return dmax(dmax(bcast(x, 0), bcast(x, 16)), dmax(bcast(x, 32), bcast(x, 48)));
}
// value of lane `lane - 2` / `lane + 2` (0 where there is none): two shifts of the whole wavefront by one
__device__ __forceinline__ double from_prev(double x, int) { return dpp<kWaveShr1>(dpp<kWaveShr1>(x)); }
__device__ __forceinline__ double from_next(double x, int) { return dpp<kWaveShl1>(dpp<kWaveShl1>(x)); }

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

// The same product for a horizon known at compile time: the vector goes through LDS and comes back as
// broadcast reads (uniform addresses), the column offsets are constants.  K is not copied: the
// blocked factorisation reads H itself.
template <int N>
__device__ __forceinline__ double matvec_fixed(const Ctx &c, double v)
{
    c.vec[c.lane] = c.act ? v : 0.0;
    __builtin_amdgcn_wave_barrier();
    double acc0 = c.g, acc1 = 0.0;
#pragma unroll
    for (int k = 0; k < N; k += 2) {
        const int t0 = (k * (2 * N - 1 - k)) >> 1, t1 = ((k + 1) * (2 * N - 2 - k)) >> 1;
        const double h0 = c.H[k <= c.row ? t0 + c.row : c.own + k];
        const double h1 = c.H[k + 1 <= c.row ? t1 + c.row : c.own + k + 1];
        acc0 = fma(h0, c.vec[k], acc0);
        acc1 = fma(h1, c.vec[k + 1], acc1);
    }
    __builtin_amdgcn_wave_barrier();
    return acc0 + acc1;
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

// The factorisation for a horizon known at compile time (N = 2T free controls, N <= 64, a multiple
// of 4), blocked for the matrix cores.  The trailing matrix
// lives in registers as ten 16 x 16 tiles of the lower triangle of a 64 x 64 frame (tile (I, J), register
// i of lane l = element (16 I + (l >> 4) + 4 i, 16 J + (l & 15)): the accumulator layout of
// v_mfma_f64_16x16x4); panels of four columns are eliminated one after the other:
//   1. the lanes holding the panel's columns put them at their place in the packed triangle in LDS,
//   2. lane r reads the four entries of row r back, and every lane the panel's 4 x 4 diagonal block
//      (uniform addresses: broadcasts), factors the block and finishes its row: four columns of L,
//      kept as the lane's row of the factor (forward substitution) and written to the triangle,
//   3. the finished columns come back in operand order (lane l: row 16 I + (l & 15), column l >> 4 of
//      the panel) and one MFMA per live tile applies the rank-4 update.
// Against the column-by-column elimination in registers (one v_readlane pair per multiply-add, half of
// them on entries right of the diagonal) the broadcasts are inside the matrix instruction; what is
// left is the chain of 60 reciprocal square roots and three LDS round trips per panel.
typedef double tile_t __attribute__((ext_vector_type(4)));
template <int N>
__device__ __forceinline__ void chol_mfma(const Ctx &c, double db, double dr, double (&R)[N], double (&C)[N])
{
    static_assert(N % 4 == 0 && N <= 64, "panels of four columns in a 64 x 64 frame");
    const int kq = c.lane >> 4, m16 = c.lane & 15;
    int tcol[4];       // tri() of the column this lane holds in the tiles of tile column J
#pragma unroll
    for (int J = 0; J < 4; ++J) tcol[J] = c.tri(16 * J + m16 < N ? 16 * J + m16 : 0);
    // K = H + the barrier (or penalty) terms: `db + dr + dr of lane - 2` on the diagonal, `-dr of
    // lane - 2` two below it (same channel, previous stage); the lanes publish both, the tile
    // registers that hold such an entry pick theirs up
    const double dprev = from_prev(dr, c.lane);
    c.vec[c.lane] = db + dr + dprev;
    c.K[c.lane] = dprev;                       // K's area is free until the first panel comes in
    __builtin_amdgcn_wave_barrier();
    tile_t Kt[10];
#pragma unroll
    for (int J = 0; J < 4; ++J)
#pragma unroll
        for (int I = J; I < 4; ++I)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = 16 * I + kq + 4 * i, col = 16 * J + m16;
                const bool in = (I > J || row >= col) && 16 * I + 4 * i < N;
                const double x = c.H[in ? tcol[J] + row : 0];
                Kt[I * (I + 1) / 2 + J][i] = in ? x : 0.0;
            }
#pragma unroll
    for (int J = 0; J < 4; ++J) {
        const int col = 16 * J + m16;
        const double dd = c.vec[col], ds = c.K[col + 2 < kWave ? col + 2 : 0];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (kq + 4 * i == m16) Kt[J * (J + 1) / 2 + J][i] += dd;
            if (kq + 4 * i == m16 + 2) Kt[J * (J + 1) / 2 + J][i] -= ds;
        }
        if (J < 3 && kq == m16 - 14) Kt[(J + 1) * (J + 2) / 2 + J][0] -= ds;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < N / 4; ++p) {
        const int c0 = 4 * p, J = p >> 2, pc = p & 3;
        // 1. the panel's columns, from the lanes that hold them
        // (rows below N: 16 I + 4 i < N, N being a multiple of 4; at or below the diagonal: always in
        // the tiles under the diagonal one, and there for registers i > pc, never for i < pc, and for
        // i == pc where l >> 4 >= the column's place in the panel -- all but one test fold away)
        if ((m16 >> 2) == pc) {
#pragma unroll
            for (int I = J; I < 4; ++I)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (16 * I + 4 * i >= N || (I == J && i < pc)) continue;
                    const int row = 16 * I + kq + 4 * i;
                    if (I > J || i > pc || kq >= (m16 & 3)) c.K[tcol[J] + row] = Kt[I * (I + 1) / 2 + J][i];
                }
        }
        __builtin_amdgcn_wave_barrier();
        // 2. rows and the diagonal block
        int tq[4];
        tq[0] = c.tri(c0);
#pragma unroll
        for (int q = 1; q < 4; ++q) tq[q] = tq[q - 1] + N - 1 - (c0 + q - 1);
        double a[4], d[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = c.K[tq[q] + c.row];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = q; r < 4; ++r) d[r][q] = c.K[tq[q] + c0 + r];
        const double r0 = rsqrt_nr(d[0][0]);
        const double l10 = d[1][0] * r0, l20 = d[2][0] * r0, l30 = d[3][0] * r0;
        const double r1 = rsqrt_nr(fma(-l10, l10, d[1][1]));
        const double l21 = fma(-l20, l10, d[2][1]) * r1, l31 = fma(-l30, l10, d[3][1]) * r1;
        const double r2 = rsqrt_nr(fma(-l21, l21, fma(-l20, l20, d[2][2])));
        const double l32 = fma(-l31, l21, fma(-l30, l20, d[3][2])) * r2;
        const double r3 = rsqrt_nr(fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, d[3][3]))));
        const double x0 = a[0] * r0;
        const double x1 = fma(-x0, l10, a[1]) * r1;
        const double x2 = fma(-x1, l21, fma(-x0, l20, a[2])) * r2;
        const double x3 = fma(-x2, l32, fma(-x1, l31, fma(-x0, l30, a[3]))) * r3;
        R[c0] = c.lane == c0 ? r0 : x0;
        R[c0 + 1] = c.lane == c0 + 1 ? r1 : x1;
        R[c0 + 2] = c.lane == c0 + 2 ? r2 : x2;
        R[c0 + 3] = c.lane == c0 + 3 ? r3 : x3;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (c.lane >= c0 + q && c.act) c.K[tq[q] + c.lane] = R[c0 + q];
        __builtin_amdgcn_wave_barrier();
        // 3. rank-4 update of the tiles that still hold live entries (rows and columns >= c0 + 4)
        if (c0 + 4 < N) {
            const int I0 = (c0 + 4) >> 4;
            const int tk = kq == 0 ? tq[0] : kq == 1 ? tq[1] : kq == 2 ? tq[2] : tq[3];
            double op[4];
#pragma unroll
            for (int I = I0; I < 4; ++I) {
                const int row = 16 * I + m16;
                const double x = c.K[tk + (row < N ? row : N - 1)];
                op[I] = row < N ? x : 0.0;
            }
#pragma unroll
            for (int Jc = I0; Jc < 4; ++Jc)
#pragma unroll
                for (int I = Jc; I < 4; ++I)
                    Kt[I * (I + 1) / 2 + Jc] = __builtin_amdgcn_mfma_f64_16x16x4f64(-op[I], op[Jc], Kt[I * (I + 1) / 2 + Jc], 0, 0, 0);
            __builtin_amdgcn_wave_barrier();
        }
    }
    // the factor comes back transposed: the lane's own column (entries above the diagonal are whatever
    // the triangle's neighbouring column holds; the backward substitution never uses them)
#pragma unroll
    for (int j = 0; j < N; ++j) C[j] = c.K[c.own + j];
}

// NFIX: the number of free controls when the kernel is built for one horizon (0: any).  The
// factor's rows and columns then stay in registers for the substitutions that follow.
template <int NFIX>
struct Factor {
    double R[NFIX > 0 ? NFIX : 1];   // lane i: row i of L (slot i: 1 / L_ii)
    double C[NFIX > 0 ? NFIX : 1];   // lane i: column i of L (slot i: 1 / L_ii)
};
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

// H w + g, and K = H + terms factored (see add_terms for db, dr)
template <int NFIX>
__device__ __forceinline__ double gradient(const Ctx &c, double v)
{
    if constexpr (NFIX > 0) return matvec_fixed<NFIX>(c, v);
    else return matvec<true>(c, v);
}
template <int NFIX>
__device__ __forceinline__ void factor(const Ctx &c, Factor<NFIX> &F, double db, double dr)
{
    if constexpr (NFIX > 0) {
        chol_mfma<NFIX>(c, db, dr, F.R, F.C);
    } else {
        add_terms(c, db, dr);
        chol(c);
    }
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

// x <- K^-1 r.  With the factor's rows and columns in registers neither substitution touches LDS,
// and both go four columns at a time: the four lanes of a block (a DPP quad, N being a multiple of 4)
// substitute among themselves with quad broadcasts, then the block's four results go to the
// remaining lanes through v_readlane and are applied there at once.  (Column by column the chain is a
// multiply, a v_readlane pair and a multiply-add per column, each waiting for the one before.)
template <int NFIX>
__device__ __forceinline__ double solve(const Ctx &c, const Factor<NFIX> &F, double r)
{
    double acc = c.act ? r : 0.0;
    if constexpr (NFIX > 0) {
        const int q = c.lane & 3, blk = c.lane >> 2;
        // forward: L y = r
#pragma unroll
        for (int p = 0; p < NFIX / 4; ++p) {
            const int c0 = 4 * p;
            double t = acc;
            const double y0 = dpp<0x00>(t * F.R[c0]);
            t = fma(-F.R[c0], y0, t);
            const double y1 = dpp<0x55>(t * F.R[c0 + 1]);
            t = fma(-F.R[c0 + 1], y1, t);
            const double y2 = dpp<0xAA>(t * F.R[c0 + 2]);
            t = fma(-F.R[c0 + 2], y2, t);
            const double y3 = dpp<0xFF>(t * F.R[c0 + 3]);
            const double own = q == 0 ? y0 : q == 1 ? y1 : q == 2 ? y2 : y3;
            const double b0 = bcast(y0, c0), b1 = bcast(y1, c0), b2 = bcast(y2, c0), b3 = bcast(y3, c0);
            const double below = fma(-F.R[c0 + 1], b1, fma(-F.R[c0], b0, acc)) - fma(F.R[c0 + 3], b3, F.R[c0 + 2] * b2);
            acc = blk == p ? own : (blk > p ? below : acc);
        }
        // backward: L' z = y
#pragma unroll
        for (int p = NFIX / 4 - 1; p >= 0; --p) {
            const int c0 = 4 * p;
            double t = acc;
            const double z3 = dpp<0xFF>(t * F.C[c0 + 3]);
            t = fma(-F.C[c0 + 3], z3, t);
            const double z2 = dpp<0xAA>(t * F.C[c0 + 2]);
            t = fma(-F.C[c0 + 2], z2, t);
            const double z1 = dpp<0x55>(t * F.C[c0 + 1]);
            t = fma(-F.C[c0 + 1], z1, t);
            const double z0 = dpp<0x00>(t * F.C[c0]);
            const double own = q == 0 ? z0 : q == 1 ? z1 : q == 2 ? z2 : z3;
            const double b0 = bcast(z0, c0), b1 = bcast(z1, c0), b2 = bcast(z2, c0), b3 = bcast(z3, c0);
            const double above = fma(-F.C[c0 + 1], b1, fma(-F.C[c0], b0, acc)) - fma(F.C[c0 + 3], b3, F.C[c0 + 2] * b2);
            acc = blk == p ? own : (blk < p ? above : acc);
        }
        return c.act ? acc : 0.0;   // lanes beyond N ran along on whatever their registers held
    } else {
        return solve_backward(c, solve_forward(c, acc));
    }
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
        MPCX_STAMP(11);
        const double hw = gradient<NFIX>(c, c.w);
        MPCX_STAMP(12);
        // slacks, barrier terms
        const double wn = from_next(c.w, i);
        const double su = dmax(c.hi - c.w, kSlackMin), sl = dmax(c.w - c.lo, kSlackMin);
        const double dv = wn - c.w;
        const double ru = dmax(c.rr - dv, kSlackMin), rl = dmax(c.rr + dv, kSlackMin);
        const double isu = c.hb ? rcp_raw(su) : 0.0, isl = c.hb ? rcp_raw(sl) : 0.0;
        const double iru = c.hr ? rcp_raw(ru) : 0.0, irl = c.hr ? rcp_raw(rl) : 0.0;
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
        MPCX_STAMP(13);
        factor<NFIX>(c, F, fma(c.lamU, isu, c.lamL * isl), fma(c.nuU, iru, c.nuL * irl));
        MPCX_STAMP(14);
        // predictor
        const double dwa = solve<NFIX>(c, F, -hw);
        MPCX_STAMP(15);
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
        const double hw = gradient<NFIX>(c, c.w);
        const double dv = from_next(c.w, i) - c.w;
        const double vu = c.w - c.hi, vl = c.lo - c.w, ru = dv - c.rr, rl = -dv - c.rr;
        const double cu = au ? vu : 0.0, cl = al ? vl : 0.0, du = bu ? ru : 0.0, dl = bl ? rl : 0.0;
        const double gl = hw + ct_apply(c, c.lamU - c.lamL, c.nuU - c.nuL);
        const double ga = rho * ct_apply(c, cu - cl, du - dl);
        double viol = 0.0, cact = dmax(dmax(fabs(cu), fabs(cl)), dmax(fabs(du), fabs(dl)));
        double neg = dmax(dmax(-c.lamU, -c.lamL), dmax(-c.nuU, -c.nuL));
        if (c.hb) viol = dmax(vu, vl);
        if (c.hr) viol = dmax(viol, dmax(ru, rl));
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
        factor<NFIX>(c, F, rho * ((au ? 1.0 : 0.0) + (al ? 1.0 : 0.0)), rho * ((bu ? 1.0 : 0.0) + (bl ? 1.0 : 0.0)));
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
