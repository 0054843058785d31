// The four-lanes-per-instance solve kernel (mpcx_quad.h); included by mpcx_kernels.hip after
// SolveArgs is defined.  One QP per call (no mpc_sim chain), 16 instances per wavefront.
namespace mpcx {

#ifndef MPCX_QUAD_OCC
#define MPCX_QUAD_OCC 1
#endif
#ifndef MPCX_QUAD_K
#define MPCX_QUAD_K 6
#endif

constexpr int kQuadIpmIters = MPCX_QUAD_K;   // interior-point iterations before the active-set rounds
constexpr int kQuadRounds0 = 6, kQuadRounds1 = 4;

// (Instantiated at TT = 5 with two wavefronts per SIMD the same code takes 107 us for the 65,536
// instances the lane-pair kernel solves in 80 us, 145 us with one: the second wavefront hides
// latency, but four lanes per instance cost more instructions per instance than two.)
template <int MODEL, int TT>
__global__ __launch_bounds__(kWave, MPCX_QUAD_OCC) void k_solve_quad(SolveArgs a)
{
    using S = quad::State<TT>;
    using quad::kInst;
    using quad::lds_double;
    constexpr int NU = MODEL == MPCX_MODEL_DIFF_DRIVE ? 2 : 4;
    constexpr int T = TT, W = T + 1;
    // LDS.  Hs: input staging (x0, controls, xbar), stage dynamics and yaw rates side by side;
    // then H and g over all of it; then, for the epilogue, the inputs and the dynamics again (H is
    // dead) and the output staging over the inputs.  Xs: the wave's reference windows.
    constexpr int SX = quad::padded(3 * W), SU = quad::padded(NU * T), S0 = quad::padded(3);
    constexpr int kInU = kInst * S0, kInXbar = kInU + kInst * SU, kDynOff = kInXbar + kInst * SX;
    constexpr int kOmOff = kDynOff + quad::kDynPerStage * T * kInst;
    constexpr int kHsDoubles = kOmOff + T * kInst > S::kLdsH ? kOmOff + T * kInst : S::kLdsH;   // T = 10: the H area holds all of it
    constexpr int kOutU = 0, kOutX = kOutU + kInst * SU, kOutJ = kOutX + kInst * SX;
    static_assert(kOutJ + kInst <= kDynOff, "output staging must fit in front of the dynamics");
    __shared__ double Hs[kHsDoubles];
    __shared__ double Xs[kInst * SX];
    __shared__ int Fs[kWave];
    const mpcx_variant &v = a.v;
    S s;
    const int lane = threadIdx.x;
    s.bind((lds_double *)Hs, (lds_double *)Hs + kDynOff, lane);
    const int64_t b0 = (int64_t)blockIdx.x * kInst;
    const int nb = a.B - b0 < kInst ? (int)(a.B - b0) : kInst;
    const int il = lane >> 2, q = lane & 3;
    const bool live = il < nb;
    const int ic = live ? il : nb - 1;   // idle quads redo the wave's last instance so votes stay uniform
    const int64_t b = b0 + ic;

    lds_double *stg = (lds_double *)Hs, *xs = (lds_double *)Xs;
    volatile lds_double *om = (volatile lds_double *)Hs + kOmOff + il;
    // non-finite input raises a per-instance flag while it is staged: [0..15] state, reference and
    // xbar; [16..31] controls (which only count for warm-started instances)
    volatile __attribute__((address_space(3))) int *flg = (volatile __attribute__((address_space(3))) int *)Fs;
    flg[lane] = 0;
    bool warm = false;
    auto stage_inputs = [&](bool with_xref) {
        quad::StageIn<kInst * 3 * W, 3 * W> in_xr, in_xb;
        quad::StageIn<kInst * 3, 3> in_x0;
        quad::StageIn<kInst * NU * T, NU * T> in_u;
        if (with_xref) in_xr.load(a.xref + b0 * 3 * W, nb * 3 * W, lane);
        in_x0.load(a.x0 + b0 * 3, nb * 3, lane);
        if (a.uin) in_u.load(a.uin + b0 * NU * T, nb * NU * T, lane);
        if (a.linear_mode) in_xb.load(a.xbar + b0 * 3 * W, nb * 3 * W, lane);
        warm = a.linear_mode || (a.uin && (!a.has_prev || a.has_prev[b]));
        if (with_xref) in_xr.put(nb * 3 * W, xs, lane, flg);
        in_x0.put(nb * 3, stg, lane, flg);
        if (a.uin) in_u.put(nb * NU * T, stg + kInU, lane, flg + kInst);
        if (a.linear_mode) in_xb.put(nb * 3 * W, stg + kInXbar, lane, flg);
    };
    stage_inputs(true);
    __builtin_amdgcn_wave_barrier();

    const volatile lds_double *xr = xs + ic * SX;
    const volatile lds_double *x0s = stg + ic * S0;
    double x0[3] = {x0s[0], x0s[1], x0s[2]};
    bool coldfin = true;
#pragma unroll
    for (int r = 0; r < NU; ++r) coldfin = coldfin && finite_d(v.cold_start[r]);
    const bool finite = flg[ic] == 0 && (warm ? flg[kInst + ic] == 0 : coldfin);
    quad::Lin<TT> lp;
    lp.u = stg + kInU + ic * SU;
    lp.cold = v.cold_start;
    lp.warm = warm;
    lp.has_yaw = a.linear_mode != 0;
    lp.yaw = stg + kInXbar + ic * SX + 2 * W;
    lp.theta0 = x0[2];
    double u0[4] = {0.0, 0.0, 0.0, 0.0};   // first-stage controls: read before H overwrites the staging
#pragma unroll
    for (int r = 0; r < NU; ++r) u0[r] = lp.control(r, 0);

    int status = MPCX_ST_OK, iters = 0;
    bool skip = !finite;
    if (!finite) {
        status = MPCX_ST_NONFINITE;
        x0[0] = x0[1] = x0[2] = 0.0;
        lp.theta0 = 0.0;
    }
    {
        FreeBounds fb;   // the first-stage window hangs on the incoming first column
        free_bounds(v, u0, fb);
        if (!fb.feasible && !skip) { skip = true; status = MPCX_ST_INFEASIBLE; }
        const double bb = s.rowodd ? fb.b[1] : fb.b[0];
        double rr = s.rowodd ? fb.rr[1] : fb.rr[0];
        // a channel without a rate bound gets one that its box makes redundant (|w| <= b at every
        // stage implies |w[t+1] - w[t]| <= 2b < 4b): same optimum, no special case
        if (!finite_d(rr)) rr = 4.0 * bb;
        s.set_bounds(bb, rr, s.rowodd ? fb.lo0[1] : fb.lo0[0], s.rowodd ? fb.hi0[1] : fb.hi0[0]);
    }
    quad::stage_eval<MODEL, TT>(s, v, lp, om, q);
    quad::condense<TT>(s, v, x0, xr);
#ifdef MPCX_QUAD_TRACE
    if (blockIdx.x == 0 && lane < 4) {
        printf("lane %d H00 %.12g H10 %.12g H11 %.12g H20 %.12g H99 %.12g g0 %.12g g1 %.12g g9 %.12g bu0 %g buR %g bl0 %g lo0 %g hi0 %g\n", lane,
               s.H(0, 0), s.H(1, 0), s.H(1, 1), s.H(2, 0), s.H(TT - 1, TT - 1), s.g(0), s.g(1), s.g(TT - 1), s.bu0, s.buR, s.bl0, s.lo0, s.hi0);
    }
#endif
    // Attempt 0: a fixed, wave-uniform number of interior-point iterations, then active-set rounds
    // that either certify the exact optimum or give up.  Attempt 1 (only for instances attempt 0
    // left unsettled): interior point to convergence, then the same rounds; if those do not settle
    // either, the converged interior-point iterate is the answer.
    bool need = !skip;
#pragma unroll 1
    for (int attempt = 0; attempt < 2; ++attempt) {
        int st = MPCX_ST_OK, it = 0;
        quad::ipm<TT>(s, attempt == 0 ? (a.max_iter < kQuadIpmIters ? a.max_iter : kQuadIpmIters) : a.max_iter, attempt == 0,
                      !need, st, it);
        if (need) { iters += it; status = st; }
#ifdef MPCX_DIAG   // diagnostic builds report the rounds in the iteration count
        int rounds = 0;
        const bool settled = quad::pdas<TT>(s, !need, attempt == 0 ? kQuadRounds0 : kQuadRounds1, &rounds);
        if (need) iters += (attempt == 0 ? 100 : 10000) * (rounds + attempt) + (settled ? 0 : 1000000 * (attempt + 1));
#else
        const bool settled = quad::pdas<TT>(s, !need, attempt == 0 ? kQuadRounds0 : kQuadRounds1);
#endif
        if (need && settled) status = MPCX_ST_OK;
        need = need && !settled;
        if (!__any(need)) break;
    }

    // epilogue (the four lanes evaluate it; lane 0 of the quad stages the controls, lane 1 the
    // states and the objective; the wave then writes its 16 records as contiguous blocks)
    const bool bad = status == MPCX_ST_NONFINITE || status == MPCX_ST_INFEASIBLE;
    const double nan = __builtin_nan("");
    // H is dead: bring the linearisation point back and rebuild the stage dynamics
    stage_inputs(false);
    __builtin_amdgcn_wave_barrier();
    // the initial state again from the staged inputs (kept in registers over the solve it would be spilled)
    const bool zeroed = status == MPCX_ST_NONFINITE;
    const double x0e[3] = {zeroed ? 0.0 : x0s[0], zeroed ? 0.0 : x0s[1], zeroed ? 0.0 : x0s[2]};
    const double sg = v.eq_mode == MPCX_EQ_ACKERMANN ? -1.0 : 1.0;
    lp.theta0 = x0e[2];
    quad::stage_eval<MODEL, TT>(s, v, lp, om, q);
    quad::Dyn dyn_all[T];
    double xr_all[3][T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        quad::load_dyn(s, t, dyn_all[t]);
#pragma unroll
        for (int r = 0; r < 3; ++r) xr_all[r][t] = xr[r * W + t + 1];
    }
    __builtin_amdgcn_wave_barrier();   // the staged inputs are dead from here on: the outputs go over them
    volatile lds_double *uo = stg + kOutU + il * SU, *xo = stg + kOutX + il * SX;
    const bool st_u = q == 0, st_x = q == 1;
    double J = 0.0, qg = 1.0;
    double x[3] = {x0e[0], x0e[1], x0e[2]};
    double up[4] = {0, 0, 0, 0};
    if (st_x) {
#pragma unroll
        for (int r = 0; r < 3; ++r) xo[r * W] = bad ? nan : x[r];
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const double other = quad::x1(s.ws[t]);
        const double w0 = s.rowodd ? other : s.ws[t], w1 = s.rowodd ? s.ws[t] : other;
        double u[4];
        if (v.eq_mode == MPCX_EQ_NONE) { u[0] = w0; u[1] = w1; u[2] = 0.0; u[3] = 0.0; }
        else if (v.eq_mode == MPCX_EQ_FIXED_STEER) { u[0] = w0; u[1] = v.steer_fixed; u[2] = w1; u[3] = v.steer_fixed; }
        else { u[0] = w0; u[1] = w1; u[2] = w0; u[3] = sg * w1; }
#pragma unroll
        for (int r = 0; r < NU; ++r) {
            J = fma(v.R[r] * u[r], u[r], J);
            if (t > 0) { const double d = u[r] - up[r]; J = fma(v.Rd[r] * d, d, J); }
            up[r] = u[r];
            if (st_u) uo[r * T + t] = bad ? nan : u[r];
        }
        const quad::Dyn &dy = dyn_all[t];
        const double n0 = x[0] + dy.a02 * x[2] + dy.bt[0][0] * w0 + dy.bt[0][1] * w1 + dy.ct[0];
        const double n1 = x[1] + dy.a12 * x[2] + dy.bt[1][0] * w0 + dy.bt[1][1] * w1 + dy.ct[1];
        const double n2 = x[2] + dy.bt[2][0] * w0 + dy.bt[2][1] * w1 + dy.ct[2];
        x[0] = n0; x[1] = n1; x[2] = n2;
        const int ts = t + 1;
        if (ts < T) qg *= v.q_growth;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double e = xr_all[r][t] - x[r];
            const double qq = ts < T ? v.Q[r] * qg : v.Qf[r];
            J = fma(qq * e, e, J);
            if (st_x) xo[r * W + ts] = bad ? nan : x[r];
        }
    }
    if (st_x) ((volatile lds_double *)stg)[kOutJ + il] = bad ? nan : J;
    __builtin_amdgcn_wave_barrier();
    quad::stage_out<kInst * NU * T, NU * T>(a.u_out + b0 * NU * T, nb * NU * T, stg + kOutU, lane);
    quad::stage_out<kInst * 3 * W, 3 * W>(a.x_out + b0 * 3 * W, nb * 3 * W, stg + kOutX, lane);
    if (lane < nb) a.obj_out[b0 + lane] = ((const volatile lds_double *)stg)[kOutJ + lane];
    if (live && q == 0) {
        a.status_out[b] = status;
        if (a.iters_out) a.iters_out[b] = iters;
    }
}

}  // namespace mpcx
