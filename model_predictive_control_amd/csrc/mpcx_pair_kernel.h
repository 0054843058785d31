// The lane-pair solve kernel; a template by inclusion like mpcx_pair.h (MPCX_PAIR_NS names the
// instantiation of the pair routines, MPCX_PAIR_KERNEL the kernel).  Included by
// mpcx_kernels.hip after SolveArgs and the stamp macros are defined.
namespace mpcx {

// CHAIN: the MPC.iterativeLMPC flavour (T chained QPs per call); otherwise exactly one QP.
// RANGE 1: only the instances a.range[0] .. a.range[1]-1 (bounds in device memory) are solved.
// RANGE 2: the batch is sorted by mode; run m, instances a.range[m] .. a.range[m+1]-1, is solved
// with the variant a.vtab[m] (device memory), every run by wavefronts of its own.
template <int MODEL, bool CHAIN, int RANGE = 0>
__global__ __launch_bounds__(kWave, MPCX_PAIR_OCC) void MPCX_PAIR_KERNEL(SolveArgs a)
{
    // LDS: Hs = input staging (x0, controls, xbar), then H and g, then output staging;
    // Ds = stage dynamics; Xs = the wave's reference windows (read again by the epilogue)
    // For horizons beyond 5 the stage dynamics share the H area (they are dead once H is
    // condensed) and are rebuilt from the inputs for the epilogue; that keeps two waves per CU.
    constexpr bool kAliasDyn = MPCX_PAIR_NS::TT > 5;
    static_assert(!(CHAIN && kAliasDyn), "chained solves keep their stage dynamics in an area of their own");
    __shared__ double Hs[MPCX_PAIR_NS::kLdsPerLane * kWave];
    __shared__ double Ds[kAliasDyn ? 1 : MPCX_PAIR_NS::kDynDoubles * MPCX_PAIR_NS::kPairs];
    __shared__ double Xs[MPCX_PAIR_NS::kPairs * MPCX_PAIR_NS::padded(3 * (MPCX_PAIR_NS::TT + 1))];
    __shared__ int Fs[kWave + 2];   // per-instance input flags; the wave's record range
    constexpr int NU = MODEL == MPCX_MODEL_DIFF_DRIVE ? 2 : 4;
    constexpr int T = MPCX_PAIR_NS::TT, W = T + 1;
    constexpr int kFixedIters = T > 5 ? kPairIpmIters10 : kPairIpmIters;
    constexpr int SX = MPCX_PAIR_NS::padded(3 * W), SU = MPCX_PAIR_NS::padded(NU * T), S0 = MPCX_PAIR_NS::padded(3);   // record strides in LDS
    constexpr int kInU = MPCX_PAIR_NS::kPairs * S0, kInXbar = kInU + MPCX_PAIR_NS::kPairs * SU;
    static_assert(kInXbar + MPCX_PAIR_NS::kPairs * SX <= MPCX_PAIR_NS::kLdsPerLane * kWave, "input staging must fit in the H area");
    MPCX_PAIR_NS::State s;
    const int lane = threadIdx.x;
    MPCX_STAMP_CLEAR();
    MPCX_STAMP(0);
    s.bind((MPCX_PAIR_NS::lds_double *)Hs, (MPCX_PAIR_NS::lds_double *)(kAliasDyn ? Hs : Ds), lane);
    int64_t b0 = (int64_t)blockIdx.x * MPCX_PAIR_NS::kPairs;
    // the wave's records [first, nb) are solved: all that exist, or those inside the caller's range
    int64_t r_lo = 0, r_hi = a.B;
    const mpcx_variant *vp = &a.v;
    if (RANGE == 1) {
        r_lo = a.range[0] > 0 ? a.range[0] : 0;
        r_hi = a.range[1] < a.B ? a.range[1] : a.B;
        if (b0 + MPCX_PAIR_NS::kPairs <= r_lo || b0 >= r_hi) return;   // wave-uniform
    }
    if (RANGE == 2) {
        // which run this wavefront belongs to: run m owns the wavefronts that its instances touch
        int w = blockIdx.x, m = 0;
        bool mine = false;
        for (; m < a.n_modes; ++m) {
            r_lo = a.range[m] > 0 ? a.range[m] : 0;
            r_hi = a.range[m + 1] < a.B ? a.range[m + 1] : a.B;
            const int nw = r_hi > r_lo ? (int)((r_hi - 1) / MPCX_PAIR_NS::kPairs - r_lo / MPCX_PAIR_NS::kPairs) + 1 : 0;
            if (w < nw) { mine = true; break; }
            w -= nw;
        }
        if (!mine) return;   // wave-uniform
        b0 = (r_lo / MPCX_PAIR_NS::kPairs + w) * MPCX_PAIR_NS::kPairs;
        vp = a.vtab + m;
    }
    const mpcx_variant &v = *vp;
    const int nb = r_hi - b0 < MPCX_PAIR_NS::kPairs ? (int)(r_hi - b0) : MPCX_PAIR_NS::kPairs;
    const int first = r_lo > b0 ? (int)(r_lo - b0) : 0;
    const int pl = lane >> 1;
    const bool live = pl >= first && pl < nb;
    const int pc = live ? pl : nb - 1;   // idle pairs redo the wave's last instance so votes stay uniform
    const int64_t b = b0 + pc;

    MPCX_PAIR_NS::lds_double *stg = (MPCX_PAIR_NS::lds_double *)Hs, *xs = (MPCX_PAIR_NS::lds_double *)Xs;
    // non-finite input raises a per-instance flag while it is staged: [0..31] state, reference
    // and xbar; [32..63] controls (which only count for warm-started instances)
    volatile __attribute__((address_space(3))) int *flg = (volatile __attribute__((address_space(3))) int *)Fs;
    flg[lane] = 0;
    if (RANGE && lane == 0) { flg[kWave] = first; flg[kWave + 1] = nb; }
    bool warm = false;
    auto stage_inputs = [&](bool with_xref) {
        MPCX_PAIR_NS::StageIn<MPCX_PAIR_NS::kPairs * 3 * W, 3 * W> in_xr, in_xb;
        MPCX_PAIR_NS::StageIn<MPCX_PAIR_NS::kPairs * 3, 3> in_x0;
        MPCX_PAIR_NS::StageIn<MPCX_PAIR_NS::kPairs * NU * T, NU * T> in_u;
        if (with_xref) in_xr.load(a.xref + b0 * 3 * W, nb * 3 * W, lane);
        in_x0.load(a.x0 + b0 * 3, nb * 3, lane);
        if (a.uin) in_u.load(a.uin + b0 * NU * T, nb * NU * T, lane);
        if (a.linear_mode) in_xb.load(a.xbar + b0 * 3 * W, nb * 3 * W, lane);
        warm = a.linear_mode || (a.uin && (!a.has_prev || a.has_prev[b]));
        if (with_xref) in_xr.put(nb * 3 * W, xs, lane, flg);
        in_x0.put(nb * 3, stg, lane, flg);
        if (a.uin) in_u.put(nb * NU * T, stg + kInU, lane, flg + MPCX_PAIR_NS::kPairs);
        if (a.linear_mode) in_xb.put(nb * 3 * W, stg + kInXbar, lane, flg);
    };
    stage_inputs(true);
    __builtin_amdgcn_wave_barrier();

    const volatile MPCX_PAIR_NS::lds_double *xr = xs + pc * SX;
    const volatile MPCX_PAIR_NS::lds_double *x0s = stg + pc * S0;
    double x0[3] = {x0s[0], x0s[1], x0s[2]};
#ifdef MPCX_STAMPS
    if (x0[0] == 1.2345e300) return;   // forces the loads to have landed before the stamp
#endif
    MPCX_STAMP(8);
    bool coldfin = true;
#pragma unroll
    for (int r = 0; r < NU; ++r) coldfin = coldfin && finite_d(v.cold_start[r]);
    const bool finite = flg[pc] == 0 && (warm ? flg[MPCX_PAIR_NS::kPairs + pc] == 0 : coldfin);
    MPCX_PAIR_NS::Lin lp;
    lp.u = stg + kInU + pc * SU;
    lp.cold = v.cold_start;
    lp.warm = warm;
    lp.has_yaw = a.linear_mode != 0;
    lp.yaw = stg + kInXbar + pc * SX + 2 * W;
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
    MPCX_STAMP(9);
    {
        FreeBounds fb;   // the first-stage window hangs on the incoming first column, in every pass
        free_bounds(v, u0, fb);
        if (!fb.feasible && !skip) { skip = true; status = MPCX_ST_INFEASIBLE; }
        s.b_ = s.odd ? fb.b[1] : fb.b[0];
        s.rr = s.odd ? fb.rr[1] : fb.rr[0];
        // a channel without a rate bound gets one that its box makes redundant (|w| <= b at
        // every stage implies |w[t+1] - w[t]| <= 2b < 4b): same optimum, no special case
        if (!finite_d(s.rr)) s.rr = 4.0 * s.b_;
        s.lo0 = s.odd ? fb.lo0[1] : fb.lo0[0];
        s.hi0 = s.odd ? fb.hi0[1] : fb.hi0[0];
    }
    // MPC.iterativeLMPC solves T QPs per call, each linearised about column `pass` of the
    // previous answer (mpc_sim.py:293-306); every other caller solves one.
    const int chain = !CHAIN ? 1 : (v.chain < T ? v.chain : T);
    const double sg = v.eq_mode == MPCX_EQ_ACKERMANN ? -1.0 : 1.0;
    MPCX_PAIR_NS::LinRegs L;
    bool inaccurate = false;
    MPCX_PAIR_NS::prepare_lin<MODEL>(s, v, lp, CHAIN, L);
    MPCX_STAMP(11);
#pragma unroll 1
    for (int pass = 0; pass < chain; ++pass) {
        if (CHAIN && pass > 0) {
            // column `pass` of the linearisation controls <- the previous answer
            double wp0 = 0.0, wp1 = 0.0;
#pragma unroll
            for (int t = 1; t < T; ++t) {
                if (t == pass) {
                    wp0 = s.odd ? s.wo[t] : s.ws[t];
                    wp1 = s.odd ? s.ws[t] : s.wo[t];
                }
            }
            double un[4];
            if (v.eq_mode == MPCX_EQ_NONE) { un[0] = wp0; un[1] = wp1; un[2] = 0.0; un[3] = 0.0; }
            else if (v.eq_mode == MPCX_EQ_FIXED_STEER) { un[0] = wp0; un[1] = v.steer_fixed; un[2] = wp1; un[3] = v.steer_fixed; }
            else { un[0] = wp0; un[1] = wp1; un[2] = wp0; un[3] = sg * wp1; }
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (L.tj[j] == pass) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) L.u[j][r] = un[r];
                }
            }
        }
        MPCX_PAIR_NS::stage_eval<MODEL>(s, v, L, CHAIN);
        if (pass == 0) MPCX_STAMP(10);
        MPCX_PAIR_NS::condense(s, v, x0, xr);
        if (pass == 0) MPCX_STAMP(1);
        // Attempt 0: a fixed, wave-uniform number of interior-point iterations, then active-set
        // rounds that either certify the exact optimum or give up.  Attempt 1 (only for lanes
        // attempt 0 left unsettled; the whole wavefront waits for them, and the launch for the
        // wavefront): kPairMoreIters further iterations from the iterate attempt 0 reached, which
        // was kept, and the rounds again.  Attempt 2: interior point from the start to convergence,
        // then the rounds; if those do not settle either, the converged iterate is the answer.
        bool need = !skip;
        MPCX_PAIR_NS::Iterate keep;
#pragma unroll 1
        for (int attempt = 0; attempt < 3; ++attempt) {
            int st = MPCX_ST_OK, it = 0;
            if (attempt == 1) keep.restore(s, need);
            const int cap = attempt == 0 ? kFixedIters : kPairMoreIters;
            MPCX_PAIR_NS::ipm(s, attempt < 2 ? (a.max_iter < cap ? a.max_iter : cap) : a.max_iter, attempt < 2, !need, st, it, attempt == 1);
            if (attempt == 0) keep.save(s);
            if (pass == 0 && attempt < 2) MPCX_STAMP(2 + 2 * attempt);
            if (need) { iters += it; status = st; }
            // the rounds run even after an interior-point solve that stalled short of its
            // tolerances: they certify their own answer, and a stalled iterate usually has the set
#ifdef MPCX_STAMPS
            int rounds = 0;
            const bool settled = MPCX_PAIR_NS::pdas(s, !need, attempt < 2 ? kPairRounds0 : kPairRounds1, &rounds);
            if (pass == 0 && attempt == 0 && threadIdx.x == 0 && blockIdx.x < kStampWaves) g_stamps[blockIdx.x * kStampSlots + 7] = rounds;
#elif defined(MPCX_DIAG)   // diagnostic builds report the rounds in the iteration count
            int rounds = 0;
            const bool settled = MPCX_PAIR_NS::pdas(s, !need, attempt < 2 ? kPairRounds0 : kPairRounds1, &rounds);
            if (need) iters += (attempt == 0 ? 100 : 10000) * (rounds + (attempt > 0 ? 1 : 0)) + (settled ? 0 : 1000000 * (attempt + 1));
#else
            const bool settled = MPCX_PAIR_NS::pdas(s, !need, attempt < 2 ? kPairRounds0 : kPairRounds1);
#endif
            if (pass == 0 && attempt < 2) MPCX_STAMP(3 + 2 * attempt);
#ifdef MPCX_TRACE_B
            if (b == MPCX_TRACE_B && live)
                printf("b %ld lane %d pass %d attempt %d need %d ipm st %d it %d settled %d ws %g %g %g %g %g\n", (long)b,
                       (int)threadIdx.x, pass, attempt, (int)need, st, it, (int)settled, s.ws[0], s.ws[1], s.ws[2],
                       s.ws[3], s.ws[4]);
#endif
            if (need && settled) status = MPCX_ST_OK;
            need = need && !settled;
            if (!__any(need)) break;
        }
        if (!skip && status == MPCX_ST_FAILED) skip = true;   // the reference's chain dies with its first failed solve
        if (CHAIN && !skip && status == MPCX_ST_INACCURATE) inaccurate = true;
    }
    // an inaccurate link makes the whole chain's answer inaccurate, whatever the last solve says
    if (CHAIN && inaccurate && status == MPCX_ST_OK) status = MPCX_ST_INACCURATE;

    // epilogue (both lanes evaluate it; the even lane stages the controls, the odd lane the
    // states and the objective; the wave then writes its 32 records as contiguous blocks)
    const bool bad = status == MPCX_ST_NONFINITE || status == MPCX_ST_INFEASIBLE;
    const double nan = __builtin_nan("");
    double J = 0.0, qg = 1.0;
    double x[3] = {x0[0], x0[1], x0[2]};
    double up[4] = {0, 0, 0, 0};
    constexpr int kOutU = kAliasDyn ? MPCX_PAIR_NS::kDynDoubles * MPCX_PAIR_NS::kPairs : 0;
    constexpr int kOutX = kOutU + MPCX_PAIR_NS::kPairs * SU, kOutJ = kOutX + MPCX_PAIR_NS::kPairs * SX;
    static_assert(kOutJ + MPCX_PAIR_NS::kPairs <= MPCX_PAIR_NS::kLdsPerLane * kWave, "output staging must fit in the H area");
    if (kAliasDyn) {   // H is dead: bring the linearisation point back and rebuild the stage dynamics
        stage_inputs(false);
        __builtin_amdgcn_wave_barrier();
        lp.theta0 = status == MPCX_ST_NONFINITE ? 0.0 : x0[2];
        MPCX_PAIR_NS::prepare_lin<MODEL>(s, v, lp, false, L);
        MPCX_PAIR_NS::stage_eval<MODEL>(s, v, L, false);
    }
    volatile MPCX_PAIR_NS::lds_double *uo = stg + kOutU + pl * SU, *xo = stg + kOutX + pl * SX;
    const bool st_u = !s.odd, st_x = s.odd;
    if (st_x) {
#pragma unroll
        for (int r = 0; r < 3; ++r) xo[r * W] = bad ? nan : x[r];
    }
    // all LDS reads of the epilogue up front (the solver's registers are free now), so that the
    // arithmetic below does not wait on them one by one
    MPCX_PAIR_NS::Dyn dyn_all[T];
    double xr_all[3][T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        MPCX_PAIR_NS::load_dyn(s, t, dyn_all[t]);
#pragma unroll
        for (int r = 0; r < 3; ++r) xr_all[r][t] = xr[r * W + t + 1];
    }
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const double w0 = s.odd ? s.wo[t] : s.ws[t], w1 = s.odd ? s.ws[t] : s.wo[t];
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
        const MPCX_PAIR_NS::Dyn &dy = dyn_all[t];
        const double a02 = dy.a02, a12 = dy.a12;
        const double (&bt)[3][2] = dy.bt;
        const double (&ct)[3] = dy.ct;
        const double n0 = x[0] + a02 * x[2] + bt[0][0] * w0 + bt[0][1] * w1 + ct[0];
        const double n1 = x[1] + a12 * x[2] + bt[1][0] * w0 + bt[1][1] * w1 + ct[1];
        const double n2 = x[2] + bt[2][0] * w0 + bt[2][1] * w1 + ct[2];
        x[0] = n0; x[1] = n1; x[2] = n2;
        const int ts = t + 1;
        if (ts < T) qg *= v.q_growth;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const double e = xr_all[r][t] - x[r];
            const double q = ts < T ? v.Q[r] * qg : v.Qf[r];
            J = fma(q * e, e, J);
            if (st_x) xo[r * W + ts] = bad ? nan : x[r];
        }
    }
    if (st_x) ((volatile MPCX_PAIR_NS::lds_double *)stg)[kOutJ + pl] = bad ? nan : J;
    MPCX_STAMP(12);
    __builtin_amdgcn_wave_barrier();
    // the wave's record range again, from LDS (kept in registers over the solve it would cost the
    // chained kernel registers it does not have)
    const int first_o = RANGE ? flg[kWave] : 0, nb_o = RANGE ? flg[kWave + 1] : nb;
    MPCX_PAIR_NS::stage_out<MPCX_PAIR_NS::kPairs * NU * T, NU * T>(a.u_out + b0 * NU * T, first_o * NU * T, nb_o * NU * T, stg + kOutU, lane);
    MPCX_PAIR_NS::stage_out<MPCX_PAIR_NS::kPairs * 3 * W, 3 * W>(a.x_out + b0 * 3 * W, first_o * 3 * W, nb_o * 3 * W, stg + kOutX, lane);
    if (lane >= first_o && lane < nb_o) a.obj_out[b0 + lane] = ((const volatile MPCX_PAIR_NS::lds_double *)stg)[kOutJ + lane];
    if (live && !s.odd) {
        a.status_out[b] = status;
        if (a.iters_out) a.iters_out[b] = iters;
    }
    MPCX_STAMP(6);
}


}  // namespace mpcx
