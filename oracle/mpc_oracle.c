/*
 * CPU restatement in plain C (float64) of the reference's iterative linear MPC
 * path, batched over independent instances.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the checker the HIP engine is compared
 * against at batch sizes the numpy twin (mpc_oracle.py) is too slow for, and the
 * CPU-baseline "port" the benchmark times.  The product library never links,
 * loads or calls it.
 *
 * Pinning: rollout / linearisation / nearest index / reference window are
 * pinned by golden vectors captured from the reference's own functions
 * (tests/golden).  The QP solve is PARITY UNPINNED: the reference delegates it
 * to cvxpy -> ECOS (not vendored, no version pinned, absent from this image;
 * call sites mpc_diff_drive.py:199-200, mpc_dsr.py:245-246,313-314,373-374,
 * mpc_sim.py:358-359,418-419,486-487, mpc.py:165-166) and holds no test or
 * fixture for it.  The QP is strictly convex in the free controls, so the
 * minimiser is unique; this file computes it with a primal-dual interior point
 * method and reports the KKT residuals of what it returns.
 *
 * Layouts (row-major, one record per instance):
 *   x0   [B][3]          xref/xbar/x_out [B][3][T+1]
 *   u..  [B][NU][T]      dual-steer rows are v_f, delta_f, v_r, delta_r
 *                        (mpc_dsr.py:186-190, mpc.py:99-102)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MODEL_DIFF_DRIVE 0
#define ORC_MODEL_DSR 1
#define ORC_EQ_NONE 0
#define ORC_EQ_ACKERMANN 1
#define ORC_EQ_CRAB 2
#define ORC_EQ_FIXED_STEER 3
#define ORC_ST_OK 0
#define ORC_ST_INACCURATE 1
#define ORC_ST_INFEASIBLE 2
#define ORC_ST_NONFINITE 3
#define ORC_MAX_T 64

typedef struct {
    int32_t model, nu, T, eq_mode, chain, pad_;
    double dt, wb;
    double R[4], Rd[4];
    double Q[3], q_growth, Qf[3];
    double umax[4], dumax[4], first_step[4];
    double steer_fixed;
    double cold_start[4];
} orc_variant;

/* ---------------------------------------------------------------- models */

/* mpc_dsr.py:141-167: closed form of (H^T H)^-1 H^T with H^T H = diag(2,2,wb^2/2) */
static void wheel_to_twist(double vf, double sf, double vr, double sr, double wb,
                           double *vx, double *vy, double *w)
{
    double v1x = cos(sf) * vf, v1y = sin(sf) * vf;
    double v2x = cos(sr) * vr, v2y = sin(sr) * vr;
    *vx = 0.5 * (v1x + v2x);
    *vy = 0.5 * (v1y + v2y);
    *w = (v1y - v2y) / wb;
}

/* one plant step; u points at one column (stride = T between rows) */
static void model_step(const orc_variant *v, double *x, double *y, double *yaw,
                       const double *u, int stride)
{
    double c = cos(*yaw), s = sin(*yaw);
    if (v->model == ORC_MODEL_DIFF_DRIVE) { /* mpc_diff_drive.py:92-100 */
        *x = *x + u[0] * c * v->dt;
        *y = *y + u[0] * s * v->dt;
        *yaw = *yaw + u[stride] * v->dt;
    } else { /* mpc_dsr.py:114-123 */
        double vx, vy, w;
        wheel_to_twist(u[0], u[stride], u[2 * stride], u[3 * stride], v->wb, &vx, &vy, &w);
        *x = *x + vx * c * v->dt - vy * s * v->dt;
        *y = *y + vx * s * v->dt + vy * c * v->dt;
        *yaw = *yaw + w * v->dt;
    }
}

/* predict_motion (mpc_diff_drive.py:126-139, mpc_dsr.py:125-139) */
static void rollout(const orc_variant *v, const double *x0, const double *u, double *xbar)
{
    int T = v->T, W = T + 1;
    double x = x0[0], y = x0[1], yaw = x0[2];
    xbar[0] = x; xbar[W] = y; xbar[2 * W] = yaw;
    for (int t = 0; t < T; ++t) {
        model_step(v, &x, &y, &yaw, u + t, T);
        xbar[t + 1] = x; xbar[W + t + 1] = y; xbar[2 * W + t + 1] = yaw;
    }
}

/* A = I + [0,2]=a02, [1,2]=a12;  Bm 3x4 (first nu columns used);  C[3] */
static void linearize(const orc_variant *v, double theta, const double *u, int stride,
                      double *a02, double *a12, double Bm[3][4], double C[3])
{
    double dt = v->dt;
    memset(Bm, 0, sizeof(double) * 12);
    if (v->model == ORC_MODEL_DIFF_DRIVE) { /* mpc_diff_drive.py:220-238 */
        double vr = u[0], s = sin(theta), c = cos(theta);
        *a02 = -vr * s * dt;
        *a12 = vr * c * dt;
        Bm[0][0] = c * dt;
        Bm[1][0] = s * dt;
        Bm[2][1] = dt;
        C[0] = vr * s * theta * dt;
        C[1] = -vr * c * theta * dt;
        C[2] = 0.0;
        return;
    }
    /* mpc_dsr.py:397-426, entry by entry (not the analytic Jacobian: :415, :416, :422-424) */
    double v_f = u[0], d_f = u[stride], v_r = u[2 * stride], d_r = u[3 * stride];
    double sf = sin(d_f), cf = cos(d_f), sr = sin(d_r), cr = cos(d_r);
    double st = sin(theta), ct = cos(theta);
    double h = 1.0 / 2.0 * dt, k = 1.0 / v->wb * dt;
    *a02 = -0.5 * dt * (v_f * cf + v_r * cr) * st - 0.5 * dt * (v_f * sf + v_r * sr) * ct;
    *a12 = 0.5 * dt * (v_f * cf + v_r * cr) * ct - 0.5 * dt * (v_f * sf + v_r * sr) * st;
    Bm[0][0] = h * (cf * ct - sf * st);
    Bm[0][1] = -h * (sf * ct + cf * st) * v_f;
    Bm[0][2] = h * (cr * ct - sr * st);
    Bm[0][3] = -h * (sr * ct + cr * st) * v_r;
    Bm[1][0] = h * (cf * st + sf * ct);
    Bm[1][1] = -h * (sf * st - cf * ct) * v_f;
    Bm[1][2] = h * (cr * st + sr * ct);
    Bm[1][3] = -h * (sr * ct - cr * st) * v_r;
    Bm[2][0] = k * sr;
    Bm[2][1] = k * cf * v_f;
    Bm[2][2] = -k * sr;
    Bm[2][3] = -k * cr * v_r;
    double af = d_f + theta, ar = d_r + theta;
    C[0] = h * (v_f * cf * st * af + v_r * cr * st * ar + v_f * sf * ct * af + v_r * sr * ct * ar);
    C[1] = h * (-v_f * cf * ct * af - v_r * cr * ct * ar + v_f * sf * st * af + v_r * sr * st * ar);
    C[2] = k * (-cf * d_f * v_f + cr * d_r * v_r);
}

/* ---------------------------------------------------------------- QP set-up */

typedef struct {
    int n, m;
    double *H, *g;        /* n*n, n */
    int *ip, *im;         /* constraint j: +w[ip] - w[im] <= d  (index -1 = absent) */
    double *d;
    double c0;
    /* per-stage reduced dynamics: x+ = A x + Bt w_t + ct */
    double *a02, *a12, *Bt, *ct; /* T, T, T*6, T*3 */
    double E[4][2], f[4];
    double *work;         /* solver scratch */
} qp_t;

static size_t qp_doubles(int T)
{
    size_t n = 2 * (size_t)T, m = 4 * n;
    /* H g d a02 a12 Bt ct | Gcur(3n) | solver: K(n*n) s lam dsv dlv(4m) rd dw rhs tmp(4n) */
    return n * n + n + m + T + T + 6 * T + 3 * T + 3 * n + n * n + 6 * m + 5 * n + 64;
}

static void control_map(const orc_variant *v, double E[4][2], double f[4])
{
    memset(E, 0, sizeof(double) * 8);
    memset(f, 0, sizeof(double) * 4);
    switch (v->eq_mode) {
    case ORC_EQ_NONE: E[0][0] = 1; E[1][1] = 1; break;
    case ORC_EQ_ACKERMANN: E[0][0] = 1; E[2][0] = 1; E[1][1] = 1; E[3][1] = -1; break; /* mpc_dsr.py:242-243 */
    case ORC_EQ_CRAB: E[0][0] = 1; E[2][0] = 1; E[1][1] = 1; E[3][1] = 1; break;       /* mpc_dsr.py:310-311 */
    default: E[0][0] = 1; E[2][1] = 1; f[1] = v->steer_fixed; f[3] = v->steer_fixed;   /* mpc_dsr.py:369-370 */
    }
}

static double dmin(double a, double b) { return a < b ? a : b; }
static double dmax(double a, double b) { return a > b ? a : b; }

/* Returns 0 if the constraint set is empty (reference would print the error and
 * return Nones), else 1.  lo/hi: [T][2]; rr[2]. */
static int free_bounds(const orc_variant *v, const double *uref, int stride,
                       double *lo, double *hi, double rr[2])
{
    const double *um = v->umax, *dm = v->dumax, *fs = v->first_step;
    double u0[4] = {0, 0, 0, 0};
    for (int r = 0; r < v->nu; ++r) u0[r] = uref[r * stride];
    double b[2], wl[2], wh[2];
    int ok = 1;
    if (v->eq_mode == ORC_EQ_NONE) {
        b[0] = um[0]; b[1] = um[1]; rr[0] = dm[0]; rr[1] = dm[1];
        wl[0] = u0[0] - fs[0]; wh[0] = u0[0] + fs[0];
        wl[1] = u0[1] - fs[1]; wh[1] = u0[1] + fs[1];
    } else if (v->eq_mode == ORC_EQ_ACKERMANN || v->eq_mode == ORC_EQ_CRAB) {
        double sg = v->eq_mode == ORC_EQ_ACKERMANN ? -1.0 : 1.0;
        b[0] = dmin(um[0], um[2]); b[1] = dmin(um[1], um[3]);
        rr[0] = dmin(dm[0], dm[2]); rr[1] = dmin(dm[1], dm[3]);
        wl[0] = dmax(u0[0] - fs[0], u0[2] - fs[2]); wh[0] = dmin(u0[0] + fs[0], u0[2] + fs[2]);
        wl[1] = dmax(u0[1] - fs[1], sg * u0[3] - fs[3]); wh[1] = dmin(u0[1] + fs[1], sg * u0[3] + fs[3]);
    } else {
        b[0] = um[0]; b[1] = um[2]; rr[0] = dm[0]; rr[1] = dm[2];
        wl[0] = u0[0] - fs[0]; wh[0] = u0[0] + fs[0];
        wl[1] = u0[2] - fs[2]; wh[1] = u0[2] + fs[2];
        for (int r = 1; r <= 3; r += 2)
            if (fabs(v->steer_fixed) > um[r] || fabs(u0[r] - v->steer_fixed) > fs[r]) ok = 0;
    }
    for (int t = 0; t < v->T; ++t)
        for (int c = 0; c < 2; ++c) { lo[2 * t + c] = -b[c]; hi[2 * t + c] = b[c]; }
    for (int c = 0; c < 2; ++c) {
        lo[c] = dmax(lo[c], wl[c]);
        hi[c] = dmin(hi[c], wh[c]);
        if (!(lo[c] <= hi[c])) ok = 0;
    }
    return ok;
}

/* Build the condensed QP of mpc_diff_drive.py:170-197 and siblings. */
static int condense(const orc_variant *v, const double *xref, const double *xbar,
                    const double *x0, const double *uref, qp_t *q, double *mem)
{
    int T = v->T, W = T + 1, n = 2 * T;
    q->n = n;
    q->H = mem; mem += (size_t)n * n;
    q->g = mem; mem += n;
    q->d = mem; mem += 4 * n;
    q->a02 = mem; mem += T;
    q->a12 = mem; mem += T;
    q->Bt = mem; mem += 6 * T;
    q->ct = mem; mem += 3 * T;
    double *G = mem; mem += 3 * n; /* rows x,y,yaw of d x_t / d w */
    q->work = mem;
    memset(q->H, 0, sizeof(double) * n * n);
    memset(q->g, 0, sizeof(double) * n);
    memset(G, 0, sizeof(double) * 3 * n);
    control_map(v, q->E, q->f);

    double rw[2], rdw[2], rconst = 0.0;
    if (v->eq_mode == ORC_EQ_NONE) {
        rw[0] = v->R[0]; rw[1] = v->R[1]; rdw[0] = v->Rd[0]; rdw[1] = v->Rd[1];
    } else if (v->eq_mode == ORC_EQ_FIXED_STEER) {
        rw[0] = v->R[0]; rw[1] = v->R[2]; rdw[0] = v->Rd[0]; rdw[1] = v->Rd[2];
        rconst = (v->R[1] + v->R[3]) * v->steer_fixed * v->steer_fixed;
    } else {
        rw[0] = v->R[0] + v->R[2]; rw[1] = v->R[1] + v->R[3];
        rdw[0] = v->Rd[0] + v->Rd[2]; rdw[1] = v->Rd[1] + v->Rd[3];
    }
    q->c0 = T * rconst;
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < 2; ++c) q->H[(2 * t + c) * n + 2 * t + c] += 2.0 * rw[c];
    for (int t = 0; t + 1 < T; ++t)
        for (int c = 0; c < 2; ++c) {
            int i = 2 * t + c, j = i + 2;
            q->H[i * n + i] += 2.0 * rdw[c];
            q->H[j * n + j] += 2.0 * rdw[c];
            q->H[i * n + j] -= 2.0 * rdw[c];
            q->H[j * n + i] -= 2.0 * rdw[c];
        }

    double p[3] = {x0[0], x0[1], x0[2]};
    double qg = 1.0;
    for (int t = 0; t < T; ++t) {
        double Bm[3][4], C[3], a02, a12;
        linearize(v, xbar[2 * W + t], uref + t, T, &a02, &a12, Bm, C);
        q->a02[t] = a02; q->a12[t] = a12;
        double *Bt = q->Bt + 6 * t, *ct = q->ct + 3 * t;
        for (int r = 0; r < 3; ++r) {
            double b0 = 0, b1 = 0, cc = C[r];
            for (int k = 0; k < v->nu; ++k) {
                b0 += Bm[r][k] * q->E[k][0];
                b1 += Bm[r][k] * q->E[k][1];
                cc += Bm[r][k] * q->f[k];
            }
            Bt[2 * r] = b0; Bt[2 * r + 1] = b1; ct[r] = cc;
        }
        /* p <- A p + ct ; G <- A G ; G[:, 2t:2t+2] = Bt */
        double pn0 = p[0] + a02 * p[2] + ct[0];
        double pn1 = p[1] + a12 * p[2] + ct[1];
        double pn2 = p[2] + ct[2];
        p[0] = pn0; p[1] = pn1; p[2] = pn2;
        for (int j = 0; j < 2 * t; ++j) {
            G[j] += a02 * G[2 * n + j];
            G[n + j] += a12 * G[2 * n + j];
        }
        for (int r = 0; r < 3; ++r) { G[r * n + 2 * t] = Bt[2 * r]; G[r * n + 2 * t + 1] = Bt[2 * r + 1]; }
        /* stage t+1 cost weight */
        double Qs[3];
        int ts = t + 1;
        if (ts < T) {
            qg *= v->q_growth; /* Q * growth^ts, mpc_dsr.py:349 */
            for (int r = 0; r < 3; ++r) Qs[r] = v->Q[r] * qg;
        } else {
            for (int r = 0; r < 3; ++r) Qs[r] = v->Qf[r];
        }
        int nn = 2 * ts;
        for (int r = 0; r < 3; ++r) {
            double e = p[r] - xref[r * W + ts];
            q->c0 += Qs[r] * e * e;
            const double *Gr = G + r * n;
            for (int i = 0; i < nn; ++i) {
                double gi = 2.0 * Qs[r] * Gr[i];
                q->g[i] += gi * e;
                for (int j = 0; j < nn; ++j) q->H[i * n + j] += gi * Gr[j];
            }
        }
    }
    return 0;
}

static int build_constraints(const double *lo, const double *hi, const double rr[2], int T,
                             int *ip, int *im, double *d)
{
    int n = 2 * T, m = 0;
    for (int i = 0; i < n; ++i) {
        if (isfinite(hi[i])) { ip[m] = i; im[m] = -1; d[m] = hi[i]; ++m; }
        if (isfinite(lo[i])) { ip[m] = -1; im[m] = i; d[m] = -lo[i]; ++m; }
    }
    for (int t = 0; t + 1 < T; ++t)
        for (int c = 0; c < 2; ++c)
            if (isfinite(rr[c])) {
                int i = 2 * t + c;
                ip[m] = i + 2; im[m] = i; d[m] = rr[c]; ++m;
                ip[m] = i; im[m] = i + 2; d[m] = rr[c]; ++m;
            }
    return m;
}

/* ---------------------------------------------------------------- IPM */

static double crow(const int *ip, const int *im, int j, const double *w)
{
    double a = 0.0;
    if (ip[j] >= 0) a += w[ip[j]];
    if (im[j] >= 0) a -= w[im[j]];
    return a;
}

static void ct_add(const int *ip, const int *im, int j, double val, double *out)
{
    if (ip[j] >= 0) out[ip[j]] += val;
    if (im[j] >= 0) out[im[j]] -= val;
}

static int cholesky(double *K, int n)
{
    for (int j = 0; j < n; ++j) {
        double s = K[j * n + j];
        for (int k = 0; k < j; ++k) s -= K[j * n + k] * K[j * n + k];
        if (!(s > 0.0)) return -1;
        double l = sqrt(s);
        K[j * n + j] = l;
        for (int i = j + 1; i < n; ++i) {
            double a = K[i * n + j];
            for (int k = 0; k < j; ++k) a -= K[i * n + k] * K[j * n + k];
            K[i * n + j] = a / l;
        }
    }
    return 0;
}

static void chol_solve(const double *L, int n, double *b)
{
    for (int i = 0; i < n; ++i) {
        double a = b[i];
        for (int k = 0; k < i; ++k) a -= L[i * n + k] * b[k];
        b[i] = a / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double a = b[i];
        for (int k = i + 1; k < n; ++k) a -= L[k * n + i] * b[k];
        b[i] = a / L[i * n + i];
    }
}

static double max_step(const double *v, const double *dv, int m)
{
    double a = 1.0;
    for (int j = 0; j < m; ++j)
        if (dv[j] < 0.0) { double r = -v[j] / dv[j]; if (r < a) a = r; }
    return a;
}

typedef struct { double erd, erp, mu; } resid_t;

static resid_t residuals(const qp_t *q, const int *ip, const int *im, int m, const double *w,
                         const double *s, const double *lam, double *rd, double *rp)
{
    int n = q->n;
    double sc = 1.0, e = 0.0;
    for (int i = 0; i < n; ++i) {
        double a = 0.0;
        for (int k = 0; k < n; ++k) a += q->H[i * n + k] * w[k];
        rd[i] = a;
        if (fabs(a) > sc) sc = fabs(a);
        if (fabs(q->g[i]) > sc) sc = fabs(q->g[i]);
    }
    /* C^T lam */
    double *cl = rp + m; /* scratch behind rp (caller sizes it) */
    for (int i = 0; i < n; ++i) cl[i] = 0.0;
    for (int j = 0; j < m; ++j) ct_add(ip, im, j, lam[j], cl);
    for (int i = 0; i < n; ++i) {
        if (fabs(cl[i]) > sc) sc = fabs(cl[i]);
        rd[i] += q->g[i] + cl[i];
        if (fabs(rd[i]) > e) e = fabs(rd[i]);
    }
    resid_t r;
    r.erd = e / sc;
    r.erp = 0.0;
    r.mu = 0.0;
    for (int j = 0; j < m; ++j) {
        rp[j] = crow(ip, im, j, w) + s[j] - q->d[j];
        if (fabs(rp[j]) > r.erp) r.erp = fabs(rp[j]);
        r.mu += s[j] * lam[j];
    }
    r.mu /= m > 0 ? m : 1;
    return r;
}

/* Dense LU with partial pivoting (row-major N x N, in place); returns -1 if singular. */
static int lu_factor(double *M, int *piv, int N)
{
    for (int c = 0; c < N; ++c) {
        int p = c;
        double best = fabs(M[c * N + c]);
        for (int r = c + 1; r < N; ++r)
            if (fabs(M[r * N + c]) > best) { best = fabs(M[r * N + c]); p = r; }
        if (!(best > 0.0)) return -1;
        piv[c] = p;
        if (p != c)
            for (int k = 0; k < N; ++k) { double t = M[c * N + k]; M[c * N + k] = M[p * N + k]; M[p * N + k] = t; }
        double inv = 1.0 / M[c * N + c];
        for (int r = c + 1; r < N; ++r) {
            double f = M[r * N + c] * inv;
            M[r * N + c] = f;
            if (f != 0.0)
                for (int k = c + 1; k < N; ++k) M[r * N + k] -= f * M[c * N + k];
        }
    }
    return 0;
}

static void lu_solve(const double *M, const int *piv, int N, double *b)
{
    for (int c = 0; c < N; ++c)   /* whole rows were swapped while factoring: P b first */
        if (piv[c] != c) { double t = b[c]; b[c] = b[piv[c]]; b[piv[c]] = t; }
    for (int c = 0; c < N; ++c)
        for (int r = c + 1; r < N; ++r) b[r] -= M[r * N + c] * b[c];
    for (int r = N - 1; r >= 0; --r) {
        double a = b[r];
        for (int k = r + 1; k < N; ++k) a -= M[r * N + k] * b[k];
        b[r] = a / M[r * N + r];
    }
}

/* Active-set polish of an interior-point answer: equality-constrained QP on the constraints
 * the iterate marks active (lam_j > s_j), regularised KKT system plus refinement against the
 * unregularised one; if the result violates a constraint left out, or gives a kept one a
 * negative multiplier, the set is fixed and the solve repeated (a few rounds of a primal-dual
 * active-set method from a very good guess).  The iterate is only ~sqrt(mu) accurate on
 * weakly active constraints; the polished point is exact to rounding.  Accepted only when
 * feasible, dual feasible, stationary to rounding and close to the iterate; otherwise the
 * iterate is kept. */
static void polish(const qp_t *q, const int *ip, const int *im, int m, double *w, const double *s,
                   double *lam)
{
    int n = q->n;
    int Nmax = n + m;
    int *active = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    int *act = (int *)malloc(sizeof(int) * (size_t)(m > 0 ? m : 1));
    double *M = (double *)malloc(sizeof(double) * (size_t)Nmax * Nmax);
    double *rhs = (double *)malloc(sizeof(double) * Nmax), *sol = (double *)malloc(sizeof(double) * Nmax);
    double *res = (double *)malloc(sizeof(double) * Nmax), *lamp = (double *)malloc(sizeof(double) * (m > 0 ? m : 1));
    int *piv = (int *)malloc(sizeof(int) * Nmax);
    const double delta = 1e-13;
    double scale = 1.0;
    for (int i = 0; i < n; ++i) {
        double a = 0.0;
        for (int j = 0; j < n; ++j) a += q->H[i * n + j] * w[j];
        if (fabs(a) + 1.0 > scale) scale = fabs(a) + 1.0;
        if (fabs(q->g[i]) + 1.0 > scale) scale = fabs(q->g[i]) + 1.0;
    }
    for (int j = 0; j < m; ++j) active[j] = lam[j] > s[j];
    for (int round = 0; round < 6; ++round) {
        int k = 0;
        for (int j = 0; j < m; ++j) if (active[j]) act[k++] = j;
        int N = n + k;
        memset(M, 0, sizeof(double) * (size_t)N * N);
        memset(sol, 0, sizeof(double) * N);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) M[i * N + j] = q->H[i * n + j];
        for (int a = 0; a < k; ++a) {
            int j = act[a];
            if (ip[j] >= 0) { M[(n + a) * N + ip[j]] = 1.0; M[ip[j] * N + n + a] = 1.0; }
            if (im[j] >= 0) { M[(n + a) * N + im[j]] = -1.0; M[im[j] * N + n + a] = -1.0; }
            M[(n + a) * N + n + a] = -delta;
        }
        for (int i = 0; i < n; ++i) rhs[i] = -q->g[i];
        for (int a = 0; a < k; ++a) rhs[n + a] = q->d[act[a]];
        if (lu_factor(M, piv, N) != 0) break;
        for (int pass = 0; pass < 4; ++pass) {
            /* res = rhs - KKT0 * sol (delta = 0) */
            for (int i = 0; i < n; ++i) {
                double a = rhs[i];
                for (int j = 0; j < n; ++j) a -= q->H[i * n + j] * sol[j];
                res[i] = a;
            }
            for (int a = 0; a < k; ++a) {
                int j = act[a];
                double nu = sol[n + a];
                if (ip[j] >= 0) res[ip[j]] -= nu;
                if (im[j] >= 0) res[im[j]] += nu;
                res[n + a] = rhs[n + a] - crow(ip, im, j, sol);
            }
            lu_solve(M, piv, N, res);
            for (int i = 0; i < N; ++i) sol[i] += res[i];
        }
        int finite = 1;
        for (int i = 0; i < N; ++i) if (!isfinite(sol[i])) finite = 0;
        if (!finite) break;
        double stat = 0.0, numin = 0.0, close = 0.0, viol = -1.0;
        double *hv = res;
        for (int i = 0; i < n; ++i) {
            double a = 0.0;
            for (int j = 0; j < n; ++j) a += q->H[i * n + j] * sol[j];
            hv[i] = a + q->g[i];
            if (fabs(sol[i] - w[i]) > close) close = fabs(sol[i] - w[i]);
        }
        for (int j = 0; j < m; ++j) lamp[j] = 0.0;
        for (int a = 0; a < k; ++a) {
            int j = act[a];
            double nu = sol[n + a];
            lamp[j] = nu;
            if (nu < numin) numin = nu;
            if (ip[j] >= 0) hv[ip[j]] += nu;
            if (im[j] >= 0) hv[im[j]] -= nu;
        }
        for (int i = 0; i < n; ++i) if (fabs(hv[i]) > stat) stat = fabs(hv[i]);
        for (int j = 0; j < m; ++j) { double g = crow(ip, im, j, sol) - q->d[j]; if (g > viol) viol = g; }
        int ok_stat = stat <= 1e-11 * scale, ok_close = close <= 1e-3;
        if (viol <= 1e-12 && numin >= -1e-9 * scale && ok_stat && ok_close) {
            for (int i = 0; i < n; ++i) w[i] = sol[i];
            for (int j = 0; j < m; ++j) lam[j] = lamp[j] > 0.0 ? lamp[j] : 0.0;
            break;
        }
        if (!ok_stat || !ok_close) break;
        for (int j = 0; j < m; ++j) {
            double g = crow(ip, im, j, sol) - q->d[j];
            active[j] = (active[j] && lamp[j] >= -1e-9 * scale) || g > 1e-12;
        }
    }
    free(active); free(act); free(M); free(rhs); free(sol); free(res); free(lamp); free(piv);
}

static int g_polish = 1; /* orc_set_polish(): tests keep it on, the timed CPU baseline turns it off */

static int ipm(const qp_t *q, const int *ip, const int *im, int m, double *w, double *lam_out,
               int *conv, double *kkt)
{
    const double tol = 1e-10;
    int n = q->n;
    double *K = q->work;
    double *s = K + (size_t)n * n, *lam = s + m, *ds = lam + m, *dl = ds + m;
    double *rd = dl + m, *dw = rd + n, *rhs = dw + n, *rp = rhs + n; /* rp: m + n */
    for (int i = 0; i < n; ++i) w[i] = 0.0;
    for (int j = 0; j < m; ++j) { s[j] = q->d[j] > 1.0 ? q->d[j] : 1.0; lam[j] = 1.0; }
    int it = 0, ok = 0;
    resid_t r;
    if (m == 0) {
        /* nothing to stay inside of: the minimiser is the Newton point (the numpy twin solves
           H w = -g directly too) */
        memcpy(K, q->H, sizeof(double) * n * n);
        ok = cholesky(K, n) == 0 ? 1 : -1;
        if (ok > 0) {
            for (int i = 0; i < n; ++i) w[i] = -q->g[i];
            chol_solve(K, n, w);
        }
        r = residuals(q, ip, im, 0, w, s, lam, rd, rp);
        *conv = ok;
        if (kkt) *kkt = r.erd;
        return 1;
    }
    for (it = 1; it <= 80; ++it) {
        r = residuals(q, ip, im, m, w, s, lam, rd, rp);
        if (r.erd <= tol && r.erp <= tol && r.mu <= 1e-12) { ok = 1; break; }
        if (r.mu < 1e-16) break;
        memcpy(K, q->H, sizeof(double) * n * n);
        for (int j = 0; j < m; ++j) {
            double D = lam[j] / s[j];
            int a = ip[j], b = im[j];
            if (a >= 0) K[a * n + a] += D;
            if (b >= 0) K[b * n + b] += D;
            if (a >= 0 && b >= 0) { K[a * n + b] -= D; K[b * n + a] -= D; }
        }
        if (cholesky(K, n) != 0) break;
        /* predictor: rc = s*lam */
        for (int i = 0; i < n; ++i) rhs[i] = -rd[i];
        for (int j = 0; j < m; ++j) ct_add(ip, im, j, -(-s[j] * lam[j] + lam[j] * rp[j]) / s[j], rhs);
        memcpy(dw, rhs, sizeof(double) * n);
        chol_solve(K, n, dw);
        for (int j = 0; j < m; ++j) {
            ds[j] = -rp[j] - crow(ip, im, j, dw);
            dl[j] = (-s[j] * lam[j] - lam[j] * ds[j]) / s[j];
        }
        double ap = max_step(s, ds, m), ad = max_step(lam, dl, m), mua = 0.0;
        for (int j = 0; j < m; ++j) mua += (s[j] + ap * ds[j]) * (lam[j] + ad * dl[j]);
        mua /= m;
        double sig = r.mu > 0.0 ? pow(mua / r.mu, 3.0) : 0.0;
        /* corrector: rc = s*lam - sig*mu + ds_aff*dl_aff (stored in dl) */
        for (int i = 0; i < n; ++i) rhs[i] = -rd[i];
        for (int j = 0; j < m; ++j) {
            double rc = s[j] * lam[j] - sig * r.mu + ds[j] * dl[j];
            dl[j] = rc;
            ct_add(ip, im, j, -(-rc + lam[j] * rp[j]) / s[j], rhs);
        }
        memcpy(dw, rhs, sizeof(double) * n);
        chol_solve(K, n, dw);
        for (int j = 0; j < m; ++j) {
            ds[j] = -rp[j] - crow(ip, im, j, dw);
            dl[j] = (-dl[j] - lam[j] * ds[j]) / s[j];
        }
        double eta = 1.0 - r.mu;
        if (eta < 0.99) eta = 0.99;
        if (eta > 0.9999) eta = 0.9999;
        ap = eta * max_step(s, ds, m); if (ap > 1.0) ap = 1.0;
        ad = eta * max_step(lam, dl, m); if (ad > 1.0) ad = 1.0;
        for (int i = 0; i < n; ++i) w[i] += ap * dw[i];
        for (int j = 0; j < m; ++j) { s[j] += ap * ds[j]; lam[j] += ad * dl[j]; }
    }
    if (!ok) {
        r = residuals(q, ip, im, m, w, s, lam, rd, rp);
        ok = (r.erd <= 1e-7 && r.erp <= 1e-7 && r.mu <= 1e-7) ? 0 : -1;
    }
    *conv = ok;
    if (ok >= 0 && g_polish && m > 0) {
        polish(q, ip, im, m, w, s, lam);
        r = residuals(q, ip, im, m, w, s, lam, rd, rp);
    }
    if (kkt) {
        /* certificate of what is returned: relative stationarity, violation, complementarity */
        double viol = 0.0, comp = 0.0;
        for (int j = 0; j < m; ++j) {
            double g = crow(ip, im, j, w) - q->d[j];
            if (g > viol) viol = g;
            if (fabs(g * lam[j]) > comp) comp = fabs(g * lam[j]);
        }
        double k = r.erd;
        if (viol > k) k = viol;
        if (comp > k) k = comp;
        *kkt = k;
    }
    if (lam_out) memcpy(lam_out, lam, sizeof(double) * m);
    return it;
}

/* ---------------------------------------------------------------- one instance */

static double objective(const orc_variant *v, const double *xref, const double *x, const double *u)
{
    int T = v->T, W = T + 1;
    double J = 0.0, qg = 1.0;
    for (int t = 0; t < T; ++t)
        for (int r = 0; r < v->nu; ++r) J += v->R[r] * u[r * T + t] * u[r * T + t];
    for (int t = 0; t + 1 < T; ++t)
        for (int r = 0; r < v->nu; ++r) {
            double dlt = u[r * T + t + 1] - u[r * T + t];
            J += v->Rd[r] * dlt * dlt;
        }
    for (int t = 1; t <= T; ++t) {
        if (t < T) qg *= v->q_growth;
        for (int r = 0; r < 3; ++r) {
            double e = xref[r * W + t] - x[r * W + t];
            J += (t < T ? v->Q[r] * qg : v->Qf[r]) * e * e;
        }
    }
    return J;
}

static int all_finite(const double *p, int n)
{
    for (int i = 0; i < n; ++i) if (!isfinite(p[i])) return 0;
    return 1;
}

static void fill_nan(double *p, int n) { for (int i = 0; i < n; ++i) p[i] = NAN; }

/* linear_mpc_control*: one QP at the given linearisation point. */
static int solve_one(const orc_variant *v, const double *xref, const double *xbar, const double *x0,
                     const double *uref, double *u_out, double *x_out, double *obj, int *iters,
                     double *kkt, double *mem)
{
    int T = v->T, W = T + 1, n = 2 * T;
    if (!all_finite(xref, 3 * W) || !all_finite(xbar, 3 * W) || !all_finite(x0, 3) ||
        !all_finite(uref, v->nu * T)) {
        fill_nan(u_out, v->nu * T); fill_nan(x_out, 3 * W);
        *obj = NAN; *iters = 0; if (kkt) *kkt = NAN;
        return ORC_ST_NONFINITE;
    }
    qp_t q;
    condense(v, xref, xbar, x0, uref, &q, mem);
    double lo[2 * ORC_MAX_T], hi[2 * ORC_MAX_T], rr[2];
    int ipb[8 * ORC_MAX_T], imb[8 * ORC_MAX_T];
    if (!free_bounds(v, uref, T, lo, hi, rr)) {
        fill_nan(u_out, v->nu * T); fill_nan(x_out, 3 * W);
        *obj = NAN; *iters = 0; if (kkt) *kkt = NAN;
        return ORC_ST_INFEASIBLE;
    }
    int m = build_constraints(lo, hi, rr, T, ipb, imb, q.d);
    q.m = m;
    double w[2 * ORC_MAX_T];
    int conv = 0;
    *iters = ipm(&q, ipb, imb, m, w, NULL, &conv, kkt);
    for (int t = 0; t < T; ++t)
        for (int r = 0; r < v->nu; ++r)
            u_out[r * T + t] = q.E[r][0] * w[2 * t] + q.E[r][1] * w[2 * t + 1] + q.f[r];
    double x[3] = {x0[0], x0[1], x0[2]};
    x_out[0] = x[0]; x_out[W] = x[1]; x_out[2 * W] = x[2];
    for (int t = 0; t < T; ++t) {
        const double *Bt = q.Bt + 6 * t, *ct = q.ct + 3 * t;
        double n0 = x[0] + q.a02[t] * x[2] + Bt[0] * w[2 * t] + Bt[1] * w[2 * t + 1] + ct[0];
        double n1 = x[1] + q.a12[t] * x[2] + Bt[2] * w[2 * t] + Bt[3] * w[2 * t + 1] + ct[1];
        double n2 = x[2] + Bt[4] * w[2 * t] + Bt[5] * w[2 * t + 1] + ct[2];
        x[0] = n0; x[1] = n1; x[2] = n2;
        x_out[t + 1] = n0; x_out[W + t + 1] = n1; x_out[2 * W + t + 1] = n2;
    }
    *obj = objective(v, xref, x_out, u_out);
    (void)n;
    return conv == 1 ? ORC_ST_OK : (conv == 0 ? ORC_ST_INACCURATE : ORC_ST_INACCURATE);
}

/* iterative_linear_mpc_control (mpc_diff_drive.py:142-167; mpc_dsr.py:169-199) and, with
 * chain > 1, iterativeLMPC's chained solves (mpc_sim.py:284-310). */
static int iterate_one(const orc_variant *v, const double *xref, const double *x0, const double *uprev,
                       int has_prev, double *u_out, double *x_out, double *obj, int *iters,
                       double *kkt, double *mem)
{
    int T = v->T, NU = v->nu;
    double uin[4 * ORC_MAX_T], xbar[3 * (ORC_MAX_T + 1)], uref[4 * ORC_MAX_T];
    for (int r = 0; r < NU; ++r)
        for (int t = 0; t < T; ++t) uin[r * T + t] = has_prev ? uprev[r * T + t] : v->cold_start[r];
    if (!all_finite(x0, 3) || !all_finite(uin, NU * T)) {
        fill_nan(u_out, NU * T); fill_nan(x_out, 3 * (T + 1));
        *obj = NAN; *iters = 0; if (kkt) *kkt = NAN;
        return ORC_ST_NONFINITE;
    }
    rollout(v, x0, uin, xbar);
    if (v->chain <= 1)
        return solve_one(v, xref, xbar, x0, uin, u_out, x_out, obj, iters, kkt, mem);
    memset(uref, 0, sizeof(double) * NU * T);
    int st = ORC_ST_OK, tot = 0, inaccurate = 0;
    const double *cur = uin;
    for (int i = 0; i < v->chain && i < T; ++i) {
        for (int r = 0; r < NU; ++r) uref[r * T + i] = cur[r * T + i];
        int it = 0;
        st = solve_one(v, xref, xbar, x0, uref, u_out, x_out, obj, &it, kkt, mem);
        tot += it;
        if (st != ORC_ST_OK && st != ORC_ST_INACCURATE) break;   /* the reference raises at ovf[i] on the next pass */
        if (st == ORC_ST_INACCURATE) inaccurate = 1;
        cur = u_out;
    }
    *iters = tot;
    /* an inaccurate link makes the whole chain's answer inaccurate, whatever the last solve says */
    return (st == ORC_ST_OK && inaccurate) ? ORC_ST_INACCURATE : st;
}

/* ---------------------------------------------------------------- batched entry points */

int orc_iterative_batch(const orc_variant *v, int64_t B, const double *x0, const double *xref,
                        const double *uprev, const uint8_t *has_prev, double *u_out, double *x_out,
                        double *obj_out, int32_t *status_out, int32_t *iters_out, double *kkt_out,
                        int nthreads)
{
    if (!v || v->T < 1 || v->T > ORC_MAX_T || (v->nu != 2 && v->nu != 4)) return -1;
    int T = v->T, W = T + 1, NU = v->nu;
    size_t per = qp_doubles(T);
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        double *mem = (double *)malloc(per * sizeof(double));
#pragma omp for schedule(static)
        for (int64_t b = 0; b < B; ++b) {
            int it = 0;
            double kk = 0.0, ob = 0.0;
            int st = iterate_one(v, xref + b * 3 * W, x0 + b * 3, uprev ? uprev + b * NU * T : NULL,
                                 (uprev && (!has_prev || has_prev[b])) ? 1 : 0,
                                 u_out + b * NU * T, x_out + b * 3 * W, &ob, &it, &kk, mem);
            obj_out[b] = ob;
            status_out[b] = st;
            if (iters_out) iters_out[b] = it;
            if (kkt_out) kkt_out[b] = kk;
        }
        free(mem);
    }
    return 0;
}

int orc_linear_mpc_batch(const orc_variant *v, int64_t B, const double *xref, const double *xbar,
                         const double *x0, const double *uref, double *u_out, double *x_out,
                         double *obj_out, int32_t *status_out, int32_t *iters_out, double *kkt_out,
                         int nthreads)
{
    if (!v || v->T < 1 || v->T > ORC_MAX_T || (v->nu != 2 && v->nu != 4)) return -1;
    int T = v->T, W = T + 1, NU = v->nu;
    size_t per = qp_doubles(T);
    if (nthreads < 1) nthreads = 1;
#pragma omp parallel num_threads(nthreads)
    {
        double *mem = (double *)malloc(per * sizeof(double));
#pragma omp for schedule(static)
        for (int64_t b = 0; b < B; ++b) {
            int it = 0;
            double kk = 0.0, ob = 0.0;
            int st = solve_one(v, xref + b * 3 * W, xbar + b * 3 * W, x0 + b * 3, uref + b * NU * T,
                               u_out + b * NU * T, x_out + b * 3 * W, &ob, &it, &kk, mem);
            obj_out[b] = ob;
            status_out[b] = st;
            if (iters_out) iters_out[b] = it;
            if (kkt_out) kkt_out[b] = kk;
        }
        free(mem);
    }
    return 0;
}

int orc_rollout_batch(const orc_variant *v, int64_t B, const double *x0, const double *u, double *xbar)
{
    if (!v || v->T < 1 || v->T > ORC_MAX_T) return -1;
    int T = v->T;
    for (int64_t b = 0; b < B; ++b) rollout(v, x0 + b * 3, u + b * v->nu * T, xbar + b * 3 * (T + 1));
    return 0;
}

/* theta[B], u[B][NU] -> A[B][3][3], Bm[B][3][NU], C[B][3] */
int orc_linearize_batch(const orc_variant *v, int64_t B, const double *theta, const double *u,
                        double *A, double *Bout, double *C)
{
    if (!v) return -1;
    for (int64_t b = 0; b < B; ++b) {
        double a02, a12, Bm[3][4], Cc[3];
        linearize(v, theta[b], u + b * v->nu, 1, &a02, &a12, Bm, Cc);
        double *Ab = A + b * 9;
        for (int i = 0; i < 9; ++i) Ab[i] = 0.0;
        Ab[0] = Ab[4] = Ab[8] = 1.0; Ab[2] = a02; Ab[5] = a12;
        for (int r = 0; r < 3; ++r) {
            for (int k = 0; k < v->nu; ++k) Bout[(b * 3 + r) * v->nu + k] = Bm[r][k];
            C[b * 3 + r] = Cc[r];
        }
    }
    return 0;
}

/* ---------------------------------------------------------------- nearest index / reference window */

static double wrap_pi(double a) /* mpc_diff_drive.py:66-73 */
{
    while (a > M_PI) a -= 2.0 * M_PI;
    while (a < -M_PI) a += 2.0 * M_PI;
    return a;
}

/* mpc_diff_drive.py:103-123: window [pind, pind+n_search) clipped to the course, first
 * minimum wins.  Returns -1 for the empty window (min([]) raises in the reference). */
static int nearest_index(double x, double y, const double *cx, const double *cy, const double *cyaw,
                         int len, int pind, int n_search, int sign, double *mind)
{
    int lo = pind < len ? pind : len, hi = pind + n_search < len ? pind + n_search : len;
    if (pind < 0 || hi <= lo) return -1;
    int bi = lo;
    double best = 0.0;
    for (int k = lo; k < hi; ++k) {
        double dx = x - cx[k], dy = y - cy[k];
        double dk = dx * dx + dy * dy;
        if (k == lo || dk < best) { best = dk; bi = k; }
    }
    double m = sqrt(best);
    if (sign) {
        double ang = wrap_pi(cyaw[bi] - atan2(cy[bi] - y, cx[bi] - x));
        if (ang < 0) m = -m;
    }
    if (mind) *mind = m;
    return bi;
}

/* Python 3 round(): half to even, then int() */
static long py_round(double x)
{
    double r = nearbyint(x); /* default rounding mode is to-nearest-even */
    return (long)r;
}

/* state[B][4] = x, y, yaw, vx.  Courses: tables cx/cy/cyaw [K][Lmax], clen[K].
 * mode 0: calc_ref_trajectory  (mpc_diff_drive.py:241-269; param = dl)
 * mode 1: calc_ref_trajectory2 (mpc_diff_drive.py:271-296; param = n_search_ind)
 * ind_out[b] = -1 and NaN xref on the empty window. */
int orc_nearest_ref_batch(int mode, int T, double dt, double param, int n_search, int sign, int64_t B,
                          const double *state, const int32_t *course_id, const int32_t *pind,
                          const double *cx, const double *cy, const double *cyaw, const int32_t *clen,
                          int64_t Lmax, int32_t *ind_out, double *mind_out, double *xref_out)
{
    int W = T + 1;
    for (int64_t b = 0; b < B; ++b) {
        int k = course_id ? course_id[b] : 0;
        const double *x_ = cx + k * Lmax, *y_ = cy + k * Lmax, *w_ = cyaw + k * Lmax;
        int len = clen[k];
        double md = 0.0;
        int ind = nearest_index(state[4 * b], state[4 * b + 1], x_, y_, w_, len, pind[b], n_search, sign, &md);
        double *xr = xref_out ? xref_out + b * 3 * W : NULL;
        if (ind < 0) {
            ind_out[b] = -1;
            if (mind_out) mind_out[b] = NAN;
            if (xr) fill_nan(xr, 3 * W);
            continue;
        }
        if (mind_out) mind_out[b] = md;
        if (pind[b] >= ind) ind = pind[b];
        ind_out[b] = ind;
        if (!xr) continue;
        if (mode == 0) {
            double travel = 0.0;
            for (int i = 0; i < W; ++i) {
                travel += fabs(state[4 * b + 3]) * dt;
                long dind = py_round(travel / param);
                long kk = ind + dind < len ? ind + dind : len - 1;
                xr[i] = x_[kk]; xr[W + i] = y_[kk]; xr[2 * W + i] = w_[kk];
            }
        } else {
            int cnt = (int)param;
            if (ind + cnt > len) cnt = len - ind;
            for (int i = 0; i < W; ++i) {
                int kk = ind + (i < cnt ? i : cnt - 1);
                xr[i] = x_[kk]; xr[W + i] = y_[kk]; xr[2 * W + i] = w_[kk];
            }
        }
    }
    return 0;
}

int orc_variant_size(void) { return (int)sizeof(orc_variant); }

void orc_set_polish(int on) { g_polish = on; }
