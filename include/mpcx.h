/*
 * mpcx -- batched iterative linear MPC for AMD MI355X (gfx950).
 *
 * C ABI of libmpcx.so.  The reference project (pure Python) has no FFI of its
 * own: its boundary is a handful of Python functions called once per robot per
 * control tick.  Each entry point below is the batched form of one of them and
 * cites the function it stands in for; the Python modules in
 * model_predictive_control_amd/ keep the reference's names and signatures and
 * forward here.
 *
 * Conventions
 *  - every data pointer is a DEVICE pointer (HIP), row-major, one record per
 *    instance:  x0 [B][3];  xref / xbar / x_out [B][3][T+1];  controls
 *    [B][NU][T] with dual-steer rows v_f, delta_f, v_r, delta_r
 *    (mpc_dsr.py:186-190, mpc.py:99-102);
 *  - `stream` is a hipStream_t (NULL = the default stream); calls are
 *    asynchronous on it and return after the launches are queued;
 *  - return value 0 = ok, negative = error (MPCX_E_*); mpcx_last_error() gives
 *    the message for the calling thread; no exception crosses this boundary;
 *  - per-instance outcome goes to status_out (MPCX_ST_*).  The reference
 *    signals failure by printing "Error: Cannot solve mpc.." and returning
 *    Nones (mpc_diff_drive.py:209-215); the Python layer maps any status other
 *    than OK / INACCURATE to that.
 */
#ifndef MPCX_H
#define MPCX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPCX_ABI_VERSION 2

#define MPCX_MODEL_DIFF_DRIVE 0 /* unicycle, NU = 2 (mpc_diff_drive.py:19-21) */
#define MPCX_MODEL_DSR 1        /* dual-steer "general bicycle", NU = 4 (mpc_dsr.py:20-22) */

/* equalities between control rows -> two free controls per stage */
#define MPCX_EQ_NONE 0        /* diff-drive                                           */
#define MPCX_EQ_ACKERMANN 1   /* u0 = u2, u1 = -u3        (mpc_dsr.py:242-243)        */
#define MPCX_EQ_CRAB 2        /* u0 = u2, u1 =  u3        (mpc_dsr.py:310-311)        */
#define MPCX_EQ_FIXED_STEER 3 /* u1 = u3 = steer_fixed    (mpc_dsr.py:369-370)        */

#define MPCX_ST_OK 0          /* solved to tolerance                                   */
#define MPCX_ST_INACCURATE 1  /* solved to the loose tolerance only (the reference also
                                 accepts OPTIMAL_INACCURATE, mpc_diff_drive.py:202)    */
#define MPCX_ST_INFEASIBLE 2  /* empty constraint set (first-step window vs box)       */
#define MPCX_ST_NONFINITE 3   /* NaN/Inf in the inputs                                 */
#define MPCX_ST_FAILED 4      /* iteration limit without reaching the loose tolerance  */

#define MPCX_E_ARG (-1)
#define MPCX_E_UNSUPPORTED (-2)
#define MPCX_E_HIP (-3)
#define MPCX_E_NOMEM (-4)

#define MPCX_MAX_T 64

/* One row of the variant table: everything that distinguishes the reference's QPs
 * (mpc_diff_drive.py:170-217; mpc_dsr.py:201-395; mpc_sim.py:312-508; mpc.py:117-187). */
typedef struct mpcx_variant {
    int32_t model;      /* MPCX_MODEL_*                                                   */
    int32_t nu;         /* 2 or 4                                                         */
    int32_t T;          /* horizon (1..MPCX_MAX_T)                                        */
    int32_t eq_mode;    /* MPCX_EQ_*                                                      */
    int32_t chain;      /* >1: mpc_sim.py:295-308's chained solves per call               */
    int32_t pad_;
    double dt;          /* time tick                                                      */
    double wb;          /* wheel base                                                     */
    double R[4];        /* input cost diagonal                                            */
    double Rd[4];       /* input-difference cost diagonal (mpc.py:152-153), else 0        */
    double Q[3];        /* state cost diagonal, stages 1..T-1                             */
    double q_growth;    /* stage t uses Q * q_growth^t (mpc_dsr.py:349), else 1           */
    double Qf[3];       /* terminal state cost diagonal                                   */
    double umax[4];     /* |u_i| <= umax_i            (+inf = none)                       */
    double dumax[4];    /* |u_i[t+1]-u_i[t]| <= ...   (+inf = none)                       */
    double first_step[4]; /* |u_i[0]-uref_i[0]| <= ... (+inf = none)                      */
    double steer_fixed; /* MPCX_EQ_FIXED_STEER only                                       */
    double cold_start[4]; /* controls used when there is no previous solution            */
} mpcx_variant;

int mpcx_abi_version(void);
const char *mpcx_last_error(void);

/* Fills *out with one of the reference's variants: "diff_drive", "dsr_ackermann",
 * "dsr_crab", "dsr_diff", "sim_ackermann", "sim_diff", "sim_crab", "mpc_py". */
int mpcx_variant_by_name(const char *name, mpcx_variant *out);

/* Bytes of device workspace mpcx_solve_batch / mpcx_linear_mpc_batch need for B
 * instances of this variant: 0 for the kernels that live in registers and LDS (T = 5 and
 * T = 10 at every batch size), otherwise the per-lane arrays of the stage-wise kernel for one
 * launch (440 T bytes per instance, at most 262,144 instances per launch).  Small batches at other
 * horizons up to T = 32 are solved by a kernel that needs none, but are followed by a stage-wise
 * pass that solves again whatever that kernel reports as MPCX_ST_INACCURATE or MPCX_ST_FAILED (rare; the pass
 * returns at once otherwise, and is left out when uprev and u_out are the same buffer, since it
 * starts from the same inputs), so they count as needing the workspace too.  The library keeps one grow-only buffer per device for
 * it; calls that need it are serialised per device (each waits on the device for the
 * previous one, whatever its stream), the first one allocates, and none of them may be
 * part of a stream capture.  The answer is for mpcx_solve_batch; for mpcx_linear_mpc_batch,
 * which never chains, it is an upper bound when v->chain > 1. */
int64_t mpcx_workspace_bytes(const mpcx_variant *v, int64_t B);

/* iterative_linear_mpc_control (mpc_diff_drive.py:142-167, mpc_dsr.py:169-199),
 * MPC.iterativeLMPC (mpc_sim.py:284-310) and the solve part of
 * ModelPredictiveControl.start (mpc.py:83-102): cold start if no previous controls,
 * rollout, T linearisations, one QP (chain QPs when v->chain > 1).
 *   uprev may be NULL (all cold); has_prev may be NULL (all warm when uprev != NULL).
 *   iters_out may be NULL. */
int mpcx_solve_batch(const mpcx_variant *v, int64_t B, const double *x0, const double *xref,
                     const double *uprev, const uint8_t *has_prev, double *u_out, double *x_out,
                     double *obj_out, int32_t *status_out, int32_t *iters_out, void *stream);

/* mpcx_solve_batch over the instances range[0] .. range[1]-1 only, the two bounds read from
 * DEVICE memory when the kernel runs (the rest of the batch is neither read nor written): for a
 * fleet sorted by motion mode on the device, where MPC.iterativeLMPC's dispatch
 * `if mode == 'ackermann' ... elif mode == 'diff' ... elif mode == 'crab'` (mpc_sim.py:301-306)
 * becomes one call per mode over its run of the sorted batch, without the host ever learning
 * where the runs begin.  T = 5 with every bound present (the lane-pair kernel); MPCX_E_UNSUPPORTED
 * otherwise. */
int mpcx_solve_range_batch(const mpcx_variant *v, int64_t B, const int32_t *range, const double *x0, const double *xref,
                           const double *uprev, const uint8_t *has_prev, double *u_out, double *x_out,
                           double *obj_out, int32_t *status_out, int32_t *iters_out, void *stream);

/* The same dispatch for a whole fleet in one launch: the batch is sorted by mode, run m (instances
 * bounds[m] .. bounds[m+1]-1, the n_modes + 1 bounds read from DEVICE memory when the kernel runs) is
 * solved with variants[m]; instances outside every run are neither read nor written.  Every run gets
 * wavefronts of its own, so one launch fills the chip where one mpcx_solve_range_batch call per mode
 * fills a part of it after the other.  variants: host array (validated here); variants_dev: the same
 * array in device memory (uploaded once by the caller; the kernel reads each run's variant from it).
 * The modes share model, horizon and chain length; T = 5 with every bound present and the
 * dual-steer model (every mpc_sim mode), MPCX_E_UNSUPPORTED otherwise. */
int mpcx_solve_modes_batch(const mpcx_variant *variants, int32_t n_modes, const mpcx_variant *variants_dev,
                           const int32_t *bounds, int64_t B, const double *x0, const double *xref,
                           const double *uprev, const uint8_t *has_prev, double *u_out, double *x_out,
                           double *obj_out, int32_t *status_out, int32_t *iters_out, void *stream);

/* The fleet side of that dispatch, on the device: a stable counting sort of B vehicles by the mode
 * of their current course-table row (seg_mode [n_courses], values 0..n_modes-1; a vehicle with
 * finished != 0 or done != 0 is at rest and sorts last), and the solve's inputs gathered in that order.
 *   was_active [B] (1 = the vehicle takes part in this tick);  order [B]: vehicle of every sorted row;
 *   bounds [n_modes + 2]: start of every mode's run, start of the vehicles at rest, B (the first
 *   n_modes + 1 entries are what mpcx_solve_modes_batch takes);  x0 / xref / uprev (any may be NULL)
 *   -> *_sorted, has_prev (NULL = all 1) -> has_prev_sorted;  scratch: mpcx_mode_sort_scratch_ints()
 *   int32 of device memory (MPCX_E_UNSUPPORTED beyond about 3 million vehicles in 3 modes; at most 7
 *   modes).  mpcx_mode_unsort_batch carries the controls and statuses back: row p of the sorted
 *   arrays to vehicle order[p]. */
int64_t mpcx_mode_sort_scratch_ints(int64_t B, int32_t n_modes);
int mpcx_mode_sort_batch(int64_t B, int32_t n_modes, int32_t nu, int32_t T, const uint8_t *finished, const uint8_t *done,
                         const int32_t *course_id, const int32_t *seg_mode, int64_t n_courses, const double *x0,
                         const double *xref, const double *uprev, const uint8_t *has_prev, uint8_t *was_active,
                         int32_t *order, int32_t *bounds, double *x0_sorted, double *xref_sorted, double *uprev_sorted,
                         uint8_t *has_prev_sorted, int32_t *scratch, void *stream);
int mpcx_mode_unsort_batch(int64_t B, int32_t nu, int32_t T, const int32_t *order, const double *u_sorted,
                           const int32_t *status_sorted, double *u, int32_t *status, void *stream);

/* No counterpart in the reference: mpcx_solve_batch for callers that cannot allocate device
 * memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
 * Same arguments, same kernels, same results; the call itself allocates and frees nothing,
 * creates no event and never synchronises.  Where the solve needs a workspace (see
 * mpcx_workspace_bytes) it uses a fixed array of mpcx_static_workspace_bytes() bytes (64 MiB)
 * that lives in the library's device image: every GPU on which the library's code object is
 * loaded holds it until the process ends, whether or not this entry point is ever called.  A
 * batch whose workspace (440 T bytes per instance, B rounded up to a multiple of 64) exceeds it
 * is rejected with MPCX_E_NOMEM and nothing is launched.
 * Set-up, once per device and outside any stream capture: call mpcx_static_workspace_bytes()
 * with that device current (it looks up the device's copy of the array, which loads the code
 * object there if need be), then make one call of this entry point of the kind that will be
 * captured or timed: the first dispatch of a kernel on a queue lets the HIP runtime set up the
 * kernel's private-memory segment (the stage-wise kernel keeps 216 bytes per lane there), which
 * is the runtime's allocation, not this library's.  After that warm-up call the device's free
 * memory does not change over further calls, and the call can be recorded into a HIP graph.
 * Every call on a GPU shares that GPU's array: calls that overlap on one GPU (different
 * streams, or different host threads) race; calls on one stream are safe. */
int mpcx_solve_batch_noalloc(const mpcx_variant *v, int64_t B, const double *x0, const double *xref,
                             const double *uprev, const uint8_t *has_prev, double *u_out, double *x_out,
                             double *obj_out, int32_t *status_out, int32_t *iters_out, void *stream);
int64_t mpcx_static_workspace_bytes(void);

/* linear_mpc_control (mpc_diff_drive.py:170-217), linear_mpc_control_{ackermann,crab,diff}
 * (mpc_dsr.py:201-395), MPC.doLMPC_* (mpc_sim.py:312-508), controlLaw (mpc.py:117-187):
 * one QP at the given linearisation point (xbar, uref). */
int mpcx_linear_mpc_batch(const mpcx_variant *v, int64_t B, const double *xref, const double *xbar,
                          const double *x0, const double *uref, double *u_out, double *x_out,
                          double *obj_out, int32_t *status_out, int32_t *iters_out, void *stream);

/* predict_motion (mpc_diff_drive.py:126-139, mpc_dsr.py:125-139, mpc_sim.py:268-282,
 * mpc.py:385-421): xbar [B][3][T+1] from x0 and the control rows u [B][NU][T]. */
int mpcx_rollout_batch(const mpcx_variant *v, int64_t B, const double *x0, const double *u,
                       double *xbar, void *stream);

/* get_linear_model_matrix (mpc_diff_drive.py:220-238, mpc_dsr.py:397-426),
 * getRobotModelMatrice (mpc_sim.py:132-163, mpc.py:354-383):
 * theta [B], u [B][NU] -> A [B][3][3], Bm [B][3][NU], C [B][3]. */
int mpcx_linearize_batch(const mpcx_variant *v, int64_t B, const double *theta, const double *u,
                         double *A, double *Bm, double *C, void *stream);

/* calc_nearest_index (mpc_diff_drive.py:103-123, mpc_sim.py:510-518) followed by
 * calc_ref_trajectory (mode 0, param = dl; mpc_diff_drive.py:241-269) or
 * calc_ref_trajectory2 / getReferenceTrajectory (mode 1, param = n_search_ind;
 * mpc_diff_drive.py:271-296, mpc_sim.py:520-539).
 *   state [B][4] = x, y, yaw, vx;  course_id [B] (NULL = course 0);  pind [B];
 *   course tables cx, cy, cyaw [K][Lmax] with clen [K] valid points each;
 *   ind_out [B] (-1 = empty search window, where the reference raises; also for a course_id
 *   outside 0..n_courses-1 and for a course whose clen is not 1..Lmax); ind_out may be the
 *   same buffer as pind (each instance reads its pind before its ind is written);
 *   mind_out [B] or NULL (signed as in mpc_diff_drive.py:116-121 when sign != 0);
 *   xref_out [B][3][T+1] or NULL;
 *   x0_out [B][3] or NULL: x0 = [state.x, state.y, state.yaw] (mpc_diff_drive.py:363), the
 *   first argument of the solve that follows, so that a closed loop needs no copy of its own. */
int mpcx_nearest_ref_batch(int32_t mode, int32_t T, double dt, double param, int32_t n_search,
                           int32_t sign, int64_t B, const double *state, const int32_t *course_id,
                           const int32_t *pind, const double *cx, const double *cy,
                           const double *cyaw, const int32_t *clen, int64_t n_courses, int64_t Lmax,
                           int32_t *ind_out, double *mind_out, double *xref_out, double *x0_out,
                           void *stream);

/* What do_simulation does between two solves (mpc_diff_drive.py:363-383, mpc_dsr.py:548-571,
 * mpc_sim.py:224-247): update_state with the first control column when the solve succeeded
 * (mpc_diff_drive.py:92-100; mpc_dsr.py:112-121 after transfrom_gbm_command, :123-150),
 * warm-start carry (has_prev = 1) or cold start after a failure (has_prev = 0), check_goal
 * (mpc_diff_drive.py:299-316) against the last point of the instance's course.
 *   u [B][NU][T], status [B]: outputs of mpcx_solve_batch;  target_ind [B]: ind_out of
 *   mpcx_nearest_ref_batch for this tick;  state [B][4] = x, y, yaw, vx (updated in place);
 *   twist [B][2] = vy, w of the applied command (or NULL);  done [B]: 0 running, 1 goal
 *   reached, 2 reference window lost, course_id outside the table or course length not
 *   1..Lmax (frozen from then on);  ticks [B]: ticks executed;
 *   log_out [B][4] or NULL: the state after this tick. */
int mpcx_advance_batch(const mpcx_variant *v, int64_t B, double goal_dis, double stop_speed, const double *u,
                       const int32_t *status, const int32_t *target_ind, const int32_t *course_id,
                       const double *cx, const double *cy, const int32_t *clen, int64_t n_courses, int64_t Lmax,
                       double *state, double *twist, uint8_t *has_prev, uint8_t *done, int32_t *ticks,
                       double *log_out, void *stream);

/* What MPC.doSimulationWithAllMotionModes (mpc_sim.py:254-265) does between two ticks for vehicles
 * whose course is a list of segments (rows seg_start[b] .. seg_start[b] + seg_count[b] - 1 of the
 * course tables), after mpcx_advance_batch: for the vehicles with was_active[b] != 0 (done == 0 and
 * not finished when the tick began) the tick is counted against the segment (seg_ticks); done == 2
 * (reference window lost) finishes the vehicle; done == 1 (goal of the segment) or seg_ticks ==
 * max_ticks_per_segment moves it to its next segment, or finishes it (done = 1) after the last.
 * Entering a segment is doSimulation's prologue (mpc_sim.py:204-222): course_id = the new row, yaw
 * moved by 2 pi towards the row's first yaw, has_prev = 0, done = 0, seg_ticks = 0, and enter[b] = 1
 * with pind_next[b] = 0 so that the caller's next mpcx_nearest_ref_batch call (pind = pind_next)
 * searches from the segment's start; for every other vehicle enter[b] = 0, pind_next[b] = target_ind[b]. */
int mpcx_segment_step_batch(int64_t B, int32_t max_ticks_per_segment, const uint8_t *was_active, const int32_t *seg_start,
                            const int32_t *seg_count, int32_t *seg, int32_t *seg_ticks, uint8_t *finished, uint8_t *done,
                            uint8_t *has_prev, int32_t *course_id, double *state, const double *cyaw, int64_t n_courses,
                            int64_t Lmax, const int32_t *target_ind, int32_t *pind_next, uint8_t *enter, void *stream);

/* calc_spline_course (cubic_spline_planner.py:311-323: CubicSpline2D :174-308 over two
 * CubicSpline1D :12-171) for K courses at once, optionally followed by smooth_yaw
 * (mpc_diff_drive.py:404-417): the producer of the course tables mpcx_nearest_ref_batch reads.
 *   wx, wy [K][n_way_max] waypoints, nway [K] of them valid (2..64);
 *   cx, cy, cyaw [K][Lmax] and ck [K][Lmax] (curvature, or NULL): samples at s = 0, ds, ...
 *   below the total chord length;  clen [K] their number, -1 for a course that does not fit
 *   Lmax or has repeated / non-finite waypoints (where the reference divides by zero). */
int mpcx_build_courses_batch(int64_t K, int32_t n_way_max, const double *wx, const double *wy, const int32_t *nway,
                             double ds, int64_t Lmax, int32_t smooth, double *cx, double *cy, double *cyaw,
                             double *ck, int32_t *clen, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MPCX_H */
