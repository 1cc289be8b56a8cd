/*
 * cvxpnpl_amd.h -- C ABI of the MI355X batched absolute-pose SDP solver.
 *
 * This is the drop-in boundary for the hot path of SergioRAgostinho/cvxpnpl: everything
 * cvxpnpl.pnp / pnl / pnpl do between receiving the correspondences and returning poses
 * (reference cvxpnpl.py:523-627), i.e. constraint assembly (:20-153), translation
 * elimination (:545-549), the SDP solve the reference delegates to scs.solve (:485-489)
 * and the pose recovery (:492-520) -- for a whole batch of independent problems in one
 * launch.  Plain pointers and sizes only; no torch / numpy types.  All device pointers
 * are HIP device memory on the current device; arrays are contiguous, problem-major,
 * float64 (the reference computes in float64 throughout).
 *
 * The reference has no FFI of its own for this path (it is pure Python calling the scs
 * C extension); the closest interface is _solve_relaxation(A, B, eps, max_iters, verbose)
 * (cvxpnpl.py:454-460) and the three public functions.  INTEGRATION.md shows the ctypes
 * binding a maintainer of the reference would add.
 */
#ifndef CVXPNPL_AMD_H
#define CVXPNPL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* per-problem outcome, written to status[] (reference behaviour in brackets) */
enum {
    CVXPNPL_CERTIFIED = 0,   /* rank-1 pose, certified globally optimal: 0 <= cost - dobj <= eps [no warning] */
    CVXPNPL_RANK_GT1 = 1,    /* rank(Z) > 1 at the 1e-3 threshold of cvxpnpl.py:502: Z is returned, poses come
                                from cvxpnpl_recover_multi (cvxpnpl.py:507, 221-343).  R, t hold ONE of the poses
                                when the pair was certified (exact two-fold ambiguity, e.g. a planar scene),
                                otherwise the better of the two rank-2 candidates of the top-2 eigenspace of Z (what cvxpnpl.py:303-315
                                computes for a rank-2 Z): finite whenever Z is; ask for Z to get all the poses */
    CVXPNPL_UNCERTIFIED = 2, /* rank-1 pose, no certificate by max_iters ["not certifiably optimal", :517-519] */
    CVXPNPL_NONFINITE = 3,   /* degenerate input: NaN pose [NaN sentinel :493-498 / LinAlgError] */
    CVXPNPL_REFLECTION = 4   /* uncertified and det(U V^T) < 0; returned as is, like the reference (:510-511) */
};

/* kernel layouts (A/B switch; all produce the same results).  Any other value of opts.layout is refused ("bad options", -1). */
enum {
    CVXPNPL_LAYOUT_AUTO = 0, /* by launch size: wave below 2560 problems, quad below 20000, lane (hybrid) from there; four-correspondence
                                problems: quad from 2560 on, with a 24-iteration first phase and the first attempt after 7 */
    CVXPNPL_LAYOUT_LANE = 1, /* one problem per lane, 64 per wavefront, for the first lane_iters iterations;
                                unfinished problems are then resumed one per wavefront (hybrid schedule) */
    CVXPNPL_LAYOUT_WAVE = 2, /* one problem per wavefront (cooperative lanes) */
    CVXPNPL_LAYOUT_QUAD = 3, /* one problem per DPP row: 16 lanes, four per wavefront, for the first lane_iters
                                iterations; the wavefront then finishes its unfinished ones itself, one at a time */
    CVXPNPL_LAYOUT_PENTA = 4 /* the same schedule with 12 lanes per problem, five per wavefront */
};

/* constraint sets of the relaxation */
enum {
    CVXPNPL_VARIANT_FULL = 0, /* the 22 equalities of cvxpnpl.py:387-451 (pnp / pnl / pnpl) */
    CVXPNPL_VARIANT_RC = 1    /* the 16 equalities of benchmarks/toolkit/methods/rc.py:9-64 (the reference's ablation "rc":
                                 the six row-orthonormality rows are left out).  Layouts: wave-per-problem below 2 560 problems, from
                                 there the quad schedule (four problems per wavefront, 36 iterations) whose survivors are finished one per
                                 wavefront, with the interior-point path on the 16 rows behind it (opts.rescue_from: 48); the lane schedule
                                 is built for the full set only and a LANE / PENTA request runs the quad schedule */
};

typedef struct {
    uint32_t struct_size; /* sizeof(cvxpnpl_opts_t) of the header the CALLER was built against: cvxpnpl_default_opts fills it in, every
                             entry point that takes options rejects a block whose size is not this library's (return -1,
                             cvxpnpl_last_error names both sizes) instead of reading fields that are not there.  cvxpnpl_opts_size()
                             returns the library's value for bindings that mirror the struct by hand (ctypes, cgo, JNI). */
    double eps;        /* absolute duality-gap tolerance; reference `eps` (cvxpnpl.py:527), default 1e-9 */
    int32_t max_iters; /* iteration cap; reference `max_iters` (cvxpnpl.py:528), default 2500 */
    double rho;        /* ADMM penalty on the trace-normalised cost, default 0.1 */
    double alpha;      /* over-relaxation, default 1.4 */
    int32_t first_check; /* first certification attempt after this many iterations; 0 (default): 5, or 6 in the lane-hybrid layout; the rc variant 11 (19 in the quad schedule);
                           launches of >= 2 560 four-correspondence problems 17 (their first phase queues its survivors, an attempt there costs the
                           whole wavefront two to three iterations; profiles/r04/minimal_tune*.txt) */
    int32_t check_every; /* then every this many (widening ~sqrt(iteration) from iteration 10 on), default 2 */
    double res_tol;    /* fixed-point residual at which an uncertifiable problem stops, default 1e-5 */
    int32_t jacobi_sweeps; /* cap on Jacobi sweeps per PSD projection, default 12 */
    double jacobi_tol; /* eigen-solve ends after a sweep whose largest column cosine is below this, default 6e-2 */
    int32_t warm_start; /* 1 (default): each eigen-solve starts from the previous iteration's eigenvectors */
    double rho_tail;    /* penalty from iteration tail_from on (dual rescaled at the switch), default 0.05 */
    int32_t tail_from;  /* default 3; <= 0 never */
    int32_t lane_iters; /* lane and quad layouts: iterations before unfinished problems are handed to one
                           wavefront each (hybrid schedule).  <= 0: default (lane: first_check, i.e. right after the
                           first attempt; quad: 7).  The lane phase is capped at 6 iterations, the quad phase at 16. */
    int32_t layout;    /* CVXPNPL_LAYOUT_* */
    int32_t variant;   /* CVXPNPL_VARIANT_*, default FULL */
    int32_t adapt_every; /* residual balancing of the penalty for long solves: every this many iterations (default 10; 0 never) */
    int32_t adapt_from;  /* ... from this iteration on (default 40: below that the well-posed problems finish on their own and an
                            adaptation only delays the odd one -- measured: slowest of 125 k problems 41 -> 61 iterations at 20) */
    double adapt_mu;     /* a primal / dual residual larger than the other by this factor (default 2) moves the penalty ... */
    double adapt_tau;    /* ... by this factor (default 2), within [1e-3, 10] */
    int32_t stall_from;  /* from this iteration on (default 300; 0 never) a solve whose Z has settled at rank > 1 -- second eigenvalue
                            above stall_lam (0.05), no longer shrinking (by less than stall_drop = 0.3 % between two certificate
                            attempts), fixed-point residual below stall_res (1e-3) -- stops as CVXPNPL_RANK_GT1: the relaxation
                            is not tight and the first-order iteration would crawl to max_iters (the reference's solve does) */
    double stall_lam, stall_res, stall_drop;
    int32_t rescue_from; /* a problem still open after this many first-order iterations is finished by the interior-point path
                            (0 never; -1, the default: 32 for problems with at most 6 correspondences, where slow convergence is
                            common, 64 for 7, 128 otherwise; full variant only): ~12 second-order iterations whatever the conditioning, then
                            the first-order iteration goes on from the interior-point solution -- same rounding, polish, certificate
                            and recovery -- so that a launch no longer waits for a 1 000-iteration straggler (minimal and
                            near-ambiguous configurations; measured on 50 k four-point RANSAC hypotheses: 7.1 -> 2.9 ms).
                            Costs one more (mostly idle) kernel launch per solve in the wave and lane layouts. */
    int32_t f32_sweeps_until; /* The reference computes in float64 throughout (numpy defaults, cvxpnpl.py:475-513).  So does this library
                            -- inputs, Gram sums, iterate, polish, dual certificate, outputs -- with ONE exception: during the first
                            iterations of a solve the Jacobi sweeps of the PSD projection (and the product (W + sigma I) V that starts them)
                            run on single-precision columns; a pose is only ever reported CERTIFIED after the float64 chain polish -> dual ->
                            LDL^T -> gap has succeeded on it.  This field bounds the exception: sweeps are single precision while the
                            iteration count of the solve is below it.  -1 (default): 64.  0: never -- every sweep, rotation angles
                            included, in float64 (the A/B mode: `value_all_f64` of bench.py, tests/test_precision_modes.py).  The quad
                            and lane phases (at most 16 / 6 iterations; four-correspondence problems 24, the rc variant up to 48) run entirely in one precision: single only if the whole phase lies
                            below the bound, float64 otherwise.  Valid: -1 ... 64 -- the window the experiments cover (host experiment of DESIGN.md section 1.2:
                            identical iteration histograms and certified counts up to 64, longer tails from 128 on; device A/B of the 24- and
                            48-iteration single-precision phases of four-point problems and the rc variant: profiles/r04/f32_phase_ab.txt);
                            larger values are rejected ("bad options"). */
    int32_t sweep_schedule; /* 1 (default): in the first phases of the quad and lane schedules -- where a wavefront runs the maximum number of
                            Jacobi sweeps over its 4 / 64 problems -- the sweeps of an eigen-solve are capped by iteration: 3 for the
                            first one (iteration 2), then 2 (lane phase: 1 from iteration 5 on); jacobi_sweeps still bounds everything.
                            A column pair that misses its last sweep is caught by the next iteration's warm start; what certifies is
                            unchanged to ~0.1 % of the first attempts (cvx::sweep_cap has the measurements), results are not affected.
                            0: only jacobi_sweeps. */
    double dual_shift;      /* -1 (default): 0.015, rc variant 0.006.  A certificate attempt whose recovered dual S fails the PSD test while the pose is fine gets second
                            tries with S + s D(R), s = dual_shift, dual_shift / 4, D(R) the projection of the identity onto the family of
                            duals that are complementary to the pose (a constant in the frame of R: cvx::dual_retry_entry6) -- one more
                            LDL^T each, no fit.  Made for the problems a quad or lane phase has handed over to the wave-per-problem phase
                            -- the slow ones that end a launch -- (the scalar core: from the second attempt of a solve on); the quad
                            and lane phases themselves, and launches small enough to run wave-per-problem from the start, make one try.
                            About half of those failed attempts pass (N = 10; fewer iterations for the stragglers: 125 k problems
                            +6 %, four- and six-point problems +8...12 %).  The certificate that is reported is the usual float64
                            statement about the S that passed.  0: no second tries. */
    int32_t dual_refine;    /* -1 (default): 1.  A dual that still fails after the tries of dual_shift (and the dual of a FIRST attempt, which gets none of
                            those) gets one eigen-gradient step inside the same family: the bottom eigenvector n of S by two inverse
                            iterations (LDL^T of S + 0.005 I, started from the runner-up eigenvector of Z), the step S + tau P_U(n n^T) with tau
                            raising n^T S n to |lambda_min| (first order), one more LDL^T (cvx::dual_refine_step; wave-per-problem and quad
                            layouts).  Rescues ~87 % of the failed attempts of ten-point problems (the shift tries: 48 %); the certificate
                            that is reported is the usual float64 statement about the S that passed.  0: never. */
} cvxpnpl_opts_t;

void cvxpnpl_default_opts(cvxpnpl_opts_t *opts);
size_t cvxpnpl_opts_size(void); /* sizeof(cvxpnpl_opts_t) in this build of the library (see struct_size) */
/* The layout (CVXPNPL_LAYOUT_LANE ... _PENTA) the calling thread's last cvxpnpl_solve_batch / cvxpnpl_solve_cost_batch actually ran: AUTO resolves by
   launch size, and a request the chosen kernels cannot serve with the given options (e.g. LANE with first_check != lane_iters or warm_start = 0,
   PENTA with float64 sweeps) runs the next-best schedule for its size -- this says which.  0 before the first solve. */
int cvxpnpl_last_layout(void);

/*
 * Solve `batch` independent problems, each with n_p point and n_l line correspondences
 * (n_l = 0: cvxpnpl.pnp, :523; n_p = 0: cvxpnpl.pnl, :555; both: cvxpnpl.pnpl, :586).
 *
 *   d_pts_2d  [batch][n_p][2]      pixels                      (pts_2d,  cvxpnpl.py:524)
 *   d_pts_3d  [batch][n_p][3]                                  (pts_3d,  cvxpnpl.py:525)
 *   d_line_2d [batch][n_l][2][2]   (line, sample, xy)          (line_2d, cvxpnpl.py:556)
 *   d_line_3d [batch][n_l][2][3]   (line, end point, xyz)      (line_3d, cvxpnpl.py:557)
 *   d_K       [3][3] (K_per_problem = 0) or [batch][3][3]; general, inverted not assumed triangular (:37)
 * outputs (caller allocated; optional ones may be NULL)
 *   d_R      [batch][3][3] row-major, world -> camera, x_c = R X + t   (R of cvxpnpl.py:520)
 *   d_t      [batch][3]                                                 (t of cvxpnpl.py:513)
 *   d_status [batch] int32   CVXPNPL_*
 *   d_iters  [batch] int32   iterations used                    (optional)
 *   d_cost   [batch][2]      ||A r||^2 and the certified lower bound dobj, reference units
 *                            (the two sides of cvxpnpl.py:517); dobj = NaN if uncertified   (optional)
 *   d_Z      [batch][55]     vech(Z) in the order of cvxpnpl.py:346-370 -- `results["x"]`
 *                            of cvxpnpl.py:492                                               (optional)
 *   d_work   [batch][2] int32 rank at exit, Jacobi sweeps (work counter)                     (optional)
 * `stream` is a hipStream_t (NULL = default stream).  The call is asynchronous.
 * Returns 0, or a negative code for launch-level failures only (bad arguments -1,
 * HIP error -2: see cvxpnpl_last_error); per-problem outcomes are in d_status.
 */
int cvxpnpl_solve_batch(int64_t batch, int32_t n_p, const double *d_pts_2d, const double *d_pts_3d, int32_t n_l,
                        const double *d_line_2d, const double *d_line_3d, const double *d_K, int32_t K_per_problem,
                        const cvxpnpl_opts_t *opts, double *d_R, double *d_t, int32_t *d_status, int32_t *d_iters,
                        double *d_cost, double *d_Z, int32_t *d_work, void *stream);

/*
 * The same solve at the seam of the reference's private _solve_relaxation(A, B, eps, max_iters, verbose)
 * (cvxpnpl.py:454-460; callers: benchmarks/toolkit/methods/pnp.py:4 and rc.py:3): the caller brings the cost and the
 * translation map instead of correspondences.
 *   d_Q45 [batch][45]  A^T A (cvxpnpl.py:475), upper triangle row by row: entry (i <= j) at i*9 - i*(i-1)/2 + (j-i)
 *                      -- what cvxpnpl_assemble_batch / cvxpnpl_assemble_large_batch write
 *   d_B27 [batch][27]  B (3x9, row-major), t = -B r (cvxpnpl.py:513)
 * Outputs, options, return codes: as cvxpnpl_solve_batch.  opts->variant selects the constraint set
 * (CVXPNPL_VARIANT_RC: _solve_relaxation_rc, rc.py:67-131).
 */
int cvxpnpl_solve_cost_batch(int64_t batch, const double *d_Q45, const double *d_B27, const cvxpnpl_opts_t *opts, double *d_R,
                             double *d_t, int32_t *d_status, int32_t *d_iters, double *d_cost, double *d_Z, int32_t *d_work, void *stream);

/*
 * Host side of the cold path: all poses of a rank > 1 solution (cvxpnpl.py:507 ->
 * _constraint_ortho_det :221-343 -> _re6q3 :156-218), from Z = vech^-1(x) and B.
 * Host pointers.  R_out [4][3][3], t_out [4][3].  Q45 (optional, may be NULL): the packed 9x9 cost
 * A^T A from cvxpnpl_assemble_batch; when given, every recovered pose is Newton-polished on SO(3)
 * (the reference does not polish).  Returns the number of poses (2 or 4), 1 for a rank-1 Z, or -1
 * if rank is 0 / Z is not finite (reference: NotImplementedError / NaN sentinel).
 */
int cvxpnpl_recover_multi(const double *Z55, const double *B27, const double *Q45, double *R_out, double *t_out);

/*
 * The same for a whole batch, on host threads (n_threads <= 0: all cores): problem i is recovered when
 * status == NULL or status[i] == CVXPNPL_RANK_GT1, skipped (n_poses[i] = 0) otherwise.  HOST pointers:
 * Z55 [batch][55], B27 [batch][27], Q45 [batch][45] or NULL (copies of d_Z and of the outputs of
 * cvxpnpl_assemble_batch); R_out [batch][4][9], t_out [batch][4][3], n_poses [batch] (2, 4, 1, or -1 as above).
 * Returns 0, or -1 for bad arguments.  (the fast batched host path.)
 */
int cvxpnpl_recover_multi_batch(int64_t batch, const int32_t *status, const double *Z55, const double *B27, const double *Q45,
                                double *R_out, double *t_out, int32_t *n_poses, int32_t n_threads);

/*
 * The same on the DEVICE, one launch for the whole batch: no copy of Z to the host, no host thread per problem (a batch of
 * minimal RANSAC hypotheses flags 0.2 - 24 % of its problems, a planar batch all of them).  DEVICE pointers, same shapes
 * and meaning as cvxpnpl_recover_multi_batch: d_status [batch] or NULL, d_Z55 [batch][55] (cvxpnpl_solve_batch's d_Z),
 * d_B27 / d_Q45 (cvxpnpl_assemble_batch; d_Q45 may be NULL: no polish), d_R_out [batch][4][9], d_t_out [batch][4][3],
 * d_n_poses [batch] (0 = skipped, 2 / 4 / 1 / -1 as cvxpnpl_recover_multi).  Same source as the host path: bit-comparable.
 */
int cvxpnpl_recover_multi_device(int64_t batch, const int32_t *d_status, const double *d_Z55, const double *d_B27, const double *d_Q45,
                                 double *d_R_out, double *d_t_out, int32_t *d_n_poses, void *stream);

/* Translation maps B (3x9 per problem, t = -B r; cvxpnpl.py:548) for callers that need them
 * on the host (cvxpnpl_recover_multi).  d_B [batch][27]. */
int cvxpnpl_assemble_batch(int64_t batch, int32_t n_p, const double *d_pts_2d, const double *d_pts_3d, int32_t n_l,
                           const double *d_line_2d, const double *d_line_3d, const double *d_K, int32_t K_per_problem,
                           double *d_B, double *d_Q45, void *stream);

/*
 * The same outputs for problems with MANY correspondences (the reference's scalability benchmark goes to 10^4 points per
 * problem, benchmarks/scalability/pnp.py:37-40): several workgroups per problem stream the correspondences at HBM rate and
 * reduce the 60 Gram sums; sums are taken about the problem's first 3D point (exact, better conditioned far from the
 * origin).  Feed d_B / d_Q45 to cvxpnpl_solve_cost_batch.  d_scratch: DEVICE memory of at least
 * cvxpnpl_assemble_large_scratch_bytes(batch, n_p, n_l) bytes (partial sums; no initialisation needed).  batch <= 65535.
 * Deterministic: partial sums are added in a fixed order.
 */
size_t cvxpnpl_assemble_large_scratch_bytes(int64_t batch, int32_t n_p, int32_t n_l);
int cvxpnpl_assemble_large_batch(int64_t batch, int32_t n_p, const double *d_pts_2d, const double *d_pts_3d, int32_t n_l,
                                 const double *d_line_2d, const double *d_line_3d, const double *d_K, int32_t K_per_problem,
                                 double *d_B, double *d_Q45, void *d_scratch, size_t scratch_bytes, void *stream);

/*
 * The reference's benchmark toolkit on the device.  DEVICE pointers, one launch each.
 *
 * cvxpnpl_synth_batch: `batch` synthetic problems with n_p points and n_l lines each, the distributions of
 *   benchmarks/toolkit/suites/synth.py:27-42 (pose), :276-346 (3D points 0.6 (U - .5), pixels K (R X + t) + N(0, sigma^2),
 *   lines = consecutive point pairs); counter-based generator (Philox4x32-10 keyed by `seed`, counter = problem, record,
 *   draw): reproducible, independent of the launch geometry.  Outputs in the layout of cvxpnpl_solve_batch's inputs;
 *   d_R_gt [batch][9], d_t_gt [batch][3] (optional).
 * cvxpnpl_pose_errors: angular error [degrees] of R_gt^-1 R after projection onto O(3), and |t - t_gt| / |t_gt|
 *   (suite.py:8-14, :22-33); a NaN estimate gives NaN.
 * cvxpnpl_disambiguate: for every problem the candidate (of d_n_poses[b] <= 4 in d_R_all [batch][4][9], d_t_all [batch][4][3],
 *   e.g. the outputs of cvxpnpl_recover_multi_batch copied to the device) whose reprojection of the n_support support
 *   points d_support [n_support][3] is closest to the ground truth's (suite.py:96-108); d_index = -1 and NaN when none.
 */
int cvxpnpl_synth_batch(int64_t batch, int32_t n_p, int32_t n_l, double sigma, uint64_t seed, const double *d_K, double *d_pts_2d,
                        double *d_pts_3d, double *d_line_2d, double *d_line_3d, double *d_R_gt, double *d_t_gt, void *stream);
int cvxpnpl_pose_errors(int64_t batch, const double *d_R_gt, const double *d_t_gt, const double *d_R, const double *d_t, double *d_ang_deg,
                        double *d_trans, void *stream);
int cvxpnpl_disambiguate(int64_t batch, const double *d_R_all, const double *d_t_all, const int32_t *d_n_poses, const double *d_K,
                         const double *d_R_gt, const double *d_t_gt, const double *d_support, int32_t n_support, double *d_R, double *d_t,
                         int32_t *d_index, void *stream);

/* Results of one shard as the [batch][13] float64 records the multi-GPU gather exchanges (north-star config 4):
 * R (9, row-major), t (3), status.  DEVICE pointers; one launch on `stream`.  Returns 0, -1 for bad arguments. */
int cvxpnpl_pack_results(int64_t batch, const double *d_R, const double *d_t, const int32_t *d_status, double *d_packed, void *stream);

/* Ordering between two streams of one device without an event on the producing stream (bench.py: the solve stream hands a finished
   step to the stream that packs and all-gathers it): the producer stores `value` to a flag in device memory after everything it has
   enqueued so far, the consumer's stream does not go on before the flag has reached `value`.  The flag only grows (the store is an
   atomic max: values written from several streams may land out of order -- with more than one producing stream a wait for step n can
   then pass on the strength of step n + 1, so use one flag per producing stream); it is polled by one sleeping wavefront.  (An event
   record between two kernels of a stream costs that stream ~17 us here, this ~2 us.)
   d_flag points to TWO 64-bit words, both zero at the start: [0] the flag, [1] set to 1 by a wait that gave up after ~0.25 s of polling --
   which happens when the two streams share a hardware queue (the producer's kernel then sits behind the wait), or when the producer
   simply takes longer than that.
   cvxpnpl_stream_wait_value FAILS CLOSED (round 6): a wait that gave up HOLDS its stream -- nothing enqueued behind it runs -- until the
   host has acknowledged the give-up with cvxpnpl_stream_wait_gave_up(d_flag, 1, stream), which tells the caller (return value 1: discard
   what the waits ordered, fall back to an event) and releases the stream.  A give-up nobody acknowledges within about half a minute
   ends in a trap on that stream (a HIP error at the next call): loud, never a consumer silently reading unfinished results.  Callers of
   this wait must therefore poll cvxpnpl_stream_wait_gave_up instead of synchronising the stream blindly. */
int cvxpnpl_stream_write_value(uint64_t *d_flag, uint64_t value, void *stream);
int cvxpnpl_stream_wait_value(uint64_t *d_flag, uint64_t value, void *stream);
/* The EXPLICITLY FAIL-OPEN form, with the bound as a parameter: max_polls polls of ~1 us each before it gives up, sets d_flag[1] and lets
   its stream go on -- what the consumer then reads may not be finished.  For callers that check d_flag[1] (cvxpnpl_stream_wait_gave_up) at
   EVERY point where they synchronise and use what the waits ordered, and throw those results away when it is set: bench.py checks after
   its warm-up and again after its timed region, and repeats a region in which a wait gave up.  0 = unbounded: never gives up, and hangs
   the consumer stream for good if the two streams do share a hardware queue -- only for callers that have established (one bounded wait,
   checked) that they do not. */
int cvxpnpl_stream_wait_value_bounded(uint64_t *d_flag, uint64_t value, uint64_t max_polls, void *stream);
/* The check a consumer of these waits owes, as one call: waits until `stream` is idle OR a wait on this flag has given up, whichever comes
   first (the give-up word is read on a stream of the library's own, so the call works while a closed wait is holding `stream`).  Returns 1
   if a wait has given up since the word was last cleared, 0 if `stream` has drained and none has, negative for an error.  clear != 0
   resets the word after reading it -- which is also what releases a closed wait. */
int cvxpnpl_stream_wait_gave_up(uint64_t *d_flag, int32_t clear, void *stream);

/*
 * Consensus scoring of pose hypotheses against one scene (RANSAC on top of the solver: BASELINE config 5;
 * SURVEY.md section 8(f) row 3 -- the reference has no RANSAC, this is the consumer of its minimal solves).
 * DEVICE pointers.  d_R [n_hyp][9] row-major, d_t [n_hyp][3] (outputs of cvxpnpl_solve_batch), d_status
 * [n_hyp] or NULL; usable_mask: bit s set = hypotheses of status s are scored, the others get 0 (ignored
 * when d_status is NULL; 0x5 = CERTIFIED | UNCERTIFIED, the rank-1 poses).  d_K [9]; scene d_pts_2d [n_corr][2] pixels,
 * d_pts_3d [n_corr][3].  Correspondence m is an inlier of hypothesis h when (R P + t)_z > 0 and its
 * reprojection K (R P + t) lies within thresh_px of the measured pixel.  d_count [n_hyp] inlier counts;
 * d_mask [n_hyp][n_corr] (0/1) or NULL.  Non-finite poses score 0.  Returns 0, -1 for bad arguments.
 */
/* The selection of a RANSAC frame on the device (round 6; replaces ~40 small torch kernels around an arg-max).  DEVICE pointers, one launch of
 * one block each.
 * cvxpnpl_select_best: arg-max of d_count [n_hyp] (cvxpnpl_score_hypotheses) with a deterministic tie-break -- the lowest index among the
 *   best counts --, the winner's pose into d_out_R [9] / d_out_t [3], its inlier mask over the scene into d_mask [n_corr] (0/1, the
 *   arithmetic of cvxpnpl_score_hypotheses), and d_head [4] = { status of the winner, its inliers, its index, number of hypotheses with
 *   status CERTIFIED }: the one read-back a frame needs.  n_hyp >= 1.
 * cvxpnpl_refit_update: the pose refitted to the consensus set (cvxpnpl_assemble_subsets on d_mask, cvxpnpl_solve_cost_batch; d_fit_*: its
 *   R [9], t [3], status [1], d_fit_count [1] = size of the set) is scored against the scene and replaces pose, mask and d_head[0..1] TOGETHER
 *   when it is usable (status 0 or 2, at least four correspondences) and has at least d_head[1] inliers; otherwise nothing changes.
 * Return 0, -1 for bad arguments. */
int cvxpnpl_select_best(int64_t n_hyp, const int32_t *d_count, const double *d_R, const double *d_t, const int32_t *d_status, const double *d_K,
                        int32_t n_corr, const double *d_pts_2d, const double *d_pts_3d, double thresh_px, double *d_out_R, double *d_out_t,
                        int32_t *d_head, uint8_t *d_mask, void *stream);
int cvxpnpl_refit_update(const double *d_fit_R, const double *d_fit_t, const int32_t *d_fit_status, const int32_t *d_fit_count, const double *d_K,
                         int32_t n_corr, const double *d_pts_2d, const double *d_pts_3d, double thresh_px, double *d_R, double *d_t, int32_t *d_head,
                         uint8_t *d_mask, void *stream);
int cvxpnpl_score_hypotheses(int64_t n_hyp, const double *d_R, const double *d_t, const int32_t *d_status, uint32_t usable_mask,
                             const double *d_K, int32_t n_corr, const double *d_pts_2d, const double *d_pts_3d, double thresh_px,
                             int32_t *d_count, uint8_t *d_mask, void *stream);

/*
 * Constraint assembly for SUBSETS of one scene (same row of SURVEY.md section 8(f); the refit of a RANSAC consensus set without a host
 * round trip for its size): problem b takes the correspondences m of d_scene_2d [n_corr][2], d_scene_3d [n_corr][3] with
 * d_mask[b * n_corr + m] != 0 -- e.g. the mask cvxpnpl_score_hypotheses wrote.  Outputs as cvxpnpl_assemble_batch (d_B [batch][27],
 * d_Q45 [batch][45]: feed cvxpnpl_solve_cost_batch) and, if not NULL, d_count [batch] = correspondences taken.  A subset of fewer than
 * three correspondences gives NaN (-> status 3 in the solve).  DEVICE pointers.  Returns 0, -1 bad arguments, -2 HIP error.
 */
int cvxpnpl_assemble_subsets(int64_t batch, int32_t n_corr, const double *d_scene_2d, const double *d_scene_3d, const uint8_t *d_mask, const double *d_K,
                             double *d_B, double *d_Q45, int32_t *d_count, void *stream);

/*
 * Minimal sets for the hypotheses of a RANSAC frame (same row of SURVEY.md section 8(f) as the scoring above; no reference counterpart):
 * for every hypothesis k (1 ... 8) DISTINCT correspondences of the scene d_scene_2d [n_corr][2], d_scene_3d [n_corr][3], uniformly at
 * random (partial Fisher-Yates on a counter-based stream, Philox4x32-10 keyed by `seed` with counter (hypothesis, draw): every hypothesis
 * is drawn independently and reproducibly), gathered into d_pts_2d [n_hyp][k][2], d_pts_3d [n_hyp][k][3] -- the inputs of
 * cvxpnpl_solve_batch(n_hyp, k, ...) -- and, if d_idx is not NULL, their indices into d_idx [n_hyp][k].  DEVICE pointers, one launch.
 * Returns 0, -1 for bad arguments (k > 8, n_corr < k), -2 HIP error.
 */
int cvxpnpl_sample_minimal_sets(int64_t n_hyp, int32_t n_corr, const double *d_scene_2d, const double *d_scene_3d, int32_t k, uint64_t seed,
                                int32_t *d_idx, double *d_pts_2d, double *d_pts_3d, void *stream);

/*
 * Scratch of the lane / quad schedules (queue of parked problems + their iterates).  By default the library
 * keeps one grow-only hipMalloc allocation per (device, stream).  A caller that wants the memory under its own
 * allocator (e.g. torch's caching allocator) registers a DEVICE buffer of at least
 * cvxpnpl_workspace_bytes(largest batch) bytes for the launches it will issue on `stream`; the call initialises
 * the buffer on that stream (the queue cleans itself after every launch).  d_workspace = NULL unregisters.
 * cvxpnpl_release_workspace frees the library's own allocation of `stream` (all_streams != 0: of every stream
 * of the current device) after synchronising with it.  Return 0, -1 bad arguments, -2 HIP error.
 */
size_t cvxpnpl_workspace_bytes(int64_t max_batch);
int cvxpnpl_set_workspace(void *d_workspace, size_t bytes, void *stream);
int cvxpnpl_release_workspace(void *stream, int32_t all_streams);

/* Diagnostics: copy nbytes (a multiple of 16) from d_src to d_dst with 4, 8 or 16 bytes per lane -- a kernel of known
 * HBM traffic, against which bench.py calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE counters for this path's access
 * widths.  DEVICE pointers.  Returns 0, -1 bad arguments, -2 HIP error. */
int cvxpnpl_calibration_copy(const void *d_src, void *d_dst, int64_t nbytes, int32_t bytes_per_lane, void *stream);

/* Diagnostics / building block: the interior-point solve of the relaxation alone (ipm_quad.h: four problems per wavefront), without
 * rounding, polish or certificate -- what the solver's interior-point path (opts.rescue_from) runs for a slow problem before the
 * first-order iteration takes over again.  d_Qs55 [batch][55]: the cost c = vech(Q) of cvxpnpl.py:475-484 WITHOUT the factor 2 on
 * the off-diagonal entries, i.e. the upper triangle of Q row by row, scaled to trace 1 (entries outside the 9x9 block are ignored);
 * variant: CVXPNPL_VARIANT_*.  Outputs (DEVICE, required): d_Z100 / d_S100 [batch][100] the primal / dual iterates as full
 * symmetric matrices, d_gap [batch] = <Z, S>, d_iters [batch] = iterations | reason << 8 (why the solve ended: 0 iteration cap, 1 gap below
 * 1e-10, 2 / 3 a factorisation failed in rounding, 4 no step keeps the iterates positive definite, 5 the gap stopped decreasing).  Returns 0, -1 bad arguments, -2 HIP error. */
int cvxpnpl_ipm_batch(int64_t batch, const double *d_Qs55, int32_t variant, double *d_Z100, double *d_S100, double *d_gap, int32_t *d_iters,
                      void *stream);

/* HIP-event timing on the launch stream (for bench.py: torch.cuda.Event only sees torch's
 * current stream).  handles are opaque. */
void *cvxpnpl_event_create(void);
int cvxpnpl_event_record(void *event, void *stream);
int cvxpnpl_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on stop */
void cvxpnpl_event_destroy(void *event);

const char *cvxpnpl_last_error(void);
const char *cvxpnpl_version(void);
int cvxpnpl_device_count(void);

#ifdef __cplusplus
}
#endif
#endif /* CVXPNPL_AMD_H */
