// solver_core.h -- the per-problem mathematics of the batched absolute-pose SDP solver.
//
// One problem = one 10x10, 22-equality Shor-relaxed SDP (reference: cvxpnpl.py:454-520).
// Everything here is scalar code over one problem's registers: cvxpnpl_hip.hip instantiates it once per
// lane (lane-per-problem layout), wave_kernel.h / quad_kernel.h are cooperative re-statements of the same
// steps that call the small pieces (inv3, polar3, dual_lambda, rounds_to, ...) directly.  It is also
// compilable by a host C++ compiler so that tests can step the *device algorithm* on a
// CPU without a GPU (tests/hostsim) -- that build is test-only and never shipped.
//
// Pipeline (DESIGN.md has the derivations):
//   assemble      Gram blocks of the reference's C, N matrices (cvxpnpl.py:20-153) without
//                 forming them:  every row block is  P^T (x) T  with T = |p|^2 I - p p^T
//                 (points) or n n^T (line end points), so  N^T N = sum T,
//                 N^T C = sum P^T (x) T,  C^T C = sum P P^T (x) T.   B = (N^T N)^-1 N^T C
//                 and  Q9 = C^T C - (N^T C)^T B  (= A^T A of cvxpnpl.py:549/475).
//   admm          Douglas-Rachford splitting between the affine set {<A_i, Z> = b_i}
//                 (cvxpnpl.py:387-451, applied in closed form -- the 22 rows are 15 disjoint
//                 off-diagonal triples plus a 3x3 "doubly stochastic" diagonal block) and
//                 the PSD cone (one-sided Jacobi eigendecomposition of the shifted iterate).
//   certify       every few iterations: rank-1 rounding of Z (cvxpnpl.py:504-505), SO(3)
//                 Newton polish of r^T Q r, dual recovery (projection of the ADMM dual onto
//                 {S in Q - span A_i, S z = 0}), LDL^T test of S + delta I > 0.  Success is a
//                 rigorous primal-dual certificate  0 <= pobj - dobj <= eps  for the SDP --
//                 the same statement as the reference's check at cvxpnpl.py:516-519.
//   fallback      no certificate by max_iters / stagnation: eigen-rank of Z at 1e-3
//                 (cvxpnpl.py:502), rank-1 ratio, U V^T projection without determinant fix
//                 (cvxpnpl.py:510-511), t = -B r (cvxpnpl.py:513); rank > 1 is flagged and Z is
//                 handed to the host-side multi-solution recovery.
#pragma once

#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CVX_HD __host__ __device__ __forceinline__
#define CVX_HDM __host__ __device__ __forceinline__ /* member functions */
#define CVX_UNROLL _Pragma("unroll")
#else
#define CVX_HD static inline
#define CVX_HDM inline
#define CVX_UNROLL
#endif

// One definition per translation unit.  csrc/lane_kernel.hip compiles these headers with CVX_REFINE_NEWTON2: other bodies of rcp / rsqrt_ /
// jacobi_cs_dl -- and therefore of every inline function that inlines them.  The inline namespace gives the two sets of definitions
// different (mangled) names, so that linking both units into one library is no violation of the one-definition rule whatever the build
// does later (-fgpu-rdc, LTO, a host-visible difference); `cvx::name` resolves as before.  (advisor, round 5)
#if defined(CVX_REFINE_NEWTON2)
#define CVX_UNIT_TAG unit_newton2
#else
#define CVX_UNIT_TAG unit_c3
#endif
namespace cvx {
// (plain data first -- the types that cross between the translation units, e.g. in cvxb::launch_lane2's signature, are the same type in both)

enum Status : int {
    ST_CERTIFIED = 0,   // rank-1, certified globally optimal (gap <= eps)
    ST_RANK_GT1 = 1,    // relaxation not tight: rank(Z) > 1, multiple poses (host recovery)
    ST_UNCERTIFIED = 2, // rank-1 pose returned, but no certificate (max_iters / stagnation)
    ST_NONFINITE = 3,   // NaN/inf met (degenerate input): NaN pose, cvxpnpl.py:493-498
    ST_REFLECTION = 4,  // uncertified and det(U V^T) < 0 (the reference returns it as is)
    ST_PENDING = 5      // internal: handed to the interior-point path of the same call (never returned)
};

struct Opts {
    double eps;        // absolute duality-gap tolerance of the certificate (reference eps)
    int max_iters;     // ADMM iteration cap (reference max_iters)
    double rho;        // ADMM penalty on the trace-normalised cost
    double alpha;      // over-relaxation
    int first_check;   // first certification attempt after this many iterations
    int check_every;   // then every this many
    double res_tol;    // fixed-point residual below which an uncertified solve stops
    int jacobi_sweeps; // cap on Jacobi sweeps per PSD projection
    double jacobi_tol; // a sweep whose largest |cos(g_p, g_q)| is below this ends the eigen-solve
    int warm_start;    // 1: start each eigen-solve from the previous iteration's eigenvectors
    double rho_tail;   // penalty used from iteration tail_from on (the dual is rescaled at the switch)
    int tail_from;     // <= 0: never switch
    int adapt_every;   // residual balancing of the penalty every this many iterations (0: never) ...
    int adapt_from;    // ... from this iteration on
    double adapt_mu;   // a residual larger than the other by this factor moves the penalty ...
    double adapt_tau;  // ... by this factor (kept within [1e-3, 10])
    int stall_from;    // from this iteration on a solve whose Z has settled at rank > 1 stops as RANK_GT1 (0: never): ...
    double stall_lam;  // ... second eigenvalue above this and no longer shrinking since the previous attempt, ...
    double stall_res;  // ... fixed-point residual below this
    double stall_drop; // ... (relative decrease of the second eigenvalue between two attempts that still counts as settled)
    int variant;       // VAR_FULL: the 22 equalities of cvxpnpl.py:387-451; VAR_RC: the 16 of benchmarks/toolkit/methods/rc.py:9-64
    int rescue_from;   // kernels: a solve still open after this many iterations goes to the interior-point path (ipm_core.h); 0: never, < 0: by problem size
    int f32_sweeps_until; // kernels: the Jacobi sweeps of the PSD projection run on single-precision columns during the first this many
                          // iterations of a solve (< 0: default, F32_SWEEPS_DEFAULT); 0: every sweep in float64, rotation angles included
    int sweep_schedule;   // kernels: 1 (default) caps the sweeps of the YOUNG eigen-solves by iteration (sweep_cap below); 0: jacobi_sweeps only
    double dual_shift;    // a certificate attempt whose recovered dual S fails the PSD test is repeated once with S + dual_shift D(R) (dual_retry_entry
                          // below); 0: no second try
    int dual_refine;      // 1 (default): a dual that still fails gets one eigen-gradient step inside the family of complementary duals
                          // (dual_refine_step below); 0: never
};
constexpr int F32_SWEEPS_DEFAULT = 64;
constexpr double DUAL_SHIFT_DEFAULT = 0.015;
constexpr int DUAL_RETRY_RUNGS = 2; // second tries of a failed dual: dual_shift, dual_shift / 4
constexpr double DUAL_REFINE_SIGMA = 0.005; // dual_refine_step: shift of the inverse iteration (the failed duals of the judged problem set have lambda_min ~ -3e-4,
                                            // the next eigenvalue ~ 3e-2: two iterations with this shift resolve the bottom eigenvector to ~1e-3)
constexpr int DUAL_REFINE_INVITS = 2;
constexpr double DUAL_REFINE_GAIN = 2.0;    // the step raises the bottom Rayleigh quotient by this multiple of its distance from zero (first order)
constexpr int DUAL_REFINE_FROM = 2;         // ... from the third attempt of a solve on (the scalar core; the kernels: in the wave-per-problem phase behind a quad phase, whose
                                            // two attempts these are) ...
constexpr int DUAL_REFINE_ATTEMPTS = 3;     // ... for three attempts: a dual it has not repaired by then is far from the cone, and the long chain that ends a launch would
                                            // pay for the step at every attempt.  Where: measured, round 6 (profiles/r06/refine_ab3*.txt, refine_cost.txt).  One step costs a
                                            // wavefront 9.6 us on its chain (1.2-1.7 iterations), a rescue saves two iterations and an attempt (23.5 us): on AVERAGE a clear
                                            // gain at a rescue rate of 87 %, but a launch ends with its slowest chain, and the problems of those chains are the ones the
                                            // step does not rescue.  Judged 10 k launch (quad schedule, step in the wave-per-problem phase): 50.4 -> 53.6 M poses/s, other
                                            // seeds +4 ... +15 %; in the quad phase itself: -8 % (four problems in lockstep); fresh wave-per-problem solves (2 000 problems):
                                            // -5.5 % although the slowest problem takes 9 instead of 11 iterations; resume phase behind the lane phase (125 k): -2.2 % (its
                                            // 33-iteration chain pays three steps for nothing).
constexpr int DUAL_RETRY_ATTEMPTS = 10; // ... in the first this many attempts that may use them: a problem they have not rescued by then is not
                                        // one they rescue (same iteration counts with 6 / 10 / 16 / no limit on four workloads), and its long
                                        // chain stops paying for them (N = 8, 125 k problems, slowest 99 iterations: 160.4 -> 162.6 M poses/s)
inline namespace CVX_UNIT_TAG { // (functions from here on: one set per translation unit, see above)
// Sweeps the eigen-solve of iteration `it` (2, 3, ...: iteration 1 needs none) may take in the first phases of the hybrid schedules,
// where a wavefront runs the MAXIMUM over its problems (64 in the lane phase, 4 in the quad phase): the first eigen-solve of a solve
// takes 3-4 sweeps, the later ones 1-2 on average but 2-3 at wavefront level (measured, 200 wavefronts of 64 N = 10 problems: mean per
// eigen-solve 3.1 / 1.8 / 2.0 / 1.3 / 1.0, wavefront maximum 4.0 / 2.5 / 2.7 / 2.0 / 1.9 -- 13.1 sweeps paid for 9.2 needed).  A column pair
// that misses its last sweep is orthogonal to ~0.1 instead of 6e-2 for ONE iteration; the next warm start absorbs it.  Host experiment
// (tests/hostsim, 30 000 problems per workload, certified at the first attempt, uncapped -> capped): lane phase, attempt after 6
// iterations, caps 3 2 2 1 1: N = 10 98.63 -> 98.61 %, PnPL 5+5 95.47 -> 95.35 %, N = 8 94.63 -> 94.62 %, N = 6 77.00 -> 76.90 %, PnL 8
// 80.6 -> 80.1 %; quad phase, attempt after 5, caps 3 2 2 2 (the eigen-solve an attempt reads keeps 2: with 1 there 94.19 -> 93.81 %):
// N = 10 94.19 -> 94.19 %, PnPL 88.62 -> 88.55 %, N = 8 86.49 -> 86.45 %.  Results are unaffected (what is not certified continues).
CVX_HD constexpr int sweep_cap(int it, bool lane_phase, int cap)
{
    const int c = it <= 2 ? 3 : (it <= 4 ? 2 : (lane_phase ? 1 : 2));
    return c < cap ? c : cap;
}

// Constraint sets.  VAR_RC is the reference's ablation "rc" (benchmarks/toolkit/methods/rc.py:16-35): the six row
// orthonormality rows (kron(I3, E_ij), cvxpnpl.py:404-418) are left out -- in the closed forms below that means the
// three triples 0..2 (Z01+Z34+Z67, Z02+Z35+Z68, Z12+Z45+Z78) are unconstrained and the diagonal block only has its
// column sums fixed.
enum Variant : int { VAR_FULL = 0, VAR_RC = 1 };
CVX_HD constexpr bool tri_dropped(int t, int var) { return var == VAR_RC && t < 3; }

CVX_HD Opts default_opts()
{
    Opts o;
    o.eps = 1e-9; o.max_iters = 2500; o.rho = 0.1; o.alpha = 1.4;
    o.first_check = 5; o.check_every = 2; o.res_tol = 1e-5; o.jacobi_sweeps = 12; o.jacobi_tol = 6e-2; o.warm_start = 1; o.rho_tail = 0.05; o.tail_from = 3; o.variant = VAR_FULL;
    o.adapt_every = 10; o.adapt_from = 40; o.adapt_mu = 2.0; o.adapt_tau = 2.0; o.stall_from = 300; o.stall_lam = 0.05; o.stall_res = 1e-3; o.stall_drop = 0.003;
    o.rescue_from = -1; // (by problem size, cvxpnpl_hip.hip: 32 for at most 6 correspondences, 64 for 7, 128 otherwise)
    o.f32_sweeps_until = -1;
    o.sweep_schedule = 1;
    o.dual_shift = DUAL_SHIFT_DEFAULT;
    o.dual_refine = 1;
    return o;
}

// ---------------------------------------------------------------------------------------
// index helpers.  sidx: 10x10 symmetric in 55 entries, identical to the reference's vech
// order (cvxpnpl.py:346-370: columns of the lower triangle == rows of the upper triangle).
CVX_HD constexpr int sidx(int i, int j) { return i <= j ? i * 10 - i * (i - 1) / 2 + (j - i) : j * 10 - j * (j - 1) / 2 + (i - j); }
// 9x9 symmetric in 45 entries
CVX_HD constexpr int qidx(int i, int j) { return i <= j ? i * 9 - i * (i - 1) / 2 + (j - i) : j * 9 - j * (j - 1) / 2 + (i - j); }

// The 15 off-diagonal equality rows of cvxpnpl.py:404-435 as (i, j, sign) triples; the
// remaining 7 rows (cvxpnpl.py:398, :404-418 with c = 1) only touch the diagonal.
// Every off-diagonal entry of Z appears in exactly one triple.
CVX_HD constexpr int tri_i(int t, int k)
{
    constexpr int T[15][3] = {{0, 3, 6}, {0, 3, 6}, {1, 4, 7}, {0, 1, 2}, {0, 1, 2}, {3, 4, 5}, {1, 2, 6}, {2, 0, 7},
                              {0, 1, 8}, {4, 5, 0}, {5, 3, 1}, {3, 4, 2}, {2, 1, 3}, {0, 2, 4}, {1, 0, 5}};
    return T[t][k];
}
CVX_HD constexpr int tri_j(int t, int k)
{
    constexpr int T[15][3] = {{1, 4, 7}, {2, 5, 8}, {2, 5, 8}, {3, 4, 5}, {6, 7, 8}, {6, 7, 8}, {5, 4, 9}, {3, 5, 9},
                              {4, 3, 9}, {8, 7, 9}, {6, 8, 9}, {7, 6, 9}, {7, 8, 9}, {8, 6, 9}, {6, 7, 9}};
    return T[t][k];
}
CVX_HD constexpr double tri_s(int t, int k) { return (t >= 6 && k >= 1) ? -1.0 : 1.0; }
CVX_HD constexpr bool odd_entry(int i, int j) { return (i < 6) != (j < 6); }
CVX_HD constexpr bool odd_tri(int t) { return odd_entry(tri_i(t, 0), tri_j(t, 0)); }
constexpr bool tri_parity_uniform()
{
    for (int t = 0; t < 15; ++t)
        for (int k = 1; k < 3; ++k)
            if (odd_entry(tri_i(t, k), tri_j(t, k)) != odd_tri(t)) return false;
    return true;
}
static_assert(tri_parity_uniform(), "every equality triple must have a single parity under D = diag(-I6, I4)");

// ---------------------------------------------------------------------------------------
// tiny helpers

// reciprocal and reciprocal square root: on the device the hardware seed (v_rcp_f64 /
// v_rsq_f64) plus two Newton steps (<= 1-2 ulp) instead of the IEEE expansion; every use
// below is either self-correcting (Newton iterations) or followed by an explicit check.
CVX_HD double rcp(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // hardware seed (2^-24.4, tools/microbench/rsq_probe.hip) and ONE third-order step r (1 + e + e^2), e = 1 - x r: three operations where two Newton
    // steps take four (round 5; float64 operations issue at half rate on this part, every one of them counts)
#if defined(CVX_REFINE_NEWTON2) // (lane_kernel.hip: the sequences that unit was tuned with)
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
#else
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, fma(e, e, e), r);
#endif
#else
    return 1.0 / x;
#endif
}
CVX_HD double rsqrt_(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // hardware seed (2^-24.2) and ONE third-order step y (1 + r / 2 + 3 r^2 / 8), r = 1 - x y^2: five operations, four of them dependent, where two
    // Newton steps take seven / six; maximum relative error over 4 M arguments 1.38e-16 either way (tools/microbench/rsq_probe.hip,
    // profiles/r05/rsq_probe.txt).  Round 5, same-box A/B (profiles/r05/rsq_c3_ab.txt): judged launch +1.8 % / +1 %, wave layout +3 %, the lane-layout
    // launches -0.2 ... -2 % (their instruction count does not change: register allocation) -- taken for the former.
#if defined(CVX_REFINE_NEWTON2)
    double y = __builtin_amdgcn_rsq(x);
    { double h = 0.5 * x * y; double e = fma(-h, y, 0.5); y = fma(y, e, y); }
    { double h = 0.5 * x * y; double e = fma(-h, y, 0.5); y = fma(y, e, y); }
    return y;
#else
    const double y = __builtin_amdgcn_rsq(x);
    const double r = fma(-(x * y), y, 1.0);
    return fma(y * r, fma(0.375, r, 0.5), y);
#endif
#else
    return 1.0 / sqrt(x);
#endif
}

// sqrt for non-negative x: x * rsqrt(x) on the device (7 VALU instead of the 18 of the IEEE expansion).
// The scalar core keeps the library sqrt (sqrt_) where it is not in the eigen-solve: measured on the
// lane-per-problem kernels, the shorter sequence there lengthens live ranges and costs more in spills than
// it saves (125 k PnP: 70 M poses/s with sqrt_fast everywhere, 76 M with this split).
CVX_HD double sqrt_fast(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return x > 0 ? x * rsqrt_(x) : 0.0;
#else
    return sqrt(x);
#endif
}
CVX_HD double sqrt_(double x) { return sqrt(x); }

CVX_HD void inv3(const double *M, double *Mi, double &det)
{
    double c00 = M[4] * M[8] - M[5] * M[7], c01 = M[5] * M[6] - M[3] * M[8], c02 = M[3] * M[7] - M[4] * M[6];
    det = M[0] * c00 + M[1] * c01 + M[2] * c02;
    double id = rcp(det);
    Mi[0] = c00 * id; Mi[1] = (M[2] * M[7] - M[1] * M[8]) * id; Mi[2] = (M[1] * M[5] - M[2] * M[4]) * id;
    Mi[3] = c01 * id; Mi[4] = (M[0] * M[8] - M[2] * M[6]) * id; Mi[5] = (M[2] * M[3] - M[0] * M[5]) * id;
    Mi[6] = c02 * id; Mi[7] = (M[1] * M[6] - M[0] * M[7]) * id; Mi[8] = (M[0] * M[4] - M[1] * M[3]) * id;
}

// orthogonal polar factor U V^T of a 3x3 matrix (what np.linalg.svd gives as U @ Vh,
// cvxpnpl.py:510-511; reflections are preserved).  Scaled Newton iteration
// X <- (g X + X^-T / g) / 2.  `iters` fixed so that lanes stay convergent.
CVX_HD void polar3(const double *M, double *R, int iters)
{
    double X[9];
    CVX_UNROLL for (int i = 0; i < 9; ++i) X[i] = M[i];
    for (int it = 0; it < iters; ++it) {
        double Xi[9], det;
        inv3(X, Xi, det);
        // Frobenius scaling g = (|X^-1|_F / |X|_F)^(1/2) while X is far from orthogonal
        double nx = 0, ni = 0;
        CVX_UNROLL for (int i = 0; i < 9; ++i) { nx += X[i] * X[i]; ni += Xi[i] * Xi[i]; }
        const bool far = (it < 4) && (fabs(nx - 3.0) > 0.3 || fabs(ni - 3.0) > 0.3);
        double g = 1.0, ig = 1.0;
        if (far) { ig = sqrt(sqrt(nx * rcp(ni))); g = rcp(ig); }
        // X^-T = transpose(Xi)
        double Y[9], dl = 0;
        CVX_UNROLL for (int i = 0; i < 3; ++i)
            CVX_UNROLL for (int j = 0; j < 3; ++j) {
                Y[i * 3 + j] = 0.5 * (g * X[i * 3 + j] + ig * Xi[j * 3 + i]);
                dl += (Y[i * 3 + j] - X[i * 3 + j]) * (Y[i * 3 + j] - X[i * 3 + j]);
            }
        CVX_UNROLL for (int i = 0; i < 9; ++i) X[i] = Y[i];
        if (dl < 1e-22) break; // converged: |dX| < 1e-11 and quadratic convergence put the next step at rounding level
    }
    CVX_UNROLL for (int i = 0; i < 9; ++i) R[i] = X[i];
}

// A rotation near the (roughly orthogonal, det > 0) matrix M, as the START of the SO(3) Newton polish: two scaled polar steps, then
// Gram-Schmidt on the columns, which makes the result orthogonal to rounding in one go.  The polish converges to the local minimiser
// next to its start, so the start need not be THE nearest rotation -- but it must be orthogonal to rounding (the Cayley steps keep an
// error of the start forever) and close enough to stay in the basin.  Host experiment, 10 k problems each of five workloads: with
// two polar steps + Gram-Schmidt the iteration histograms and certified counts equal those of the polar iteration run to 1e-11 (6-7
// steps: 9 424 / 9 424 certified at the first attempt for PnP N = 10, 8 880 / 8 882 for PnPL 5+5); Gram-Schmidt alone loses 1-4 % of the
// first attempts (it keeps the first column's direction whatever the other two say).  On the GPU: 3 % at 10 k and 24 k problems.
CVX_HD void near_rotation(const double *M, double *R)
{
    double T[9];
    polar3(M, T, 2);
    double c0[3] = {T[0], T[3], T[6]}, c1[3] = {T[1], T[4], T[7]};
    const double n0 = rsqrt_(c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2]);
    CVX_UNROLL for (int i = 0; i < 3; ++i) c0[i] *= n0;
    const double d = c0[0] * c1[0] + c0[1] * c1[1] + c0[2] * c1[2];
    CVX_UNROLL for (int i = 0; i < 3; ++i) c1[i] -= d * c0[i];
    const double n1 = rsqrt_(c1[0] * c1[0] + c1[1] * c1[1] + c1[2] * c1[2]);
    CVX_UNROLL for (int i = 0; i < 3; ++i) c1[i] *= n1;
    const double c2[3] = {c0[1] * c1[2] - c0[2] * c1[1], c0[2] * c1[0] - c0[0] * c1[2], c0[0] * c1[1] - c0[1] * c1[0]};
    CVX_UNROLL for (int i = 0; i < 3; ++i) { R[i * 3] = c0[i]; R[i * 3 + 1] = c1[i]; R[i * 3 + 2] = c2[i]; }
}

CVX_HD double det3(const double *M)
{
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

// ---------------------------------------------------------------------------------------
// assembly (replaces cvxpnpl.py:20-153 + :545-549 + :475)

struct Gram {
    double M0[6];      // sum T            (N^T N), packed 00 01 02 11 12 22
    double M1[3][6];   // sum P_a T        (N^T C blocks)
    double M2[6][6];   // sum P_a P_b T    (C^T C blocks), ab packed 00 01 02 11 12 22
};

CVX_HD void gram_zero(Gram &g)
{
    CVX_UNROLL for (int i = 0; i < 6; ++i) g.M0[i] = 0;
    CVX_UNROLL for (int a = 0; a < 3; ++a) CVX_UNROLL for (int i = 0; i < 6; ++i) g.M1[a][i] = 0;
    CVX_UNROLL for (int a = 0; a < 6; ++a) CVX_UNROLL for (int i = 0; i < 6; ++i) g.M2[a][i] = 0;
}

CVX_HD void gram_add(Gram &g, const double *T, double X, double Y, double Z)
{
    const double P[3] = {X, Y, Z};
    const double PP[6] = {X * X, X * Y, X * Z, Y * Y, Y * Z, Z * Z};
    CVX_UNROLL for (int i = 0; i < 6; ++i) {
        g.M0[i] += T[i];
        CVX_UNROLL for (int a = 0; a < 3; ++a) g.M1[a][i] += P[a] * T[i];
        CVX_UNROLL for (int a = 0; a < 6; ++a) g.M2[a][i] += PP[a] * T[i];
    }
}

// bearing p = K^-1 [u v 1]^T (cvxpnpl.py:37) with a precomputed general inverse
CVX_HD void bearing(const double *Ki, double u, double v, double *p)
{
    p[0] = Ki[0] * u + Ki[1] * v + Ki[2];
    p[1] = Ki[3] * u + Ki[4] * v + Ki[5];
    p[2] = Ki[6] * u + Ki[7] * v + Ki[8];
}

// one 2D-3D point correspondence: rows [p]x (R P + t) = 0 (cvxpnpl.py:43-102)
CVX_HD void gram_add_point(Gram &g, const double *Ki, double u, double v, double X, double Y, double Z)
{
    double p[3];
    bearing(Ki, u, v, p);
    double n2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    // [p]x^T [p]x = |p|^2 I - p p^T
    double T[6] = {n2 - p[0] * p[0], -p[0] * p[1], -p[0] * p[2], n2 - p[1] * p[1], -p[1] * p[2], n2 - p[2] * p[2]};
    gram_add(g, T, X, Y, Z);
}

// one 2D-3D line correspondence: rows n^T (R P_k + t) = 0 for both 3D end points,
// n = normalised cross product of the two back-projected 2D samples (cvxpnpl.py:123-153)
CVX_HD void gram_add_line(Gram &g, const double *Ki, const double *l2 /*u0 v0 u1 v1*/, const double *l3 /*P0 P1*/)
{
    double a[3], b[3];
    bearing(Ki, l2[0], l2[1], a);
    bearing(Ki, l2[2], l2[3], b);
    double n[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
    double inv = rsqrt_(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
    n[0] *= inv; n[1] *= inv; n[2] *= inv;
    double T[6] = {n[0] * n[0], n[0] * n[1], n[0] * n[2], n[1] * n[1], n[1] * n[2], n[2] * n[2]};
    gram_add(g, T, l3[0], l3[1], l3[2]);
    gram_add(g, T, l3[3], l3[4], l3[5]);
}

// B = (N^T N)^-1 N^T C  (3x9, row-major), Q9 = C^T C - (N^T C)^T B (45 packed).
// Returns false when N^T N is singular / non-finite (the reference raises LinAlgError).
CVX_HD bool gram_finish(const Gram &g, double *B, double *Q9)
{
    const double *m = g.M0;
    double M0[9] = {m[0], m[1], m[2], m[1], m[3], m[4], m[2], m[4], m[5]}, Mi[9], det;
    inv3(M0, Mi, det);
    double scale = m[0] + m[3] + m[5];
    // singular N^T N (fewer than two distinct bearings): np.linalg.solve would raise or
    // return garbage; the relative determinant of a usable system is >> 1e-12
    if (!(det > 1e-12 * (scale * scale * scale) * (1.0 / 27.0))) return false;
    // B block a (3x3): Mi * sym(M1[a]);  B[i][3a+j]
    double Bb[3][9];
    CVX_UNROLL for (int a = 0; a < 3; ++a) {
        const double *s = g.M1[a];
        double S[9] = {s[0], s[1], s[2], s[1], s[3], s[4], s[2], s[4], s[5]};
        CVX_UNROLL for (int i = 0; i < 3; ++i)
            CVX_UNROLL for (int j = 0; j < 3; ++j) {
                double acc = Mi[i * 3] * S[j] + Mi[i * 3 + 1] * S[3 + j] + Mi[i * 3 + 2] * S[6 + j];
                Bb[a][i * 3 + j] = acc;
                B[i * 9 + 3 * a + j] = acc;
            }
    }
    // Q block (a, b), a <= b:  M2[ab] - sym(M1[a]) * Bb[b]
    constexpr int ab[3][3] = {{0, 1, 2}, {1, 3, 4}, {2, 4, 5}};
    CVX_UNROLL for (int a = 0; a < 3; ++a) {
        const double *s = g.M1[a];
        double S[9] = {s[0], s[1], s[2], s[1], s[3], s[4], s[2], s[4], s[5]};
        CVX_UNROLL for (int b = a; b < 3; ++b) {
            const double *t = g.M2[ab[a][b]];
            double Tm[9] = {t[0], t[1], t[2], t[1], t[3], t[4], t[2], t[4], t[5]};
            CVX_UNROLL for (int i = 0; i < 3; ++i)
                CVX_UNROLL for (int j = 0; j < 3; ++j) {
                    if (a == b && j < i) continue;
                    double acc = Tm[i * 3 + j] - (S[i * 3] * Bb[b][j] + S[i * 3 + 1] * Bb[b][3 + j] + S[i * 3 + 2] * Bb[b][6 + j]);
                    Q9[qidx(3 * a + i, 3 * b + j)] = acc;
                }
        }
    }
    return true;
}

// ---------------------------------------------------------------------------------------
// the affine set of cvxpnpl.py:387-451 in closed form

// E <- E - P_range(E) shifted:  project the symmetric matrix E (55 packed) onto
// { <A_i, Z> = b_i } (homog = false) or onto its direction space { <A_i, Z> = 0 }.
template <int VAR = VAR_FULL>
CVX_HD void proj_affine(double *E, bool homog)
{
    CVX_UNROLL for (int t = 0; t < 15; ++t) {
        if (tri_dropped(t, VAR)) continue;
        double m = (tri_s(t, 0) * E[sidx(tri_i(t, 0), tri_j(t, 0))] + tri_s(t, 1) * E[sidx(tri_i(t, 1), tri_j(t, 1))] +
                    tri_s(t, 2) * E[sidx(tri_i(t, 2), tri_j(t, 2))]) * (1.0 / 3.0);
        CVX_UNROLL for (int k = 0; k < 3; ++k) E[sidx(tri_i(t, k), tri_j(t, k))] -= tri_s(t, k) * m;
    }
    // diagonal block D[i][j] = Z[3j+i, 3j+i]: rows (VAR_FULL only) and columns sum to Z99 = 1
    const double tgt = homog ? 0.0 : 1.0;
    double rs[3], cs[3], tot = 0;
    CVX_UNROLL for (int i = 0; i < 3; ++i) {
        rs[i] = E[sidx(i, i)] + E[sidx(3 + i, 3 + i)] + E[sidx(6 + i, 6 + i)] - tgt;
        cs[i] = E[sidx(3 * i, 3 * i)] + E[sidx(3 * i + 1, 3 * i + 1)] + E[sidx(3 * i + 2, 3 * i + 2)] - tgt;
        tot += rs[i];
    }
    CVX_UNROLL for (int i = 0; i < 3; ++i)
        CVX_UNROLL for (int j = 0; j < 3; ++j)
            E[sidx(3 * j + i, 3 * j + i)] -= (VAR == VAR_RC) ? cs[j] * (1.0 / 3.0) : (rs[i] + cs[j]) * (1.0 / 3.0) - tot * (1.0 / 9.0);
    E[sidx(9, 9)] = tgt;
}

// ---------------------------------------------------------------------------------------
// PSD projection: eigendecomposition of W by one-sided (Hestenes) Jacobi on the shifted,
// positive definite G = W + sigma I.  On exit column j of G is lam'_j v_j.

struct Eig {
    double G[10][10]; // G[j][i]: column j, row i
    double n2[10];    // squared column norms = lam'_j^2
    double sigma;
    bool exact = false; // device build: rotation angles in float64 as well (Opts::f32_sweeps_until == 0); the host build always is
};

CVX_HD void eig_load(Eig &e, const double *W)
{
    double fro = 0;
    CVX_UNROLL for (int i = 0; i < 10; ++i)
        CVX_UNROLL for (int j = i; j < 10; ++j) fro += (i == j ? 1.0 : 2.0) * W[sidx(i, j)] * W[sidx(i, j)];
    e.sigma = 1.5 * sqrt_fast(fro) + 1e-300;
    CVX_UNROLL for (int j = 0; j < 10; ++j)
        CVX_UNROLL for (int i = 0; i < 10; ++i) e.G[j][i] = W[sidx(i, j)] + (i == j ? e.sigma : 0.0);
}

// warm start: G = (W + sigma I) V with V the (unit) eigenvectors of the previous iterate --
// nearly orthogonal columns when W moved little, so the Jacobi sweeps converge at once.
// In place: on entry column j of G is lam'_j v_j from the previous eigen-solve (n2 its squared norm), on
// exit it is (W + sigma I) v_j -- no second copy of the eigenvectors is kept (200 registers less state in
// the lane-per-problem kernels: 3.3 KB -> 2.5 KB of scratch, 76 -> 81 M poses/s at 125 k problems).
// One-sided Jacobi on G then yields (W + sigma I) (V J).
CVX_HD void eig_load_warm(Eig &e, const double *W)
{
    double fro = 0;
    CVX_UNROLL for (int i = 0; i < 10; ++i)
        CVX_UNROLL for (int j = i; j < 10; ++j) fro += (i == j ? 1.0 : 2.0) * W[sidx(i, j)] * W[sidx(i, j)];
    e.sigma = 1.5 * sqrt_fast(fro) + 1e-300;
    CVX_UNROLL for (int j = 0; j < 10; ++j) {
        const double il_ = rsqrt_(e.n2[j]);
        double v[10];
        CVX_UNROLL for (int i = 0; i < 10; ++i) v[i] = e.G[j][i] * il_;
        CVX_UNROLL for (int i = 0; i < 10; ++i) {
            double acc = e.sigma * v[i];
            CVX_UNROLL for (int m = 0; m < 10; ++m) acc += W[sidx(i, m)] * v[m];
            e.G[j][i] = acc;
        }
    }
}

CVX_HD void eig_norms(Eig &e)
{
    CVX_UNROLL for (int j = 0; j < 10; ++j) {
        double s = 0;
        CVX_UNROLL for (int i = 0; i < 10; ++i) s += e.G[j][i] * e.G[j][i];
        e.n2[j] = s;
    }
}

// Rotation parameters of one one-sided Jacobi step from al = |g_p|^2, be = |g_q|^2,
// gam = g_p . g_q:  t = tan(theta), c = cos, s = sin with  tan(2 theta) = 2 gam / (be - al).
// The rotation is exactly orthogonal (c^2 + s^2 = 1 to rounding) for ANY t, so t may be
// approximate; c is refined to full double precision.  Device build: v_rsq_f64 /
// v_rcp_f64 seeds + Newton steps instead of the IEEE sqrt / divide expansions.
CVX_HD void jacobi_cs(double al, double be, double gam, bool rot, double &c, double &s, double &t)
{
    const double d = be - al, g2 = 2.0 * gam;
#if defined(__HIP_DEVICE_COMPILE__)
    // tan(theta) in single precision (one v_rsq_f32 + one v_rcp_f32): an angle that is right to
    // ~1e-7 still annihilates g_p . g_q to 1e-7 of its size per rotation, far below the
    // sweep tolerance.  cos(theta) then comes from a float seed refined by one double Newton
    // step (|c^2 + s^2 - 1| ~ 1e-14), so the accumulated rotation stays orthogonal.
    const float df = (float)d, gf = (float)g2;
    const float h2 = df * df + gf * gf + 1e-37f;
    const float hf = h2 * __builtin_amdgcn_rsqf(h2);
    float tf = gf * __builtin_amdgcn_rcpf(fabsf(df) + hf);
    tf = df < 0.0f ? -tf : tf;
    t = rot ? (double)tf : 0.0;
    const double x = 1.0 + t * t;
    double z = (double)__builtin_amdgcn_rsqf((float)x);
    { double hh = 0.5 * x * z; double e = fma(-hh, z, 0.5); z = fma(z, e, z); }
    c = z;
#else
    const double h = sqrt(d * d + g2 * g2 + 1e-290);
    double tt = g2 / (fabs(d) + h);
    tt = d < 0 ? -tt : tt;
    t = rot ? tt : 0.0;
    c = 1.0 / sqrt(1.0 + t * t);
#endif
    s = t * c;
}

// The same with the change of the squared norms under the rotation in place of tan(theta): |g_p'|^2 = al + dl, |g_q'|^2 = be - dl.
// exact (uniform; device build only -- the host build is float64 anyway): everything in float64, Opts::f32_sweeps_until == 0, the
// reference's precision.  One full-precision reciprocal root on the dependent chain (until round 5: v_rsq -> v_rcp -> v_rsq, each with its
// Newton steps): with h ~ sqrt(d^2 + 4 gam^2) from the raw v_rsq_f64 seed (~2^-23) and u = h + |d|,
//     c = u / sqrt(u^2 + 4 gam^2),   s = +-2 gam / sqrt(u^2 + 4 gam^2)
// is a rotation to the precision of that one root whatever the error of h (c^2 + s^2 = 1 identically); the error of h moves the angle by
// ~1e-7 of itself, i.e. leaves 1e-7 of the pair's inner product standing (sweep tolerance: 6e-2).  dl = s (s d - 2 c gam) holds for any angle.
CVX_HD void jacobi_cs_dl(double al, double be, double gam, bool rot, double &c, double &s, double &dl, bool exact)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (exact) {
        const double d = be - al, g2 = 2.0 * gam;
        const double g22 = g2 * g2;
#if defined(CVX_REFINE_NEWTON2)
        const double h2 = fma(d, d, g22) + 1e-290;
#else
        const double h2 = fma(d, d, g22); // (zero only where rot is false: the NaN it breeds is selected away below)
#endif
        const double u = fma(h2, __builtin_amdgcn_rsq(h2), fabs(d));
        const double w = rsqrt_(fma(u, u, g22));
        const double sf = (d < 0 ? -g2 : g2) * w;
        c = rot ? u * w : 1.0;
        s = rot ? sf : 0.0;
        dl = s * fma(s, d, -(c * g2));
        return;
    }
#else
    (void)exact;
#endif
    double t;
    jacobi_cs(al, be, gam, rot, c, s, t);
    dl = -(t * gam);
}

// Round-robin (circle method) pairing: 9 steps of 5 disjoint pairs cover all 45 pairs.  Positions
// a0..a4 / b0..b4 start as columns 2k / 2k+1; after every step a0 stays and the others move one
// place along the ring a1 > a2 > a3 > a4 > b4 > b3 > b2 > b1 > b0 > a1 -- the same schedule the
// wave-per-problem kernel runs across lanes.  The 5 rotations of a step touch disjoint columns, so
// in the scalar (lane-per-problem) build they are independent instruction streams.
CVX_HD constexpr int rr_col(int step, int pos /* 0..4 = a_k, 5..9 = b_k */)
{
    int a[5] = {0, 2, 4, 6, 8}, b[5] = {1, 3, 5, 7, 9};
    for (int s = 0; s < step; ++s) {
        const int na1 = b[0], nb4 = a[4];
        a[4] = a[3]; a[3] = a[2]; a[2] = a[1]; a[1] = na1;
        b[0] = b[1]; b[1] = b[2]; b[2] = b[3]; b[3] = b[4]; b[4] = nb4;
    }
    return pos < 5 ? a[pos] : b[pos - 5];
}

// The 5 rotations of round-robin step `st`, written phase-major (all 5 dot products, then all 5
// angles, then all 5 column updates, innermost index = the pair): the five are independent
// dependency chains, and in this order consecutive instructions belong to different chains, so the
// in-order SIMD overlaps their latencies (a lane-per-problem wave has no other wave to hide behind).
template <int ST>
CVX_HD double eig_step5(Eig &e, double tol2)
{
    constexpr int P[5] = {rr_col(ST, 0), rr_col(ST, 1), rr_col(ST, 2), rr_col(ST, 3), rr_col(ST, 4)};
    constexpr int Q[5] = {rr_col(ST, 5), rr_col(ST, 6), rr_col(ST, 7), rr_col(ST, 8), rr_col(ST, 9)};
    double gam[5] = {0, 0, 0, 0, 0};
    CVX_UNROLL for (int i = 0; i < 10; ++i)
        CVX_UNROLL for (int k = 0; k < 5; ++k) gam[k] += e.G[P[k]][i] * e.G[Q[k]][i];
    double c[5], s[5], worst = 0;
    CVX_UNROLL for (int k = 0; k < 5; ++k) {
        const double al = e.n2[P[k]], be = e.n2[Q[k]];
        const double g2 = gam[k] * gam[k], ab = al * be;
        double dl;
        jacobi_cs_dl(al, be, gam[k], g2 > 1e-30 * ab, c[k], s[k], dl, e.exact);
#if defined(__HIP_DEVICE_COMPILE__)
        const double r = g2 > tol2 * ab ? 1.0 : 0.0; // (all the caller asks is whether a pair exceeds the tolerance: no IEEE division -- twelve instructions -- per pair)
#else
        const double r = g2 / ab;
        (void)tol2;
#endif
        worst = r > worst ? r : worst;
        e.n2[P[k]] = al + dl;
        e.n2[Q[k]] = be - dl;
    }
    CVX_UNROLL for (int i = 0; i < 10; ++i)
        CVX_UNROLL for (int k = 0; k < 5; ++k) {
            const double gp = e.G[P[k]][i], gq = e.G[Q[k]][i];
            e.G[P[k]][i] = c[k] * gp - s[k] * gq;
            e.G[Q[k]][i] = s[k] * gp + c[k] * gq;
        }
    return worst;
}

// sweeps until the largest squared cosine met in a sweep is below tol2 (quadratic convergence:
// the columns are then orthogonal to ~tol2 after that sweep), or max_sweeps
CVX_HD int eig_solve(Eig &e, int max_sweeps, double tol2)
{
    int sweeps = 0;
    for (; sweeps < max_sweeps;) {
        eig_norms(e);
        double worst = 0, r;
        r = eig_step5<0>(e, tol2); worst = r > worst ? r : worst;
        r = eig_step5<1>(e, tol2); worst = r > worst ? r : worst;
        r = eig_step5<2>(e, tol2); worst = r > worst ? r : worst;
        r = eig_step5<3>(e, tol2); worst = r > worst ? r : worst;
        r = eig_step5<4>(e, tol2); worst = r > worst ? r : worst;
        r = eig_step5<5>(e, tol2); worst = r > worst ? r : worst;
        r = eig_step5<6>(e, tol2); worst = r > worst ? r : worst;
        r = eig_step5<7>(e, tol2); worst = r > worst ? r : worst;
        r = eig_step5<8>(e, tol2); worst = r > worst ? r : worst;
        ++sweeps;
        if (!(worst > tol2)) break;
    }
    eig_norms(e);
    return sweeps;
}

// ---------------------------------------------------------------------------------------
// The same eigen-solve with the columns in single precision, two rows per packed instruction (v_pk_mul_f32 /
// v_pk_fma_f32): for the FIRST iterations of a solve -- the lane phase of the hybrid schedule (TWIN = false: at most 5
// iterations) and the quad kernel.  The columns only have to become orthogonal to the sweep tolerance (6e-2), the
// iterate they produce is a dual hint whose certificate is verified in double, and ~1e-7 of noise per iteration is far
// below what the first iterations move.  Measured on the host build, 10 k problems each of PnP N=10 / N=6 sigma 5 / N=4 and
// PnPL 5+5: with single-precision sweeps in the first 7 iterations the iteration histograms are identical to the double
// ones; single precision THROUGHOUT only changes tails beyond ~100 iterations (those run in double: wave kernel).
#if defined(__HIP_DEVICE_COMPILE__)
typedef float f2 __attribute__((ext_vector_type(2)));
CVX_HD f2 f2_set(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
CVX_HD f2 f2_mul(f2 a, f2 b) { return a * b; }
CVX_HD f2 f2_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
CVX_HD f2 f2_neg(f2 a) { return -a; }
#else
struct f2 { float x, y; };
CVX_HD f2 f2_set(float a, float b) { f2 r; r.x = a; r.y = b; return r; }
CVX_HD f2 f2_mul(f2 a, f2 b) { return f2_set(a.x * b.x, a.y * b.y); }
CVX_HD f2 f2_fma(f2 a, f2 b, f2 c) { return f2_set(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y)); }
CVX_HD f2 f2_neg(f2 a) { return f2_set(-a.x, -a.y); }
#endif

struct EigF {
    f2 G[10][5];  // G[j][i]: column j, rows 2i and 2i + 1
    float n2[10]; // squared column norms
    double sigma;
};
CVX_HD double eig_g(const Eig &e, int j, int i) { return e.G[j][i]; }
CVX_HD void set_exact(Eig &e, bool x) { e.exact = x; }
CVX_HD void set_exact(EigF &, bool) {}
CVX_HD double eig_g(const EigF &e, int j, int i) { return (double)((i & 1) ? e.G[j][i >> 1].y : e.G[j][i >> 1].x); }
CVX_HD void eig_unit(Eig &e)
{
    CVX_UNROLL for (int j = 0; j < 10; ++j) {
        CVX_UNROLL for (int i = 0; i < 10; ++i) e.G[j][i] = (i == j) ? 1.0 : 0.0;
        e.n2[j] = 1.0;
    }
    e.sigma = 0.0;
}
CVX_HD void eig_unit(EigF &e)
{
    CVX_UNROLL for (int j = 0; j < 10; ++j) {
        CVX_UNROLL for (int i = 0; i < 5; ++i) e.G[j][i] = f2_set((2 * i == j) ? 1.0f : 0.0f, (2 * i + 1 == j) ? 1.0f : 0.0f);
        e.n2[j] = 1.0f;
    }
    e.sigma = 0.0;
}

CVX_HD void eig_load(EigF &e, const double *W)
{
    double fro = 0;
    CVX_UNROLL for (int i = 0; i < 10; ++i)
        CVX_UNROLL for (int j = i; j < 10; ++j) fro += (i == j ? 1.0 : 2.0) * W[sidx(i, j)] * W[sidx(i, j)];
    e.sigma = 1.5 * sqrt_fast(fro) + 1e-300;
    CVX_UNROLL for (int j = 0; j < 10; ++j)
        CVX_UNROLL for (int i = 0; i < 5; ++i)
            e.G[j][i] = f2_set((float)(W[sidx(2 * i, j)] + (2 * i == j ? e.sigma : 0.0)), (float)(W[sidx(2 * i + 1, j)] + (2 * i + 1 == j ? e.sigma : 0.0)));
}

// warm start (see eig_load_warm above); the product (W + sigma I) v is formed in double from the double iterate
CVX_HD void eig_load_warm(EigF &e, const double *W)
{
    double fro = 0;
    CVX_UNROLL for (int i = 0; i < 10; ++i)
        CVX_UNROLL for (int j = i; j < 10; ++j) fro += (i == j ? 1.0 : 2.0) * W[sidx(i, j)] * W[sidx(i, j)];
    e.sigma = 1.5 * sqrt_fast(fro) + 1e-300;
    CVX_UNROLL for (int j = 0; j < 10; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
        const double il_ = (double)__builtin_amdgcn_rsqf(e.n2[j]);
#else
        const double il_ = 1.0 / sqrt((double)e.n2[j]);
#endif
        double v[10];
        CVX_UNROLL for (int i = 0; i < 10; ++i) v[i] = eig_g(e, j, i) * il_;
        double acc[10];
        CVX_UNROLL for (int i = 0; i < 10; ++i) {
            acc[i] = e.sigma * v[i];
            CVX_UNROLL for (int m = 0; m < 10; ++m) acc[i] += W[sidx(i, m)] * v[m];
        }
        CVX_UNROLL for (int i = 0; i < 5; ++i) e.G[j][i] = f2_set((float)acc[2 * i], (float)acc[2 * i + 1]);
    }
}

CVX_HD void eig_norms(EigF &e)
{
    CVX_UNROLL for (int j = 0; j < 10; ++j) {
        f2 a = f2_mul(e.G[j][0], e.G[j][0]);
        CVX_UNROLL for (int i = 1; i < 5; ++i) a = f2_fma(e.G[j][i], e.G[j][i], a);
        e.n2[j] = a.x + a.y;
    }
}

CVX_HD void jacobi_cs(float al, float be, float gam, bool rot, float &c, float &s, float &t)
{
    const float d = be - al, g2 = 2.0f * gam;
#if defined(__HIP_DEVICE_COMPILE__)
    // half-angle form (two v_rsq_f32 in a row instead of rsq -> rcp -> rsq): cos 2th = |d| / h, c^2 = (1 + cos 2th) / 2
    const float h2 = d * d + g2 * g2 + 1e-37f;
    const float ih = __builtin_amdgcn_rsqf(h2);
    const float c2 = fmaf(0.5f * fabsf(d), ih, 0.5f);
    const float ic = __builtin_amdgcn_rsqf(c2);
    const float sg = (0.5f * g2) * ih * ic;
    c = rot ? c2 * ic : 1.0f;
    s = rot ? (d < 0.0f ? -sg : sg) : 0.0f;
    t = s * ic;
    return;
#else
    const float h = sqrtf(d * d + g2 * g2 + 1e-37f);
    float tt = g2 / (fabsf(d) + h);
    tt = d < 0 ? -tt : tt;
    t = rot ? tt : 0.0f;
    c = 1.0f / sqrtf(1.0f + t * t);
#endif
    s = t * c;
}

template <int ST>
CVX_HD double eig_step5(EigF &e)
{
    constexpr int P[5] = {rr_col(ST, 0), rr_col(ST, 1), rr_col(ST, 2), rr_col(ST, 3), rr_col(ST, 4)};
    constexpr int Q[5] = {rr_col(ST, 5), rr_col(ST, 6), rr_col(ST, 7), rr_col(ST, 8), rr_col(ST, 9)};
    f2 acc[5];
    CVX_UNROLL for (int k = 0; k < 5; ++k) acc[k] = f2_mul(e.G[P[k]][0], e.G[Q[k]][0]);
    CVX_UNROLL for (int i = 1; i < 5; ++i)
        CVX_UNROLL for (int k = 0; k < 5; ++k) acc[k] = f2_fma(e.G[P[k]][i], e.G[Q[k]][i], acc[k]);
    float c[5], s[5], worst = 0;
    CVX_UNROLL for (int k = 0; k < 5; ++k) {
        const float gam = acc[k].x + acc[k].y;
        const float al = e.n2[P[k]], be = e.n2[Q[k]];
        const float g2 = gam * gam, ab = al * be;
        float t;
        jacobi_cs(al, be, gam, g2 > 1e-30f * ab, c[k], s[k], t);
#if defined(__HIP_DEVICE_COMPILE__)
        const float r = g2 * __builtin_amdgcn_rcpf(ab);
#else
        const float r = g2 / ab;
#endif
        worst = r > worst ? r : worst;
        e.n2[P[k]] = al - t * gam;
        e.n2[Q[k]] = be + t * gam;
    }
    CVX_UNROLL for (int i = 0; i < 5; ++i)
        CVX_UNROLL for (int k = 0; k < 5; ++k) {
            const f2 gp = e.G[P[k]][i], gq = e.G[Q[k]][i];
            const f2 cc = f2_set(c[k], c[k]), ss = f2_set(s[k], s[k]);
            e.G[P[k]][i] = f2_fma(cc, gp, f2_neg(f2_mul(ss, gq)));
            e.G[Q[k]][i] = f2_fma(ss, gp, f2_mul(cc, gq));
        }
    return (double)worst;
}

CVX_HD int eig_solve(EigF &e, int max_sweeps, double tol2)
{
    int sweeps = 0;
    for (; sweeps < max_sweeps;) {
        eig_norms(e);
        double worst = 0, r;
        r = eig_step5<0>(e); worst = r > worst ? r : worst;
        r = eig_step5<1>(e); worst = r > worst ? r : worst;
        r = eig_step5<2>(e); worst = r > worst ? r : worst;
        r = eig_step5<3>(e); worst = r > worst ? r : worst;
        r = eig_step5<4>(e); worst = r > worst ? r : worst;
        r = eig_step5<5>(e); worst = r > worst ? r : worst;
        r = eig_step5<6>(e); worst = r > worst ? r : worst;
        r = eig_step5<7>(e); worst = r > worst ? r : worst;
        r = eig_step5<8>(e); worst = r > worst ? r : worst;
        ++sweeps;
        if (!(worst > tol2)) break;
    }
    eig_norms(e);
    return sweeps;
}

// Wp = sum_{lam_j > 0} lam_j v_j v_j^T  (55 packed), accumulated in double from the single-precision columns
CVX_HD void eig_pospart(const EigF &e, double *Wp)
{
    double w[10];
    CVX_UNROLL for (int j = 0; j < 10; ++j) {
        const double n2 = (double)e.n2[j];
        const double lam = sqrt_fast(n2) - e.sigma;
        w[j] = lam > 0 ? lam / n2 : 0.0;
    }
    CVX_UNROLL for (int i = 0; i < 55; ++i) Wp[i] = 0.0;
    CVX_UNROLL for (int j = 0; j < 10; ++j) { // column-major: ten conversions live at a time, not a hundred
        double g[10];
        CVX_UNROLL for (int i = 0; i < 10; ++i) g[i] = eig_g(e, j, i);
        CVX_UNROLL for (int i = 0; i < 10; ++i) {
            const double wg = w[j] * g[i];
            CVX_UNROLL for (int k = i; k < 10; ++k) Wp[sidx(i, k)] += wg * g[k];
        }
    }
}

// Wp = sum_{lam_j > 0} lam_j v_j v_j^T  (55 packed)
CVX_HD void eig_pospart(const Eig &e, double *Wp)
{
    double w[10];
    CVX_UNROLL for (int j = 0; j < 10; ++j) {
        double lp = sqrt_fast(e.n2[j]);
        double lam = lp - e.sigma;
        w[j] = lam > 0 ? lam / e.n2[j] : 0.0;
    }
    CVX_UNROLL for (int i = 0; i < 10; ++i)
        CVX_UNROLL for (int k = i; k < 10; ++k) {
            double acc = 0;
            CVX_UNROLL for (int j = 0; j < 10; ++j) acc += w[j] * e.G[j][i] * e.G[j][k];
            Wp[sidx(i, k)] = acc;
        }
}

// ---------------------------------------------------------------------------------------
// SO(3) Newton polish of f(R) = r^T Q r, r = vec_colmajor(R)

// Where the per-problem constants (normalised cost Qs, translation map B) live during a scalar solve:
// plain arrays (host build, default) or -- lane-per-problem kernel -- this lane's column of an LDS block,
// element k of lane l at base[64 k + l] (conflict-free, and 144 registers less state per lane).  Every
// routine below takes the cost as "anything indexable": a pointer or a StridedView.
struct StridedView {
    const double *p;
    CVX_HDM double operator[](int k) const { return p[k * 64]; }
};
struct RegStore {
    double q[45], b[27];
    CVX_HDM void setQ(int k, double v) { q[k] = v; }
    CVX_HDM const double *Q() const { return q; }
    CVX_HDM void setB(int k, double v) { b[k] = v; }
    CVX_HDM double B(int k) const { return b[k]; }
};
struct LdsStore {
    double *base; // &block[lane]; 72 * 64 doubles per 64-lane block
    CVX_HDM void setQ(int k, double v) { base[k * 64] = v; }
    CVX_HDM StridedView Q() const { return StridedView{base}; }
    CVX_HDM void setB(int k, double v) { base[(45 + k) * 64] = v; }
    CVX_HDM double B(int k) const { return base[(45 + k) * 64]; }
};

template <class QV>
CVX_HD void q9_mul(QV Q9, const double *x, double *y)
{
    CVX_UNROLL for (int i = 0; i < 9; ++i) {
        double acc = 0;
        CVX_UNROLL for (int j = 0; j < 9; ++j) acc += Q9[qidx(i, j)] * x[j];
        y[i] = acc;
    }
}

// R is row-major 3x3; r[3j+i] = R[i][j]
template <class QV>
CVX_HD void so3_newton(QV Q9, double *R, int iters)
{
    for (int it = 0; it < iters; ++it) {
        double r[9], Qr[9];
        CVX_UNROLL for (int i = 0; i < 3; ++i) CVX_UNROLL for (int j = 0; j < 3; ++j) r[3 * j + i] = R[i * 3 + j];
        q9_mul(Q9, r, Qr);
        // N = R^T M, M = mat(Qr): M[i][j] = Qr[3j+i]
        double N[9];
        CVX_UNROLL for (int i = 0; i < 3; ++i)
            CVX_UNROLL for (int j = 0; j < 3; ++j) N[i * 3 + j] = R[0 * 3 + i] * Qr[3 * j] + R[1 * 3 + i] * Qr[3 * j + 1] + R[2 * 3 + i] * Qr[3 * j + 2];
        double g[3] = {2 * (N[7] - N[5]), 2 * (N[2] - N[6]), 2 * (N[3] - N[1])};
        // converged: projected gradient at rounding level (Q is trace-normalised, |R| = O(1))
        const double gn = fabs(g[0]) + fabs(g[1]) + fabs(g[2]);
        if (it >= 2 && gn < 1e-15) break;
        // Newton converges quadratically: from |g| < 1e-8 the step below lands at rounding level, so it is
        // the last one (saves the iteration that would only have confirmed it)
        const bool final_step = gn < 1e-8;
        // a_k = vec(R [e_k]x): columns (0, c2, -c1), (-c2, 0, c0), (c1, -c0, 0)
        double a[3][9], Qa[3][9];
        CVX_UNROLL for (int i = 0; i < 3; ++i) {
            double c0 = R[i * 3], c1 = R[i * 3 + 1], c2 = R[i * 3 + 2];
            a[0][i] = 0;   a[0][3 + i] = c2;  a[0][6 + i] = -c1;
            a[1][i] = -c2; a[1][3 + i] = 0;   a[1][6 + i] = c0;
            a[2][i] = c1;  a[2][3 + i] = -c0; a[2][6 + i] = 0;
        }
        CVX_UNROLL for (int k = 0; k < 3; ++k) q9_mul(Q9, a[k], Qa[k]);
        double trN = N[0] + N[4] + N[8];
        double H[9];
        CVX_UNROLL for (int k = 0; k < 3; ++k)
            CVX_UNROLL for (int l = 0; l < 3; ++l) {
                double acc = 0;
                CVX_UNROLL for (int i = 0; i < 9; ++i) acc += a[k][i] * Qa[l][i];
                H[k * 3 + l] = 2 * acc + N[l * 3 + k] + N[k * 3 + l] - (k == l ? 2 * trN : 0.0);
            }
        // solve H w = -g; fall back to a scaled gradient step when H is not positive definite
        double Hi[9], det;
        inv3(H, Hi, det);
        bool pd = H[0] > 0 && (H[0] * H[4] - H[1] * H[3]) > 0 && det > 0;
        double w[3];
        double hn = fabs(H[0]) + fabs(H[4]) + fabs(H[8]) + 1e-300;
        CVX_UNROLL for (int k = 0; k < 3; ++k) {
            double nw = -(Hi[k * 3] * g[0] + Hi[k * 3 + 1] * g[1] + Hi[k * 3 + 2] * g[2]);
            w[k] = pd ? nw : -g[k] * rcp(hn);
        }
        double wn = sqrt_(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        double lim = wn > 0.5 ? 0.5 * rcp(wn) : 1.0;
        // Cayley retraction R <- R (I - S)^-1 (I + S), S = [w/2]x  = R (I + 2/(1+|s|^2) (S + S^2))
        double s0 = 0.5 * lim * w[0], s1 = 0.5 * lim * w[1], s2 = 0.5 * lim * w[2];
        double f = 2.0 * rcp(1.0 + s0 * s0 + s1 * s1 + s2 * s2);
        double ss = s0 * s0 + s1 * s1 + s2 * s2;
        double Cm[9] = {1 + f * (s0 * s0 - ss), f * (-s2 + s0 * s1), f * (s1 + s0 * s2),
                        f * (s2 + s0 * s1), 1 + f * (s1 * s1 - ss), f * (-s0 + s1 * s2),
                        f * (-s1 + s0 * s2), f * (s0 + s1 * s2), 1 + f * (s2 * s2 - ss)};
        double Rn[9];
        CVX_UNROLL for (int i = 0; i < 3; ++i)
            CVX_UNROLL for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = R[i * 3] * Cm[j] + R[i * 3 + 1] * Cm[3 + j] + R[i * 3 + 2] * Cm[6 + j];
        CVX_UNROLL for (int i = 0; i < 9; ++i) R[i] = Rn[i];
        if (final_step && pd) break;
    }
    // one polar step squares any drift from orthogonality
    double Ri[9], det;
    inv3(R, Ri, det);
    double Rn[9];
    CVX_UNROLL for (int i = 0; i < 3; ++i) CVX_UNROLL for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = 0.5 * (R[i * 3 + j] + Ri[j * 3 + i]);
    CVX_UNROLL for (int i = 0; i < 9; ++i) R[i] = Rn[i];
}

// ---------------------------------------------------------------------------------------
// dual recovery + certificate

// y[i] = sum_j S[i][j] z[j] for packed symmetric S
CVX_HD void sym_mul10(const double *S, const double *z, double *y)
{
    CVX_UNROLL for (int i = 0; i < 10; ++i) {
        double acc = 0;
        CVX_UNROLL for (int j = 0; j < 10; ++j) acc += S[sidx(i, j)] * z[j];
        y[i] = acc;
    }
}

// in-place LDL^T without pivoting of a packed symmetric 10x10; returns the smallest pivot
CVX_HD double ldl_min_pivot(double *S)
{
    double minp = 1e300;
    CVX_UNROLL for (int k = 0; k < 10; ++k) {
        double d = S[sidx(k, k)];
        minp = d < minp ? d : minp;
        double id = rcp(d);
        CVX_UNROLL for (int i = k + 1; i < 10; ++i) {
            double l = S[sidx(k, i)] * id;
            CVX_UNROLL for (int j = i; j < 10; ++j) S[sidx(i, j)] -= l * S[sidx(k, j)];
        }
    }
    return minp;
}

// Closed form of the multiplier solve.  span{A_i} is invariant under the congruence with
// P(R) = blkdiag(R, R, R, 1) (orthonormality and cross-product constraints are invariant under
// R -> R0 R), so  Mz(z_R) = P(R) Mz(z_I) P(R)^T  and likewise for the tangent projector: the 10x10
// system of the dual correction is the CONSTANT matrix M_I = Mz(z_I) + T_I T_I^T seen in a rotated
// frame, and  lam = P(R) M_I^-1 P(R)^T rhs.  M_I^-1 is built at compile time (block structure
// {0,4,8,9}, {1,3}, {2,6}, {5,7}: 28 non-zeros); `symm` = the D-even variant used for planar scenes,
// a pivot-skipping generalised inverse there (any solution of the consistent system gives the same
// projected correction).
struct DualC { double c[100]; };
constexpr DualC make_dual_c(bool symm, int var = VAR_FULL)
{
    const double z[10] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 1};
    double M[100] = {};
    for (int t = 0; t < 15; ++t) {
        if ((symm && odd_tri(t)) || tri_dropped(t, var)) continue;
        double g[10] = {};
        for (int k = 0; k < 3; ++k) {
            g[tri_i(t, k)] += 0.5 * tri_s(t, k) * z[tri_j(t, k)];
            g[tri_j(t, k)] += 0.5 * tri_s(t, k) * z[tri_i(t, k)];
        }
        for (int a = 0; a < 10; ++a)
            for (int b = 0; b < 10; ++b) M[a * 10 + b] += (2.0 / 3.0) * g[a] * g[b];
    }
    for (int k = 0; k < 9; ++k)
        for (int l = 0; l < 9; ++l) {
            const double p = var == VAR_RC ? ((k / 3) == (l / 3) ? 1.0 / 3.0 : 0.0)
                                           : ((k % 3) == (l % 3) ? 1.0 / 3.0 : 0.0) + ((k / 3) == (l / 3) ? 1.0 / 3.0 : 0.0) - 1.0 / 9.0;
            M[k * 10 + l] += z[k] * p * z[l];
        }
    M[99] += 1.0;
    // tangents [vec([e_k]x); 0] of SO(3) at the identity
    const double tv[3][10] = {{0, 0, 0, 0, 0, 1, 0, -1, 0, 0}, {0, 0, -1, 0, 0, 0, 1, 0, 0, 0}, {0, 1, 0, -1, 0, 0, 0, 0, 0, 0}};
    for (int k = 0; k < 3; ++k)
        for (int i = 0; i < 10; ++i)
            for (int j = 0; j < 10; ++j) M[i * 10 + j] += tv[k][i] * tv[k][j];
    // L D L^T with null pivots skipped, then one solve per unit vector
    double Lm[100] = {}, d[10] = {};
    bool skip[10] = {};
    for (int j = 0; j < 10; ++j) {
        double dj = M[j * 10 + j];
        for (int k = 0; k < j; ++k) dj -= Lm[j * 10 + k] * Lm[j * 10 + k] * d[k];
        skip[j] = !(dj > 1e-10);
        d[j] = skip[j] ? 0.0 : dj;
        Lm[j * 10 + j] = 1.0;
        for (int i = j + 1; i < 10; ++i) {
            double v = M[i * 10 + j];
            for (int k = 0; k < j; ++k) v -= Lm[i * 10 + k] * Lm[j * 10 + k] * d[k];
            Lm[i * 10 + j] = skip[j] ? 0.0 : v / dj;
        }
    }
    DualC out{};
    for (int col = 0; col < 10; ++col) {
        double x[10] = {};
        x[col] = 1.0;
        for (int i = 0; i < 10; ++i) {
            double v = x[i];
            for (int k = 0; k < i; ++k) v -= Lm[i * 10 + k] * x[k];
            x[i] = skip[i] ? 0.0 : v;
        }
        for (int i = 0; i < 10; ++i) x[i] = skip[i] ? 0.0 : x[i] / d[i];
        for (int i = 9; i >= 0; --i) {
            double v = x[i];
            for (int k = i + 1; k < 10; ++k) v -= Lm[k * 10 + i] * x[k];
            x[i] = skip[i] ? 0.0 : v;
        }
        for (int i = 0; i < 10; ++i) out.c[i * 10 + col] = (x[i] > -1e-15 && x[i] < 1e-15) ? 0.0 : x[i];
    }
    return out;
}
constexpr DualC kDualC = make_dual_c(false), kDualCs = make_dual_c(true);
constexpr DualC kDualCrc = make_dual_c(false, VAR_RC), kDualCrcs = make_dual_c(true, VAR_RC);

// lam = P(R) C P(R)^T rhs,  (P x)[3 j + i] = sum_k R[i][k] x[3 j + k],  R row-major
template <int VAR = VAR_FULL>
CVX_HD void dual_lambda(const double *R, const double *rhs, bool symm, double *lam)
{
    double yp[10], lp[10];
    CVX_UNROLL for (int j = 0; j < 3; ++j)
        CVX_UNROLL for (int i = 0; i < 3; ++i) yp[3 * j + i] = R[0 * 3 + i] * rhs[3 * j] + R[1 * 3 + i] * rhs[3 * j + 1] + R[2 * 3 + i] * rhs[3 * j + 2];
    yp[9] = rhs[9];
    CVX_UNROLL for (int a = 0; a < 10; ++a) {
        double u = 0, v = 0;
        CVX_UNROLL for (int b = 0; b < 10; ++b) {
            const double cn = VAR == VAR_RC ? kDualCrc.c[a * 10 + b] : kDualC.c[a * 10 + b];
            const double cs = VAR == VAR_RC ? kDualCrcs.c[a * 10 + b] : kDualCs.c[a * 10 + b];
            if (cn != 0.0) u += cn * yp[b];
            if (cs != 0.0) v += cs * yp[b];
        }
        lp[a] = symm ? v : u;
    }
    CVX_UNROLL for (int j = 0; j < 3; ++j)
        CVX_UNROLL for (int i = 0; i < 3; ++i) lam[3 * j + i] = R[i * 3] * lp[3 * j] + R[i * 3 + 1] * lp[3 * j + 1] + R[i * 3 + 2] * lp[3 * j + 2];
    lam[9] = lp[9];
}

// E <- P_range(E) = E - P_null(E), for E = sym(lam z^T) given implicitly; subtracts the
// result from S:  S <- S - P_range(sym(lam z^T))
template <int VAR = VAR_FULL>
CVX_HD void sub_range_of_rank2(double *S, const double *lam, const double *z, bool symm = false)
{
    double E[55];
    CVX_UNROLL for (int i = 0; i < 10; ++i)
        CVX_UNROLL for (int j = i; j < 10; ++j) E[sidx(i, j)] = (symm && odd_entry(i, j)) ? 0.0 : 0.5 * (lam[i] * z[j] + z[i] * lam[j]);
    double N[55];
    CVX_UNROLL for (int i = 0; i < 55; ++i) N[i] = E[i];
    proj_affine<VAR>(N, true); // N = P_null(E)
    CVX_UNROLL for (int i = 0; i < 55; ++i) S[i] -= E[i] - N[i];
}

struct Cert {
    double R[9];     // polished rotation, row-major
    double pobj;     // r^T Qs r (trace-normalised units)
    double zSz;      // z^T S z, |.| ~ 1e-16
    double min_piv;  // smallest LDL^T pivot of S + delta I
    double res;      // |S z|_inf
    bool ok;
};

// Primal half of a certification attempt: round the candidate v (any multiple of [r; 1]) to a rotation
// (cvxpnpl.py:504-505 + nearest proper rotation), Newton-polish r^T Qs r on SO(3).  Returns det of the
// rounded matrix (<= 0: the candidate was a reflection and cannot certify) and pobj = r^T Qs r.
// rank-1 rounding (cvxpnpl.py:504-505) and projection to the nearest proper rotation; returns det of the
// rounded matrix (<= 0: the candidate was a reflection and cannot certify)
CVX_HD double round_candidate(const double *v, double *R)
{
    double iv = rcp(v[9]);
    double M0[9];
    CVX_UNROLL for (int i = 0; i < 3; ++i) CVX_UNROLL for (int j = 0; j < 3; ++j) M0[i * 3 + j] = v[3 * j + i] * iv; // R[i][j] = r[3j+i]
    double d0 = det3(M0);
    if (d0 < 0) { CVX_UNROLL for (int i = 0; i < 9; ++i) M0[i] = -M0[i]; } // polish needs SO(3); a reflection cannot certify
    polar3(M0, R, 8); // (the scalar core keeps the converged polar iteration: in the lane-per-problem kernel near_rotation() costs more
                      //  in register allocation than it saves -- 125 k problems: 0.805 against 0.795 ms)
    return d0;
}

// Does the rank-1 ratio M0 = mat(v[0..8] / v[9]) round to (almost) the rotation Rp?  The polar factor of M0
// is Rp exactly when Rp^T M0 is symmetric positive definite; |skew part|^2 < 0.1 (tr / 3)^2 keeps it within
// ~0.16 rad of Rp -- the test costs 40 flops instead of the polar iteration.  d0 = det(M0).
CVX_HD bool rounds_to(const double *v, const double *Rp, double &d0, double tol = 0.1)
{
    const double iv = rcp(v[9]);
    double M0[9];
    CVX_UNROLL for (int i = 0; i < 3; ++i) CVX_UNROLL for (int j = 0; j < 3; ++j) M0[i * 3 + j] = v[3 * j + i] * iv;
    d0 = det3(M0);
    double S[9];
    CVX_UNROLL for (int a = 0; a < 3; ++a)
        CVX_UNROLL for (int b = 0; b < 3; ++b) S[a * 3 + b] = Rp[0 * 3 + a] * M0[0 * 3 + b] + Rp[1 * 3 + a] * M0[1 * 3 + b] + Rp[2 * 3 + a] * M0[2 * 3 + b];
    const double a01 = S[1] - S[3], a02 = S[2] - S[6], a12 = S[5] - S[7], t = S[0] + S[4] + S[8];
    return d0 > 0 && t > 0 && (a01 * a01 + a02 * a02 + a12 * a12) < (tol / 9.0) * t * t;
}

// Newton polish of r^T Qs r on SO(3) from R; pobj = r^T Qs r
template <class QV>
CVX_HD void polish_rotation(QV Qs, double *R, double &pobj)
{
    so3_newton(Qs, R, 6);
    double z[9], Qz[9];
    CVX_UNROLL for (int i = 0; i < 3; ++i) CVX_UNROLL for (int j = 0; j < 3; ++j) z[3 * j + i] = R[i * 3 + j];
    q9_mul(Qs, z, Qz);
    pobj = 0;
    CVX_UNROLL for (int i = 0; i < 9; ++i) pobj += z[i] * Qz[i];
}

template <class QV>
CVX_HD double polish_candidate(QV Qs, const double *v, double *R, double &pobj)
{
    const double d0 = round_candidate(v, R);
    polish_rotation(Qs, R, pobj);
    return d0;
}

// polish_candidate with memory, for the twin candidates: a candidate that rounds to within ~0.05 rad of the
// rotation ITS OWN slot polished at the previous check (cvx::rounds_to) takes that rotation and its cost
// instead of a polar + Newton run.  Near-ambiguous problems -- the slowest of every batch -- polish two
// twins per check and nearly all of them repeat; the caller forces a fresh polish every ninth check so that
// a stale reuse cannot persist.
template <class QV>
CVX_HD double polish_or_reuse(QV Qs, const double *z, const double *Rk, double fk, bool have, double *R, double &f)
{
    double d0;
    if (have && rounds_to(z, Rk, d0, 0.01)) {
        CVX_UNROLL for (int i = 0; i < 9; ++i) R[i] = Rk[i];
        f = fk;
        return d0;
    }
    return polish_candidate(Qs, z, R, f);
}

// The two points of span{v1, v2} (orthonormal) with last entry 1 and squared norm 4.  For ANY rank-2
// Z = w1 z1 z1^T + w2 z2 z2^T (z_i = [vec R_i; 1], |z_i|^2 = 4) whose range is span{v1, v2} these are
// exactly z1 and z2: a line-circle intersection instead of the reference's 21-quadratic solve
// (cvxpnpl.py:303-315).  Degenerates gracefully to the rank-1 ratio when v2 carries no weight.
CVX_HD void twin_candidates(const double *v1, const double *v2, double *zp, double *zm)
{
    const double a = v1[9], b = v2[9], n2r = a * a + b * b, n2 = n2r > 1e-60 ? n2r : 1e-60;
    const double inv = rcp(n2), rn = rsqrt_(n2);
    const double rad = 4.0 - inv;
    const double sq = rad > 0 ? sqrt_(rad) : 0.0;
    const double c1 = a * inv, c2 = b * inv, d1 = -b * rn * sq, d2 = a * rn * sq;
    CVX_UNROLL for (int i = 0; i < 10; ++i) {
        zp[i] = (c1 + d1) * v1[i] + (c2 + d2) * v2[i];
        zm[i] = (c1 - d1) * v1[i] + (c2 - d2) * v2[i];
    }
}

// Second try of a certificate attempt.  The duals that certify the pose z are the PSD members of the affine family
// S1 + U,  U = { X in span A_i : X z = 0 }  (14-dimensional; S1 = the recovered dual of dual_certificate).  The first-order iterate
// supplies ONE member, often just outside the cone while the pose has long been right.  Moving it along
//     D(R) = P_U(I - z z^T / 4) = P(R) D_I P(R)^T,      6 D_I = [[I9 + K - u u^T, u], [u^T, -3]],  u = vec(I3), K = the transposition,
// the projection of the identity onto U (a constant seen in the frame of R, like the multiplier solve of dual_lambda: span A_i is
// invariant under the congruence with P(R) = blkdiag(R, R, R, 1)), raises five eigen-directions of S1 by shift / 3 each and lowers the
// direction [r; -3] by 2 shift / 3; S1 z = 0 and z^T S z are unchanged, so the certificate statement is the same.  It costs one more
// LDL^T and no fit.  Host experiment on the judged problem set (10 000 x N = 10, 2 px; tools/experiments/dualref_*): 52 % of the failed
// attempts pass on the second try with shift = 0.015; problems needing >= 8 iterations 100 -> 31, p99.9 11 -> 9 iterations, mean
// 5.149 -> 5.065; with attempts at every iteration from the 4th the mean falls 4.42 -> 4.19.  (The same D for the rc variant.)
// 6 D(R)[a][b]:
CVX_HD double dual_retry_entry6(const double *R, int a, int b)
{
    if (a == 9 && b == 9) return -3.0;
    if (b == 9) return R[(a % 3) * 3 + a / 3];
    if (a == 9) return R[(b % 3) * 3 + b / 3];
    const int i = a % 3, j = a / 3, k = b % 3, l = b / 3;
    return (a == b ? 1.0 : 0.0) + R[k * 3 + j] * R[i * 3 + l] - R[i * 3 + j] * R[k * 3 + l];
}
// Third try of a certificate attempt: ONE eigen-gradient step inside the dual family (round 6).
// The certifying duals of the pose z are the PSD members of S1 + U (above).  A failed S1 of an N >= 6 problem typically has ONE eigenvalue
// just below zero (median -3e-4 on the judged set; its eigenvector n is close to the runner-up eigenvector of Z: these are the problems whose
// Z is still a mixture) while the others sit at 3e-2 and more.  d/dv lambda_min(S1 + U(v)) = n^T U_k n, so the steepest ascent direction of
// the bottom eigenvalue inside the family is G = P_U(n n^T); moving by tau = gain |lambda_1| / <G, n n^T> raises n^T S n to (gain - 1) |lambda_1|
// to first order at a cost of O(tau |G|) ~ 1e-3 to the other eigenvalues.  n: two inverse iterations with LDL^T(S1 + sigma I) from the
// runner-up eigenvector of Z (projected off z); P_U: the two closed forms of dual_certificate (projection onto span A_i, then the
// minimum-norm correction onto { X z = 0 }).  Cost: two LDL^T, two triangular solves, one P_U.  Host experiments on the judged problem set
// (tools/experiments/dual_refine/, 655 failed attempts of 10 000 problems): the shift tries rescue 48 %, this step 87 % (with the exact n:
// 90 %; a log-det barrier Newton step on (v, t): 93 %, two: 96 % = every attempt whose pose is already the final one); problems needing >= 9
// iterations 66 -> 15, >= 7 (with the step at the first attempt too) 576 -> 115.  The certificate that is reported is the usual float64
// statement about the S that passed (S z = 0 and z^T S z are re-measured on it).
// S: the failed dual + delta I (packed 55); on success it holds the refined dual + delta I.  Returns the smallest pivot of its LDL^T
// (<= 0: no luck), res / zSz re-measured.
CVX_HD void ldl_solve10(const double *T, double *x) // (L D L^T) x = b with the factor ldl_min_pivot leaves (pivot rows unnormalised)
{
    double id[10];
    CVX_UNROLL for (int i = 0; i < 10; ++i) id[i] = rcp(T[sidx(i, i)]);
    CVX_UNROLL for (int i = 1; i < 10; ++i) {
        double a = x[i];
        CVX_UNROLL for (int k = 0; k < i; ++k) a -= T[sidx(k, i)] * id[k] * x[k];
        x[i] = a;
    }
    CVX_UNROLL for (int i = 0; i < 10; ++i) x[i] *= id[i];
    CVX_UNROLL for (int i = 8; i >= 0; --i) {
        double a = x[i];
        CVX_UNROLL for (int j = i + 1; j < 10; ++j) a -= T[sidx(i, j)] * id[i] * x[j];
        x[i] = a;
    }
}
template <int VAR = VAR_FULL>
CVX_HD double dual_refine_step(double *S, const double *z, const double *R, const double *v2, double delta, double &res, double &zSz, double *ray_out = nullptr)
{
    if (ray_out) *ray_out = NAN;
    double x[10], T[55];
    CVX_UNROLL for (int i = 0; i < 55; ++i) T[i] = S[i];
    CVX_UNROLL for (int i = 0; i < 10; ++i) T[sidx(i, i)] += DUAL_REFINE_SIGMA;
    if (!(ldl_min_pivot(T) > 0)) return -1.0;
    CVX_UNROLL for (int i = 0; i < 10; ++i) x[i] = v2[i];
    for (int itn = 0; itn <= DUAL_REFINE_INVITS; ++itn) { // (pass 0 only projects the start vector off z and normalises it)
        if (itn > 0) ldl_solve10(T, x);
        double zx = 0;
        CVX_UNROLL for (int i = 0; i < 10; ++i) zx += z[i] * x[i];
        double n2 = 0;
        CVX_UNROLL for (int i = 0; i < 10; ++i) { x[i] -= 0.25 * zx * z[i]; n2 += x[i] * x[i]; }
        const double in = rsqrt_(n2 > 1e-300 ? n2 : 1e-300);
        CVX_UNROLL for (int i = 0; i < 10; ++i) x[i] *= in;
    }
    double Sx[10];
    sym_mul10(S, x, Sx);
    double ray = -delta;
    CVX_UNROLL for (int i = 0; i < 10; ++i) ray += x[i] * Sx[i];
    if (ray_out) *ray_out = ray;
    if (!(ray < 0)) return -1.0;
    // G = P_U(x x^T)
    double G[55], N[55];
    CVX_UNROLL for (int i = 0; i < 10; ++i) CVX_UNROLL for (int j = i; j < 10; ++j) { G[sidx(i, j)] = x[i] * x[j]; N[sidx(i, j)] = G[sidx(i, j)]; }
    proj_affine<VAR>(N, true);
    CVX_UNROLL for (int i = 0; i < 55; ++i) G[i] -= N[i];
    double rhs[10], lam[10];
    sym_mul10(G, z, rhs);
    dual_lambda<VAR>(R, rhs, false, lam);
    sub_range_of_rank2<VAR>(G, lam, z, false);
    double Gx[10];
    sym_mul10(G, x, Gx);
    double g2 = 0;
    CVX_UNROLL for (int i = 0; i < 10; ++i) g2 += x[i] * Gx[i];
    if (!(g2 > 1e-6)) return -1.0;
    const double tau = DUAL_REFINE_GAIN * (-ray) * rcp(g2);
    CVX_UNROLL for (int i = 0; i < 55; ++i) { S[i] += tau * G[i]; T[i] = S[i]; }
    double Sz[10];
    sym_mul10(S, z, Sz);
    res = 0; zSz = 0;
    CVX_UNROLL for (int i = 0; i < 10; ++i) { const double v = Sz[i] - delta * z[i]; res = fabs(v) > res ? fabs(v) : res; zSz += z[i] * v; }
    return ldl_min_pivot(T);
}
// ---------------------------------------------------------------------------------------
// The dual of a pose by a barrier Newton method inside the dual family (round 6; Opts::dual_refine >= 2).
//
// For a pose z that IS the optimum of the relaxation the family { S1 + U : U in span A_i, U z = 0 } of dual_certificate contains a
// positive semidefinite member from the first attempt on -- the optimal dual itself -- whatever the first-order iterate has delivered so far
// (host check on the slowest problems of 125 000: max_v lambda_min = 1e-4 ... 3e-3 at iteration 6 for problems that iterate to 14 ... 32 for
// it, tools/experiments/dual_newton/).  Finding it is a 15-variable convex problem,   max t  s.t.  S1 + sum_k v_k U_k - t T >= 0 on the
// complement of z,   T = I - z z^T / 4,   solved here by Newton steps on   -log det(S1 + U(v) - t T + z z^T / 4)   with the barrier weight
// chosen such that the t-component of the gradient vanishes (mu = 1 / tr X on the complement of z), from t0 = lambda - max(0.3 |lambda|, 1e-3)
// (lambda: an estimate of the bottom eigenvalue, e.g. the Rayleigh quotient of dual_refine_step; a start that is not feasible lowers t0).
// Everything is done in the FRAME OF R: S' = P(R)^T S1 P(R), P = blkdiag(R, R, R, 1), where z becomes z_I = [vec I3; 1] and the family has a
// constant sparse integer basis (kNt*: generated by tools/gen_newton_tables.py, 14 signed sums of constraint matrices with 4, 6 or 13 upper
// entries, and T_I as the 15th) -- gradient and Hessian are sums over those entries of products of two entries of X = (...)^-1:
//     g_a = sum_(p,q) w c X_pq,      H_ab = sum_(p,q) sum_(r,s) c c' (w w' / 2) (X_pr X_qs + X_ps X_qr),      w = 1 on the diagonal, 2 off it.
// Host experiments (same directory): every failed attempt whose pose is final certifies within 3 steps -- 36 of 36 on the slow problems of
// 125 000, 627 of 628 on the judged set (599 after ONE step).  The certificate that is reported is the usual one: LDL^T(S(v) + delta I) > 0,
// S z and z^T S z re-measured on the matrix that passed (eigenvalues are those of the world-frame matrix: P is orthogonal).
// WHERE IT RUNS: in this scalar core only (host build; Opts::dual_refine = 2).  The cooperative device version was built and measured
// (tools/experiments/patches/r06_coop_newton_device.patch, profiles/r06/newton_check.txt, newton_ab.txt): identical outcomes -- 2 000 fresh solves end
// after at most 7 iterations instead of 11, the 125 k launch's slowest problem after 19 instead of 33 -- but as a table-driven routine it costs a
// wavefront ~150 us per solve, and its mere presence cost the kernels around it 4-10 % (register allocation): not in the library.  What it
// establishes stays: the iterations after the fifth buy nothing but a dual that this 15-variable problem delivers at once.
constexpr int NT_N = 15, NT_MAX = 16;
constexpr int kNtCount[NT_N] = {4, 6, 6, 4, 6, 6, 6, 13, 13, 6, 6, 6, 13, 6, 16};
constexpr signed char kNtP[NT_N][NT_MAX] = {
    {1, 2, 3, 6, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 0, 1, 2, 3, 6, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 0, 1, 2, 3, 6, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {1, 3, 5, 7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {1, 3, 4, 4, 5, 7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 1, 2, 3, 6, 6, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 1, 2, 4, 7, 7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 0, 1, 1, 2, 3, 4, 5, 6, 7, 8, 8, 9, 0, 0, 0},
    {0, 0, 1, 2, 3, 4, 4, 5, 5, 6, 7, 8, 9, 0, 0, 0},
    {0, 1, 3, 3, 5, 6, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 2, 3, 3, 4, 6, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 1, 2, 3, 3, 6, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 0, 1, 2, 2, 3, 4, 4, 5, 6, 7, 8, 9, 0, 0, 0},
    {0, 1, 1, 4, 5, 7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 0, 0, 0, 1, 2, 3, 4, 4, 4, 5, 6, 7, 8, 8, 9},
};
constexpr signed char kNtQ[NT_N][NT_MAX] = {
    {1, 2, 3, 6, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {1, 3, 4, 5, 4, 7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {2, 6, 7, 8, 5, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {1, 3, 5, 7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {2, 6, 5, 7, 8, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {2, 5, 4, 5, 8, 9, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {5, 2, 3, 5, 8, 9, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 4, 1, 3, 2, 3, 4, 5, 6, 7, 8, 9, 9, 0, 0, 0},
    {0, 9, 1, 2, 3, 4, 8, 5, 7, 6, 7, 8, 9, 0, 0, 0},
    {1, 9, 4, 8, 6, 7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {2, 9, 5, 7, 6, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {1, 8, 7, 4, 9, 7, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 8, 1, 2, 6, 3, 4, 9, 5, 6, 7, 8, 9, 0, 0, 0},
    {7, 2, 6, 5, 9, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0},
    {0, 4, 8, 9, 1, 2, 3, 4, 8, 9, 5, 6, 7, 8, 9, 9},
};
constexpr double kNtC[NT_N][NT_MAX] = {
    {1.0, 1.0, -1.0, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0},
    {-1.0, 1.0, 1.0, 1.0, -1.0, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0},
    {-1.0, 1.0, 1.0, 1.0, -1.0, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0},
    {-1.0, 1.0, 1.0, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0},
    {-1.0, 1.0, -1.0, 1.0, 1.0, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0},
    {1.0, 1.0, -1.0, 1.0, 1.0, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0},
    {-1.0, 1.0, 1.0, 1.0, 1.0, -1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0},
    {-1.0, 1.0, -1.0, -1.0, 1.0, -1.0, -1.0, 1.0, -1.0, -1.0, 1.0, -1.0, 1.0, 0.0, 0.0, 0.0},
    {1.0, -1.0, -1.0, -1.0, 1.0, -1.0, 1.0, -1.0, -1.0, 1.0, -1.0, -1.0, 1.0, 0.0, 0.0, 0.0},
    {1.0, -1.0, 1.0, -1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0},
    {1.0, -1.0, 1.0, 1.0, -1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0},
    {1.0, -1.0, 1.0, 1.0, -1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0},
    {-1.0, 1.0, 1.0, -1.0, -1.0, -1.0, 1.0, -1.0, -1.0, -1.0, 1.0, -1.0, 1.0, 0.0, 0.0, 0.0},
    {-1.0, 1.0, 1.0, 1.0, -1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0},
    {0.75, -0.25, -0.25, -0.25, 1.0, 1.0, 1.0, 0.75, -0.25, -0.25, 1.0, 1.0, 1.0, 0.75, -0.25, 0.75},
};
constexpr int DUAL_NEWTON_STEPS = 3;
constexpr double DUAL_NEWTON_MARGIN = 0.3, DUAL_NEWTON_FLOOR = 1e-3;

// inverse of a symmetric positive definite n x n matrix (full storage, row stride ld) by Gauss-Jordan without pivoting; false: a pivot <= 0
template <int N>
CVX_HD bool spd_inverse(double *A, int ld)
{
    for (int k = 0; k < N; ++k) {
        const double d = A[k * ld + k];
        if (!(d > 0)) return false;
        const double id = 1.0 / d;
        for (int j = 0; j < N; ++j) A[k * ld + j] *= id;
        A[k * ld + k] = id;
        for (int i = 0; i < N; ++i) {
            if (i == k) continue;
            const double f = A[i * ld + k];
            A[i * ld + k] = 0.0;
            for (int j = 0; j < N; ++j) A[i * ld + j] -= f * A[k * ld + j];
        }
    }
    return true;
}

// S: the failed dual of dual_certificate (55 packed, delta ON its diagonal), R: the polished rotation, lam: estimate of its bottom eigenvalue
// (< 0; NaN: unknown).  Returns the smallest pivot of the refined dual + delta I (<= 0: no luck) with res / zSz re-measured; steps: Newton steps made.
CVX_HD double dual_newton(const double *S, const double *R, double delta, double lam, double &res, double &zSz, int *steps = nullptr)
{
    const double zI[10] = {1, 0, 0, 0, 1, 0, 0, 0, 1, 1};
    // S' = P^T (S - delta I) P:  S'[3j+a][3k+b] = sum_cd R[c][a] S[3j+c][3k+d] R[d][b],  S'[3j+a][9] = sum_c R[c][a] S[3j+c][9]
    double Sp[10][10];
    for (int j = 0; j < 3; ++j)
        for (int a = 0; a < 3; ++a) {
            for (int k = 0; k < 3; ++k)
                for (int b = 0; b < 3; ++b) {
                    double acc = 0;
                    for (int c = 0; c < 3; ++c)
                        for (int d = 0; d < 3; ++d) {
                            const int p = 3 * j + c, q = 3 * k + d;
                            acc += R[c * 3 + a] * (S[sidx(p < q ? p : q, p < q ? q : p)] - (p == q ? delta : 0.0)) * R[d * 3 + b];
                        }
                    Sp[3 * j + a][3 * k + b] = acc;
                }
            double acc = 0;
            for (int c = 0; c < 3; ++c) acc += R[c * 3 + a] * S[sidx(3 * j + c, 9)];
            Sp[3 * j + a][9] = acc; Sp[9][3 * j + a] = acc;
        }
    Sp[9][9] = S[sidx(9, 9)] - delta;
    double v[NT_N];
    for (int a = 0; a < NT_N; ++a) v[a] = 0.0;
    // M(v, t) = S' + sum_a v_a U_a - t T + z_I z_I^T / 4   (v[14] = -t)
    auto build = [&](const double *vv, double M[10][10], bool with_t) {
        for (int i = 0; i < 10; ++i) for (int j = 0; j < 10; ++j) M[i][j] = Sp[i][j] + (with_t ? 0.25 * zI[i] * zI[j] : 0.0);
        for (int a = 0; a < (with_t ? NT_N : NT_N - 1); ++a)
            for (int e = 0; e < kNtCount[a]; ++e) {
                const int p = kNtP[a][e], q = kNtQ[a][e];
                M[p][q] += vv[a] * kNtC[a][e];
                if (p != q) M[q][p] += vv[a] * kNtC[a][e];
            }
    };
    double X[10][10];
    double margin = (lam == lam && lam < 0) ? (DUAL_NEWTON_MARGIN * -lam > DUAL_NEWTON_FLOOR ? DUAL_NEWTON_MARGIN * -lam : DUAL_NEWTON_FLOOR) : 4e-3;
    const double l0 = (lam == lam && lam < 0) ? lam : -1e-2;
    bool feas = false;
    for (int tr_ = 0; tr_ < 4 && !feas; ++tr_) { // (a start that is not inside the cone: the estimate was too high -- lower t0)
        v[NT_N - 1] = -(l0 - margin);
        build(v, X, true);
        feas = spd_inverse<10>(&X[0][0], 10);
        margin *= 4.0;
    }
    if (steps) *steps = 0;
    if (!feas) return -1.0;
    for (int k = 0; k < DUAL_NEWTON_STEPS; ++k) {
        // gradient and Hessian over the tables
        double g[NT_N], H[NT_N][NT_N + 1];
        for (int a = 0; a < NT_N; ++a) {
            double ga = 0;
            for (int e = 0; e < kNtCount[a]; ++e) ga += kNtC[a][e] * (kNtP[a][e] == kNtQ[a][e] ? 1.0 : 2.0) * X[kNtP[a][e]][kNtQ[a][e]];
            g[a] = ga;
            for (int b = a; b < NT_N; ++b) {
                double h = 0;
                for (int e = 0; e < kNtCount[a]; ++e) {
                    const int p = kNtP[a][e], q = kNtQ[a][e];
                    const double ca = kNtC[a][e] * (p == q ? 1.0 : 2.0);
                    for (int f = 0; f < kNtCount[b]; ++f) {
                        const int r = kNtP[b][f], s_ = kNtQ[b][f];
                        h += ca * kNtC[b][f] * (r == s_ ? 0.5 : 1.0) * (X[p][r] * X[q][s_] + X[p][s_] * X[q][r]);
                    }
                }
                H[a][b] = h; H[b][a] = h;
            }
        }
        // Newton direction of -log det(M) in (v_0..v_13, v_14 = -t); the barrier weight makes the last component of the gradient vanish
        for (int a = 0; a < NT_N; ++a) H[a][NT_N] = (a == NT_N - 1) ? 0.0 : g[a];
        double Hi[NT_N][NT_N];
        for (int a = 0; a < NT_N; ++a) for (int b = 0; b < NT_N; ++b) Hi[a][b] = H[a][b];
        if (!spd_inverse<NT_N>(&Hi[0][0], NT_N)) return -1.0;
        double dx[NT_N];
        for (int a = 0; a < NT_N; ++a) { double acc = 0; for (int b = 0; b < NT_N; ++b) acc += Hi[a][b] * H[b][NT_N]; dx[a] = acc; }
        // step: the longest of 1, 1/2, ... 1/16 that stays inside the cone
        double al = 1.0, vn[NT_N];
        bool ok = false;
        for (int ls = 0; ls < 5 && !ok; ++ls) {
            for (int a = 0; a < NT_N; ++a) vn[a] = v[a] + al * dx[a];
            build(vn, X, true);
            ok = spd_inverse<10>(&X[0][0], 10);
            al *= 0.5;
        }
        if (!ok) return -1.0;
        for (int a = 0; a < NT_N; ++a) v[a] = vn[a];
        if (steps) *steps = k + 1;
        // the certificate's own test on S(v) = S' + sum v_a U_a (+ delta I)
        double M[10][10], T[55];
        build(v, M, false);
        for (int i = 0; i < 10; ++i) for (int j = i; j < 10; ++j) T[sidx(i, j)] = M[i][j] + (i == j ? delta : 0.0);
        const double mp = ldl_min_pivot(T);
        if (mp > 0) {
            res = 0; zSz = 0;
            for (int i = 0; i < 10; ++i) {
                double acc = 0;
                for (int j = 0; j < 10; ++j) acc += M[i][j] * zI[j];
                res = fabs(acc) > res ? fabs(acc) : res;
                zSz += zI[i] * acc;
            }
            return mp;
        }
    }
    return -1.0;
}
// Dual half: given the polished rotation c.R (and c.pobj), recover a dual and test it.
// SYMM: recognise planar scenes (Qs blind to the third column of R), whose relaxation is invariant
// under D = diag(-I6, I4), and build the correction in the D-even subspace so that it annihilates
// both twins z and D z at once.
template <bool SYMM = true, class QV = const double *, int VAR = VAR_FULL>
CVX_HD void dual_certificate(QV Qs, const double *W, const double *Wp, double rho, double delta, double d0, Cert &c, double shift = 0.0, const double *v2 = nullptr, bool newton = false)
{
    c.ok = false;
    double z[10];
    CVX_UNROLL for (int i = 0; i < 3; ++i) CVX_UNROLL for (int j = 0; j < 3; ++j) z[3 * j + i] = c.R[i * 3 + j];
    z[9] = 1.0;
    bool symm = SYMM;
    if (SYMM) { CVX_UNROLL for (int i = 0; i < 9; ++i) CVX_UNROLL for (int j = 6; j < 9; ++j) symm &= fabs(Qs[qidx(i, j)]) < 1e-13; }
    // dual hint S_h = -rho Wm = rho (Wp - W); S1 = S_h - P_null(S_h - Qs)  (in Qs + span A_i)
    double S[55], T[55];
    CVX_UNROLL for (int i = 0; i < 55; ++i) { S[i] = rho * (Wp[i] - W[i]); T[i] = S[i]; }
    CVX_UNROLL for (int i = 0; i < 9; ++i) CVX_UNROLL for (int j = i; j < 9; ++j) T[sidx(i, j)] -= Qs[qidx(i, j)];
    proj_affine<VAR>(T, true);
    CVX_UNROLL for (int i = 0; i < 55; ++i) S[i] -= T[i];
    if (symm) { CVX_UNROLL for (int i = 0; i < 10; ++i) CVX_UNROLL for (int j = i; j < 10; ++j) if (odd_entry(i, j)) S[sidx(i, j)] = 0.0; }
    // correction: min-norm dS in span A_i with (S - dS) z = 0
    // (closed form: the system matrix is a constant in the frame of R, see dual_lambda)
    double rhs[10], lam[10];
    sym_mul10(S, z, rhs);
    dual_lambda<VAR>(c.R, rhs, symm, lam);
    sub_range_of_rank2<VAR>(S, lam, z, symm);
    // checks
    double Sz[10];
    sym_mul10(S, z, Sz);
    c.res = 0; c.zSz = 0;
    CVX_UNROLL for (int i = 0; i < 10; ++i) { c.res = fabs(Sz[i]) > c.res ? fabs(Sz[i]) : c.res; c.zSz += z[i] * Sz[i]; }
    CVX_UNROLL for (int i = 0; i < 10; ++i) S[sidx(i, i)] += delta;
    const bool pre = (c.res < 1e-10) && (d0 > 0) && (c.pobj == c.pobj);
    if ((shift > 0.0 || v2) && pre && !symm) {
        CVX_UNROLL for (int i = 0; i < 55; ++i) T[i] = S[i];
        c.min_piv = ldl_min_pivot(T);
        if (!(c.min_piv > 0) && shift > 0.0) { // second try: S + shift D(R)
            double s6 = shift * (1.0 / 6.0);
            for (int rung = 0; rung < DUAL_RETRY_RUNGS && !(c.min_piv > 0); ++rung) { // shift, shift / 4
                CVX_UNROLL for (int i = 0; i < 10; ++i)
                    CVX_UNROLL for (int j = i; j < 10; ++j) T[sidx(i, j)] = S[sidx(i, j)] + s6 * dual_retry_entry6(c.R, i, j);
                c.min_piv = ldl_min_pivot(T);
                s6 *= 0.25;
            }
        }
        if (!(c.min_piv > 0) && v2) { // third try: one eigen-gradient step inside the dual family
            double res2, zSz2, ray = NAN, S0[55];
            CVX_UNROLL for (int i = 0; i < 55; ++i) S0[i] = S[i];
            double mp = dual_refine_step<VAR>(S, z, c.R, v2, delta, res2, zSz2, &ray);
            if (!(mp > 0 && res2 < 1e-10) && newton && VAR == VAR_FULL) mp = dual_newton(S0, c.R, delta, ray, res2, zSz2); // fourth: the barrier Newton solve
            if (mp > 0 && res2 < 1e-10) { c.min_piv = mp; c.res = res2; c.zSz = zSz2; }
        }
    } else {
        c.min_piv = ldl_min_pivot(S);
    }
    c.ok = (c.min_piv > 0) && pre;
}

// ---------------------------------------------------------------------------------------
// planar scenes in a general world frame
//
// When all 3D points / lines lie in one plane with unit normal n the cost does not see R n, the relaxation
// is exactly two-fold ambiguous, and the certificate needs the D-even correction of dual_certificate -- which
// is written for n = e3 (Qs blind to the third column of R).  The cost is blind to R n exactly when the
// partial trace T_ij = sum_a Qs[3i+a][3j+a] (3x3, PSD) has n in its null space.  With U = [u1 u2 n] in SO(3)
// the substitution R' = R U, r' = (U^T (x) I) r turns the problem into the canonical one: Qs' = P Qs P^T,
// P = U^T (x) I3; the constraint set is invariant under it (orthonormality and cross-product relations hold
// for R U iff they hold for R).  The solve then runs in primed variables; R = R' U^T and Z = Pt^T Z' Pt
// (Pt = blkdiag(P, 1)) go back to the caller, so the change of frame is invisible outside.
struct Canon {
    bool on;
    double U[9]; // row-major, columns u1, u2, n
};

// T: partial trace of the trace-normalised cost (3x3, PSD, tr T = 1).  True, with U, when the cost is blind
// to one direction n that is not e3 already.
CVX_HD bool planar_frame(const double *T, double *U)
{
    // cheap reject first: det T = product of the eigenvalues
    if (!(fabs(det3(T)) < 1e-12)) return false;
    // null vector of a rank-2 symmetric 3x3: the largest cross product of two rows
    double c[3][3], n2[3];
    CVX_UNROLL for (int k = 0; k < 3; ++k) {
        const int a = (k + 1) % 3, b = (k + 2) % 3;
        c[k][0] = T[a * 3 + 1] * T[b * 3 + 2] - T[a * 3 + 2] * T[b * 3 + 1];
        c[k][1] = T[a * 3 + 2] * T[b * 3 + 0] - T[a * 3 + 0] * T[b * 3 + 2];
        c[k][2] = T[a * 3 + 0] * T[b * 3 + 1] - T[a * 3 + 1] * T[b * 3 + 0];
        n2[k] = c[k][0] * c[k][0] + c[k][1] * c[k][1] + c[k][2] * c[k][2];
    }
    const int kb = (n2[0] >= n2[1] && n2[0] >= n2[2]) ? 0 : (n2[1] >= n2[2] ? 1 : 2);
    double nb = n2[0], n[3] = {c[0][0], c[0][1], c[0][2]};
    CVX_UNROLL for (int k = 1; k < 3; ++k)
        if (kb == k) { nb = n2[k]; n[0] = c[k][0]; n[1] = c[k][1]; n[2] = c[k][2]; }
    if (!(nb > 1e-12)) return false; // rank < 2: a degenerate configuration, left to the general path
    const double inb = 1.0 / sqrt(nb);
    n[0] *= inb; n[1] *= inb; n[2] *= inb;
    double qf = 0;
    CVX_UNROLL for (int i = 0; i < 3; ++i) qf += n[i] * (T[i * 3] * n[0] + T[i * 3 + 1] * n[1] + T[i * 3 + 2] * n[2]);
    if (!(fabs(qf) < 1e-13)) return false;                      // the cost sees every direction: not planar
    if (fabs(n[0]) < 1e-12 && fabs(n[1]) < 1e-12) return false; // already canonical (plane Z = const)
    // U = [u1 u2 n]: u1 = e_k x n (k: the smallest |n_k|), u2 = n x u1
    const int km = (fabs(n[0]) <= fabs(n[1]) && fabs(n[0]) <= fabs(n[2])) ? 0 : (fabs(n[1]) <= fabs(n[2]) ? 1 : 2);
    const double e[3] = {km == 0 ? 1.0 : 0.0, km == 1 ? 1.0 : 0.0, km == 2 ? 1.0 : 0.0};
    double u1[3] = {e[1] * n[2] - e[2] * n[1], e[2] * n[0] - e[0] * n[2], e[0] * n[1] - e[1] * n[0]};
    const double iu = 1.0 / sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
    u1[0] *= iu; u1[1] *= iu; u1[2] *= iu;
    const double u2[3] = {n[1] * u1[2] - n[2] * u1[1], n[2] * u1[0] - n[0] * u1[2], n[0] * u1[1] - n[1] * u1[0]};
    CVX_UNROLL for (int i = 0; i < 3; ++i) { U[i * 3] = u1[i]; U[i * 3 + 1] = u2[i]; U[i * 3 + 2] = n[i]; }
    return true;
}

// detects the blind direction from the packed cost (45) and, if there is one that is not e3 already,
// rewrites q in place in the canonical frame
CVX_HD void canonicalise_planar(double *q, Canon &cn)
{
    double T[9];
    CVX_UNROLL for (int i = 0; i < 3; ++i)
        CVX_UNROLL for (int j = 0; j < 3; ++j) T[i * 3 + j] = q[qidx(3 * i, 3 * j)] + q[qidx(3 * i + 1, 3 * j + 1)] + q[qidx(3 * i + 2, 3 * j + 2)];
    cn.on = planar_frame(T, cn.U);
    if (!cn.on) return;
    // Qs'[3i+x][3j+y] = sum_kl U[k][i] U[l][j] Qs[3k+x][3l+y]
    double qn[45];
    CVX_UNROLL for (int i = 0; i < 3; ++i)
        CVX_UNROLL for (int j = i; j < 3; ++j)
            CVX_UNROLL for (int x = 0; x < 3; ++x)
                CVX_UNROLL for (int y = 0; y < 3; ++y) {
                    if (i == j && y < x) continue;
                    double acc = 0;
                    CVX_UNROLL for (int k = 0; k < 3; ++k)
                        CVX_UNROLL for (int l = 0; l < 3; ++l) acc += cn.U[k * 3 + i] * cn.U[l * 3 + j] * q[qidx(3 * k + x, 3 * l + y)];
                    // the blind block is zero up to rounding (~1e-17): made exactly zero, so that the iterates stay
                    // exactly D-even like for a plane given as Z = 0 (rounding-level asymmetry is amplified by the iteration)
                    qn[qidx(3 * i + x, 3 * j + y)] = (j == 2) ? 0.0 : acc;
                }
    CVX_UNROLL for (int i = 0; i < 45; ++i) q[i] = qn[i];
}

// R = R' U^T
CVX_HD void canon_rotation_back(const Canon &cn, double *R)
{
    double Rn[9];
    CVX_UNROLL for (int i = 0; i < 3; ++i)
        CVX_UNROLL for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = R[i * 3] * cn.U[j * 3] + R[i * 3 + 1] * cn.U[j * 3 + 1] + R[i * 3 + 2] * cn.U[j * 3 + 2];
    CVX_UNROLL for (int i = 0; i < 9; ++i) R[i] = Rn[i];
}

// Z = Pt^T Z' Pt (to_canon = false) or Z' = Pt Z Pt^T (to_canon = true), Pt = blkdiag(U^T (x) I3, 1), packed 55
CVX_HD void canon_congruence(const Canon &cn, double *Z, bool to_canon)
{
    // A = U (back) or U^T (to the canonical frame):  out[3i+x][3j+y] = sum_kl A[i][k] A[j][l] Z[3k+x][3l+y]
    double A[9];
    CVX_UNROLL for (int i = 0; i < 3; ++i) CVX_UNROLL for (int k = 0; k < 3; ++k) A[i * 3 + k] = to_canon ? cn.U[k * 3 + i] : cn.U[i * 3 + k];
    double Zn[55];
    CVX_UNROLL for (int i = 0; i < 3; ++i)
        CVX_UNROLL for (int x = 0; x < 3; ++x) {
            CVX_UNROLL for (int j = i; j < 3; ++j)
                CVX_UNROLL for (int y = 0; y < 3; ++y) {
                    if (i == j && y < x) continue;
                    double acc = 0;
                    CVX_UNROLL for (int k = 0; k < 3; ++k)
                        CVX_UNROLL for (int l = 0; l < 3; ++l) acc += A[i * 3 + k] * A[j * 3 + l] * Z[sidx(3 * k + x, 3 * l + y)];
                    Zn[sidx(3 * i + x, 3 * j + y)] = acc;
                }
            double acc = 0;
            CVX_UNROLL for (int k = 0; k < 3; ++k) acc += A[i * 3 + k] * Z[sidx(3 * k + x, 9)];
            Zn[sidx(3 * i + x, 9)] = acc;
        }
    Zn[sidx(9, 9)] = Z[sidx(9, 9)];
    CVX_UNROLL for (int i = 0; i < 55; ++i) Z[i] = Zn[i];
}

// ---------------------------------------------------------------------------------------
// the solve

// certification attempts: at first_check (5), then with a spacing of check_every (2) iterations up to 10, then
// spaced more and more widely, ~sqrt(it): 5 7 9 11 15 19 25 ...  An attempt costs about one iteration -- of the
// whole wavefront: in the quad and lane layouts the other problems of the wavefront wait for it.  For a problem
// that needs N iterations a spacing s costs N / s failed attempts plus s / 2 iterations of overshoot.  Measured
// over 8 problem sets per size (tools/schedule_tune.sh, M poses/s, first_check : check_every):
//     problems/launch    4:1     5:1     5:2     5:3
//     2 k (wave)          -     10.3    13.1    13.9
//     10 k (quad)        34.9   37.3    38.1    37.5
//     24 k (quad)         -     53.1    57.2    57.3
//     125 k (lane)        -    114.0   119.7   121.9
// An attempt at iteration 4 fails for 40 % of the problems (13 % of the quad wavefronts end there); the slow
// tail, which sets the end of every launch, gains most from the sparser attempts.
constexpr int REUSE_MAX = 3; // consecutive certificate attempts that may take over the previous attempt's polished pose

// residual balancing: rp2, rd2 = squared primal / dual residuals
CVX_HD double adapted_rho(double rho, double rp2, double rd2, const Opts &o)
{
    const double m2 = o.adapt_mu * o.adapt_mu;
    double rn = rho;
    if (rp2 > m2 * rd2) rn = rho * o.adapt_tau;
    else if (rd2 > m2 * rp2) rn = rho / o.adapt_tau;
    return rn > 10.0 ? 10.0 : (rn < 1e-3 ? 1e-3 : rn);
}

// rank > 1 stall test (Opts::stall_*): lam1 >= lam2 the two largest eigenvalues of Z at this attempt, lam2_prev the
// second one at the previous attempt
CVX_HD bool rank_stalled(int it, double lam2, double lam2_prev, double fp_res, const Opts &o)
{
    // (a second eigenvalue that still shrinks -- by more than stall_drop of itself between two attempts -- is on its way to
    // rank one: such a problem is left alone)
    return o.stall_from > 0 && it >= o.stall_from && lam2 > o.stall_lam && fp_res < o.stall_res && (lam2_prev - lam2) <= o.stall_drop * lam2 && lam2_prev > 0;
}

CVX_HD int next_check_after(int it, const Opts &o)
{
    int s = 1 + (it >= 10) + (it >= 16) + (it >= 24) + (it >= 36) + (it >= 54) + (it >= 80) + (it >= 104) + (it >= 128);
    if (it >= 128) s += (it - 128) / 32;
    return it + o.check_every * s;
}

struct Solution {
    double R[9];    // row-major, world -> camera, x_c = R X + t
    double t[3];
    double cost;    // ||A r||^2 in the reference's (unnormalised) units
    double dobj;    // certified lower bound on the SDP optimum (same units); NaN if uncertified
    int status;
    int iters;
    int rank;       // #eig(Z) > 1e-3 at exit (cvxpnpl.py:502)
    int sweeps;     // total Jacobi sweeps (work counter for the flop model)
};

// reference-style recovery from an uncertified ADMM iterate Z = Wp (cvxpnpl.py:499-513):
// v is the unit top eigenvector of Z, v2 the runner-up, rank = #eig(Z) > 1e-3.  Rank 1: R = U V^T of the
// rank-1 ratio (no determinant correction, cvxpnpl.py:510-511).  Rank > 1 is flagged for the multi-solution
// recovery (cvxpnpl.py:506-507), and R, t then hold ONE candidate pose, never NaN while Z is finite: the
// rank-1 ratio of the top eigenvector is meaningless there (for an exact two-fold ambiguity it is z1 - z2
// with last entry 0), so the pose is the better of the two rank-2 candidates of the top-2 eigenspace
// (twin_candidates: what the reference's rank-2 branch, cvxpnpl.py:303-315, computes for a rank-2 Z) --
// proper rotations before reflections, then the lower cost; a proper rotation is Newton-polished on SO(3).
template <class QV>
CVX_HD double rounded_cost(QV Qs, const double *z, double *R, bool &fin)
{
    double M0[9], iv = 1.0 / z[9];
    CVX_UNROLL for (int i = 0; i < 3; ++i) CVX_UNROLL for (int j = 0; j < 3; ++j) M0[i * 3 + j] = z[3 * j + i] * iv;
    polar3(M0, R, 12);
    double r[9], Qr[9];
    CVX_UNROLL for (int i = 0; i < 3; ++i) CVX_UNROLL for (int j = 0; j < 3; ++j) r[3 * j + i] = R[i * 3 + j];
    q9_mul(Qs, r, Qr);
    double c = 0;
    fin = true;
    CVX_UNROLL for (int i = 0; i < 9; ++i) { c += r[i] * Qr[i]; fin &= (R[i] == R[i]); }
    fin &= (c == c);
    return c;
}

template <class QV>
CVX_HD void fallback_pose(QV Qs, double tr, const double *v, const double *v2, int rank, Solution &sol)
{
    sol.rank = rank;
    bool okf;
    double c = rounded_cost(Qs, v, sol.R, okf);
    if (rank > 1) {
        double zp[10], zm[10], Rp[9], Rm[9];
        twin_candidates(v, v2, zp, zm);
        bool okp, okm;
        const double fp = rounded_cost(Qs, zp, Rp, okp), fm = rounded_cost(Qs, zm, Rm, okm);
        const bool pp = okp && det3(Rp) > 0, pm = okm && det3(Rm) > 0;
        const bool take_m = okm && (!okp || (pm && !pp) || (pm == pp && fm < fp));
        if (okp || okm) {
            CVX_UNROLL for (int i = 0; i < 9; ++i) sol.R[i] = take_m ? Rm[i] : Rp[i];
            c = take_m ? fm : fp;
            // a proper rotation is Newton-polished on SO(3): the pose returned is then a stationary point of the cost,
            // the same one cvxpnpl_recover_multi (with Q45) reports for this candidate
            if (take_m ? pm : pp) {
                double Rq[9], fq;
                CVX_UNROLL for (int i = 0; i < 9; ++i) Rq[i] = sol.R[i];
                polish_rotation(Qs, Rq, fq);
                bool fin = (fq == fq);
                CVX_UNROLL for (int i = 0; i < 9; ++i) fin &= (Rq[i] == Rq[i]);
                if (fin) { CVX_UNROLL for (int i = 0; i < 9; ++i) sol.R[i] = Rq[i]; c = fq; }
            }
        }
    }
    sol.cost = tr * c;
    sol.dobj = NAN;
    sol.status = rank > 1 ? ST_RANK_GT1 : (!okf ? ST_NONFINITE : (rank != 1 ? ST_RANK_GT1 : (det3(sol.R) < 0 ? ST_REFLECTION : ST_UNCERTIFIED)));
}

// Q9: 45 packed (unnormalised A^T A), B: 3x9.  Zout (optional, 55): final Z in vech order.
// handoff (optional, 56 doubles) with handoff_at > 0: a solve that is not finished after handoff_at
// iterations stores W and the iteration count there, sets status = -1 and returns (hybrid
// schedule: the wave-per-problem kernel resumes it).
// TWIN = false compiles the two-fold-ambiguity branch out (the lane phase of the hybrid schedule hands
// off before iteration 6, where that branch starts, and the extra live state costs it registers).
template <bool TWIN> struct EigOf { typedef Eig type; };
template <> struct EigOf<false> { typedef EigF type; };

// DBL: the eigen-solve on float64 columns (Eig) instead of packed single precision (EigF); the lane phase (TWIN = false) runs
// single precision unless Opts::f32_sweeps_until asks for less than its length (cvxpnpl_hip.hip)
template <bool TWIN = true, class ST = RegStore, int VAR = VAR_FULL, bool DBL = TWIN>
CVX_HD void solve_sdp(const double *Q9, const double *B, const Opts &o, Solution &sol, double *Zout, int handoff_at = 0,
                      double *handoff = nullptr, ST st = ST())
{
    double tr = 0;
    CVX_UNROLL for (int i = 0; i < 9; ++i) tr += Q9[qidx(i, i)];
    sol.sweeps = 0; sol.iters = 0; sol.rank = 0;
    bool finite = (tr == tr) && (tr > 0) && (tr < 1e300);
    double itr = finite ? 1.0 / tr : 0.0;
    Canon cn;
    cn.on = false;
    {
        double q[45];
        CVX_UNROLL for (int i = 0; i < 45; ++i) { q[i] = Q9[i] * itr; finite &= (q[i] == q[i]); }
        if (TWIN && finite) canonicalise_planar(q, cn);
        if (!TWIN && finite && handoff) {
            // first phase of a hybrid schedule: a planar scene in a general frame goes to the wave-per-problem
            // kernel at once (it solves in the canonical frame, with the D-even certificate and the twin logic)
            double T[9], U[9];
            CVX_UNROLL for (int i = 0; i < 3; ++i)
                CVX_UNROLL for (int j = 0; j < 3; ++j) T[i * 3 + j] = q[qidx(3 * i, 3 * j)] + q[qidx(3 * i + 1, 3 * j + 1)] + q[qidx(3 * i + 2, 3 * j + 2)];
            if (planar_frame(T, U)) {
                CVX_UNROLL for (int i = 0; i < 55; ++i) handoff[i] = (i == 54) ? 1.0 : 0.0;
                handoff[55] = 0.0;
                sol.status = -1;
                return;
            }
        }
        CVX_UNROLL for (int i = 0; i < 45; ++i) st.setQ(i, q[i]);
    }
    CVX_UNROLL for (int i = 0; i < 27; ++i) st.setB(i, B[i]);
    const auto Qs = st.Q();
    if (!finite) {
        CVX_UNROLL for (int i = 0; i < 9; ++i) sol.R[i] = NAN;
        CVX_UNROLL for (int i = 0; i < 3; ++i) sol.t[i] = NAN;
        sol.cost = NAN; sol.dobj = NAN; sol.status = ST_NONFINITE;
        if (Zout) { CVX_UNROLL for (int i = 0; i < 55; ++i) Zout[i] = NAN; }
        return;
    }
    // PSD slack of the certificate: gap <= tr (zSz + 4 delta) <= eps
    double delta = o.eps / (8.0 * tr);
    delta = delta < 1e-13 ? 1e-13 : delta;
    double rho = o.rho, irho = 1.0 / o.rho;

    double W[55], Wp[55];
    CVX_UNROLL for (int i = 0; i < 55; ++i) W[i] = 0;
    W[sidx(9, 9)] = 1.0;
    typename EigOf<DBL>::type e; // (TWIN = false: the lane phase, at most 6 iterations: single-precision sweeps by default)
    set_exact(e, o.f32_sweeps_until == 0);
    Cert c;
    c.ok = false;
    int it = 0, next_check = o.first_check;
    bool done = false, have_prev = false;
    double Rprev[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, fprev = 0;
    double Rk[2][9], fk[2] = {0, 0}; // the twins polished by the previous check (cvx::polish_or_reuse)
    bool hk[2] = {false, false};
    int tw_reused = 0, reused = 0, attempts = 0; // attempts: certificate attempts made so far (second tries of the dual from the second one on)
    double fp_res = 1e300, lam2_prev = -1.0;
    while (!done) {
        if (handoff_at > 0 && it >= handoff_at) { // W is the iterate after `it` completed iterations
            CVX_UNROLL for (int i = 0; i < 55; ++i) handoff[i] = W[i];
            handoff[55] = (double)it;
            sol.status = -1;
            sol.iters = it;
            return;
        }
        if (it == 0 && o.first_check > 1 && o.max_iters > 1) {
            // the initial iterate W0 = e9 e9^T is diagonal and PSD: its projection is itself and its
            // eigenvectors are the unit vectors -- the first iteration needs no eigen-solve
            CVX_UNROLL for (int i = 0; i < 55; ++i) Wp[i] = W[i];
            eig_unit(e);
        } else {
            if (o.warm_start && it > 0) eig_load_warm(e, W);
            else eig_load(e, W);
            sol.sweeps += eig_solve(e, (!TWIN && o.sweep_schedule) ? sweep_cap(it + 1, true, o.jacobi_sweeps) : o.jacobi_sweeps, o.jacobi_tol * o.jacobi_tol); // (lane phase: sweep_cap)
            eig_pospart(e, Wp);
        }
        ++it;
        bool check = it >= next_check;
        bool last = (it >= o.max_iters) || (fp_res < o.res_tol);
        if (check || last) {
            // a failed dual gets its second tries (dual_certificate) from the second attempt of a solve on: the first attempt of every
            // problem would pay for them, the later ones are the slow problems that end a launch (the lane phase makes one attempt)
            const double retry_shift = (TWIN && attempts > 0 && attempts <= DUAL_RETRY_ATTEMPTS) ? o.dual_shift : 0.0;
            // (dual_refine: low two bits 1 = the eigen-gradient step, 2 = and the barrier Newton solve behind it; the higher bits say in which phases
            //  of the kernels' schedules -- this scalar core is one phase.  CVX_REFINE_FROM_EXPERIMENT: tools/experiments/dual_newton/policy_host.py)
#ifdef CVX_REFINE_FROM_EXPERIMENT
            const int refine_from = (o.dual_refine >> 4) > 0 ? (o.dual_refine >> 4) - 1 : DUAL_REFINE_FROM;
#else
            constexpr int refine_from = DUAL_REFINE_FROM;
#endif
            const bool refine = TWIN && (o.dual_refine & 3) && attempts >= refine_from && attempts < refine_from + DUAL_REFINE_ATTEMPTS; // (the eigen-gradient step)
            const bool newton = refine && (o.dual_refine & 3) >= 2;
            ++attempts;
            // top eigenvector of Wp (and the runner-up, see below)
            int jm = 0, j2 = 0;
            double best = -1, second = -1;
            CVX_UNROLL for (int j = 0; j < 10; ++j) {
                const double n2 = (double)e.n2[j];
                const bool b1 = n2 > best, b2 = !b1 && n2 > second;
                second = b1 ? best : (b2 ? n2 : second);
                j2 = b1 ? jm : (b2 ? j : j2);
                best = b1 ? n2 : best;
                jm = b1 ? j : jm;
            }
            if (TWIN && o.stall_from > 0 && it >= o.stall_from - 32) { // a Z that has settled at rank > 1: the relaxation is not tight
                const double lam2 = sqrt_fast(second) - e.sigma;
                last = last || rank_stalled(it, lam2, lam2_prev, fp_res, o);
                lam2_prev = lam2;
            }
            // Two-fold ambiguous problems (two poses with equal or nearly equal cost -- every planar scene is
            // one: R diag(-1,-1,1) has exactly the same algebraic cost) make Z converge to a rank-2 mixture
            // of both, eigenvalues (~2, ~2), and the top eigenvector need not be either pose.  From iteration
            // 6 on, when the second eigenvalue is comparable, both poses are read off the top-2 eigenspace
            // in closed form (twin_candidates) and polished: equal cost => the problem IS two-fold ambiguous:
            // stop with rank 2 and the exact Z = (z+ z+^T + z- z-^T) / 2, like the reference's rank-2 branch;
            // otherwise the better one goes through the dual certificate.
            // l2 > l1 / 2 with l = sqrt(n2) - sigma, evaluated only where the twin logic can fire
            bool two = false;
            if (TWIN && it >= 6) two = (sqrt_fast(second) - e.sigma) > 0.5 * (sqrt_fast(best) - e.sigma);
            double vt[10], v2[10], il1 = rsqrt_(best), il2 = rsqrt_(second);
            CVX_UNROLL for (int i = 0; i < 10; ++i) {
                double s1 = 0, s2 = 0;
                CVX_UNROLL for (int j = 0; j < 10; ++j) { s1 = (j == jm) ? eig_g(e, j, i) : s1; s2 = (j == j2) ? eig_g(e, j, i) : s2; }
                vt[i] = s1 * il1;
                v2[i] = s2 * il2;
            }
            bool ambiguous = false, twin_tested = false;
            double Rm[9], fm = 0;
            if (!two) {
                // 9 of 10 repeated checks polish to the pose the previous check already had (it was the dual
                // that was not ready): when the candidate rounds to that rotation (rounds_to: within ~0.16 rad,
                // 99.9 % of those polish back onto it) reuse it, skipping the polar and the Newton iterations
                // The shortcut must not outlive its premise: two local minima can lie within the rounding
                // tolerance of each other, and a pose taken over forever from an early check (or from before a
                // spell in the twin branch) would then never be certified although the iterate has long
                // converged to the other one.  So at most REUSE_MAX checks in a row reuse, and the twin branch
                // invalidates the stored pose.
                double d0;
                // (TWIN = false is the lane phase of the hybrid schedule: at most two checks, nothing to bound)
                if (have_prev && (!TWIN || reused < REUSE_MAX) && rounds_to(vt, Rprev, d0)) {
                    CVX_UNROLL for (int i = 0; i < 9; ++i) c.R[i] = Rprev[i];
                    c.pobj = fprev;
                    if (TWIN) ++reused;
                } else {
                    d0 = round_candidate(vt, c.R);
                    polish_rotation(Qs, c.R, c.pobj);
                    if (TWIN) reused = 0;
                }
                dual_certificate<TWIN, decltype(Qs), VAR>(Qs, W, Wp, rho, delta, d0, c, retry_shift, refine ? v2 : nullptr, newton);
                have_prev = d0 > 0 && (c.pobj == c.pobj);
                CVX_UNROLL for (int i = 0; i < 9; ++i) Rprev[i] = c.R[i];
                fprev = c.pobj;
            } else {
                double zp[10], zm[10], fp;
                have_prev = false; // (see above)
                twin_candidates(vt, v2, zp, zm);
                const bool may = tw_reused < 8; // every ninth check polishes afresh
                const double dp = polish_or_reuse(Qs, zp, Rk[0], fk[0], may && hk[0], c.R, fp);
                const double dm = polish_or_reuse(Qs, zm, Rk[1], fk[1], may && hk[1], Rm, fm);
                tw_reused = may ? tw_reused + 1 : 0;
                CVX_UNROLL for (int i = 0; i < 9; ++i) { Rk[0][i] = c.R[i]; Rk[1][i] = Rm[i]; }
                fk[0] = fp; fk[1] = fm;
                hk[0] = dp > 0 && (fp == fp);
                hk[1] = dm > 0 && (fm == fm);
                double trc = 0;
                bool fin = (fp == fp) && (fm == fm);
                CVX_UNROLL for (int i = 0; i < 9; ++i) { trc += c.R[i] * Rm[i]; fin &= (c.R[i] == c.R[i]) && (Rm[i] == Rm[i]); }
                const double gtol = (o.eps > 8e-13 * tr ? o.eps : 8e-13 * tr) * itr;
                ambiguous = fin && dp > 0 && dm > 0 && fabs(fp - fm) <= gtol && trc < 2.9;
                // Equal cost alone is not enough: a symmetric problem maps every stationary point to a twin of
                // equal cost, local minima included.  The pair is accepted only with a certificate: z+ must pass
                // the dual test (it is then a global optimum, and so is z- with the same cost).
                if (ambiguous) {
                    c.pobj = fp;
                    dual_certificate<TWIN, decltype(Qs), VAR>(Qs, W, Wp, rho, delta, dp, c, retry_shift, refine ? v2 : nullptr, newton);
                    ambiguous = c.ok && (tr * (fabs(c.zSz) + 4.0 * delta) <= (o.eps > 8e-13 * tr ? o.eps : 8e-13 * tr));
                    twin_tested = true;
                }
                if (!ambiguous && !twin_tested) {
                    const bool take_m = dm > 0 && (fm == fm) && (!(dp > 0) || !(fp == fp) || fm < fp);
                    if (take_m) { CVX_UNROLL for (int i = 0; i < 9; ++i) c.R[i] = Rm[i]; }
                    c.pobj = take_m ? fm : fp;
                    dual_certificate<TWIN, decltype(Qs), VAR>(Qs, W, Wp, rho, delta, take_m ? dm : dp, c, retry_shift, refine ? v2 : nullptr, newton);
                } else if (!ambiguous) {
                    c.ok = false; // equal-cost twins whose certificate is not there yet: keep iterating
                }
            }
            next_check = next_check_after(it, o);
            if (ambiguous) {
                CVX_UNROLL for (int i = 0; i < 9; ++i) sol.R[i] = c.R[i];
                sol.cost = tr * c.pobj;
                sol.dobj = tr * (c.pobj - c.zSz - 4.0 * delta); // the pair IS certified: both twins attain this bound to eps
                sol.status = ST_RANK_GT1;
                sol.rank = 2;
                if (Zout) {
                    double za[10], zb[10];
                    CVX_UNROLL for (int i = 0; i < 3; ++i) CVX_UNROLL for (int j = 0; j < 3; ++j) { za[3 * j + i] = c.R[i * 3 + j]; zb[3 * j + i] = Rm[i * 3 + j]; }
                    za[9] = 1.0; zb[9] = 1.0;
                    CVX_UNROLL for (int i = 0; i < 10; ++i) CVX_UNROLL for (int j = i; j < 10; ++j) Zout[sidx(i, j)] = 0.5 * (za[i] * za[j] + zb[i] * zb[j]);
                }
                done = true;
            }
            bool gap_ok = c.ok && (tr * (fabs(c.zSz) + 4.0 * delta) <= (o.eps > 8e-13 * tr ? o.eps : 8e-13 * tr));
            if (gap_ok && !done) {
                CVX_UNROLL for (int i = 0; i < 9; ++i) sol.R[i] = c.R[i];
                sol.cost = tr * c.pobj;
                sol.dobj = tr * (c.pobj - c.zSz - 4.0 * delta);
                sol.status = ST_CERTIFIED;
                sol.rank = 1;
                if (Zout) {
                    double z[10];
                    CVX_UNROLL for (int i = 0; i < 3; ++i) CVX_UNROLL for (int j = 0; j < 3; ++j) z[3 * j + i] = c.R[i * 3 + j];
                    z[9] = 1.0;
                    CVX_UNROLL for (int i = 0; i < 10; ++i) CVX_UNROLL for (int j = i; j < 10; ++j) Zout[sidx(i, j)] = z[i] * z[j];
                }
                done = true;
            } else if (last && !done) {
                int rank = 0;
                const double thr = (e.sigma + 1e-3) * (e.sigma + 1e-3); // eigenvalue > 1e-3 (cvxpnpl.py:501) without the roots
                CVX_UNROLL for (int j = 0; j < 10; ++j) rank += e.n2[j] > thr;
                fallback_pose(Qs, tr, vt, v2, rank, sol);
                if (Zout) { CVX_UNROLL for (int i = 0; i < 55; ++i) Zout[i] = Wp[i]; }
                done = true;
            }
        }
        if (!done && it == o.tail_from) {
            // the few problems still running get a smaller penalty (measured: shorter tail); W = Wp + Wm
            // with Wm = -S / rho, so keeping the dual S means rescaling Wm by rho / rho_tail
            const double sc = rho / o.rho_tail;
            CVX_UNROLL for (int i = 0; i < 55; ++i) W[i] = Wp[i] + (W[i] - Wp[i]) * sc;
            rho = o.rho_tail;
            irho = 1.0 / rho;
        }
        if (TWIN && !done && o.adapt_every > 0 && it >= o.adapt_from && (it - o.adapt_from) % o.adapt_every == 0) { // (TWIN = false: the lane phase ends at iteration 5)
            // Residual balancing for the slow tail (from iteration adapt_from on, every adapt_every): primal residual =
            // distance of the PSD iterate from the affine set, dual residual = the part of S - Qs outside span A_i (S =
            // rho (Wp - W)).  The larger one by more than adapt_mu gets the penalty moved its way by adapt_tau; the dual is
            // kept (Wm rescaled), like at the tail_from switch.  Measured on 12 k minimal (N = 4) hypotheses: problems
            // beyond 300 iterations 145 -> 59, uncertified exits 35 -> 8; planar scenes p99 968 -> 147 iterations.
            double P[55], T[55];
            CVX_UNROLL for (int i = 0; i < 55; ++i) { P[i] = Wp[i]; T[i] = rho * (Wp[i] - W[i]); }
            CVX_UNROLL for (int i = 0; i < 9; ++i) CVX_UNROLL for (int j = i; j < 9; ++j) T[sidx(i, j)] -= Qs[qidx(i, j)];
            proj_affine<VAR>(P, false);
            proj_affine<VAR>(T, true);
            double rp = 0, rd = 0;
            CVX_UNROLL for (int i = 0; i < 10; ++i)
                CVX_UNROLL for (int j = i; j < 10; ++j) {
                    const double w_ = (i == j ? 1.0 : 2.0), dp_ = P[sidx(i, j)] - Wp[sidx(i, j)];
                    rp += w_ * dp_ * dp_;
                    rd += w_ * T[sidx(i, j)] * T[sidx(i, j)];
                }
            const double rn = adapted_rho(rho, rp, rd, o);
            if (rn != rho) {
                const double sc = rho / rn;
                CVX_UNROLL for (int i = 0; i < 55; ++i) W[i] = Wp[i] + (W[i] - Wp[i]) * sc;
                rho = rn; irho = 1.0 / rho;
            }
        }
        if (!done) {
            // X = Pi_aff(2 Wp - W - Qs / rho);  W <- W + alpha (X - Wp)
            double X[55];
            CVX_UNROLL for (int i = 0; i < 55; ++i) X[i] = 2.0 * Wp[i] - W[i];
            CVX_UNROLL for (int i = 0; i < 9; ++i) CVX_UNROLL for (int j = i; j < 9; ++j) X[sidx(i, j)] -= irho * Qs[qidx(i, j)];
            proj_affine<VAR>(X, false);
            double r2 = 0;
            CVX_UNROLL for (int i = 0; i < 10; ++i)
                CVX_UNROLL for (int j = i; j < 10; ++j) {
                    double d = X[sidx(i, j)] - Wp[sidx(i, j)];
                    r2 += (i == j ? 1.0 : 2.0) * d * d;
                    W[sidx(i, j)] += o.alpha * d;
                }
            fp_res = sqrt_(r2);
#ifdef CVX_TRACE
            {
                double lam[10];
                for (int j = 0; j < 10; ++j) lam[j] = sqrt((double)e.n2[j]) - e.sigma;
                for (int a = 0; a < 10; ++a) for (int b = a + 1; b < 10; ++b) if (lam[b] > lam[a]) { double t_ = lam[a]; lam[a] = lam[b]; lam[b] = t_; }
                if (it <= 30 || it % 50 == 0)
                    printf("it %4d fp_res %.3e eig+ %.4f %.4f %.4f %.4f  eig- %.2e  cert: ok=%d minpiv %.2e res %.1e pobj %.3e\n", it, fp_res, lam[0], lam[1], lam[2], lam[3], lam[9], (int)c.ok, c.min_piv, c.res, c.pobj);
            }
#endif
            if (!(fp_res == fp_res)) { // NaN guard
                CVX_UNROLL for (int i = 0; i < 9; ++i) sol.R[i] = NAN;
                sol.cost = NAN; sol.dobj = NAN; sol.status = ST_NONFINITE;
                if (Zout) { CVX_UNROLL for (int i = 0; i < 55; ++i) Zout[i] = NAN; }
                done = true;
            }
        }
    }
    sol.iters = it;
    if (cn.on) { // back to the caller's frame
        canon_rotation_back(cn, sol.R);
        if (Zout) canon_congruence(cn, Zout, false);
    }
    // t = -B r (cvxpnpl.py:513)
    {
        double r[9];
        CVX_UNROLL for (int i = 0; i < 3; ++i) CVX_UNROLL for (int j = 0; j < 3; ++j) r[3 * j + i] = sol.R[i * 3 + j];
        CVX_UNROLL for (int i = 0; i < 3; ++i) {
            double acc = 0;
            CVX_UNROLL for (int j = 0; j < 9; ++j) acc += st.B(i * 9 + j) * r[j];
            sol.t[i] = -acc;
        }
    }
}

} // inline namespace CVX_UNIT_TAG
} // namespace cvx
