// hostsim.cpp -- TEST-ONLY host build of the device algorithm (cvxpnpl_amd/csrc/solver_core.h).
// Lets the CPU test-suite step the exact per-lane mathematics of the HIP kernel without a
// GPU.  Never imported by the cvxpnpl_amd package; built by tests/hostsim/__init__.py.
#include "../../cvxpnpl_amd/csrc/solver_core.h"
#include "../../cvxpnpl_amd/csrc/problem_io.h"
#include "../../cvxpnpl_amd/csrc/ipm_core.h"
#include "../../cvxpnpl_amd/csrc/lane_core.h"

#include <omp.h>
extern "C" {

// thread count of the OpenMP loops below (bench.py: the all-cores and the single-core host baselines); returns the previous maximum
int hs_set_threads(int n) { const int p = omp_get_max_threads(); if (n > 0) omp_set_num_threads(n); return p; }

void hs_default_opts(cvx::Opts *o) { *o = cvx::default_opts(); }

// closed-form multipliers of the dual correction (solver_core.h: dual_lambda)
void hs_dual_lambda(const double *R9, const double *rhs10, int symm, double *lam10) { cvx::dual_lambda(R9, rhs10, symm != 0, lam10); }

// reuse test of the certificate (solver_core.h: rounds_to); returns 1/0, det of the rank-1 ratio in d0
int hs_rounds_to(const double *v10, const double *Rp9, double tol, double *d0) { return cvx::rounds_to(v10, Rp9, *d0, tol) ? 1 : 0; }

// the direction of the dual's second tries (solver_core.h: dual_retry_entry6): D(R) as a full 10 x 10
void hs_dual_retry_direction(const double *R9, double *D100)
{
    for (int a = 0; a < 10; ++a)
        for (int b = 0; b < 10; ++b) D100[a * 10 + b] = cvx::dual_retry_entry6(R9, a < b ? a : b, a < b ? b : a) * (1.0 / 6.0);
}
// the barrier Newton solve of the dual (solver_core.h: dual_newton) on one failed dual: S55 with delta on its diagonal, R9 row-major;
// out = {smallest pivot of the dual that passed (<= 0: none), |S z|_inf, z^T S z}; returns the Newton steps made
int hs_dual_newton(const double *S55, const double *R9, double delta, double lam, double *out3)
{
    int steps = 0;
    double res = 0, zSz = 0;
    out3[0] = cvx::dual_newton(S55, R9, delta, lam, res, zSz, &steps);
    out3[1] = res; out3[2] = zSz;
    return steps;
}
// its constant tables as dense matrices [15][100] (14 basis matrices of the dual family in the frame of R, then T_I)
void hs_newton_tables(double *M)
{
    for (int a = 0; a < cvx::NT_N; ++a) {
        for (int e = 0; e < 100; ++e) M[a * 100 + e] = 0.0;
        for (int e = 0; e < cvx::kNtCount[a]; ++e) {
            const int p = cvx::kNtP[a][e], q = cvx::kNtQ[a][e];
            M[a * 100 + p * 10 + q] += cvx::kNtC[a][e];
            if (p != q) M[a * 100 + q * 10 + p] += cvx::kNtC[a][e];
        }
    }
}
// homogeneous projection onto the direction space of the equalities (solver_core.h: proj_affine), packed 55 in place
void hs_proj_affine_homog(double *E55, int variant)
{
    if (variant == cvx::VAR_RC) cvx::proj_affine<cvx::VAR_RC>(E55, true);
    else cvx::proj_affine<cvx::VAR_FULL>(E55, true);
}

// same argument meaning as cvxpnpl_solve_batch (include/cvxpnpl_amd.h), host pointers
int hs_solve_batch(int batch, int n_p, const double *pts_2d, const double *pts_3d, int n_l, const double *line_2d,
                   const double *line_3d, const double *K, int K_per_problem, const cvx::Opts *opts,
                   double *R_out, double *t_out, int *status, int *iters, double *cost, int *rank, int *sweeps, double *Z_out)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (int b = 0; b < batch; ++b) {
        cvx::ProblemView pv = cvx::make_view(b, n_p, pts_2d, pts_3d, n_l, line_2d, line_3d, K, K_per_problem);
        cvx::Solution sol;
        double Z[55];
        cvx::solve_problem(pv, *opts, sol, Z_out ? Z : nullptr);
        for (int i = 0; i < 9; ++i) R_out[(size_t)b * 9 + i] = sol.R[i];
        for (int i = 0; i < 3; ++i) t_out[(size_t)b * 3 + i] = sol.t[i];
        if (status) status[b] = sol.status;
        if (iters) iters[b] = sol.iters;
        if (cost) { cost[2 * (size_t)b] = sol.cost; cost[2 * (size_t)b + 1] = sol.dobj; }
        if (rank) rank[b] = sol.rank;
        if (sweeps) sweeps[b] = sol.sweeps;
        if (Z_out) for (int i = 0; i < 55; ++i) Z_out[(size_t)b * 55 + i] = Z[i];
    }
    return 0;
}

// the interior-point path (ipm_core.h) for a batch: same outputs as hs_solve_batch
int hs_ipm_batch(int batch, int n_p, const double *pts_2d, const double *pts_3d, int n_l, const double *line_2d,
                 const double *line_3d, const double *K, int K_per_problem, const cvx::Opts *opts,
                 double *R_out, double *t_out, int *status, int *iters, double *cost, int *rank, double *Z_out)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (int b = 0; b < batch; ++b) {
        cvx::ProblemView pv = cvx::make_view(b, n_p, pts_2d, pts_3d, n_l, line_2d, line_3d, K, K_per_problem);
        double B[27], Q9[45], Z[55];
        cvx::Solution sol;
        sol.iters = 0;
        if (!cvx::assemble(pv, B, Q9)) { for (int i = 0; i < 45; ++i) Q9[i] = NAN; for (int i = 0; i < 27; ++i) B[i] = NAN; }
        if (opts->variant == cvx::VAR_RC) cvx::ipm_problem<cvx::VAR_RC>(Q9, B, *opts, sol, Z_out ? Z : nullptr);
        else cvx::ipm_problem(Q9, B, *opts, sol, Z_out ? Z : nullptr);
        for (int i = 0; i < 9; ++i) R_out[(size_t)b * 9 + i] = sol.R[i];
        for (int i = 0; i < 3; ++i) t_out[(size_t)b * 3 + i] = sol.t[i];
        if (status) status[b] = sol.status;
        if (iters) iters[b] = sol.iters;
        if (cost) { cost[2 * (size_t)b] = sol.cost; cost[2 * (size_t)b + 1] = sol.dobj; }
        if (rank) rank[b] = sol.rank;
        if (Z_out) for (int i = 0; i < 55; ++i) Z_out[(size_t)b * 55 + i] = Z[i];
    }
    return 0;
}

// the interior-point solve alone (ipm_core.h: ipm_solve) on trace-normalised costs Qs (55, vech order, zero outside the 9x9 block):
// iterates as full 10x10 matrices, gap, iterations -- the host statement tests/test_ipm_quad.py holds cvxpnpl_ipm_batch against
int hs_ipm_solve(int batch, const double *Qs55, int variant, double *Z100, double *S100, double *gap_out, int *iters)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (int b = 0; b < batch; ++b) {
        double q[45], Z[10][10], S[10][10], y[cvx::IPM_M], gap;
        for (int i = 0; i < 9; ++i)
            for (int j = i; j < 9; ++j) q[cvx::qidx(i, j)] = Qs55[(size_t)b * 55 + cvx::sidx(i, j)];
        const int nit = variant == cvx::VAR_RC ? cvx::ipm_solve<cvx::VAR_RC>(q, Z, S, y, 1e-10, 40, gap) : cvx::ipm_solve<cvx::VAR_FULL>(q, Z, S, y, 1e-10, 40, gap);
        for (int i = 0; i < 10; ++i)
            for (int j = 0; j < 10; ++j) { Z100[(size_t)b * 100 + i * 10 + j] = Z[i][j]; S100[(size_t)b * 100 + i * 10 + j] = S[i][j]; }
        gap_out[b] = gap;
        iters[b] = nit;
    }
    return 0;
}
// the constraint rows of the interior-point solve (ipm_core.h: ipm_term) as dense matrices A_i [rows][100]; returns the row count
int hs_ipm_rows(int variant, double *A)
{
    const int nr = cvx::ipm_rows(variant);
    for (int i = 0; i < nr; ++i) {
        for (int e = 0; e < 100; ++e) A[i * 100 + e] = 0.0;
        for (int k = 0; k < 3; ++k) {
            int r, c; double cf;
            if (variant == cvx::VAR_RC) cvx::ipm_term<cvx::VAR_RC>(i, k, r, c, cf); else cvx::ipm_term<cvx::VAR_FULL>(i, k, r, c, cf);
            if (r == c) A[i * 100 + r * 10 + r] += cf;
            else { A[i * 100 + r * 10 + c] += 0.5 * cf; A[i * 100 + c * 10 + r] += 0.5 * cf; }
        }
    }
    return nr;
}

// assembly only: B (27) and Q9 (45 packed)
int hs_assemble(int n_p, const double *pts_2d, const double *pts_3d, int n_l, const double *line_2d, const double *line_3d,
                const double *K, double *B, double *Q9)
{
    cvx::ProblemView pv = cvx::make_view(0, n_p, pts_2d, pts_3d, n_l, line_2d, line_3d, K, 0);
    return cvx::assemble(pv, B, Q9) ? 0 : -1;
}

void hs_proj_affine(double *E, int homog) { cvx::proj_affine(E, homog != 0); }

// eigen test hook: W (55 packed) -> Wp (55), eigenvalues (10)
int hs_pospart(const double *W, double *Wp, double *lam)
{
    cvx::Eig e;
    cvx::eig_load(e, W);
    int s = cvx::eig_solve(e, 30, 1e-30);
    cvx::eig_pospart(e, Wp);
    for (int j = 0; j < 10; ++j) lam[j] = sqrt(e.n2[j]) - e.sigma;
    return s;
}

void hs_polar3(const double *M, double *R, int iters) { cvx::polar3(M, R, iters); }
}

// the solve at the cost seam (cvxpnpl_solve_cost_batch), host pointers; variant: cvx::VAR_FULL / cvx::VAR_RC
extern "C" int hs_solve_cost_batch(int batch, const double *Q45, const double *B27, const cvx::Opts *opts, double *R_out, double *t_out,
                                   int *status, int *iters, double *cost, int *rank, double *Z_out)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (int b = 0; b < batch; ++b) {
        cvx::Solution sol;
        double Z[55];
        if (opts->variant == cvx::VAR_RC) cvx::solve_sdp<true, cvx::RegStore, cvx::VAR_RC>(Q45 + (size_t)b * 45, B27 + (size_t)b * 27, *opts, sol, Z_out ? Z : nullptr);
        else cvx::solve_sdp<true, cvx::RegStore, cvx::VAR_FULL>(Q45 + (size_t)b * 45, B27 + (size_t)b * 27, *opts, sol, Z_out ? Z : nullptr);
        for (int i = 0; i < 9; ++i) R_out[(size_t)b * 9 + i] = sol.R[i];
        for (int i = 0; i < 3; ++i) t_out[(size_t)b * 3 + i] = sol.t[i];
        if (status) status[b] = sol.status;
        if (iters) iters[b] = sol.iters;
        if (cost) { cost[2 * (size_t)b] = sol.cost; cost[2 * (size_t)b + 1] = sol.dobj; }
        if (rank) rank[b] = sol.rank;
        if (Z_out) for (int i = 0; i < 55; ++i) Z_out[(size_t)b * 55 + i] = Z[i];
    }
    return 0;
}

extern "C" void hs_proj_affine_rc(double *E, int homog) { cvx::proj_affine<cvx::VAR_RC>(E, homog != 0); }

// the first phase of the lane-hybrid schedule on the host: the general scalar core (impl 0: cvx::solve_problem<TWIN = false>, what
// solve_lane_kernel<DBL> instantiates; dbl != 0: float64 eigen-solve) or the register-budgeted restatement (impl 1: cvxl::lane_phase,
// what solve_lane2_kernel<false> instantiates; impl 2: cvxl::lane_phase_f64, its float64 instantiation).  status -1 = parked: handoff[b][0..54] = W, [55] = iteration count.
extern "C" int hs_lane_phase(int batch, int n_p, const double *pts_2d, const double *pts_3d, int n_l, const double *line_2d, const double *line_3d,
                             const double *K, int K_per_problem, const cvx::Opts *opts, int iters, int impl, int dbl, double *R_out, double *t_out,
                             int *status, int *its, double *cost, int *sweeps, double *handoff, double *Z_out)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (int b = 0; b < batch; ++b) {
        cvx::ProblemView pv = cvx::make_view(b, n_p, pts_2d, pts_3d, n_l, line_2d, line_3d, K, K_per_problem);
        cvx::Solution sol;
        sol.status = -7; sol.cost = NAN; sol.dobj = NAN;
        for (int i = 0; i < 9; ++i) sol.R[i] = NAN;
        for (int i = 0; i < 3; ++i) sol.t[i] = NAN;
        double Z[55];
        double *ho = handoff + (size_t)b * 56;
        for (int i = 0; i < 56; ++i) ho[i] = NAN;
        if (impl == 2) cvxl::lane_phase_f64(pv, *opts, sol, Z_out ? Z : nullptr, iters, ho, cvx::RegStore());
        else if (impl == 1) cvxl::lane_phase(pv, *opts, sol, Z_out ? Z : nullptr, iters, ho, cvx::RegStore());
        else if (dbl) cvx::solve_problem<false, cvx::RegStore, cvx::VAR_FULL, true>(pv, *opts, sol, Z_out ? Z : nullptr, iters, ho);
        else cvx::solve_problem<false, cvx::RegStore, cvx::VAR_FULL, false>(pv, *opts, sol, Z_out ? Z : nullptr, iters, ho);
        for (int i = 0; i < 9; ++i) R_out[(size_t)b * 9 + i] = sol.R[i];
        for (int i = 0; i < 3; ++i) t_out[(size_t)b * 3 + i] = sol.t[i];
        status[b] = sol.status;
        its[b] = sol.iters;
        cost[2 * (size_t)b] = sol.cost; cost[2 * (size_t)b + 1] = sol.dobj;
        sweeps[b] = sol.sweeps;
        if (Z_out && sol.status >= 0) for (int i = 0; i < 55; ++i) Z_out[(size_t)b * 55 + i] = Z[i];
    }
    return 0;
}
