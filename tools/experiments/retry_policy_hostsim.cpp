// Host experiment: which shifts to try when a dual fails (cvx::dual_certificate's second try), counted in LDL^T factorisations spent.
//   g++ -O2 -fopenmp -shared -fPIC -std=c++17 -o /tmp/librp.so tools/experiments/retry_policy_hostsim.cpp ; python tools/experiments/retry_policy_check.py
#define CVX_RETRY_POLICY_EXPERIMENT
#include "../../cvxpnpl_amd/csrc/solver_core.h"
#include "../../cvxpnpl_amd/csrc/problem_io.h"
static int rp_policy = 0;
static double rp_k = 3.0;
static thread_local long rp_ldl = 0, rp_ok = 0;
namespace cvx {
void retry_policy(double *S, Cert &c, double shift)
{
    double S0[55], D[55];
    for (int i = 0; i < 10; ++i) for (int j = i; j < 10; ++j) D[sidx(i, j)] = dual_retry_entry6(c.R, i, j) * (1.0 / 6.0);
    for (int i = 0; i < 55; ++i) S0[i] = S[i];
    const double p0 = -c.min_piv; // > 0: the most negative pivot met
    double rungs[4];
    int n = 0;
    if (rp_policy == 0) { rungs[n++] = shift; }
    else if (rp_policy == 1) { rungs[n++] = shift; rungs[n++] = shift / 3; rungs[n++] = shift / 9; }
    else if (rp_policy == 2) { double a = rp_k * p0; a = a < 1e-4 ? 1e-4 : (a > 0.03 ? 0.03 : a); rungs[n++] = a; }
    else if (rp_policy == 3) { double a = rp_k * p0; a = a < 1e-4 ? 1e-4 : (a > 0.03 ? 0.03 : a); rungs[n++] = a; rungs[n++] = shift; }
    else if (rp_policy == 4) { rungs[n++] = shift; rungs[n++] = shift / 4; }
    for (int r = 0; r < n; ++r) {
        for (int i = 0; i < 55; ++i) S[i] = S0[i] + rungs[r] * D[i];
        c.min_piv = ldl_min_pivot(S);
        ++rp_ldl;
        if (c.min_piv > 0) { ++rp_ok; return; }
    }
}
}
extern "C" {
void rp_config(int policy, double k) { rp_policy = policy; rp_k = k; }
void rp_default_opts(cvx::Opts *o) { *o = cvx::default_opts(); }
int rp_solve_batch(int batch, int n_p, const double *pts_2d, const double *pts_3d, const double *K, const cvx::Opts *opts, int *status, int *iters, long *stats)
{
    long l = 0, k = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : l, k)
    for (int b = 0; b < batch; ++b) {
        rp_ldl = 0; rp_ok = 0;
        cvx::ProblemView pv = cvx::make_view(b, n_p, pts_2d, pts_3d, 0, nullptr, nullptr, K, 0);
        cvx::Solution sol;
        cvx::solve_problem(pv, *opts, sol, nullptr);
        status[b] = sol.status; iters[b] = sol.iters;
        l += rp_ldl; k += rp_ok;
    }
    stats[0] = l; stats[1] = k;
    return 0;
}
}
