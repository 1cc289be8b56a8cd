// Host experiment: does Anderson acceleration of the Douglas-Rachford map W -> W + alpha (X - W+) shorten the slow tail (the 11-15
// iteration problems that end a 10 000-problem launch, the 41-iteration one of 125 000)?  Builds solver_core.h with CVX_AA_EXPERIMENT;
// memory AA_M (0 = off), from iteration AA_FROM, safeguard AA_SAFE (reject a step whose mixing coefficients exceed it).
//   g++ -O2 -fopenmp -shared -fPIC -o /tmp/libaa.so tools/experiments/aa_hostsim.cpp ; python tools/experiments/aa_check.py
#include <cmath>
#include <cstdlib>
static int aa_m = 0, aa_from = 3;
static double aa_safe = 10.0;
struct AAState { double G[3][55], F[3][55]; int n; };
static thread_local AAState aa_;
static inline double wdot(const double *a, const double *b)
{
    static const int diag[10] = {0, 10, 19, 27, 34, 40, 45, 49, 52, 54};
    double s = 0;
    for (int i = 0; i < 55; ++i) s += 2.0 * a[i] * b[i];
    for (int i = 0; i < 10; ++i) s -= a[diag[i]] * b[diag[i]];
    return s;
}
static inline void aa_step(int it, double *W, const double *g, const double *f)
{
    if (it <= 1) aa_.n = 0;
    const int m = aa_m;
    int n = aa_.n;
    bool done = false;
    if (m > 0 && it >= aa_from && n >= 1) {
        const int k = n < m ? n : m; // differences available
        // dF_i = f - F[i] (i-th most recent), dG_i likewise: W+ = g - sum gamma_i (g - G[i])  with gamma = argmin |f - sum gamma_i (f - F[i])|
        double dF[3][55], dG[3][55], A[3][3], b[3], gam[3] = {0, 0, 0};
        for (int i = 0; i < k; ++i)
            for (int e = 0; e < 55; ++e) { dF[i][e] = f[e] - aa_.F[i][e]; dG[i][e] = g[e] - aa_.G[i][e]; }
        for (int i = 0; i < k; ++i) { b[i] = wdot(dF[i], f); for (int j = 0; j < k; ++j) A[i][j] = wdot(dF[i], dF[j]); }
        for (int i = 0; i < k; ++i) A[i][i] *= 1.0 + 1e-10;
        // tiny Gaussian elimination
        bool ok = true;
        double M[3][4];
        for (int i = 0; i < k; ++i) { for (int j = 0; j < k; ++j) M[i][j] = A[i][j]; M[i][k] = b[i]; }
        for (int c = 0; c < k && ok; ++c) {
            int p = c;
            for (int r = c + 1; r < k; ++r) if (fabs(M[r][c]) > fabs(M[p][c])) p = r;
            if (!(fabs(M[p][c]) > 1e-300)) { ok = false; break; }
            for (int j = 0; j <= k; ++j) { double t = M[c][j]; M[c][j] = M[p][j]; M[p][j] = t; }
            for (int r = c + 1; r < k; ++r) { const double q = M[r][c] / M[c][c]; for (int j = c; j <= k; ++j) M[r][j] -= q * M[c][j]; }
        }
        if (ok) {
            for (int c = k - 1; c >= 0; --c) { double v = M[c][k]; for (int j = c + 1; j < k; ++j) v -= M[c][j] * gam[j]; gam[c] = v / M[c][c]; }
            double gs = 0;
            for (int i = 0; i < k; ++i) gs += fabs(gam[i]);
            if (gs == gs && gs <= aa_safe) {
                for (int e = 0; e < 55; ++e) { double v = g[e]; for (int i = 0; i < k; ++i) v -= gam[i] * dG[i][e]; W[e] = v; }
                done = true;
            }
        }
    }
    if (!done) for (int e = 0; e < 55; ++e) W[e] = g[e];
    // history: most recent first
    for (int i = 2; i > 0; --i) for (int e = 0; e < 55; ++e) { aa_.G[i][e] = aa_.G[i - 1][e]; aa_.F[i][e] = aa_.F[i - 1][e]; }
    for (int e = 0; e < 55; ++e) { aa_.G[0][e] = g[e]; aa_.F[0][e] = f[e]; }
    aa_.n = n + 1 > 3 ? 3 : n + 1;
}
#define CVX_AA_EXPERIMENT
#include "../../cvxpnpl_amd/csrc/solver_core.h"
#include "../../cvxpnpl_amd/csrc/problem_io.h"

extern "C" {
void aa_config(int m, int from, double safe) { aa_m = m; aa_from = from; aa_safe = safe; }
void aa_default_opts(cvx::Opts *o) { *o = cvx::default_opts(); }
int aa_solve_batch(int batch, int n_p, const double *pts_2d, const double *pts_3d, const double *K, const cvx::Opts *opts, int *status, int *iters)
{
#pragma omp parallel for schedule(dynamic, 16)
    for (int b = 0; b < batch; ++b) {
        cvx::ProblemView pv = cvx::make_view(b, n_p, pts_2d, pts_3d, 0, nullptr, nullptr, K, 0);
        cvx::Solution sol;
        cvx::solve_problem(pv, *opts, sol, nullptr);
        status[b] = sol.status; iters[b] = sol.iters;
    }
    return 0;
}
}
