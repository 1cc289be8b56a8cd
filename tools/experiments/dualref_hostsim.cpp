// Host experiment: when a certificate attempt fails (S = corrected dual hint has an eigenvalue below -delta), refine the DUAL alone --
// alternating projections between the PSD cone and the affine family { S in Qs + span A_i, S z = 0 } (the pose z is already right) --
// for DR_CYCLES cycles before giving up.  Does that certify earlier than waiting for the first-order iteration to deliver a better hint?
//   g++ -O2 -fopenmp -shared -fPIC -std=c++17 -o /tmp/libdr.so tools/experiments/dualref_hostsim.cpp ; python tools/experiments/dualref_check.py
#define CVX_DUALREF_EXPERIMENT
#include "../../cvxpnpl_amd/csrc/solver_core.h"
#include "../../cvxpnpl_amd/csrc/problem_io.h"
#include <cmath>
static int dr_cycles = 0;
static double dr_margin = 0.0;
static int dr_from = 0;
static double dr_over = 1.0; // over-projection: S+ = S + dr_over * (P_psd(S) - S)
static thread_local long dr_used = 0, dr_saved = 0;

// cyclic Jacobi eigen-decomposition of a symmetric 10x10 (full storage), V columns = eigenvectors
static void eig10(double A[10][10], double V[10][10])
{
    for (int i = 0; i < 10; ++i) for (int j = 0; j < 10; ++j) V[i][j] = i == j;
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0;
        for (int p = 0; p < 10; ++p) for (int q = p + 1; q < 10; ++q) off += A[p][q] * A[p][q];
        if (off < 1e-30) break;
        for (int p = 0; p < 10; ++p)
            for (int q = p + 1; q < 10; ++q) {
                if (fabs(A[p][q]) < 1e-300) continue;
                const double th = (A[q][q] - A[p][p]) / (2 * A[p][q]);
                const double t = (th >= 0 ? 1.0 : -1.0) / (fabs(th) + sqrt(th * th + 1)), c = 1 / sqrt(t * t + 1), s = t * c;
                for (int k = 0; k < 10; ++k) { const double a = A[k][p], b = A[k][q]; A[k][p] = c * a - s * b; A[k][q] = s * a + c * b; }
                for (int k = 0; k < 10; ++k) { const double a = A[p][k], b = A[q][k]; A[p][k] = c * a - s * b; A[q][k] = s * a + c * b; }
                for (int k = 0; k < 10; ++k) { const double a = V[k][p], b = V[k][q]; V[k][p] = c * a - s * b; V[k][q] = s * a + c * b; }
            }
    }
}

namespace cvx {
template <bool SYMM, class QV, int VAR>
void dualref_cycles(QV Qs, double *S, const double *z, Cert &c, bool symm, double delta, double d0)
{
    c.ok = false;
    if (dr_cycles <= 0 || !(d0 > 0) || !(c.pobj == c.pobj) || (dr_from >= 0 && dualref_it < dr_from)) return;
    {   // already good?
        double T[55];
        for (int i = 0; i < 55; ++i) T[i] = S[i];
        for (int i = 0; i < 10; ++i) T[sidx(i, i)] += delta;
        if (dr_from >= 0 && ldl_min_pivot(T) > 0 && c.res < 1e-10) return; // the caller's own test will pass
    }
    double S0[55];
    for (int i = 0; i < 55; ++i) S0[i] = S[i];
    const double res0 = c.res, zSz0 = c.zSz;
    const double ladder[6] = {0.03, 0.01, 0.003, 0.001, 0.0003, 0.0001};
    const int nl = dr_margin < 0 ? (int)(-dr_margin) : 1;
    for (int li = 0; li < nl; ++li) {
    const double margin_ = dr_margin < 0 ? ladder[li] : dr_margin;
    for (int i = 0; i < 55; ++i) S[i] = S0[i];
    for (int cyc = 0; cyc < dr_cycles; ++cyc) {
        double A[10][10], V[10][10];
        if (dr_over < 0) { // no eigen-solve: a uniform shift on the complement of z (|z|^2 = 4)
            for (int i = 0; i < 10; ++i) for (int j = i; j < 10; ++j) S[sidx(i, j)] += margin_ * ((i == j ? 1.0 : 0.0) - 0.25 * z[i] * z[j]);
        } else {
        for (int i = 0; i < 10; ++i) for (int j = 0; j < 10; ++j) A[i][j] = S[sidx(i, j)];
        eig10(A, V);
        // S <- S + over * (neg part removed)
        for (int k = 0; k < 10; ++k) {
            const double lam = A[k][k];
            double vz = 0;
            for (int i = 0; i < 10; ++i) vz += V[i][k] * z[i];
            if (fabs(vz) > 1.0) continue; // (|z| = 2: the null direction z itself)
            if (lam < margin_)
                for (int i = 0; i < 10; ++i) for (int j = i; j < 10; ++j) S[sidx(i, j)] += dr_over * (margin_ - lam) * V[i][k] * V[j][k];
        }
        }
        // back onto Qs + span A_i
        double T[55];
        for (int i = 0; i < 55; ++i) T[i] = S[i];
        for (int i = 0; i < 9; ++i) for (int j = i; j < 9; ++j) T[sidx(i, j)] -= Qs[qidx(i, j)];
        proj_affine<VAR>(T, true);
        for (int i = 0; i < 55; ++i) S[i] -= T[i];
        if (symm) { for (int i = 0; i < 10; ++i) for (int j = i; j < 10; ++j) if (odd_entry(i, j)) S[sidx(i, j)] = 0.0; }
        // and onto S z = 0
        double rhs[10], lam[10], Sz[10];
        sym_mul10(S, z, rhs);
        dual_lambda<VAR>(c.R, rhs, symm, lam);
        sub_range_of_rank2<VAR>(S, lam, z, symm);
        sym_mul10(S, z, Sz);
        c.res = 0; c.zSz = 0;
        for (int i = 0; i < 10; ++i) { c.res = fabs(Sz[i]) > c.res ? fabs(Sz[i]) : c.res; c.zSz += z[i] * Sz[i]; }
        double T2[55];
        for (int i = 0; i < 55; ++i) T2[i] = S[i];
        for (int i = 0; i < 10; ++i) T2[sidx(i, i)] += delta;
        const double mp = ldl_min_pivot(T2);
        ++dr_used;
        if (mp > 0 && c.res < 1e-10) { c.min_piv = mp; c.ok = true; ++dr_saved; return; }
    }
    }
    for (int i = 0; i < 55; ++i) S[i] = S0[i];
    c.res = res0; c.zSz = zSz0;
}
}

extern "C" {
void dr_config(int cycles, double over, double margin, int from) { dr_cycles = cycles; dr_over = over; dr_margin = margin; dr_from = from; }
void dr_default_opts(cvx::Opts *o) { *o = cvx::default_opts(); }
int dr_solve_batch(int batch, int n_p, const double *pts_2d, const double *pts_3d, const double *K, const cvx::Opts *opts, int *status, int *iters, double *R_out, long *stats)
{
    long used = 0, saved = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : used, saved)
    for (int b = 0; b < batch; ++b) {
        dr_used = 0; dr_saved = 0;
        cvx::ProblemView pv = cvx::make_view(b, n_p, pts_2d, pts_3d, 0, nullptr, nullptr, K, 0);
        cvx::Solution sol;
        cvx::solve_problem(pv, *opts, sol, nullptr);
        status[b] = sol.status; iters[b] = sol.iters;
        for (int i = 0; i < 9; ++i) R_out[9 * (size_t)b + i] = sol.R[i];
        used += dr_used; saved += dr_saved;
    }
    stats[0] = used; stats[1] = saved;
    return 0;
}
}
