// problem_io.h -- how one problem's correspondences are read and turned into (B, Q9).
//
// Batch layout at the C-ABI (include/cvxpnpl_amd.h): contiguous, problem-major
//   pts_2d  [batch][n_p][2]     pts_3d  [batch][n_p][3]
//   line_2d [batch][n_l][2][2]  line_3d [batch][n_l][2][3]     K [3][3] or [batch][3][3]
// the layout np.stack of the reference's per-problem arguments gives (cvxpnpl.py:523-595).
#pragma once
#include "solver_core.h"

namespace cvx {
inline namespace CVX_UNIT_TAG {

struct ProblemView {
    int n_p, n_l;
    const double *p2, *p3, *l2, *l3, *K;
    const double *Q45, *B27; // cost entry (cvxpnpl_solve_cost_batch): A^T A packed and B given, nothing to assemble
};

CVX_HD ProblemView make_view(long b, int n_p, const double *pts_2d, const double *pts_3d, int n_l, const double *line_2d,
                             const double *line_3d, const double *K, int K_per_problem)
{
    ProblemView v;
    v.n_p = n_p; v.n_l = n_l;
    v.p2 = n_p ? pts_2d + b * n_p * 2 : nullptr;
    v.p3 = n_p ? pts_3d + b * n_p * 3 : nullptr;
    v.l2 = n_l ? line_2d + b * n_l * 4 : nullptr;
    v.l3 = n_l ? line_3d + b * n_l * 6 : nullptr;
    v.K = K ? K + (K_per_problem ? b * 9 : 0) : nullptr;
    v.Q45 = nullptr; v.B27 = nullptr;
    return v;
}

// The point c the Gram sums are taken about (P' = P - c; exact for any c: the cost is invariant, t = -B' r - R c): per coordinate the
// MEDIAN of the first three 3D records (points, then line end points), so that one far outlier among them -- a bad RANSAC sample, a
// sentinel point in first position -- cannot drag the centre away from the scene and bring the |c|^2 / spread^2 cancellation back
// (round-2 advisor finding; rounds 1-2 used the first record).  Fewer than three records: the first one.
CVX_HD void shift_centre(int n_p, const double *p3, int n_l, const double *l3, double *c)
{
    const int nrec = n_p + 2 * n_l;
    CVX_UNROLL for (int k = 0; k < 3; ++k) {
        const double a = n_p > 0 ? p3[k] : l3[k];
        double m = a;
        if (nrec >= 3) {
            const double b = n_p > 1 ? p3[3 + k] : l3[3 * (1 - n_p) + k];
            const double d = n_p > 2 ? p3[6 + k] : l3[3 * (2 - n_p) + k];
            const double lo = a < b ? a : b, hi = a < b ? b : a;
            m = d < lo ? lo : (d > hi ? hi : d);
        }
        c[k] = m;
    }
}

// cvxpnpl.py:523-627 up to the call of _solve_relaxation: B (3x9) and Q9 = A^T A (45 packed)
CVX_HD bool assemble(const ProblemView &v, double *B, double *Q9)
{
    double Kc[9], Ki[9], det;
    CVX_UNROLL for (int i = 0; i < 9; ++i) Kc[i] = v.K[i];
    inv3(Kc, Ki, det);
    Gram g;
    gram_zero(g);
    // The sums are taken about a point c of the scene (shift_centre; P' = P - c): the cost r^T Q r is invariant under that
    // shift (the translation absorbs R c) and t = -B' r - R c, i.e. B[i][3j+i] += c_j.  Exact, and the Gram difference
    // C^T C - (N^T C)^T B no longer cancels |c|^2 / spread^2 digits when the world origin is far from the scene.
    double c[3];
    shift_centre(v.n_p, v.p3, v.n_l, v.l3, c);
    for (int i = 0; i < v.n_p; ++i)
        gram_add_point(g, Ki, v.p2[2 * i], v.p2[2 * i + 1], v.p3[3 * i] - c[0], v.p3[3 * i + 1] - c[1], v.p3[3 * i + 2] - c[2]);
    for (int i = 0; i < v.n_l; ++i) {
        const double *e = v.l3 + 6 * i;
        const double l3s[6] = {e[0] - c[0], e[1] - c[1], e[2] - c[2], e[3] - c[0], e[4] - c[1], e[5] - c[2]};
        gram_add_line(g, Ki, v.l2 + 4 * i, l3s);
    }
    bool ok = gram_finish(g, B, Q9);
    CVX_UNROLL for (int i = 0; i < 3; ++i) CVX_UNROLL for (int j = 0; j < 3; ++j) B[i * 9 + 3 * j + i] += c[j];
    return ok && (det == det) && det != 0.0;
}

template <bool TWIN = true, class ST = RegStore, int VAR = VAR_FULL, bool DBL = TWIN>
CVX_HD void solve_problem(const ProblemView &v, const Opts &o, Solution &sol, double *Zout, int handoff_at = 0,
                          double *handoff = nullptr, ST st = ST())
{
    double B[27], Q9[45];
    bool ok = true;
    if (v.Q45) { // the seam of cvxpnpl.py:454-460: the caller brings A^T A and B
        CVX_UNROLL for (int i = 0; i < 45; ++i) Q9[i] = v.Q45[i];
        CVX_UNROLL for (int i = 0; i < 27; ++i) B[i] = v.B27[i];
    } else {
        ok = assemble(v, B, Q9);
    }
    if (!ok) { // singular N^T N or K: the reference raises LinAlgError; report a NaN pose
        CVX_UNROLL for (int i = 0; i < 45; ++i) Q9[i] = NAN;
        CVX_UNROLL for (int i = 0; i < 27; ++i) B[i] = NAN;
    }
    solve_sdp<TWIN, ST, VAR, DBL>(Q9, B, o, sol, Zout, handoff_at, handoff, st);
    if (!ok) { CVX_UNROLL for (int i = 0; i < 3; ++i) sol.t[i] = NAN; }
}

} // inline namespace CVX_UNIT_TAG
} // namespace cvx
