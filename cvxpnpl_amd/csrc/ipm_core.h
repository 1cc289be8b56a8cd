// ipm_core.h -- a second-order solve of the 10x10 relaxation for the problems the first-order iteration is slow on.
//
// The Douglas-Rachford iteration of solver_core.h certifies a typical problem in 5 iterations, but its rate depends on
// the problem: minimal and near-ambiguous configurations need hundreds to thousands of iterations,
// and a launch lasts as long as its slowest problem.  A primal-dual interior-point method does not care: 9-13
// iterations for every problem tried, at ~25 k flops each -- four typical first-order solves, but a hundredth of a
// 1 200-iteration one.  So the kernels hand a problem that is still open after opts.rescue_from iterations to this
// solver -- on the device the cooperative version of ipm_wave.h (cvxw::coop_ipm, one wavefront per problem, in
// cvxw::rescue_wave_kernel); this file is the scalar statement of the same mathematics, which the CPU tests run (tests/hostsim:
// hs_ipm_batch) -- and what comes out goes through the SAME rounding, Newton polish and dual certificate as everywhere
// else: the interior-point iterate is only a better starting point.
//
//   min <Q, Z>  s.t. <A_i, Z> = b_i (21 independent rows of the reference's 22, cvxpnpl.py:387-451), Z >= 0
//   max b^T y   s.t. S = Q - sum y_i A_i >= 0
// Feasible start: Z0 = blkdiag(I_9 / 3, 1) is the mean of z z^T over SO(3), hence feasible and positive definite; the three
// row-sum rows and the Z_99 row add up to the identity, so y0 = -c on them gives S0 = Q + c I.  HKM direction with
// Mehrotra's predictor-corrector, step to the boundary by Cholesky backtracking; both iterates stay feasible, so the
// duality gap is <Z, S> throughout.
#pragma once

#include "solver_core.h"

namespace cvx {
inline namespace CVX_UNIT_TAG {

constexpr int IPM_M = 21;
// Independent rows of the constraint set: 21 of the reference's 22 (cvxpnpl.py:387-451), or -- VAR_RC, the ablation of
// benchmarks/toolkit/methods/rc.py:9-64 -- its 16: the twelve triples 3..14, the three column sums (none of them implied once the
// row sums are gone) and Z_99 = 1.
CVX_HD constexpr int ipm_rows(int var) { return var == VAR_RC ? 16 : IPM_M; }

// term k (0..2) of constraint i: A_i = sum_k coef_k sym(E_{r_k c_k}), sym(E_rc) = (E_rc + E_cr) / 2
template <int VAR = VAR_FULL>
CVX_HD constexpr void ipm_term(int i, int k, int &r, int &c, double &coef)
{
    if (VAR == VAR_RC) {
        if (i < 12) { r = tri_i(i + 3, k); c = tri_j(i + 3, k); coef = tri_s(i + 3, k); return; }
        if (i < 15) { r = c = 3 * (i - 12) + k; coef = 1.0; return; }      // column sums of the diagonal block D[r][c] = Z[3c+r, 3c+r]
        r = c = 9; coef = k == 0 ? 1.0 : 0.0;                              // Z_99 = 1
        return;
    }
    if (i < 15) { r = tri_i(i, k); c = tri_j(i, k); coef = tri_s(i, k); return; }
    if (i < 18) { r = c = 3 * k + (i - 15); coef = 1.0; return; }          // row sums of the diagonal block D[r][c] = Z[3c+r, 3c+r]
    if (i < 20) { r = c = 3 * (i - 18) + k; coef = 1.0; return; }          // column sums (the third one is implied)
    r = c = 9; coef = k == 0 ? 1.0 : 0.0;                                  // Z_99 = 1
}

// <A_i, X> for a symmetric X
template <int VAR = VAR_FULL>
CVX_HD double ipm_adot(int i, const double (*X)[10])
{
    double s = 0;
    for (int k = 0; k < 3; ++k) {
        int r, c; double cf;
        ipm_term<VAR>(i, k, r, c, cf);
        s += cf * X[r][c];
    }
    return s;
}
// Y += sum_i y_i A_i (sign: scale)
template <int VAR = VAR_FULL>
CVX_HD void ipm_adjoint(const double *y, double scale, double (*Y)[10])
{
    for (int i = 0; i < ipm_rows(VAR); ++i)
        for (int k = 0; k < 3; ++k) {
            int r, c; double cf;
            ipm_term<VAR>(i, k, r, c, cf);
            const double v = scale * cf * y[i];
            if (r == c) Y[r][r] += v;
            else { Y[r][c] += 0.5 * v; Y[c][r] += 0.5 * v; }
        }
}

// in-place Cholesky (lower triangle) of the n x n matrix a (row stride ld); false if a pivot is not > floor
CVX_HD bool ipm_chol(double *a, int n, int ld, double floor_)
{
    for (int j = 0; j < n; ++j) {
        double d = a[j * ld + j];
        for (int k = 0; k < j; ++k) d -= a[j * ld + k] * a[j * ld + k];
        if (!(d > floor_)) return false;
        const double l = sqrt(d), il = 1.0 / l;
        a[j * ld + j] = l;
        for (int i = j + 1; i < n; ++i) {
            double s = a[i * ld + j];
            for (int k = 0; k < j; ++k) s -= a[i * ld + k] * a[j * ld + k];
            a[i * ld + j] = s * il;
        }
    }
    return true;
}
// x <- (L L^T)^-1 x
CVX_HD void ipm_chol_solve(const double *L, int n, int ld, double *x)
{
    for (int i = 0; i < n; ++i) {
        double s = x[i];
        for (int k = 0; k < i; ++k) s -= L[i * ld + k] * x[k];
        x[i] = s / L[i * ld + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = x[i];
        for (int k = i + 1; k < n; ++k) s -= L[k * ld + i] * x[k];
        x[i] = s / L[i * ld + i];
    }
}
// is X + a dX positive definite?
CVX_HD bool ipm_pd(const double (*X)[10], const double (*dX)[10], double a)
{
    double T[10][10];
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j <= i; ++j) T[i][j] = X[i][j] + a * dX[i][j];
    return ipm_chol(&T[0][0], 10, 10, 0.0);
}
// a step that keeps X + a dX positive definite, at most 1: backtracking by 0.7, then 0.9 of what passed
CVX_HD double ipm_step(const double (*X)[10], const double (*dX)[10])
{
    if (ipm_pd(X, dX, 1.0)) return 1.0;
    double a = 0.7;
    for (int t = 0; t < 40 && !ipm_pd(X, dX, a); ++t) a *= 0.7;
    return 0.9 * a;
}
CVX_HD void ipm_mul(const double (*A)[10], const double (*B)[10], double (*C)[10])
{
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 10; ++j) {
            double s = 0;
            for (int k = 0; k < 10; ++k) s += A[i][k] * B[k][j];
            C[i][j] = s;
        }
}

// Interior-point solve of the relaxation for the trace-normalised cost q (45 packed, 9x9).  Z, S: 10x10 (full, symmetric).
// Returns the number of iterations; gap = <Z, S> at exit.
template <int VAR = VAR_FULL>
CVX_HD int ipm_solve(const double *q, double (*Z)[10], double (*S)[10], double *y, double tol, int max_iters, double &gap)
{
    constexpr int NR = ipm_rows(VAR);
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 10; ++j) {
            Z[i][j] = (i == j) ? (i < 9 ? 1.0 / 3.0 : 1.0) : 0.0;
            S[i][j] = (i < 9 && j < 9) ? q[qidx(i, j)] : 0.0;
        }
    for (int i = 0; i < NR; ++i) y[i] = 0.0;
    // S0 = q + I (q is positive semidefinite with trace 1): the three row-sum rows (rc: column-sum rows) and the Z_99 row add up to I
    if (VAR == VAR_RC) y[12] = y[13] = y[14] = y[15] = -1.0;
    else y[15] = y[16] = y[17] = y[20] = -1.0;
    for (int i = 0; i < 10; ++i) S[i][i] += 1.0;
    int it = 0;
    gap = 0;
    for (int i = 0; i < 10; ++i)
        for (int j = 0; j < 10; ++j) gap += Z[i][j] * S[i][j];
    for (; it < max_iters; ++it) {
        if (gap < tol) break;
        const double mu = gap * 0.1;
        // Si = S^-1
        double Ls[10][10], Si[10][10];
        for (int i = 0; i < 10; ++i)
            for (int j = 0; j <= i; ++j) Ls[i][j] = S[i][j];
        if (!ipm_chol(&Ls[0][0], 10, 10, 0.0)) break;
        for (int c = 0; c < 10; ++c) {
            double e[10];
            for (int i = 0; i < 10; ++i) e[i] = (i == c) ? 1.0 : 0.0;
            ipm_chol_solve(&Ls[0][0], 10, 10, e);
            for (int i = 0; i < 10; ++i) Si[i][c] = e[i];
        }
        // Schur matrix M_ij = <A_i, Z A_j Si> (symmetric positive definite), factored once per iteration
        double M[NR][NR];
        for (int i = 0; i < NR; ++i)
            for (int j = 0; j <= i; ++j) {
                double s = 0;
                for (int ka = 0; ka < 3; ++ka) {
                    int a, b; double ca;
                    ipm_term<VAR>(i, ka, a, b, ca);
                    if (ca == 0.0) continue;
                    for (int kb = 0; kb < 3; ++kb) {
                        int p, r; double cb;
                        ipm_term<VAR>(j, kb, p, r, cb);
                        if (cb == 0.0) continue;
                        s += ca * cb * 0.25 * (Z[a][p] * Si[r][b] + Z[a][r] * Si[p][b] + Z[b][p] * Si[r][a] + Z[b][r] * Si[p][a]);
                    }
                }
                M[i][j] = s;
            }
        if (!ipm_chol(&M[0][0], NR, NR, 0.0)) break;
        // predictor (sigma = 0), then corrector with Mehrotra's sigma and second-order term
        double dZ[10][10], dS[10][10], dy[NR], Rc[10][10], T1[10][10], T2[10][10];
        double sig_mu = 0.0;
        double ap = 1.0, ad = 1.0;
        for (int pass = 0; pass < 2; ++pass) {
            for (int i = 0; i < 10; ++i)
                for (int j = 0; j < 10; ++j) Rc[i][j] = sig_mu * Si[i][j] - Z[i][j];
            if (pass == 1) { // second-order correction: - (dZ dS Si + Si dS dZ) / 2 of the predictor
                ipm_mul(dZ, dS, T1);
                ipm_mul(T1, Si, T2);
                for (int i = 0; i < 10; ++i)
                    for (int j = 0; j < 10; ++j) Rc[i][j] -= 0.5 * (T2[i][j] + T2[j][i]);
            }
            for (int i = 0; i < NR; ++i) dy[i] = -ipm_adot<VAR>(i, Rc);
            ipm_chol_solve(&M[0][0], NR, NR, dy);
            for (int i = 0; i < 10; ++i)
                for (int j = 0; j < 10; ++j) dS[i][j] = 0.0;
            ipm_adjoint<VAR>(dy, -1.0, dS);
            ipm_mul(Z, dS, T1);
            ipm_mul(T1, Si, T2);
            for (int i = 0; i < 10; ++i)
                for (int j = 0; j < 10; ++j) dZ[i][j] = Rc[i][j] - 0.5 * (T2[i][j] + T2[j][i]);
            ap = ipm_step(Z, dZ);
            ad = ipm_step(S, dS);
            if (pass == 0) {
                double g_aff = 0;
                for (int i = 0; i < 10; ++i)
                    for (int j = 0; j < 10; ++j) g_aff += (Z[i][j] + ap * dZ[i][j]) * (S[i][j] + ad * dS[i][j]);
                const double r = g_aff / gap;
                sig_mu = r * r * r * mu;
            }
        }
        double g = 0;
        for (int i = 0; i < 10; ++i)
            for (int j = 0; j < 10; ++j) g += (Z[i][j] + ap * dZ[i][j]) * (S[i][j] + ad * dS[i][j]);
        if (!(g == g) || !(g < gap)) break; // no progress: rounding has taken over; the iterate of the last good step stands
        for (int i = 0; i < 10; ++i)
            for (int j = 0; j < 10; ++j) { Z[i][j] += ap * dZ[i][j]; S[i][j] += ad * dS[i][j]; }
        for (int i = 0; i < NR; ++i) y[i] += ad * dy[i];
        gap = g;
    }
    return it;
}

// From the interior-point iterate to the reference's outputs: eigen-decomposition of Z, rank-1 rounding + Newton polish + dual
// certificate with S as the hint (or the twin-candidate logic for a rank-2 Z), exactly the decisions of cvx::solve_sdp at a
// certificate attempt; the reference's own recovery when nothing certifies.  Qs: trace-normalised cost (45 packed), tr: trace.
template <int VAR = VAR_FULL>
CVX_HD void ipm_finish(const double *Qs, double tr, const Opts &o, const double (*Zf)[10], const double (*Sf)[10], Solution &sol, double *Zout)
{
    double delta = o.eps / (8.0 * tr);
    delta = delta < 1e-13 ? 1e-13 : delta;
    const double gap_tol = (o.eps > 8e-13 * tr ? o.eps : 8e-13 * tr);
    double Zp[55], Wd[55], Wz[55];
    for (int i = 0; i < 10; ++i)
        for (int j = i; j < 10; ++j) { Zp[sidx(i, j)] = Zf[i][j]; Wd[sidx(i, j)] = -Sf[i][j]; Wz[sidx(i, j)] = 0.0; }
    Eig e;
    eig_load(e, Zp);
    sol.sweeps += eig_solve(e, 40, 1e-30);
    int jm = 0, j2 = 0;
    double best = -1, second = -1;
    for (int j = 0; j < 10; ++j) {
        const double n2 = e.n2[j];
        const bool b1 = n2 > best, b2 = !b1 && n2 > second;
        second = b1 ? best : (b2 ? n2 : second);
        j2 = b1 ? jm : (b2 ? j : j2);
        best = b1 ? n2 : best;
        jm = b1 ? j : jm;
    }
    const double l1 = sqrt(best) - e.sigma, l2 = sqrt(second) - e.sigma;
    double vt[10], v2[10];
    for (int i = 0; i < 10; ++i) { vt[i] = e.G[jm][i] / sqrt(best); v2[i] = e.G[j2][i] / sqrt(second); }
    Cert c;
    c.ok = false;
    bool ambiguous = false;
    double Rm[9];
    if (!(l2 > 0.5 * l1)) {
        const double d0 = round_candidate(vt, c.R);
        polish_rotation(Qs, c.R, c.pobj);
        dual_certificate<true, const double *, VAR>(Qs, Wd, Wz, 1.0, delta, d0, c); // hint: rho (Wp - W) = S
    } else {
        double zp[10], zm[10], fp, fm;
        twin_candidates(vt, v2, zp, zm);
        const double dp = polish_candidate(Qs, zp, c.R, fp), dm = polish_candidate(Qs, zm, Rm, fm);
        double trc = 0;
        bool fin = (fp == fp) && (fm == fm);
        for (int i = 0; i < 9; ++i) { trc += c.R[i] * Rm[i]; fin &= (c.R[i] == c.R[i]) && (Rm[i] == Rm[i]); }
        const double gtol = gap_tol / tr;
        ambiguous = fin && dp > 0 && dm > 0 && fabs(fp - fm) <= gtol && trc < 2.9;
        if (ambiguous) {
            c.pobj = fp;
            dual_certificate<true, const double *, VAR>(Qs, Wd, Wz, 1.0, delta, dp, c);
            ambiguous = c.ok && (tr * (fabs(c.zSz) + 4.0 * delta) <= gap_tol);
            if (!ambiguous) c.ok = false;
        } else {
            const bool take_m = dm > 0 && (fm == fm) && (!(dp > 0) || !(fp == fp) || fm < fp);
            if (take_m) { for (int i = 0; i < 9; ++i) c.R[i] = Rm[i]; }
            c.pobj = take_m ? fm : fp;
            dual_certificate<true, const double *, VAR>(Qs, Wd, Wz, 1.0, delta, take_m ? dm : dp, c);
        }
    }
    if (ambiguous) {
        for (int i = 0; i < 9; ++i) sol.R[i] = c.R[i];
        sol.cost = tr * c.pobj;
        sol.dobj = tr * (c.pobj - c.zSz - 4.0 * delta);
        sol.status = ST_RANK_GT1;
        sol.rank = 2;
        if (Zout) {
            double za[10], zb[10];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) { za[3 * j + i] = c.R[i * 3 + j]; zb[3 * j + i] = Rm[i * 3 + j]; }
            za[9] = 1.0; zb[9] = 1.0;
            for (int i = 0; i < 10; ++i)
                for (int j = i; j < 10; ++j) Zout[sidx(i, j)] = 0.5 * (za[i] * za[j] + zb[i] * zb[j]);
        }
        return;
    }
    const bool gap_ok = c.ok && (tr * (fabs(c.zSz) + 4.0 * delta) <= gap_tol);
    if (gap_ok) {
        for (int i = 0; i < 9; ++i) sol.R[i] = c.R[i];
        sol.cost = tr * c.pobj;
        sol.dobj = tr * (c.pobj - c.zSz - 4.0 * delta);
        sol.status = ST_CERTIFIED;
        sol.rank = 1;
        if (Zout) {
            double z[10];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) z[3 * j + i] = c.R[i * 3 + j];
            z[9] = 1.0;
            for (int i = 0; i < 10; ++i)
                for (int j = i; j < 10; ++j) Zout[sidx(i, j)] = z[i] * z[j];
        }
        return;
    }
    int rank = 0;
    const double thr = (e.sigma + 1e-3) * (e.sigma + 1e-3);
    for (int j = 0; j < 10; ++j) rank += e.n2[j] > thr;
    fallback_pose(Qs, tr, vt, v2, rank, sol);
    if (Zout) { for (int i = 0; i < 55; ++i) Zout[i] = Zp[i]; }
}

// One problem through the interior-point path: Q9 (45, unnormalised A^T A), B (27) -> Solution, like cvx::solve_sdp.
template <int VAR = VAR_FULL>
CVX_HD void ipm_problem(const double *Q9, const double *B, const Opts &o, Solution &sol, double *Zout)
{
    double tr = 0;
    for (int i = 0; i < 9; ++i) tr += Q9[qidx(i, i)];
    sol.sweeps = 0; sol.rank = 0;
    bool finite = (tr == tr) && (tr > 0) && (tr < 1e300);
    const double itr = finite ? 1.0 / tr : 0.0;
    double q[45];
    for (int i = 0; i < 45; ++i) { q[i] = Q9[i] * itr; finite &= (q[i] == q[i]); }
    if (!finite) {
        for (int i = 0; i < 9; ++i) sol.R[i] = NAN;
        for (int i = 0; i < 3; ++i) sol.t[i] = NAN;
        sol.cost = NAN; sol.dobj = NAN; sol.status = ST_NONFINITE;
        if (Zout) { for (int i = 0; i < 55; ++i) Zout[i] = NAN; }
        return;
    }
    Canon cn;
    cn.on = false;
    canonicalise_planar(q, cn);
    double Z[10][10], S[10][10], y[IPM_M], gap;
    const int nit = ipm_solve<VAR>(q, Z, S, y, 1e-10, 40, gap);
    sol.iters += nit;
    ipm_finish<VAR>(q, tr, o, Z, S, sol, Zout);
    if (cn.on) {
        canon_rotation_back(cn, sol.R);
        if (Zout) canon_congruence(cn, Zout, false);
    }
    double r[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r[3 * j + i] = sol.R[i * 3 + j];
    for (int i = 0; i < 3; ++i) {
        double acc = 0;
        for (int j = 0; j < 9; ++j) acc += B[i * 9 + j] * r[j];
        sol.t[i] = -acc;
    }
}

} // inline namespace CVX_UNIT_TAG
} // namespace cvx
