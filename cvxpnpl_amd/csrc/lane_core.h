// lane_core.h -- the first phase of the lane-hybrid schedule, one problem per lane, written for a register budget.
//
// Same mathematics as cvx::solve_sdp<TWIN = false> (solver_core.h; reference path cvxpnpl.py:454-520) for the ONE schedule the
// launch policy uses: `iters` ADMM iterations without a certificate attempt, one attempt after the last of them (first_check ==
// hand-off point), then either the certified pose or the iterate parked for cvxw::resume_wave_kernel.  What is different from the
// general scalar core is only HOW it is written:
//   * straight line: trivial first iteration, a counted loop of eigen-solve / positive part / update, ONE certificate at the end --
//     the general core keeps the certificate inside its loop with four exits, and everything the certificate touches is live
//     across the loop (compiled for the device it needs ~1 500 registers: 256 VGPR + 256 AGPR + 1 006 spilled, 2.6 KB of scratch
//     per lane, 1.1 GB of HBM traffic per 125 k launch);
//   * no 55-entry temporaries: the affine projections (ADMM update, dual fit, dual correction) are streamed triple by triple --
//     every off-diagonal entry of Z belongs to exactly one equality triple, so a projection is "subtract the signed mean of three
//     entries" fifteen times plus the 3x3 diagonal block, with three to twelve values live instead of 55 (or 110);
//   * the persistent state is W (55), Wp (55) and the eigen columns (100 floats): 320 registers; the certificate works on S in
//     place of Wp.  The cost matrix and the translation map stay in this lane's LDS column (cvx::LdsStore).
// Host-compilable like solver_core.h (tests/hostsim steps it on the CPU against cvx::solve_problem<false>).
#pragma once
#include "problem_io.h"
#include "solver_core.h"

namespace cvxl {

using namespace cvx;

// cost entry (i <= j) of the 10x10 Qs: the packed 9x9 block, zero in the last row / column
template <class QV>
CVX_HD double qent(QV Qs, int i, int j) { return j < 9 ? Qs[qidx(i, j)] : 0.0; }

// The affine set { <A_i, Z> = b_i } (cvxpnpl.py:387-451) streamed: calls f(k, x, x_projected) for every entry k of vech order, where
// the caller's g(i, j) yields the entry x (i <= j) of the matrix to be projected.  homog: onto the direction space (targets 0).
// g is evaluated for all members of a triple (and for the whole diagonal block) before f is called for any of them, and every
// entry belongs to exactly one triple or to the diagonal block: f may overwrite what g reads.
// Fifteen disjoint triples (sign s, mean m: x_k - s_k m) and the diagonal block D[i][j] = Z[3j+i][3j+i] with row and column sums
// equal to Z99 (= 1, or 0 when homog).
template <class G, class F>
CVX_HD void proj_stream(G g, F f, bool homog)
{
    CVX_UNROLL for (int t = 0; t < 15; ++t) {
        const double x0 = g(tri_i(t, 0), tri_j(t, 0)), x1 = g(tri_i(t, 1), tri_j(t, 1)), x2 = g(tri_i(t, 2), tri_j(t, 2));
        const double m = (tri_s(t, 0) * x0 + tri_s(t, 1) * x1 + tri_s(t, 2) * x2) * (1.0 / 3.0);
        f(sidx(tri_i(t, 0), tri_j(t, 0)), x0, x0 - tri_s(t, 0) * m);
        f(sidx(tri_i(t, 1), tri_j(t, 1)), x1, x1 - tri_s(t, 1) * m);
        f(sidx(tri_i(t, 2), tri_j(t, 2)), x2, x2 - tri_s(t, 2) * m);
    }
    const double tgt = homog ? 0.0 : 1.0;
    double d[9];
    CVX_UNROLL for (int k = 0; k < 9; ++k) d[k] = g(k, k);
    const double r0 = d[0] + d[3] + d[6] - tgt, r1 = d[1] + d[4] + d[7] - tgt, r2 = d[2] + d[5] + d[8] - tgt;
    const double c0 = d[0] + d[1] + d[2] - tgt, c1 = d[3] + d[4] + d[5] - tgt, c2 = d[6] + d[7] + d[8] - tgt;
    const double tot = (r0 + r1 + r2) * (1.0 / 9.0);
    const double rs[3] = {r0, r1, r2}, cs[3] = {c0, c1, c2};
    CVX_UNROLL for (int k = 0; k < 9; ++k) f(sidx(k, k), d[k], d[k] - (rs[k % 3] + cs[k / 3]) * (1.0 / 3.0) + tot);
    f(sidx(9, 9), g(9, 9), tgt);
}

// X = Pi_aff(2 Wp - W - Qs / rho);  W <- W + alpha (X - Wp)  (cvx::solve_sdp, end of the loop body).  Returns sum (X - Wp)^2 over
// the 55 entries -- the lane phase only tests it for NaN (the Frobenius weights of the fixed-point residual are not applied).
template <class QV>
CVX_HD double dr_update(double *W, const double *Wp, QV Qs, double irho, double alpha)
{
    double r2 = 0.0;
    proj_stream([&](int i, int j) { return 2.0 * Wp[sidx(i, j)] - W[sidx(i, j)] - irho * qent(Qs, i, j); },
                [&](int k, double, double x) {
                    const double dd = x - Wp[k];
                    W[k] += alpha * dd;
                    r2 += dd * dd;
                },
                false);
    return r2;
}

// G = (W + sigma I) V in single precision (the sweeps that follow are single precision, solver_core.h EigF): in place, column j of
// e.G is lam'_j v_j from the previous eigen-solve on entry.  W is rounded to float once (55 conversions, 1 000 float FMAs instead
// of the 1 000 double ones of cvx::eig_load_warm -- what the quad kernel does as well).
CVX_HD void eig_load_warm_f32(EigF &e, const double *W)
{
    double fro = 0;
    CVX_UNROLL for (int i = 0; i < 10; ++i)
        CVX_UNROLL for (int j = i; j < 10; ++j) fro += (i == j ? 1.0 : 2.0) * W[sidx(i, j)] * W[sidx(i, j)];
    e.sigma = 1.5 * sqrt_fast(fro) + 1e-300;
    float Wf[55];
    CVX_UNROLL for (int k = 0; k < 55; ++k) Wf[k] = (float)W[k];
    const float sg = (float)e.sigma;
    CVX_UNROLL for (int j = 0; j < 10; ++j) {
#if defined(__HIP_DEVICE_COMPILE__)
        const float il_ = __builtin_amdgcn_rsqf(e.n2[j]);
#else
        const float il_ = 1.0f / sqrtf(e.n2[j]);
#endif
        float v[10], acc[10];
        CVX_UNROLL for (int i = 0; i < 10; ++i) v[i] = (float)eig_g(e, j, i) * il_;
        CVX_UNROLL for (int i = 0; i < 10; ++i) {
            float a = sg * v[i];
            CVX_UNROLL for (int m = 0; m < 10; ++m) a = fmaf(Wf[sidx(i, m)], v[m], a);
            acc[i] = a;
        }
        CVX_UNROLL for (int i = 0; i < 5; ++i) e.G[j][i] = f2_set(acc[2 * i], acc[2 * i + 1]);
    }
}

// One certificate attempt.  Mirrors the !two branch of cvx::solve_sdp + cvx::dual_certificate<SYMM = false>, streamed; the dual
// is built in place.  S: on entry the dual hint rho (Wp - W), destroyed.  vt: unit top eigenvector of Wp.  Returns true when the
// pose is certified (gap <= gap_tol); R, pobj, zSz are set either way.
template <class QV>
CVX_HD bool certify_in_place(QV Qs, double *S, const double *vt, double delta, double tr, double gap_tol, double *R, double &pobj, double &zSz)
{
    // primal half: rank-1 rounding (cvxpnpl.py:504-505), a rotation next to it, Newton polish of r^T Qs r on SO(3)
    double d0;
    {
        const double iv = rcp(vt[9]);
        double M0[9];
        CVX_UNROLL for (int i = 0; i < 3; ++i) CVX_UNROLL for (int j = 0; j < 3; ++j) M0[i * 3 + j] = vt[3 * j + i] * iv;
        d0 = det3(M0);
        if (d0 < 0) { CVX_UNROLL for (int i = 0; i < 9; ++i) M0[i] = -M0[i]; } // the polish needs SO(3); a reflection cannot certify
        near_rotation(M0, R);
    }
    polish_rotation(Qs, R, pobj);
    double z[10];
    CVX_UNROLL for (int i = 0; i < 3; ++i) CVX_UNROLL for (int j = 0; j < 3; ++j) z[3 * j + i] = R[i * 3 + j];
    z[9] = 1.0;
    // S1 = S_h - P_0(S_h - Qs): in Qs + span A_i  (P_0: projection onto { <A_i, .> = 0 })
    proj_stream([&](int i, int j) { return S[sidx(i, j)] - qent(Qs, i, j); },
                [&](int k, double, double p) { S[k] -= p; }, true);
    // correction: min-norm dS in span A_i with (S - dS) z = 0;  lam = P(R) M_I^-1 P(R)^T (S z)  (cvx::dual_lambda)
    double rhs[10], lam[10];
    sym_mul10(S, z, rhs);
    dual_lambda<VAR_FULL>(R, rhs, false, lam);
    // S2 = S1 - (E - P_0(E)),  E = sym(lam z^T)
    proj_stream([&](int i, int j) { return 0.5 * (lam[i] * z[j] + z[i] * lam[j]); },
                [&](int k, double E, double p) { S[k] -= E - p; }, true);
    double Sz[10], res = 0.0;
    sym_mul10(S, z, Sz);
    zSz = 0.0;
    CVX_UNROLL for (int i = 0; i < 10; ++i) { res = fabs(Sz[i]) > res ? fabs(Sz[i]) : res; zSz += z[i] * Sz[i]; }
    CVX_UNROLL for (int i = 0; i < 10; ++i) S[sidx(i, i)] += delta;
    const double minp = ldl_min_pivot(S); // all pivots of S + delta I positive  <=>  lambda_min(S) > -delta
    const bool ok = (minp > 0) && (res < 1e-10) && (d0 > 0) && (pobj == pobj);
    return ok && (tr * (fabs(zSz) + 4.0 * delta) <= gap_tol);
}

// The lane phase.  Requirements (checked by the launch code, which otherwise uses the general core): variant FULL,
// 2 <= iters <= 6, o.first_check == iters (one attempt, after the last iteration), o.max_iters > iters, warm start on.
// Outputs as cvx::solve_sdp: sol.status = -1 and handoff[0..54] = W, handoff[55] = iteration count when the problem is parked.
template <class ST>
CVX_HD void lane_phase(const ProblemView &pv, const Opts &o, Solution &sol, double *Zout, int iters, double *handoff, ST st)
{
    double B[27], Q9[45];
    bool ok = true;
    if (pv.Q45) {
        CVX_UNROLL for (int i = 0; i < 45; ++i) Q9[i] = pv.Q45[i];
        CVX_UNROLL for (int i = 0; i < 27; ++i) B[i] = pv.B27[i];
    } else {
        ok = assemble(pv, B, Q9);
    }
    double tr = 0;
    CVX_UNROLL for (int i = 0; i < 9; ++i) tr += Q9[qidx(i, i)];
    sol.sweeps = 0; sol.iters = 0; sol.rank = 0;
    bool finite = ok && (tr == tr) && (tr > 0) && (tr < 1e300);
    const double itr = finite ? 1.0 / tr : 0.0;
    CVX_UNROLL for (int i = 0; i < 45; ++i) { Q9[i] *= itr; finite &= (Q9[i] == Q9[i]); }
    if (!finite) { // degenerate input: NaN pose (cvxpnpl.py:493-498 / LinAlgError)
        CVX_UNROLL for (int i = 0; i < 9; ++i) sol.R[i] = NAN;
        CVX_UNROLL for (int i = 0; i < 3; ++i) sol.t[i] = NAN;
        sol.cost = NAN; sol.dobj = NAN; sol.status = ST_NONFINITE;
        if (Zout) { CVX_UNROLL for (int i = 0; i < 55; ++i) Zout[i] = NAN; }
        return;
    }
    {   // a planar scene in a general frame goes to the wave-per-problem kernel at once (cvx::solve_sdp, same test)
        double T[9], U[9];
        CVX_UNROLL for (int i = 0; i < 3; ++i)
            CVX_UNROLL for (int j = 0; j < 3; ++j) T[i * 3 + j] = Q9[qidx(3 * i, 3 * j)] + Q9[qidx(3 * i + 1, 3 * j + 1)] + Q9[qidx(3 * i + 2, 3 * j + 2)];
        if (planar_frame(T, U)) {
            CVX_UNROLL for (int i = 0; i < 55; ++i) handoff[i] = (i == 54) ? 1.0 : 0.0;
            handoff[55] = 0.0;
            sol.status = -1;
            return;
        }
    }
    CVX_UNROLL for (int i = 0; i < 45; ++i) st.setQ(i, Q9[i]);
    CVX_UNROLL for (int i = 0; i < 27; ++i) st.setB(i, B[i]);
    const auto Qs = st.Q();
    double delta = o.eps / (8.0 * tr);
    delta = delta < 1e-13 ? 1e-13 : delta;
    const double gap_tol = o.eps > 8e-13 * tr ? o.eps : 8e-13 * tr;
    double rho = o.rho, irho = 1.0 / o.rho;

    double W[55], Wp[55];
    CVX_UNROLL for (int i = 0; i < 55; ++i) { W[i] = 0.0; Wp[i] = 0.0; }
    W[54] = 1.0; Wp[54] = 1.0;
    EigF e;
    eig_unit(e);
    bool bad = false;
    // iteration 1: W0 = e9 e9^T is diagonal and PSD: Wp = W0, eigenvectors = unit vectors, no eigen-solve
    int it = 1;
    if (it == o.tail_from) { rho = o.rho_tail; irho = 1.0 / rho; } // (W - Wp = 0: nothing to rescale)
    { const double r2 = dr_update(W, Wp, Qs, irho, o.alpha); bad |= !(r2 == r2); }
    const double tol2 = o.jacobi_tol * o.jacobi_tol;
    for (; it < iters;) {
        eig_load_warm_f32(e, W);
        sol.sweeps += eig_solve(e, o.sweep_schedule ? sweep_cap(it + 1, true, o.jacobi_sweeps) : o.jacobi_sweeps, tol2); // (the wavefront pays the maximum over its 64 lanes: cvx::sweep_cap)
        eig_pospart(e, Wp);
        ++it;
        if (it == o.tail_from) { // smaller penalty from here on; the dual is kept: Wm scales by rho / rho_tail
            const double sc = rho / o.rho_tail;
            CVX_UNROLL for (int i = 0; i < 55; ++i) W[i] = Wp[i] + (W[i] - Wp[i]) * sc;
            rho = o.rho_tail;
            irho = 1.0 / rho;
        }
        if (it < iters) { const double r2 = dr_update(W, Wp, Qs, irho, o.alpha); bad |= !(r2 == r2); }
    }
    // ---- the one certificate attempt (it == iters == first_check): unit top eigenvector of Wp
    double vt[10];
    {
        int jm = 0;
        float best = -1.0f;
        CVX_UNROLL for (int j = 0; j < 10; ++j) { const bool b1 = e.n2[j] > best; best = b1 ? e.n2[j] : best; jm = b1 ? j : jm; }
        const double il1 = rsqrt_((double)best);
        CVX_UNROLL for (int i = 0; i < 5; ++i) {
            f2 s1 = e.G[0][i];
            CVX_UNROLL for (int j = 1; j < 10; ++j) { s1.x = (j == jm) ? e.G[j][i].x : s1.x; s1.y = (j == jm) ? e.G[j][i].y : s1.y; }
            vt[2 * i] = (double)s1.x * il1;
            vt[2 * i + 1] = (double)s1.y * il1;
        }
    }
    // dual hint S = rho (Wp - W) first, then the iterate the next phase continues from, W <- W + alpha (X - Wp), in place:
    // two 55-entry arrays live from here on (W, S)
    double S[55];
    CVX_UNROLL for (int i = 0; i < 55; ++i) S[i] = rho * (Wp[i] - W[i]);
    { const double r2 = dr_update(W, Wp, Qs, irho, o.alpha); bad |= !(r2 == r2); }
    double R[9], pobj, zSz;
    const bool certified = certify_in_place(Qs, S, vt, delta, tr, gap_tol, R, pobj, zSz);
    sol.iters = it;
    if (bad) {
        CVX_UNROLL for (int i = 0; i < 9; ++i) sol.R[i] = NAN;
        CVX_UNROLL for (int i = 0; i < 3; ++i) sol.t[i] = NAN;
        sol.cost = NAN; sol.dobj = NAN; sol.status = ST_NONFINITE;
        if (Zout) { CVX_UNROLL for (int i = 0; i < 55; ++i) Zout[i] = NAN; }
        return;
    }
    if (!certified) {
        CVX_UNROLL for (int i = 0; i < 55; ++i) handoff[i] = W[i];
        handoff[55] = (double)it;
        sol.status = -1;
        return;
    }
    CVX_UNROLL for (int i = 0; i < 9; ++i) sol.R[i] = R[i];
    sol.cost = tr * pobj;
    sol.dobj = tr * (pobj - zSz - 4.0 * delta);
    sol.status = ST_CERTIFIED;
    sol.rank = 1;
    double r[9];
    CVX_UNROLL for (int i = 0; i < 3; ++i) CVX_UNROLL for (int j = 0; j < 3; ++j) r[3 * j + i] = R[i * 3 + j];
    if (Zout) {
        CVX_UNROLL for (int i = 0; i < 10; ++i)
            CVX_UNROLL for (int j = i; j < 10; ++j) Zout[sidx(i, j)] = (i < 9 ? r[i] : 1.0) * (j < 9 ? r[j] : 1.0);
    }
    CVX_UNROLL for (int i = 0; i < 3; ++i) { // t = -B r (cvxpnpl.py:513)
        double acc = 0;
        CVX_UNROLL for (int j = 0; j < 9; ++j) acc += st.B(i * 9 + j) * r[j];
        sol.t[i] = -acc;
    }
}

} // namespace cvxl
