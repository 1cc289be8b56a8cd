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
inline namespace CVX_UNIT_TAG {

using namespace cvx;

#if defined(CVXL_MARKS) && defined(__HIP_DEVICE_COMPILE__)
#define CVXL_MARK(x) asm volatile("; CVXL_MARK " x)
#else
#define CVXL_MARK(x)
#endif

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

// ---------------------------------------------------------------------------------------
// The float64 instantiation (Opts::f32_sweeps_until below the length of the phase: every sweep, the product (W + sigma I) V and the
// rotation angles in float64 -- the precision of the reference, cvxpnpl.py:475-513).
//
// Register plan.  The state is W (55 doubles) + the eigen columns (100 doubles) + their norms = 330 registers; a lane owns 256 VGPRs +
// 256 AGPRs, VALU instructions read VGPRs only, and a sweep has all 200 registers of the columns as operands.  Left to itself the
// compiler's allocator cannot place this: the straightforward float64 copy of lane_phase (W, Wp and the columns: 420 registers)
// compiles to 581 spilled registers and 1 784 B of scratch per lane -- 1 500 scratch loads per wavefront and launch, each of them
// exposed latency with one wavefront per SIMD (SQ counters: the wavefronts issue 41 % of their cycles and sit in s_waitcnt for 51 %;
// the single-precision kernel issues 84 %), 838 us per 125 k launch against 284 us for the single-precision kernel with 1.44x fewer
// instructions.  Measured over the variants of this round (tools/microbench/lane_bench.*, same box, 125 k problems): launch time
// ~ 410 us + 0.27 us per byte of scratch per lane -- the spilled set no longer fits the L2 of its XCD (4 MB / 8 192 lanes = 512 B).
// What is done about it:
//   * Wp is never stored (pos_update_cols below): the update is linear in Wp and is added into W one COLUMN at a time, so that the
//     operands of every loop outside the sweeps are W and one column;
//   * W is parked BY HAND in accumulation registers across the sweeps (cvxl::Bank: "a"-class inline-asm values, one
//     v_accvgpr_write_b32 / _read_b32 per half), likewise the constraint sums of the update while the columns are added -- and nowhere
//     else: parked around the last positive part and around the certificate as well, the allocator spilled the parked values
//     themselves (796 B, 524 us; without: 468 B, 457 us);
//   * the last iteration forms Wp column by column into the registers that then hold the dual hint (dr_update_with_hint).
// Result (kernel of the library: tools/resource_table.py): ~240 spilled registers, ~470 B, ~115 scratch operations per iteration
// instead of 250; 125 k launch 838 -> 457 us (tools/microbench/lane_bench.py; the single-precision kernel: 275 us).  Measured and NOT
// kept: the whole idle state in a fixed bank of 210 AGPRs with staged swaps around the sweeps (the allocator then spills the "a"-class
// values themselves: 2 240-2 504 B, 1 100 us); parking with tied ("+a") operands (whole-kernel live ranges: same effect); the
// row-ordered two-pass update (1 784 B); walking the update in the order of the equality triples (2 152 B); a dozen -mllvm scheduling /
// allocation switches (no effect).  profiles/r04/lane_f64_variants.txt, lane_bench_2.txt.

// The bank: float64 slots in accumulation registers.  A put defines a fresh "a"-class value, a get reads it: a slot lives from its put
// to its last get, and the asm statements are volatile so that the moves stay where they are written.  Host build: a plain array.
constexpr int BANK_SLOTS = 55;
// (accumulation registers that plain VALU code can park values in exist on gfx90a / gfx942 / gfx950 -- the unified 512-entry file; any other
// target gets the plain array, like the host build)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CVXL_NO_PARK) && (defined(__gfx90a__) || defined(__gfx942__) || defined(__gfx950__))
struct Bank { unsigned r[2 * BANK_SLOTS]; };
CVX_HD void bank_put(Bank &b, int s, double x)
{
    const unsigned l = (unsigned)__double2loint(x), h = (unsigned)__double2hiint(x);
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(b.r[2 * s]) : "v"(l));
    asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(b.r[2 * s + 1]) : "v"(h));
}
CVX_HD double bank_get(const Bank &b, int s)
{
    unsigned l, h;
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(l) : "a"(b.r[2 * s]));
    asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(h) : "a"(b.r[2 * s + 1]));
    return __hiloint2double((int)h, (int)l);
}
#else
struct Bank { double v[BANK_SLOTS]; };
CVX_HD void bank_put(Bank &b, int s, double x) { b.v[s] = x; }
CVX_HD double bank_get(const Bank &b, int s) { return b.v[s]; }
#endif
// W <- A(W) = W + alpha Pi_aff(-W - Qs / rho): the part of the update that does not depend on the positive part (dr_update with Wp = 0)
template <class QV>
CVX_HD void dr_update_affine_part(double *W, QV Qs, double irho, double alpha)
{
    proj_stream([&](int i, int j) { return -W[sidx(i, j)] - irho * qent(Qs, i, j); },
                [&](int k, double, double x) { W[k] += alpha * x; }, false);
}

// signed sums of the 15 equality triples, row and column sums of the diagonal block, Z99
CVX_HD void constraint_sums(const double *W, double *m, double *rs, double *cs, double &z99)
{
    CVX_UNROLL for (int t = 0; t < 15; ++t)
        m[t] = tri_s(t, 0) * W[sidx(tri_i(t, 0), tri_j(t, 0))] + tri_s(t, 1) * W[sidx(tri_i(t, 1), tri_j(t, 1))] + tri_s(t, 2) * W[sidx(tri_i(t, 2), tri_j(t, 2))];
    CVX_UNROLL for (int i = 0; i < 3; ++i) {
        rs[i] = W[sidx(i, i)] + W[sidx(3 + i, 3 + i)] + W[sidx(6 + i, 6 + i)];
        cs[i] = W[sidx(3 * i, 3 * i)] + W[sidx(3 * i + 1, 3 * i + 1)] + W[sidx(3 * i + 2, 3 * i + 2)];
    }
    z99 = W[54];
}

// One update of the iterate straight from the eigen columns, one COLUMN at a time.  With W1 = sc W + (1 - sc) Wp (sc != 1 only at the
// tail_from switch) the update  W <- W1 + alpha (Pi_aff(2 Wp - W1 - Qs / rho) - Wp)  is affine in W and LINEAR in Wp = sum_c w_c g_c g_c^T:
//     W <- A(sc W) + [ a I - b R0 ] (Wp),      a = 1 - sc + alpha sc,   b = alpha (1 + sc),
// with A as above and R0 = I - P0 the projector onto span A_i (signed triple means; row / column means of the diagonal block; Z99).
// So: scale, apply A in place, note the constraint sums, add a w_c g_c g_c^T column by column -- the operands of that loop are W and
// ONE column --, and take R0 of what was added from the DIFFERENCE of the constraint sums after and before (they are linear).
// Nothing of the size of Wp is ever formed; rounding differs from the stored form by a few ulp of |W| (tests/hostsim: parked
// iterates agree with the general core to 1e-13).  COLS: a callable col(c, g) that yields column c (lam'_c v_c) in g[0..9].
template <class QV, class COLS, class N2T>
CVX_HD void pos_update_cols(double *W, const N2T *n2, double sigma, COLS col, QV Qs, double irho, double alpha, double sc)
{
    CVX_UNROLL for (int i = 0; i < 55; ++i) W[i] *= sc;
    dr_update_affine_part(W, Qs, irho, alpha);
    double mb[15], rb[3], cb[3], zb;
    constraint_sums(W, mb, rb, cb, zb);
#ifndef CVXL_NO_PARK_SUMS
    Bank sb; // the sums are idle while the columns are added: parked
    CVX_UNROLL for (int t = 0; t < 15; ++t) bank_put(sb, t, mb[t]);
    CVX_UNROLL for (int i = 0; i < 3; ++i) { bank_put(sb, 15 + i, rb[i]); bank_put(sb, 18 + i, cb[i]); }
    bank_put(sb, 21, zb);
#endif
    const double a = 1.0 - sc + alpha * sc, b = alpha * (1.0 + sc);
    CVX_UNROLL for (int c = 0; c < 10; ++c) {
        const double n2c = (double)n2[c];
        const double lam = sqrt_fast(n2c) - sigma;
        const double wc = lam > 0 ? a * lam * rcp(n2c) : 0.0; // a (lam'_c - sigma)_+ / lam'_c^2  (cvx::eig_pospart)
        double g[10];
        col(c, g);
        CVX_UNROLL for (int i = 0; i < 10; ++i) {
            const double wg = wc * g[i];
            CVX_UNROLL for (int k = i; k < 10; ++k) W[sidx(i, k)] = fma(wg, g[k], W[sidx(i, k)]);
        }
    }
#ifndef CVXL_NO_PARK_SUMS
    CVX_UNROLL for (int t = 0; t < 15; ++t) mb[t] = bank_get(sb, t);
    CVX_UNROLL for (int i = 0; i < 3; ++i) { rb[i] = bank_get(sb, 15 + i); cb[i] = bank_get(sb, 18 + i); }
    zb = bank_get(sb, 21);
#endif
    const double boa = b * rcp(a);
    CVX_UNROLL for (int t = 0; t < 15; ++t) {
        const double ma = tri_s(t, 0) * W[sidx(tri_i(t, 0), tri_j(t, 0))] + tri_s(t, 1) * W[sidx(tri_i(t, 1), tri_j(t, 1))] + tri_s(t, 2) * W[sidx(tri_i(t, 2), tri_j(t, 2))];
        const double cm = boa * (1.0 / 3.0) * (ma - mb[t]);
        CVX_UNROLL for (int k = 0; k < 3; ++k) W[sidx(tri_i(t, k), tri_j(t, k))] -= tri_s(t, k) * cm;
    }
    double dr[3], dc[3];
    CVX_UNROLL for (int i = 0; i < 3; ++i) {
        dr[i] = W[sidx(i, i)] + W[sidx(3 + i, 3 + i)] + W[sidx(6 + i, 6 + i)] - rb[i];
        dc[i] = W[sidx(3 * i, 3 * i)] + W[sidx(3 * i + 1, 3 * i + 1)] + W[sidx(3 * i + 2, 3 * i + 2)] - cb[i];
    }
    const double za = W[54];
    const double tot = (dr[0] + dr[1] + dr[2]) * (1.0 / 9.0);
    CVX_UNROLL for (int k = 0; k < 9; ++k) W[sidx(k, k)] -= boa * ((dr[k % 3] + dc[k / 3]) * (1.0 / 3.0) - tot);
    W[54] = zb + (1.0 - sc - alpha) * rcp(a) * (za - zb);
}

// X = Pi_aff(2 Wp - W - Qs / rho);  W <- W + alpha (X - Wp);  WpS <- rho (Wp - W_old), the dual hint, in place of Wp
template <class QV>
CVX_HD void dr_update_with_hint(double *W, double *WpS, QV Qs, double irho, double alpha, double rho)
{
    proj_stream([&](int i, int j) { return 2.0 * WpS[sidx(i, j)] - W[sidx(i, j)] - irho * qent(Qs, i, j); },
                [&](int k, double, double x) {
                    const double wp = WpS[k], w0 = W[k];
                    W[k] = w0 + alpha * (x - wp);
                    WpS[k] = rho * (wp - w0);
                },
                false);
}

// The lane phase on float64 columns.  Same requirements, inputs and outputs as lane_phase below.
template <class ST>
CVX_HD void lane_phase_f64(const ProblemView &pv, const Opts &o, Solution &sol, double *Zout, int iters, double *handoff, ST st)
{
    double tr = 0, delta, gap_tol;
    {
        double B[27], Q9[45];
        bool ok = true;
        if (pv.Q45) {
            CVX_UNROLL for (int i = 0; i < 45; ++i) Q9[i] = pv.Q45[i];
            CVX_UNROLL for (int i = 0; i < 27; ++i) B[i] = pv.B27[i];
        } else {
            ok = assemble(pv, B, Q9);
        }
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
    }
    const auto Qs = st.Q();
    delta = o.eps / (8.0 * tr);
    delta = delta < 1e-13 ? 1e-13 : delta;
    gap_tol = o.eps > 8e-13 * tr ? o.eps : 8e-13 * tr;
    double rho = o.rho, irho = 1.0 / o.rho;

    double W[55];
    CVX_UNROLL for (int i = 0; i < 55; ++i) W[i] = 0.0;
    W[54] = 1.0;
    Eig e;
    eig_unit(e);
    set_exact(e, true);
    // iteration 1: W0 = e9 e9^T is diagonal and PSD: Wp = W0, eigenvectors = unit vectors, no eigen-solve
    int it = 1;
    if (it == o.tail_from) { rho = o.rho_tail; irho = 1.0 / rho; } // (W - Wp = 0: nothing to rescale)
    {
        double Wp0[55];
        CVX_UNROLL for (int i = 0; i < 55; ++i) Wp0[i] = (i == 54) ? 1.0 : 0.0;
        (void)dr_update(W, Wp0, Qs, irho, o.alpha);
    }
    const double tol2 = o.jacobi_tol * o.jacobi_tol;
    for (;;) {
        CVXL_MARK("load_warm");
        eig_load_warm(e, W);
        CVXL_MARK("to_sweeps");
        Bank bk; // (a fresh set of values per iteration: a slot lives from its put to its get, not across the loop)
        CVX_UNROLL for (int k = 0; k < 55; ++k) bank_put(bk, k, W[k]);
        CVXL_MARK("eig_solve");
        sol.sweeps += eig_solve(e, o.sweep_schedule ? sweep_cap(it + 1, true, o.jacobi_sweeps) : o.jacobi_sweeps, tol2);
        CVXL_MARK("from_sweeps");
        CVX_UNROLL for (int k = 0; k < 55; ++k) W[k] = bank_get(bk, k);
        ++it;
        double sc = 1.0;
        if (it == o.tail_from) { sc = rho / o.rho_tail; rho = o.rho_tail; irho = 1.0 / rho; } // the dual is kept: W - Wp scales by rho / rho_tail
        if (it >= iters) break;
        CVXL_MARK("update");
        pos_update_cols(W, e.n2, e.sigma, [&](int c, double *g) { CVX_UNROLL for (int i = 0; i < 10; ++i) g[i] = e.G[c][i]; }, Qs, irho, o.alpha, sc);
        CVXL_MARK("update_done");
    }
    // ---- the one certificate attempt (it == iters == first_check).  Here: W in the bank, the ten columns in registers.
    // Unit top eigenvector of Wp, Wp column by column (the columns die as they are used), W back, the dual hint S = rho (Wp - W) in
    // place of Wp and the iterate the next phase continues from.
    CVXL_MARK("final_update");
    double S[55], vt[10]; // S: Wp first, then the dual hint

    {
        int jm = 0;
        double best = -1.0;
        CVX_UNROLL for (int j = 0; j < 10; ++j) { const bool b1 = e.n2[j] > best; best = b1 ? e.n2[j] : best; jm = b1 ? j : jm; }
        const double il1 = rsqrt_(best);
        CVX_UNROLL for (int i = 0; i < 10; ++i) {
            double s1 = e.G[0][i];
            CVX_UNROLL for (int j = 1; j < 10; ++j) s1 = (j == jm) ? e.G[j][i] : s1;
            vt[i] = s1 * il1;
        }
        CVX_UNROLL for (int i = 0; i < 55; ++i) S[i] = 0.0;
        CVX_UNROLL for (int c = 0; c < 10; ++c) {
            const double lam = sqrt_fast(e.n2[c]) - e.sigma;
            const double wc = lam > 0 ? lam * rcp(e.n2[c]) : 0.0;
            CVX_UNROLL for (int i = 0; i < 10; ++i) {
                const double wg = wc * e.G[c][i];
                CVX_UNROLL for (int k = i; k < 10; ++k) S[sidx(i, k)] = fma(wg, e.G[c][k], S[sidx(i, k)]);
            }
        }
    }
    if (it == o.tail_from) { // (the switch fell on the last iteration of the phase: rho, irho are the new ones already)
        const double sc = o.rho / o.rho_tail;
        CVX_UNROLL for (int i = 0; i < 55; ++i) W[i] = S[i] + (W[i] - S[i]) * sc;
    }
    dr_update_with_hint(W, S, Qs, irho, o.alpha, rho);
    double chk = 0.0;
    CVX_UNROLL for (int i = 0; i < 55; ++i) chk += W[i];
    const bool bad = !(chk == chk);

    CVXL_MARK("certify");
    double R[9], pobj, zSz;
    const bool certified = certify_in_place(Qs, S, vt, delta, tr, gap_tol, R, pobj, zSz);
    CVXL_MARK("certify_done");
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
        CVXL_MARK("load_warm");
        eig_load_warm_f32(e, W);
        CVXL_MARK("eig_solve");
        sol.sweeps += eig_solve(e, o.sweep_schedule ? sweep_cap(it + 1, true, o.jacobi_sweeps) : o.jacobi_sweeps, tol2); // (the wavefront pays the maximum over its 64 lanes: cvx::sweep_cap)
        CVXL_MARK("pospart");
        eig_pospart(e, Wp);
        CVXL_MARK("update");
        ++it;
        if (it == o.tail_from) { // smaller penalty from here on; the dual is kept: Wm scales by rho / rho_tail
            const double sc = rho / o.rho_tail;
            CVX_UNROLL for (int i = 0; i < 55; ++i) W[i] = Wp[i] + (W[i] - Wp[i]) * sc;
            rho = o.rho_tail;
            irho = 1.0 / rho;
        }
        if (it < iters) { const double r2 = dr_update(W, Wp, Qs, irho, o.alpha); bad |= !(r2 == r2); }
        CVXL_MARK("update_done");
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

} // inline namespace CVX_UNIT_TAG
} // namespace cvxl
