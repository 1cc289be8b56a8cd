// ipm_wave.h -- the interior-point solve of ipm_core.h with the 64 lanes of one wavefront working on ONE problem.
//
// cvxw::solve_one_wave<.., IPM = true> (cvxw::rescue_wave_kernel only) calls this once for a problem that is still open after
// opts.rescue_from first-order iterations (minimal and near-ambiguous configurations need hundreds to thousands of them, and a
// launch lasts as long as its slowest problem): ~13 second-order iterations of ~26 us each whatever the conditioning, then
// the Douglas-Rachford iteration goes on from W = Z - S / rho -- whose positive part is Z and whose dual hint rho (W+ - W) is
// S -- so that the usual attempt (rounding, Newton polish, dual certificate, twin logic for rank 2, the reference's recovery)
// finishes the problem one iteration later.  Same mathematics as cvx::ipm_solve (HKM direction, Mehrotra predictor-corrector,
// feasible start); what is different is the step length: three candidate steps each for Z and for S are Cholesky-tested
// side by side (one ten-lane group each) instead of backtracking.  Z, S, their directions and the 21 x 21 Schur matrix are
// full matrices in LDS; the Cholesky factors are computed with one row per lane in registers (chol_rows*).  Nothing of the
// first-order solver's LDS state survives the solve except the translation map and the canonical frame, which lie outside
// [0, 684) u [908, LDSW_IPM); the caller assembles the problem again.
#pragma once
#include <hip/hip_runtime.h>

#include "ipm_core.h" // (the scalar statement of the same solve; host-tested, and the constraint tables)

namespace cvxw {

#ifdef CVXW_IPM_CLOCK // diagnostic build (tools/ipm_clock.py): 100 MHz ticks per stage of the solve, summed at L[I_COL + 100 + stage]
#define IPM_CLK(k) do { const long long now_ = wall_clock64(); if (lane == 0) L[I_COL + 100 + (k)] += (double)(now_ - clk_); clk_ = now_; } while (0)
#else
#define IPM_CLK(k) do { } while (0)
#endif

constexpr int I_Z = 0, I_S = 100, I_SI = 200, I_DZ = 300, I_DS = 400, I_RC = 500, I_DY = 600, I_RHS = 624;               // [0, 684)
constexpr int I_M = 908, I_T1 = 1350, I_T2 = 1450, I_TEST = 1550, I_COL = 2150, I_TAB = 2272;                            // [908, 2304)
constexpr int I_QS = I_T1, I_W = I_M; // hand-over between cvxw::solve_pass and the solve: cost in (64), iterate out (55)
constexpr int LDS_IPM_END = 2304;
static_assert(LDS_IPM_END <= LDSW_IPM, "the interior-point solve needs LDSW_IPM doubles per wavefront");
// Where the solve keeps its matrices (doubles from the start of the wavefront's LDS slice).  Fused: inside cvxw::rescue_wave_kernel, around
// the first-order solver's state ([684, 908) survives: the translation map and the canonical frame).  Compact: cvxw::ipm_wave_kernel, which
// runs nothing else -- 1 480 doubles = 11.6 KB per wavefront, so that three wavefronts per SIMD fit the 160 KB of a CU (fused: 18 KB, two).
struct IpmLayFused { static constexpr int Z = I_Z, S = I_S, SI = I_SI, DZ = I_DZ, DS = I_DS, RC = I_RC, DY = I_DY, RHS = I_RHS, M = I_M, T1 = I_T1, T2 = I_T2, COL = I_COL, TAB = I_TAB; };
struct IpmLayCompact { static constexpr int Z = 0, S = 100, SI = 200, DZ = 300, DS = 400, RC = 500, DY = 600, RHS = 624, M = 684, T1 = 1126, T2 = 1226, TAB = 1326, COL = 1358, END = 1480; };

struct IpmTab { signed char r[21][3], c[21][3], s[21][3]; signed char ent_con[55], ent_sgn[55]; signed char d1[10], d2[10]; };
// (rows: cvx::ipm_term<VAR>; ent_con / ent_sgn: the row an off-diagonal entry belongs to and its sign; d1 / d2: the row(s) a
// diagonal entry belongs to, -1: none)
template <int VAR>
constexpr IpmTab make_ipm_tab()
{
    IpmTab t{};
    constexpr int NR = cvx::ipm_rows(VAR), T0 = VAR == cvx::VAR_RC ? 3 : 0, NT = 15 - T0; // triples T0..14 are rows 0..NT-1
    for (int i = 0; i < 21; ++i)
        for (int k = 0; k < 3; ++k) {
            int r = 9, c = 9, s = 0;
            if (i < NT) { r = cvx::tri_i(i + T0, k); c = cvx::tri_j(i + T0, k); s = cvx::tri_s(i + T0, k) < 0 ? -1 : 1; }
            else if (VAR == cvx::VAR_RC) {
                if (i < 15) { r = c = 3 * (i - 12) + k; s = 1; }
                else if (i == 15) { r = c = 9; s = k == 0 ? 1 : 0; }
            } else if (i < 18) { r = c = 3 * k + (i - 15); s = 1; }
            else if (i < 20) { r = c = 3 * (i - 18) + k; s = 1; }
            else { r = c = 9; s = k == 0 ? 1 : 0; }
            if (i >= NR) { r = c = 9; s = 0; }
            t.r[i][k] = (signed char)r; t.c[i][k] = (signed char)c; t.s[i][k] = (signed char)s;
        }
    for (int e = 0; e < 55; ++e) { t.ent_con[e] = -1; t.ent_sgn[e] = 0; }
    for (int i = 0; i < NT; ++i)
        for (int k = 0; k < 3; ++k) {
            const int e = cvx::sidx(cvx::tri_i(i + T0, k), cvx::tri_j(i + T0, k));
            t.ent_con[e] = (signed char)i;
            t.ent_sgn[e] = (signed char)(cvx::tri_s(i + T0, k) < 0 ? -1 : 1);
        }
    for (int i = 0; i < 10; ++i) {
        t.d1[i] = -1; t.d2[i] = -1;
        if (i == 9) t.d1[i] = (signed char)(NR - 1);
        else if (VAR == cvx::VAR_RC) t.d1[i] = (signed char)(12 + i / 3);
        else { t.d1[i] = (signed char)(15 + i % 3); if (i / 3 < 2) t.d2[i] = (signed char)(18 + i / 3); }
    }
    return t;
}
constexpr bool ipm_tab_ok()
{
    const IpmTab t = make_ipm_tab<cvx::VAR_FULL>();
    int hit[55] = {};
    for (int i = 0; i < 15; ++i)
        for (int k = 0; k < 3; ++k) {
            if (cvx::tri_i(i, k) == cvx::tri_j(i, k)) return false;
            ++hit[cvx::sidx(cvx::tri_i(i, k), cvx::tri_j(i, k))];
        }
    for (int e = 0; e < 55; ++e)
        if (hit[e] > 1) return false;
    // the tables restate cvx::ipm_term
    for (int v = 0; v < 2; ++v) {
        const IpmTab u = v ? make_ipm_tab<cvx::VAR_RC>() : make_ipm_tab<cvx::VAR_FULL>();
        for (int i = 0; i < cvx::ipm_rows(v); ++i)
            for (int k = 0; k < 3; ++k) {
                int r = 0, c = 0; double cf = 0;
                if (v) cvx::ipm_term<cvx::VAR_RC>(i, k, r, c, cf); else cvx::ipm_term<cvx::VAR_FULL>(i, k, r, c, cf);
                if (u.s[i][k] != (cf > 0 ? 1 : (cf < 0 ? -1 : 0))) return false;
                if (cf != 0 && (u.r[i][k] != r || u.c[i][k] != c)) return false;
            }
    }
    return t.ent_con[0] == -1;
}
static_assert(ipm_tab_ok(), "coop_ipm builds dS entry-wise: every off-diagonal entry belongs to at most one of the 15 triples");
__device__ const IpmTab kIpmTab = make_ipm_tab<cvx::VAR_FULL>();
__device__ const IpmTab kIpmTabRc = make_ipm_tab<cvx::VAR_RC>();

// Cholesky with the rows in registers: lane i < N holds row i in a[] (entries 0..i; the rest is ignored and comes back as
// garbage), lanes >= N hold zeros.  Left-looking, fully unrolled; the finished entries of row j reach the other lanes through lane
// reads -- no LDS, no barrier (a version with the matrix in LDS cost ~1 200 cycles per column in dependent round trips: 12 of the 39 us of an
// interior-point iteration went into the 21 x 21 factor).  Returns false (wave-uniform) if the matrix is not positive definite.
template <int N>
__device__ __forceinline__ bool chol_rows(double (&a)[N])
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        double s = a[j];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= a[k] * wave_lane(a[k], j);
        const double d = wave_lane(s, j);
        ok = ok && (d > 0.0);
        a[j] = s * cvx::rsqrt_(d > 0.0 ? d : 1.0); // (row == j: d / sqrt(d) = sqrt(d))
    }
    return ok;
}
// The same for several matrices side by side, one per group of N consecutive lanes starting at lane `base` (`mine`: this lane
// belongs to a group).  Right-looking, so that a column needs ONE exchange: the pivot and the column's entries, all read from
// the lanes that own them with ds_bpermute in one batch.  Returns false in the lanes of a group whose matrix is not positive definite.
template <int N>
__device__ __forceinline__ bool chol_rows_grouped(double (&a)[N], int base, bool mine)
{
    bool ok = true;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double d = __shfl(a[j], base + j);
        double c[N];
#pragma unroll
        for (int k = j + 1; k < N; ++k) c[k] = __shfl(a[j], base + k);
        ok = ok && (!mine || d > 0.0);
        const double il = cvx::rsqrt_(d > 0.0 ? d : 1.0);
        const double f = a[j] * il * il; // a_ik -= l_ij l_kj = (a_ij / d) a_kj
#pragma unroll
        for (int k = j + 1; k < N; ++k) a[k] -= f * c[k];
        a[j] *= il;
    }
    return ok;
}

// C = A B for 10 x 10 matrices in LDS (entries e and e + 64 per lane)
__device__ __forceinline__ void coop_mul10(const double *A, const double *B, double *C, int lane)
{
    for (int e = lane; e < 100; e += 64) {
        const int i = e / 10, j = e % 10;
        double s = 0;
#pragma unroll
        for (int k = 0; k < 10; ++k) s += A[i * 10 + k] * B[k * 10 + j];
        C[e] = s;
    }
    CVXW_SYNC();
}

// Step lengths for Z and S in one go: ten-lane groups 0..2 Cholesky-test Z + a dZ for a = {1, .7, .45} x scale_z, groups 3..5
// S + a dS for {1, .7, .45} x scale_s; the largest step that keeps the matrix positive definite each (0 if none).
__device__ __forceinline__ void coop_steps(double *L, const double *Z, const double *dZ, const double *S, const double *dS, int lane,
                                           double scale_z, double scale_s, double &ap, double &ad)
{
    const int g = lane / 10, r = lane % 10;
    const bool mine = lane < 60;
    const int c = g % 3;
    const double f = c == 0 ? 1.0 : (c == 1 ? 0.7 : 0.45);
    const bool isz = g < 3;
    const double cand = (isz ? scale_z : scale_s) * f;
    const double *X = isz ? Z : S, *dX = isz ? dZ : dS;
    double a[10];
#pragma unroll
    for (int j = 0; j < 10; ++j) a[j] = (mine && j <= r) ? X[r * 10 + j] + cand * dX[r * 10 + j] : (j == r ? 1.0 : 0.0);
    const bool ok = chol_rows_grouped<10>(a, mine ? g * 10 : 60, mine);
    const unsigned long long m = __ballot(mine && ok && r == 0);
    ap = ((m >> 0) & 1ull) ? scale_z : (((m >> 10) & 1ull) ? 0.7 * scale_z : (((m >> 20) & 1ull) ? 0.45 * scale_z : 0.0));
    ad = ((m >> 30) & 1ull) ? scale_s : (((m >> 40) & 1ull) ? 0.7 * scale_s : (((m >> 50) & 1ull) ? 0.45 * scale_s : 0.0));
}

// The solve.  qe: this lane's entry (ei, ej) of the trace-normalised cost (lanes < 55, 0 outside the 9 x 9 block).
// On exit Z and S (full, symmetric) are at L[I_Z], L[I_S]; returns the iterations, gap = <Z, S>.
template <int VAR, class LAY>
__device__ __forceinline__ int coop_ipm_body(double *L, int lane, double qe, int ei, int ej, double tol, int max_iters, double *gap_out)
{
    constexpr int NR = cvx::ipm_rows(VAR), NSCH = NR * (NR + 1) / 2; // constraint rows; entries of the Schur matrix's lower triangle
    const IpmTab &tab = VAR == cvx::VAR_RC ? kIpmTabRc : kIpmTab;
    constexpr int I_COL = LAY::COL; // (diagnostic build only)
    double *Z = L + LAY::Z, *S = L + LAY::S, *Si = L + LAY::SI, *dZ = L + LAY::DZ, *dS = L + LAY::DS, *Rc = L + LAY::RC, *dy = L + LAY::DY, *rhs = L + LAY::RHS;
    double *M = L + LAY::M, *T1 = L + LAY::T1, *T2 = L + LAY::T2;
    CVXW_SYNC();
    for (int e = lane; e < 100; e += 64) {
        const int i = e / 10, j = e % 10;
        Z[e] = (i == j) ? (i < 9 ? 1.0 / 3.0 : 1.0) : 0.0;
        S[e] = (i == j) ? 1.0 : 0.0;
    }
    CVXW_SYNC();
    if (lane < 55 && ej < 9) { S[ei * 10 + ej] += qe; if (ei != ej) S[ej * 10 + ei] += qe; }
    CVXW_SYNC();
    // The constraint tables, once per solve (kIpmTab lives in global memory: read inside the iteration -- nine dependent byte
    // loads per Schur entry -- it was most of the solve's time).  The 63 terms, packed r | c << 4 | (s + 1) << 8, go to LDS;
    // rhs: the terms of row `lane`;  dS entries e = lane + 64 r (r < 2): dS[e] = c1 dy[i1] + c2 dy[i2].
    auto pack_term = [&](int i, int k) { return (int)tab.r[i][k] | ((int)tab.c[i][k] << 4) | (((int)tab.s[i][k] + 1) << 8); };
    int *TI = reinterpret_cast<int *>(L + LAY::TAB);
    if (lane < 63) TI[lane] = pack_term(lane / 3, lane % 3);
    int sch_i[4], sch_j[4]; // Schur entries e = lane + 64 r (lower triangle, 231 of them): row and column
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int e = lane + 64 * r;
        int i = 0, acc = 0;
        while (acc + i + 1 <= e) { acc += i + 1; ++i; } // row i starts at i (i + 1) / 2
        sch_i[r] = i < NR ? i : 0; sch_j[r] = i < NR ? e - acc : 0;
    }
    int rhs_t[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) rhs_t[k] = lane < NR ? pack_term(lane, k) : 0x100;
    int ds_i1[2], ds_i2[2];
    double ds_c1[2], ds_c2[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int e = lane + 64 * r;
        const int i = (e < 100 ? e : 0) / 10, j = (e < 100 ? e : 0) % 10;
        ds_i1[r] = 0; ds_i2[r] = 0; ds_c1[r] = 0.0; ds_c2[r] = 0.0;
        if (i != j) {
            const int se = cvx::sidx(i, j), t = tab.ent_con[se];
            if (t >= 0) { ds_i1[r] = t; ds_c1[r] = -0.5 * (double)tab.ent_sgn[se]; }
        } else {
            if (tab.d1[i] >= 0) { ds_i1[r] = tab.d1[i]; ds_c1[r] = -1.0; }
            if (tab.d2[i] >= 0) { ds_i2[r] = tab.d2[i]; ds_c2[r] = -1.0; }
        }
    }
    auto frob = [&](const double *A, const double *B) { // <A, B> over all 100 entries
        double s = 0;
        for (int e = lane; e < 100; e += 64) s += A[e] * B[e];
        return wave_sum(s);
    };
    double gap = frob(Z, S);
    int it = 0;
#ifdef CVXW_IPM_CLOCK
    if (lane < 16) L[I_COL + 100 + lane] = 0.0;
    long long clk_ = wall_clock64();
#endif
    for (; it < max_iters; ++it) {
        if (gap < tol) break;
        const double mu = gap * 0.1;
        // ---- Si = S^-1: Cholesky in T1 (lanes 0..9 own a row), then one column of the inverse per lane
        {
            double a[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) a[k] = (lane < 10 && k <= lane) ? S[lane * 10 + k] : 0.0;
            if (!chol_rows<10>(a)) break;
#pragma unroll
            for (int k = 0; k < 10; ++k)
                if (lane < 10 && k <= lane) T1[lane * 10 + k] = a[k];
            CVXW_SYNC();
        }
        IPM_CLK(0);
        if (lane < 10) {
            double x[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                double s = (i == lane) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < 10; ++k)
                    if (k < i) s -= T1[i * 10 + k] * x[k];
                x[i] = s / T1[i * 10 + i];
            }
#pragma unroll
            for (int i = 9; i >= 0; --i) {
                double s = x[i];
#pragma unroll
                for (int k = 0; k < 10; ++k)
                    if (k > i) s -= T1[k * 10 + i] * x[k];
                x[i] = s / T1[i * 10 + i];
            }
#pragma unroll
            for (int i = 0; i < 10; ++i) Si[i * 10 + lane] = x[i];
        }
        CVXW_SYNC();
        IPM_CLK(1);
        // ---- Schur matrix M_ij = <A_i, Z A_j Si>, lower triangle: 231 entries dealt to the lanes
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = lane + 64 * r;
            if (e >= NSCH) break;
            const int i = sch_i[r], j = sch_j[r];
            int wi[3], wj[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { wi[k] = TI[i * 3 + k]; wj[k] = TI[j * 3 + k]; }
            double s = 0;
#pragma unroll
            for (int ka = 0; ka < 3; ++ka) {
                const int a = wi[ka] & 15, b = (wi[ka] >> 4) & 15;
                const double ca = (double)(((wi[ka] >> 8) & 3) - 1);
#pragma unroll
                for (int kb = 0; kb < 3; ++kb) {
                    const int p = wj[kb] & 15, q = (wj[kb] >> 4) & 15;
                    const double cb = (double)(((wj[kb] >> 8) & 3) - 1);
                    s += ca * cb * 0.25 * (Z[a * 10 + p] * Si[q * 10 + b] + Z[a * 10 + q] * Si[p * 10 + b] + Z[b * 10 + p] * Si[q * 10 + a] + Z[b * 10 + q] * Si[p * 10 + a]);
                }
            }
            M[i * NR + j] = s;
        }
        CVXW_SYNC();
        IPM_CLK(2);
        const int li = lane < NR ? lane : 0;
        double Lr[NR];
#pragma unroll
        for (int k = 0; k < NR; ++k) Lr[k] = (lane < NR && k <= lane) ? M[li * NR + k] : 0.0;
        if (!chol_rows<NR>(Lr)) break;
#pragma unroll
        for (int k = 0; k < NR; ++k)
            if (lane < NR && k <= lane) M[li * NR + k] = Lr[k]; // (the columns are read back from here)
        CVXW_SYNC();
        IPM_CLK(3);
        // the factor into registers for the two solves of this iteration: lane i keeps row i (left of the diagonal) and column i
        // (below it), zero elsewhere, and 1 / L_ii -- a substitution step is then two lane reads and one multiply-add, not two
        // round trips through LDS
        double Lc[NR];
        const double dinv = 1.0 / M[li * NR + li];
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            Lr[k] = (lane < NR && k < lane) ? Lr[k] : 0.0;
            Lc[k] = (lane < NR && k > lane) ? M[k * NR + li] : 0.0;
        }
        IPM_CLK(4);
        // ---- predictor (sigma = 0), then corrector
        double sig_mu = 0.0, ap = 0.0, ad = 0.0;
        for (int pass = 0; pass < 2; ++pass) {
            if (pass == 1) { // second-order term of the predictor: (dZ dS Si + its transpose) / 2
                coop_mul10(dZ, dS, T1, lane);
                coop_mul10(T1, Si, T2, lane);
            }
            for (int e = lane; e < 100; e += 64) {
                const int i = e / 10, j = e % 10;
                Rc[e] = sig_mu * Si[e] - Z[e] - (pass == 1 ? 0.5 * (T2[e] + T2[j * 10 + i]) : 0.0);
            }
            CVXW_SYNC();
            if (lane < NR) {
                double s = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) s += (double)(((rhs_t[k] >> 8) & 3) - 1) * Rc[(rhs_t[k] & 15) * 10 + ((rhs_t[k] >> 4) & 15)];
                rhs[lane] = -s;
            }
            CVXW_SYNC();
            {
                double r = lane < NR ? rhs[lane] : 0.0;
#pragma unroll
                for (int k = 0; k < NR; ++k) { // forward
                    const double xk = wave_lane(r, k) * wave_lane(dinv, k);
                    r = lane == k ? xk : r - Lr[k] * xk;
                }
#pragma unroll
                for (int k = NR - 1; k >= 0; --k) { // backward
                    const double xk = wave_lane(r, k) * wave_lane(dinv, k);
                    r = lane == k ? xk : r - Lc[k] * xk;
                }
                if (lane < NR) dy[lane] = r;
            }
            CVXW_SYNC();
            // dS = - sum dy_i A_i, entry-wise: an off-diagonal entry belongs to one triple, a diagonal one to a row and a column sum
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (lane + 64 * r < 100) dS[lane + 64 * r] = ds_c1[r] * dy[ds_i1[r]] + ds_c2[r] * dy[ds_i2[r]];
            CVXW_SYNC();
            IPM_CLK(5);
            coop_mul10(Z, dS, T1, lane);
            coop_mul10(T1, Si, T2, lane);
            for (int e = lane; e < 100; e += 64) {
                const int i = e / 10, j = e % 10;
                dZ[e] = Rc[e] - 0.5 * (T2[e] + T2[j * 10 + i]);
            }
            CVXW_SYNC();
            IPM_CLK(6);
            ap = 0.0; ad = 0.0;
            {
                double sz = 1.0, ss = 1.0;
                for (int round = 0; round < 4 && (ap == 0.0 || ad == 0.0); ++round) { // (wave-uniform)
                    double tp, td;
                    coop_steps(L, Z, dZ, S, dS, lane, sz, ss, tp, td);
                    if (ap == 0.0) { ap = tp; sz *= 0.3; }
                    if (ad == 0.0) { ad = td; ss *= 0.3; }
                }
            }
            IPM_CLK(7);
            if (ap < 1.0) ap *= 0.95;
            if (ad < 1.0) ad *= 0.95;
            if (pass == 0) {
                double s = 0;
                for (int e = lane; e < 100; e += 64) s += (Z[e] + ap * dZ[e]) * (S[e] + ad * dS[e]);
                const double r = wave_sum(s) / gap;
                sig_mu = r * r * r * mu;
            }
        }
        if (ap == 0.0 || ad == 0.0) break;
        double s = 0;
        for (int e = lane; e < 100; e += 64) s += (Z[e] + ap * dZ[e]) * (S[e] + ad * dS[e]);
        const double g = wave_sum(s);
        if (!(g == g) || !(g < gap)) break; // no progress: rounding has taken over; the last good iterate stands
        for (int e = lane; e < 100; e += 64) { Z[e] += ap * dZ[e]; S[e] += ad * dS[e]; }
        CVXW_SYNC();
        gap = g;
        IPM_CLK(8);
    }
    *gap_out = gap;
    return it;
}

// inside cvxw::rescue_wave_kernel: a call, not inlined -- merely compiled into the first-order loop of that kernel the solve costs the
// loop its register allocation
template <int VAR>
__device__ __noinline__ int coop_ipm(double *L, int lane, double qe, int ei, int ej, double tol, int max_iters, double *gap_out)
{
    return coop_ipm_body<VAR, IpmLayFused>(L, lane, qe, ei, ej, tol, max_iters, gap_out);
}

} // namespace cvxw
