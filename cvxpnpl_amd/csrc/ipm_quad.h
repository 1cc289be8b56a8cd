// ipm_quad.h -- the interior-point solve of ipm_core.h, FOUR problems per gfx950 wavefront (one per DPP row of sixteen lanes).
//
// Until round 5 the solve ran one problem per wavefront (ipm_wave.h, cvxw::coop_ipm -- still the safety net of the workloads that hardly
// ever need it): ~13 iterations of ~25 us, nearly all of it the dependent chains of two factorisations, four triangular solves and the
// step-length tests on 10-21 of the 64 lanes, plus a Schur matrix built from ~17 000 scattered LDS reads per iteration
// (profiles/r04/split_ipm_experiment.txt: "what would help: two problems per wavefront").  Here the same chains serve four problems, the
// data-parallel parts have their operands in registers addressed at compile time, and the broadcasts of the chains are DPP moves:
//   * matrices: six full 10 x 10 matrices per problem (Z, S^-1, S, dZ, dS and a scratch matrix) in the row's LDS slice, row c (= column
//     c) written by lane c; a lane re-reads its own row where it needs it.  20 KB per wavefront: two wavefronts per SIMD;
//   * Schur matrix M_ij = <A_i, Z A_j S^-1>: lane l owns row l (and lanes 0..4 rows 16..20).  For each of the (at most three) terms
//     (a, b) of ITS row a lane loads rows a, b of Z and of S^-1 (four contiguous 80-byte reads) and then runs the SAME straight-line
//     code as every other lane: for every column j and term (p, q) of A_j -- all compile-time -- four multiply-adds on registers.
//     ~900 multiply-adds per lane and iteration instead of 1 152 eight-byte gathers with bank conflicts;
//   * dS = -sum dy_i A_i applied to a vector is straight-line code on the 21 multipliers (every off-diagonal entry belongs to one triple);
//   * factorisations: LDL^T with one row (M: two rows) per lane, right-looking; pivot and column reach the other lanes of the row by
//     DPP row_newbcast (gfx90a and later: two v_mov_b32_dpp per double, 20-26 cycles on a dependent chain where a ds_bpermute round trip
//     takes 78-87 -- tools/microbench/lat_probe.hip); forward substitution the same way, backward substitution by DPP row reductions;
//   * step lengths: three candidate steps {1, .7, .45} x scale of Z AND of S per call, each Cholesky-tested by five lanes that hold
//     two rows (r and 9 - r) -- the candidate ladder of cvxw::coop_steps; the two eliminations run interleaved, their pivot columns
//     travel through the scratch matrix;
//   * right-hand sides b_i - <A_i, Z + Rc> (the residual of the equalities goes into every step) and, when the gap does not decrease
//     although it is still large, one retry with EQUAL steps: the two changes to cvx::ipm_solve's rules, both measured (below).
// Control flow is wave-uniform; a problem that has converged (or whose factorisation failed in rounding: the last good iterate stands,
// as in ipm_core.h) idles until its three neighbours are done.  Mathematics: cvx::ipm_solve (HKM direction, Mehrotra
// predictor-corrector, feasible start), constraint rows cvx::ipm_term<VAR> (cvxpnpl.py:387-451; VAR_RC: benchmarks/toolkit/methods/rc.py:9-64).
// Measured (MI355X, profiles/r05/ipm_quad_*.txt): 8 192 solves of four-point problems in 0.79 ms (cvxw::coop_ipm: 2 048 in ~0.34 ms);
// one wavefront alone: ~50 000 cycles per iteration for its four problems (step tests 21 %, triangular solves 19 %, dS / dZ 15 %,
// Schur matrix 14 %, its factorisation 12 %).
#pragma once
#include <hip/hip_runtime.h>

#include "ipm_core.h"
#include "wave_kernel.h"

namespace cvxi {

// LDS slice of one problem (doubles): six full 10 x 10 matrices, row c (= column c) written by lane c, and the multipliers of the current
// direction.  628 * 2 dwords = 40 mod 64 banks: the four slices of a wavefront start 0, 40, 16, 56 banks in, so that the broadcast reads
// of the four rows (same offset in each slice) never collide.  20 096 B per wavefront: eight wavefronts -- two per SIMD -- fit the
// 160 KB of a CU exactly (allocation granularity 512 B).
constexpr int P_Z = 0, P_SI = 100, P_S = 200, P_DZ = 300, P_DS = 400, P_X = 500, P_DY = 600, P_SLICE = 628;

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// all-reduce over the 16 lanes of a DPP row
__device__ __forceinline__ double row_sum(double x)
{
    x += dpp_mov<0xB1>(x);
    x += dpp_mov<0x4E>(x);
    x += dpp_mov<0x141>(x);
    x += dpp_mov<0x140>(x);
    return x;
}
// value of lane K (compile-time after unrolling) of the caller's row, in every lane of the row: DPP row_newbcast (gfx90a and later; two
// v_mov_b32_dpp -- no LDS crossbar, no address register: 20-26 cycles on a dependent chain where the ds_bpermute round trip takes 78-87,
// tools/microbench/lat_probe.hip)
template <int K>
__device__ __forceinline__ double row_bcast_k(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + K, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + K, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_bcast(double v, int k) // k: a constant once the calling loop is unrolled
{
    switch (k & 15) {
    case 0: return row_bcast_k<0>(v);   case 1: return row_bcast_k<1>(v);   case 2: return row_bcast_k<2>(v);   case 3: return row_bcast_k<3>(v);
    case 4: return row_bcast_k<4>(v);   case 5: return row_bcast_k<5>(v);   case 6: return row_bcast_k<6>(v);   case 7: return row_bcast_k<7>(v);
    case 8: return row_bcast_k<8>(v);   case 9: return row_bcast_k<9>(v);   case 10: return row_bcast_k<10>(v); case 11: return row_bcast_k<11>(v);
    case 12: return row_bcast_k<12>(v); case 13: return row_bcast_k<13>(v); case 14: return row_bcast_k<14>(v); default: return row_bcast_k<15>(v);
    }
}
__device__ __forceinline__ double rcp_(double x) { return cvxw::fast_rcp(x); }

#ifdef CVXI_CLOCK // diagnostic build (tools/ipmq_clock.py): shader-clock cycles per stage of the solve, summed per wavefront
#define CVXI_CLK(k) do { const long long now_ = (long long)__builtin_readcyclecounter(); clk_[k] += now_ - clk_t_; clk_t_ = now_; } while (0)
#else
#define CVXI_CLK(k) do { } while (0)
#endif

// constraint rows as compile-time data: term k of row i is coef * sym(E_rc)
template <int VAR> struct Rows {
    static constexpr int NR = cvx::ipm_rows(VAR);
    static constexpr int r(int i, int k) { int a = 0, b = 0; double c = 0; cvx::ipm_term<VAR>(i, k, a, b, c); return a; }
    static constexpr int c(int i, int k) { int a = 0, b = 0; double cf = 0; cvx::ipm_term<VAR>(i, k, a, b, cf); return b; }
    static constexpr int s(int i, int k) { int a = 0, b = 0; double cf = 0; cvx::ipm_term<VAR>(i, k, a, b, cf); return cf > 0 ? 1 : (cf < 0 ? -1 : 0); }
};
// the terms of row i, packed r | c << 4 | (s + 1) << 8, for the lanes that look their own row up at run time
struct RowTab { int w[21][3]; };
template <int VAR>
constexpr RowTab make_row_tab()
{
    RowTab t{};
    for (int i = 0; i < 21; ++i)
        for (int k = 0; k < 3; ++k) t.w[i][k] = i < Rows<VAR>::NR ? (Rows<VAR>::r(i, k) | (Rows<VAR>::c(i, k) << 4) | ((Rows<VAR>::s(i, k) + 1) << 8)) : (9 | (9 << 4) | (1 << 8));
    return t;
}
__device__ const RowTab kRowTab = make_row_tab<cvx::VAR_FULL>();
__device__ const RowTab kRowTabRc = make_row_tab<cvx::VAR_RC>();

// y = dS x with dS = -sum_i dy_i A_i, straight-line (x, y: 10-vectors in registers; dy: the NR multipliers)
template <int VAR>
__device__ __forceinline__ void apply_dS(const double (&dy)[21], const double (&x)[10], double (&y)[10])
{
    using R = Rows<VAR>;
#pragma unroll
    for (int i = 0; i < 10; ++i) y[i] = 0.0;
#pragma unroll
    for (int i = 0; i < R::NR; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int a = R::r(i, k), b = R::c(i, k), s = R::s(i, k);
            if (s == 0) continue;
            const double v = s > 0 ? -dy[i] : dy[i];
            if (a == b) y[a] += v * x[a];
            else { y[a] += 0.5 * v * x[b]; y[b] += 0.5 * v * x[a]; }
        }
}

// LDL^T of a symmetric N x N matrix (N <= 16), row i on lane i of the DPP row: a[j] = A_ij for j <= i, 0 beyond.  On exit a[j] = L_ij
// (j < i), lane i's a[i]... is returned as dinv = 1 / d_i.  Returns whether every pivot was positive (uniform over the row).
template <int N>
__device__ __forceinline__ bool ldl_rows(double (&a)[N], int gl, double &dinv)
{
    bool ok = true;
    dinv = 1.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
        const double d = row_bcast(a[j], j);
        double c[N];
#pragma unroll
        for (int k = j + 1; k < N; ++k) c[k] = row_bcast(a[j], k);
        ok = ok && (d > 0.0);
        const double inv = rcp_(d > 0.0 ? d : 1.0);
        const double f = a[j] * inv;
#pragma unroll
        for (int k = j + 1; k < N; ++k) a[k] -= f * c[k];
        if (gl == j) dinv = inv;
        if (gl > j) a[j] = f;
    }
    return ok;
}

// The same for the Schur matrix: rows 0..15 on lanes 0..15 (lo), rows 16..NR-1 on lanes 0..NR-17 (hi).  (Tried: the pivot column through
// the slice -- one or two 8-byte writes per lane and 16-byte broadcast reads instead of two ds_bpermute per entry.  hipcc 7.2 then spills
// 234 registers INSIDE the iteration loop, against none with the register exchange: not taken.)
template <int NR>
__device__ __forceinline__ bool ldl_schur(double (&lo)[16], double (&hi)[21], int gl, double &dinv_lo, double &dinv_hi)
{
    bool ok = true;
    dinv_lo = 1.0; dinv_hi = 1.0;
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const double d = j < 16 ? row_bcast(lo[j < 16 ? j : 0], j) : row_bcast(hi[j], j - 16);
        double c[21];
#pragma unroll
        for (int k = j + 1; k < NR; ++k) c[k] = k < 16 ? row_bcast(lo[j < 16 ? j : 0], k) : row_bcast(hi[j], k - 16);
        ok = ok && (d > 0.0);
        const double inv = rcp_(d > 0.0 ? d : 1.0);
        if (j < 16) {
            const double f = lo[j < 16 ? j : 0] * inv;
#pragma unroll
            for (int k = j + 1; k < 16; ++k) lo[k] -= f * c[k];
            if (gl == j) dinv_lo = inv;
            if (gl > j) lo[j < 16 ? j : 0] = f;
        }
        if (NR > 16) {
            const double f = hi[j] * inv;
#pragma unroll
            for (int k = j + 1; k < NR; ++k) hi[k] -= f * c[k];
            if (16 + gl == j) dinv_hi = inv;
            if (16 + gl > j) hi[j] = f;
        }
    }
    return ok;
}

// x <- M^-1 x with the factor of ldl_schur: xlo belongs to row gl, xhi to row 16 + gl
template <int NR>
__device__ __forceinline__ void solve_schur(const double (&lo)[16], const double (&hi)[21], int gl, double dinv_lo, double dinv_hi, double &xlo, double &xhi)
{
#pragma unroll
    for (int i = 0; i < NR; ++i) { // L y = b: y_i is final once the steps before it are applied
        const double yi = i < 16 ? row_bcast(xlo, i) : row_bcast(xhi, i - 16);
        if (i < 16 && gl > i) xlo -= lo[i < 16 ? i : 0] * yi;
        if (NR > 16 && 16 + gl > i) xhi -= hi[i] * yi;
    }
    xlo *= dinv_lo;
    xhi *= dinv_hi;
#pragma unroll
    for (int i = NR - 1; i >= 0; --i) { // L^T x = z: the rows below i have their x, each contributes L_ki x_k
        double p = 0.0;
        if (i < 16 && gl > i) p = lo[i < 16 ? i : 0] * xlo;
        if (NR > 16 && 16 + gl > i && gl < NR - 16) p += hi[i] * xhi;
        const double s = row_sum(p);
        if (i < 16) { if (gl == i) xlo -= s; }
        else if (16 + gl == i) xhi -= s;
    }
}

// Step lengths for Z and S in one go.  For each of the two matrices: the largest of the steps {1, .7, .45} x scale that keeps X + a dX
// positive definite (0: none).  Lanes 5 c + r (c = 0..2, r = 0..4) test candidate c with rows r and 9 - r, read from the slice; the two
// eliminations are independent and run interleaved, so that one chain of column exchanges serves both.  The pivot columns travel through the scratch matrix P_X (free while steps are tested).
__device__ __forceinline__ void step_tests(const double *L, int gl, int row_lane0, double scale_z, double scale_s, double &az, double &as)
{
    const int sub = gl / 5, r = gl - 5 * sub;
    const bool mine = gl < 15;
    const double f3 = sub == 0 ? 1.0 : (sub == 1 ? 0.7 : 0.45);
    const double cz = scale_z * f3, cs = scale_s * f3;
    double lz[5], hz[10], ls[5], hs[10];
    const int rl = mine ? r : 0, rh = mine ? 9 - r : 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        lz[k] = (mine && k <= r) ? L[P_Z + rl * 10 + k] + cz * L[P_DZ + rl * 10 + k] : 0.0;
        ls[k] = (mine && k <= r) ? L[P_S + rl * 10 + k] + cs * L[P_DS + rl * 10 + k] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        hz[k] = (mine && k <= 9 - r) ? L[P_Z + rh * 10 + k] + cz * L[P_DZ + rh * 10 + k] : 0.0;
        hs[k] = (mine && k <= 9 - r) ? L[P_S + rh * 10 + k] + cs * L[P_DS + rh * 10 + k] : 0.0;
    }
    // the pivot columns travel through the scratch matrix of the slice: 20 doubles per candidate (Z: 0..9, S: 10..19; lane 15: a dummy set)
    double *B = const_cast<double *>(L) + P_X + 20 * sub;
    bool okz = true, oks = true;
#pragma unroll
    for (int j = 0; j < 10; ++j) {
        if (j < 5) { B[r] = lz[j < 5 ? j : 0]; B[10 + r] = ls[j < 5 ? j : 0]; }
        B[9 - r] = hz[j]; B[19 - r] = hs[j];
        CVXW_SYNC();
        double cz_[10], cs_[10];
#pragma unroll
        for (int k = j; k < 10; ++k) { cz_[k] = B[k]; cs_[k] = B[10 + k]; }
        CVXW_SYNC();
        const double dz = cz_[j], ds = cs_[j];
        okz = okz && (dz > 0.0);
        oks = oks && (ds > 0.0);
        const double iz = rcp_(dz > 0.0 ? dz : 1.0), is = rcp_(ds > 0.0 ? ds : 1.0);
        if (j < 5) {
            const double fz = lz[j < 5 ? j : 0] * iz, fs = ls[j < 5 ? j : 0] * is;
#pragma unroll
            for (int k = j + 1; k < 5; ++k) { lz[k] -= fz * cz_[k]; ls[k] -= fs * cs_[k]; }
        }
        const double fz = hz[j] * iz, fs = hs[j] * is;
#pragma unroll
        for (int k = j + 1; k < 10; ++k) { hz[k] -= fz * cz_[k]; hs[k] -= fs * cs_[k]; }
    }
    const unsigned long long mz = __ballot(mine && okz && r == 0) >> row_lane0, ms = __ballot(mine && oks && r == 0) >> row_lane0;
    az = (mz & 1ull) ? scale_z : (((mz >> 5) & 1ull) ? 0.7 * scale_z : (((mz >> 10) & 1ull) ? 0.45 * scale_z : 0.0));
    as = (ms & 1ull) ? scale_s : (((ms >> 5) & 1ull) ? 0.7 * scale_s : (((ms >> 10) & 1ull) ? 0.45 * scale_s : 0.0));
}

// The solve for the four problems of the wavefront.  L: the caller's slice, with S0 = Qs + I at L[P_S..] (Qs the trace-normalised cost,
// zero outside its 9 x 9 block; row c written by lane c).  live: this row has a problem.  On exit Z and S (full) are at L[P_Z..], L[P_S..];
// returns the iterations of this row's problem, gap = <Z, S> at exit.
// Register plan: what lives across an iteration is the factor of the Schur matrix (lo / hi: 74 registers); every 10 x 10 matrix lives in the
// slice and a lane re-reads its own row where it is used -- a spilled register costs this kernel a trip to scratch memory on a chain
// that has nothing to hide it behind (first build: 848 B of scratch per lane, 2.3x slower).
template <int VAR>
__device__ __forceinline__ int ipm4_solve(double *L, const int gl_in, const int row_lane0, bool live, double tol, int max_iters, double &gap_out, long long *clk_)
{
#if defined(__HIP_DEVICE_COMPILE__)
    using R = Rows<VAR>;
    constexpr int NR = R::NR;
    // Everything the code below derives from the lane's position in its row -- the masks gl > j / gl == j of 21 elimination steps, the
    // addresses of its rows, the signs of its terms -- is loop-invariant, and LLVM hoists all of it out of the iteration loop and keeps
    // it live across every phase (first build: ~90 scalar pairs and ~60 vector registers of invariants, 217 spilled registers).  So the
    // position is hidden behind an empty asm at the top of every phase: the derived values are recomputed where they are used (a
    // compare or a shift each) and die there.
    int gl = gl_in;
    int t_lo0, t_lo1, t_lo2, t_hi0, t_hi1, t_hi2;
    {
        const RowTab &tab = VAR == cvx::VAR_RC ? kRowTabRc : kRowTab;
        const int rl = gl < NR ? gl : NR - 1, rh = 16 + gl < NR ? 16 + gl : NR - 1;
        t_lo0 = tab.w[rl][0]; t_lo1 = tab.w[rl][1]; t_lo2 = tab.w[rl][2];
        t_hi0 = tab.w[rh][0]; t_hi1 = tab.w[rh][1]; t_hi2 = tab.w[rh][2];
    }
#define CVXI_REFRESH() asm volatile("" : "+v"(gl), "+v"(t_lo0), "+v"(t_lo1), "+v"(t_lo2), "+v"(t_hi0), "+v"(t_hi1), "+v"(t_hi2))
#define col (gl < 10)
#define cg (gl < 10 ? gl : 0)
#define has_lo (gl < NR)
#define has_hi (16 + gl < NR)
#ifdef CVXI_CLOCK
    long long clk_t_ = (long long)__builtin_readcyclecounter();
#else
    (void)clk_;
#endif
    auto own_row = [&](int base, double (&v)[10]) { // row cg of a matrix of the slice (zero in the lanes that own none)
#pragma unroll
        for (int i = 0; i < 10; ++i) v[i] = col ? L[base + cg * 10 + i] : 0.0;
    };
    auto put_row = [&](int base, const double (&v)[10]) {
        if (col) {
#pragma unroll
            for (int i = 0; i < 10; ++i) L[base + cg * 10 + i] = v[i];
        }
    };
    // <Z + a dZ, S + b dS> over the row's problem
    auto gap_at = [&](double a, double b2) {
        double z[10], dz[10], sv[10], ds[10];
        own_row(P_Z, z); own_row(P_DZ, dz); own_row(P_S, sv); own_row(P_DS, ds);
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < 10; ++i) acc += (z[i] + a * dz[i]) * (sv[i] + b2 * ds[i]);
        return row_sum(acc);
    };
    // Z0 = blkdiag(I_9 / 3, 1); no direction yet
    if (col) {
#pragma unroll
        for (int i = 0; i < 10; ++i) { L[P_Z + cg * 10 + i] = (i == cg) ? (cg < 9 ? 1.0 / 3.0 : 1.0) : 0.0; L[P_DZ + cg * 10 + i] = 0.0; L[P_DS + cg * 10 + i] = 0.0; }
    }
    CVXW_SYNC();
    double gap = gap_at(0.0, 0.0);
    int it = 0, why = 0; // why the solve ended: 0 iteration cap, 1 gap below tol, 2 / 3 a factorisation (S / Schur matrix) failed, 4 no step, 5 no progress
    bool done = !live;
#pragma unroll 1
    for (int iter = 0; iter < max_iters; ++iter) {
        if (!done && gap < tol) { done = true; why = 1; }
        if (!__any(!done)) break; // wave-uniform
        const double mu = gap * 0.1;
        CVXI_REFRESH();
        // ---- S = L D L^T, then row gl of S^-1
        {
            double dinv_s;
            double a[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) a[k] = (col && k <= gl) ? L[P_S + cg * 10 + k] : 0.0;
            const bool ok = ldl_rows<10>(a, gl, dinv_s);
            if (!ok && !done) { done = true; why = 2; } // (S is positive definite by construction: rounding only)
            if (col) {
#pragma unroll
                for (int k = 0; k < 10; ++k) L[P_X + cg * 10 + k] = (k == gl) ? dinv_s : a[k];
            }
            CVXW_SYNC();
            double x[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) x[i] = (i == gl) ? 1.0 : 0.0;
#pragma unroll
            for (int j2 = 0; j2 < 9; ++j2) { // L y = e: column-oriented (each y_j is final when its turn comes)
#pragma unroll
                for (int i = j2 + 1; i < 10; ++i) x[i] -= L[P_X + i * 10 + j2] * x[j2];
            }
#pragma unroll
            for (int i = 0; i < 10; ++i) x[i] *= L[P_X + i * 10 + i];
#pragma unroll
            for (int k = 9; k >= 1; --k) { // L^T x = z, column-oriented as well: x_k is final, rows above it take their share
#pragma unroll
                for (int i = 0; i < k; ++i) x[i] -= L[P_X + k * 10 + i] * x[k];
            }
            CVXW_SYNC();
            put_row(P_SI, x);
            CVXW_SYNC();
        }
        CVXI_CLK(0);
        CVXI_REFRESH();
        // ---- Schur matrix, row gl (lo) and row 16 + gl (hi)
        double lo[16], hi[21];
        {
            double acc[21];
#pragma unroll
            for (int j2 = 0; j2 < 21; ++j2) acc[j2] = 0.0;
#pragma unroll 1
            for (int ka = 0; ka < 3; ++ka) {
                const int w = ka == 0 ? t_lo0 : (ka == 1 ? t_lo1 : t_lo2);
                const int a = w & 15, b = (w >> 4) & 15;
                const double ca = 0.25 * (double)(((w >> 8) & 3) - 1);
                double za[10], zb[10], sa[10], sb[10];
#pragma unroll
                for (int p = 0; p < 10; ++p) { za[p] = ca * L[P_Z + a * 10 + p]; zb[p] = ca * L[P_Z + b * 10 + p]; sa[p] = L[P_SI + a * 10 + p]; sb[p] = L[P_SI + b * 10 + p]; }
#pragma unroll
                for (int j2 = 0; j2 < NR; ++j2)
#pragma unroll
                    for (int kb = 0; kb < 3; ++kb) {
                        const int p = R::r(j2, kb), q = R::c(j2, kb), sg = R::s(j2, kb);
                        if (sg == 0) continue;
                        const double v = za[p] * sb[q] + za[q] * sb[p] + zb[p] * sa[q] + zb[q] * sa[p];
                        acc[j2] += sg > 0 ? v : -v;
                    }
            }
#pragma unroll
            for (int j2 = 0; j2 < 16; ++j2) lo[j2] = (has_lo && j2 <= gl) ? acc[j2] : 0.0;
#pragma unroll
            for (int j2 = 0; j2 < 21; ++j2) hi[j2] = 0.0;
            if (NR > 16) {
                // the entries (16 + g, j <= 15) of the hi rows are the transposed tail of the lo rows: through the scratch matrix
#pragma unroll
                for (int t = 0; t < NR - 16; ++t) L[P_X + gl * 5 + t] = acc[16 + t];
                CVXW_SYNC();
                const int g5 = has_hi ? gl : 0;
#pragma unroll
                for (int j2 = 0; j2 < 16; ++j2) hi[j2] = has_hi ? L[P_X + j2 * 5 + g5] : 0.0;
                CVXW_SYNC();
                double acc2[5];
#pragma unroll
                for (int j2 = 0; j2 < 5; ++j2) acc2[j2] = 0.0;
#pragma unroll 1
                for (int ka = 0; ka < 3; ++ka) {
                    const int w = ka == 0 ? t_hi0 : (ka == 1 ? t_hi1 : t_hi2);
                    const int a = w & 15, b = (w >> 4) & 15;
                    const double ca = 0.25 * (double)(((w >> 8) & 3) - 1);
                    double za[10], zb[10], sa[10], sb[10];
#pragma unroll
                    for (int p = 0; p < 10; ++p) { za[p] = ca * L[P_Z + a * 10 + p]; zb[p] = ca * L[P_Z + b * 10 + p]; sa[p] = L[P_SI + a * 10 + p]; sb[p] = L[P_SI + b * 10 + p]; }
#pragma unroll
                    for (int j2 = 16; j2 < NR; ++j2)
#pragma unroll
                        for (int kb = 0; kb < 3; ++kb) {
                            const int p = R::r(j2, kb), q = R::c(j2, kb), sg = R::s(j2, kb);
                            if (sg == 0) continue;
                            const double v = za[p] * sb[q] + za[q] * sb[p] + zb[p] * sa[q] + zb[q] * sa[p];
                            acc2[j2 - 16] += sg > 0 ? v : -v;
                        }
                }
#pragma unroll
                for (int j2 = 16; j2 < NR; ++j2) hi[j2] = (has_hi && j2 <= 16 + gl) ? acc2[j2 - 16] : 0.0;
            }
        }
        CVXI_CLK(1);
        CVXI_REFRESH();
        double dinv_lo, dinv_hi;
        if (!ldl_schur<NR>(lo, hi, gl, dinv_lo, dinv_hi) && !done) { done = true; why = 3; }
        CVXI_CLK(2);
        // ---- predictor (sigma = 0), then corrector
        double sig_mu = 0.0, ap = 0.0, ad = 0.0;
#pragma unroll 1
        for (int pass = 0; pass < 2; ++pass) {
            CVXI_REFRESH();
            // Right-hand side b_i - <A_i, Z + Rc> rather than -<A_i, Rc>: the same for a feasible Z, and an infeasibility that the solves of
            // an ill-conditioned Schur system leave in <A_i, Z> is taken out again by the next step instead of adding up (2.6e-7 after
            // 20 iterations without this).  b_i = 1 on the rows of diagonal sums and of Z_99, 0 on the triples.
            double rc[10];
            {
                double Zc[10], Sic[10], corr[10], rz[10];
                own_row(P_Z, Zc); own_row(P_SI, Sic); own_row(P_X, corr); // (P_X: the second-order term, left there by the predictor pass)
#pragma unroll
                for (int i = 0; i < 10; ++i) { rz[i] = sig_mu * Sic[i] - (pass == 1 ? corr[i] : 0.0); rc[i] = rz[i] - Zc[i]; }
                CVXW_SYNC();
                put_row(P_X, rz);
                CVXW_SYNC();
            }
            constexpr int NB0 = VAR == cvx::VAR_RC ? 12 : 15; // first row with b_i = 1
            double xlo = gl >= NB0 ? 1.0 : 0.0, xhi = 16 + gl >= NB0 ? 1.0 : 0.0;
            {
                const int tl[3] = {t_lo0, t_lo1, t_lo2}, th[3] = {t_hi0, t_hi1, t_hi2};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    xlo -= (double)(((tl[k] >> 8) & 3) - 1) * L[P_X + (tl[k] & 15) * 10 + ((tl[k] >> 4) & 15)];
                    xhi -= (double)(((th[k] >> 8) & 3) - 1) * L[P_X + (th[k] & 15) * 10 + ((th[k] >> 4) & 15)];
                }
            }
            if (!has_lo) xlo = 0.0;
            if (!has_hi) xhi = 0.0;
            CVXI_REFRESH();
            solve_schur<NR>(lo, hi, gl, dinv_lo, dinv_hi, xlo, xhi);
            if (has_lo) L[P_DY + gl] = xlo;
            if (has_hi) L[P_DY + 16 + gl] = xhi;
            CVXW_SYNC();
            CVXI_CLK(3);
            CVXI_REFRESH();
            double t1[10];
            {
                double dy[21];
#pragma unroll
                for (int i = 0; i < 21; ++i) dy[i] = i < NR ? L[P_DY + i] : 0.0;
                CVXW_SYNC();
                {
                    double Sic[10];
                    own_row(P_SI, Sic);
                    apply_dS<VAR>(dy, Sic, t1);
                }
                {
                    double e[10], dS[10];
#pragma unroll
                    for (int i = 0; i < 10; ++i) e[i] = (i == gl) ? 1.0 : 0.0;
                    apply_dS<VAR>(dy, e, dS);
                    put_row(P_DS, dS);
                }
            }
            // dZ = Rc - sym(Z dS S^-1): row gl
            {
                double t2[10];
#pragma unroll
                for (int r = 0; r < 10; ++r) {
                    double s = 0.0;
#pragma unroll
                    for (int c = 0; c < 10; ++c) s += L[P_Z + r * 10 + c] * t1[c];
                    t2[r] = s;
                }
                CVXW_SYNC(); // (every gather of rc from P_X is long done; the barrier orders the overwrite for the compiler)
                put_row(P_X, t2); // (column cg of T2, stored as row cg)
                CVXW_SYNC();
                double dZ[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) dZ[i] = rc[i] - 0.5 * (t2[i] + L[P_X + i * 10 + cg]);
                CVXW_SYNC();
                put_row(P_DZ, dZ);
                CVXW_SYNC();
            }
            CVXI_CLK(4);
            // ---- step lengths
            ap = 0.0; ad = 0.0;
            {
                double sz = 1.0, ss = 1.0;
#pragma unroll 1
                for (int round = 0; round < 8; ++round) { // (the ladder goes down to 0.45 x 0.3^7 = 1e-4 of the full step; rounds beyond the first run only while some problem of the wavefront has found no step yet)
                    if (!__any(!done && (ap == 0.0 || ad == 0.0))) break;
                    CVXI_REFRESH();
                    double tp, td;
                    step_tests(L, gl, row_lane0, sz, ss, tp, td);
#ifdef CVXI_CLOCK
                    clk_[8 + pass] += 1;
#endif
                    if (ap == 0.0) { ap = tp; sz *= 0.3; }
                    if (ad == 0.0) { ad = td; ss *= 0.3; }
                }
                if (ap < 1.0) ap *= 0.95;
                if (ad < 1.0) ad *= 0.95;
            }
            CVXI_CLK(5);
            if (pass == 0) {
                const double r = gap_at(ap, ad) / gap;
                sig_mu = r * r * r * mu;
                // second-order term of the corrector: sym(dZ dS S^-1) of THIS direction; it waits in P_X (row cg) for the corrector's rc
                CVXI_REFRESH();
                double t1b[10];
                {
                    double dy[21], Sic[10];
#pragma unroll
                    for (int i = 0; i < 21; ++i) dy[i] = i < NR ? L[P_DY + i] : 0.0;
                    own_row(P_SI, Sic);
                    apply_dS<VAR>(dy, Sic, t1b);
                }
                double t2[10];
#pragma unroll
                for (int r2 = 0; r2 < 10; ++r2) {
                    double s2 = 0.0;
#pragma unroll
                    for (int c = 0; c < 10; ++c) s2 += L[P_DZ + r2 * 10 + c] * t1b[c];
                    t2[r2] = s2;
                }
                CVXW_SYNC();
                put_row(P_X, t2);
                CVXW_SYNC();
                double corr[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) corr[i] = 0.5 * (t2[i] + L[P_X + i * 10 + cg]);
                CVXW_SYNC();
                put_row(P_X, corr);
                CVXW_SYNC();
                CVXI_CLK(6);
            }
        }
        if ((ap == 0.0 || ad == 0.0) && !done) { done = true; why = 4; }
        {
            double g = gap_at(ap, ad);
            // No progress while the gap is still large is not rounding: the two steps are taken to the edge of the cone separately, and
            // <dZ, S> or <Z, dS> alone may be positive.  With EQUAL steps the gap is gap + a (sigma mu n - gap) (the directions are
            // orthogonal): the shorter step for both, then halved -- measured: 1 of 1 000 four-point problems (5 of 1 000 in the
            // 16-equality variant) stopped at a gap of 1e-2 ... 1e-5 without this.
#pragma unroll 1
            for (int t = 0; t < 4; ++t) {
                const bool retry = !done && !(g < gap) && gap > 1e-7;
                if (!__any(retry)) break;
                const double a = (t == 0 ? 1.0 : 0.5) * (ap < ad ? ap : ad);
                if (retry) { ap = a; ad = a; }
                const double g2 = gap_at(ap, ad);
                if (retry) g = g2;
            }
            if ((!(g == g) || !(g < gap)) && !done) { done = true; why = 5; } // no progress: rounding has taken over; the last good iterate stands
            if (!done) {
                double z[10], dz[10], sv[10], ds[10];
                own_row(P_Z, z); own_row(P_DZ, dz); own_row(P_S, sv); own_row(P_DS, ds);
#pragma unroll
                for (int i = 0; i < 10; ++i) { z[i] += ap * dz[i]; sv[i] += ad * ds[i]; }
                put_row(P_Z, z); put_row(P_S, sv);
                gap = g;
                ++it;
            }
        }
        CVXW_SYNC();
        CVXI_CLK(7);
    }
    gap_out = gap;
    return it | (why << 8);
#undef CVXI_REFRESH
#undef col
#undef cg
#undef has_lo
#undef has_hi
#else
    (void)L; (void)gl_in; (void)row_lane0; (void)live; (void)tol; (void)max_iters; (void)clk_; gap_out = 0; return 0;
#endif
}

// ---------------------------------------------------------------------------------------
// The kernel of the split interior-point path (workloads whose problems really go through the solve: at most six correspondences, the
// 16-equality variant -- cvxpnpl_hip.hip).  It consumes the rescue queue in groups of four (same self-cleaning discipline as
// cvxw::resume_body: block i owns positions 4 i .. 4 i + 3 and then draws further groups), reads the cost a first-order kernel left in
// the problem's slot (cvxw::solve_pass: the trace-normalised cost in the solver's frame at RS_W.., its iteration count at RS_IT), writes
// W = Z - S / rho -- positive part Z, dual hint rho (W+ - W) = S -- with a NEGATIVE iteration count into the slot and appends the
// problem to the RESUME queue: the plain cvxw::resume_wave_kernel launched behind it makes the attempt one iteration later with the
// code every other problem runs through (solve_pass: post_ipm).
#ifndef CVXI_OCC
#define CVXI_OCC 2
#endif
constexpr int IPMQ_GRID_MAX = 1024 * CVXI_OCC;
struct IpmQuadArgs {
    int64_t batch;
    double rho, rho_tail;
    int tail_from;
    int32_t *rq_count, *rq_entries; // consumed: the rescue queue
    int32_t *count, *entries;       // produced: the resume queue
    double *ws;
    int stride;
    int entries_cap;                // positions of rq_entries that may be read
    // diagnostic entry (cvxpnpl_ipm_batch): costs in, iterates out, no queues
    const double *qs_in;
    double *z_out, *s_out, *gap_out;
    int32_t *it_out;
};

template <int VAR>
__global__ void __launch_bounds__(64, CVXI_OCC) ipm_quad_kernel(IpmQuadArgs k)
{
    __shared__ __attribute__((aligned(16))) double lds[4 * P_SLICE];
    const int lane = threadIdx.x & 63, grp = lane >> 4, gl = lane & 15;
    const int row_lane0 = lane & 48;
    double *L = lds + grp * P_SLICE;
    const bool direct = k.qs_in != nullptr;
    int64_t q = blockIdx.x;
    int32_t b;
    if (direct) b = 4 * q + grp < k.batch ? (int32_t)(4 * q + grp) : -1;
    else {
        if (k.rq_entries[4 * q] < 0) return; // nothing queued for this block: leave without touching anything
        b = k.rq_entries[4 * q + grp];
    }
    const int gsz = (int)gridDim.x;
    const int pushed = direct ? 0 : __hip_atomic_load(k.rq_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (stable until the last block resets it)
    for (;;) { // wave-uniform
        if (!direct && gl == 0) k.rq_entries[4 * q + grp] = -1;
        const bool live = b >= 0 && b < k.batch;
        const double *src = direct ? k.qs_in + (int64_t)(live ? b : 0) * 55 : k.ws + (int64_t)(live ? b : 0) * k.stride + cvxw::RS_W;
        if (gl < 10) {
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const bool in = live && gl < 9 && i < 9;
                const double qv = in ? __hip_atomic_load(src + cvx::sidx(i, gl < 9 ? gl : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
                L[P_S + gl * 10 + i] = qv + ((i == gl) ? 1.0 : 0.0);
            }
        }
        const int it0 = (!direct && live) ? (int)__hip_atomic_load(k.ws + (int64_t)b * k.stride + cvxw::RS_IT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        double gap;
        long long clk[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const int nit_why = ipm4_solve<VAR>(L, gl, row_lane0, live, 1e-10, 40, gap, clk);
        const int nit = nit_why & 255;
        if (live && gl < 10) {
            if (direct) {
#pragma unroll
                for (int i = 0; i < 10; ++i) { k.z_out[(int64_t)b * 100 + gl * 10 + i] = L[P_Z + gl * 10 + i]; k.s_out[(int64_t)b * 100 + gl * 10 + i] = L[P_S + gl * 10 + i]; }
                if (gl == 0) { k.it_out[b] = nit_why; k.gap_out[b] = gap; } // (iterations | reason << 8)
#ifdef CVXI_CLOCK
                if (gl == 0) { for (int c = 0; c < 10; ++c) k.s_out[(int64_t)b * 100 + c] = (double)clk[c]; }
#endif
            } else {
                double *slot = k.ws + (int64_t)b * k.stride;
                const int it1 = it0 + nit;
                const double rho = (k.tail_from > 0 && it1 >= k.tail_from) ? k.rho_tail : k.rho; // the penalty the resumed solve runs with
                const double irho = 1.0 / rho;
#pragma unroll
                for (int i = 0; i < 10; ++i)
                    if (i <= gl) slot[cvxw::RS_W + cvx::sidx(i, gl)] = L[P_Z + gl * 10 + i] - L[P_S + gl * 10 + i] * irho;
                if (gl == 0) slot[cvxw::RS_IT] = -(double)(it1 > 0 ? it1 : 1);
            }
        }
        CVXW_SYNC();
        if (direct) return;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (live && gl == 0) {
            const int p = atomicAdd(k.count, 1);
            k.entries[p] = b;
        }
        CVXW_SYNC();
        int pn = 0;
        if (lane == 0) pn = atomicAdd(k.rq_count + 1, 1);
        q = (int64_t)gsz + __builtin_amdgcn_readfirstlane(pn);
        const bool in_range = 4 * q + 3 < (int64_t)k.entries_cap;
        const int32_t first = in_range ? k.rq_entries[4 * q] : -1;
        if (first < 0) break; // an empty group: the queue is exhausted -- every drawing block ends with exactly one such draw
        b = k.rq_entries[4 * q + grp];
    }
    if (lane == 0) {
        const int groups = (pushed + 3) / 4;
        const int drawing = groups < gsz ? groups : gsz;
        if (atomicAdd(k.rq_count + 2, 1) == drawing - 1) {
            __hip_atomic_store(k.rq_count, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(k.rq_count + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(k.rq_count + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

} // namespace cvxi
