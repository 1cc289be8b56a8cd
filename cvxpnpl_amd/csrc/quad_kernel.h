// quad_kernel.h -- sixteen lanes per problem, four problems per gfx950 wavefront (and, as a template parameter, twelve lanes per
// problem, five per wavefront: CVXPNPL_LAYOUT_PENTA, an experiment that stayed an experiment -- see Geo<> below and DESIGN.md section 3).
//
// The 10x10 SDP is too small for 64 lanes: in the wave-per-problem layout most instructions are 3x3 /
// scalar algebra replicated in every lane, and the kernel is VALU-issue bound on that redundancy.
// Here a DPP row (16 lanes) owns one problem, so every replicated instruction serves four problems:
//   * eigen-solve: lane j < 10 owns one COLUMN of G = (W + sigma I) V in registers.  A one-sided Jacobi rotation needs the partner's
//     column.  Nine steps per sweep, every pair of columns exactly once (kATab): steps 0, 2, 4, 6, 8 pair lanes (2k, 2k + 1) and read
//     the partner through DPP quad_perm [1,0,3,2] -- in single precision as an OPERAND of the FMAs themselves (v_fmac_f32_dpp: no
//     exchange instruction at all), in double as 22 v_mov_b32_dpp; steps 1, 3, 5, 7 fetch it with ds_bpermute (11 / 22 per step, no LDS
//     memory, no barrier) and may leave a lane with the PARTNER's rotated column, which is what lines the next DPP step up.  Both lanes
//     of a pair compute the same rotation.  (Round 5; until then all nine steps went through ds_bpermute -- round-robin order --
//     tools/microbench/eig16x.hip, profiles/r05/eig16x.txt: LDS round trips per sweep 9 -> 4);
//   * entry work (affine projection, ADMM update, PSD reconstruction, LDL^T): the 55 entries of the
//     symmetric iterate are dealt round-robin, entry e to lane e % 16 (3-4 entries per lane);
//   * reductions over a problem are 4-step DPP butterflies inside the row (quad_perm, row_half_mirror,
//     row_mirror); the four problems of a wave run in lock step, control flow is wave-uniform and a
//     finished problem just idles until its three neighbours are done.
// The kernel runs the first `handoff_at` ADMM iterations (certificate attempts from first_check on);
// the few problems that are not certified by then park their iterate in ws[] and the wavefront finishes them
// itself, one at a time, in the wave-per-problem layout (cvxw::solve_one_wave: twin candidates, slow tails
// and the reference's uncertified exits live there).  Planar scenes are the exception: recognised before the
// first iteration, they are queued for cvxw::resume_wave_kernel, launched behind this kernel.
// Mathematics identical to solver_core.h / wave_kernel.h; see those files for the derivations.
#pragma once
#include <hip/hip_runtime.h>

#include "problem_io.h"
#include "solver_core.h"
#include "wave_kernel.h"

namespace cvxq {

using cvxw::WaveArgs;

// LDS slice per problem, in doubles (even offsets: 16-byte aligned): 3.1 KB, 12.4 KB per wave, so that LDS
// never limits occupancy (12 waves fit the 160 KB of a CU; registers allow 8).  388 * 2 dwords = 8 mod 64 banks: the four slices of a wave start
// 8 banks apart.  Regions are reused across phases (noted per region).
constexpr int Q_WF = 0;    // 100  full 10x10: W (warm start), S (certificate); assembly: records, then the 60 Gram sums
constexpr int Q_QF = 100;  // 100  full Qs, row stride 10, zero last column (certificate mat-vecs); assembly: records
constexpr int Q_Y = 200;   // 124  eigen columns g [column][row] + weights (100..109); certificate scratch
constexpr int Q_X = Q_Y + 40; // 56 entry scratch of the affine projection (over C_YV.. : never live together)
constexpr int Q_B = 324;   // 28   translation map B (27)
constexpr int Q_M = 352;   // 36   0.. R out, 12.. previous polished R, 22.. candidate eigenvector
#ifndef CVXQ_SLICE
#define CVXQ_SLICE 388
#endif
constexpr int QLDS = CVXQ_SLICE; // (A/B knob: slice stride in doubles; 400 puts the two slices of a 32-lane half 32 banks apart)
constexpr int C_RED = 388;  // 12   (twelve-lane groups only) partial values of a reduction over the group
constexpr int QLDS12 = 404; // slice of a twelve-lane group: 404 * 2 dwords = 40 mod 64 banks, the six slices of a wave start 0, 40, 16, 56, 32, 8 banks in
// Lanes per problem: 16 (a DPP row; four problems per wavefront) or 12 (five problems per wavefront + four dummy lanes that own a
// sixth LDS slice and never write results): 2 000 instead of 2 500 wavefronts for 10 k problems -- one round of the chip's 2 048
// slots instead of two -- for 5 instead of 4 entries per lane and group reductions through LDS instead of DPP butterflies.
template <int LPP> struct Geo {
    static constexpr int NPW = LPP == 16 ? 4 : 5;            // problems per wavefront
    static constexpr int NGRP = LPP == 16 ? 4 : 6;           // LDS slices per wavefront (incl. the dummy group of lanes 60..63)
    static constexpr int EPL = (55 + LPP - 1) / LPP;         // entries of the symmetric iterate per lane
    static constexpr int SLICE = LPP == 16 ? QLDS : QLDS12;  // doubles of LDS per group
    static constexpr int M27 = (27 + LPP - 1) / LPP, M40 = (40 + LPP - 1) / LPP, M36 = (36 + LPP - 1) / LPP, M60 = (60 + LPP - 1) / LPP;
};
constexpr int C_XV = Q_Y;         // 40  x-vectors z, a_0, a_1, a_2 (stride 10)
constexpr int C_YV = Q_Y + 40;    // 40  Qs x
constexpr int C_H1 = Q_Y + 80;    // 10  a_k . Q a_l
constexpr int C_RL = Q_Y + 90;    // 10  R (row-major)
constexpr int C_ROW = Q_Y + 100;  // 12  pivot row of the elimination / gathered rhs
constexpr int C_LAM = Q_Y + 112;  // 10  multipliers of the dual correction

// entry e of the symmetric 10x10 (vech order): ei | ej << 4 | p1 << 8 | p2 << 14 | neg0 << 20 | neg1 << 21 |
// neg2 << 22 | diag << 23, with (p1, p2, signs) the other two members of its equality triple
struct ETab { unsigned w[64]; };
constexpr ETab make_etab()
{
    const cvxw::LaneTab t = cvxw::make_lane_tab();
    ETab o{};
    for (int e = 0; e < 64; ++e) {
        const int s = e < 55 ? e : 0;
        o.w[e] = (unsigned)t.ei[s] | ((unsigned)t.ej[s] << 4) | ((unsigned)t.p1[s] << 8) | ((unsigned)t.p2[s] << 14) |
                 ((unsigned)(t.s0[s] < 0) << 20) | ((unsigned)(t.s1[s] < 0) << 21) | ((unsigned)(t.s2[s] < 0) << 22) |
                 ((unsigned)(t.diag[s] != 0) << 23);
    }
    return o;
}
__device__ const ETab kETab = make_etab();

// The alternating ordering of the sweeps (round 5).  A sweep is nine steps: A X A X A X A X A.  An A step pairs the columns held by lanes
// (2k, 2k + 1) of the group (DPP quad_perm [1,0,3,2]).  An X step s = 0..3 pairs lane l with lane x_partner(s, l) through ds_bpermute and,
// where x_take(s, l) is set, BOTH lanes of the pair keep the other one's rotated column (each has both columns in registers after the
// exchange, so the swap is free) -- that is what makes the next A step's neighbours a set of pairs that have not met yet.  Found by search
// (tools/microbench/jacobi_orderings.py: P1, M2, P3, ..., P9 a 1-factorisation of K10 with P(t+2) = f(P(t)), f a sub-involution of M(t+1));
// the table is in terms of LANES, so it is a full sweep from any placement of the columns, in particular from the one the previous sweep
// left.  Word l: bits 4s..4s+3 = partner lane of X step s, bit 16 + s = take.  Lanes 10..15 (and 10, 11 of a twelve-lane group) pair
// with themselves.
struct ATab { unsigned w[16]; };
constexpr ATab make_atab()
{
    // X steps as matchings of the COLUMNS (columns named by the lane that holds them at the start of the sweep) + which pairs swap
    constexpr int M[4][5][2] = {{{0, 2}, {1, 3}, {4, 6}, {5, 8}, {7, 9}}, {{0, 4}, {1, 6}, {2, 9}, {3, 5}, {7, 8}},
                                {{0, 9}, {1, 8}, {2, 4}, {3, 7}, {5, 6}}, {{0, 8}, {1, 5}, {2, 7}, {3, 6}, {4, 9}}};
    constexpr unsigned SW[4] = {0xd, 0x1a, 0x1b, 0x1d};
    int lane_of[10] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9};
    ATab t{};
    for (int l = 0; l < 16; ++l) t.w[l] = l >= 10 ? (unsigned)(l | (l << 4) | (l << 8) | (l << 12)) : 0u;
    for (int s = 0; s < 4; ++s)
        for (int k = 0; k < 5; ++k) {
            const int a = M[s][k][0], b = M[s][k][1], la = lane_of[a], lb = lane_of[b];
            t.w[la] |= (unsigned)lb << (4 * s);
            t.w[lb] |= (unsigned)la << (4 * s);
            if ((SW[s] >> k) & 1u) {
                t.w[la] |= 1u << (16 + s);
                t.w[lb] |= 1u << (16 + s);
                lane_of[a] = lb;
                lane_of[b] = la;
            }
        }
    return t;
}
__device__ const ATab kATab = make_atab();
// every pair of columns meets exactly once per sweep, whatever the placement (checked at compile time from the lane-level table)
constexpr bool atab_is_a_sweep()
{
    const ATab t = make_atab();
    int col[10] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9};
    bool met[10][10] = {};
    int n = 0;
    for (int st = 0; st < 9; ++st) {
        if ((st & 1) == 0) {
            for (int k = 0; k < 5; ++k) { const int a = col[2 * k], b = col[2 * k + 1]; if (met[a][b]) return false; met[a][b] = met[b][a] = true; ++n; }
        } else {
            const int s = st >> 1;
            for (int l = 0; l < 10; ++l) {
                const int p = (int)((t.w[l] >> (4 * s)) & 15u);
                if (p >= 10 || p == l || (int)((t.w[p] >> (4 * s)) & 15u) != l || ((t.w[l] >> (16 + s)) & 1u) != ((t.w[p] >> (16 + s)) & 1u)) return false;
                if (p > l) {
                    const int a = col[l], b = col[p];
                    if (met[a][b]) return false;
                    met[a][b] = met[b][a] = true; ++n;
                    if ((t.w[l] >> (16 + s)) & 1u) { col[l] = b; col[p] = a; }
                }
            }
        }
    }
    return n == 45;
}
static_assert(atab_is_a_sweep(), "kATab: a sweep must rotate every pair of columns exactly once");

// x-vector table of the polish: element (v, i) of z / a_k as sign * R[src]  (cvxw::kXTab), 40 entries
// dealt to lane idx % 16

template <int CTRL>
__device__ __forceinline__ double dpp_mov(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// all-reduce over the 16 lanes of a DPP row: quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ double row_sum(double x)
{
    x += dpp_mov<0xB1>(x);
    x += dpp_mov<0x4E>(x);
    x += dpp_mov<0x141>(x);
    x += dpp_mov<0x140>(x);
    return x;
}
__device__ __forceinline__ double row_max(double x)
{
    x = fmax(x, dpp_mov<0xB1>(x));
    x = fmax(x, dpp_mov<0x4E>(x));
    x = fmax(x, dpp_mov<0x141>(x));
    x = fmax(x, dpp_mov<0x140>(x));
    return x;
}
__device__ __forceinline__ double bperm(int addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
typedef float f2 __attribute__((ext_vector_type(2))); // packed single precision (v_pk_mul_f32 / v_pk_fma_f32)
__device__ __forceinline__ float bpermf(int addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); }
__device__ __forceinline__ double flip(double x, unsigned neg) // neg in {0, 1}
{
    return __hiloint2double(__double2hiint(x) ^ (int)(neg << 31), __double2loint(x));
}
template <int LPP>
__device__ __forceinline__ unsigned grp_bits(unsigned long long m, int grp) { return (unsigned)(m >> (LPP * grp)) & ((1u << LPP) - 1u); }
// all-reduce over the lanes of a group: DPP butterflies inside the row (16) or through the group's LDS slice (12; wave-uniform call)
template <int LPP>
__device__ __forceinline__ double grp_sum(double *L, int gl, double x)
{
    if (LPP == 16) return row_sum(x);
    L[C_RED + gl] = x;
    CVXW_SYNC();
    const double2 *r = reinterpret_cast<const double2 *>(L + C_RED);
    const double2 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4], f = r[5];
    CVXW_SYNC();
    return ((a.x + a.y) + (b.x + b.y)) + ((c.x + c.y) + (d.x + d.y)) + ((e.x + e.y) + (f.x + f.y));
}
template <int LPP>
__device__ __forceinline__ double grp_max(double *L, int gl, double x)
{
    if (LPP == 16) return row_max(x);
    L[C_RED + gl] = x;
    CVXW_SYNC();
    const double2 *r = reinterpret_cast<const double2 *>(L + C_RED);
    const double2 a = r[0], b = r[1], c = r[2], d = r[3], e = r[4], f = r[5];
    CVXW_SYNC();
    return fmax(fmax(fmax(a.x, a.y), fmax(b.x, b.y)), fmax(fmax(fmax(c.x, c.y), fmax(d.x, d.y)), fmax(fmax(e.x, e.y), fmax(f.x, f.y))));
}

// entries owned by a lane
// Only the packed words are state; everything else is re-derived where it is used (a few bit-field extracts), and the
// main loop hides the words behind an empty asm once per iteration so that the compiler cannot hoist the derived values
// out of the loop and carry ~28 registers of them across every phase (they were the bulk of the kernel's spills).
template <int LPP>
struct Own {
    static constexpr int EPL = Geo<LPP>::EPL;
    unsigned pk[EPL];
    int gl;
    __device__ __forceinline__ int e(int m) const { return gl + LPP * m; }
    __device__ __forceinline__ int ei(int m) const { return (int)(pk[m] & 15u); }
    __device__ __forceinline__ int ej(int m) const { return (int)((pk[m] >> 4) & 15u); }
    __device__ __forceinline__ bool ok(int m) const { return gl + LPP * m < 55; }
    __device__ __forceinline__ double wgt(int m) const { return ok(m) ? (ei(m) == ej(m) ? 1.0 : 2.0) : 0.0; }
    __device__ __forceinline__ void refresh()
    {
#pragma unroll
        for (int m = 0; m < EPL; ++m) asm volatile("" : "+v"(pk[m]));
    }
};

// Projection of the symmetric matrix held 3-4 entries per lane onto { <A_i, Z> = b_i } (tgt = 1) or its
// direction space (tgt = 0); closed form of cvx::proj_affine.
// VAR_RC (the reference's "rc" ablation, benchmarks/toolkit/methods/rc.py:9-64): the row-orthonormality rows are absent -- the entries
// inside one diagonal 3x3 block (triples 0..2) are free and a diagonal entry only sees its column sum (cvxw::coop_proj).
template <int VAR = cvx::VAR_FULL, class OWN>
__device__ __forceinline__ void quad_proj(double *L, const OWN &w, double *X, double tgt)
{
#pragma unroll
    for (int m = 0; m < OWN::EPL; ++m)
        if (w.ok(m)) L[Q_X + w.e(m)] = X[m];
    CVXW_SYNC();
    double d[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) d[k] = L[Q_X + cvx::sidx(k, k)];
    const double r0 = d[0] + d[3] + d[6] - tgt, r1 = d[1] + d[4] + d[7] - tgt, r2 = d[2] + d[5] + d[8] - tgt;
    const double c0 = d[0] + d[1] + d[2] - tgt, c1 = d[3] + d[4] + d[5] - tgt, c2 = d[6] + d[7] + d[8] - tgt;
    const double tot = r0 + r1 + r2;
#pragma unroll
    for (int m = 0; m < OWN::EPL; ++m) {
        const unsigned pk = w.pk[m];
        const int ri = w.ei(m) % 3, ci = w.ei(m) / 3; // diagonal entry (ei, ei), ei < 9, is D[ri][ci]
        const double rr = ri == 0 ? r0 : (ri == 1 ? r1 : r2), cc = ci == 0 ? c0 : (ci == 1 ? c1 : c2);
        const double xdiag = (w.ei(m) == 9) ? tgt : X[m] - (VAR == cvx::VAR_RC ? cc * (1.0 / 3.0) : (rr + cc) * (1.0 / 3.0) - tot * (1.0 / 9.0));
        const double x1 = L[Q_X + ((pk >> 8) & 63)], x2 = L[Q_X + ((pk >> 14) & 63)];
        const unsigned n0 = (pk >> 20) & 1;
        const double mm = (flip(X[m], n0) + flip(x1, (pk >> 21) & 1) + flip(x2, (pk >> 22) & 1)) * (1.0 / 3.0);
        const bool free_entry = VAR == cvx::VAR_RC && w.ei(m) != w.ej(m) && w.ej(m) < 9 && (w.ei(m) / 3 == w.ej(m) / 3);
        X[m] = ((pk >> 23) & 1) ? xdiag : (free_entry ? X[m] : X[m] - flip(mm, n0));
    }
    CVXW_SYNC();
}

// Rotation of one one-sided Jacobi step seen from ONE of the two lanes of a pair: d = |other|^2 - |own|^2,
// gam = own . other.  own' = c own - s other; the partner, with d -> -d, gets s -> -s: together the same
// plane rotation as cvx::jacobi_cs (the inner one, |angle| <= 45 degrees).  tie_neg breaks d == 0 consistently (the pair must not both
// pick +s).  Half-angle form (round 5): with h = sqrt(d^2 + (2 gam)^2), cos^2 = (h + |d|) / 2h and sin cos = gam / h -- two v_rsq and no
// v_rcp on the dependent chain (until round 5: t = 2 gam / (|d| + h) by v_rsq, v_rcp, then c = rsq(1 + t^2): three transcendentals in a
// row; tools/microbench/eig16x.hip: -1 % / -6 % per sweep in single / double precision).  t = s / c is still returned: the norms are
// updated incrementally, |own'|^2 = |own|^2 - t gam, |other'|^2 = |other|^2 + t gam.
// Single precision throughout (see the eigen-solve in solve_quad_kernel): v_rsq_f32 is good to 1 ulp.
__device__ __forceinline__ void pair_cs(float d, float gam, bool rot, bool tie_neg, float &c, float &s, float &t)
{
    const float g2 = 2.0f * gam;
    const float h2 = d * d + g2 * g2 + 1e-37f;
    const float rh = __builtin_amdgcn_rsqf(h2);       // 1 / h
    const float c2 = fmaf(0.5f * fabsf(d), rh, 0.5f);  // cos^2 in [1/2, 1]
    const float rc = __builtin_amdgcn_rsqf(c2);       // 1 / cos
    float sf = gam * rh * rc;                          // sin = (sin cos) / cos
    const bool neg = d < 0.0f || (d == 0.0f && tie_neg);
    sf = neg ? -sf : sf;
    c = rot ? c2 * rc : 1.0f;
    s = rot ? sf : 0.0f;
    t = s * rc;
}

// The same in float64 throughout (F64SW instantiation: Opts::f32_sweeps_until below the length of this phase -- the reference's
// precision).  ONE full-precision reciprocal root on the chain instead of two: with h ~ sqrt(d^2 + 4 gam^2) from the raw hardware seed
// (v_rsq_f64, ~2^-23) and u = h + |d|,
//     c = u / sqrt(u^2 + 4 gam^2),   s = +-2 gam / sqrt(u^2 + 4 gam^2)
// is a rotation to the precision of that one root WHATEVER the error of h (c^2 + s^2 = 1 identically); the error of h only moves the angle
// by ~1e-7 of itself, i.e. leaves 1e-7 of the pair's inner product standing -- the sweeps end at |cos| < jacobi_tol = 6e-2 and start from
// the previous iteration's vectors.  dl: the change of the squared norms under the rotation that is actually applied,
// |own'|^2 = |own|^2 + dl, |other'|^2 = |other|^2 - dl (exact for any angle; the norms are recomputed once per sweep as before).
__device__ __forceinline__ void pair_cs_f64(double d, double gam, bool rot, bool tie_neg, double &c, double &s, double &dl)
{
    const double g2 = 2.0 * gam;
    const double g22 = g2 * g2;
    const double h2 = fma(d, d, g22); // (zero only where rot is false: the NaN it breeds is selected away below)
#if defined(__HIP_DEVICE_COMPILE__)
    const double h = h2 * __builtin_amdgcn_rsq(h2);
#else
    const double h = sqrt(h2);
#endif
    const double u = h + fabs(d);
    const double w = cvx::rsqrt_(fma(u, u, g22));
    const bool neg = d < 0.0 || (d == 0.0 && tie_neg);
    const double sf = (neg ? -g2 : g2) * w;
    c = rot ? u * w : 1.0;
    s = rot ? sf : 0.0;
    dl = s * fma(s, d, -(c * g2));
}

// ---- the steps of a sweep (see kATab).  real: this lane holds a column (gl < 10) and has a partner.
// DPP as an OPERAND of the arithmetic (VOP2 DPP encoding): hipcc 7.2 does not fold v_mov_b32_dpp into the consuming FMA, so the two
// blocks are written by hand.  s_nop 4 in front: the hazard recogniser does not look inside inline asm (VALU write -> DPP read of the
// same VGPR needs 2 wait states, EXEC write -> DPP 5).
#define CVXQ_DPP_XOR1 "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:0"
// g0 + g1 = sum_i e_i * e_i[partner]
#define CVXQ_DOT10_DPP(g0, g1, e)                                                                                                   \
    asm volatile("s_nop 4\n\t"                                                                                                      \
                 "v_mul_f32_dpp %0, %2, %2 " CVXQ_DPP_XOR1 "\n\tv_mul_f32_dpp %1, %3, %3 " CVXQ_DPP_XOR1 "\n\t"                     \
                 "v_fmac_f32_dpp %0, %4, %4 " CVXQ_DPP_XOR1 "\n\tv_fmac_f32_dpp %1, %5, %5 " CVXQ_DPP_XOR1 "\n\t"                   \
                 "v_fmac_f32_dpp %0, %6, %6 " CVXQ_DPP_XOR1 "\n\tv_fmac_f32_dpp %1, %7, %7 " CVXQ_DPP_XOR1 "\n\t"                   \
                 "v_fmac_f32_dpp %0, %8, %8 " CVXQ_DPP_XOR1 "\n\tv_fmac_f32_dpp %1, %9, %9 " CVXQ_DPP_XOR1 "\n\t"                   \
                 "v_fmac_f32_dpp %0, %10, %10 " CVXQ_DPP_XOR1 "\n\tv_fmac_f32_dpp %1, %11, %11 " CVXQ_DPP_XOR1                       \
                 : "=&v"(g0), "=&v"(g1)                                                                                             \
                 : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(e[4]), "v"(e[5]), "v"(e[6]), "v"(e[7]), "v"(e[8]), "v"(e[9]))
// n_i += k * e_i[partner]
#define CVXQ_AXPY10_DPP(n, e, k)                                                                                                    \
    asm volatile("s_nop 4\n\t"                                                                                                      \
                 "v_fmac_f32_dpp %0, %10, %20 " CVXQ_DPP_XOR1 "\n\tv_fmac_f32_dpp %1, %11, %20 " CVXQ_DPP_XOR1 "\n\t"               \
                 "v_fmac_f32_dpp %2, %12, %20 " CVXQ_DPP_XOR1 "\n\tv_fmac_f32_dpp %3, %13, %20 " CVXQ_DPP_XOR1 "\n\t"               \
                 "v_fmac_f32_dpp %4, %14, %20 " CVXQ_DPP_XOR1 "\n\tv_fmac_f32_dpp %5, %15, %20 " CVXQ_DPP_XOR1 "\n\t"               \
                 "v_fmac_f32_dpp %6, %16, %20 " CVXQ_DPP_XOR1 "\n\tv_fmac_f32_dpp %7, %17, %20 " CVXQ_DPP_XOR1 "\n\t"               \
                 "v_fmac_f32_dpp %8, %18, %20 " CVXQ_DPP_XOR1 "\n\tv_fmac_f32_dpp %9, %19, %20 " CVXQ_DPP_XOR1                        \
                 : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]), "+v"(n[4]), "+v"(n[5]), "+v"(n[6]), "+v"(n[7]), "+v"(n[8]), "+v"(n[9]) \
                 : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(e[4]), "v"(e[5]), "v"(e[6]), "v"(e[7]), "v"(e[8]), "v"(e[9]), "v"(k))

// A step, single precision: lanes (2k, 2k + 1); no exchange instruction
__device__ __forceinline__ void jstep_dpp(f2 (&q)[5], float &alf, bool real, bool tie_neg, bool active, float tol2, bool &coarse)
{
#if defined(__HIP_DEVICE_COMPILE__)
    float e[10], g0, g1;
#pragma unroll
    for (int i = 0; i < 5; ++i) { e[2 * i] = q[i].x; e[2 * i + 1] = q[i].y; }
    CVXQ_DOT10_DPP(g0, g1, e);
    const float gam = g0 + g1;
    const float be = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(alf), 0xB1, 0xF, 0xF, true));
    const float g2 = gam * gam, ab = alf * be;
    coarse |= real && g2 > tol2 * ab;
    float c, sn, t;
    pair_cs(be - alf, gam, active && real && g2 > 1e-30f * ab, tie_neg, c, sn, t);
    const f2 cc = {c, c};
    const float ms = -sn;
    float n[10];
#pragma unroll
    for (int i = 0; i < 5; ++i) { const f2 p = cc * q[i]; n[2 * i] = p.x; n[2 * i + 1] = p.y; }
    CVXQ_AXPY10_DPP(n, e, ms);
#pragma unroll
    for (int i = 0; i < 5; ++i) { q[i].x = n[2 * i]; q[i].y = n[2 * i + 1]; }
    alf = fmaf(-t, gam, alf);
#endif
}
// X step, single precision: partner through ds_bpermute (11), take = keep the partner's rotated column
__device__ __forceinline__ void jstep_bperm(f2 (&q)[5], float &alf, int addr, bool real, bool tie_neg, bool take, bool active, float tol2, bool &coarse)
{
    f2 oq[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) { oq[i].x = bpermf(addr, q[i].x); oq[i].y = bpermf(addr, q[i].y); }
    const float be = bpermf(addr, alf);
    f2 acc = q[0] * oq[0];
#pragma unroll
    for (int i = 1; i < 5; ++i) acc = __builtin_elementwise_fma(q[i], oq[i], acc);
    const float gam = acc.x + acc.y;
    const float g2 = gam * gam, ab = alf * be;
    coarse |= real && g2 > tol2 * ab;
    float c, sn, t;
    pair_cs(be - alf, gam, active && real && g2 > 1e-30f * ab, tie_neg, c, sn, t);
    // own' = c own - s other ; other' = s own + c other
    const float ka = take ? sn : c, kb = take ? c : -sn;
    const f2 ka2 = {ka, ka}, kb2 = {kb, kb};
#pragma unroll
    for (int i = 0; i < 5; ++i) q[i] = __builtin_elementwise_fma(ka2, q[i], kb2 * oq[i]);
    alf = take ? fmaf(t, gam, be) : fmaf(-t, gam, alf);
}
// the same two steps in float64 (22 v_mov_b32_dpp / 22 ds_bpermute)
// (the ten products on two accumulators: a dependent float64 FMA takes 8 cycles, tools/microbench/lat_probe.hip -- one chain of ten sits on
//  the critical path of every step between the exchange and the rotation parameters)
__device__ __forceinline__ double dot10_f64(const double (&a)[10], const double (&b)[10])
{
    double s0 = a[0] * b[0], s1 = a[1] * b[1];
#pragma unroll
    for (int i = 2; i < 10; i += 2) { s0 = fma(a[i], b[i], s0); s1 = fma(a[i + 1], b[i + 1], s1); }
    return s0 + s1;
}
__device__ __forceinline__ void jstep_dpp(double (&qd)[10], double &alq, bool real, bool tie_neg, bool active, double tol2, bool &coarse)
{
    double oq[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) oq[i] = dpp_mov<0xB1>(qd[i]);
    const double be = dpp_mov<0xB1>(alq);
    const double gam = dot10_f64(qd, oq);
    const double g2 = gam * gam, ab = alq * be;
    coarse |= real && g2 > tol2 * ab;
    double c, sn, t;
    pair_cs_f64(be - alq, gam, active && real && g2 > 1e-30 * ab, tie_neg, c, sn, t);
#pragma unroll
    for (int i = 0; i < 10; ++i) qd[i] = c * qd[i] - sn * oq[i];
    alq += t; // (t: pair_cs_f64's dl)
}
__device__ __forceinline__ void jstep_bperm(double (&qd)[10], double &alq, int addr, bool real, bool tie_neg, bool take, bool active, double tol2, bool &coarse)
{
    double oq[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) oq[i] = bperm(addr, qd[i]);
    const double be = bperm(addr, alq);
    const double gam = dot10_f64(qd, oq);
    const double g2 = gam * gam, ab = alq * be;
    coarse |= real && g2 > tol2 * ab;
    double c, sn, t;
    pair_cs_f64(be - alq, gam, active && real && g2 > 1e-30 * ab, tie_neg, c, sn, t);
    const double ka = take ? sn : c, kb = take ? c : -sn;
#pragma unroll
    for (int i = 0; i < 10; ++i) qd[i] = fma(ka, qd[i], kb * oq[i]);
    alq = take ? be - t : alq + t; // (t: pair_cs_f64's dl)
}
// one sweep: A X A X A X A X A
template <class COL, class T>
__device__ __forceinline__ void jacobi_sweep(COL &q, T &al, unsigned atab, int gl, int lane_base4, bool active, T tol2, bool &coarse)
{
    const bool col_lane = gl < 10;
#pragma unroll
    for (int st = 0; st < 9; ++st) {
        if ((st & 1) == 0) {
            jstep_dpp(q, al, col_lane, (gl & 1) != 0, active, tol2, coarse);
        } else {
            const int xs = st >> 1;
            const int partner = (int)((atab >> (4 * xs)) & 15u);
            jstep_bperm(q, al, lane_base4 + (partner << 2), partner != gl, gl > partner, ((atab >> (16 + xs)) & 1u) != 0, active, tol2, coarse);
        }
    }
}

// In-place LDL^T of the symmetric matrix held 3-4 entries per lane, pivot rows broadcast through LDS;
// returns the smallest pivot (cvx::ldl_min_pivot).
template <class OWN>
__device__ __forceinline__ double quad_ldl(double *L, const OWN &w, double *Me)
{
    double minp = 1e300;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
#pragma unroll
        for (int m = 0; m < OWN::EPL; ++m)
            if (w.ok(m) && w.ei(m) == k) L[C_ROW + w.ej(m)] = Me[m];
        CVXW_SYNC();
        const double d = L[C_ROW + k];
        minp = d < minp ? d : minp;
        const double id = cvxw::fast_rcp(d);
#pragma unroll
        for (int m = 0; m < OWN::EPL; ++m) {
            const double ra = L[C_ROW + w.ei(m)], rb = L[C_ROW + w.ej(m)];
            if (w.ei(m) > k) Me[m] -= ra * id * rb;
        }
        CVXW_SYNC();
    }
    return minp;
}


// NPW problems per wavefront (four 16-lane or five 12-lane groups): problem b = NPW * blockIdx.x + group.
// Second phase: the wavefront finishes the problems it could not certify within handoff_at iterations itself,
// one after the other in the wave-per-problem layout (low latency per problem; they are the slow / ambiguous
// ones, twin logic included).  The iterate travels through ws[] (written with park(), read back with
// device-scope loads: no cache maintenance, only program order).
// Tried first: a device-wide queue from which idle wavefronts steal parked problems.  The cache line with
// its head / tail words bounces between the eight XCDs' L2s: 1.6 ms per launch for 2500 wavefronts, measured.
// Inlined (round 2): as a separate function the second phase paid the calling convention -- ~110 callee-saved VGPRs
// stored on entry and reloaded on exit, 47 KB of scratch traffic per call: 4.7 MB of HBM writes and 3.4 MB of reads
// per 10 k launch from the ~75 wavefronts that get here, more than all inputs and outputs together (PMC: 14.6 ->
// 8.7 MB per launch, and 1 % faster).  Its spills (46 registers) sit in this tail only; the quad loop has none.
// (History of the separate function: handing down the kernel's own a / o made every wavefront of the grid write them
// to its scratch frame at kernel entry, 190 B per lane = 30 MB of HBM writes per 10 k launch; re-reading them from the
// kernarg segment inside the callee fixed that.)
__device__ __forceinline__ void park(double *p, double x) { __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// All kernel arguments as ONE struct, so that the kernarg segment IS this struct: the second phase reads what it needs
// straight from there (scalar loads through the segment pointer the kernel hands down) instead of from copies on the
// stack.  History: handing down references to the kernel's own by-value arguments made every wavefront of the grid
// write them to its scratch frame at kernel entry (190 B per lane = 30 MB of HBM writes per 10 k launch in round 1;
// copies made inside the calling branch fixed that until the option block grew and LLVM hoisted part of the copy back
// to the entry: 68 B per lane again).
struct QuadArgs {
    WaveArgs a;
    cvx::Opts o;
    int handoff_at;
    int32_t *qcount, *qentries;
    double *ws;
};
typedef const __attribute__((address_space(4))) QuadArgs *QuadArgsPtr;

template <int NPW, int VAR>
__device__ __forceinline__ void finish_own(QuadArgsPtr kp, unsigned parked, double *lds)
{
#if defined(__HIP_DEVICE_COMPILE__) // (the host pass cannot copy out of the constant address space; it never calls this)
    const WaveArgs a = kp->a; // scalar loads from the kernarg segment, only in the wavefronts that get here
    const cvx::Opts o = kp->o;
    const double *ws = kp->ws;
    const int64_t b0 = (int64_t)blockIdx.x * NPW;
    for (int g = 0; g < NPW; ++g) { // wave-uniform
        if (!((parked >> g) & 1u)) continue;
        cvxw::solve_one_wave<VAR>(a, o, b0 + g, lds, ws + (b0 + g) * cvxw::RS_FULL, true);
        CVXW_SYNC();
    }
#endif
}

// MODE 0: the schedule described above.  MODE 2 (experiment, round 4: layouts 11 / 12): the first phase with its certificate attempts, but
// every survivor is queued for the resume kernel instead of being finished by its own wavefront -- the kernel then holds no
// wave-per-problem code and can run three wavefronts per SIMD (one round of 3 072 slots for the 2 500 wavefronts of a 10 k launch).
// MODE 1 (experiment builds, layout 9): the iterations only -- no certificate code is compiled in, every problem is parked after handoff_at
// iterations and queued for the resume kernel behind the launch, which makes the first attempt: the "one-round geometry" of round 5
// (146 registers, three wavefronts per SIMD, 3 072 slots for the 2 500 wavefronts of a 10 k launch; profiles/r05/one_round.txt).
// F64SW: the Jacobi sweeps, G = (W + sigma I) V and the warm-start eigenvectors in float64 (see pair_cs_f64)
template <int MODE, int OCC = 2, int LPP = 16, bool F64SW = false, int VAR = cvx::VAR_FULL>
__global__ void __launch_bounds__(64, OCC) solve_quad_kernel(QuadArgs k)
{
    typedef Geo<LPP> G_;
    constexpr int NPW = G_::NPW, EPL = G_::EPL, SLICE = G_::SLICE;
    const WaveArgs &a = k.a;
    const cvx::Opts &o = k.o;
    const int handoff_at = k.handoff_at;
    int32_t *const qcount = k.qcount, *const qentries = k.qentries;
    double *const ws = k.ws;
    __shared__ __attribute__((aligned(16))) double lds_all[G_::NGRP * SLICE];
    const int lane = threadIdx.x & 63;
    const int grp = LPP == 16 ? lane >> 4 : (lane * 43) >> 9; // lane / 12: lanes 60..63 are a dummy sixth group (own LDS slice, no problem)
    const int gl = lane - grp * LPP;
    double *L = lds_all + grp * SLICE;
    double2 *L2 = reinterpret_cast<double2 *>(L);
    const int64_t b_raw = (int64_t)blockIdx.x * NPW + grp;
    const bool gvalid = grp < NPW && b_raw < a.batch;
    const int64_t b = gvalid ? b_raw : a.batch - 1; // surplus rows redo the last problem, outputs suppressed
#ifdef CVXQ_TIMELINE // diagnostics build (tools/timeline.py): shader-clock stamps of this wavefront, written over cost[] at the end
    unsigned long long tl_[4];
    // s_memrealtime: the 100 MHz reference clock, one time base for the whole device (s_memtime runs per XCD / SE)
    auto tl_now = []() { unsigned long long t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory"); return t; };
    tl_[0] = tl_now();
#endif

    // ---------------------------------------------------------------- roles
    Own<LPP> w;
    w.gl = gl;
#pragma unroll
    for (int m = 0; m < EPL; ++m) w.pk[m] = kETab.w[gl + LPP * m];
    const unsigned atab = kATab.w[gl];
    const int lane_base4 = (grp * LPP) << 2;

    // ---------------------------------------------------------------- assembly (cvxpnpl.py:20-153, :545-549)
    bool okK = true, okG = true;
    double Qs[EPL];
    if (a.Q45) {
        // cost entry (the seam of cvxpnpl.py:454-460): A^T A (packed 9x9) and B come from the caller
#pragma unroll
        for (int m = 0; m < EPL; ++m) Qs[m] = (w.ok(m) && w.ej(m) < 9) ? a.Q45[b * 45 + cvx::qidx(w.ei(m), w.ej(m))] : 0.0;
#pragma unroll
        for (int m = 0; m < G_::M27; ++m)
            if (gl + LPP * m < 27) L[Q_B + gl + LPP * m] = a.B27[b * 27 + gl + LPP * m];
    } else {
        cvx::ProblemView pv = cvx::make_view(b, a.n_p, a.p2, a.p3, a.n_l, a.l2, a.l3, a.K, a.K_per_problem);
        double Ki[9];
        {
            double Kc[9], det;
#pragma unroll
            for (int i = 0; i < 9; ++i) Kc[i] = pv.K[i];
            cvx::inv3(Kc, Ki, det);
            okK = (det == det) && det != 0.0;
        }
        // Gram sums: role r = gl + 16 m < 60 is  sum rec[6 + qa] rec[6 + qb] rec[te]  with rec = (T[6], 1, P[3]):
        // M0 (6): qa = qb = 0 | M1 (3 x 6): qa = 1 + a | M2 (6 x 6): qa = 1 + a, qb = 1 + b
        constexpr int M60 = G_::M60;
        int rqa[M60], rqb[M60], rte[M60];
        double acc[M60];
#pragma unroll
        for (int m = 0; m < M60; ++m) acc[m] = 0.0;
#pragma unroll
        for (int m = 0; m < M60; ++m) {
            const int r = gl + LPP * m;
            const int al = r < 60 ? r : 0;
            int qa = 0, qb = 0, te = al;
            if (al >= 6 && al < 24) { qa = 1 + (al - 6) / 6; te = (al - 6) % 6; }
            if (al >= 24) {
                const int ab = (al - 24) / 6;
                te = (al - 24) % 6;
                qa = 1 + (ab < 3 ? 0 : (ab < 5 ? 1 : 2));
                qb = 1 + (ab < 3 ? ab : (ab < 5 ? ab - 2 : 2));
            }
            rqa[m] = 6 + qa; rqb[m] = 6 + qb; rte[m] = te;
        }
        const int nrec = pv.n_p + 2 * pv.n_l;
        // sums about a point of the scene (cvx::shift_centre: exact, and well conditioned far from the world origin)
        double cs_[3];
        cvx::shift_centre(pv.n_p, pv.p3, pv.n_l, pv.l3, cs_);
        const double cs0 = cs_[0], cs1 = cs_[1], cs2 = cs_[2];
        for (int base = 0; base < nrec; base += LPP) {
            const int cnt = nrec - base < LPP ? nrec - base : LPP;
            if (gl < cnt) {
                const int r = base + gl;
                double T[6], P[3];
                if (r < pv.n_p) {
                    double p[3];
                    cvx::bearing(Ki, pv.p2[2 * r], pv.p2[2 * r + 1], p);
                    const double n2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
                    T[0] = n2 - p[0] * p[0]; T[1] = -p[0] * p[1]; T[2] = -p[0] * p[2];
                    T[3] = n2 - p[1] * p[1]; T[4] = -p[1] * p[2]; T[5] = n2 - p[2] * p[2];
                    P[0] = pv.p3[3 * r] - cs0; P[1] = pv.p3[3 * r + 1] - cs1; P[2] = pv.p3[3 * r + 2] - cs2;
                } else {
                    const int li = (r - pv.n_p) >> 1, en = (r - pv.n_p) & 1;
                    const double *l2 = pv.l2 + 4 * li, *l3 = pv.l3 + 6 * li + 3 * en;
                    double u[3], v[3];
                    cvx::bearing(Ki, l2[0], l2[1], u);
                    cvx::bearing(Ki, l2[2], l2[3], v);
                    double n[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
                    const double inv = cvx::rsqrt_(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                    n[0] *= inv; n[1] *= inv; n[2] *= inv;
                    T[0] = n[0] * n[0]; T[1] = n[0] * n[1]; T[2] = n[0] * n[2]; T[3] = n[1] * n[1]; T[4] = n[1] * n[2]; T[5] = n[2] * n[2];
                    P[0] = l3[0] - cs0; P[1] = l3[1] - cs1; P[2] = l3[2] - cs2;
                }
                double *rec = L + Q_WF + gl * 10;
#pragma unroll
                for (int i = 0; i < 6; ++i) rec[i] = T[i];
                rec[6] = 1.0; rec[7] = P[0]; rec[8] = P[1]; rec[9] = P[2];
            }
            CVXW_SYNC();
            for (int c = 0; c < cnt; ++c) {
                const double *rec = L + Q_WF + c * 10;
#pragma unroll
                for (int m = 0; m < M60; ++m) acc[m] += rec[rqa[m]] * rec[rqb[m]] * rec[rte[m]];
            }
            CVXW_SYNC();
        }
#pragma unroll
        for (int m = 0; m < M60; ++m)
            if (gl + LPP * m < 60) L[Q_WF + gl + LPP * m] = acc[m];
        CVXW_SYNC();
        // B = M0^-1 [M1_0 M1_1 M1_2], Q = M2 - M1^T B
        {
            const double *mm = L + Q_WF;
            double M0[9] = {mm[0], mm[1], mm[2], mm[1], mm[3], mm[4], mm[2], mm[4], mm[5]}, Mi[9], det;
            cvx::inv3(M0, Mi, det);
            const double sc = mm[0] + mm[3] + mm[5];
            okG = det > 1e-12 * (sc * sc * sc) * (1.0 / 27.0);
            if (gl < 9) {
                double sel = Mi[0];
#pragma unroll
                for (int i = 1; i < 9; ++i) sel = gl == i ? Mi[i] : sel;
                L[Q_X + gl] = sel;
            }
        }
        CVXW_SYNC();
        // packed index of (i, j) in a symmetric 3x3 (00 01 02 11 12 22)
        auto psym = [](int i, int j) { const int lo = i < j ? i : j, hi = i < j ? j : i; return lo * 3 - (lo == 2 ? 1 : 0) + (hi - lo); };
#pragma unroll
        for (int m = 0; m < G_::M27; ++m) {
            const int idx = gl + LPP * m;
            if (idx < 27) {
                const int bb = idx / 9, i = (idx % 9) / 3, j = idx % 3;
                const double *m1 = L + Q_WF + 6 + 6 * bb;
                double v = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) v += L[Q_X + i * 3 + k] * m1[psym(k, j)];
                L[Q_B + i * 9 + 3 * bb + j] = v; // B'[i][3 bb + j] (about the shifted origin)
            }
        }
        CVXW_SYNC();
#pragma unroll
        for (int m = 0; m < EPL; ++m) {
            double v = 0.0;
            if (w.ok(m) && w.ej(m) < 9) {
                const int qa = w.ei(m) / 3, qi = w.ei(m) % 3, qb = w.ej(m) / 3, qj = w.ej(m) % 3;
                const double *m1 = L + Q_WF + 6 + 6 * qa, *m2 = L + Q_WF + 24 + 6 * psym(qa, qb);
                v = m2[psym(qi, qj)];
#pragma unroll
                for (int k = 0; k < 3; ++k) v -= m1[psym(qi, k)] * L[Q_B + k * 9 + 3 * qb + qj];
            }
            Qs[m] = v;
        }
        CVXW_SYNC(); // (every read of B' for Q is done: the shift goes back into B, t = -B' r - R c)
        if (gl < 9) L[Q_B + (gl / 3) * 9 + 3 * (gl % 3) + gl / 3] += (gl % 3 == 0 ? cs0 : (gl % 3 == 1 ? cs1 : cs2));
    }
    CVXW_SYNC();
#pragma unroll
    for (int m = 0; m < EPL; ++m)
        if (w.ok(m)) L[Q_X + w.e(m)] = Qs[m];
    CVXW_SYNC();
    double tr = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) tr += L[Q_X + cvx::sidx(k, k)];
    bool finite = okK && okG && (tr == tr) && tr > 0 && tr < 1e300;
#pragma unroll
    for (int m = 0; m < EPL; ++m) finite = finite && (Qs[m] == Qs[m]);
    finite = grp_bits<LPP>(__ballot(!finite), grp) == 0;
    const double itr = finite ? cvx::rcp(tr) : 0.0;
    // a planar scene in a general world frame goes to the wave-per-problem kernel at once: it solves in the
    // canonical frame (cvx::canonicalise_planar), with the D-even certificate and the twin logic
    bool planar_handoff = false;
    if (finite) {
        double T[9], U[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                T[i * 3 + j] = (L[Q_X + cvx::sidx(3 * i, 3 * j)] + L[Q_X + cvx::sidx(3 * i + 1, 3 * j + 1)] + L[Q_X + cvx::sidx(3 * i + 2, 3 * j + 2)]) * itr;
        planar_handoff = cvx::planar_frame(T, U); // replicated in the row: row-uniform
    }
    CVXW_SYNC();
#pragma unroll
    for (int m = 0; m < EPL; ++m) {
        Qs[m] *= itr;
        if (w.ok(m) && w.ej(m) < 9) { L[Q_QF + w.ei(m) * 10 + w.ej(m)] = Qs[m]; L[Q_QF + w.ej(m) * 10 + w.ei(m)] = Qs[m]; }
    }
    if (gl < 9) L[Q_QF + gl * 10 + 9] = 0.0;
    // planar scene: the cost is blind to the third column of R (cvx::dual_certificate, SYMM)
    bool symm;
    {
        bool zero = true;
#pragma unroll
        for (int m = 0; m < EPL; ++m) zero = zero && !(w.ok(m) && w.ej(m) >= 6 && w.ej(m) < 9 && !(fabs(Qs[m]) < 1e-13));
        symm = grp_bits<LPP>(__ballot(!zero), grp) == 0;
    }
    bool odd[EPL];
#pragma unroll
    for (int m = 0; m < EPL; ++m) odd[m] = symm && ((w.ei(m) < 6) != (w.ej(m) < 6));

#ifdef CVXQ_TIMELINE
    tl_[1] = tl_now();
#endif
#ifdef CVXQ_PHASES // diagnostics (tools/quad_phases.py): shader cycles per phase of this wavefront, summed over its iterations
    unsigned long long ph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t;
    auto ph_now = []() { unsigned long long t; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory"); return t; };
    ph_t = ph_now();
#define CVXQ_PH(k) do { const unsigned long long n_ = ph_now(); ph_[k] += n_ - ph_t; ph_t = n_; } while (0)
#else
#define CVXQ_PH(k)
#endif
    // ---------------------------------------------------------------- ADMM
    double delta = o.eps / (8.0 * tr);
    delta = delta < 1e-13 ? 1e-13 : delta;
    const double gap_tol = o.eps > 8e-13 * tr ? o.eps : 8e-13 * tr;
    double rho = o.rho, irho = 1.0 / o.rho;
    double W[EPL], Wp[EPL];
#pragma unroll
    for (int m = 0; m < EPL; ++m) { W[m] = (w.e(m) == 54) ? 1.0 : 0.0; Wp[m] = W[m]; }
    f2 v[5]; // unit eigenvector owned by this lane (warm start of the next eigen-solve), rows 2i and 2i + 1, single precision
#pragma unroll
    for (int i = 0; i < 5; ++i) { v[i].x = (gl == 2 * i) ? 1.0f : 0.0f; v[i].y = (gl == 2 * i + 1) ? 1.0f : 0.0f; }
    double vd[10]; // (F64SW) the same in float64
#pragma unroll
    for (int i = 0; i < 10; ++i) vd[i] = (gl == i) ? 1.0 : 0.0;
    auto vrow = [&](int i) -> double { if constexpr (F64SW) return vd[i]; else return (double)((i & 1) ? v[i >> 1].y : v[i >> 1].x); };
    int it = 0, total_sweeps = 0, next_check = o.first_check;
    bool have_prev = false;
    int reused = 0; // consecutive checks that took over the previous check's pose (cvx::REUSE_MAX, see cvx::solve_sdp)
    double fprev = 0.0;
    bool done = !gvalid || !finite;
    bool parked = false; // this problem goes to the second phase
    const double tol2 = o.jacobi_tol * o.jacobi_tol;
    CVXW_SYNC();
    if (!done && (planar_handoff || symm)) {
        // A planar scene (general or canonical frame) is two-fold ambiguous: rank 2, nothing for this layout to
        // certify, and the slow ones among them run for hundreds of iterations.  They go to the queue of
        // cvxw::resume_wave_kernel (launched behind this kernel), one wavefront each, scheduled dynamically:
        // finishing them here, four in a row per wavefront, was 1.6x slower on a planar batch.
        double *slot = ws + b * cvxw::RS_FULL;
#pragma unroll
        for (int m = 0; m < EPL; ++m)
            if (w.ok(m)) { slot[cvxw::RS_W + w.e(m)] = W[m]; slot[cvxw::RS_Q + w.e(m)] = w.ej(m) < 9 ? L[Q_QF + w.ei(m) * 10 + w.ej(m)] * tr : 0.0; }
#pragma unroll
        for (int m = 0; m < G_::M27; ++m)
            if (gl + LPP * m < 27) slot[cvxw::RS_B + gl + LPP * m] = L[Q_B + gl + LPP * m];
        if (gl == 0) {
            slot[cvxw::RS_IT] = 0.0;
            slot[cvxw::RS_NC] = 0.0;
            const int q = atomicAdd(qcount, 1);
            qentries[q] = (int32_t)b;
        }
        done = true;
    }

    if (gvalid && !finite) { // degenerate input: NaN pose (cvxpnpl.py:493-498)
        if (gl < 9) a.R[b * 9 + gl] = NAN;
        if (gl < 3) a.t[b * 3 + gl] = NAN;
        if (gl == 0) {
            a.status[b] = cvx::ST_NONFINITE;
            if (a.iters) a.iters[b] = 0;
            if (a.cost) { a.cost[2 * b] = NAN; a.cost[2 * b + 1] = NAN; }
            if (a.work) { a.work[2 * b] = 0; a.work[2 * b + 1] = 0; }
        }
        if (a.Z) {
#pragma unroll
            for (int m = 0; m < EPL; ++m)
                if (w.ok(m)) a.Z[b * 55 + w.e(m)] = NAN;
        }
    }

    int lane_h = lane;
    unsigned atab_h = atab;
    while (__any(!done)) {
#ifndef CVXQ_NO_ROLE_REFRESH
        // Everything derived from the lane index (the row's LDS slice, the problem index and the output addresses that hang on it, the exchange
        // partners) is re-derived HERE from a copy the compiler cannot see through: as loop invariants these values were hoisted in front of the
        // loop, a few dozen registers of them, and spilled (cf. cvxw::solve_pass, CVXW_ROLES).  The names shadow the function scope's.
        asm volatile("" : "+v"(lane_h), "+v"(atab_h));
        const int grp = LPP == 16 ? lane_h >> 4 : (lane_h * 43) >> 9;
        const int gl = lane_h - grp * LPP;
        double *const L = lds_all + grp * SLICE;
        double2 *const L2 = reinterpret_cast<double2 *>(L);
        const int64_t b_raw = (int64_t)blockIdx.x * NPW + grp;
        const int64_t b = (grp < NPW && b_raw < a.batch) ? b_raw : a.batch - 1;
        const unsigned atab = atab_h;
        const int lane_base4 = (grp * LPP) << 2;
        w.gl = gl;
#endif
        w.refresh();
        double al = 0.0, sigma = 0.0;
        if (!(it == 0 && o.first_check > 1)) {
            // ---- eigendecomposition of W: one-sided Jacobi on G = (W + sigma I) V_prev, column gl in this lane
            double fro = 0.0;
#pragma unroll
            for (int m = 0; m < EPL; ++m) fro += w.wgt(m) * W[m] * W[m];
            sigma = 1.5 * cvx::sqrt_fast(grp_sum<LPP>(L, gl, fro)) + 1e-300;
            int sweeps = 0;
            const int sw_cap = o.sweep_schedule ? cvx::sweep_cap(it + 1, false, o.jacobi_sweeps) : o.jacobi_sweeps; // (the wavefront pays the maximum over its problems)
            double g[10];
            float alf = 0.0f; // (single-precision path: squared norm of this lane's column)
            f2 q[5];
            if constexpr (F64SW) {
                // float64 throughout: W to LDS as the full symmetric matrix (row stride 10), g = (W + sigma I) v; 22 v_mov_b32_dpp / 22 ds_bpermute per step
#pragma unroll
                for (int m = 0; m < EPL; ++m)
                    if (w.ok(m)) { L[Q_WF + w.ei(m) * 10 + w.ej(m)] = W[m]; L[Q_WF + w.ej(m) * 10 + w.ei(m)] = W[m]; }
                CVXW_SYNC();
                double qd[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) {
                    const double2 *row = L2 + (Q_WF + i * 10) / 2;
                    double a = 0.0;
#pragma unroll
                    for (int kk = 0; kk < 5; ++kk) { const double2 r = row[kk]; a = fma(r.x, vd[2 * kk], a); a = fma(r.y, vd[2 * kk + 1], a); }
                    qd[i] = fma(sigma, vd[i], a);
                }
                double alq = dot10_f64(qd, qd);
CVXQ_PH(0);
                bool active = !done; // row-uniform
                do {
                    bool coarse = false;
                    jacobi_sweep(qd, alq, atab, gl, lane_base4, active, tol2, coarse);
                    alq = dot10_f64(qd, qd); // exact norms once per sweep (the incremental update drifts)
                    const bool grp_more = grp_bits<LPP>(__ballot(coarse && active), grp) != 0;
                    if (active) ++sweeps;
                    active = active && grp_more && sweeps < sw_cap;
                } while (__any(active));
CVXQ_PH(1); /* jacobi (float64) */
#pragma unroll
                for (int i = 0; i < 10; ++i) g[i] = qd[i];
                al = alq;
                {
                    const double il = gl < 10 ? cvx::rsqrt_(alq) : 0.0;
#pragma unroll
                    for (int i = 0; i < 10; ++i) vd[i] = qd[i] * il;
                }
            } else {
            // g = (W + sigma I) v in single precision like the sweeps that follow (see there): W goes to LDS as floats, rows
            // padded to 12 (three b128 reads per row instead of five), two rows of v per packed FMA
            float *Lf = reinterpret_cast<float *>(L + Q_WF);
#pragma unroll
            for (int m = 0; m < EPL; ++m)
                if (w.ok(m)) { const float wf = (float)W[m]; Lf[w.ei(m) * 12 + w.ej(m)] = wf; Lf[w.ej(m) * 12 + w.ei(m)] = wf; }
            CVXW_SYNC();
            const float sigf = (float)sigma;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const float4 *row = reinterpret_cast<const float4 *>(Lf + i * 12);
                const float4 r0 = row[0], r1 = row[1];
                const float2 r2 = *reinterpret_cast<const float2 *>(Lf + i * 12 + 8);
                f2 a = f2{r0.x, r0.y} * v[0];
                a = __builtin_elementwise_fma(f2{r0.z, r0.w}, v[1], a);
                a = __builtin_elementwise_fma(f2{r1.x, r1.y}, v[2], a);
                a = __builtin_elementwise_fma(f2{r1.z, r1.w}, v[3], a);
                a = __builtin_elementwise_fma(f2{r2.x, r2.y}, v[4], a);
                const float vi = (i & 1) ? v[i >> 1].y : v[i >> 1].x;
                const float gi = fmaf(sigf, vi, a.x + a.y);
                if (i & 1) q[i >> 1].y = gi; else q[i >> 1].x = gi;
            }
            {
                f2 a = q[0] * q[0];
#pragma unroll
                for (int i = 1; i < 5; ++i) a = __builtin_elementwise_fma(q[i], q[i], a);
                alf = a.x + a.y;
            }
CVXQ_PH(0); /* fro, LDS copy of W, g = (W + sigma I) v */
            // The sweeps run in single precision, two rows per packed instruction: the columns only have to become
            // orthogonal to the sweep tolerance (6e-2), the iterate is a dual hint whose certificate is verified in
            // double, and this phase ends after handoff_at <= 16 iterations -- measured on the host build (10 k problems each
            // of PnP N=10 / N=6 sigma 5 / N=4, PnPL 5+5): iteration histograms identical to the double sweeps as long as
            // the first 7-16 iterations are concerned (single precision throughout only hurts tails of > 100 iterations,
            // which are the wave-per-problem kernel's).  Half the exchange (11 ds_bpermute per X step, none per A step), half the arithmetic.
            const float tol2f = (float)tol2;
            bool active = !done; // row-uniform
            do {
                bool coarse = false;
                jacobi_sweep(q, alf, atab, gl, lane_base4, active, tol2f, coarse);
                { // exact norms once per sweep (the incremental update drifts)
                    f2 acc = q[0] * q[0];
#pragma unroll
                    for (int i = 1; i < 5; ++i) acc = __builtin_elementwise_fma(q[i], q[i], acc);
                    alf = acc.x + acc.y;
                }
                const bool grp_more = grp_bits<LPP>(__ballot(coarse && active), grp) != 0;
                if (active) ++sweeps;
                active = active && grp_more && sweeps < sw_cap;
            } while (__any(active));
#pragma unroll
            for (int i = 0; i < 5; ++i) { g[2 * i] = (double)q[i].x; g[2 * i + 1] = (double)q[i].y; }
            al = 0.0;
#pragma unroll
            for (int i = 0; i < 10; ++i) al += g[i] * g[i];
CVXQ_PH(1); /* jacobi */
            }
            total_sweeps += sweeps;
            // ---- Wp = sum_{lam > 0} lam u u^T from (g, w g), w = lam / |g|^2
            const double lp = cvx::sqrt_fast(al), lam = lp - sigma;
            const bool col = gl < 10;
            const double wpos = (col && lam > 0) ? lam * cvx::rcp(al) : 0.0;
            if (col) {
#pragma unroll
                for (int i = 0; i < 5; ++i) L2[(Q_Y + gl * 10) / 2 + i] = make_double2(g[2 * i], g[2 * i + 1]);
                L[Q_Y + 100 + gl] = wpos;
            }
            if constexpr (!F64SW) {
                const float ilf = col ? __builtin_amdgcn_rsqf(alf) : 0.0f;
                const f2 il2 = {ilf, ilf};
#pragma unroll
                for (int i = 0; i < 5; ++i) v[i] = q[i] * il2;
            }
            const unsigned long long pm = __ballot(wpos > 0);
            unsigned long long pu = pm;
#pragma unroll
            for (int g = 1; g < NPW; ++g) pu |= pm >> (LPP * g);
            const unsigned anypos = (unsigned)(pu & 0x3FFull);
            CVXW_SYNC();
#pragma unroll
            for (int m = 0; m < EPL; ++m) Wp[m] = 0.0;
#pragma unroll
            for (int s = 0; s < 10; ++s) {
                if ((anypos >> s) & 1u) { // wave-uniform; a column with no weight in THIS problem adds 0
                    const double ws_ = L[Q_Y + 100 + s];
#pragma unroll
                    for (int m = 0; m < EPL; ++m) Wp[m] += ws_ * L[Q_Y + s * 10 + w.ei(m)] * L[Q_Y + s * 10 + w.ej(m)];
                }
            }
        }
CVXQ_PH(2); /* Wp */
        ++it;
        const bool check = MODE != 1 && it >= next_check;
        if (check) {
            // ---- certificate attempt (cvx::solve_sdp, non-twin branch): top eigenvector of Wp
            const double best = grp_max<LPP>(L, gl, gl < 10 ? al : -1.0);
            const unsigned tm = grp_bits<LPP>(__ballot(gl < 10 && al == best), grp);
            const int jmax = __builtin_ctz(tm | 0x10000u);
            CVXW_SYNC(); // (the Wp gathers above are done with Q_Y)
            if (gl == jmax) {
#pragma unroll
                for (int i = 0; i < 10; ++i) L[Q_M + 22 + i] = vrow(i);
            }
            CVXW_SYNC();
            double vloc[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) vloc[i] = L[Q_M + 22 + i];
            double Rc[9], pobj = 0.0, d0;
            bool reuse;
            {   // does the candidate round to the pose the previous check already polished?  (cvx::rounds_to)
                double Rp[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) Rp[i] = L[Q_M + 12 + i];
                reuse = have_prev && reused < cvx::REUSE_MAX && cvx::rounds_to(vloc, Rp, d0);
                reused = reuse ? reused + 1 : 0;
            }
            // the four problems polish together; when none needs it (done, or the rounded candidate is the pose
            // the previous check already polished) the Newton iterations are skipped altogether
CVXQ_PH(3); /* check: top eigenvector, reuse test */
            if (__any(!done && !reuse)) {
                // (problems whose candidate would be reused polish along: same result, no divergence)
                d0 = cvxw::coop_round(vloc, Rc);
                // ---- Newton on SO(3) for f(R) = r^T Qs r (cvx::so3_newton)
                int xsrc[G_::M40];
                double xsgn[G_::M40];
#pragma unroll
                for (int m = 0; m < G_::M40; ++m) {
                    const int idx = gl + LPP * m < 40 ? gl + LPP * m : 0;
                    xsrc[m] = cvxw::kXTab.src[idx];
                    xsgn[m] = (double)cvxw::kXTab.sgn[idx];
                }
                for (int nit = 0; nit < 6; ++nit) {
                    if (gl == 0) {
#pragma unroll
                        for (int i = 0; i < 9; ++i) L[C_RL + i] = Rc[i];
                    }
                    CVXW_SYNC();
#pragma unroll
                    for (int m = 0; m < G_::M40; ++m) {
                        const int idx = gl + LPP * m;
                        if (idx < 40) L[C_XV + idx] = (idx == 9) ? 1.0 : xsgn[m] * L[C_RL + xsrc[m]];
                    }
                    CVXW_SYNC();
#pragma unroll
                    for (int m = 0; m < G_::M36; ++m) {
                        const int idx = gl + LPP * m;
                        if (idx < 36) {
                            const int vv = idx / 9, i = idx % 9;
                            L[C_YV + vv * 10 + i] = cvxw::dot10(L2 + (Q_QF + i * 10) / 2, L2 + (C_XV + vv * 10) / 2);
                        }
                    }
                    if (gl < 4) L[C_YV + gl * 10 + 9] = 0.0;
                    CVXW_SYNC();
                    if (gl < 9) {
                        const int k = gl / 3, l = gl % 3;
                        L[C_H1 + gl] = cvxw::dot10(L2 + (C_XV + (1 + k) * 10) / 2, L2 + (C_YV + (1 + l) * 10) / 2);
                    }
                    CVXW_SYNC();
                    double Qr[9], H1[9];
#pragma unroll
                    for (int i = 0; i < 9; ++i) { Qr[i] = L[C_YV + i]; H1[i] = L[C_H1 + i]; }
                    double N[9];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) N[i * 3 + j] = Rc[0 * 3 + i] * Qr[3 * j] + Rc[1 * 3 + i] * Qr[3 * j + 1] + Rc[2 * 3 + i] * Qr[3 * j + 2];
                    const double g3[3] = {2 * (N[7] - N[5]), 2 * (N[2] - N[6]), 2 * (N[3] - N[1])};
                    const double gn = fabs(g3[0]) + fabs(g3[1]) + fabs(g3[2]);
                    CVXW_SYNC();
                    if (nit >= 2 && !__any(!done && !(gn < 1e-15))) break;
                    const bool final_step = gn < 1e-8;
                    const double trN = N[0] + N[4] + N[8];
                    double H[9];
#pragma unroll
                    for (int k = 0; k < 3; ++k)
#pragma unroll
                        for (int l = 0; l < 3; ++l) H[k * 3 + l] = 2 * H1[k * 3 + l] + N[l * 3 + k] + N[k * 3 + l] - (k == l ? 2 * trN : 0.0);
                    double Hi[9], det;
                    cvx::inv3(H, Hi, det);
                    const bool pd = H[0] > 0 && (H[0] * H[4] - H[1] * H[3]) > 0 && det > 0;
                    const double hn = fabs(H[0]) + fabs(H[4]) + fabs(H[8]) + 1e-300;
                    double ww[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const double nw = -(Hi[k * 3] * g3[0] + Hi[k * 3 + 1] * g3[1] + Hi[k * 3 + 2] * g3[2]);
                        ww[k] = pd ? nw : -g3[k] * cvx::rcp(hn);
                    }
                    const double wn = cvx::sqrt_fast(ww[0] * ww[0] + ww[1] * ww[1] + ww[2] * ww[2]);
                    const double lim = wn > 0.5 ? 0.5 * cvx::rcp(wn) : 1.0;
                    const double q0 = 0.5 * lim * ww[0], q1 = 0.5 * lim * ww[1], q2 = 0.5 * lim * ww[2];
                    const double ss = q0 * q0 + q1 * q1 + q2 * q2;
                    const double f = 2.0 * cvx::rcp(1.0 + ss);
                    const double Cm[9] = {1 + f * (q0 * q0 - ss), f * (-q2 + q0 * q1), f * (q1 + q0 * q2),
                                          f * (q2 + q0 * q1), 1 + f * (q1 * q1 - ss), f * (-q0 + q1 * q2),
                                          f * (-q1 + q0 * q2), f * (q0 + q1 * q2), 1 + f * (q2 * q2 - ss)};
                    double Rn[9];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = Rc[i * 3] * Cm[j] + Rc[i * 3 + 1] * Cm[3 + j] + Rc[i * 3 + 2] * Cm[6 + j];
#pragma unroll
                    for (int i = 0; i < 9; ++i) Rc[i] = Rn[i];
                    if (!__any(!done && !(final_step && pd))) break;
                }
                { // one polar step squares any drift from orthogonality
                    double Ri[9], det, Rn[9];
                    cvx::inv3(Rc, Ri, det);
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = 0.5 * (Rc[i * 3 + j] + Ri[j * 3 + i]);
#pragma unroll
                    for (int i = 0; i < 9; ++i) Rc[i] = Rn[i];
                }
                // z of the final R; pobj = z^T Qs z
                if (gl == 0) {
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) L[C_XV + 3 * j + i] = Rc[i * 3 + j];
                    L[C_XV + 9] = 1.0;
                }
                CVXW_SYNC();
                double part = 0.0;
                if (gl < 9) part = L[C_XV + gl] * cvxw::dot10(L2 + (Q_QF + gl * 10) / 2, L2 + C_XV / 2);
                pobj = grp_sum<LPP>(L, gl, part);
            } else {
#pragma unroll
                for (int i = 0; i < 9; ++i) Rc[i] = L[Q_M + 12 + i];
                pobj = fprev;
                CVXW_SYNC();
                if (gl == 0) {
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) L[C_XV + 3 * j + i] = Rc[i * 3 + j];
                    L[C_XV + 9] = 1.0;
                }
                CVXW_SYNC();
            }
CVXQ_PH(4); /* polar + Newton polish */
            // ---- dual half (cvx::dual_certificate): hint S_h = rho (Wp - W); S1 = S_h - P_null(S_h - Qs)
            double S[EPL];
            {
                double T[EPL];
#pragma unroll
                for (int m = 0; m < EPL; ++m) { S[m] = rho * (Wp[m] - W[m]); T[m] = S[m] - (w.ej(m) < 9 ? L[Q_QF + w.ei(m) * 10 + w.ej(m)] : 0.0); }
                quad_proj<VAR>(L, w, T, 0.0);
#pragma unroll
                for (int m = 0; m < EPL; ++m) {
                    S[m] = odd[m] ? 0.0 : S[m] - T[m];
                    if (w.ok(m)) { L[Q_WF + w.ei(m) * 10 + w.ej(m)] = S[m]; L[Q_WF + w.ej(m) * 10 + w.ei(m)] = S[m]; }
                }
            }
            CVXW_SYNC();
            if (gl < 10) L[C_ROW + gl] = cvxw::dot10(L2 + (Q_WF + gl * 10) / 2, L2 + C_XV / 2); // rhs = S z
            CVXW_SYNC();
            {
                double rhs[10], lamv[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) rhs[i] = L[C_ROW + i];
                cvx::dual_lambda<VAR>(Rc, rhs, symm, lamv);
                if (gl == 0) {
#pragma unroll
                    for (int i = 0; i < 10; ++i) L[C_LAM + i] = lamv[i];
                }
            }
            CVXW_SYNC();
            {   // S2 = S1 - P_range(sym(lam z^T))
                double E[EPL], Nn[EPL];
#pragma unroll
                for (int m = 0; m < EPL; ++m) {
                    E[m] = odd[m] ? 0.0 : 0.5 * (L[C_LAM + w.ei(m)] * L[C_XV + w.ej(m)] + L[C_XV + w.ei(m)] * L[C_LAM + w.ej(m)]);
                    Nn[m] = E[m];
                }
                quad_proj<VAR>(L, w, Nn, 0.0);
#pragma unroll
                for (int m = 0; m < EPL; ++m) {
                    S[m] -= E[m] - Nn[m];
                    if (w.ok(m)) { L[Q_WF + w.ei(m) * 10 + w.ej(m)] = S[m]; L[Q_WF + w.ej(m) * 10 + w.ei(m)] = S[m]; }
                }
            }
            CVXW_SYNC();
            double res, zSz;
            {
                const int aa = gl < 10 ? gl : 0;
                const double sz = cvxw::dot10(L2 + (Q_WF + aa * 10) / 2, L2 + C_XV / 2);
                res = grp_max<LPP>(L, gl, gl < 10 ? fabs(sz) : 0.0);
                zSz = grp_sum<LPP>(L, gl, gl < 10 ? L[C_XV + aa] * sz : 0.0);
            }
            CVXW_SYNC();
            double Se[EPL];
#pragma unroll
            for (int m = 0; m < EPL; ++m) Se[m] = S[m] + (((w.pk[m] >> 23) & 1) ? delta : 0.0);
CVXQ_PH(5); /* dual fit + correction */
            const double minp = quad_ldl(L, w, Se);
            const bool pre = (res < 1e-10) && (d0 > 0) && (pobj == pobj);
            // (The eigen-gradient step of the dual, cvx::dual_refine_step, is NOT made here -- measured, round 6, profiles/r06/refine_ab3.txt: with the
            //  step in this phase the four problems of a wavefront pay for it in lockstep whenever one of them fails, and the wavefronts that end
            //  a launch pay it twice before the hand-over: judged launch 50.9 M poses/s without the code, 48.4 with it compiled in and switched
            //  off (register allocation), 46.6 with it on; 16 k 66.3 / 62.7 / 60.8.  The wave-per-problem phase behind the hand-over makes it.)
            const bool cok = (minp > 0) && pre;
            const bool gap_ok = cok && (tr * (fabs(zSz) + 4.0 * delta) <= gap_tol);
            have_prev = d0 > 0 && (pobj == pobj);
            fprev = pobj;
            if (gl == 0) {
#pragma unroll
                for (int i = 0; i < 9; ++i) { L[Q_M + 12 + i] = Rc[i]; L[Q_M + i] = Rc[i]; }
            }
            CVXW_SYNC();
            next_check = cvx::next_check_after(it, o);
            if (gap_ok && !done) {
                // ---- outputs of a certified problem
                if (gl < 9) a.R[b * 9 + gl] = L[Q_M + gl];
                if (gl < 3) { // t = -B r (cvxpnpl.py:513), r = vec_colmajor(R)
                    double tv = 0;
#pragma unroll
                    for (int c3 = 0; c3 < 3; ++c3)
#pragma unroll
                        for (int r3 = 0; r3 < 3; ++r3) tv += L[Q_B + gl * 9 + 3 * c3 + r3] * L[Q_M + r3 * 3 + c3];
                    a.t[b * 3 + gl] = -tv;
                }
                if (gl == 0) {
                    a.status[b] = cvx::ST_CERTIFIED;
                    if (a.iters) a.iters[b] = it;
                    if (a.cost) { a.cost[2 * b] = tr * pobj; a.cost[2 * b + 1] = tr * (pobj - zSz - 4.0 * delta); }
                    if (a.work) { a.work[2 * b] = 1; a.work[2 * b + 1] = total_sweeps; }
                }
                if (a.Z) { // Z = z z^T with z = [vec_colmajor(R); 1]
#pragma unroll
                    for (int m = 0; m < EPL; ++m)
                        if (w.ok(m)) a.Z[b * 55 + w.e(m)] = L[C_XV + w.ei(m)] * L[C_XV + w.ej(m)];
                }
                done = true;
            }
        }
CVXQ_PH(6); /* LDL + outputs (or nothing when no check) */
        if (!done && it == o.tail_from) { // smaller penalty for the slow tail; keeps the dual: Wm scales by rho / rho_tail
#pragma unroll
            for (int m = 0; m < EPL; ++m) W[m] = Wp[m] + (W[m] - Wp[m]) * (rho / o.rho_tail);
        }
        if (it == o.tail_from) { rho = o.rho_tail; irho = 1.0 / rho; }
        // ---- X = Pi_aff(2 Wp - W - Qs / rho);  W <- W + alpha (X - Wp)
        {
            double X[EPL];
#pragma unroll
            for (int m = 0; m < EPL; ++m) X[m] = 2.0 * Wp[m] - W[m] - irho * (w.ej(m) < 9 ? L[Q_QF + w.ei(m) * 10 + w.ej(m)] : 0.0); // (the cost entries stay in LDS: 8 registers less to carry through the loop)
            quad_proj<VAR>(L, w, X, 1.0);
            double r2 = 0.0;
#pragma unroll
            for (int m = 0; m < EPL; ++m) {
                const double dd = X[m] - Wp[m];
                W[m] += o.alpha * dd;
                r2 += w.wgt(m) * dd * dd;
            }
            const double fp_res = cvx::sqrt_fast(grp_sum<LPP>(L, gl, r2));
            if (!done && !(fp_res == fp_res)) { // NaN guard
                if (gl < 9) a.R[b * 9 + gl] = NAN;
                if (gl < 3) a.t[b * 3 + gl] = NAN;
                if (gl == 0) {
                    a.status[b] = cvx::ST_NONFINITE;
                    if (a.iters) a.iters[b] = it;
                    if (a.cost) { a.cost[2 * b] = NAN; a.cost[2 * b + 1] = NAN; }
                    if (a.work) { a.work[2 * b] = 0; a.work[2 * b + 1] = total_sweeps; }
                }
                if (a.Z) {
#pragma unroll
                    for (int m = 0; m < EPL; ++m)
                        if (w.ok(m)) a.Z[b * 55 + w.e(m)] = NAN;
                }
                done = true;
            }
        }
CVXQ_PH(7); /* projection + update */
        if (it >= handoff_at) {
            // ---- hand the unfinished problems to the wave-per-problem kernel
            if (!done) {
                // the iterate, and what this wavefront has that the next one would otherwise rebuild: cost, B, eigenvectors
                double *slot = ws + b * cvxw::RS_FULL;
#pragma unroll
                for (int m = 0; m < EPL; ++m)
                    if (w.ok(m)) {
                        park(slot + cvxw::RS_W + w.e(m), W[m]);
                        park(slot + cvxw::RS_Q + w.e(m), w.ej(m) < 9 ? L[Q_QF + w.ei(m) * 10 + w.ej(m)] * tr : 0.0);
                    }
#pragma unroll
                for (int m = 0; m < G_::M27; ++m)
                    if (gl + LPP * m < 27) park(slot + cvxw::RS_B + gl + LPP * m, L[Q_B + gl + LPP * m]);
                if (gl < 10) {
#pragma unroll
                    for (int i = 0; i < 10; ++i) park(slot + cvxw::RS_V + gl * 10 + i, vrow(i));
                }
                if (gl == 0) { park(slot + cvxw::RS_IT, (double)it); park(slot + cvxw::RS_NC, (double)next_check); }
                parked = true;
            }
            break;
        }
    }

#ifdef CVXQ_TIMELINE
    tl_[2] = tl_now();
#endif
    // ---------------------------------------------------------------- second phase (wave per problem)
    const unsigned long long pm = __ballot(parked);
    unsigned pmask = 0;
#pragma unroll
    for (int g = 0; g < NPW; ++g) pmask |= (unsigned)((pm >> (LPP * g)) & 1ull) << g;
    if (MODE == 1 || MODE == 2) { // every survivor (MODE 1: every problem -- it makes no attempts) goes to the queue of the resume kernel launched behind this one: no wave-per-problem code in this kernel
        if (parked && gvalid && gl == 0) {
            const int q = atomicAdd(qcount, 1);
            qentries[q] = (int32_t)b;
        }
        return;
    }
    if (MODE == 3 && (pmask & (pmask - 1))) { // (experiment, layout 13) more than one survivor: this wavefront keeps the first, the others are
        // queued for the resume kernel behind this launch instead of being finished one after the other here
        const unsigned keep = pmask & (0u - pmask);
        const int g = grp < NPW ? grp : 0;
        if (parked && gvalid && gl == 0 && !((keep >> g) & 1u)) {
            const int q = atomicAdd(qcount, 1);
            qentries[q] = (int32_t)b;
        }
        pmask = keep;
    }
    if (pmask) { // wave-uniform
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); // every park() acknowledged before the iterate is read back
        CVXW_SYNC();
        finish_own<NPW, VAR>((QuadArgsPtr)__builtin_amdgcn_kernarg_segment_ptr(), pmask, lds_all);
    }
#ifdef CVXQ_PHASES
    if (lane == 0 && a.cost && (int64_t)blockIdx.x * NPW + 3 < a.batch) {
        double *c = a.cost + 2 * ((int64_t)blockIdx.x * NPW);
#pragma unroll
        for (int k = 0; k < 8; ++k) c[k] = (double)ph_[k];
        if (a.work) { a.work[2 * ((int64_t)blockIdx.x * NPW)] = it; a.work[2 * ((int64_t)blockIdx.x * NPW) + 1] = total_sweeps; }
    }
#endif
#ifdef CVXQ_TIMELINE
    tl_[3] = tl_now();
    if (lane == 0 && a.cost && (int64_t)blockIdx.x * NPW + 1 < a.batch) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        double *c = a.cost + 2 * ((int64_t)blockIdx.x * NPW);
        c[0] = (double)tl_[0]; c[1] = (double)tl_[1]; c[2] = (double)tl_[2]; c[3] = (double)tl_[3];
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        if (a.work) { int32_t *wk = a.work + 2 * ((int64_t)blockIdx.x * NPW); wk[0] = (int)hw; wk[1] = it; wk[2] = (int)(xcc & 15); }
    }
#endif
}

} // namespace cvxq
