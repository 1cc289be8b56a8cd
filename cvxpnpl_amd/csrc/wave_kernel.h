// wave_kernel.h -- wave-per-problem layout: one gfx950 wavefront owns one SDP.
//
// The 64 lanes of a wave cooperate on ONE problem; all state lives in registers and in
// a private 7 KB LDS slice (no inter-wave communication, no __syncthreads):
//   * "entry lanes"  e = 0..54   own entry (i <= j) of the symmetric 10x10 iterate W
//                                (vech order of cvxpnpl.py:346-370) -- affine projection,
//                                ADMM update, PSD reconstruction are one entry per lane;
//   * "jacobi lanes" 10 k + i    own row i of the two columns paired at position k of a
//                                round-robin (circle) ordering -- the one-sided Jacobi
//                                rotation of pair k is local to its 10 lanes, the dot
//                                product g_p . g_q is a 10-lane LDS reduction, and the
//                                pairing advances by shifting one value per lane to the
//                                neighbouring group (lane +-10) through LDS;
//   * "accumulator lanes" 0..59  own one of the 60 Gram accumulators of the assembly.
// The certificate is cooperative as well (coop_round / coop_polish / coop_dual): 9x9 matrix-vector
// products one row per lane, dual fit and correction one entry per lane, the 3x3 algebra and the
// closed-form multipliers replicated in every lane, one LDL^T with pivot rows broadcast through LDS.
// Control flow (sweeps, iterations, exit) is wave-uniform: no divergence, per-problem
// early exit frees the SIMD slot for the next wave.
#pragma once
#include <hip/hip_runtime.h>

#include "problem_io.h"
#include "solver_core.h"

#ifndef CVX_DUAL_REFINE_COMPILED
#define CVX_DUAL_REFINE_COMPILED 1 // (diagnostic builds: 0 compiles the eigen-gradient step of the dual out of the kernels)
#endif
namespace cvxw {

constexpr int WPB = 1; // waves (= problems) per block: 1, so a finished problem frees its SIMD slot at once

// LDS slice per wave, in doubles (all offsets even => 16-byte aligned)
constexpr int L_P = 0;      // 64   per-lane products for 10-lane reductions
constexpr int L_EX = 64;    // 256  exchange buffer: lane -> (a, alpha), (b, beta)
constexpr int L_Y = 320;    // 200  (g, w g) per [slot][row]
constexpr int L_X = 520;    // 64   entry scratch (affine projection gathers)
constexpr int L_G = 584;    // 100  full 10x10
constexpr int L_B = 684;    // 28   translation map B (27)
constexpr int L_M = 712;    // 64   misc: 0.. slot norms, 16.. R, 25.. cost, dobj, status, rank, 30../40.. twin R's, 50.. previous R
constexpr int L_V = 776;    // 20   candidate eigenvectors (top, runner-up)
constexpr int L_VN = 796;   // 100  unit eigenvectors of the previous iterate [position][row] (warm start)
constexpr int L_U = 896;    // 10   planar scene in a general frame: the rotation U to the canonical frame (row-major)
constexpr int LDSW = 908;
constexpr int LDSW_IPM = 2304; // with the interior-point solve (ipm_wave.h uses [908, 2304) on top): cvxw::rescue_wave_kernel

struct LaneTab {
    signed char ei[64], ej[64], p1[64], p2[64], s0[64], s1[64], s2[64], diag[64];
};

constexpr LaneTab make_lane_tab()
{
    LaneTab t{};
    int e = 0;
    for (int i = 0; i < 10; ++i)
        for (int j = i; j < 10; ++j) { t.ei[e] = (signed char)i; t.ej[e] = (signed char)j; t.diag[e] = (i == j); ++e; }
    for (int l = 55; l < 64; ++l) { t.ei[l] = t.ei[l - 55]; t.ej[l] = t.ej[l - 55]; t.diag[l] = t.diag[l - 55]; }
    for (int tr = 0; tr < 15; ++tr)
        for (int k = 0; k < 3; ++k) {
            int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
            int ee = cvx::sidx(cvx::tri_i(tr, k), cvx::tri_j(tr, k));
            t.p1[ee] = (signed char)cvx::sidx(cvx::tri_i(tr, k1), cvx::tri_j(tr, k1));
            t.p2[ee] = (signed char)cvx::sidx(cvx::tri_i(tr, k2), cvx::tri_j(tr, k2));
            t.s0[ee] = (signed char)cvx::tri_s(tr, k);
            t.s1[ee] = (signed char)cvx::tri_s(tr, k1);
            t.s2[ee] = (signed char)cvx::tri_s(tr, k2);
        }
    for (int l = 55; l < 64; ++l) { t.p1[l] = t.p1[l - 55]; t.p2[l] = t.p2[l - 55]; t.s0[l] = t.s0[l - 55]; t.s1[l] = t.s1[l - 55]; t.s2[l] = t.s2[l - 55]; }
    return t;
}


#define CVXW_SYNC()                                              \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

// Reductions over the wavefront without the LDS crossbar: a 4-step DPP butterfly inside each row of 16
// lanes (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror: every lane then holds its row's
// result), then the four rows are combined through v_readlane.  ~25 VALU instructions and no wait on
// ds_bpermute round trips (the __shfl_xor version: 12 of them per reduction).
template <int CTRL>
__device__ __forceinline__ double wave_dpp(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_lane(double x, int lane)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), lane), __builtin_amdgcn_readlane(__double2loint(x), lane));
}
__device__ __forceinline__ double wave_sum(double x)
{
    x += wave_dpp<0xB1>(x);
    x += wave_dpp<0x4E>(x);
    x += wave_dpp<0x141>(x);
    x += wave_dpp<0x140>(x);
    return (wave_lane(x, 0) + wave_lane(x, 16)) + (wave_lane(x, 32) + wave_lane(x, 48));
}
__device__ __forceinline__ double wave_max(double x)
{
    x = fmax(x, wave_dpp<0xB1>(x));
    x = fmax(x, wave_dpp<0x4E>(x));
    x = fmax(x, wave_dpp<0x141>(x));
    x = fmax(x, wave_dpp<0x140>(x));
    return fmax(fmax(wave_lane(x, 0), wave_lane(x, 16)), fmax(wave_lane(x, 32), wave_lane(x, 48)));
}

} // namespace cvxw
#include "ipm_wave.h" // cvxw::coop_ipm (needs CVXW_SYNC, wave_sum and LDSW from above)
namespace cvxw {

// ---------------------------------------------------------------------------------------
// cooperative certificate: constant tables

// element i (0..9) of x-vector v (0: z = [vec(R); 1]; 1..3: [vec(R [e_k]x); 0]) as sign * R[src]
struct XTab { signed char src[40]; signed char sgn[40]; };
constexpr XTab make_xtab()
{
    XTab t{};
    for (int v = 0; v < 4; ++v)
        for (int i = 0; i < 10; ++i) {
            int src = 0, sg = 0;
            if (i < 9) {
                const int row = i % 3, c = i / 3;
                if (v == 0) { src = row * 3 + c; sg = 1; }
                else {
                    const int k = v - 1; // (R [e_k]x)[row][c] : column c of [e_k]x
                    if (k == 0) { if (c == 1) { src = row * 3 + 2; sg = 1; } if (c == 2) { src = row * 3 + 1; sg = -1; } }
                    if (k == 1) { if (c == 0) { src = row * 3 + 2; sg = -1; } if (c == 2) { src = row * 3 + 0; sg = 1; } }
                    if (k == 2) { if (c == 0) { src = row * 3 + 1; sg = 1; } if (c == 1) { src = row * 3 + 0; sg = -1; } }
                }
            }
            t.src[v * 10 + i] = (signed char)src;
            t.sgn[v * 10 + i] = (signed char)sg;
        }
    return t;
}
__device__ const XTab kXTab = make_xtab();

// everything a lane needs to know about its roles in ONE 32-bit word (one global load at kernel start instead
// of ten byte loads scattered over the phases): ei | ej << 4 | p1 << 8 | p2 << 14 | (s0 < 0) << 20 | (s1 < 0) << 21 |
// (s2 < 0) << 22 | diag << 23 | xsrc << 24 | (xsgn: 0 zero, 1 plus, 2 minus) << 28
struct LanePack { unsigned w[64]; };
constexpr LanePack make_lane_pack()
{
    const LaneTab t = make_lane_tab();
    const XTab x = make_xtab();
    LanePack o{};
    for (int l = 0; l < 64; ++l) {
        const int xs = l < 40 ? x.src[l] : 0, xg = l < 40 ? x.sgn[l] : 0;
        o.w[l] = (unsigned)t.ei[l] | ((unsigned)t.ej[l] << 4) | ((unsigned)t.p1[l] << 8) | ((unsigned)t.p2[l] << 14) |
                 ((unsigned)(t.s0[l] < 0) << 20) | ((unsigned)(t.s1[l] < 0) << 21) | ((unsigned)(t.s2[l] < 0) << 22) |
                 ((unsigned)(t.diag[l] != 0) << 23) | ((unsigned)xs << 24) | ((unsigned)(xg == 0 ? 0 : (xg > 0 ? 1 : 2)) << 28);
    }
    return o;
}
__device__ const LanePack kLanePack = make_lane_pack();

__device__ __forceinline__ double fast_rcp(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    return fma(r, e, r);
}

__device__ __forceinline__ double dot10(const double2 *a, const double2 *b)
{
    const double2 a0 = a[0], a1 = a[1], a2 = a[2], a3 = a[3], a4 = a[4];
    const double2 b0 = b[0], b1 = b[1], b2 = b[2], b3 = b[3], b4 = b[4];
    return ((a0.x * b0.x + a0.y * b0.y) + (a1.x * b1.x + a1.y * b1.y)) + ((a2.x * b2.x + a2.y * b2.y) + (a3.x * b3.x + a3.y * b3.y)) +
           (a4.x * b4.x + a4.y * b4.y);
}

// Optional phase profile (-DCVXW_PROFILE, tools/phase_profile.py): shader-clock cycles of every wave,
// accumulated per phase in SGPRs and added to g_phase_cycles at the end.  Not part of the product build.
#ifdef CVXW_PROFILE
__device__ unsigned long long g_phase_cycles[32];
__device__ __forceinline__ unsigned long long cvxw_clock()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
// ph_[0..22]: cycles per phase / event counts; ph_[23]: time of the last marker
#define CVXW_PH_DECL unsigned long long ph_[24]; { _Pragma("unroll") for (int k_ = 0; k_ < 24; ++k_) ph_[k_] = 0; ph_[23] = cvxw_clock(); }
#define CVXW_PH_PARAM , unsigned long long (&ph_)[24]
#define CVXW_PH_ARG , ph_
#define CVXW_PHX(A, P) do { const unsigned long long n_ = cvxw_clock(); (A)[P] += n_ - (A)[23]; (A)[23] = n_; } while (0)
#define CVXW_PH(P) CVXW_PHX(ph_, P)
#define CVXW_PHR(P) CVXW_PHX(ph_, P)
#define CVXW_CNT(P) do { ph_[P] += 1; } while (0)
#define CVXW_PH_FLUSH() do { if ((threadIdx.x & 63) == 0) { _Pragma("unroll") for (int k_ = 0; k_ < 23; ++k_) atomicAdd(&g_phase_cycles[k_], ph_[k_]); atomicAdd(&g_phase_cycles[31], 1ull); } } while (0)
#else
#define CVXW_PH_DECL
#define CVXW_PH_PARAM
#define CVXW_PH_ARG
#define CVXW_PH(P)
#define CVXW_PHR(P)
#define CVXW_CNT(P)
#define CVXW_PH_FLUSH()
#endif
enum { PH_ASSEMBLE = 0, PH_EIG_SETUP, PH_JACOBI, PH_WP, PH_TOPSEL, PH_POLISH, PH_DUAL, PH_CHECK_TAIL, PH_UPDATE, PH_OUTPUT,
       PH_P_POLAR = 10, PH_P_NEWTON, PH_P_FINAL, PH_D_HINT, PH_D_MBUILD, PH_D_LDL1, PH_D_BACKSUB, PH_D_RANGE, PH_D_LDL2,
       CNT_NEWTON = 20, CNT_POLISH, CNT_DUAL };

struct Roles {
    int lane, el, ei, ej, p1, p2, xsrc;
    bool is_diag;
    double s0, s1, s2, xsgn;
};

// projection of the symmetric matrix held one entry per lane onto { <A_i, Z> = b_i }
// (tgt = 1) or its direction space (tgt = 0); same closed form as cvx::proj_affine.
// VAR_RC (the reference's "rc" ablation): the row-orthonormality rows are absent -- entries (i, j) inside one
// diagonal 3x3 block (i / 3 == j / 3, i != j: triples 0..2) are free, diagonal entries only see their column sum.
template <int VAR = cvx::VAR_FULL>
__device__ __forceinline__ double coop_proj(double *L, const Roles &r, double X, double tgt)
{
    L[L_X + r.el] = X;
    CVXW_SYNC();
    double d[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) d[k] = L[L_X + cvx::sidx(k, k)];
    const double r0 = d[0] + d[3] + d[6] - tgt, r1 = d[1] + d[4] + d[7] - tgt, r2 = d[2] + d[5] + d[8] - tgt;
    const double c0 = d[0] + d[1] + d[2] - tgt, c1 = d[3] + d[4] + d[5] - tgt, c2 = d[6] + d[7] + d[8] - tgt;
    const double tot = r0 + r1 + r2;
    const int ri = r.ei % 3, ci = r.ei / 3; // diagonal entry (ei, ei), ei < 9, is D[ri][ci]
    const double rr = ri == 0 ? r0 : (ri == 1 ? r1 : r2), cc = ci == 0 ? c0 : (ci == 1 ? c1 : c2);
    const double corr = VAR == cvx::VAR_RC ? cc * (1.0 / 3.0) : (rr + cc) * (1.0 / 3.0) - tot * (1.0 / 9.0);
    const double xdiag = (r.ei == 9) ? tgt : X - corr;
    const double m = (r.s0 * X + r.s1 * L[L_X + r.p1] + r.s2 * L[L_X + r.p2]) * (1.0 / 3.0);
    CVXW_SYNC();
    const bool free_entry = VAR == cvx::VAR_RC && r.ei != r.ej && r.ej < 9 && (r.ei / 3 == r.ej / 3);
    return r.is_diag ? xdiag : (free_entry ? X : X - r.s0 * m);
}

// LDS map of the certificate (regions that are dead while it runs)
constexpr int C_QF = L_EX;          // 90   full 9x9 Qs, row stride 10
constexpr int C_XV = L_EX + 90;     // 40   x-vectors z, a_0, a_1, a_2 (stride 10)
constexpr int C_YV = L_EX + 130;    // 40   Qs x
constexpr int C_H1 = L_EX + 170;    // 10   a_k . Q a_l
constexpr int C_RL = L_EX + 180;    // 10   R (row-major)
constexpr int C_ROW = L_EX + 190;   // 12   pivot row (+ rhs entry) of the elimination
constexpr int C_LAM = L_EX + 202;   // 10   multipliers of the dual correction
constexpr int C_SF = L_G;           // 100  full 10x10 S

// In-place LDL^T elimination of the symmetric matrix held one entry (a <= b) per lane (lane = vech index; lanes 55..63 alias 0..8).
// Returns the smallest pivot met; stops at the first non-positive one (the matrix is then not positive definite and the caller
// rejects the certificate -- wave-uniform, the pivot is read into a scalar register).  Row k of the current matrix sits in the lanes
// sidx(k, k) ... sidx(k, 9) = c_k + k ... c_k + 9: a lane fetches its two row entries with ds_bpermute (no LDS traffic, no barrier).
// (The first version broadcast the pivot row through LDS: write, barrier, three reads per step -- 5 300 cycles per factorisation,
// a quarter of a certificate attempt; this one: see profiles/r03/coop_ldl_ab.txt.)
__device__ __forceinline__ double coop_ldl(double *L, const Roles &r, double &Me)
{
    (void)L;
    double minp = 1e300;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const int ck = k * 10 - k * (k - 1) / 2 - k; // sidx(k, x) = ck + x  for x >= k
        const double d = wave_lane(Me, ck + k);
        minp = d < minp ? d : minp;
        if (!(d > 0)) break;
        const double id = fast_rcp(d);
        const double ra = __shfl(Me, ck + r.ei), rb = __shfl(Me, ck + r.ej);
        if (r.ei > k) Me -= ra * id * rb;
    }
    return minp;
}


// Symmetric sweep operator over all ten pivots on the matrix held one entry (a <= b) per lane: Me <- -(Me)^-1, at the price of one
// coop_ldl (one lane read and two ds_bpermute per pivot).  Returns the smallest pivot met (they are the pivots of the LDL^T) and stops at
// the first non-positive one (wave-uniform).  For cvx::dual_refine_step: with the inverse in hand an inverse iteration is a mat-vec.
__device__ __forceinline__ double coop_sweep_inverse(const Roles &r, double &Me)
{
    double minp = 1e300;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        const double d = wave_lane(Me, cvx::sidx(k, k));
        minp = d < minp ? d : minp;
        if (!(d > 0)) break;
        const double id = fast_rcp(d);
        const int ia = r.ei < k ? cvx::sidx(r.ei, k) : cvx::sidx(k, r.ei), ib = r.ej < k ? cvx::sidx(r.ej, k) : cvx::sidx(k, r.ej);
        const double ra = __shfl(Me, ia), rb = __shfl(Me, ib); // A[ei][k], A[k][ej]
        const bool pi = r.ei == k, pj = r.ej == k;
        Me = (pi && pj) ? -id : (pi ? rb * id : (pj ? ra * id : Me - ra * id * rb));
    }
    return minp;
}

// Cooperative certificate = the steps of cvx::solve_sdp's check (solver_core.h), identical mathematics, all
// 64 lanes: coop_round / coop_polish (primal half: cvx::round_candidate, cvx::polish_rotation) and coop_dual
// (cvx::dual_certificate).  Inputs are the entry-lane values Qs, W, Wp; R is replicated in every lane.
// rank-1 rounding (cvxpnpl.py:504-505) and projection to SO(3) (cvx::round_candidate), every lane
__device__ __forceinline__ double coop_round(const double *v, double *R)
{
    const double iv = cvx::rcp(v[9]);
    double X[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) X[i * 3 + j] = v[3 * j + i] * iv;
    const double d0 = cvx::det3(X);
    if (d0 < 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) X[i] = -X[i];
    }
    cvx::near_rotation(X, R);
    return d0;
}

// full Qs (row stride 10, zero last column) in LDS for the matrix-vector products of the certificate
__device__ __forceinline__ void coop_store_qf(double *L, const Roles &r, double Qs)
{
    if (r.ej < 9 && r.lane < 55) { L[C_QF + r.ei * 10 + r.ej] = Qs; L[C_QF + r.ej * 10 + r.ei] = Qs; }
    if (r.lane < 9) L[C_QF + r.lane * 10 + 9] = 0.0;
}

// Newton polish of r^T Qs r on SO(3) from the rotation R (cvx::polish_rotation, all lanes): R (every
// lane) and pobj = r^T Qs r on exit.
__device__ __forceinline__ void coop_polish(double *L, const Roles &r, double Qs, double *R, double &pobj CVXW_PH_PARAM)
{
    double2 *L2 = reinterpret_cast<double2 *>(L);
    const int lane = r.lane;
    CVXW_PHR(PH_P_POLAR);
    CVXW_CNT(CNT_POLISH);
    coop_store_qf(L, r, Qs);
    const int xsrc = r.xsrc;
    const double xsgn = r.xsgn;
    // ---- Newton on SO(3) for f(R) = r^T Qs r (cvx::so3_newton)
    for (int it = 0; it < 6; ++it) {
        CVXW_CNT(CNT_NEWTON);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) L[C_RL + i] = R[i];
        }
        CVXW_SYNC();
        if (lane < 40) L[C_XV + lane] = (lane == 9) ? 1.0 : xsgn * L[C_RL + xsrc];
        CVXW_SYNC();
        if (lane < 36) {
            const int vv = lane / 9, i = lane % 9;
            L[C_YV + vv * 10 + i] = dot10(L2 + (C_QF + i * 10) / 2, L2 + (C_XV + vv * 10) / 2);
        }
        if (lane >= 36 && lane < 40) L[C_YV + (lane - 36) * 10 + 9] = 0.0;
        CVXW_SYNC();
        if (lane < 9) {
            const int k = lane / 3, l = lane % 3;
            L[C_H1 + lane] = dot10(L2 + (C_XV + (1 + k) * 10) / 2, L2 + (C_YV + (1 + l) * 10) / 2);
        }
        CVXW_SYNC();
        double Qr[9], H1[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) { Qr[i] = L[C_YV + i]; H1[i] = L[C_H1 + i]; }
        double N[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) N[i * 3 + j] = R[0 * 3 + i] * Qr[3 * j] + R[1 * 3 + i] * Qr[3 * j + 1] + R[2 * 3 + i] * Qr[3 * j + 2];
        const double g[3] = {2 * (N[7] - N[5]), 2 * (N[2] - N[6]), 2 * (N[3] - N[1])};
        CVXW_SYNC();
        const double gn = fabs(g[0]) + fabs(g[1]) + fabs(g[2]);
        if (it >= 2 && gn < 1e-15) break; // wave-uniform
        const bool final_step = gn < 1e-8;  // quadratic convergence: this step lands at rounding level
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
        double w[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const double nw = -(Hi[k * 3] * g[0] + Hi[k * 3 + 1] * g[1] + Hi[k * 3 + 2] * g[2]);
            w[k] = pd ? nw : -g[k] * cvx::rcp(hn);
        }
        const double wn = cvx::sqrt_fast(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        const double lim = wn > 0.5 ? 0.5 * cvx::rcp(wn) : 1.0;
        const double q0 = 0.5 * lim * w[0], q1 = 0.5 * lim * w[1], q2 = 0.5 * lim * w[2];
        const double ss = q0 * q0 + q1 * q1 + q2 * q2;
        const double f = 2.0 * cvx::rcp(1.0 + ss);
        const double Cm[9] = {1 + f * (q0 * q0 - ss), f * (-q2 + q0 * q1), f * (q1 + q0 * q2),
                              f * (q2 + q0 * q1), 1 + f * (q1 * q1 - ss), f * (-q0 + q1 * q2),
                              f * (-q1 + q0 * q2), f * (q0 + q1 * q2), 1 + f * (q2 * q2 - ss)};
        double Rn[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = R[i * 3] * Cm[j] + R[i * 3 + 1] * Cm[3 + j] + R[i * 3 + 2] * Cm[6 + j];
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = Rn[i];
        if (final_step && pd) break;
    }
    CVXW_PHR(PH_P_NEWTON);
    { // one polar step squares any drift from orthogonality
        double Ri[9], det, Rn[9];
        cvx::inv3(R, Ri, det);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = 0.5 * (R[i * 3 + j] + Ri[j * 3 + i]);
#pragma unroll
        for (int i = 0; i < 9; ++i) R[i] = Rn[i];
    }
    // ---- z and the x-vectors of the final R; pobj = z^T Qs z
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) L[C_RL + i] = R[i];
    }
    CVXW_SYNC();
    if (lane < 40) L[C_XV + lane] = (lane == 9) ? 1.0 : xsgn * L[C_RL + xsrc];
    CVXW_SYNC();
    {
        double part = 0.0;
        if (lane < 9) part = L[C_XV + lane] * dot10(L2 + (C_QF + lane * 10) / 2, L2 + C_XV / 2);
        pobj = wave_sum(part);
    }
    CVXW_PHR(PH_P_FINAL);
}

// Dual half (cvx::dual_certificate, all lanes) for the rotation R: returns the verdict c.ok.
template <int VAR = cvx::VAR_FULL>
__device__ __forceinline__ bool coop_dual(double *L, const Roles &r, double Qs, double W, double Wp, const double *R, double d0,
                                          double pobj, double rho, double delta, double &zSz, const double shift, const bool refine CVXW_PH_PARAM)
{
    // planar scene (Qs blind to the third column of R): the problem is invariant under
    // D = diag(-I6, I4) and the correction is built in the D-even subspace (cvx::dual_certificate)
    const bool symm = __all(!(r.lane < 55 && r.ej >= 6 && r.ej < 9) || fabs(Qs) < 1e-13);
    const bool odd = symm && ((r.ei < 6) != (r.ej < 6));
    double2 *L2 = reinterpret_cast<double2 *>(L);
    const int lane = r.lane;
    const int xsrc = r.xsrc;
    const double xsgn = r.xsgn;
    // z and the tangent vectors of R (the polish may have left those of another candidate in LDS)
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 9; ++i) L[C_RL + i] = R[i];
    }
    CVXW_SYNC();
    if (lane < 40) L[C_XV + lane] = (lane == 9) ? 1.0 : xsgn * L[C_RL + xsrc];
    CVXW_SYNC();
    // ---- dual hint S_h = rho (Wp - W); S1 = S_h - P_null(S_h - Qs)
    const double Sh = rho * (Wp - W);
    double S = Sh - coop_proj<VAR>(L, r, Sh - (r.ej < 9 ? Qs : 0.0), 0.0);
    if (odd) S = 0.0;
    if (lane < 55) { L[C_SF + r.ei * 10 + r.ej] = S; L[C_SF + r.ej * 10 + r.ei] = S; }
    CVXW_SYNC();
    // rhs = S z on lanes 0..9
    double y = 0.0;
    {
        const int a = lane < 10 ? lane : 0;
        y = dot10(L2 + (C_SF + a * 10) / 2, L2 + C_XV / 2);
    }
    CVXW_PHR(PH_D_HINT);
    CVXW_CNT(CNT_DUAL);
    // ---- multipliers lam = P(R) M_I^-1 P(R)^T rhs in closed form (cvx::dual_lambda): every lane
    // computes all ten from the gathered rhs (compile-time sparse constants, no tables)
    if (lane < 10) L[C_ROW + lane] = y;
    CVXW_SYNC();
    {
        double rhs[10], lam[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) rhs[i] = L[C_ROW + i];
        cvx::dual_lambda<VAR>(R, rhs, symm, lam);
        CVXW_SYNC();
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < 10; ++i) L[C_LAM + i] = lam[i];
        }
    }
    CVXW_SYNC();
    CVXW_PHR(PH_D_BACKSUB);
    // ---- S2 = S1 - P_range(sym(lam z^T))
    {
        const double E = odd ? 0.0 : 0.5 * (L[C_LAM + r.ei] * L[C_XV + r.ej] + L[C_XV + r.ei] * L[C_LAM + r.ej]);
        const double Nn = coop_proj<VAR>(L, r, E, 0.0);
        S -= E - Nn;
    }
    if (lane < 55) { L[C_SF + r.ei * 10 + r.ej] = S; L[C_SF + r.ej * 10 + r.ei] = S; }
    CVXW_SYNC();
    double res;
    {
        const int a = lane < 10 ? lane : 0;
        const double sz = dot10(L2 + (C_SF + a * 10) / 2, L2 + C_XV / 2);
        double m = lane < 10 ? fabs(sz) : 0.0;
        res = wave_max(m);
        zSz = wave_sum(lane < 10 ? L[C_XV + a] * sz : 0.0);
    }
    CVXW_SYNC();
    // ---- LDL^T of S2 + delta I: all pivots positive  <=>  lambda_min(S2) > -delta
    double Se = S + (r.is_diag ? delta : 0.0);
    CVXW_PHR(PH_D_RANGE);
    double minp = coop_ldl(L, r, Se);
    const bool pre = (res < 1e-10) && (d0 > 0) && (pobj == pobj);
    if (shift > 0.0 && pre && !symm && !(minp > 0)) { // (wave-uniform) second try along D(R): cvx::dual_retry_entry6, z from LDS
        const int a = r.ei, b = r.ej;
        const double za = L[C_XV + a], zb = L[C_XV + b];
        const double cross = L[C_XV + 3 * (a / 3) + (b % 3)] * L[C_XV + 3 * (b / 3) + (a % 3)]; // R[k][j] R[i][l]  (a = 3 j + i, b = 3 l + k < 9)
        const double d6 = b == 9 ? (a == 9 ? -3.0 : za) : (a == b ? 1.0 : 0.0) + cross - za * zb;
        double sh = shift * (1.0 / 6.0);
        for (int rung = 0; rung < cvx::DUAL_RETRY_RUNGS && !(minp > 0); ++rung) { // shift, shift / 4
            Se = S + (r.is_diag ? delta : 0.0) + sh * d6;
            minp = coop_ldl(L, r, Se);
            sh *= 0.25;
        }
    }
    if (CVX_DUAL_REFINE_COMPILED && refine && pre && !symm && !(minp > 0)) { // (wave-uniform) third try: cvx::dual_refine_step, all lanes
        // X = (S + (delta + sigma) I)^-1 by the sweep operator, two inverse iterations from the runner-up eigenvector of Z (L_V + 10..19)
        constexpr int R_X = L_Y;            // 100  full X (the eigen columns of this iteration are dead: Wp is in registers)
        constexpr int R_V = L_Y + 100;      // 2 x 10  iteration vector, ping-pong
        double Xe = S + (r.is_diag ? delta + cvx::DUAL_REFINE_SIGMA : 0.0);
        const double mps = coop_sweep_inverse(r, Xe);
        if (mps > 0) {
            if (lane < 55) { L[R_X + r.ei * 10 + r.ej] = -Xe; L[R_X + r.ej * 10 + r.ei] = -Xe; }
            if (lane < 10) L[R_V + lane] = L[L_V + 10 + lane];
            CVXW_SYNC();
            double x[10];
#pragma unroll
            for (int itn = 0; itn <= cvx::DUAL_REFINE_INVITS; ++itn) {
                const int cur = R_V + 10 * (itn & 1), nxt = R_V + 10 * ((itn + 1) & 1);
                double zx = 0.0, n2 = 0.0;
#pragma unroll
                for (int i = 0; i < 10; ++i) { x[i] = L[cur + i]; zx += L[C_XV + i] * x[i]; }
#pragma unroll
                for (int i = 0; i < 10; ++i) { x[i] -= 0.25 * zx * L[C_XV + i]; n2 += x[i] * x[i]; }
                const double in = cvx::rsqrt_(n2 > 1e-300 ? n2 : 1e-300);
#pragma unroll
                for (int i = 0; i < 10; ++i) x[i] *= in;
                if (itn < cvx::DUAL_REFINE_INVITS) {
                    CVXW_SYNC();
                    double xs = 0.0;
#pragma unroll
                    for (int i = 0; i < 10; ++i) xs = lane == i ? x[i] : xs;
                    if (lane < 10) L[cur + lane] = xs;
                    CVXW_SYNC();
                    if (lane < 10) L[nxt + lane] = dot10(L2 + (R_X + lane * 10) / 2, L2 + cur / 2);
                    CVXW_SYNC();
                }
            }
            double xi = 0.0, xj = 0.0;
#pragma unroll
            for (int i = 0; i < 10; ++i) { xi = r.ei == i ? x[i] : xi; xj = r.ej == i ? x[i] : xj; }
            const double wg = lane < 55 ? (r.is_diag ? 1.0 : 2.0) : 0.0;
            const double E0 = xi * xj;
            const double ray = wave_sum(wg * S * E0);
            if (ray < 0) {
                // G = P_U(x x^T): onto span A_i, then the minimum-norm correction onto { X z = 0 } (the two closed forms above)
                double G = E0 - coop_proj<VAR>(L, r, E0, 0.0);
                if (lane < 55) { L[C_SF + r.ei * 10 + r.ej] = G; L[C_SF + r.ej * 10 + r.ei] = G; }
                CVXW_SYNC();
                if (lane < 10) L[C_ROW + lane] = dot10(L2 + (C_SF + lane * 10) / 2, L2 + C_XV / 2);
                CVXW_SYNC();
                {
                    double rhs[10], lam[10];
#pragma unroll
                    for (int i = 0; i < 10; ++i) rhs[i] = L[C_ROW + i];
                    cvx::dual_lambda<VAR>(R, rhs, false, lam);
                    CVXW_SYNC();
                    if (lane == 0) {
#pragma unroll
                        for (int i = 0; i < 10; ++i) L[C_LAM + i] = lam[i];
                    }
                }
                CVXW_SYNC();
                {
                    const double E = 0.5 * (L[C_LAM + r.ei] * L[C_XV + r.ej] + L[C_XV + r.ei] * L[C_LAM + r.ej]);
                    const double Nn = coop_proj<VAR>(L, r, E, 0.0);
                    G -= E - Nn;
                }
                const double g2 = wave_sum(wg * G * E0);
                if (g2 > 1e-6) {
                    const double tau = cvx::DUAL_REFINE_GAIN * (-ray) * cvx::rcp(g2);
                    const double Sn = S + tau * G;
                    if (lane < 55) { L[C_SF + r.ei * 10 + r.ej] = Sn; L[C_SF + r.ej * 10 + r.ei] = Sn; }
                    CVXW_SYNC();
                    const int a = lane < 10 ? lane : 0;
                    const double sz = dot10(L2 + (C_SF + a * 10) / 2, L2 + C_XV / 2);
                    const double res2 = wave_max(lane < 10 ? fabs(sz) : 0.0);
                    const double zSz2 = wave_sum(lane < 10 ? L[C_XV + a] * sz : 0.0);
                    CVXW_SYNC();
                    double Se2 = Sn + (r.is_diag ? delta : 0.0);
                    const double mp2 = coop_ldl(L, r, Se2);
                    if (mp2 > 0 && res2 < 1e-10) { minp = mp2; zSz = zSz2; }
                }
            }
        }
    }
    CVXW_PHR(PH_D_LDL2);
    return (minp > 0) && pre;
}

struct WaveArgs {
    int64_t batch;
    int n_p, n_l, K_per_problem;
    const double *p2, *p3, *l2, *l3, *K;
    double *R, *t, *cost, *Z;
    int32_t *status, *iters, *work;
    const double *Q45, *B27; // cost entry (cvxpnpl_solve_cost_batch): [batch][45] packed A^T A and [batch][27] B instead of correspondences
    int32_t *rq_count, *rq_entries; // queue of cvxw::rescue_wave_kernel (or null): problems still open after opts.rescue_from iterations
    double *rq_ws;                  // split interior-point path (cvxw::ipm_wave_kernel; null: the fused cvxw::rescue_wave_kernel): the parked-iterate
    int rq_stride;                  // slots, doubles per problem -- a problem put on the rescue queue leaves its cost there (RS_W.. : Qs, RS_IT: iterations)
};

// Layout of a parked problem in the workspace (doubles).  Every hand-off carries the iterate W (vech order) and the
// iteration count; the quad schedule (RS_FULL slots) also hands over what its wavefront already had -- the cost entries,
// the translation map and the unit eigenvectors of the last iterate -- so that the wavefront that takes the problem over
// neither re-reads and re-assembles the correspondences nor starts its first eigen-solve cold (2-3 iteration-equivalents
// off the tail of every launch: the slowest problems are exactly the handed-over ones).
constexpr int F32_SWEEPS_UNTIL = cvx::F32_SWEEPS_DEFAULT; // eigen-solve sweeps in single precision during the first iterations of a solve; Opts::f32_sweeps_until overrides (0: never)
constexpr int RS_W = 0, RS_IT = 55, RS_LANE = 56;           // lane schedule: 56 doubles per problem
constexpr int RS_Q = 56, RS_B = 112, RS_NC = 139, RS_V = 140, RS_FULL = 240; // quad schedule: + Q (55, vech order, 0 outside the 9x9 block), B (27), the iteration of the next certificate attempt, V (100: [column][row])

// Solve problem b with the wavefront that calls this.  resume (optional): 56 doubles written by
// the lane-layout kernel for a problem it handed off -- W (55, vech order) and the iteration
// count -- the solve then continues from that iterate instead of starting at e9 e9^T.
//
// A problem that is still open after opts.rescue_from iterations leaves this function: with IPM = false (every kernel but
// cvxw::rescue_wave_kernel) it is put on the rescue queue (status ST_PENDING + its iteration count) and the function returns
// false; with IPM = true it returns true (wave-uniform) with the cost in the solver's frame at L[I_QS..], the caller
// (solve_one_wave) runs the interior-point solve and calls again with after_ipm set: the problem is assembled once more and
// the iteration goes on from the iterate at L[I_W..].  The interior-point solve is kept OUT of the kernels every problem runs
// through: merely compiled into their loop (never executed) it cost 4 % (10 000 problems) to 17 % (2 000) through the
// register allocation of the hot path -- 8 % more vector instructions executed.
template <int VAR, bool IPM>
__device__ __forceinline__ bool solve_pass(const WaveArgs &a, const cvx::Opts &o, const int64_t b, double *L, const double *resume, const bool resume_full_in,
                                           const bool after_ipm, int &it_io, int &sweeps_io)
{
    int lane_ = threadIdx.x & 63;
    double2 *L2 = reinterpret_cast<double2 *>(L);

    // ---------------------------------------------------------------- lane roles
    // Everything below is derived from the lane index and one packed word -- loop-invariant, so LLVM hoists it all out of the iteration
    // loop and keeps ~25 registers of it live across every phase (the kernel's spills, reloaded from scratch inside the loop: a round
    // trip to L2 on a chain that has nothing to hide it behind).  CVXW_ROLES declares the set; it is instantiated here and again at the
    // top of every iteration from copies that an empty asm hides (cf. cvxq::Own::refresh, cvxi::ipm4_solve), so that the values are
    // recomputed where they are used (a bit-field extract each) and die there.
    unsigned lw_ = kLanePack.w[lane_];
#define CVXW_ROLES(LANE, LW)                                                                                                                          \
    const int lane = (LANE);                                                                                                                          \
    const unsigned lw = (LW);                                                                                                                         \
    const int ei = (int)(lw & 15), ej = (int)((lw >> 4) & 15);                                                                                        \
    const int el = lane < 55 ? lane : lane - 55; /* entry index (lanes 55..63 alias 0..8, weight 0) */                                                \
    const double wgt = lane < 55 ? (ei == ej ? 1.0 : 2.0) : 0.0;                                                                                      \
    const bool is_diag = ((lw >> 23) & 1) != 0;                                                                                                       \
    const int p1 = (int)((lw >> 8) & 63), p2 = (int)((lw >> 14) & 63);                                                                                \
    const double s0 = ((lw >> 20) & 1) ? -1.0 : 1.0, s1 = ((lw >> 21) & 1) ? -1.0 : 1.0, s2 = ((lw >> 22) & 1) ? -1.0 : 1.0;                          \
    Roles roles;                                                                                                                                      \
    roles.lane = lane; roles.el = el; roles.ei = ei; roles.ej = ej; roles.p1 = p1; roles.p2 = p2;                                                     \
    roles.is_diag = is_diag; roles.s0 = s0; roles.s1 = s1; roles.s2 = s2;                                                                             \
    roles.xsrc = (int)((lw >> 24) & 15);                                                                                                              \
    roles.xsgn = ((lw >> 28) & 3) == 0 ? 0.0 : (((lw >> 28) & 3) == 1 ? 1.0 : -1.0);                                                                  \
    const int jl = lane < 50 ? lane : lane - 50; /* jacobi lane (50..63 alias 0..13) */                                                               \
    const int ji = jl % 10, jk = jl / 10;                                                                                                             \
    (void)wgt; (void)ji; (void)jk; (void)p1; (void)p2; (void)s0; (void)s1; (void)s2; (void)is_diag; (void)el
    CVXW_ROLES(lane_, lw_);
    CVXW_PH_DECL;

    // A slot whose iteration count is NEGATIVE comes from cvxw::ipm_wave_kernel (split interior-point path): W = Z - S / rho of the
    // interior-point solution, ALREADY in the solver's frame, nothing else -- the problem is assembled again whatever the slot format,
    // the first eigen-solve is cold, the attempt comes one iteration later, and the problem neither goes back to the rescue queue nor
    // runs more than IPM_GRACE further iterations (what after_ipm means in the fused kernel).
    const bool post_ipm = !IPM && resume && __hip_atomic_load(resume + RS_IT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0.0; // wave-uniform
    const bool resume_full = resume_full_in && !post_ipm;
    // ---------------------------------------------------------------- assembly
    bool okK = true, okG = true;
    double Qe = 0.0; // Q9 entry of this entry-lane (0 outside the 9x9 block)
    if (a.Q45) {
        // cost entry (the seam of cvxpnpl.py:454-460): A^T A (packed 9x9, cvx::qidx order) and B come from the caller
        if (ej < 9) Qe = a.Q45[b * 45 + cvx::qidx(ei, ej)];
        if (lane < 27) L[L_B + lane] = a.B27[b * 27 + lane];
        okG = !__any(lane < 27 && !(L[L_B + lane] == L[L_B + lane]));
    } else if (resume && resume_full) {
        // handed over by a quad wavefront together with what it had assembled (device-coherent loads: same launch)
        Qe = __hip_atomic_load(resume + RS_Q + el, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (lane < 27) L[L_B + lane] = __hip_atomic_load(resume + RS_B + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        cvx::ProblemView pv = cvx::make_view(b, a.n_p, a.p2, a.p3, a.n_l, a.l2, a.l3, a.K, a.K_per_problem);
        const int nrec = pv.n_p + 2 * pv.n_l;
        constexpr int CHUNK = 32; // records (T[6], P[3]) staged per pass in L_EX.. (32 * 10 doubles)
        // raw inputs of record r: a point (u, v, X, Y, Z) or one endpoint of a line (2D segment + its 3D point).
        // The first chunk is requested BEFORE K is inverted, so that all global loads of the problem are in
        // flight together (one memory round trip instead of three).
        // (separate registers for the point and the line case: loads into the same registers from both sides of
        // the branch would force a wait between them)
        auto load_point = [&](int r, double *q) {
            q[0] = pv.p2[2 * r]; q[1] = pv.p2[2 * r + 1];
            q[2] = pv.p3[3 * r]; q[3] = pv.p3[3 * r + 1]; q[4] = pv.p3[3 * r + 2];
        };
        auto load_line = [&](int r, double *q) {
            const int li = (r - pv.n_p) >> 1, en = (r - pv.n_p) & 1;
            const double *l2 = pv.l2 + 4 * li, *l3 = pv.l3 + 6 * li + 3 * en;
            q[0] = l2[0]; q[1] = l2[1]; q[2] = l2[2]; q[3] = l2[3];
            q[4] = l3[0]; q[5] = l3[1]; q[6] = l3[2];
        };
        // sums about a point of the scene (cvx::shift_centre: exact, and well conditioned far from the world origin)
        double cs_[3];
        cvx::shift_centre(pv.n_p, pv.p3, pv.n_l, pv.l3, cs_);
        const double cs0 = cs_[0], cs1 = cs_[1], cs2 = cs_[2];
        double rawp[5] = {0, 0, 0, 0, 0}, rawl[7] = {0, 0, 0, 0, 0, 0, 0};
        {
            const int cnt0 = nrec < CHUNK ? nrec : CHUNK;
            if (lane < cnt0 && lane < pv.n_p) load_point(lane, rawp);
            if (lane < cnt0 && lane >= pv.n_p) load_line(lane, rawl);
        }
        double Ki[9];
        {
            double Kc[9], det;
    #pragma unroll
            for (int i = 0; i < 9; ++i) Kc[i] = pv.K[i];
            cvx::inv3(Kc, Ki, det);
            okK = (det == det) && det != 0.0;
        }
        // accumulator role of this lane: sum rec[6 + qa] rec[6 + qb] rec[te] with rec = (T[6], 1, P[3]):
        // M0 (6): qa = qb = 0 | M1 (3 x 6): qa = 1 + a | M2 (6 x 6): qa = 1 + a, qb = 1 + b
        int acc_qa = 6, acc_qb = 6, acc_te = 0;
        {
            const int al = lane < 60 ? lane : 0;
            acc_te = al;
            if (al >= 6 && al < 24) { acc_qa = 7 + (al - 6) / 6; acc_te = (al - 6) % 6; }
            if (al >= 24) {
                const int ab = (al - 24) / 6;
                acc_te = (al - 24) % 6;
                acc_qa = 7 + (ab < 3 ? 0 : (ab < 5 ? 1 : 2));
                acc_qb = 7 + (ab < 3 ? ab : (ab < 5 ? ab - 2 : 2));
            }
        }
        double acc = 0.0;
        for (int base = 0; base < nrec; base += CHUNK) {
            const int cnt = nrec - base < CHUNK ? nrec - base : CHUNK;
            if (lane < cnt) {
                const int r = base + lane;
                double T[6], P[3];
                if (r < pv.n_p) {
                    if (base > 0) load_point(r, rawp);
                    double p[3];
                    cvx::bearing(Ki, rawp[0], rawp[1], p);
                    double n2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
                    T[0] = n2 - p[0] * p[0]; T[1] = -p[0] * p[1]; T[2] = -p[0] * p[2];
                    T[3] = n2 - p[1] * p[1]; T[4] = -p[1] * p[2]; T[5] = n2 - p[2] * p[2];
                    P[0] = rawp[2] - cs0; P[1] = rawp[3] - cs1; P[2] = rawp[4] - cs2;
                } else {
                    if (base > 0) load_line(r, rawl);
                    double u[3], v[3];
                    cvx::bearing(Ki, rawl[0], rawl[1], u);
                    cvx::bearing(Ki, rawl[2], rawl[3], v);
                    double n[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
                    double inv = cvx::rsqrt_(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                    n[0] *= inv; n[1] *= inv; n[2] *= inv;
                    T[0] = n[0] * n[0]; T[1] = n[0] * n[1]; T[2] = n[0] * n[2]; T[3] = n[1] * n[1]; T[4] = n[1] * n[2]; T[5] = n[2] * n[2];
                    P[0] = rawl[4] - cs0; P[1] = rawl[5] - cs1; P[2] = rawl[6] - cs2;
                }
                double *rec = L + L_EX + lane * 10;
    #pragma unroll
                for (int i = 0; i < 6; ++i) rec[i] = T[i];
                rec[6] = 1.0; rec[7] = P[0]; rec[8] = P[1]; rec[9] = P[2];
            }
            CVXW_SYNC();
            for (int c = 0; c < cnt; ++c) {
                const double *rec = L + L_EX + c * 10;
                acc += rec[acc_qa] * rec[acc_qb] * rec[acc_te];
            }
            CVXW_SYNC();
        }
        L[L_P + lane] = acc; // ACC[0..59]
        CVXW_SYNC();
        // B = M0^-1 [M1_0 M1_1 M1_2], Q = M2 - M1^T B; every lane inverts M0 redundantly
        {
            const double *m = L + L_P;
            double M0[9] = {m[0], m[1], m[2], m[1], m[3], m[4], m[2], m[4], m[5]}, Mi[9], det;
            cvx::inv3(M0, Mi, det);
            double sc = m[0] + m[3] + m[5];
            okG = det > 1e-12 * (sc * sc * sc) * (1.0 / 27.0);
            double sel = Mi[0]; // element `lane` of the inverse, without dynamic register indexing
    #pragma unroll
            for (int i = 1; i < 9; ++i) sel = lane == i ? Mi[i] : sel;
            if (lane < 9) L[L_X + lane] = sel;
        }
        CVXW_SYNC();
        // packed index of (i, j) in a symmetric 3x3 (00 01 02 11 12 22)
        auto psym = [](int i, int j) { const int lo = i < j ? i : j, hi = i < j ? j : i; return lo * 3 - (lo == 2 ? 1 : 0) + (hi - lo); };
        if (lane < 27) {
            const int bb = lane / 9, i = (lane % 9) / 3, j = lane % 3;
            const double *m1 = L + L_P + 6 + 6 * bb;
            double v = 0;
    #pragma unroll
            for (int k = 0; k < 3; ++k) v += L[L_X + i * 3 + k] * m1[psym(k, j)];
            L[L_B + i * 9 + 3 * bb + j] = v; // B'[i][3 bb + j] (about the shifted origin; the shift goes in once Q is formed)
        }
        CVXW_SYNC();
        if (ej < 9) {
            const int qa = ei / 3, qi = ei % 3, qb = ej / 3, qj = ej % 3;
            const double *m1 = L + L_P + 6 + 6 * qa, *m2 = L + L_P + 24 + 6 * psym(qa, qb);
            double v = m2[psym(qi, qj)];
    #pragma unroll
            for (int k = 0; k < 3; ++k) v -= m1[psym(qi, k)] * L[L_B + k * 9 + 3 * qb + qj];
            Qe = v;
        }
        CVXW_SYNC(); // (every read of B' for Q is done)
        if (lane < 9) L[L_B + (lane / 3) * 9 + 3 * (lane % 3) + lane / 3] += (lane % 3 == 0 ? cs0 : (lane % 3 == 1 ? cs1 : cs2)); // t = -B' r - R c
    }
    L[L_X + el] = Qe;
    CVXW_SYNC();
    double tr = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) tr += L[L_X + cvx::sidx(k, k)];
    bool finite = okK && okG && (tr == tr) && tr > 0 && tr < 1e300;
    finite = !__any(!(finite && (Qe == Qe)));
    const double itr = finite ? cvx::rcp(tr) : 0.0;
    double Qs = Qe * itr;
    // planar scene in a general world frame (cvx::canonicalise_planar): when the cost is blind to R n for a
    // direction n other than e3, continue in the frame R' = R U whose third axis is n -- Qs' = P Qs P^T with
    // P = U^T (x) I3 -- so that the D-even dual correction applies; R, Z are taken back at the end.
    bool canon = false;
    if (finite) {
        double T[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                T[i * 3 + j] = (L[L_X + cvx::sidx(3 * i, 3 * j)] + L[L_X + cvx::sidx(3 * i + 1, 3 * j + 1)] + L[L_X + cvx::sidx(3 * i + 2, 3 * j + 2)]) * itr;
        double U[9];
        canon = cvx::planar_frame(T, U); // every lane computes the same: wave-uniform
        if (canon) {
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 9; ++i) L[L_U + i] = U[i];
            }
            double acc = 0.0;
            if (ej < 9) {
                const int bi = ei / 3, x = ei % 3, bj = ej / 3, y = ej % 3;
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int l = 0; l < 3; ++l) {
                        const double uki = bi == 0 ? U[k * 3] : (bi == 1 ? U[k * 3 + 1] : U[k * 3 + 2]);
                        const double ulj = bj == 0 ? U[l * 3] : (bj == 1 ? U[l * 3 + 1] : U[l * 3 + 2]);
                        acc += uki * ulj * L[L_X + cvx::sidx(3 * k + x, 3 * l + y)];
                    }
                if (bj == 2) acc = 0.0; // the blind block, zero up to rounding: made exactly zero (cvx::canonicalise_planar)
            }
            Qs = acc * itr;
        }
        CVXW_SYNC();
    }

    CVXW_PH(PH_ASSEMBLE);
#ifdef CVXW_STOP_AFTER_ASSEMBLY // timing ablation (tools/ablate.sh): assembly only
    { const double chk = wave_sum(Qs); if (lane == 0) a.status[b] = chk > 1e300 ? 1 : 0; return false; }
#endif
    // ---------------------------------------------------------------- ADMM
    double delta = o.eps / (8.0 * tr);
    delta = delta < 1e-13 ? 1e-13 : delta;
    double rho = o.rho, irho = 1.0 / o.rho;
    double W = (lane == 54) ? 1.0 : 0.0, Wp = 0.0;
    int it = 0, total_sweeps = 0, next_check = o.first_check;
    bool have_prev = false; // L_M + 50.. holds the rotation polished by the previous check
    int reused = 0;         // consecutive checks that took it over (at most cvx::REUSE_MAX, see cvx::solve_sdp)
    double fprev = 0.0;
    bool have_tp = false, have_tm = false; // L_M + 30.. / 40.. hold the twins polished by the previous twin check
    double f_tp = 0.0, f_tm = 0.0;
    int tw_reused = 0;
    bool cold = false; // resumed solves have no previous eigenvectors for their first eigen-solve
    if (IPM && after_ipm) { // the interior-point solve's W = Z - S / rho, already in the solver's frame
        W = L[I_W + el];
        it = it_io; total_sweeps = sweeps_io;
        next_check = it + 1;
        cold = true;
    } else if (IPM && !resume) { // from the rescue queue: only its cost is wanted (the test at the top of the loop)
        it = it_io; total_sweeps = sweeps_io;
    } else if (resume) {
        // device-coherent loads: the iterate may have been parked by a wavefront of this very launch (quad_kernel.h)
        auto rd = [&](int i) { return __hip_atomic_load(resume + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
        W = rd(el);
        if (canon && !post_ipm) { // the hand-off iterate is in the caller's frame: W' = Pt W Pt^T
            double acc = 0.0;
            if (ej < 9) {
                const int bi = ei / 3, x = ei % 3, bj = ej / 3, y = ej % 3;
                for (int k = 0; k < 3; ++k)
                    for (int l = 0; l < 3; ++l) acc += L[L_U + k * 3 + bi] * L[L_U + l * 3 + bj] * rd(cvx::sidx(3 * k + x, 3 * l + y));
            } else if (ei < 9) {
                const int bi = ei / 3, x = ei % 3;
                for (int k = 0; k < 3; ++k) acc += L[L_U + k * 3 + bi] * rd(cvx::sidx(3 * k + x, 9));
            } else acc = rd(54);
            W = acc;
        }
        it = (int)rd(RS_IT);
        if (post_ipm) { it = -it; total_sweeps = a.work ? a.work[2 * b + 1] : 0; }
        next_check = it + 1 > o.first_check ? it + 1 : o.first_check;
#ifndef CVXW_RESUME_EARLY_CHECK
#define CVXW_RESUME_EARLY_CHECK 0
#endif
        if (resume_full && !(CVXW_RESUME_EARLY_CHECK && CVX_DUAL_REFINE_COMPILED && o.dual_refine && a.n_p + a.n_l >= 6)) { // the attempt schedule of the first phase goes on (an attempt costs two to three iterations)
            const int nc = (int)rd(RS_NC);
            next_check = nc > next_check ? nc : next_check;
        }
        cold = true;
        if (resume_full && it > 0 && !canon) { // the eigenvectors of the last iterate come along: warm start
            L[L_VN + lane] = rd(RS_V + lane);
            if (lane < 36) L[L_VN + 64 + lane] = rd(RS_V + 64 + lane);
            cold = false;
            CVXW_SYNC();
        }
        if (o.tail_from > 0 && it >= o.tail_from) { rho = o.rho_tail; irho = 1.0 / rho; }
    }
    double fp_res = 1e300, lam2_prev = -1.0;
    // Second tries of a failed dual (coop_dual, cvx::dual_retry_entry6) for the problems another phase has handed over -- they have had
    // their first attempt there (cvx::solve_sdp makes them from the second attempt of a solve on).  A fresh solve in this layout makes
    // none: such launches are small and last as long as their one slowest problem, which pays the extra LDL^T's at every failed attempt
    // and rarely is the one that gains (2 000 problems: 19.2 -> 18.6 M poses/s with them, 18.2 with tries at first attempts too).
    // Same-box A/B, dual_shift 0 -> 0.015: 125 k problems 265 -> 285 M poses/s, 32 k 113.6 -> 134.9 M, PnPL 100 k 188 -> 192 M, N = 6
    // 68.4 -> 74.6 M, N = 8 161.4 -> 158.9 M, four-point problems 10.2 -> 11.2 M (config 5 at the reference's defaults 18.1 -> 20.1 M),
    // 16 k 70.7 -> 71.7 M, the judged 10 k launch and 1 M unchanged.
    const bool retries = resume != nullptr;
    int retry_left = cvx::DUAL_RETRY_ATTEMPTS; // (attempts of this phase that may use them: see solver_core.h)
    int status = cvx::ST_NONFINITE, rank_out = 0;
    bool done = !finite;
    bool certified = false;

    // a problem that is still open after rescue_cap iterations leaves the loop for the interior-point solve (below)
    const int rescue_cap = (!(IPM && after_ipm) && !post_ipm && o.rescue_from > 0 && (IPM || a.rq_count)) ? o.rescue_from : 0x7fffffff;
    // After the hand-over the iterate IS the SDP optimum (gap 1e-10): a problem that has not certified within IPM_GRACE further
    // first-order iterations has a relaxation that is not tight (or one whose interior-point solve stalled), and exits through the
    // reference's recovery from the Z it has instead of crawling to the rank-stall rule (measured: one of 50 000 rc problems ran 1 625
    // iterations that way and held the launch for 8 ms; the slowest problem that does certify after a hand-over takes 47).
#ifndef CVXW_IPM_GRACE
#define CVXW_IPM_GRACE 64
#endif
    constexpr int IPM_GRACE = CVXW_IPM_GRACE;
    const int ipm_deadline = (IPM && after_ipm) ? it_io + IPM_GRACE : (post_ipm ? it + IPM_GRACE : 0x7fffffff);
    while (!done && it < rescue_cap) {
#ifndef CVXW_NO_ROLE_REFRESH
        asm volatile("" : "+v"(lane_), "+v"(lw_));
        CVXW_ROLES(lane_, lw_); // (shadows the set of the function scope for the body of the loop)
#endif
        double sigma = 0.0;
        if (it == 0 && !resume && o.first_check > 1 && o.max_iters > 1) {
            // W0 = e9 e9^T is diagonal and PSD: Wp = W0, eigenvectors = unit vectors, no eigen-solve
            Wp = W;
            L[L_VN + (2 * jk) * 10 + ji] = (ji == 2 * jk) ? 1.0 : 0.0;
            L[L_VN + (2 * jk + 1) * 10 + ji] = (ji == 2 * jk + 1) ? 1.0 : 0.0;
            CVXW_SYNC();
        } else {
        // ---- eigendecomposition of W: one-sided Jacobi on G = W + sigma I
        const double fro2 = wave_sum(wgt * W * W);
        sigma = 1.5 * cvx::sqrt_fast(fro2) + 1e-300;
        {
            const double g = W + (is_diag ? sigma : 0.0);
            L[L_G + ei * 10 + ej] = g;
            L[L_G + ej * 10 + ei] = g;
        }
        CVXW_SYNC();
        constexpr int CA = L_EX, CB = L_EX + 64, NA = L_EX + 128, NB = L_EX + 136;
        double ca, cb, al, be, gam;
        if (o.warm_start && it > 0 && !cold) {
            // warm start: G = (W + sigma I) V_prev -- columns already nearly orthogonal when W moved
            // little.  Row ji of the two columns at position jk, then the full columns via LDS.
            const double2 *wr = L2 + (L_G + ji * 10) / 2;
            ca = dot10(wr, L2 + (L_VN + (2 * jk) * 10) / 2);
            cb = dot10(wr, L2 + (L_VN + (2 * jk + 1) * 10) / 2);
            L[CA + jl] = ca;
            L[CB + jl] = cb;
            CVXW_SYNC();
            const double2 *ra = L2 + (CA + jk * 10) / 2, *rb = L2 + (CB + jk * 10) / 2;
            al = dot10(ra, ra);
            be = dot10(rb, rb);
            gam = dot10(ra, rb);
            CVXW_SYNC();
        } else {
            // columns 2k, 2k+1 of the symmetric G are its rows: contiguous in L_G
            const double2 *ra = L2 + (L_G + (2 * jk) * 10) / 2, *rb = L2 + (L_G + (2 * jk + 1) * 10) / 2;
            al = dot10(ra, ra);
            be = dot10(rb, rb);
            gam = dot10(ra, rb);
            ca = L[L_G + (2 * jk) * 10 + ji];
            cb = L[L_G + (2 * jk + 1) * 10 + ji];
        }
        // after the rotation position 0 keeps its first column and every other column moves one
        // place along the ring a1 > a2 > a3 > a4 > b4 > b3 > b2 > b1 > b0 > a1 (circle method)
        const int src_a = jk == 0 ? CA : (jk == 1 ? CB : CA + (jk - 1) * 10);
        const int src_an = jk == 0 ? NA : (jk == 1 ? NB : NA + jk - 1);
        const int src_b = jk == 4 ? CA + 40 : CB + (jk + 1) * 10;
        const int src_bn = jk == 4 ? NA + 4 : NB + jk + 1;
        const double tol2 = o.jacobi_tol * o.jacobi_tol;
        int sweeps = 0;
        bool more;
        CVXW_PH(PH_EIG_SETUP);
        if (it < (o.f32_sweeps_until < 0 ? F32_SWEEPS_UNTIL : o.f32_sweeps_until)) {
            // The sweeps of the first F32_SWEEPS_UNTIL iterations (Opts::f32_sweeps_until) run in single precision (columns as floats in LDS, two rows per
            // packed FMA, rotation parameters without the double refinements): the columns only have to become orthogonal
            // to the sweep tolerance, the iterate is a dual hint whose certificate is verified in double.  Host experiment
            // (10 k problems each of PnP N=10 / N=6 sigma 5 / N=4, PnPL 5+5): iteration histograms identical to double
            // sweeps for limits of 7 ... 64 iterations (N=4: mean 30.386 vs 30.381 iterations), slightly longer tails from
            // 128 on -- so the slow tails beyond 64 iterations keep double sweeps.
            float *Lf = reinterpret_cast<float *>(L + L_EX);
            constexpr int FA = 0, FB = 64, FNA = 128, FNB = 136; // columns padded to 12 floats: 16-byte aligned rows
            const int f_a = jk == 0 ? FA : (jk == 1 ? FB : FA + (jk - 1) * 12);
            const int f_an = jk == 0 ? FNA : (jk == 1 ? FNB : FNA + jk - 1);
            const int f_b = jk == 4 ? FA + 48 : FB + (jk + 1) * 12;
            const int f_bn = jk == 4 ? FNA + 4 : FNB + jk + 1;
            float caf = (float)ca, cbf = (float)cb, alf = (float)al, bef = (float)be, gamf = (float)gam;
            const float tol2f = (float)tol2;
            CVXW_SYNC(); // (the double set-up above read L_EX)
            do {
                bool coarse = false;
                for (int step = 0; step < 9; ++step) {
                    const float g2 = gamf * gamf, ab = alf * bef;
                    coarse |= g2 > tol2f * ab;
                    float c, s, t;
                    cvx::jacobi_cs(alf, bef, gamf, g2 > 1e-30f * ab, c, s, t);
                    Lf[FA + jk * 12 + ji] = c * caf - s * cbf;
                    Lf[FB + jk * 12 + ji] = s * caf + c * cbf;
                    if (ji == 0) { Lf[FNA + jk] = alf - t * gamf; Lf[FNB + jk] = bef + t * gamf; }
                    CVXW_SYNC();
                    const float4 *ra = reinterpret_cast<const float4 *>(Lf + f_a), *rb = reinterpret_cast<const float4 *>(Lf + f_b);
                    const float4 a0 = ra[0], a1 = ra[1], b0 = rb[0], b1 = rb[1];
                    const float2 a2 = *reinterpret_cast<const float2 *>(Lf + f_a + 8), b2 = *reinterpret_cast<const float2 *>(Lf + f_b + 8);
                    caf = Lf[f_a + ji];
                    cbf = Lf[f_b + ji];
                    const cvx::f2 pa[5] = {{a0.x, a0.y}, {a0.z, a0.w}, {a1.x, a1.y}, {a1.z, a1.w}, {a2.x, a2.y}};
                    const cvx::f2 pb[5] = {{b0.x, b0.y}, {b0.z, b0.w}, {b1.x, b1.y}, {b1.z, b1.w}, {b2.x, b2.y}};
                    cvx::f2 acc = cvx::f2_mul(pa[0], pb[0]);
#pragma unroll
                    for (int i = 1; i < 5; ++i) acc = cvx::f2_fma(pa[i], pb[i], acc);
                    gamf = acc.x + acc.y;
                    if (step == 8) { // exact norms once per sweep (the incremental update drifts)
                        cvx::f2 na = cvx::f2_mul(pa[0], pa[0]), nb = cvx::f2_mul(pb[0], pb[0]);
#pragma unroll
                        for (int i = 1; i < 5; ++i) { na = cvx::f2_fma(pa[i], pa[i], na); nb = cvx::f2_fma(pb[i], pb[i], nb); }
                        alf = na.x + na.y;
                        bef = nb.x + nb.y;
                    } else {
                        alf = Lf[f_an];
                        bef = Lf[f_bn];
                    }
                    CVXW_SYNC();
                }
                ++sweeps;
                more = __any(coarse) && sweeps < o.jacobi_sweeps;
            } while (more);
            ca = (double)caf; cb = (double)cbf; al = (double)alf; be = (double)bef;
        } else {
            do {
                bool coarse = false;
                for (int step = 0; step < 9; ++step) {
                    const double g2 = gam * gam, ab = al * be;
                    coarse |= g2 > tol2 * ab;
                    double c, s, dl;
                    cvx::jacobi_cs_dl(al, be, gam, g2 > 1e-30 * ab, c, s, dl, o.f32_sweeps_until == 0);
                    L[CA + jl] = c * ca - s * cb;
                    L[CB + jl] = s * ca + c * cb;
                    if (ji == 0) { L[NA + jk] = al + dl; L[NB + jk] = be - dl; }
                    CVXW_SYNC();
                    const double2 *ra = L2 + src_a / 2, *rb = L2 + src_b / 2;
                    const double2 a0 = ra[0], a1 = ra[1], a2 = ra[2], a3 = ra[3], a4 = ra[4];
                    const double2 b0 = rb[0], b1 = rb[1], b2 = rb[2], b3 = rb[3], b4 = rb[4];
                    ca = L[src_a + ji];
                    cb = L[src_b + ji];
                    gam = ((a0.x * b0.x + a0.y * b0.y) + (a1.x * b1.x + a1.y * b1.y)) + ((a2.x * b2.x + a2.y * b2.y) + (a3.x * b3.x + a3.y * b3.y)) + (a4.x * b4.x + a4.y * b4.y);
                    if (step == 8) { // exact norms once per sweep (the incremental update drifts)
                        al = ((a0.x * a0.x + a0.y * a0.y) + (a1.x * a1.x + a1.y * a1.y)) + ((a2.x * a2.x + a2.y * a2.y) + (a3.x * a3.x + a3.y * a3.y)) + (a4.x * a4.x + a4.y * a4.y);
                        be = ((b0.x * b0.x + b0.y * b0.y) + (b1.x * b1.x + b1.y * b1.y)) + ((b2.x * b2.x + b2.y * b2.y) + (b3.x * b3.x + b3.y * b3.y)) + (b4.x * b4.x + b4.y * b4.y);
                    } else {
                        al = L[src_an];
                        be = L[src_bn];
                    }
                    CVXW_SYNC();
                }
                ++sweeps;
                more = __any(coarse) && sweeps < o.jacobi_sweeps;
            } while (more);
        }
        total_sweeps += sweeps;
        cold = false;
        CVXW_PH(PH_JACOBI);
        // ---- Wp = sum_{lam > 0} lam v v^T, from (g, w g) with w = lam / lam'^2
        const double lpa = cvx::sqrt_fast(al), lpb = cvx::sqrt_fast(be);
        const double lama = lpa - sigma, lamb = lpb - sigma;
        const double wa = lama > 0 ? lama * cvx::rcp(al) : 0.0, wb = lamb > 0 ? lamb * cvx::rcp(be) : 0.0;
        if (o.warm_start) {
            L[L_VN + (2 * jk) * 10 + ji] = ca * cvx::rsqrt_(al);
            L[L_VN + (2 * jk + 1) * 10 + ji] = cb * cvx::rsqrt_(be);
        }
        L2[(L_Y + 2 * ((2 * jk) * 10 + ji)) / 2] = make_double2(ca, wa * ca);
        L2[(L_Y + 2 * ((2 * jk + 1) * 10 + ji)) / 2] = make_double2(cb, wb * cb);
        if (ji == 0) { // slot eigenvalue data for the top-eigenvector / rank decisions
            L[L_M + 2 * jk] = al;
            L[L_M + 2 * jk + 1] = be;
        }
        CVXW_SYNC();
        {   // all ten slots unconditionally (a slot without weight holds w g = 0): twenty independent LDS reads
            // in flight instead of a chain of branches, each waiting for its own pair of reads
            double acc0 = 0.0, acc1 = 0.0;
#pragma unroll
            for (int s = 0; s < 10; s += 2) {
                const double2 yi0 = L2[(L_Y + 2 * (s * 10 + ei)) / 2], yj0 = L2[(L_Y + 2 * (s * 10 + ej)) / 2];
                const double2 yi1 = L2[(L_Y + 2 * ((s + 1) * 10 + ei)) / 2], yj1 = L2[(L_Y + 2 * ((s + 1) * 10 + ej)) / 2];
                acc0 += yi0.y * yj0.x;
                acc1 += yi1.y * yj1.x;
            }
            Wp = acc0 + acc1;
        }
        }
        ++it;
        CVXW_PH(PH_WP);
        const bool check = it >= next_check;
        bool last = (it >= o.max_iters) || (fp_res < o.res_tol) || (it >= ipm_deadline);
        if (check || last) {
            const double retry_shift = (retries && retry_left > 0) ? o.dual_shift : 0.0;
            const bool refine = CVX_DUAL_REFINE_COMPILED && o.dual_refine && resume_full_in && retry_left > cvx::DUAL_RETRY_ATTEMPTS - cvx::DUAL_REFINE_ATTEMPTS; // (cvx::dual_refine_step: the first attempts of the phase behind a QUAD phase; where it pays and where not: solver_core.h at DUAL_REFINE_ATTEMPTS)
            --retry_left;
            // top eigenvector slot and the runner-up (wave-uniform)
            int smax = 0, s2nd = 0;
            double best = -1.0, second = -1.0;
#pragma unroll
            for (int s = 0; s < 10; ++s) {
                const double n2 = L[L_M + s];
                const bool b1 = n2 > best, b2 = !b1 && n2 > second;
                second = b1 ? best : (b2 ? n2 : second);
                s2nd = b1 ? smax : (b2 ? s : s2nd);
                best = b1 ? n2 : best;
                smax = b1 ? s : smax;
            }
            if (o.stall_from > 0 && it >= o.stall_from - 32) { // a Z that has settled at rank > 1 (cvx::rank_stalled): not tight, stop
                const double lam2 = cvx::sqrt_fast(second) - sigma;
                last = last || cvx::rank_stalled(it, lam2, lam2_prev, fp_res, o);
                lam2_prev = lam2;
            }
            int rank = 0;
#pragma unroll
            for (int s = 0; s < 10; ++s) rank += L[L_M + s] > (sigma + 1e-3) * (sigma + 1e-3); // eigenvalue > 1e-3, no roots
            // candidates (see cvx::solve_sdp): the top eigenvector; from iteration 6 on, with a comparable
            // second eigenvalue, the two poses of the top-2 eigenspace in closed form (cvx::twin_candidates)
            bool two = false;
            if (it >= 6) two = (cvx::sqrt_fast(second) - sigma) > 0.5 * (cvx::sqrt_fast(best) - sigma); // wave-uniform
            double Rc[9], pobj = 0, zSz = 0;
            {   // unit eigenvectors saved to LDS first: the certificate reuses the L_Y region
                const double il1 = cvx::rsqrt_(best), il2 = cvx::rsqrt_(second);
                if (lane < 10) {
                    L[L_V + lane] = L[L_Y + 2 * (smax * 10 + lane)] * il1;
                    L[L_V + 10 + lane] = L[L_Y + 2 * (s2nd * 10 + lane)] * il2;
                }
            }
            CVXW_SYNC();
            bool gap_ok = false, ambiguous = false;
            const double gap_tol = o.eps > 8e-13 * tr ? o.eps : 8e-13 * tr;
            if (!two) {
                double vloc[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) vloc[i] = L[L_V + i];
                // repeated checks mostly polish to the pose the previous check already had (see
                // cvx::solve_sdp): reuse it when the candidate rounds to that rotation (cvx::rounds_to)
                double d0, Rp[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) Rp[i] = L[L_M + 50 + i];
                const bool reuse = have_prev && reused < cvx::REUSE_MAX && cvx::rounds_to(vloc, Rp, d0);
                reused = reuse ? reused + 1 : 0;
                CVXW_PH(PH_TOPSEL);
                if (reuse) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) Rc[i] = Rp[i];
                    pobj = fprev;
                } else {
                    d0 = coop_round(vloc, Rc);
                    coop_polish(L, roles, Qs, Rc, pobj CVXW_PH_ARG);
                }
                CVXW_PH(PH_POLISH);
                const bool cok = coop_dual<VAR>(L, roles, Qs, W, Wp, Rc, d0, pobj, rho, delta, zSz, retry_shift, refine CVXW_PH_ARG);
                CVXW_PH(PH_DUAL);
                gap_ok = cok && (tr * (fabs(zSz) + 4.0 * delta) <= gap_tol);
                have_prev = d0 > 0 && (pobj == pobj);
                fprev = pobj;
                CVXW_SYNC();
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) L[L_M + 50 + i] = Rc[i];
                }
            } else {
                have_prev = false; // the pose of the last single-candidate check is stale after a spell here
                // z+- = (c1 +- d1) v1 + (c2 +- d2) v2: last entry 1, squared norm 4
                const double ta = L[L_V + 9], tb = L[L_V + 19], n2 = ta * ta + tb * tb;
                const double inv = cvx::rcp(n2), rn = cvx::rsqrt_(n2), rad = 4.0 - inv;
                const double sq = rad > 0 ? cvx::sqrt_fast(rad) : 0.0;
                const double c1 = ta * inv, c2 = tb * inv, d1 = -tb * rn * sq, d2 = ta * rn * sq;
                double zc[10], fp, fm;
#pragma unroll
                for (int i = 0; i < 10; ++i) zc[i] = (c1 + d1) * L[L_V + i] + (c2 + d2) * L[L_V + 10 + i];
                // each twin may take over the rotation its slot polished at the previous check (cvx::polish_or_reuse:
                // within ~0.05 rad, a fresh polish every ninth check) -- near-ambiguous problems are the slowest
                // of every batch and would otherwise run two polar + Newton polishes per check
                const bool may = tw_reused < 8;
                double dp, dm, Rold[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) Rold[i] = L[L_M + 30 + i];
                if (may && have_tp && cvx::rounds_to(zc, Rold, dp, 0.01)) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) Rc[i] = Rold[i];
                    fp = f_tp;
                } else {
                    dp = coop_round(zc, Rc);
                    coop_polish(L, roles, Qs, Rc, fp CVXW_PH_ARG);
                }
                CVXW_SYNC();
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) L[L_M + 30 + i] = Rc[i];
                }
#pragma unroll
                for (int i = 0; i < 10; ++i) zc[i] = (c1 - d1) * L[L_V + i] + (c2 - d2) * L[L_V + 10 + i];
#pragma unroll
                for (int i = 0; i < 9; ++i) Rold[i] = L[L_M + 40 + i];
                CVXW_SYNC();
                if (may && have_tm && cvx::rounds_to(zc, Rold, dm, 0.01)) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) Rc[i] = Rold[i];
                    fm = f_tm;
                } else {
                    dm = coop_round(zc, Rc);
                    coop_polish(L, roles, Qs, Rc, fm CVXW_PH_ARG);
                }
                CVXW_SYNC();
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) L[L_M + 40 + i] = Rc[i];
                }
                CVXW_SYNC();
                tw_reused = may ? tw_reused + 1 : 0;
                f_tp = fp; f_tm = fm;
                have_tp = dp > 0 && (fp == fp);
                have_tm = dm > 0 && (fm == fm);
                double trc = 0;
                bool fin = (fp == fp) && (fm == fm);
#pragma unroll
                for (int i = 0; i < 9; ++i) { const double rp = L[L_M + 30 + i]; trc += rp * Rc[i]; fin = fin && (rp == rp) && (Rc[i] == Rc[i]); }
                const bool twins = fin && dp > 0 && dm > 0 && fabs(fp - fm) <= gap_tol * itr && trc < 2.9;
                // equal-cost twins: the pair is accepted only if z+ passes the dual test (then both are
                // global optima: the problem is exactly two-fold ambiguous -> rank 2, like the reference)
                const bool take_m = !twins && dm > 0 && (fm == fm) && (!(dp > 0) || !(fp == fp) || fm < fp);
                if (!take_m) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) Rc[i] = L[L_M + 30 + i];
                }
                pobj = take_m ? fm : fp;
                const bool cok = coop_dual<VAR>(L, roles, Qs, W, Wp, Rc, take_m ? dm : dp, pobj, rho, delta, zSz, retry_shift, refine CVXW_PH_ARG);
                const bool ok = cok && (tr * (fabs(zSz) + 4.0 * delta) <= gap_tol);
                ambiguous = twins && ok;
                gap_ok = !twins && ok;
            }
            CVXW_SYNC();
            if (lane == 0) L[L_M + 29] = ambiguous ? 1.0 : 0.0;
            if (ambiguous) {
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) L[L_M + 16 + i] = Rc[i];
                    L[L_M + 25] = tr * pobj; L[L_M + 26] = tr * (pobj - zSz - 4.0 * delta); // certified pair: its lower bound
                    L[L_M + 27] = (double)cvx::ST_RANK_GT1; L[L_M + 28] = 2.0;
                }
            }
            if (gap_ok && !ambiguous) {
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) L[L_M + 16 + i] = Rc[i];
                    L[L_M + 25] = tr * pobj; L[L_M + 26] = tr * (pobj - zSz - 4.0 * delta);
                    L[L_M + 27] = (double)cvx::ST_CERTIFIED; L[L_M + 28] = 1.0;
                }
            } else if (last && !ambiguous) {
                // cold path: the reference's recovery from the uncertified iterate (cvx::fallback_pose):
                // R = U V^T of the rank-1 ratio (no determinant fix), cost = r^T Q r via the LDS copy of Qs;
                // rank > 1: the better of the two rank-2 candidates of the top-2 eigenspace instead (never NaN
                // while Z is finite, see cvx::fallback_pose)
                coop_store_qf(L, roles, Qs); // (a reused pose skipped the polish that would have stored it)
                auto rounded_cost = [&](const double *z, double *Rr, bool &fin) { // cvx::rounded_cost, every lane
                    double M0[9];
                    const double iv = 1.0 / z[9];
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) M0[i * 3 + j] = z[3 * j + i] * iv;
                    cvx::polar3(M0, Rr, 12);
                    CVXW_SYNC();
                    if (lane == 0) {
#pragma unroll
                        for (int i = 0; i < 3; ++i)
#pragma unroll
                            for (int j = 0; j < 3; ++j) L[C_XV + 3 * j + i] = Rr[i * 3 + j];
                        L[C_XV + 9] = 1.0;
                    }
                    CVXW_SYNC();
                    double part = 0.0;
                    if (lane < 9) part = L[C_XV + lane] * dot10(reinterpret_cast<double2 *>(L) + (C_QF + lane * 10) / 2, reinterpret_cast<double2 *>(L) + C_XV / 2);
                    const double c = wave_sum(part);
                    fin = (c == c);
#pragma unroll
                    for (int i = 0; i < 9; ++i) fin = fin && (Rr[i] == Rr[i]);
                    return c;
                };
                double v1[10], v2[10], Rf[9];
#pragma unroll
                for (int i = 0; i < 10; ++i) { v1[i] = L[L_V + i]; v2[i] = L[L_V + 10 + i]; }
                bool okf;
                double fc = rounded_cost(v1, Rf, okf);
                if (rank > 1) { // wave-uniform
                    double zp[10], zm[10], Rp[9], Rm[9];
                    cvx::twin_candidates(v1, v2, zp, zm);
                    bool okp, okm;
                    const double fp = rounded_cost(zp, Rp, okp), fm = rounded_cost(zm, Rm, okm);
                    const bool pp = okp && cvx::det3(Rp) > 0, pm = okm && cvx::det3(Rm) > 0;
                    const bool take_m = okm && (!okp || (pm && !pp) || (pm == pp && fm < fp));
                    if (okp || okm) {
#pragma unroll
                        for (int i = 0; i < 9; ++i) Rf[i] = take_m ? Rm[i] : Rp[i];
                        fc = take_m ? fm : fp;
                        if (take_m ? pm : pp) { // wave-uniform: polish a proper rotation (cvx::fallback_pose)
                            double Rq[9], fq;
#pragma unroll
                            for (int i = 0; i < 9; ++i) Rq[i] = Rf[i];
                            CVXW_SYNC();
                            coop_polish(L, roles, Qs, Rq, fq CVXW_PH_ARG);
                            bool fin = (fq == fq);
#pragma unroll
                            for (int i = 0; i < 9; ++i) fin = fin && (Rq[i] == Rq[i]);
                            if (fin) {
#pragma unroll
                                for (int i = 0; i < 9; ++i) Rf[i] = Rq[i];
                                fc = fq;
                            }
                        }
                    }
                }
                const int fst = rank > 1 ? cvx::ST_RANK_GT1
                                         : (!okf ? cvx::ST_NONFINITE : (rank != 1 ? cvx::ST_RANK_GT1 : (cvx::det3(Rf) < 0 ? cvx::ST_REFLECTION : cvx::ST_UNCERTIFIED)));
                if (lane == 0) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) L[L_M + 16 + i] = Rf[i];
                    L[L_M + 25] = tr * fc; L[L_M + 26] = NAN;
                    L[L_M + 27] = (double)fst; L[L_M + 28] = (double)rank;
                }
            }
            if (lane == 0) L[L_M + 15] = (gap_ok && !ambiguous) ? 1.0 : 0.0;
            CVXW_SYNC();
            certified = L[L_M + 15] != 0.0;
            next_check = cvx::next_check_after(it, o);
            if (certified || last || ambiguous) {
                status = (int)L[L_M + 27];
                rank_out = (int)L[L_M + 28];
                done = true;
            }
            CVXW_PH(PH_CHECK_TAIL);
        }
        if (!done && it == o.tail_from) { // smaller penalty for the slow tail; keeps the dual: Wm scales by rho / rho_tail
            W = Wp + (W - Wp) * (rho / o.rho_tail);
            rho = o.rho_tail;
            irho = 1.0 / rho;
        }
        if (!done && o.adapt_every > 0 && it >= o.adapt_from && (it - o.adapt_from) % o.adapt_every == 0) { // wave-uniform
            // residual balancing of the penalty for the slow tail (cvx::solve_sdp has the rationale and the numbers)
            const double Pa = coop_proj<VAR>(L, roles, Wp, 1.0);
            const double Td = coop_proj<VAR>(L, roles, rho * (Wp - W) - (ej < 9 ? Qs : 0.0), 0.0);
            const double rp2 = wave_sum(wgt * (Pa - Wp) * (Pa - Wp)), rd2 = wave_sum(wgt * Td * Td);
            const double rn = cvx::adapted_rho(rho, rp2, rd2, o);
            if (rn != rho) {
                W = Wp + (W - Wp) * (rho / rn);
                rho = rn;
                irho = 1.0 / rho;
            }
        }
        if (!done) {
            // X = Pi_aff(2 Wp - W - Qs / rho);  W <- W + alpha (X - Wp)
            const double Xn = coop_proj<VAR>(L, roles, 2.0 * Wp - W - irho * Qs, 1.0);
            const double dd = Xn - Wp;
            W += o.alpha * dd;
            fp_res = cvx::sqrt_fast(wave_sum(wgt * dd * dd));
            if (!(fp_res == fp_res)) { status = cvx::ST_NONFINITE; done = true; }
            CVXW_SYNC();
            CVXW_PH(PH_UPDATE);
        }
    }
    if (!done) { // wave-uniform: a slow one
        if (IPM) {
            L[I_QS + lane] = (lane < 55 && ej < 9) ? Qs : 0.0;
            CVXW_SYNC();
            it_io = it; sweeps_io = total_sweeps;
            return true;
        }
        if (a.rq_ws) { // split path: the interior-point kernel runs nothing but the solve -- it finds the cost (solver's frame) in the slot
            double *slot = a.rq_ws + (int64_t)b * a.rq_stride;
            if (lane < 55) __hip_atomic_store(slot + RS_W + lane, ej < 9 ? Qs : 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane == 0) __hip_atomic_store(slot + RS_IT, (double)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        }
        if (lane == 0) {
            a.status[b] = cvx::ST_PENDING + (it << 8);
            if (a.work) a.work[2 * b + 1] = total_sweeps;
            const int q = atomicAdd(a.rq_count, 1);
            a.rq_entries[q] = (int32_t)b;
        }
        return false;
    }

    // ---------------------------------------------------------------- outputs
    if (canon) { // back to the caller's frame: R = R' U^T for the pose and the twins, Z = Pt^T Z' Pt for an uncertified Z
        CVXW_SYNC();
        if (lane < 3) {
            const int off = lane == 0 ? 16 : (lane == 1 ? 30 : 40);
            double Rn[9];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    Rn[i * 3 + j] = L[L_M + off + i * 3] * L[L_U + j * 3] + L[L_M + off + i * 3 + 1] * L[L_U + j * 3 + 1] + L[L_M + off + i * 3 + 2] * L[L_U + j * 3 + 2];
#pragma unroll
            for (int i = 0; i < 9; ++i) L[L_M + off + i] = Rn[i];
        }
        L[L_X + el] = Wp;
        CVXW_SYNC();
        double acc = 0.0;
        if (ej < 9) {
            const int bi = ei / 3, x = ei % 3, bj = ej / 3, y = ej % 3;
            for (int k = 0; k < 3; ++k)
                for (int l = 0; l < 3; ++l) acc += L[L_U + bi * 3 + k] * L[L_U + bj * 3 + l] * L[L_X + cvx::sidx(3 * k + x, 3 * l + y)];
        } else if (ei < 9) {
            const int bi = ei / 3, x = ei % 3;
            for (int k = 0; k < 3; ++k) acc += L[L_U + bi * 3 + k] * L[L_X + cvx::sidx(3 * k + x, 9)];
        } else acc = L[L_X + 54];
        Wp = acc;
        CVXW_SYNC();
    }
    const bool have_pose = finite && status != cvx::ST_NONFINITE;
    if (lane < 9) a.R[b * 9 + lane] = have_pose ? L[L_M + 16 + lane] : NAN;
    if (lane < 3) { // t = -B r (cvxpnpl.py:513), r = vec_colmajor(R)
        double tv = 0;
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3)
#pragma unroll
            for (int r3 = 0; r3 < 3; ++r3) tv += L[L_B + lane * 9 + 3 * c3 + r3] * L[L_M + 16 + r3 * 3 + c3];
        a.t[b * 3 + lane] = have_pose ? -tv : NAN;
    }
    if (lane == 0) {
        a.status[b] = status;
        if (a.iters) a.iters[b] = it;
        if (a.cost) { a.cost[2 * b] = have_pose ? L[L_M + 25] : NAN; a.cost[2 * b + 1] = have_pose ? L[L_M + 26] : NAN; }
        if (a.work) { a.work[2 * b] = rank_out; a.work[2 * b + 1] = total_sweeps; }
    }
    if (a.Z && lane < 55) {
        double zv;
        if (!have_pose) zv = NAN;
        else if (certified) { // Z = z z^T with z = [vec_colmajor(R); 1]
            const double zi = ei == 9 ? 1.0 : L[L_M + 16 + (ei % 3) * 3 + ei / 3];
            const double zj = ej == 9 ? 1.0 : L[L_M + 16 + (ej % 3) * 3 + ej / 3];
            zv = zi * zj;
        } else if (L[L_M + 29] != 0.0) { // exactly two-fold ambiguous: Z = (z+ z+^T + z- z-^T) / 2
            const double ai = ei == 9 ? 1.0 : L[L_M + 30 + (ei % 3) * 3 + ei / 3], aj = ej == 9 ? 1.0 : L[L_M + 30 + (ej % 3) * 3 + ej / 3];
            const double bi = ei == 9 ? 1.0 : L[L_M + 40 + (ei % 3) * 3 + ei / 3], bj = ej == 9 ? 1.0 : L[L_M + 40 + (ej % 3) * 3 + ej / 3];
            zv = 0.5 * (ai * aj + bi * bj);
        } else zv = Wp;
        a.Z[b * 55 + lane] = zv;
    }
    CVXW_PH(PH_OUTPUT);
    CVXW_PH_FLUSH();
    return false;
}

// Solve problem b with the wavefront that calls this.  IPM (cvxw::rescue_wave_kernel only; L then has LDSW_IPM doubles): a
// problem that comes from the rescue queue (resume == nullptr, it0 iterations behind it) or reaches opts.rescue_from
// iterations here goes through the interior-point solve of ipm_wave.h (~12 second-order
// iterations whatever the conditioning), then the first-order iteration from W = Z - S / rho, whose positive part is Z and
// whose dual hint rho (W+ - W) is S: the attempt of the next iteration certifies (or finds the twin pair, or reports the rank)
// from the interior-point solution, with the rounding, polish, certificate and recovery every other problem goes through.
template <int VAR = cvx::VAR_FULL, bool IPM = false>
__device__ __forceinline__ void solve_one_wave(const WaveArgs &a, const cvx::Opts &o, const int64_t b, double *L, const double *resume, const bool resume_full = false,
                                               const int it0 = 0, const int sweeps0 = 0)
{
    int it = it0, sweeps = sweeps0;
    if constexpr (!IPM) {
        (void)solve_pass<VAR, false>(a, o, b, L, resume, resume_full, false, it, sweeps);
    } else {
#ifdef CVXW_IPM_CLOCK // diagnostic build (tools/ipm_clock.py): 100 MHz ticks of the three stages into cost[2b], cost[2b+1], t[3b]
    const long long c0 = wall_clock64();
#endif
    if (solve_pass<VAR, true>(a, o, b, L, resume, resume_full, false, it, sweeps)) {
        const int lane = threadIdx.x & 63;
        const unsigned lw = kLanePack.w[lane];
        const int ei = (int)(lw & 15), ej = (int)((lw >> 4) & 15);
        double gap;
#ifdef CVXW_IPM_CLOCK
        const long long c1 = wall_clock64();
#endif
#ifndef CVXW_IPM_TOL
#define CVXW_IPM_TOL 1e-10
#endif
        const int nit = coop_ipm<VAR>(L, lane, L[I_QS + lane], ei, ej, CVXW_IPM_TOL, 40, &gap);
#ifdef CVXW_IPM_CLOCK
        const long long c2 = wall_clock64();
#endif
        const double w = L[I_Z + ei * 10 + ej] - L[I_S + ei * 10 + ej] / o.rho;
        CVXW_SYNC();
        if (lane < 55) L[I_W + lane] = w;
        CVXW_SYNC();
        it += nit;
        (void)solve_pass<VAR, true>(a, o, b, L, resume, resume_full, true, it, sweeps);
#ifdef CVXW_IPM_CLOCK
        if (lane == 0 && a.cost) { a.cost[2 * b] = (double)(c2 - c1) + 1e-3 * nit; a.cost[2 * b + 1] = (double)(wall_clock64() - c2); a.t[3 * b] = (double)(c1 - c0); for (int k = 0; k < 9; ++k) a.R[9 * b + k] = L[I_COL + 100 + k]; }
#endif
    }
    }
}

template <int VAR>
__global__ void __launch_bounds__(64 * WPB, 2) solve_wave_kernel(WaveArgs a, cvx::Opts o)
{
    __shared__ __attribute__((aligned(16))) double lds_all[WPB][LDSW];
    const int wib = threadIdx.x >> 6;
    const int64_t b = (int64_t)blockIdx.x * WPB + wib;
    if (b >= a.batch) return; // wave-uniform
    solve_one_wave<VAR>(a, o, b, lds_all[wib], nullptr);
}

// Second phase of the hybrid schedules: the problems the first kernel parked, one wavefront each.  The queue is
// self-cleaning: entries[] is -1 wherever nothing is queued and the three counters (count[0]: positions filled by the first
// kernel, count[1]: positions drawn, count[2]: blocks that have left) are zero between launches.  The first kernel appends problem
// indices at atomicAdd(count[0]) positions; resume block i owns position i and, when that is done, draws further positions from
// count[1] (dynamic: a slow problem does not hold up a fixed share of the queue) until it meets a -1; it puts -1 back over every
// entry it consumes, and the last block to leave zeroes the counters.  A block whose own position is empty leaves at once without
// touching anything -- with an empty queue (most launches) that is every block.  So every launch leaves the queue as it found it --
// no host-side bookkeeping, no memset per launch, nothing a hipGraph replay could desynchronise -- and every index is range
// checked, so a corrupted workspace cannot turn into an out-of-bounds write.  entries[] has RESUME_GRID_MAX spare slots.
#ifndef CVXW_OCC_RESUME
#define CVXW_OCC_RESUME 2
#endif
#ifndef CVXW_OCC_RESCUE
#define CVXW_OCC_RESCUE 2
#endif
constexpr int RESUME_GRID_MAX = 1024 * (CVXW_OCC_RESUME > CVXW_OCC_RESCUE ? CVXW_OCC_RESUME : CVXW_OCC_RESCUE); // one block per resident wavefront slot: more blocks only add launch time to the (usual) empty-queue case
// A block whose first queue slot is empty (every block of most launches) leaves after one load: the arguments of the
// solve are read from the kernarg segment only behind that test, so that nothing is live -- and nothing spilled -- before
// it (with by-value arguments the 8192 mostly idle wavefronts of a launch wrote 126 MB of spilled registers at kernel
// entry; as a separate non-inlined function the body paid the calling convention instead: ~185 register saves and
// restores per call, 0.8 GB per 125 k launch).
struct ResumeArgs {
    WaveArgs a;
    cvx::Opts o;
    int32_t *count_p, *entries;
    const double *ws;
    int ws_stride, ws_full; // doubles per parked problem (RS_LANE / RS_FULL) and whether the slots carry Q, B, V
    // rescue_wave_kernel only: the blocks from grid1 on serve this second queue (parked problems; null: none)
    int32_t *count2, *entries2;
    int grid1;
};
typedef const __attribute__((address_space(4))) ResumeArgs *ResumeArgsPtr;

template <bool IPM, int VAR = cvx::VAR_FULL>
__device__ __forceinline__ void resume_body(ResumeArgsPtr kp, int32_t first, double *lds)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const WaveArgs a = kp->a;
    const cvx::Opts o = kp->o;
    // (rescue_wave_kernel: two queues, each with its own range of blocks)
    const bool second = IPM && (int)blockIdx.x >= kp->grid1;
    const int bid = second ? (int)blockIdx.x - kp->grid1 : (int)blockIdx.x;
    const int gsz = !IPM ? (int)gridDim.x : (second ? (int)gridDim.x - kp->grid1 : kp->grid1);
    int32_t *entries = second ? kp->entries2 : kp->entries;
    int32_t *count_p = second ? kp->count2 : kp->count_p;
    const int pushed = __hip_atomic_load(count_p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (stable until the last block resets it)
    const double *ws = kp->ws;
    const int stride = kp->ws_stride;
    const bool full = kp->ws_full != 0;
    int32_t b = first;
    for (int64_t q = bid;;) { // wave-uniform
        if ((threadIdx.x & 63) == 0) entries[q] = -1;
        if (b < a.batch) {
            if constexpr (IPM) {
                if (second) solve_one_wave<VAR, true>(a, o, b, lds, ws + (int64_t)b * stride, full); // a parked problem
                else { // rescue queue: nothing parked but the iteration count, in the status word
                    const int st = a.status[b];
                    solve_one_wave<VAR, true>(a, o, b, lds, nullptr, false, st >> 8, a.work ? a.work[2 * b + 1] : 0);
                }
            } else solve_one_wave<VAR>(a, o, b, lds, ws + (int64_t)b * stride, full);
            CVXW_SYNC();
        }
        // The next position nobody has taken yet: blocks own the positions below gridDim.x by index and DRAW the ones behind them
        // from count_p[1] -- a wavefront that got a 40-iteration problem must not also own every 2048th entry behind it.
        int pn = 0;
        if ((threadIdx.x & 63) == 0) pn = atomicAdd(count_p + 1, 1);
        q = (int64_t)gsz + __builtin_amdgcn_readfirstlane(pn);
        b = q < a.batch + RESUME_GRID_MAX ? entries[q] : -1;
        if (b < 0) break; // an empty position: the queue is exhausted -- every drawing block ends with exactly one such draw
    }
    // The last block to leave puts the counters back to zero for the next launch (pushed = positions the first kernel filled; the
    // blocks that draw are the ones whose own position was filled: min(pushed, gridDim.x) of them).
    if ((threadIdx.x & 63) == 0) {
        const int drawing = pushed < gsz ? pushed : gsz;
        if (atomicAdd(count_p + 2, 1) == drawing - 1) {
            __hip_atomic_store(count_p, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(count_p + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(count_p + 2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#endif
}

__global__ void __launch_bounds__(64, CVXW_OCC_RESUME) resume_wave_kernel(ResumeArgs k)
{
    __shared__ __attribute__((aligned(16))) double lds_all[LDSW];
    const int32_t first = k.entries[blockIdx.x];
    if (first < 0) return;
    resume_body<false>((ResumeArgsPtr)__builtin_amdgcn_kernarg_segment_ptr(), first, lds_all);
}
// the same for the 16-equality variant (quad schedule of cvxpnpl_solve_cost_batch / solve_batch with opts.variant = RC)
__global__ void __launch_bounds__(64, CVXW_OCC_RESUME) resume_wave_kernel_rc(ResumeArgs k)
{
    __shared__ __attribute__((aligned(16))) double lds_all[LDSW];
    const int32_t first = k.entries[blockIdx.x];
    if (first < 0) return;
    resume_body<false, cvx::VAR_RC>((ResumeArgsPtr)__builtin_amdgcn_kernarg_segment_ptr(), first, lds_all);
}

// Last phase of a solve with opts.rescue_from in force: the problems the other kernels put on the rescue queue (k.count_p =
// rq_count, k.entries = rq_entries; same self-cleaning queue discipline), one wavefront each, through the interior-point solve.
// In the quad layout the same launch also takes the place of resume_wave_kernel: its blocks from k.grid1 on serve the resume queue
// (k.count2, k.entries2, k.ws), and a parked problem that reaches opts.rescue_from goes through the interior-point solve right here.
__global__ void __launch_bounds__(64, CVXW_OCC_RESCUE) rescue_wave_kernel(ResumeArgs k)
{
    __shared__ __attribute__((aligned(16))) double lds_all[LDSW_IPM];
    const int32_t first = (int)blockIdx.x < k.grid1 ? k.entries[blockIdx.x] : k.entries2[(int)blockIdx.x - k.grid1];
    if (first < 0) return;
    resume_body<true>((ResumeArgsPtr)__builtin_amdgcn_kernarg_segment_ptr(), first, lds_all);
}
// the same for the 16-equality variant: the interior-point solve on its 16 rows (cvx::ipm_rows)
__global__ void __launch_bounds__(64, CVXW_OCC_RESCUE) rescue_wave_kernel_rc(ResumeArgs k)
{
    __shared__ __attribute__((aligned(16))) double lds_all[LDSW_IPM];
    const int32_t first = (int)blockIdx.x < k.grid1 ? k.entries[blockIdx.x] : k.entries2[(int)blockIdx.x - k.grid1];
    if (first < 0) return;
    resume_body<true, cvx::VAR_RC>((ResumeArgsPtr)__builtin_amdgcn_kernarg_segment_ptr(), first, lds_all);
}


} // namespace cvxw
