// NOT BUILT -- the record of a round-3 experiment (profiles/r03/lane4_experiment.txt).  It plugs into cvxl::lane_phase through an eigen-engine template
// parameter (eig_unit / eig_load_warm_f32 / eig_solve / eig_pospart / eig_top found by ADL) and a `writer` flag that the experiment added to
// lane_core.h and that were taken out again with it.
// lane4_kernel.h -- four lanes per problem: the lane phase of lane_core.h with its eigen-solve divided between the lanes of a DPP quad.
//
// The lane-per-problem kernel (solve_lane2_kernel) is the cheapest layout per problem, but a wavefront of 64 problems needs ~150 us for
// its six iterations and a launch needs ~20 000 problems before every SIMD has one; the quad kernel (16 lanes per problem) has the
// latency but replicates most of its arithmetic sixteen times.  Two thirds of the lane phase are the Jacobi sweeps of the PSD
// projection, and those divide cleanly: here every problem belongs to the four lanes of a DPP quad, each lane owns three of twelve
// column slots (ten columns + two empty slots), rotations inside a lane are local, rotations between lanes read the partner's column
// through quad_perm DPP operands (full-rate VALU modifiers: no LDS, no ds_bpermute, no barrier).  Everything else of cvxl::lane_phase --
// assembly, the iterate W, the affine projection, the certificate -- is computed identically by all four lanes (same inputs, same
// instructions, bit-identical results; the one reduction, W+ = sum of the lanes' partial sums, is a symmetric butterfly), so the
// register-budgeted code of lane_core.h runs unchanged.  16 problems per wavefront: ~4x the wavefronts and ~1/3 of the latency of the
// lane kernel at twice its instructions per problem (a quarter of the quad kernel's).
#pragma once
#include <hip/hip_runtime.h>

#include "lane_core.h"
#include "quad_kernel.h" // cvxq::pair_cs

namespace cvxl {

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)

constexpr int Q_X1 = 0xB1, Q_X2 = 0x4E, Q_X3 = 0x1B; // quad_perm [1,0,3,2], [2,3,0,1], [3,2,1,0]: lane ^ 1, ^ 2, ^ 3

template <int CTRL>
__device__ __forceinline__ float qdpp(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ double qdppd(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// Column slots: lane ql of a quad owns columns ql, ql + 4 and ql + 8 (slots 0..2); slots 10 and 11 (lanes 2, 3, slot 2) are empty:
// zero columns with zero norm, which every rotation leaves alone.
struct EigQ {
    float g[3][10]; // g[s][i]: slot s, row i -- after an eigen-solve lam'_j v_j like cvx::EigF
    float n2[3];
    double sigma;
    int ql;
};
__device__ __forceinline__ bool slot_real(int ql, int s) { return ql + 4 * s < 10; }

__device__ __forceinline__ void eig_unit(EigQ &e)
{
    e.ql = (int)(threadIdx.x & 3);
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int col = e.ql + 4 * s;
#pragma unroll
        for (int i = 0; i < 10; ++i) e.g[s][i] = (i == col) ? 1.0f : 0.0f; // (col >= 10: a zero column)
        e.n2[s] = col < 10 ? 1.0f : 0.0f;
    }
    e.sigma = 0.0;
}

// G = (W + sigma I) V for this lane's columns (cvxl::eig_load_warm_f32)
__device__ __forceinline__ void eig_load_warm_f32(EigQ &e, const double *W)
{
    double fro = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i)
#pragma unroll
        for (int j = i; j < 10; ++j) fro += (i == j ? 1.0 : 2.0) * W[sidx(i, j)] * W[sidx(i, j)];
    e.sigma = 1.5 * sqrt_fast(fro) + 1e-300;
    float Wf[55];
#pragma unroll
    for (int k = 0; k < 55; ++k) Wf[k] = (float)W[k];
    const float sg = (float)e.sigma;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const float il_ = slot_real(e.ql, s) ? __builtin_amdgcn_rsqf(e.n2[s]) : 0.0f;
        float v[10], acc[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) v[i] = e.g[s][i] * il_;
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            float a = sg * v[i];
#pragma unroll
            for (int m = 0; m < 10; ++m) a = fmaf(Wf[sidx(i, m)], v[m], a);
            acc[i] = a;
        }
#pragma unroll
        for (int i = 0; i < 10; ++i) e.g[s][i] = acc[i];
    }
}

__device__ __forceinline__ void eig_norms(EigQ &e)
{
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        float a = 0.0f;
#pragma unroll
        for (int i = 0; i < 10; ++i) a = fmaf(e.g[s][i], e.g[s][i], a);
        e.n2[s] = a;
    }
}

// rotation of two columns of this lane (cvx::eig_step5 for one pair)
template <int A, int B>
__device__ __forceinline__ void step_local(EigQ &e, float tol2, bool active, bool &coarse)
{
    float gam = 0.0f;
#pragma unroll
    for (int i = 0; i < 10; ++i) gam = fmaf(e.g[A][i], e.g[B][i], gam);
    const float al = e.n2[A], be = e.n2[B], g2 = gam * gam, ab = al * be;
    coarse |= g2 > tol2 * ab;
    float c, s, t;
    jacobi_cs(al, be, gam, active && g2 > 1e-30f * ab, c, s, t);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const float gp = e.g[A][i], gq = e.g[B][i];
        e.g[A][i] = c * gp - s * gq;
        e.g[B][i] = s * gp + c * gq;
    }
    e.n2[A] = al - t * gam;
    e.n2[B] = be + t * gam;
}

// Rotations between this lane and lane ^ X (CTRL = its quad_perm).  A == B: the pair (slot A here, slot A there) -- both lanes
// compute the same rotation from their side (cvxq::pair_cs: the partner sees d -> -d, t -> -t) and update their own column.
// A != B: two pairs at once, (slot A here, slot B there) and (slot A there, slot B here); a lane computes the rotation of the pair
// whose first column it owns and takes the parameters of the other one from its partner.  All updates from the old columns.
template <int A, int B, int CTRL, int X>
__device__ __forceinline__ void step_cross(EigQ &e, float tol2, bool active, bool &coarse)
{
    float oth[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) oth[i] = qdpp<CTRL>(e.g[B][i]);
    float gam = 0.0f;
#pragma unroll
    for (int i = 0; i < 10; ++i) gam = fmaf(e.g[A][i], oth[i], gam);
    const float al = e.n2[A], be = qdpp<CTRL>(e.n2[B]);
    const float g2 = gam * gam, ab = al * be;
    coarse |= g2 > tol2 * ab;
    float c, s, t;
    cvxq::pair_cs(be - al, gam, active && g2 > 1e-30f * ab, e.ql > (e.ql ^ X), c, s, t); // own' = c own - s other
    if (A != B) {
        const float cf = qdpp<CTRL>(c), sf = qdpp<CTRL>(s), tf = qdpp<CTRL>(t), gf = qdpp<CTRL>(gam);
        float nb[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) nb[i] = fmaf(sf, qdpp<CTRL>(e.g[A][i]), cf * e.g[B][i]); // q' = s p + c q of the partner's pair
#pragma unroll
        for (int i = 0; i < 10; ++i) oth[i] = fmaf(c, e.g[A][i], -(s * oth[i]));
#pragma unroll
        for (int i = 0; i < 10; ++i) { e.g[A][i] = oth[i]; e.g[B][i] = nb[i]; }
        e.n2[B] += tf * gf;
    } else {
#pragma unroll
        for (int i = 0; i < 10; ++i) e.g[A][i] = fmaf(c, e.g[A][i], -(s * oth[i]));
    }
    e.n2[A] = al - t * gam;
}

template <int CTRL, int X>
__device__ __forceinline__ void steps_with(EigQ &e, float tol2, bool active, bool &coarse)
{
    // (scheduling barriers: without them the scheduler overlaps the steps of a sweep and spills inside the loop)
    step_cross<0, 0, CTRL, X>(e, tol2, active, coarse); __builtin_amdgcn_sched_barrier(0);
    step_cross<1, 1, CTRL, X>(e, tol2, active, coarse); __builtin_amdgcn_sched_barrier(0);
    step_cross<2, 2, CTRL, X>(e, tol2, active, coarse); __builtin_amdgcn_sched_barrier(0);
    step_cross<0, 1, CTRL, X>(e, tol2, active, coarse); __builtin_amdgcn_sched_barrier(0);
    step_cross<0, 2, CTRL, X>(e, tol2, active, coarse); __builtin_amdgcn_sched_barrier(0);
    step_cross<1, 2, CTRL, X>(e, tol2, active, coarse); __builtin_amdgcn_sched_barrier(0);
}

// One-sided Jacobi until the largest squared cosine met in a sweep is below tol2 (cvx::eig_solve); a sweep is 21 steps: three
// inside the lanes, six with each of the three partners -- every pair of the twelve slots exactly once.  Returns this problem's sweeps.
__device__ __forceinline__ int eig_solve(EigQ &e, int max_sweeps, double tol2d)
{
    const float tol2 = (float)tol2d;
    const int q0 = (int)(threadIdx.x & 63) & ~3;
    int sweeps = 0;
    bool active = true;
    do {
        eig_norms(e);
        bool coarse = false;
        step_local<0, 1>(e, tol2, active, coarse); __builtin_amdgcn_sched_barrier(0);
        step_local<0, 2>(e, tol2, active, coarse); __builtin_amdgcn_sched_barrier(0);
        step_local<1, 2>(e, tol2, active, coarse); __builtin_amdgcn_sched_barrier(0);
        steps_with<Q_X1, 1>(e, tol2, active, coarse);
        steps_with<Q_X2, 2>(e, tol2, active, coarse);
        steps_with<Q_X3, 3>(e, tol2, active, coarse);
        const bool more = ((__ballot(coarse && active) >> q0) & 0xFull) != 0; // any lane of this quad
        if (active) ++sweeps;
        active = active && more && sweeps < max_sweeps;
    } while (__any(active));
    eig_norms(e);
    return sweeps;
}

// Wp = sum_{lam_j > 0} lam_j v_j v_j^T: partial sums over this lane's columns, then the sum over the quad (a symmetric butterfly:
// the four lanes end with bit-identical values)
__device__ __forceinline__ void eig_pospart(const EigQ &e, double *Wp)
{
#pragma unroll
    for (int i = 0; i < 55; ++i) Wp[i] = 0.0;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const double n2 = (double)e.n2[s];
        const double lam = sqrt_fast(n2) - e.sigma;
        const double w = (slot_real(e.ql, s) && lam > 0) ? lam / n2 : 0.0;
        double g[10];
#pragma unroll
        for (int i = 0; i < 10; ++i) g[i] = (double)e.g[s][i];
#pragma unroll
        for (int i = 0; i < 10; ++i) {
            const double wg = w * g[i];
#pragma unroll
            for (int k = i; k < 10; ++k) Wp[sidx(i, k)] += wg * g[k];
        }
    }
#pragma unroll
    for (int i = 0; i < 55; ++i) {
        double x = Wp[i];
        x += qdppd<Q_X1>(x);
        x += qdppd<Q_X2>(x);
        Wp[i] = x;
    }
}

// unit eigenvector of the largest eigenvalue: the owning lane's column, to all four
__device__ __forceinline__ void eig_top(const EigQ &e, double *vt)
{
    float best = -1.0f;
    int bs = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const float ns = e.n2[s];
        const bool b1 = slot_real(e.ql, s) && ns > best;
        best = b1 ? ns : best;
        bs = b1 ? s : bs;
    }
    float m = fmaxf(best, qdpp<Q_X1>(best));
    m = fmaxf(m, qdpp<Q_X2>(m));
    const int q0 = (int)(threadIdx.x & 63) & ~3;
    const unsigned own = (unsigned)((__ballot(best == m) >> q0) & 0xFull);
    const int owner = q0 + __builtin_ctz(own | 0x10u);
    const double il1 = rsqrt_((double)m);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const float c0 = e.g[0][i], c1 = e.g[1][i], c2 = e.g[2][i]; // (values first: a select between the addresses keeps `e` in memory)
        const float ci = bs == 0 ? c0 : (bs == 1 ? c1 : c2);
        vt[i] = (double)__shfl(ci, owner) * il1;
    }
}

// the per-problem constants (normalised cost, translation map) in LDS, 16 problems per wavefront: element k of problem p at base[16 k + p]
struct StridedView16 {
    const double *p;
    __device__ __forceinline__ double operator[](int k) const { return p[k * 16]; }
};
struct LdsStore16 {
    double *base; // &block[problem]; (72 + 55) * 16 doubles per wavefront: cost, translation map, and the iterate during the sweeps
    __device__ __forceinline__ void setQ(int k, double v) { base[k * 16] = v; }
    __device__ __forceinline__ StridedView16 Q() const { return StridedView16{base}; }
    __device__ __forceinline__ void setB(int k, double v) { base[(45 + k) * 16] = v; }
    __device__ __forceinline__ double B(int k) const { return base[(45 + k) * 16]; }
};

// The iterate W (55 doubles, identical in the four lanes) waits in LDS while the sweeps run: with it in registers the sweep loop
// spilled one of its own columns to scratch (all 256 VGPRs referenced, the AGPRs taken by longer-lived values).
#ifdef CVXL4_STASH_W
__device__ __forceinline__ void stash_iterate(LdsStore16 &st, double *W)
{
#pragma unroll
    for (int k = 0; k < 55; ++k) st.base[(72 + k) * 16] = W[k];
}
__device__ __forceinline__ void unstash_iterate(LdsStore16 &st, double *W)
{
#pragma unroll
    for (int k = 0; k < 55; ++k) W[k] = st.base[(72 + k) * 16];
}
#endif

#endif

} // namespace cvxl
