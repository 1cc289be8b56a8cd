// assemble_kernel.h -- blocked constraint assembly for problems with many correspondences.
//
// The reference's scalability benchmark runs pnp() on up to 10^4 points (benchmarks/scalability/pnp.py:37-40); its
// assembly builds C (3n x 9), N (3n x 3) and A = C - N B (cvxpnpl.py:20-104, :545-549).  Here the 60 Gram sums that
// determine B and Q = A^T A (solver_core.h: Gram) are accumulated by many workgroups per problem: this is the one
// stage of the path that is bandwidth-shaped (40 B read per point against 60 FMAs), so it is laid out for HBM --
// every lane streams its own records with 16-byte (pixels) and 8-byte (points) coalesced loads, keeps the 60 sums in
// registers, one DPP + LDS reduction per workgroup at the end, partial sums to a scratch buffer, and a second, tiny
// kernel adds them in a fixed order (deterministic results) and finishes B and Q.
//
// Conditioning: the sums are taken about a point c of the scene (cvx::shift_centre: per coordinate the median of the first three 3D records; P' = P - c).  The cost r^T Q r is
// invariant under that shift (the translation absorbs R c) and t = -B' r - R c, i.e. B[i][3j+i] += c_j: exact, and the
// Gram difference C^T C - (N^T C)^T B no longer cancels |c|^2 / spread^2 digits when the world origin is far away.
#pragma once
#include <hip/hip_runtime.h>

#include "problem_io.h"
#include "solver_core.h"

namespace cvxa {

// Threads per workgroup.  The 60-value reduction that ends a workgroup costs as much as ~8 records per lane, so a workgroup of four
// wavefronts pays only when a problem is long: from 4 096 records on (measured on ~10^7 points per launch, TB/s with four wavefronts /
// one wavefront per workgroup: 2 263 records 4.65 / 5.44, 3 294: 5.2 / 5.5, 4 326: 5.3 / 5.1, 5 357: 5.2 / 4.4, 6 389: 5.0 / 4.1,
// 10^4 x 1 000 problems: 5.7-5.9 / 4.3).
constexpr int ASM_TPB = 256, ASM_TPB_NARROW = 64;
constexpr int ASM_MAX_BLOCKS = 64; // workgroups per problem (upper bound)
__host__ __device__ inline int asm_tpb(int64_t nrec, int64_t batch) { (void)batch; return nrec >= 4096 ? ASM_TPB : ASM_TPB_NARROW; }

struct AsmArgs {
    int64_t batch;
    int n_p, n_l, K_per_problem, nblk;
    const double *p2, *p3, *l2, *l3, *K;
    double *partial; // [batch][nblk][60]
    double *Bout, *Qout; // nblk == 1: the workgroup of a problem finishes B and Q itself (no second kernel)
};

// Workgroups per problem for nrec = n_p + 2 n_l records.  The 60-value reduction at the end of a workgroup costs as
// much as ~8 records per lane, so a lane should stream many records: as few workgroups per problem as still give the
// chip ~512 workgroups in total (first build: 4 records per lane, 89 M wave-instructions per 10^7 points, VALU-bound
// at 1.9 TB/s; the reduction was 4/5 of them).
__host__ __device__ inline int asm_blocks(int64_t nrec, int64_t batch)
{
    const int tpb = asm_tpb(nrec, batch);
    const int64_t fill = 512 * (ASM_TPB / tpb);
    int64_t want = (fill + batch - 1) / (batch > 0 ? batch : 1);              // fill the chip (2 048 wavefronts) ...
    const int64_t cap = (nrec + 8 * tpb - 1) / (8 * tpb);                     // ... with at least 8 records per lane
    want = want > cap ? cap : want;
    return (int)(want < 1 ? 1 : (want > ASM_MAX_BLOCKS ? ASM_MAX_BLOCKS : want));
}

template <int CTRL>
__device__ __forceinline__ double asm_dpp(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a DPP row (every lane of the row ends with the row's sum): quad_perm [1,0,3,2], [2,3,0,1],
// row_half_mirror, row_mirror -- 12 instructions per double; the 16 row sums of a workgroup are added through LDS
__device__ __forceinline__ double asm_row_sum(double x)
{
    x += asm_dpp<0xB1>(x);
    x += asm_dpp<0x4E>(x);
    x += asm_dpp<0x141>(x);
    x += asm_dpp<0x140>(x);
    return x;
}

// B and Q (cvx::gram_finish) of problem b from its 60 Gram sums about the shift, the shift put back into B; NaN for a singular problem
__device__ __forceinline__ void asm_finish_problem(const AsmArgs &a, const int64_t b, const cvx::Gram &g, double *Bout, double *Qout)
{
    double B[27], Q9[45];
    bool ok = cvx::gram_finish(g, B, Q9);
    {   // K must be invertible too (cvx::assemble)
        const double *K = a.K + (a.K_per_problem ? b * 9 : 0);
        double Kc[9], Ki[9], det;
#pragma unroll
        for (int i = 0; i < 9; ++i) Kc[i] = K[i];
        cvx::inv3(Kc, Ki, det);
        ok = ok && (det == det) && det != 0.0;
    }
    double c[3];
    cvx::shift_centre(a.n_p, a.n_p ? a.p3 + b * a.n_p * 3 : nullptr, a.n_l, a.n_l ? a.l3 + b * a.n_l * 6 : nullptr, c);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) B[i * 9 + 3 * j + i] += c[j]; // t = -B' r - R c
#pragma unroll
    for (int i = 0; i < 27; ++i) Bout[b * 27 + i] = ok ? B[i] : NAN;
    if (Qout) {
#pragma unroll
        for (int i = 0; i < 45; ++i) Qout[b * 45 + i] = ok ? Q9[i] : NAN;
    }
}

// The point records of a problem are two arrays (pixels: 16 B each, 3D points: 24 B each).  A wavefront takes them in tiles of 64
// records -- 1 KiB + 1.5 KiB of contiguous memory -- and has the memory system copy each tile straight into its own LDS ring
// (global_load_lds_dwordx4: 16 B per lane, perfectly coalesced, no registers held while in flight), ASM_STAGES tiles ahead of the
// arithmetic; a lane then picks its record out of the tile (ds_read_b128 + three ds_read_b64).  The ring is private to the wavefront:
// no workgroup barrier in the stream, only the wavefront's own vmcnt.  (The first build let every lane load its own record --
// three 8-byte loads at a 24-byte stride per 3D point, two records in flight per lane in registers: 4.1-4.7 TB/s; deeper register
// pipelines were slower)
#ifndef CVXA_STAGES
#define CVXA_STAGES 4
#endif
constexpr int ASM_STAGES = CVXA_STAGES;       // tiles in the ring of a wavefront
constexpr int ASM_TILE_BYTES = 64 * (16 + 24); // one tile: 64 pixel records, then 64 point records
#ifndef CVXA_AUX
#define CVXA_AUX 2 // cache policy of the copies: nt (every byte is read once; measured 4.45 -> 5.0 TB/s against the default policy)
#endif

typedef __attribute__((address_space(1))) const void *asm_gptr_t;
typedef __attribute__((address_space(3))) void *asm_lptr_t;

__device__ __forceinline__ void asm_point(cvx::Gram &g, const double *Ki, const double2 uv, double X, double Y, double Z)
{
    double p[3];
    cvx::bearing(Ki, uv.x, uv.y, p);
    const double n2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
    const double T[6] = {n2 - p[0] * p[0], -p[0] * p[1], -p[0] * p[2], n2 - p[1] * p[1], -p[1] * p[2], n2 - p[2] * p[2]};
    cvx::gram_add(g, T, X, Y, Z);
}

template <int TPB>
__global__ void __launch_bounds__(TPB) assemble_large_kernel(AsmArgs a)
{
    constexpr int S = ASM_STAGES;
    static_assert((S & (S - 1)) == 0 && S >= 2, "ring size: a power of two");
    __shared__ __attribute__((aligned(16))) char ring[TPB / 64][S][ASM_TILE_BYTES];
    __shared__ double red[TPB / 16][60];
    const int64_t b = blockIdx.y;
    const int chunk = blockIdx.x, tid = threadIdx.x;
    const double *p2 = a.n_p ? a.p2 + b * a.n_p * 2 : nullptr, *p3 = a.n_p ? a.p3 + b * a.n_p * 3 : nullptr;
    const double *l2 = a.n_l ? a.l2 + b * a.n_l * 4 : nullptr, *l3 = a.n_l ? a.l3 + b * a.n_l * 6 : nullptr;
    const int stride = a.nblk * TPB;
    // points: [p]x (R P + t) = 0  (cvxpnpl.py:43-102).  Tile k of wavefront wv of this workgroup: records r0(k) .. r0(k) + 63,
    // r0(k) = (k nblk + chunk) TPB + 64 wv; lane l accumulates record r0(k) + l -- the same assignment and order on both ways below,
    // so the sums do not depend on which one a problem takes.
    const int lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t r_first = (int64_t)chunk * TPB + 64 * wv;
    // full tiles go through the ring -- when both arrays of this problem start on a 16-byte boundary (an odd number of points
    // puts every second problem's 3D points 8 bytes off: those, and every partial last tile, take the direct loads below)
    const bool ring_ok = a.n_p && (((uintptr_t)p2 | (uintptr_t)p3) & 15) == 0;
    const int nfull = (ring_ok && a.n_p >= r_first + 64) ? (int)((a.n_p - 64 - r_first) / stride) + 1 : 0;
    auto issue = [&](int k) {
        const int64_t r0 = r_first + (int64_t)k * stride;
        char *dst = &ring[wv][k & (S - 1)][0];
        const char *g2 = reinterpret_cast<const char *>(p2) + 16 * r0 + 16 * lane;
        const char *g3 = reinterpret_cast<const char *>(p3) + 24 * r0 + 16 * lane;
        __builtin_amdgcn_global_load_lds((asm_gptr_t)g2, (asm_lptr_t)dst, 16, 0, CVXA_AUX);
        __builtin_amdgcn_global_load_lds((asm_gptr_t)g3, (asm_lptr_t)(dst + 1024), 16, 0, CVXA_AUX);
        if (lane < 32) __builtin_amdgcn_global_load_lds((asm_gptr_t)(g3 + 1024), (asm_lptr_t)(dst + 2048), 16, 0, CVXA_AUX);
    };
    // K^-1 and the shift are the same in every lane: kept in scalar registers -- 186 -> 168 vector registers (the 60 sums are 120 of
    // them), i.e. three wavefronts per SIMD instead of two: 5.2 -> 5.7-5.9 TB/s at 10^4 points x 1 000 problems, 3.9 -> 5.1 at 10^3 x 10^4
    // (same-box A/B; requesting the first tiles before K and the shift are fetched, on top of that, LOSES: 4.7 / 4.5 TB/s)
    auto uni = [](double x) {
        return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(x)), __builtin_amdgcn_readfirstlane(__double2loint(x)));
    };
    const double *K = a.K + (a.K_per_problem ? b * 9 : 0);
    double Kc[9], Ki[9], det;
#pragma unroll
    for (int i = 0; i < 9; ++i) Kc[i] = K[i];
    cvx::inv3(Kc, Ki, det);
#pragma unroll
    for (int i = 0; i < 9; ++i) Ki[i] = uni(Ki[i]);
    double c_[3]; // the shift (cvx::shift_centre: the same point in every workgroup of the problem and in assemble_finish_kernel)
    cvx::shift_centre(a.n_p, p3, a.n_l, l3, c_);
    const double cx = uni(c_[0]), cy = uni(c_[1]), cz = uni(c_[2]);
    cvx::Gram g;
    cvx::gram_zero(g);
    if (a.n_p) {
        for (int k = 0; k < S - 1 && k < nfull; ++k) issue(k);
        auto consume = [&](int k) {
            const char *src = &ring[wv][k & (S - 1)][0];
            const double2 uv = *reinterpret_cast<const double2 *>(src + 16 * lane);
            const double *q = reinterpret_cast<const double *>(src + 1024 + 24 * lane);
            const double X = q[0], Y = q[1], Z = q[2];
            // the slot read one tile ago is free (its record has been consumed): the copy of tile k + S - 1 goes there
            if (k + S - 1 < nfull) issue(k + S - 1);
            asm_point(g, Ki, uv, X - cx, Y - cy, Z - cz);
        };
        int k = 0;
        // three copy instructions per tile, completing in order: tile k has landed when at most those of the S - 2 tiles behind it are out
        for (; k + (S - 2) < nfull; ++k) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (S - 2)) : "memory");
            consume(k);
        }
        for (; k < nfull; ++k) { // the last S - 2 tiles: nothing is issued any more
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            consume(k);
        }
        for (int64_t r = r_first + (int64_t)nfull * stride + lane; r < a.n_p; r += stride) {
            const double2 uv = reinterpret_cast<const double2 *>(p2)[r];
            asm_point(g, Ki, uv, p3[3 * r] - cx, p3[3 * r + 1] - cy, p3[3 * r + 2] - cz);
        }
    }
    // lines: n^T (R P_k + t) = 0 for both end points  (cvxpnpl.py:123-153)
    for (int r = chunk * TPB + tid; r < a.n_l; r += stride) {
        const double2 s0 = reinterpret_cast<const double2 *>(l2)[2 * r], s1 = reinterpret_cast<const double2 *>(l2)[2 * r + 1];
        double u[3], v[3];
        cvx::bearing(Ki, s0.x, s0.y, u);
        cvx::bearing(Ki, s1.x, s1.y, v);
        double n[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
        const double inv = cvx::rsqrt_(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
        n[0] *= inv; n[1] *= inv; n[2] *= inv;
        const double T[6] = {n[0] * n[0], n[0] * n[1], n[0] * n[2], n[1] * n[1], n[1] * n[2], n[2] * n[2]};
        const double *e = l3 + 6 * r;
        cvx::gram_add(g, T, e[0] - cx, e[1] - cy, e[2] - cz);
        cvx::gram_add(g, T, e[3] - cx, e[4] - cy, e[5] - cz);
    }
    // workgroup reduction of the 60 sums: DPP inside each row of 16 lanes, then the 16 rows through LDS
    const int row = tid >> 4;
    double *flat = &g.M0[0]; // M0[6], M1[3][6], M2[6][6] are contiguous: 60 doubles
#pragma unroll
    for (int k = 0; k < 60; ++k) {
        const double s = asm_row_sum(flat[k]);
        if ((tid & 15) == 0) red[row][k] = s;
    }
    __syncthreads();
    double s = 0;
    if (tid < 60) {
#pragma unroll
        for (int w = 0; w < TPB / 16; ++w) s += red[w][tid];
        if (a.nblk > 1) a.partial[(b * a.nblk + chunk) * 60 + tid] = s;
    }
    if (a.nblk == 1) { // the only workgroup of its problem: the sums are complete -- B and Q here, one entry per lane (cvx::gram_finish)
        __syncthreads();
        double *m = &red[0][0], *mi = &red[1][0]; // the 60 sums; the inverse of N^T N (9) and the verdict
        if (tid < 60) m[tid] = s;
        __syncthreads();
        if (tid == 0) {
            const double M0[9] = {m[0], m[1], m[2], m[1], m[3], m[4], m[2], m[4], m[5]};
            double Mi[9], d0;
            cvx::inv3(M0, Mi, d0);
            const double scale = m[0] + m[3] + m[5];
            const bool ok = (d0 > 1e-12 * (scale * scale * scale) * (1.0 / 27.0)) && (det == det) && det != 0.0; // (K must be invertible too)
#pragma unroll
            for (int i = 0; i < 9; ++i) mi[i] = Mi[i];
            mi[9] = ok ? 1.0 : 0.0;
        }
        __syncthreads();
        const bool ok = mi[9] != 0.0;
        auto psym = [](int i, int j) { const int lo = i < j ? i : j, hi = i < j ? j : i; return lo * 3 - (lo == 2 ? 1 : 0) + (hi - lo); };
        for (int w = tid; w < 27 + 45; w += TPB) {
            if (w < 27) { // B[i][3 a + j] = sum_k Mi[i][k] sym(M1[a])[k][j], the shift put back: t = -B' r - R c
                const int i = w / 9, a3 = (w % 9) / 3, j = w % 3;
                const double *s1 = m + 6 + 6 * a3;
                double acc = mi[i * 3] * s1[psym(0, j)] + mi[i * 3 + 1] * s1[psym(1, j)] + mi[i * 3 + 2] * s1[psym(2, j)];
                if (j == i) acc += a3 == 0 ? cx : (a3 == 1 ? cy : cz);
                a.Bout[b * 27 + w] = ok ? acc : NAN;
            } else if (a.Qout) { // Q[3 a + i][3 b + j] = sym(M2[ab])[i][j] - sum_k sym(M1[a])[i][k] B'[k][3 b + j]
                const int q = w - 27;
                int r = 0, rem = q;
                while (rem >= 9 - r) { rem -= 9 - r; ++r; }
                const int cc = r + rem, a3 = r / 3, i = r % 3, b3 = cc / 3, j = cc % 3;
                const double *sa = m + 6 + 6 * a3, *sb = m + 6 + 6 * b3, *t2 = m + 24 + 6 * psym(a3, b3);
                double acc = t2[psym(i, j)];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const double bk = mi[k * 3] * sb[psym(0, j)] + mi[k * 3 + 1] * sb[psym(1, j)] + mi[k * 3 + 2] * sb[psym(2, j)];
                    acc -= sa[psym(i, k)] * bk;
                }
                a.Qout[b * 45 + q] = ok ? acc : NAN;
            }
        }
    }
}
static_assert(sizeof(cvx::Gram) == 60 * sizeof(double), "Gram must be 60 contiguous doubles");

// fixed-order sum of the partial Gram sums, then B and Q (cvx::gram_finish) with the shift put back into B
__global__ void __launch_bounds__(64) assemble_finish_kernel(AsmArgs a, double *Bout, double *Qout)
{
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= a.batch) return;
    cvx::Gram g;
    double *flat = &g.M0[0];
#pragma unroll
    for (int k = 0; k < 60; ++k) flat[k] = 0.0;
    for (int c = 0; c < a.nblk; ++c) {
        const double *p = a.partial + (b * a.nblk + c) * 60;
#pragma unroll
        for (int k = 0; k < 60; ++k) flat[k] += p[k];
    }
    asm_finish_problem(a, b, g, Bout, Qout);
}

} // namespace cvxa
