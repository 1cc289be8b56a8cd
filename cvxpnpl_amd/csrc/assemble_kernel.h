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

constexpr int ASM_TPB = 256;      // threads per workgroup
constexpr int ASM_MAX_BLOCKS = 64; // workgroups per problem (upper bound)

struct AsmArgs {
    int64_t batch;
    int n_p, n_l, K_per_problem, nblk;
    const double *p2, *p3, *l2, *l3, *K;
    double *partial; // [batch][nblk][60]
};

// Workgroups per problem for nrec = n_p + 2 n_l records.  The 60-value reduction at the end of a workgroup costs as
// much as ~8 records per lane, so a lane should stream many records: as few workgroups per problem as still give the
// chip ~512 workgroups in total (first build: 4 records per lane, 89 M wave-instructions per 10^7 points, VALU-bound
// at 1.9 TB/s; the reduction was 4/5 of them).
__host__ __device__ inline int asm_blocks(int64_t nrec, int64_t batch)
{
    int64_t want = (512 + batch - 1) / (batch > 0 ? batch : 1);               // fill the chip (512 workgroups of 4 wavefronts) ...
    const int64_t cap = (nrec + 8 * ASM_TPB - 1) / (8 * ASM_TPB);             // ... with at least 8 records per lane
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

__global__ void __launch_bounds__(ASM_TPB) assemble_large_kernel(AsmArgs a)
{
    __shared__ double red[ASM_TPB / 16][60];
    const int64_t b = blockIdx.y;
    const int chunk = blockIdx.x, tid = threadIdx.x;
    const double *K = a.K + (a.K_per_problem ? b * 9 : 0);
    double Kc[9], Ki[9], det;
#pragma unroll
    for (int i = 0; i < 9; ++i) Kc[i] = K[i];
    cvx::inv3(Kc, Ki, det);
    const double *p2 = a.n_p ? a.p2 + b * a.n_p * 2 : nullptr, *p3 = a.n_p ? a.p3 + b * a.n_p * 3 : nullptr;
    const double *l2 = a.n_l ? a.l2 + b * a.n_l * 4 : nullptr, *l3 = a.n_l ? a.l3 + b * a.n_l * 6 : nullptr;
    double c_[3]; // the shift (cvx::shift_centre: the same point in every workgroup of the problem and in assemble_finish_kernel)
    cvx::shift_centre(a.n_p, p3, a.n_l, l3, c_);
    const double cx = c_[0], cy = c_[1], cz = c_[2];
    cvx::Gram g;
    cvx::gram_zero(g);
    const int stride = a.nblk * ASM_TPB;
    // points: [p]x (R P + t) = 0  (cvxpnpl.py:43-102).  Software-pipelined: the next record's five doubles are requested
    // before the current one's 100 instructions, so that every lane keeps two records in flight.
    {
        int r = chunk * ASM_TPB + tid;
        double2 uv = make_double2(0.0, 0.0);
        double X = 0.0, Y = 0.0, Z = 0.0;
        if (r < a.n_p) { uv = reinterpret_cast<const double2 *>(p2)[r]; X = p3[3 * r]; Y = p3[3 * r + 1]; Z = p3[3 * r + 2]; }
        while (r < a.n_p) {
            const int rn = r + stride;
            double2 uvn = make_double2(0.0, 0.0);
            double Xn = 0.0, Yn = 0.0, Zn = 0.0;
            if (rn < a.n_p) { uvn = reinterpret_cast<const double2 *>(p2)[rn]; Xn = p3[3 * rn]; Yn = p3[3 * rn + 1]; Zn = p3[3 * rn + 2]; }
            double p[3];
            cvx::bearing(Ki, uv.x, uv.y, p);
            const double n2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
            const double T[6] = {n2 - p[0] * p[0], -p[0] * p[1], -p[0] * p[2], n2 - p[1] * p[1], -p[1] * p[2], n2 - p[2] * p[2]};
            cvx::gram_add(g, T, X - cx, Y - cy, Z - cz);
            uv = uvn; X = Xn; Y = Yn; Z = Zn;
            r = rn;
        }
    }
    // lines: n^T (R P_k + t) = 0 for both end points  (cvxpnpl.py:123-153)
    for (int r = chunk * ASM_TPB + tid; r < a.n_l; r += stride) {
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
    if (tid < 60) {
        double s = 0;
#pragma unroll
        for (int w = 0; w < ASM_TPB / 16; ++w) s += red[w][tid];
        a.partial[(b * a.nblk + chunk) * 60 + tid] = s;
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

} // namespace cvxa
