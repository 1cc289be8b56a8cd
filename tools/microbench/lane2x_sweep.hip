// lane2x_sweep.hip -- probe for the round-5 verdict's item 2: a float64 lane phase with TWO lanes per problem (32 problems per wavefront).
//
// What sinks the shipped float64 lane phase (cvxl::lane_phase_f64, solve_lane2_kernel<true>: 256 VGPR + 256 AGPR, 520 B of scratch, one
// wavefront per SIMD) is the eigen-solve: a sweep has all 100 doubles of the ten columns as operands.  This file times THAT part alone in
// the two geometries, same matrices, same number of sweeps:
//   A  one lane per problem: cvx::Eig + cvx::eig_solve, the code the shipped kernel runs (rotation parameters in float64: exact = true);
//   B  two lanes per problem: five columns per lane; per sweep 10 local pairs per lane, the 5 pairs (A_i, B_i) where both lanes rotate
//      their own column, and the 20 pairs (A_i, B_{i+r}), r = 1..4, taken two at a time -- lane 0 drives (A_i, B_{i+r}) while lane 1
//      drives (B_i, A_{i+r}) for r = 1, 2, which are lane 0's offsets 4, 3: each lane updates its own column of its own pair and its
//      column of the partner's pair (partner's column and rotation read through DPP quad_perm [1,0,3,2]).  Every pair once per sweep.
// Output per problem: the ten squared column norms after the sweeps (= squared eigenvalues of the SPD input: checked by the driver).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Rpass-analysis=kernel-resource-usage -o liblane2x_sweep.so lane2x_sweep.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../cvxpnpl_amd/csrc/solver_core.h"

#ifndef B_WAVES
#define B_WAVES 2 // wavefronts per SIMD the two-lane kernel is compiled for
#endif

__global__ void __launch_bounds__(64) sweep_one_lane_kernel(int64_t batch, const double *W55, int sweeps, double *n2_out)
{
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    double W[55];
#pragma unroll
    for (int i = 0; i < 55; ++i) W[i] = W55[b * 55 + i];
    cvx::Eig e;
    cvx::set_exact(e, true);
    cvx::eig_load(e, W);
    cvx::eig_solve(e, sweeps, 0.0);
#pragma unroll
    for (int j = 0; j < 10; ++j) n2_out[b * 10 + j] = e.n2[j];
}

__device__ __forceinline__ double xlane(double x) // the partner lane's value (lanes 2k <-> 2k + 1)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0xB1, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0xB1, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dot10(const double (&a)[10], const double (&b)[10])
{
    double s0 = a[0] * b[0], s1 = a[1] * b[1];
#pragma unroll
    for (int i = 2; i < 10; i += 2) { s0 = fma(a[i], b[i], s0); s1 = fma(a[i + 1], b[i + 1], s1); }
    return s0 + s1;
}

__global__ void __launch_bounds__(64, B_WAVES) sweep_two_lane_kernel(int64_t batch, const double *W55, int sweeps, double *n2_out)
{
    const int64_t t = (int64_t)blockIdx.x * 64 + threadIdx.x;
    const int64_t b = t >> 1;
    const int half = (int)(t & 1);
    const int64_t bb = b < batch ? b : batch - 1; // (a wavefront's tail lanes redo the last problem: DPP partners stay meaningful)
    // columns 5 * half .. 5 * half + 4 of G = W + sigma I
    double fro = 0.0;
    for (int i = 0; i < 55; ++i) { const double w = W55[bb * 55 + i]; fro += w * w; }
    double g[5][10], n2[5];
    {
        double dsum = 0.0;
        for (int i = 0; i < 10; ++i) { const double w = W55[bb * 55 + cvx::sidx(i, i)]; dsum += w * w; }
        const double sigma = 1.5 * sqrt(2.0 * fro - dsum) + 1e-300;
#pragma unroll
        for (int c = 0; c < 5; ++c) {
            const int j = 5 * half + c;
#pragma unroll
            for (int i = 0; i < 10; ++i) g[c][i] = W55[bb * 55 + cvx::sidx(i, j)] + (i == j ? sigma : 0.0);
        }
    }
    for (int sw = 0; sw < sweeps; ++sw) {
#pragma unroll
        for (int c = 0; c < 5; ++c) n2[c] = dot10(g[c], g[c]); // exact norms once per sweep (the incremental update drifts)
        // ---- local pairs
#pragma unroll
        for (int p = 0; p < 5; ++p)
#pragma unroll
            for (int q = p + 1; q < 5; ++q) {
                const double gam = dot10(g[p], g[q]);
                double c, s, dl;
                cvx::jacobi_cs_dl(n2[p], n2[q], gam, gam * gam > 1e-30 * n2[p] * n2[q], c, s, dl, true);
#pragma unroll
                for (int i = 0; i < 10; ++i) { const double a = g[p][i], bq = g[q][i]; g[p][i] = c * a - s * bq; g[q][i] = s * a + c * bq; }
                n2[p] += dl; n2[q] -= dl;
            }
        // ---- cross pairs (own i, partner i): each lane rotates its own column
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            double pb[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) pb[k] = xlane(g[i][k]);
            const double pbn = xlane(n2[i]);
            const double gam = dot10(g[i], pb);
            double c, s, dl;
            cvx::jacobi_cs_dl(n2[i], pbn, gam, gam * gam > 1e-30 * n2[i] * pbn, c, s, dl, true);
#pragma unroll
            for (int k = 0; k < 10; ++k) g[i][k] = c * g[i][k] - s * pb[k];
            n2[i] += dl;
        }
        // ---- cross pairs (own i, partner (i + r) % 5), r = 1, 2: two disjoint pairs per sub-step, one driven by each lane
#pragma unroll
        for (int r = 1; r <= 2; ++r)
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int j = (i + r) % 5;
                double pb[10], pa[10];
#pragma unroll
                for (int k = 0; k < 10; ++k) { pb[k] = xlane(g[j][k]); pa[k] = xlane(g[i][k]); } // partner's columns j and i, before any update
                const double pbn = xlane(n2[j]);
                const double gam = dot10(g[i], pb);
                double c, s, dl;
                cvx::jacobi_cs_dl(n2[i], pbn, gam, gam * gam > 1e-30 * n2[i] * pbn, c, s, dl, true);
                const double pc = xlane(c), ps = xlane(s), pdl = xlane(dl); // the rotation the partner drives: (its i, my j)
#pragma unroll
                for (int k = 0; k < 10; ++k) {
                    const double mi = g[i][k], mj = g[j][k];
                    g[i][k] = c * mi - s * pb[k];     // my pair:        a' = c a - s b
                    g[j][k] = ps * pa[k] + pc * mj;   // partner's pair: b' = s a + c b
                }
                n2[i] += dl; n2[j] -= pdl;
            }
    }
    if (b < batch) {
#pragma unroll
        for (int c = 0; c < 5; ++c) n2_out[b * 10 + 5 * half + c] = dot10(g[c], g[c]);
    }
}

extern "C" int lane2x_run(int which, int64_t batch, const double *W55, int sweeps, double *n2_out, int reps, float *ms_out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned gridA = (unsigned)((batch + 63) / 64), gridB = (unsigned)((2 * batch + 63) / 64);
    for (int rep = -2; rep < reps; ++rep) {
        if (rep == 0) hipEventRecord(e0, 0);
        if (which == 0) hipLaunchKernelGGL(sweep_one_lane_kernel, dim3(gridA), dim3(64), 0, 0, batch, W55, sweeps, n2_out);
        else hipLaunchKernelGGL(sweep_two_lane_kernel, dim3(gridB), dim3(64), 0, 0, batch, W55, sweeps, n2_out);
    }
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    hipEventElapsedTime(ms_out, e0, e1);
    *ms_out /= reps;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return (int)hipGetLastError();
}
