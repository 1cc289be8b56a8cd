// Probe: can a second kernel on a LOW-PRIORITY stream fill the SIMDs that the last round of a long kernel leaves free, without taking
// slots from its pending workgroups?  Kernel A: the lane phase (1 wavefront per SIMD, two rounds at 125 k problems), every block stamps
// its start and end; kernel B (low-priority stream, enqueued right behind A on the host): 2048 one-wavefront blocks (2 per SIMD: 256
// registers) that stamp their start and spin for `spin_us`.  tools/microbench/overlap_probe.py prints when B's blocks started relative to A's.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "problem_io.h"
#include "solver_core.h"
#include "lane_core.h"

struct LArgs {
    int64_t batch;
    int n_p;
    const double *p2, *p3, *K;
    double *R, *t;
    int32_t *status;
    int32_t *qcount, *qentries;
    double *ws;
    unsigned long long *t0, *t1;
};

__global__ void __launch_bounds__(64) probe_lane_kernel(LArgs a, cvx::Opts o, int handoff_at)
{
    __shared__ double lds_const[72 * 64];
    const unsigned long long ts = wall_clock64();
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    cvx::Solution sol;
    sol.status = 3;
    if (b < a.batch) {
        cvx::ProblemView pv = cvx::make_view(b, a.n_p, a.p2, a.p3, 0, nullptr, nullptr, a.K, 0);
        cvxl::lane_phase(pv, o, sol, nullptr, handoff_at, a.ws + b * 56, cvx::LdsStore{lds_const + threadIdx.x});
        a.status[b] = sol.status;
        if (sol.status == -1) {
            const int q = atomicAdd(a.qcount, 1);
            a.qentries[q] = (int32_t)b;
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) a.R[b * 9 + i] = sol.R[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) a.t[b * 3 + i] = sol.t[i];
        }
    }
    if (threadIdx.x == 0) { a.t0[blockIdx.x] = ts; a.t1[blockIdx.x] = wall_clock64(); }
}

__global__ void __launch_bounds__(64, 2) probe_b_kernel(unsigned long long *t0, unsigned long long *t1, int spin_ticks, double *sink)
{
    __shared__ double pad[900]; // (LDS footprint of the resume kernel: 7.2 KB)
    const unsigned long long ts = wall_clock64();
    double x = threadIdx.x;
    // hold 200+ registers like the resume kernel so that two of these fill a SIMD's register file with a lane wavefront absent only
    double r[96];
#pragma unroll
    for (int i = 0; i < 96; ++i) r[i] = x + i;
    while ((long long)(wall_clock64() - ts) < spin_ticks) {
#pragma unroll
        for (int i = 0; i < 96; ++i) r[i] = fma(r[i], 1.0000001, 1e-9);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 96; ++i) s += r[i];
    pad[threadIdx.x] = s;
    if (s == 12345.678) sink[0] = pad[(threadIdx.x + 1) & 63];
    if (threadIdx.x == 0) { t0[blockIdx.x] = ts; t1[blockIdx.x] = wall_clock64(); }
}

extern "C" int overlap_probe_run(int64_t batch, int n_p, const double *p2, const double *p3, const double *K, double *R, double *t, int32_t *status,
                                 int32_t *qcount, int32_t *qentries, double *ws, unsigned long long *ta0, unsigned long long *ta1,
                                 unsigned long long *tb0, unsigned long long *tb1, int nb, int spin_ticks, int mode, double *sink)
{
    LArgs a{batch, n_p, p2, p3, K, R, t, status, qcount, qentries, ws, ta0, ta1};
    cvx::Opts o = cvx::default_opts();
    o.first_check = 6;
    o.f32_sweeps_until = 64;
    int lo = 0, hi = 0;
    hipDeviceGetStreamPriorityRange(&lo, &hi); // lo = least priority (numerically greatest)
    hipStream_t sa, sb;
    hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, mode == 2 ? hi : (lo + hi) / 2);
    hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, mode == 0 ? (lo + hi) / 2 : lo);
    const unsigned grid = (unsigned)((batch + 63) / 64);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemsetAsync(qcount, 0, 4, sa);
        hipStreamSynchronize(sa);
        hipLaunchKernelGGL(probe_lane_kernel, dim3(grid), dim3(64), 0, sa, a, o, 6);
        hipLaunchKernelGGL(probe_b_kernel, dim3(nb), dim3(64), 0, sb, tb0, tb1, spin_ticks, sink);
        hipStreamSynchronize(sa);
        hipStreamSynchronize(sb);
    }
    hipStreamDestroy(sa); hipStreamDestroy(sb);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
