// Microbenchmark of the lane phase alone (first kernel of the lane-hybrid schedule): builds ONE variant of cvxl::lane_phase /
// lane_phase_f64 per shared object, so that register-allocation experiments on lane_core.h (compile flags, -D switches) can be
// compiled in seconds and timed side by side on one box.   tools/microbench/lane_bench.py drives it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I cvxpnpl_amd/csrc [-DLANE_F64] [-D...] -o lane_bench_X.so lane_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "problem_io.h"
#include "solver_core.h"
#ifdef LANE_CORE_H
#include LANE_CORE_H
#else
#include "lane_core.h"
#endif

struct LArgs {
    int64_t batch;
    int n_p;
    const double *p2, *p3, *K;
    double *R, *t;
    int32_t *status, *iters, *sweeps;
    int32_t *qcount, *qentries;
    double *ws;
};

__global__ void __launch_bounds__(64) lane_bench_kernel(LArgs a, cvx::Opts o, int handoff_at)
{
    __shared__ double lds_const[72 * 64];
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.batch) return;
    cvx::ProblemView pv = cvx::make_view(b, a.n_p, a.p2, a.p3, 0, nullptr, nullptr, a.K, 0);
    cvx::Solution sol;
#ifdef LANE_F64
    cvxl::lane_phase_f64(pv, o, sol, nullptr, handoff_at, a.ws + b * 56, cvx::LdsStore{lds_const + threadIdx.x});
#else
    cvxl::lane_phase(pv, o, sol, nullptr, handoff_at, a.ws + b * 56, cvx::LdsStore{lds_const + threadIdx.x});
#endif
    a.iters[b] = sol.iters;
    a.sweeps[b] = sol.sweeps;
    a.status[b] = sol.status;
    if (sol.status == -1) {
        const int q = atomicAdd(a.qcount, 1);
        a.qentries[q] = (int32_t)b;
        return;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) a.R[b * 9 + i] = sol.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) a.t[b * 3 + i] = sol.t[i];
}

extern "C" int lane_bench_run(int64_t batch, int n_p, const double *p2, const double *p3, const double *K, double *R, double *t, int32_t *status,
                              int32_t *iters, int32_t *sweeps, int32_t *qcount, int32_t *qentries, double *ws, int handoff_at, int f32_until, int reps,
                              float *ms_out)
{
    LArgs a{batch, n_p, p2, p3, K, R, t, status, iters, sweeps, qcount, qentries, ws};
    cvx::Opts o = cvx::default_opts();
    o.first_check = handoff_at;
    o.f32_sweeps_until = f32_until;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned grid = (unsigned)((batch + 63) / 64);
    for (int w = 0; w < 2; ++w) {
        hipMemsetAsync(qcount, 0, 4, 0);
        hipLaunchKernelGGL(lane_bench_kernel, dim3(grid), dim3(64), 0, 0, a, o, handoff_at);
    }
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) {
        hipMemsetAsync(qcount, 0, 4, 0);
        hipLaunchKernelGGL(lane_bench_kernel, dim3(grid), dim3(64), 0, 0, a, o, handoff_at);
    }
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return -2;
    hipEventElapsedTime(ms_out, e0, e1);
    *ms_out /= reps;
    hipEventDestroy(e0); hipEventDestroy(e1);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
