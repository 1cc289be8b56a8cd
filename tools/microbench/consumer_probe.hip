// Experiment (round 4, verdict item 7): the parked problems of the lane phase consumed WHILE the lane kernel still runs, by a persistent
// kernel on a second stream that is gated behind "every workgroup of the lane kernel has been dispatched" (a low-priority stream alone
// takes the slots of the lane kernel's pending workgroups: overlap_probe.hip).  tools/microbench/consumer_probe.py times
//   mode 0: lane kernel, then the consumer kernel on the same stream (= what the library does: lane phase, then resume phase);
//   mode 1: lane kernel on stream A; on stream B a one-wavefront gate that polls the `started` counter, then the consumer kernel.
// Queue: the lane kernel appends at atomicAdd(pushed) and stores the index with release; consumers claim positions by CAS on `claimed`
// while claimed < pushed, wait for the entry, and leave when every lane workgroup is done and nothing is left to claim.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "problem_io.h"
#include "solver_core.h"
#include "lane_core.h"
#include "wave_kernel.h"

struct LArgs {
    int64_t batch;
    int n_p;
    const double *p2, *p3, *K;
    double *R, *t;
    int32_t *status, *iters;
    int32_t *ctr;      // [0] started, [32] pushed, [64] claimed, [96] done  (one counter per 128-byte line)
    int32_t *entries;  // -1 = empty
    double *ws;
    unsigned long long *tend; // per lane block: end stamp (100 MHz)
};

__global__ void __launch_bounds__(64) cp_lane_kernel(LArgs a, cvx::Opts o, int handoff_at)
{
    __shared__ double lds_const[72 * 64];
    if (threadIdx.x == 0) atomicAdd(a.ctr + 0, 1);
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b < a.batch) {
        cvx::ProblemView pv = cvx::make_view(b, a.n_p, a.p2, a.p3, 0, nullptr, nullptr, a.K, 0);
        cvx::Solution sol;
        cvxl::lane_phase(pv, o, sol, nullptr, handoff_at, a.ws + b * 56, cvx::LdsStore{lds_const + threadIdx.x});
        if (sol.status == -1) {
            __threadfence();
            const int q = atomicAdd(a.ctr + 32, 1);
            __hip_atomic_store(a.entries + q, (int32_t)b, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
#pragma unroll
            for (int i = 0; i < 9; ++i) a.R[b * 9 + i] = sol.R[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) a.t[b * 3 + i] = sol.t[i];
            a.status[b] = sol.status;
            a.iters[b] = sol.iters;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    __threadfence();
    if (threadIdx.x == 0) { a.tend[blockIdx.x] = wall_clock64(); __hip_atomic_fetch_add(a.ctr + 96, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT); }
}

__global__ void cp_gate_kernel(int32_t *ctr, int grid)
{
    for (int spin = 0; spin < (1 << 22); ++spin) {
        if (__hip_atomic_load(ctr + 0, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= grid) return;
        __builtin_amdgcn_s_sleep(64);
    }
}

struct CArgs {
    cvxw::WaveArgs a;
    cvx::Opts o;
    int32_t *ctr, *entries;
    const double *ws;
    int lane_grid;
    unsigned long long *t0, *t1; // per consumer block: first claim, exit
    int32_t *nsolved;
};
typedef const __attribute__((address_space(4))) CArgs *CArgsPtr;

__global__ void __launch_bounds__(64, 2) cp_consumer_kernel(CArgs k)
{
#if defined(__HIP_DEVICE_COMPILE__) // (the host pass cannot copy out of the constant address space)
    __shared__ __attribute__((aligned(16))) double lds_all[cvxw::LDSW];
    CArgsPtr kp = (CArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    const int lane = threadIdx.x;
    int32_t *ctr = kp->ctr, *entries = kp->entries;
    const int lane_grid = kp->lane_grid;
    unsigned long long first = 0;
    int solved = 0;
    for (;;) {
        int pos = -1;
        if (lane == 0) {
            for (int spin = 0; spin < (1 << 20); ++spin) {
                const int done = __hip_atomic_load(ctr + 96, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                const int pushed = __hip_atomic_load(ctr + 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                int cur = __hip_atomic_load(ctr + 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (cur < pushed) {
                    if (__hip_atomic_compare_exchange_strong(ctr + 64, &cur, cur + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { pos = cur; break; }
                    continue;
                }
                if (done >= lane_grid) { pos = -2; break; }
                __builtin_amdgcn_s_sleep(127);
            }
            if (pos == -1) pos = -2; // (gave up)
        }
        pos = __builtin_amdgcn_readfirstlane(pos);
        if (pos < 0) break;
        int32_t b = -1;
        for (int spin = 0; spin < (1 << 20) && b < 0; ++spin) b = __hip_atomic_load(entries + pos, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b < 0) break;
        if (!first) first = wall_clock64();
        const cvxw::WaveArgs a = kp->a;
        const cvx::Opts o = kp->o;
        cvxw::solve_one_wave<cvx::VAR_FULL>(a, o, b, lds_all, kp->ws + (int64_t)b * 56, false);
        CVXW_SYNC();
        ++solved;
    }
    if (lane == 0) { kp->t0[blockIdx.x] = first; kp->t1[blockIdx.x] = wall_clock64(); kp->nsolved[blockIdx.x] = solved; }
#endif
}

extern "C" int consumer_probe_run(int64_t batch, int n_p, const double *p2, const double *p3, const double *K, double *R, double *t, int32_t *status,
                                  int32_t *iters, double *cost, int32_t *work, int32_t *ctr, int32_t *entries, double *ws, unsigned long long *tend,
                                  unsigned long long *t0, unsigned long long *t1, int32_t *nsolved, int cgrid, int mode, int reps, float *ms_out)
{
    const unsigned grid = (unsigned)((batch + 63) / 64);
    LArgs la{batch, n_p, p2, p3, K, R, t, status, iters, ctr, entries, ws, tend};
    cvx::Opts o = cvx::default_opts();
    o.first_check = 6;
    o.f32_sweeps_until = 64;
    o.rescue_from = 0;
    CArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.a.batch = batch; ca.a.n_p = n_p; ca.a.n_l = 0; ca.a.K_per_problem = 0;
    ca.a.p2 = p2; ca.a.p3 = p3; ca.a.K = K; ca.a.R = R; ca.a.t = t; ca.a.cost = cost; ca.a.status = status; ca.a.iters = iters; ca.a.work = work;
    ca.o = o; ca.ctr = ctr; ca.entries = entries; ca.ws = ws; ca.lane_grid = (int)grid; ca.t0 = t0; ca.t1 = t1; ca.nsolved = nsolved;
    hipStream_t sa, sb;
    hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    hipEvent_t e0, e1, eb;
    hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&eb);
    float total = 0;
    for (int rep = 0; rep < reps + 2; ++rep) {
        hipMemsetAsync(ctr, 0, 128 * 4, sa);
        hipMemsetAsync(entries, 0xFF, (size_t)(batch + 64) * 4, sa);
        hipStreamSynchronize(sa);
        hipEventRecord(e0, sa);
        hipLaunchKernelGGL(cp_lane_kernel, dim3(grid), dim3(64), 0, sa, la, o, 6);
        if (mode == 0) {
            hipLaunchKernelGGL(cp_consumer_kernel, dim3(cgrid), dim3(64), 0, sa, ca);
            hipEventRecord(e1, sa);
            hipEventSynchronize(e1);
        } else {
            hipLaunchKernelGGL(cp_gate_kernel, dim3(1), dim3(1), 0, sb, ctr, (int)grid);
            hipLaunchKernelGGL(cp_consumer_kernel, dim3(cgrid), dim3(64), 0, sb, ca);
            hipEventRecord(eb, sb);
            hipStreamWaitEvent(sa, eb, 0);
            hipEventRecord(e1, sa);
            hipEventSynchronize(e1);
        }
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) total += ms;
    }
    *ms_out = total / reps;
    hipStreamDestroy(sa); hipStreamDestroy(sb);
    hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(eb);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
