// lane_twin_repro.hip -- reproducer for the oddity parked since round 1 (HISTORY.md, end of old section 9): a one-problem-per-lane kernel that
// carries the WHOLE scalar solve including the twin-candidate logic -- cvx::solve_problem<TWIN = true> -- "produced NaN columns on the device
// for some register allocations although the same source is clean on the host under ASan/UBSan".  The product never instantiates it (the lane
// phase ends before the twin logic starts); this file does, as a kernel of its own, so that it can be compiled under different options
// and held against the host build of the same header (tools/microbench/lane_twin_repro.py).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC [-DREPRO_BOUNDS=..] [-mllvm -enable-ipra=0] -o liblane_twin_repro.so lane_twin_repro.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../cvxpnpl_amd/csrc/problem_io.h"
#include "../../cvxpnpl_amd/csrc/solver_core.h"

#ifndef REPRO_WAVES
#define REPRO_WAVES 1 // __launch_bounds__(64, REPRO_WAVES): 1 -> 512 registers per lane, 2 -> 256, 4 -> 128 (more spills, other allocations)
#endif

template <bool TWIN, bool DBL = TWIN>
__global__ void __launch_bounds__(64, REPRO_WAVES) lane_full_kernel(int64_t batch, int n_p, const double *p2, const double *p3, const double *K, cvx::Opts o,
                                                                    double *R, int32_t *status, int32_t *iters, double *Z)
{
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    cvx::ProblemView pv = cvx::make_view(b, n_p, p2, p3, 0, nullptr, nullptr, K, 0);
    cvx::Solution sol;
    double Zl[55];
    cvx::solve_problem<TWIN, cvx::RegStore, cvx::VAR_FULL, DBL>(pv, o, sol, Zl);
    for (int i = 0; i < 9; ++i) R[b * 9 + i] = sol.R[i];
    for (int i = 0; i < 55; ++i) Z[b * 55 + i] = Zl[i];
    status[b] = sol.status;
    iters[b] = sol.iters;
}

extern "C" int repro_run(int twin, int64_t batch, int n_p, const double *p2, const double *p3, const double *K, int max_iters, int f64, double *R, int32_t *status,
                         int32_t *iters, double *Z, void *stream)
{
    cvx::Opts o = cvx::default_opts();
    o.max_iters = max_iters;
    o.sweep_schedule = (f64 & 2) ? 0 : 1; // (bit 1: no cap on the sweeps of an eigen-solve -- without the twin logic the cap is the LANE PHASE's: one sweep from iteration 5 on)
    f64 &= 1;
    o.f32_sweeps_until = f64 ? 0 : 64;
    o.rescue_from = 0;
    const unsigned grid = (unsigned)((batch + 63) / 64);
    if (twin == 2) hipLaunchKernelGGL((lane_full_kernel<false, true>), dim3(grid), dim3(64), 0, (hipStream_t)stream, batch, n_p, p2, p3, K, o, R, status, iters, Z); // no twin logic, float64 sweeps
    else if (twin) hipLaunchKernelGGL(lane_full_kernel<true>, dim3(grid), dim3(64), 0, (hipStream_t)stream, batch, n_p, p2, p3, K, o, R, status, iters, Z);
    else hipLaunchKernelGGL(lane_full_kernel<false>, dim3(grid), dim3(64), 0, (hipStream_t)stream, batch, n_p, p2, p3, K, o, R, status, iters, Z);
    return (int)hipGetLastError();
}
