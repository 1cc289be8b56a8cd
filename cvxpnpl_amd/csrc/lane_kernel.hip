// lane_kernel.hip -- solve_lane2_kernel, the lane-per-problem first phase of the lane-hybrid schedule, as a translation unit of its own.
//
// Why its own: the reciprocal root / reciprocal of the device code (cvx::rsqrt_, cvx::rcp) refine the hardware seed with ONE third-order step since the
// end of round 5 -- two / one float64 operations fewer per call, which the quad and wave-per-problem kernels turn into 1-3 % -- and the same change,
// which does not alter the instruction count of THIS kernel (14 376 -> 14 372), moved its register allocation (scratch 84 -> 184 B / 520 -> 600 B per
// lane) and cost the lane-layout launches 0.2-2 % (profiles/r05/rsq_c3_ab.txt).  Built with CVX_REFINE_NEWTON2 this unit keeps the two-Newton
// sequences and with them the allocation it had.  The shared headers put their definitions into an inline namespace named after the macro
// (solver_core.h: CVX_UNIT_TAG), so the two units' versions of the same inline function are different symbols.
#define CVX_REFINE_NEWTON2
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "problem_io.h"
#include "solver_core.h"
#include "lane_core.h"
#include "batch_args.h"

namespace cvxb {

// Lane-per-problem, the first phase of the lane-hybrid schedule: each lane owns one problem (assembly -> ADMM -> one certificate attempt ->
// pose) for the first handoff_at (2..6) iterations; 64 independent problems per wavefront, no cross-lane traffic.  A lane that is not
// finished by then parks its iterate in ws[b] and queues b for resume_wave_kernel, so that one slow problem cannot hold the other 63 lanes.
// The register-budgeted restatement of the scalar core (lane_core.h): the schedule the launch policy
// actually uses -- handoff_at iterations, one certificate attempt after the last, single-precision sweeps -- written straight
// line with streamed projections.  512 registers (256 + 256), ~20 spilled, 60 B of scratch per lane; the general core above needs
// 2 640 B per lane (1 006 spilled registers, 1.1 GB of HBM traffic per 125 k launch) and stays for every other combination of
// options (float64 sweeps, hand-off point != first attempt).
// F64SW: every sweep, the product that starts them and the rotation angles in float64 (opts.f32_sweeps_until below the length of the
// phase) -- cvxl::lane_phase_f64, the positive part streamed row by row instead of stored.
template <bool F64SW>
__global__ void __launch_bounds__(64) solve_lane2_kernel(BatchArgs a, cvx::Opts o, int handoff_at, int32_t *qcount, int32_t *qentries, double *ws)
{
    __shared__ double lds_const[72 * 64];
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.batch) return;
    cvx::ProblemView pv = cvx::make_view(b, a.n_p, a.p2, a.p3, a.n_l, a.l2, a.l3, a.K, a.K_per_problem);
    if (a.Q45) { pv.Q45 = a.Q45 + b * 45; pv.B27 = a.B27 + b * 27; }
    cvx::Solution sol;
    if (F64SW) cvxl::lane_phase_f64(pv, o, sol, a.Z ? a.Z + b * 55 : nullptr, handoff_at, ws + b * 56, cvx::LdsStore{lds_const + threadIdx.x});
    else cvxl::lane_phase(pv, o, sol, a.Z ? a.Z + b * 55 : nullptr, handoff_at, ws + b * 56, cvx::LdsStore{lds_const + threadIdx.x});
    if (sol.status == -1) {
        const int q = atomicAdd(qcount, 1);
        qentries[q] = (int32_t)b;
        return;
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) a.R[b * 9 + i] = sol.R[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) a.t[b * 3 + i] = sol.t[i];
    a.status[b] = sol.status;
    if (a.iters) a.iters[b] = sol.iters;
    if (a.cost) { a.cost[2 * b] = sol.cost; a.cost[2 * b + 1] = sol.dobj; }
    if (a.work) { a.work[2 * b] = sol.rank; a.work[2 * b + 1] = sol.sweeps; }
}

void launch_lane2(bool f64_sweeps, unsigned grid, unsigned block, void *stream, const BatchArgs &a, const cvx::Opts &o, int handoff_at, int32_t *qcount,
                  int32_t *qentries, double *ws)
{
    hipStream_t s = (hipStream_t)stream;
    if (f64_sweeps) hipLaunchKernelGGL(solve_lane2_kernel<true>, dim3(grid), dim3(block), 0, s, a, o, handoff_at, qcount, qentries, ws);
    else hipLaunchKernelGGL(solve_lane2_kernel<false>, dim3(grid), dim3(block), 0, s, a, o, handoff_at, qcount, qentries, ws);
}

} // namespace cvxb
