// batch_args.h -- the argument block of the lane-per-problem kernels, shared by the two translation units of the library
// (cvxpnpl_hip.hip: every other kernel and the launch policy; lane_kernel.hip: solve_lane2_kernel, built with its own choice of the
// reciprocal-root refinement, see there).
#pragma once
#include <stdint.h>

#include "solver_core.h"

namespace cvxb {

struct BatchArgs {
    int64_t batch;
    int n_p, n_l, K_per_problem;
    const double *p2, *p3, *l2, *l3, *K;
    double *R, *t, *cost, *Z;
    int32_t *status, *iters, *work;
    const double *Q45, *B27; // cost entry (cvxpnpl_solve_cost_batch)
};

// solve_lane2_kernel<f64_sweeps> on `stream` (lane_kernel.hip)
void launch_lane2(bool f64_sweeps, unsigned grid, unsigned block, void *stream, const BatchArgs &a, const cvx::Opts &o, int handoff_at, int32_t *qcount,
                  int32_t *qentries, double *ws);

} // namespace cvxb
