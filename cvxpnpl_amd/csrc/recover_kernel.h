// recover_kernel.h -- rank > 1 multi-solution recovery on the device (cvxpnpl.py:507 -> :221-343 -> :156-218).
//
// One lane per problem; lanes whose problem is not flagged CVXPNPL_RANK_GT1 leave at once.  The flagged ones are rare
// (0.2 - 24 % of minimal RANSAC hypotheses, every planar scene) and the work is branchy scalar code (a 10x10 symmetric
// eigen-solve, 21 quadratic forms, a quartic): it runs out of registers and scratch on purpose -- what matters is that a
// batch needs no D2H copy of Z and no host thread per problem.  Same source as the host path: recover_core.h.
#pragma once
#include <hip/hip_runtime.h>

#include "recover_core.h"

namespace cvxr {

__global__ void __launch_bounds__(64) recover_multi_kernel(int64_t batch, const int32_t *status, const double *Z55, const double *B27, const double *Q45,
                                                           double *R_out, double *t_out, int32_t *n_poses)
{
    const int64_t b = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (b >= batch) return;
    if (status && status[b] != cvx::ST_RANK_GT1) { n_poses[b] = 0; return; }
    double R[36], t[12];
    for (int i = 0; i < 36; ++i) R[i] = NAN;
    for (int i = 0; i < 12; ++i) t[i] = NAN;
    const int n = recover_multi(Z55 + b * 55, B27 + b * 27, Q45 ? Q45 + b * 45 : nullptr, R, t);
    for (int i = 0; i < 36; ++i) R_out[b * 36 + i] = R[i];
    for (int i = 0; i < 12; ++i) t_out[b * 12 + i] = t[i];
    n_poses[b] = n;
}

} // namespace cvxr
