// host_recover.cpp -- host entry points of the cold path: all poses of a rank > 1 SDP solution
// (cvxpnpl.py:507 -> :221-343 -> :156-218).  The mathematics is recover_core.h, shared with the device kernel.
#include <algorithm>
#include <thread>
#include <vector>

#include "../../include/cvxpnpl_amd.h"
#include "recover_core.h"

extern "C" int cvxpnpl_recover_multi(const double *Z55, const double *B27, const double *Q45, double *R_out, double *t_out)
{
    return cvxr::recover_multi(Z55, B27, Q45, R_out, t_out);
}

// Batched form of the cold path: every problem of the batch whose status is CVXPNPL_RANK_GT1 (or every
// problem when status == NULL) goes through cvxpnpl_recover_multi; the work is split over host threads
// (n_threads <= 0: hardware concurrency).  n_poses[i] = 0 for problems that were skipped.
extern "C" int cvxpnpl_recover_multi_batch(int64_t batch, const int32_t *status, const double *Z55, const double *B27, const double *Q45,
                                           double *R_out, double *t_out, int32_t *n_poses, int32_t n_threads)
{
    if (batch < 0 || !Z55 || !B27 || !R_out || !t_out || !n_poses) return -1;
    int nt = n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency();
    if (nt < 1) nt = 1;
    if ((int64_t)nt > batch) nt = batch > 0 ? (int)batch : 1;
    auto work = [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            if (status && status[i] != CVXPNPL_RANK_GT1) { n_poses[i] = 0; continue; }
            n_poses[i] = cvxpnpl_recover_multi(Z55 + i * 55, B27 + i * 27, Q45 ? Q45 + i * 45 : nullptr, R_out + i * 36, t_out + i * 12);
        }
    };
    if (nt == 1) { work(0, batch); return 0; }
    std::vector<std::thread> pool;
    const int64_t chunk = (batch + nt - 1) / nt;
    for (int k = 0; k < nt; ++k) {
        const int64_t lo = k * chunk, hi = std::min<int64_t>(batch, lo + chunk);
        if (lo < hi) pool.emplace_back(work, lo, hi);
    }
    for (auto &th : pool) th.join();
    return 0;
}
