// cvxpnpl_hip.hip -- HIP kernels (gfx950) and the C ABI of include/cvxpnpl_amd.h.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

#include <map>
#include <memory>
#include <mutex>
#include <vector>
#include <utility>

#include "../../include/cvxpnpl_amd.h"
#include "problem_io.h"
#include "solver_core.h"
#include "lane_core.h"
#include "batch_args.h"
#include "wave_kernel.h"
#include "quad_kernel.h"
#include "ipm_quad.h"
#include "score_kernel.h"
#include "assemble_kernel.h"
#include "synth_kernel.h"
#include "recover_kernel.h"

namespace {

using cvxb::BatchArgs; // (batch_args.h: shared with lane_kernel.hip)

#ifdef CVXPNPL_EXPERIMENTS // the general scalar core on lanes: experiment builds only (layout 10) since round 5 -- see the lane branch of launch_solve
// ---------------------------------------------------------------------------------------
// lane-per-problem: each lane owns one problem (assembly -> ADMM -> certificate -> pose) for the first
// handoff_at (1..5) iterations; 64 independent problems per wavefront, no cross-lane traffic, no LDS.
// Hybrid schedule: a lane that is not finished by then parks its iterate in ws[b] and queues b for
// resume_wave_kernel, so that one slow problem cannot hold the other 63 lanes (and the whole launch)
// for hundreds of lane-serial iterations.
template <bool DBL>
__global__ void __launch_bounds__(64) solve_lane_kernel(BatchArgs a, cvx::Opts o, int handoff_at, int32_t *qcount, int32_t *qentries, double *ws)
{
    __shared__ double lds_const[72 * 64];
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.batch) return;
    cvx::ProblemView pv = cvx::make_view(b, a.n_p, a.p2, a.p3, a.n_l, a.l2, a.l3, a.K, a.K_per_problem);
    if (a.Q45) { pv.Q45 = a.Q45 + b * 45; pv.B27 = a.B27 + b * 27; }
    cvx::Solution sol;
    double Z[55];
    // TWIN = false: the hand-off comes before iteration 6, where the twin-candidate logic would start.
    // The cost matrix and the translation map (72 doubles) live in this lane's LDS column, not in registers.
    // DBL: the eigen-solve on float64 columns (opts.f32_sweeps_until below the length of this phase; the default is packed single precision)
    cvx::solve_problem<false, cvx::LdsStore, cvx::VAR_FULL, DBL>(pv, o, sol, a.Z ? Z : nullptr, handoff_at, ws + b * 56, cvx::LdsStore{lds_const + threadIdx.x});
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
    if (a.Z) {
#pragma unroll
        for (int i = 0; i < 55; ++i) a.Z[b * 55 + i] = Z[i];
    }
}

#endif // CVXPNPL_EXPERIMENTS

// (solve_lane2_kernel, the first phase of the lane-hybrid schedule: lane_kernel.hip, a translation unit of its own -- cvxb::launch_lane2)

// results of one shard as the 13-doubles-per-pose records the multi-GPU gather exchanges: R (9, row-major), t (3), status
__global__ void __launch_bounds__(256) pack_kernel(int64_t batch, const double *R, const double *t, const int32_t *status, double *out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= batch * 13) return;
    const int64_t b = i / 13;
    const int k = (int)(i - b * 13);
    out[i] = k < 9 ? R[b * 9 + k] : (k < 12 ? t[b * 3 + (k - 9)] : (double)status[b]);
}

__global__ void __launch_bounds__(64) assemble_kernel(BatchArgs a, double *Bout, double *Qout)
{
    int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.batch) return;
    cvx::ProblemView pv = cvx::make_view(b, a.n_p, a.p2, a.p3, a.n_l, a.l2, a.l3, a.K, a.K_per_problem);
    double B[27], Q9[45];
    bool ok = cvx::assemble(pv, B, Q9);
    for (int i = 0; i < 27; ++i) Bout[b * 27 + i] = ok ? B[i] : NAN;
    if (Qout)
        for (int i = 0; i < 45; ++i) Qout[b * 45 + i] = ok ? Q9[i] : NAN;
}

// Assembly for subsets of ONE scene: problem b takes the correspondences m with mask[b][m] != 0 (the refit of a RANSAC consensus set:
// the mask is what cvxs::score_kernel wrote, so that the size of the set never has to travel to the host).  One lane per problem; the
// Gram sums are the ones of cvx::assemble, taken about the same kind of centre (median of the subset's first three points: any centre is
// exact).  Fewer than three correspondences give a singular N^T N: NaN, as cvx::assemble reports it.
__global__ void __launch_bounds__(64) assemble_subsets_kernel(int64_t batch, int n_corr, const double *s2, const double *s3, const uint8_t *mask, const double *K,
                                                              double *Bout, double *Qout, int32_t *count)
{
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    double Kc[9], Ki[9], det;
    for (int i = 0; i < 9; ++i) Kc[i] = K[i];
    cvx::inv3(Kc, Ki, det);
    cvx::Gram g;
    cvx::gram_zero(g);
    // centre of the Gram sums: the median of the subset's OWN first three correspondences (round 5 took the scene's first three whether
    // selected or not: a non-finite or far-away outlier among them spoilt every subset, also those that mask it out -- advisor)
    const uint8_t *mk = mask + b * n_corr;
    double c[3] = {0.0, 0.0, 0.0}, first3[9];
    int nf = 0;
    for (int m = 0; m < n_corr && nf < 3; ++m)
        if (mk[m]) {
#pragma unroll
            for (int k = 0; k < 3; ++k) // (static indices: the array stays in registers)
                if (nf == k) { first3[3 * k] = s3[3 * m]; first3[3 * k + 1] = s3[3 * m + 1]; first3[3 * k + 2] = s3[3 * m + 2]; }
            ++nf;
        }
    if (nf > 0) cvx::shift_centre(nf, first3, 0, nullptr, c);
    int n = 0;
    for (int m = 0; m < n_corr; ++m) {
        if (!mk[m]) continue;
        cvx::gram_add_point(g, Ki, s2[2 * m], s2[2 * m + 1], s3[3 * m] - c[0], s3[3 * m + 1] - c[1], s3[3 * m + 2] - c[2]);
        ++n;
    }
    double B[27], Q9[45];
    bool ok = n >= 3 && cvx::gram_finish(g, B, Q9) && (det == det) && det != 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) B[i * 9 + 3 * j + i] += c[j];
    for (int i = 0; i < 27; ++i) Bout[b * 27 + i] = ok ? B[i] : NAN;
    for (int i = 0; i < 45; ++i) Qout[b * 45 + i] = ok ? Q9[i] : NAN;
    if (count) count[b] = n;
}

// Scratch of the hybrid schedules, one per (device, stream): [0, 256) the counters of the two queues (resume: ints 0..2,
// rescue: ints 16..18), then the entries of the resume queue and of the rescue queue
// (int32, -1 = empty; batch + RESUME_GRID_MAX of them each) and the parked iterates [batch][56] doubles (only the
// slots of parked problems are touched).  Either the library's own allocation (grow-only while in use, freed by
// cvxpnpl_release_workspace) or memory the caller registered with cvxpnpl_set_workspace (e.g. from torch's
// caching allocator).  The queue is self-cleaning (cvxw::resume_wave_kernel): it is initialised once.
// Offsets depend on the CAPACITY (problems) of the allocation, not on the batch of a launch, so that launches of
// different sizes agree on where the queue ends and the iterates begin.  The parked region is sized for the slot stride
// of the schedule that asked for the most (lane: 56 doubles per problem, quad: 240).
struct Workspace { void *ptr = nullptr; size_t bytes = 0; int64_t cap = 0; size_t parked_bytes = 0; bool owned = true; };
std::mutex g_ws_mutex;
std::map<std::pair<int, void *>, Workspace> g_ws;
thread_local char g_err[512] = "";
thread_local int g_last_layout = 0; // the layout the last solve of this thread ran (cvxpnpl_last_layout)

// One solve is two or three dependent launches that share the queue and the parked-iterate buffer of their (device, stream):
// host threads calling on the SAME stream (ctypes drops the GIL; torch's default stream is shared by every thread) must not
// interleave them (A.first, B.first, A.resume: B would overwrite A's parked slots and A's resume kernel would drain B's queue
// entries with A's pointers), and the workspace must not be freed / regrown between a thread's pointer fetch and its launches.
// So the launches of a solve, from get_workspace to the last kernel, run under the mutex of their (device, stream); threads on
// different streams do not contend.  (Round 2 had dropped the global launch mutex of round 1: the advisor's finding.)
std::mutex g_lm_mutex;
std::map<std::pair<int, void *>, std::unique_ptr<std::mutex>> g_launch_mutexes; // never erased: a mutex may be held while its workspace goes
std::mutex &launch_mutex(int dev, void *stream)
{
    std::lock_guard<std::mutex> lock(g_lm_mutex);
    std::unique_ptr<std::mutex> &p = g_launch_mutexes[std::make_pair(dev, stream)];
    if (!p) p.reset(new std::mutex);
    return *p;
}

size_t queue_entries_bytes(int64_t cap) { return ((size_t)(cap + cvxw::RESUME_GRID_MAX) * sizeof(int32_t) + 255) & ~(size_t)255; }
size_t hybrid_queue_bytes(int64_t cap) { return 256 + 2 * queue_entries_bytes(cap); }
size_t hybrid_ws_bytes(int64_t cap) { return hybrid_queue_bytes(cap) + (size_t)cap * cvxw::RS_FULL * sizeof(double); } // any schedule
int64_t hybrid_capacity(size_t bytes) // largest capacity whose layout fits (any schedule)
{
    if (bytes < hybrid_ws_bytes(1)) return 0;
    int64_t cap = (int64_t)(bytes / (2 * sizeof(int32_t) + cvxw::RS_FULL * sizeof(double))) + 1; // an upper bound, then down to the first fit
    while (cap > 0 && hybrid_ws_bytes(cap) > bytes) --cap;
    return cap;
}

// counter zero, every queue entry -1 (0xFF bytes), on the stream; the parked iterates need no initialisation
bool init_workspace(void *p, int64_t cap, void *stream)
{
    if (hipMemsetAsync(p, 0xFF, hybrid_queue_bytes(cap), (hipStream_t)stream) != hipSuccess) return false;
    return hipMemsetAsync(p, 0, 256, (hipStream_t)stream) == hipSuccess;
}

struct WsView { int32_t *count, *entries, *rq_count, *rq_entries; double *parked; };

bool get_workspace(int64_t batch, int stride, void *stream, WsView &v)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { snprintf(g_err, sizeof(g_err), "cvxpnpl: hipGetDevice failed"); return false; }
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    Workspace &w = g_ws[std::make_pair(dev, stream)];
    const size_t need_parked = (size_t)batch * stride * sizeof(double);
    if (w.cap < batch || w.parked_bytes < need_parked) {
        if (!w.owned) {
            snprintf(g_err, sizeof(g_err), "cvxpnpl: the registered workspace holds %lld problems, the launch has %lld (cvxpnpl_workspace_bytes)",
                     (long long)w.cap, (long long)batch);
            return false;
        }
        const int64_t cap = w.cap > batch ? w.cap : batch;
        const size_t parked = w.parked_bytes > need_parked ? w.parked_bytes : need_parked;
        if (w.ptr) { (void)hipStreamSynchronize((hipStream_t)stream); (void)hipFree(w.ptr); }
        w.ptr = nullptr; w.bytes = 0; w.cap = 0; w.parked_bytes = 0;
        const size_t bytes = hybrid_queue_bytes(cap) + parked;
        if (hipMalloc(&w.ptr, bytes) != hipSuccess || !init_workspace(w.ptr, cap, stream)) {
            if (w.ptr) (void)hipFree(w.ptr);
            w.ptr = nullptr;
            snprintf(g_err, sizeof(g_err), "cvxpnpl: workspace allocation failed (%zu bytes)", bytes);
            return false;
        }
        w.bytes = bytes; w.cap = cap; w.parked_bytes = parked;
    }
    v.count = (int32_t *)w.ptr;
    v.entries = (int32_t *)((char *)w.ptr + 256);
    v.rq_count = v.count + 16;
    v.rq_entries = (int32_t *)((char *)w.ptr + 256 + queue_entries_bytes(w.cap));
    v.parked = (double *)((char *)w.ptr + hybrid_queue_bytes(w.cap));
    return true;
}

void launch_wave(int64_t wgrid, hipStream_t s, const cvxw::WaveArgs &w, const cvx::Opts &o)
{
    if (o.variant == cvx::VAR_RC) hipLaunchKernelGGL(cvxw::solve_wave_kernel<cvx::VAR_RC>, dim3((unsigned)wgrid), dim3(64 * cvxw::WPB), 0, s, w, o);
    else hipLaunchKernelGGL(cvxw::solve_wave_kernel<cvx::VAR_FULL>, dim3((unsigned)wgrid), dim3(64 * cvxw::WPB), 0, s, w, o);
}

void launch_resume(int64_t rgrid, hipStream_t s, const cvxw::WaveArgs &w, const cvx::Opts &o, int32_t *count, int32_t *entries, const double *ws, bool full)
{
    cvxw::ResumeArgs ra;
    ra.a = w; ra.o = o; ra.count_p = count; ra.entries = entries; ra.ws = ws;
    ra.ws_stride = full ? cvxw::RS_FULL : cvxw::RS_LANE; ra.ws_full = full ? 1 : 0;
    ra.count2 = nullptr; ra.entries2 = nullptr; ra.grid1 = 0;
    if (o.variant == cvx::VAR_RC) hipLaunchKernelGGL(cvxw::resume_wave_kernel_rc, dim3((unsigned)rgrid), dim3(64), 0, s, ra);
    else hipLaunchKernelGGL(cvxw::resume_wave_kernel, dim3((unsigned)rgrid), dim3(64), 0, s, ra);
}

// last launch of a solve with opts.rescue_from in force: the problems the other kernels gave up on (w.rq_count / w.rq_entries) and,
// in the quad layout (count != null), the parked problems of the resume queue as well -- one launch for both
void launch_rescue(int64_t batch, hipStream_t s, const cvxw::WaveArgs &w, const cvx::Opts &o, int32_t *count = nullptr, int32_t *entries = nullptr,
                   const double *ws = nullptr)
{
    const int64_t rgrid = batch < cvxw::RESUME_GRID_MAX ? batch : cvxw::RESUME_GRID_MAX;
    cvxw::ResumeArgs ra;
    ra.a = w; ra.o = o; ra.count_p = w.rq_count; ra.entries = w.rq_entries; ra.ws = ws; ra.ws_stride = cvxw::RS_FULL; ra.ws_full = 1;
    ra.count2 = count; ra.entries2 = entries; ra.grid1 = (int)rgrid;
    if (o.variant == cvx::VAR_RC) hipLaunchKernelGGL(cvxw::rescue_wave_kernel_rc, dim3((unsigned)(count ? 2 * rgrid : rgrid)), dim3(64), 0, s, ra);
    else hipLaunchKernelGGL(cvxw::rescue_wave_kernel, dim3((unsigned)(count ? 2 * rgrid : rgrid)), dim3(64), 0, s, ra);
}

// split interior-point path: the rescue queue through cvxi::ipm_quad_kernel (four solves per wavefront, ipm_quad.h) into the resume queue
// (a launch of the resume kernel follows)
void launch_ipm(int64_t batch, hipStream_t s, const cvxw::WaveArgs &w, const cvx::Opts &o, int32_t *count, int32_t *entries)
{
    const int64_t groups = (batch + 3) / 4;
    const int64_t grid = groups < cvxi::IPMQ_GRID_MAX ? groups : cvxi::IPMQ_GRID_MAX;
    cvxi::IpmQuadArgs ia;
    ia.batch = batch; ia.rho = o.rho; ia.rho_tail = o.rho_tail; ia.tail_from = o.tail_from;
    ia.rq_count = w.rq_count; ia.rq_entries = w.rq_entries; ia.count = count; ia.entries = entries; ia.ws = w.rq_ws; ia.stride = w.rq_stride;
    ia.entries_cap = (int)(batch + cvxw::RESUME_GRID_MAX < 0x7fffffffLL ? batch + cvxw::RESUME_GRID_MAX : 0x7fffffffLL);
    ia.qs_in = nullptr; ia.z_out = nullptr; ia.s_out = nullptr; ia.gap_out = nullptr; ia.it_out = nullptr;
    if (o.variant == cvx::VAR_RC) hipLaunchKernelGGL(cvxi::ipm_quad_kernel<cvx::VAR_RC>, dim3((unsigned)grid), dim3(64), 0, s, ia);
    else hipLaunchKernelGGL(cvxi::ipm_quad_kernel<cvx::VAR_FULL>, dim3((unsigned)grid), dim3(64), 0, s, ia);
}

int set_err(const char *what, hipError_t e)
{
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return -2;
}

cvx::Opts to_core(const cvxpnpl_opts_t *opts)
{
    cvx::Opts o = cvx::default_opts();
    if (opts) {
        o.eps = opts->eps; o.max_iters = opts->max_iters; o.rho = opts->rho; o.alpha = opts->alpha;
        o.first_check = opts->first_check; o.check_every = opts->check_every; o.res_tol = opts->res_tol;
        o.jacobi_sweeps = opts->jacobi_sweeps; o.jacobi_tol = opts->jacobi_tol; o.warm_start = opts->warm_start; o.rho_tail = opts->rho_tail; o.tail_from = opts->tail_from;
        o.variant = opts->variant;
        o.adapt_every = opts->adapt_every; o.adapt_from = opts->adapt_from; o.adapt_mu = opts->adapt_mu; o.adapt_tau = opts->adapt_tau;
        o.stall_from = opts->stall_from; o.stall_lam = opts->stall_lam; o.stall_res = opts->stall_res; o.stall_drop = opts->stall_drop;
        o.rescue_from = opts->rescue_from;
        o.f32_sweeps_until = opts->f32_sweeps_until;
        o.sweep_schedule = opts->sweep_schedule != 0;
        o.dual_shift = opts->dual_shift < 0.0 ? (opts->variant == CVXPNPL_VARIANT_RC ? 0.006 : cvx::DUAL_SHIFT_DEFAULT) : opts->dual_shift; // (rc, 50 k problems: 17.3 M poses/s without, 17.3 / 18.4 / 18.3 M with 0.015 / 0.006 / 0.001)
        o.dual_refine = opts->dual_refine != 0;
    }
    if (o.f32_sweeps_until < 0) o.f32_sweeps_until = cvx::F32_SWEEPS_DEFAULT;
    return o;
}

} // namespace

// diagnostics: a copy with W bytes per lane and a known byte count, against which bench.py calibrates rocprofv3's
// FETCH_SIZE / WRITE_SIZE for this path's access widths (the MI355X guide calibrates only wide streaming reads)
namespace cvxd {
template <int W>
__global__ void __launch_bounds__(256) calibration_copy_kernel(const char *src, char *dst, int64_t nbytes)
{
    const int64_t n = nbytes / W;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (W == 4) reinterpret_cast<uint32_t *>(dst)[i] = reinterpret_cast<const uint32_t *>(src)[i];
        if (W == 8) reinterpret_cast<double *>(dst)[i] = reinterpret_cast<const double *>(src)[i];
        if (W == 16) reinterpret_cast<double2 *>(dst)[i] = reinterpret_cast<const double2 *>(src)[i];
    }
}
} // namespace cvxd

extern "C" {

void cvxpnpl_default_opts(cvxpnpl_opts_t *opts)
{
    cvx::Opts o = cvx::default_opts();
    opts->eps = o.eps; opts->max_iters = o.max_iters; opts->rho = o.rho; opts->alpha = o.alpha;
    opts->first_check = 0 /* by layout, see cvxpnpl_amd.h */; opts->check_every = o.check_every; opts->res_tol = o.res_tol;
    opts->jacobi_sweeps = o.jacobi_sweeps; opts->jacobi_tol = o.jacobi_tol; opts->warm_start = o.warm_start; opts->rho_tail = o.rho_tail; opts->tail_from = o.tail_from; opts->lane_iters = -1; opts->layout = CVXPNPL_LAYOUT_AUTO; opts->variant = CVXPNPL_VARIANT_FULL;
    opts->adapt_every = o.adapt_every; opts->adapt_from = o.adapt_from; opts->adapt_mu = o.adapt_mu; opts->adapt_tau = o.adapt_tau;
    opts->stall_from = o.stall_from; opts->stall_lam = o.stall_lam; opts->stall_res = o.stall_res; opts->stall_drop = o.stall_drop;
    opts->rescue_from = o.rescue_from;
    opts->f32_sweeps_until = -1;
    opts->sweep_schedule = 1;
    opts->dual_shift = -1.0; /* by variant */
    opts->dual_refine = -1;
    opts->struct_size = (uint32_t)sizeof(cvxpnpl_opts_t);
}

size_t cvxpnpl_opts_size(void) { return sizeof(cvxpnpl_opts_t); }

int cvxpnpl_last_layout(void) { return g_last_layout; }

static int check_args(int64_t batch, int32_t n_p, const double *p2, const double *p3, int32_t n_l, const double *l2,
                      const double *l3, const double *K)
{
    if (batch < 0 || n_p < 0 || n_l < 0 || (n_p == 0 && n_l == 0) || !K || (n_p > 0 && (!p2 || !p3)) || (n_l > 0 && (!l2 || !l3))) {
        snprintf(g_err, sizeof(g_err), "cvxpnpl: bad arguments (batch=%lld n_p=%d n_l=%d)", (long long)batch, n_p, n_l);
        return -1;
    }
    return 0;
}

// the launches of one solve: layout policy + kernels (shared by the correspondence entry and the cost entry)
static int launch_solve(const BatchArgs &a, const cvxpnpl_opts_t *opts, void *stream)
{
    const int64_t batch = a.batch;
    if (!a.R || !a.t || !a.status) { snprintf(g_err, sizeof(g_err), "cvxpnpl: R, t and status outputs are required"); return -1; }
    if (opts && opts->struct_size != (uint32_t)sizeof(cvxpnpl_opts_t)) {
        // a caller built against another revision of cvxpnpl_opts_t: refuse rather than read fields that are not there
        snprintf(g_err, sizeof(g_err), "cvxpnpl: options block of %u bytes, this library's cvxpnpl_opts_t has %zu (cvxpnpl_default_opts / cvxpnpl_opts_size)",
                 opts->struct_size, sizeof(cvxpnpl_opts_t));
        return -1;
    }
    if (opts && (opts->max_iters < 1 || opts->f32_sweeps_until < -1 || opts->f32_sweeps_until > cvx::F32_SWEEPS_DEFAULT || !(opts->rho > 0) || !(opts->eps > 0) || opts->check_every < 1 || opts->first_check < 0 ||
                 (opts->variant != CVXPNPL_VARIANT_FULL && opts->variant != CVXPNPL_VARIANT_RC) || opts->adapt_every < 0 ||
                 (opts->adapt_every > 0 && !(opts->adapt_mu >= 1.0 && opts->adapt_tau > 1.0)) || opts->rescue_from < -1 || !((opts->dual_shift >= 0.0 && opts->dual_shift <= 1.0) || opts->dual_shift == -1.0) || opts->dual_refine < -1 || opts->dual_refine > 1)) {
        snprintf(g_err, sizeof(g_err), "cvxpnpl: bad options");
        return -1;
    }
    // layouts: the public enum only.  The experiment layouts of rounds 2-4 (9: quad iterations only at three wavefronts per SIMD, 10: the lane
    // schedule on the general scalar core, 11-13: the tail experiments of profiles/r04/tail_experiments.txt) exist in experiment builds
    // (-DCVXPNPL_EXPERIMENTS, tools/experiments/build_experiments.sh) and nowhere else
#ifdef CVXPNPL_EXPERIMENTS
    const bool layout_known = !opts || (opts->layout >= CVXPNPL_LAYOUT_AUTO && opts->layout <= CVXPNPL_LAYOUT_PENTA) || (opts->layout >= 9 && opts->layout <= 13);
#else
    const bool layout_known = !opts || (opts->layout >= CVXPNPL_LAYOUT_AUTO && opts->layout <= CVXPNPL_LAYOUT_PENTA);
#endif
    if (!layout_known) { snprintf(g_err, sizeof(g_err), "cvxpnpl: bad options (layout %d is not one of CVXPNPL_LAYOUT_*)", opts->layout); return -1; }
    cvx::Opts o = to_core(opts);
    if (!opts) o.first_check = 0; // (by layout, below)
    int cur_dev = 0;
    if (hipGetDevice(&cur_dev) != hipSuccess) { snprintf(g_err, sizeof(g_err), "cvxpnpl: hipGetDevice failed"); return -2; }
    std::lock_guard<std::mutex> launch_lock(launch_mutex(cur_dev, stream)); // (see launch_mutex)
    hipStream_t s = (hipStream_t)stream;
    const int block = 64;
    int64_t grid = (batch + block - 1) / block;
    if (grid > 0x7fffffffLL) { snprintf(g_err, sizeof(g_err), "cvxpnpl: batch too large for one launch"); return -1; }
    int layout = opts ? opts->layout : CVXPNPL_LAYOUT_AUTO;
    // AUTO, by launch size (measured on one MI355X, M poses/s, PnP N = 10, one launch stream; wave / quad: profiles/r02/layout_sweep.txt,
    // quad / lane-hybrid with the register-budgeted first phase: profiles/r03/layout_sweep2.txt, two problem sets per size):
    //   wave / quad      2 k: 19.7 / 17.3    5 k: 27.9 / 34.1    10 k: 33.8 / 51.7
    //   quad / lane     10 k: 51.6, 49.1 / 36.7, 41.0    16 k: 52.0, 73.5 / 48.7, 63.3    20 k: 74.5, 72.8 / 79.6, 73.4
    //                   24 k: 74.5, 75.6 / 95.4, 86.8    32 k: 87.2, 87.7 / 115.0, 124.8   125 k: 113 / 247
    // * below 2560 problems a wavefront per problem: every SIMD gets work and a finished problem frees its slot at once;
    // * from there four problems per wavefront (one per DPP row): 2.5x fewer instructions per problem;
    // * from 20 000 the lane-hybrid schedule (64 problems per wavefront for the first lane_iters iterations): fewest instructions
    //   per problem, but it needs ~20 k problems to give every SIMD a wavefront (round 2, general scalar core: crossover 24 576).
    //   (with every sweep in float64 the crossover is the same: quad / lane 16 k: 58.0 / 57.0, 20 k: 59.0 / 59.6, 24 k: 62.2 / 75.9, 32 k: 67.9 / 99.0 -- profiles/r04/f64_layout_crossover.txt)
    // The problems the quad phase leaves open are finished by the same wavefront, those of the lane phase by a second kernel,
    // one per wavefront in both cases.
    // Minimal problems (four correspondences; the cost seam does not say): 18 iterations on average and a fifth of them beyond 32 --
    // the lane-hybrid schedule would park nearly all of them for the one-problem-per-wavefront phase.  They stay four per wavefront
    // for 24 iterations instead (first attempt after 7 -- round 4: 17, and their survivors are queued: minimal_queued below), like the rc variant: 50 k problems 11.2 -> 12.4 M poses/s (lane_iters 16 / 24 /
    // 32 / 40: 12.1 / 12.4 / 11.7-12.2 / 12.3; five correspondences and more: the lane-hybrid schedule wins, 36.8 against 33.6 M at N = 5).
#ifdef CVXPNPL_EXPERIMENTS
    const bool layout_auto_like = layout == CVXPNPL_LAYOUT_AUTO || layout == 11 || layout == 12 || layout == 13;
#else
    const bool layout_auto_like = layout == CVXPNPL_LAYOUT_AUTO;
#endif
#ifdef CVXPNPL_EXPERIMENTS // (tuning builds: opts.lane_iters sets the length of the first phase of four-point problems)
    const bool minimal = !a.Q45 && a.n_p + a.n_l <= 4 && o.variant == cvx::VAR_FULL && layout_auto_like && batch >= 2560 && o.max_iters > 24;
#else
    const bool minimal = !a.Q45 && a.n_p + a.n_l <= 4 && o.variant == cvx::VAR_FULL && layout_auto_like && batch >= 2560 && o.max_iters > 24 &&
                         (!opts || opts->lane_iters <= 0);
#endif
    if (layout == CVXPNPL_LAYOUT_AUTO) layout = batch < 2560 ? CVXPNPL_LAYOUT_WAVE : ((batch < 20000 || minimal) ? CVXPNPL_LAYOUT_QUAD : CVXPNPL_LAYOUT_LANE);
    // The 16-equality variant (benchmarks/toolkit/methods/rc.py): wave-per-problem and, since round 3, the quad schedule (the
    // constraint set is a template parameter of the kernels); the lane kernels and the interior-point path are built for the full set.
    const bool rc = o.variant == cvx::VAR_RC;
    if (rc && (layout == CVXPNPL_LAYOUT_LANE || layout == CVXPNPL_LAYOUT_PENTA || layout == 9 || layout == 10 || layout == 12 || layout == 13)) layout = CVXPNPL_LAYOUT_QUAD; // (9-13: experiment builds only)
    cvxw::WaveArgs w;
    w.batch = batch; w.n_p = a.n_p; w.n_l = a.n_l; w.K_per_problem = a.K_per_problem;
    w.p2 = a.p2; w.p3 = a.p3; w.l2 = a.l2; w.l3 = a.l3; w.K = a.K;
    w.R = a.R; w.t = a.t; w.cost = a.cost; w.Z = a.Z; w.status = a.status; w.iters = a.iters; w.work = a.work;
    w.Q45 = a.Q45; w.B27 = a.B27;
    w.rq_count = nullptr; w.rq_entries = nullptr;
    int quad_iters = opts ? opts->lane_iters : -1;
    // rc: the weaker relaxation certifies after ~21 iterations instead of 5 (N = 10): a longer first phase, first attempt later
    if (quad_iters <= 0 && rc) quad_iters = 36; // (profiles/r03/rc_tune.txt: 28 / 36 / 44 within 1 %)
    if (quad_iters <= 0 && minimal) quad_iters = 24;
    if (quad_iters <= 0) quad_iters = 7; // measured (4 problem sets at 10 k, same box): 5: 0.242 ms, 7: 0.233, 8: 0.235, 10: 0.240; 24 k: 7 = 10 (round 1, second phase as a call: 8-12)
    if (quad_iters > 48) quad_iters = 48; // (rc: up to 48 -- still inside the wave kernel's own single-precision window of 64)
    if (quad_iters > 16 && !rc && !minimal) quad_iters = 16; // (a caller's lane_iters: the quad phase is the YOUNG part of a solve, its slow survivors belong to the wave-per-problem phase.
    // Minimal problems and the rc variant run 24 / 36-48 iterations here, in single-precision sweeps by default -- inside the window of 64 that
    // opts.f32_sweeps_until allows and the host experiment covers; device A/B against float64 sweeps: profiles/r04/f32_phase_ab.txt)
#ifdef CVXPNPL_EXPERIMENTS
    const bool lane_general = layout == 10; // experiment / A-B (tools/README.md): the lane schedule with the general scalar core (solve_lane_kernel)
    if (lane_general) layout = CVXPNPL_LAYOUT_LANE;
#else
    const bool lane_general = false;
#endif
    // the REQUEST (five problems per wavefront) and the kernel that serves it are separate: with float64 sweeps the twelve-lane geometry
    // does not exist and the request runs the sixteen-lane quad kernel -- either way the layout from here on is QUAD (round-3 advisor:
    // a PENTA request with f32_sweeps_until = 0 used to fall through to the lane branch with a workspace fetched for another stride)
    const bool penta_req = layout == CVXPNPL_LAYOUT_PENTA;
    const bool penta = penta_req && !(o.f32_sweeps_until < quad_iters); // (float64 sweeps: built for the sixteen-lane geometry only)
    if (layout == 9 || layout == 11 || layout == 12 || layout == 13 || penta_req) layout = CVXPNPL_LAYOUT_QUAD; // (9: experiment (tools/README.md): quad iterations only, 3 waves/SIMD, solve_quad_kernel<1>)
    // The register-budgeted lane kernel (lane_core.h) covers the schedule of the defaults: one attempt, right at the hand-off point, warm-started
    // eigen-solves.  Any other combination of options (first_check != lane_iters, warm_start = 0) used to fall through to the general scalar
    // core on lanes (solve_lane_kernel: 1 008-1 014 spilled registers, 2.2-2.7 KB of scratch per lane, a third of the speed) without saying
    // so; round 5 sent such a request to the wave-per-problem layout -- one problem per wavefront, also at 125 000 problems, and with the
    // attempt schedule already derived for the lane layout (advisor).  Now the layout is settled HERE, before anything is derived from it:
    // a lane request the budgeted kernel cannot serve runs the next-best schedule for its size (quad from 2 560 problems, wave below), the
    // first attempt then follows THAT layout's default, and cvxpnpl_last_layout() says what ran.
    int lane_iters = opts ? opts->lane_iters : -1;
    {
        const int fc_lane = o.first_check > 0 ? o.first_check : 6;
        if (lane_iters <= 0) lane_iters = fc_lane;
        if (lane_iters > 6) lane_iters = 6; // (see the lane branch below)
#ifndef CVXPNPL_EXPERIMENTS
        const bool budgeted = !lane_general && fc_lane == lane_iters && lane_iters >= 2 && o.warm_start != 0;
        if (layout == CVXPNPL_LAYOUT_LANE && !budgeted) {
            layout = batch >= 2560 ? CVXPNPL_LAYOUT_QUAD : CVXPNPL_LAYOUT_WAVE;
            quad_iters = 7; // (the caller's lane_iters was meant for the lane phase)
        }
#endif
    }
    if (layout == CVXPNPL_LAYOUT_QUAD && !(quad_iters >= 1 && o.max_iters > quad_iters)) layout = CVXPNPL_LAYOUT_WAVE;
    // (rc with float64 sweeps: solve_quad_kernel<0, 2, 16, true, VAR_RC> since round 5 -- until then such a request ran the wave layout)
    // First certificate attempt (0 = by layout): after 5 iterations 94 % of N = 10 problems certify, after 6 99 %.  In the lane-hybrid
    // schedule every problem that fails the first attempt is parked and resumed one per wavefront, so the later attempt pays for its
    // extra iteration: 125 k problems 157 -> 164 M poses/s, PnPL 100 k 116 -> 126 M.  The quad and wave layouts keep 5 (quad with 6, launch
    // time relative to 5 over 4 problem sets per size: 3 k 0.93, 5 k 1.07, 8 k 1.02, 10 k 0.98, 12 k 1.07, 16 k 1.03, 20 k 1.02, 24 k 0.97;
    // wave: -12 % at 2 k).
    // Four-correspondence problems in the schedule that queues its survivors (below): an attempt costs the whole wavefront two to three
    // iterations' worth -- at three wavefronts per SIMD the certificate's code is the part that spills -- and these problems need 14-20
    // iterations on average: first attempt after 17 (profiles/r04/minimal_tune*.txt, 50 k problems / config 5, M per second: first attempt
    // after 7: 13.0 / 21.3, 9: 13.5 / 22.4, 13: 14.5 / 23.5, 17: 14.9 / 24.0, 21: 14.5 / 23.9; every third iteration instead of every second: same).
    // (round 5: in both precision modes -- with float64 sweeps the kernel keeps two wavefronts per SIMD, solve_quad_kernel<2, 2, 16, true>)
    const bool minimal_queued = minimal && layout == CVXPNPL_LAYOUT_QUAD;
    const bool minimal_queued_f64 = minimal_queued && o.f32_sweeps_until < quad_iters;
    // (rc in the quad schedule: 19 instead of 11 -- profiles/r04/rc_tune_r04.txt: 50 k problems 18.3 -> 19.3 M poses/s, 10 k 9.0 -> 9.3 M; the same effect, smaller)
    if (o.first_check <= 0) o.first_check = rc ? (layout == CVXPNPL_LAYOUT_QUAD ? 19 : 11) : (minimal_queued ? 17 : (minimal ? 7 : (layout == CVXPNPL_LAYOUT_LANE ? 6 : 5))); // (rc: nothing certifies before ~10 iterations; 5 ... 15 within 3 %)
    // interior-point path for the problems still open after rescue_from iterations (ipm_wave.h): its queue lives in the workspace
    // -1 (default): by problem size.  Slow convergence is a property of minimal and near-minimal configurations
    // (profiles/r02/remaining_iters.jsonl, 100 k problems each, first-order iterations only: with N = 4 / 5 / 6 / 7 correspondences
    // 21 % / 3.8 % / 0.7 % / 0.14 % of the problems are still open after 32 iterations and 27 % / 17 % / 12 % / 6 % of those need more
    // than the ~75 iterations an interior-point solve costs, slowest 1 455 / 1 037 / 581 / 227; with N = 8 the slowest takes 99, with
    // N = 10 (1 M problems) 61, and a problem that is open after 48 finishes within the next 3-25).  A threshold below the natural tail
    // of a workload sends problems through a 0.3 ms solve they did not need and ends the launch later: 100 k problems with N = 8
    // 1.05 ms without the path, 1.35 ms with 96; 1 M with N = 10 4.46 / 4.75 ms with 96 / 32; against that 10 k problems with N = 4
    // 4.77 / 1.78 ms, N = 6 (100 k) 3.42 / 1.91 ms without / with 32, N = 7 (100 k) 1.71 / 1.49 ms without / with 64.
    if (o.rescue_from < 0) {
        const int n = a.Q45 ? 8 : a.n_p + a.n_l;
        o.rescue_from = n <= 6 ? 32 : (n == 7 ? 64 : 128);
        // rc: the weaker relaxation is tight less often, and a problem whose relaxation is not tight crawls to max_iters -- 13 of 10 000
        // N = 10 problems run all 2 500 iterations, 10 ms per launch whatever the layout (profiles/r03/rc_rate.txt) -- while the
        // typical problem certifies after ~19 iterations (median; p90 33): hand over at 48 whatever the size
        if (rc) o.rescue_from = 48; // (profiles/r03/rc_tune.txt, 50 k problems: 48 / 64 / 80 / 96 -> 2.92 / 3.10 / 3.46 / 3.63 ms, same outcomes)
    }
    const bool rescue = o.rescue_from > 0 && o.max_iters > o.rescue_from;
    // ONE workspace view per solve, fetched after the layout is settled and with the stride of the schedule that will run: a second
    // fetch with a larger stride may free and reallocate the buffer, and queue pointers taken from the first would dangle
    // (lane_iters and the budgeted / not budgeted decision: above, where the layout is settled)
    const bool lane_budgeted = !lane_general && o.first_check == lane_iters && lane_iters >= 2 && o.warm_start != 0;
    g_last_layout = (layout == CVXPNPL_LAYOUT_QUAD && penta) ? CVXPNPL_LAYOUT_PENTA : layout;
    const bool lane_hybrid = layout == CVXPNPL_LAYOUT_LANE && o.max_iters > lane_iters;
    // The interior-point path comes in two builds.  Fused (cvxw::rescue_wave_kernel: the solve compiled into a resume kernel, one launch
    // behind the first kernel) where it is a safety net -- seven correspondences and more: its queue is empty in nearly every launch and
    // one more (empty) launch would cost the 10 k-problem step 2 %.  Split (cvxw::ipm_wave_kernel + the plain resume kernel) where
    // problems really go through it -- at most six correspondences, the 16-equality variant: a fifth of the four-point problems.
#ifndef CVXPNPL_FUSED_IPM
    const bool split = rescue && (rc || (!a.Q45 && a.n_p + a.n_l <= 5)); // (measured, split / fused, M poses/s: N = 4 50 k 17.96 / 15.07, config 5 25.6 / 24.7, N = 5 100 k 37.3 / 35.8, N = 6 125 k 73.1 / 75.5: profiles/r05/ipm_quad_ab.txt)
#else
    const bool split = false; // (A/B builds: every workload through the fused kernel, as until round 4)
#endif
    WsView wv = {nullptr, nullptr, nullptr, nullptr, nullptr};
    const int ws_stride = layout == CVXPNPL_LAYOUT_QUAD ? cvxw::RS_FULL : ((lane_hybrid || split) ? cvxw::RS_LANE : 0);
    if (rescue || layout == CVXPNPL_LAYOUT_QUAD || lane_hybrid) {
        if (!get_workspace(batch, ws_stride, stream, wv)) return -2;
    }
    w.rq_ws = nullptr; w.rq_stride = 0;
    if (rescue) { w.rq_count = wv.rq_count; w.rq_entries = wv.rq_entries; }
    if (split) { w.rq_ws = wv.parked; w.rq_stride = ws_stride; }
    const int64_t rgrid_all = batch < cvxw::RESUME_GRID_MAX ? batch : cvxw::RESUME_GRID_MAX;
    if (layout == CVXPNPL_LAYOUT_QUAD) {
        // four problems per wavefront for the first quad_iters iterations, survivors resumed one per wavefront
        // (a wavefront finishes its own survivors; only planar scenes, recognised before the first iteration,
        // are queued for the resume kernel behind it -- an empty queue costs that launch a few microseconds.
        // The resume kernel leaves the queue counter at zero for the next launch: no memset per call.)
        int32_t *count = wv.count, *entries = wv.entries;
        double *ws = wv.parked;
        const int64_t qgrid = penta ? (batch + 4) / 5 : (batch + 3) / 4;
        cvxq::QuadArgs qa;
        qa.a = w; qa.o = o; qa.handoff_at = quad_iters; qa.qcount = count; qa.qentries = entries; qa.ws = ws;
#ifdef CVXPNPL_EXPERIMENTS // 9: iterations only (tools/phase_a_time.sh); round 4, profiles/r04/tail_experiments.txt: survivors queued (layouts 11: three, 12: two wavefronts per SIMD), extras queued (13)
        if (opts && opts->layout == 9) hipLaunchKernelGGL((cvxq::solve_quad_kernel<1, 3>), dim3((unsigned)qgrid), dim3(64), 0, s, qa);
        else if (opts && opts->layout == 11 && rc) hipLaunchKernelGGL((cvxq::solve_quad_kernel<2, 3, 16, false, cvx::VAR_RC>), dim3((unsigned)qgrid), dim3(64), 0, s, qa);
        else if (opts && opts->layout == 11) hipLaunchKernelGGL((cvxq::solve_quad_kernel<2, 3>), dim3((unsigned)qgrid), dim3(64), 0, s, qa);
        else if (opts && opts->layout == 13) hipLaunchKernelGGL((cvxq::solve_quad_kernel<3, 2>), dim3((unsigned)qgrid), dim3(64), 0, s, qa);
        else if (opts && opts->layout == 12) hipLaunchKernelGGL((cvxq::solve_quad_kernel<2, 2>), dim3((unsigned)qgrid), dim3(64), 0, s, qa);
        else
#endif
        if (minimal_queued_f64) hipLaunchKernelGGL((cvxq::solve_quad_kernel<2, 2, 16, true>), dim3((unsigned)qgrid), dim3(64), 0, s, qa);
        else if (minimal_queued)
            // Four-correspondence problems: every survivor of the 24-iteration first phase goes to the queue of the launch behind this one
            // instead of being finished by its own wavefront -- 59 % of these wavefronts end with survivors, most of which are headed for
            // the interior-point path anyway, and without the wave-per-problem code the kernel runs three wavefronts per SIMD (168 registers).
            // Measured (profiles/r04/quad_mode2_minimal.txt): 50 k four-point problems 12.4 -> 13.1 M poses/s, config 5 19.8 -> 21.3 M
            // hypotheses/s (with the first attempt after 17 iterations, above: 14.9 / 24.0 M); the same schedule LOSES on the N = 10 launches, whose few survivors then start late (tail_experiments.txt).
            hipLaunchKernelGGL((cvxq::solve_quad_kernel<2, 3>), dim3((unsigned)qgrid), dim3(64), 0, s, qa);
        else if (penta) hipLaunchKernelGGL((cvxq::solve_quad_kernel<0, 2, 12>), dim3((unsigned)qgrid), dim3(64), 0, s, qa);
        else if (rc && o.f32_sweeps_until < quad_iters) hipLaunchKernelGGL((cvxq::solve_quad_kernel<0, 2, 16, true, cvx::VAR_RC>), dim3((unsigned)qgrid), dim3(64), 0, s, qa);
        else if (rc) hipLaunchKernelGGL((cvxq::solve_quad_kernel<0, 2, 16, false, cvx::VAR_RC>), dim3((unsigned)qgrid), dim3(64), 0, s, qa);
        else if (o.f32_sweeps_until < quad_iters) hipLaunchKernelGGL((cvxq::solve_quad_kernel<0, 2, 16, true>), dim3((unsigned)qgrid), dim3(64), 0, s, qa); // float64 sweeps (A/B mode)
        else hipLaunchKernelGGL((cvxq::solve_quad_kernel<0, 2>), dim3((unsigned)qgrid), dim3(64), 0, s, qa);
        const int64_t rgrid = batch < cvxw::RESUME_GRID_MAX ? batch : cvxw::RESUME_GRID_MAX;
        if (split) {
            // the wavefronts' own slow survivors (rescue queue) through the interior-point kernel into the resume queue, behind the planar
            // scenes parked there; a parked problem that reaches rescue_from in the resume kernel takes the second round
            for (int round = 0; round < 2; ++round) {
                launch_ipm(batch, s, w, o, count, entries);
                launch_resume(rgrid, s, w, o, count, entries, ws, true);
            }
        } else if (rescue) launch_rescue(batch, s, w, o, count, entries, ws); // (both queues in one launch)
        else launch_resume(rgrid, s, w, o, count, entries, ws, true);
    } else if (layout == CVXPNPL_LAYOUT_WAVE) {
        int64_t wgrid = (batch + cvxw::WPB - 1) / cvxw::WPB;
        if (wgrid > 0x7fffffffLL) { snprintf(g_err, sizeof(g_err), "cvxpnpl: batch too large for one launch"); return -1; }
        launch_wave(wgrid, s, w, o);
    } else {
        // hand-off point of the hybrid schedule (<= 0: default): right after the first certificate attempt (opts.first_check: 6
        // by default in this layout) -- later is slower (round 1, first attempt at 5: hand-off at 5 / 6 / 7: 110 / 105 / 100 M at 125 k).
        // (lane_iters: above.)  The lane phase never runs past 6 iterations: from then on the few problems still open are the
        // slow / ambiguous ones (twin candidates, tails), which belong to the wave-per-problem kernel -- one of
        // them would hold 63 idle lanes, so the lane kernel is built without that logic.
        if (lane_hybrid) {
            // hybrid: lanes for the first lane_iters iterations, survivors resumed one per wavefront
            int32_t *count = wv.count, *entries = wv.entries;
            double *ws = wv.parked;
            const bool budgeted = lane_budgeted;
            if (budgeted) cvxb::launch_lane2(!(o.f32_sweeps_until >= lane_iters), (unsigned)grid, (unsigned)block, (void *)s, a, o, lane_iters, count, entries, ws); // (true: float64 sweeps)
#ifdef CVXPNPL_EXPERIMENTS
            else if (o.f32_sweeps_until < lane_iters) hipLaunchKernelGGL(solve_lane_kernel<true>, dim3((unsigned)grid), dim3(block), 0, s, a, o, lane_iters, count, entries, ws); // float64 sweeps (A/B mode)
            else hipLaunchKernelGGL(solve_lane_kernel<false>, dim3((unsigned)grid), dim3(block), 0, s, a, o, lane_iters, count, entries, ws);
#endif
            const int64_t rgrid = batch < cvxw::RESUME_GRID_MAX ? batch : cvxw::RESUME_GRID_MAX;
            launch_resume(rgrid, s, w, o, count, entries, ws, false);
        } else {
            // fewer iterations allowed than the lane phase would run: the wave kernel does the whole solve
            int64_t wgrid = (batch + cvxw::WPB - 1) / cvxw::WPB;
            launch_wave(wgrid, s, w, o);
        }
    }
    if (split && layout != CVXPNPL_LAYOUT_QUAD) {
        launch_ipm(batch, s, w, o, wv.count, wv.entries);
        launch_resume(rgrid_all, s, w, o, wv.count, wv.entries, wv.parked, false);
    } else if (rescue && layout != CVXPNPL_LAYOUT_QUAD) launch_rescue(batch, s, w, o);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err("solve kernel launch", e);
    return 0;
}

int cvxpnpl_solve_batch(int64_t batch, int32_t n_p, const double *d_pts_2d, const double *d_pts_3d, int32_t n_l,
                        const double *d_line_2d, const double *d_line_3d, const double *d_K, int32_t K_per_problem,
                        const cvxpnpl_opts_t *opts, double *d_R, double *d_t, int32_t *d_status, int32_t *d_iters,
                        double *d_cost, double *d_Z, int32_t *d_work, void *stream)
{
    if (batch == 0) return 0; /* empty batch: nothing to do (pointers may be NULL) */
    if (check_args(batch, n_p, d_pts_2d, d_pts_3d, n_l, d_line_2d, d_line_3d, d_K)) return -1;
    BatchArgs a;
    a.batch = batch; a.n_p = n_p; a.n_l = n_l; a.K_per_problem = K_per_problem;
    a.p2 = d_pts_2d; a.p3 = d_pts_3d; a.l2 = d_line_2d; a.l3 = d_line_3d; a.K = d_K;
    a.R = d_R; a.t = d_t; a.cost = d_cost; a.Z = d_Z; a.status = d_status; a.iters = d_iters; a.work = d_work;
    a.Q45 = nullptr; a.B27 = nullptr;
    return launch_solve(a, opts, stream);
}

int cvxpnpl_solve_cost_batch(int64_t batch, const double *d_Q45, const double *d_B27, const cvxpnpl_opts_t *opts, double *d_R,
                             double *d_t, int32_t *d_status, int32_t *d_iters, double *d_cost, double *d_Z, int32_t *d_work, void *stream)
{
    if (batch == 0) return 0;
    if (batch < 0 || !d_Q45 || !d_B27) { snprintf(g_err, sizeof(g_err), "cvxpnpl_solve_cost_batch: bad arguments (batch=%lld)", (long long)batch); return -1; }
    BatchArgs a;
    memset(&a, 0, sizeof(a));
    a.batch = batch;
    a.R = d_R; a.t = d_t; a.cost = d_cost; a.Z = d_Z; a.status = d_status; a.iters = d_iters; a.work = d_work;
    a.Q45 = d_Q45; a.B27 = d_B27;
    return launch_solve(a, opts, stream);
}

int cvxpnpl_assemble_batch(int64_t batch, int32_t n_p, const double *d_pts_2d, const double *d_pts_3d, int32_t n_l,
                           const double *d_line_2d, const double *d_line_3d, const double *d_K, int32_t K_per_problem,
                           double *d_B, double *d_Q45, void *stream)
{
    if (check_args(batch, n_p, d_pts_2d, d_pts_3d, n_l, d_line_2d, d_line_3d, d_K) || !d_B) return -1;
    if (batch == 0) return 0;
    BatchArgs a;
    memset(&a, 0, sizeof(a));
    a.batch = batch; a.n_p = n_p; a.n_l = n_l; a.K_per_problem = K_per_problem;
    a.p2 = d_pts_2d; a.p3 = d_pts_3d; a.l2 = d_line_2d; a.l3 = d_line_3d; a.K = d_K;
    const int block = 64;
    hipLaunchKernelGGL(assemble_kernel, dim3((unsigned)((batch + block - 1) / block)), dim3(block), 0, (hipStream_t)stream, a, d_B, d_Q45);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err("assemble_kernel launch", e);
    return 0;
}

int cvxpnpl_assemble_subsets(int64_t batch, int32_t n_corr, const double *d_scene_2d, const double *d_scene_3d, const uint8_t *d_mask, const double *d_K,
                             double *d_B, double *d_Q45, int32_t *d_count, void *stream)
{
    if (batch < 0 || n_corr < 1 || (batch + 63) / 64 > 0x7fffffffLL || (batch > 0 && (!d_scene_2d || !d_scene_3d || !d_mask || !d_K || !d_B || !d_Q45))) {
        snprintf(g_err, sizeof(g_err), "cvxpnpl_assemble_subsets: bad arguments");
        return -1;
    }
    if (batch == 0) return 0;
    hipLaunchKernelGGL(assemble_subsets_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, (hipStream_t)stream, batch, (int)n_corr, d_scene_2d, d_scene_3d,
                       d_mask, d_K, d_B, d_Q45, d_count);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : set_err("assemble_subsets_kernel launch", e);
}

size_t cvxpnpl_assemble_large_scratch_bytes(int64_t batch, int32_t n_p, int32_t n_l)
{
    if (batch <= 0 || n_p < 0 || n_l < 0) return 0;
    return (size_t)batch * cvxa::asm_blocks((int64_t)n_p + 2 * (int64_t)n_l, batch) * 60 * sizeof(double);
}

int cvxpnpl_assemble_large_batch(int64_t batch, int32_t n_p, const double *d_pts_2d, const double *d_pts_3d, int32_t n_l,
                                 const double *d_line_2d, const double *d_line_3d, const double *d_K, int32_t K_per_problem,
                                 double *d_B, double *d_Q45, void *d_scratch, size_t scratch_bytes, void *stream)
{
    if (check_args(batch, n_p, d_pts_2d, d_pts_3d, n_l, d_line_2d, d_line_3d, d_K) || !d_B) return -1;
    if (batch == 0) return 0;
    if (batch > 65535) { snprintf(g_err, sizeof(g_err), "cvxpnpl_assemble_large_batch: at most 65535 problems per call (got %lld)", (long long)batch); return -1; }
    const size_t need = cvxpnpl_assemble_large_scratch_bytes(batch, n_p, n_l);
    if (!d_scratch || scratch_bytes < need) {
        snprintf(g_err, sizeof(g_err), "cvxpnpl_assemble_large_batch: scratch of %zu bytes needed (cvxpnpl_assemble_large_scratch_bytes), got %zu", need, scratch_bytes);
        return -1;
    }
    cvxa::AsmArgs a;
    a.batch = batch; a.n_p = n_p; a.n_l = n_l; a.K_per_problem = K_per_problem;
    a.nblk = cvxa::asm_blocks((int64_t)n_p + 2 * (int64_t)n_l, batch);
    a.p2 = d_pts_2d; a.p3 = d_pts_3d; a.l2 = d_line_2d; a.l3 = d_line_3d; a.K = d_K;
    a.partial = (double *)d_scratch;
    a.Bout = d_B; a.Qout = d_Q45;
    if (cvxa::asm_tpb((int64_t)n_p + 2 * (int64_t)n_l, batch) == cvxa::ASM_TPB)
        hipLaunchKernelGGL(cvxa::assemble_large_kernel<cvxa::ASM_TPB>, dim3((unsigned)a.nblk, (unsigned)batch), dim3(cvxa::ASM_TPB), 0, (hipStream_t)stream, a);
    else // short problems or many of them: one wavefront per workgroup (cvxa::asm_tpb)
        hipLaunchKernelGGL(cvxa::assemble_large_kernel<cvxa::ASM_TPB_NARROW>, dim3((unsigned)a.nblk, (unsigned)batch), dim3(cvxa::ASM_TPB_NARROW), 0, (hipStream_t)stream, a);
    if (a.nblk > 1) // several workgroups per problem: their partial sums are added in a fixed order by a second kernel
        hipLaunchKernelGGL(cvxa::assemble_finish_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, (hipStream_t)stream, a, d_B, d_Q45);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err("assemble_large_kernel launch", e);
    return 0;
}

int cvxpnpl_synth_batch(int64_t batch, int32_t n_p, int32_t n_l, double sigma, uint64_t seed, const double *d_K, double *d_pts_2d,
                        double *d_pts_3d, double *d_line_2d, double *d_line_3d, double *d_R_gt, double *d_t_gt, void *stream)
{
    if (batch < 0 || n_p < 0 || n_l < 0 || !d_K || !(sigma >= 0.0) || (n_p > 0 && (!d_pts_2d || !d_pts_3d)) || (n_l > 0 && (!d_line_2d || !d_line_3d))) {
        snprintf(g_err, sizeof(g_err), "cvxpnpl_synth_batch: bad arguments");
        return -1;
    }
    if (batch == 0) return 0;
    cvxg::SynthArgs a;
    a.batch = batch; a.n_p = n_p; a.n_l = n_l; a.sigma = sigma; a.length = 0.6 /* synth.py:55 */; a.seed = seed; a.K = d_K;
    a.p2 = d_pts_2d; a.p3 = d_pts_3d; a.l2 = d_line_2d; a.l3 = d_line_3d; a.R_gt = d_R_gt; a.t_gt = d_t_gt;
    const int64_t nrec = (int64_t)n_p + 2 * (int64_t)n_l, n = batch * (nrec > 0 ? nrec : 1), grid = (n + 255) / 256;
    if (grid > 0x7fffffffLL) { snprintf(g_err, sizeof(g_err), "cvxpnpl_synth_batch: too many records for one launch"); return -1; }
    hipLaunchKernelGGL(cvxg::synth_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : set_err("synth_kernel launch", e);
}

int cvxpnpl_pose_errors(int64_t batch, const double *d_R_gt, const double *d_t_gt, const double *d_R, const double *d_t, double *d_ang_deg,
                        double *d_trans, void *stream)
{
    if (batch < 0 || !d_R_gt || !d_t_gt || !d_R || !d_t || !d_ang_deg || !d_trans) { snprintf(g_err, sizeof(g_err), "cvxpnpl_pose_errors: bad arguments"); return -1; }
    if (batch == 0) return 0;
    hipLaunchKernelGGL(cvxg::pose_error_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, batch, d_R_gt, d_t_gt, d_R, d_t,
                       d_ang_deg, d_trans);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : set_err("pose_error_kernel launch", e);
}

int cvxpnpl_disambiguate(int64_t batch, const double *d_R_all, const double *d_t_all, const int32_t *d_n_poses, const double *d_K,
                         const double *d_R_gt, const double *d_t_gt, const double *d_support, int32_t n_support, double *d_R, double *d_t,
                         int32_t *d_index, void *stream)
{
    if (batch < 0 || !d_R_all || !d_t_all || !d_n_poses || !d_K || !d_R_gt || !d_t_gt || (n_support > 0 && !d_support) || n_support < 0 || !d_R || !d_t || !d_index) {
        snprintf(g_err, sizeof(g_err), "cvxpnpl_disambiguate: bad arguments");
        return -1;
    }
    if (batch == 0) return 0;
    hipLaunchKernelGGL(cvxg::disambiguate_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, batch, d_R_all, d_t_all,
                       d_n_poses, d_K, d_R_gt, d_t_gt, d_support, n_support, d_R, d_t, d_index);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : set_err("disambiguate_kernel launch", e);
}

int cvxpnpl_recover_multi_device(int64_t batch, const int32_t *d_status, const double *d_Z55, const double *d_B27, const double *d_Q45,
                                 double *d_R_out, double *d_t_out, int32_t *d_n_poses, void *stream)
{
    if (batch < 0 || !d_Z55 || !d_B27 || !d_R_out || !d_t_out || !d_n_poses) { snprintf(g_err, sizeof(g_err), "cvxpnpl_recover_multi_device: bad arguments"); return -1; }
    if (batch == 0) return 0;
    hipLaunchKernelGGL(cvxr::recover_multi_kernel, dim3((unsigned)((batch + 63) / 64)), dim3(64), 0, (hipStream_t)stream, batch, d_status, d_Z55, d_B27,
                       d_Q45, d_R_out, d_t_out, d_n_poses);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : set_err("recover_multi_kernel launch", e);
}

// Ordering between two streams of one device without a HIP event on the producing stream: the producer enqueues a one-lane kernel
// that stores `value` to a flag in device memory (release, device scope); the consumer enqueues a one-wavefront kernel that sleeps
// and polls until the flag has reached it.  (A hipEventRecord between two kernels of a stream costs that stream ~17 us on this
// stack -- rocprofv3 trace of bench.py --force-dist: 17.6 us between the end of a solve and the start of the next against 2 us
// without the event; this pair costs it ~2 us.)  The flag only ever grows (the store is an atomic max); d_flag points to TWO words, the
// flag and the wait's "gave up" mark.  The wait is BOUNDED and therefore fails open: a consumer must read d_flag[1] after every
// synchronisation at which it consumes what the waits ordered, and discard those results when it is set (include/cvxpnpl_amd.h).
__global__ void stream_write_value_kernel(unsigned long long *flag, unsigned long long value)
{
    __threadfence();
    // max, not store: write kernels enqueued on different streams may complete out of order and the flag must never move backwards
    __hip_atomic_fetch_max(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// closed = true (cvxpnpl_stream_wait_value): a wait that gives up HOLDS its stream until the host has acknowledged the give-up
// (cvxpnpl_stream_wait_gave_up clears flag[1]); nobody acknowledging within ack_polls (~2^25 polls, half a minute) is a caller bug and ends in a
// trap -- loud, never a consumer silently reading unfinished results.  closed = false (cvxpnpl_stream_wait_value_bounded): the explicit
// fail-open form of rounds 3-5.
__global__ void stream_wait_value_kernel(unsigned long long *flag, unsigned long long value, unsigned long long max_polls, int closed, unsigned long long ack_polls)
{
    // bounded (max_polls > 0): if the two streams share a hardware queue the producer's kernel sits BEHIND this one and the flag can never
    // arrive -- after max_polls polls of ~1 us (cvxpnpl_stream_wait_value: 2^18, ~0.25 s) the wait gives up and says so in flag[1]
    // (the caller falls back to an event); max_polls = 0: wait for as long as it takes
    for (unsigned long long spin = 0; max_polls == 0 || spin < max_polls; ++spin) {
        if (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= value) return;
        __builtin_amdgcn_s_sleep(32);
    }
    __hip_atomic_store(flag + 1, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!closed) return;
    for (unsigned long long spin = 0; spin < ack_polls; ++spin) {
        if (__hip_atomic_load(flag + 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) == 0ull) return; // acknowledged: the caller knows
        __builtin_amdgcn_s_sleep(64);
    }
    __builtin_trap();
}

int cvxpnpl_stream_write_value(uint64_t *d_flag, uint64_t value, void *stream)
{
    if (!d_flag) { snprintf(g_err, sizeof(g_err), "cvxpnpl_stream_write_value: null flag"); return -1; }
    hipLaunchKernelGGL(stream_write_value_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long *)d_flag, (unsigned long long)value);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : set_err("stream_write_value_kernel launch", e);
}

static int launch_wait(uint64_t *d_flag, uint64_t value, uint64_t max_polls, int closed, void *stream)
{
    if (!d_flag) { snprintf(g_err, sizeof(g_err), "cvxpnpl_stream_wait_value: null flag"); return -1; }
    hipLaunchKernelGGL(stream_wait_value_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long *)d_flag, (unsigned long long)value,
                       (unsigned long long)max_polls, closed, 1ull << 25);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : set_err("stream_wait_value_kernel launch", e);
}

int cvxpnpl_stream_wait_value_bounded(uint64_t *d_flag, uint64_t value, uint64_t max_polls, void *stream) { return launch_wait(d_flag, value, max_polls, 0, stream); }

int cvxpnpl_stream_wait_value(uint64_t *d_flag, uint64_t value, void *stream) { return launch_wait(d_flag, value, 1ull << 18, 1, stream); }

// The give-up word is read (and cleared) on a stream of the library's own, so that the call works while a closed wait holds `stream`.
static hipStream_t ack_stream()
{
    static std::mutex mu;
    static hipStream_t s[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> g(mu);
    if (!s[dev] && hipStreamCreateWithFlags(&s[dev], hipStreamNonBlocking) != hipSuccess) s[dev] = nullptr;
    return s[dev];
}

int cvxpnpl_stream_wait_gave_up(uint64_t *d_flag, int32_t clear, void *stream)
{
    if (!d_flag) { snprintf(g_err, sizeof(g_err), "cvxpnpl_stream_wait_gave_up: null flag"); return -1; }
    hipStream_t as = ack_stream();
    if (!as) { snprintf(g_err, sizeof(g_err), "cvxpnpl_stream_wait_gave_up: no stream for the acknowledgement"); return -2; }
    for (;;) {
        // (the state of `stream` is sampled BEFORE the word: when the stream was already idle no wait of it can set the word afterwards)
        const hipError_t q = hipStreamQuery((hipStream_t)stream);
        if (q != hipSuccess && q != hipErrorNotReady) return set_err("cvxpnpl_stream_wait_gave_up (stream)", q);
        unsigned long long w = 0;
        hipError_t e = hipMemcpyAsync(&w, d_flag + 1, sizeof(w), hipMemcpyDeviceToHost, as);
        if (e == hipSuccess) e = hipStreamSynchronize(as);
        if (e == hipSuccess && w != 0 && clear) {
            e = hipMemsetAsync(d_flag + 1, 0, sizeof(w), as); // releases a closed wait that is holding `stream`
            if (e == hipSuccess) e = hipStreamSynchronize(as);
        }
        if (e != hipSuccess) return set_err("cvxpnpl_stream_wait_gave_up", e);
        if (w != 0) return 1;
        if (q == hipSuccess) return 0; // everything the waits ordered is complete, and none gave up
        usleep(50);
    }
}

int cvxpnpl_pack_results(int64_t batch, const double *d_R, const double *d_t, const int32_t *d_status, double *d_packed, void *stream)
{
    if (batch < 0 || !d_R || !d_t || !d_status || !d_packed) { snprintf(g_err, sizeof(g_err), "cvxpnpl_pack_results: bad arguments"); return -1; }
    if (batch == 0) return 0;
    const int64_t n = batch * 13, grid = (n + 255) / 256;
    if (grid > 0x7fffffffLL) { snprintf(g_err, sizeof(g_err), "cvxpnpl_pack_results: batch too large for one launch"); return -1; }
    hipLaunchKernelGGL(pack_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, batch, d_R, d_t, d_status, d_packed);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err("pack_kernel launch", e);
    return 0;
}

int cvxpnpl_select_best(int64_t n_hyp, const int32_t *d_count, const double *d_R, const double *d_t, const int32_t *d_status, const double *d_K,
                        int32_t n_corr, const double *d_pts_2d, const double *d_pts_3d, double thresh_px, double *d_out_R, double *d_out_t,
                        int32_t *d_head, uint8_t *d_mask, void *stream)
{
    if (n_hyp < 1 || n_hyp > 0x7fffffffLL || n_corr < 0 || !d_count || !d_R || !d_t || !d_status || !d_K || !d_out_R || !d_out_t || !d_head ||
        (n_corr > 0 && (!d_pts_2d || !d_pts_3d || !d_mask)) || !(thresh_px > 0.0)) {
        snprintf(g_err, sizeof(g_err), "cvxpnpl_select_best: bad arguments");
        return -1;
    }
    cvxs::SelectArgs a{n_hyp, d_count, d_R, d_t, d_status, d_K, n_corr, d_pts_2d, d_pts_3d, thresh_px, d_out_R, d_out_t, d_head, d_mask};
    hipLaunchKernelGGL(cvxs::select_best_kernel, dim3(1), dim3(cvxs::SELECT_BLOCK), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : set_err("select_best_kernel launch", e);
}

int cvxpnpl_refit_update(const double *d_fit_R, const double *d_fit_t, const int32_t *d_fit_status, const int32_t *d_fit_count, const double *d_K,
                         int32_t n_corr, const double *d_pts_2d, const double *d_pts_3d, double thresh_px, double *d_R, double *d_t, int32_t *d_head,
                         uint8_t *d_mask, void *stream)
{
    if (n_corr < 0 || !d_fit_R || !d_fit_t || !d_fit_status || !d_fit_count || !d_K || !d_R || !d_t || !d_head ||
        (n_corr > 0 && (!d_pts_2d || !d_pts_3d || !d_mask)) || !(thresh_px > 0.0)) {
        snprintf(g_err, sizeof(g_err), "cvxpnpl_refit_update: bad arguments");
        return -1;
    }
    cvxs::RefitArgs a{d_fit_R, d_fit_t, d_fit_status, d_fit_count, d_K, n_corr, d_pts_2d, d_pts_3d, thresh_px, d_R, d_t, d_head, d_mask};
    hipLaunchKernelGGL(cvxs::refit_update_kernel, dim3(1), dim3(cvxs::SELECT_BLOCK), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : set_err("refit_update_kernel launch", e);
}

int cvxpnpl_score_hypotheses(int64_t n_hyp, const double *d_R, const double *d_t, const int32_t *d_status, uint32_t usable_mask,
                             const double *d_K, int32_t n_corr, const double *d_pts_2d, const double *d_pts_3d, double thresh_px,
                             int32_t *d_count, uint8_t *d_mask, void *stream)
{
    if (n_hyp < 0 || n_corr < 0 || !d_R || !d_t || !d_K || !d_count || (n_corr > 0 && (!d_pts_2d || !d_pts_3d)) || !(thresh_px > 0.0)) {
        snprintf(g_err, sizeof(g_err), "cvxpnpl_score_hypotheses: bad arguments");
        return -1;
    }
    if (n_hyp == 0) return 0;
    const int64_t grid = (n_hyp + cvxs::SCORE_BLOCK - 1) / cvxs::SCORE_BLOCK;
    if (grid > 0x7fffffffLL) { snprintf(g_err, sizeof(g_err), "cvxpnpl_score_hypotheses: too many hypotheses for one launch"); return -1; }
    cvxs::ScoreArgs a;
    a.n_hyp = n_hyp; a.R = d_R; a.t = d_t; a.status = d_status; a.usable_mask = usable_mask; a.K = d_K;
    a.n_corr = n_corr; a.p2 = d_pts_2d; a.p3 = d_pts_3d; a.thresh = thresh_px; a.count = d_count; a.mask = d_mask;
    hipLaunchKernelGGL(cvxs::score_kernel, dim3((unsigned)grid), dim3(cvxs::SCORE_BLOCK), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_err("score_kernel launch", e);
    return 0;
}

int cvxpnpl_sample_minimal_sets(int64_t n_hyp, int32_t n_corr, const double *d_scene_2d, const double *d_scene_3d, int32_t k, uint64_t seed,
                                int32_t *d_idx, double *d_pts_2d, double *d_pts_3d, void *stream)
{
    if (n_hyp == 0) return 0; /* a no-op, as the header says: the (empty) outputs may be NULL */
    if (n_hyp < 0 || k < 1 || k > cvxs::SAMPLE_KMAX || n_corr < k || !d_scene_2d || !d_scene_3d || !d_pts_2d || !d_pts_3d) {
        snprintf(g_err, sizeof(g_err), "cvxpnpl_sample_minimal_sets: bad arguments (n_hyp=%lld n_corr=%d k=%d)", (long long)n_hyp, n_corr, k);
        return -1;
    }
    const int64_t grid = (n_hyp + 255) / 256;
    if (grid > 0x7fffffffLL) { snprintf(g_err, sizeof(g_err), "cvxpnpl_sample_minimal_sets: too many hypotheses for one launch"); return -1; }
    cvxs::SampleArgs a;
    a.n_hyp = n_hyp; a.n_corr = n_corr; a.k = k; a.seed = seed; a.s2 = d_scene_2d; a.s3 = d_scene_3d; a.idx = d_idx; a.p2 = d_pts_2d; a.p3 = d_pts_3d;
    hipLaunchKernelGGL(cvxs::sample_sets_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, a);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : set_err("sample_sets_kernel launch", e);
}

size_t cvxpnpl_workspace_bytes(int64_t max_batch) { return max_batch > 0 ? hybrid_ws_bytes(max_batch) : 0; }

int cvxpnpl_set_workspace(void *d_workspace, size_t bytes, void *stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { snprintf(g_err, sizeof(g_err), "cvxpnpl: hipGetDevice failed"); return -2; }
    std::lock_guard<std::mutex> launch_lock(launch_mutex(dev, stream)); // no solve of this stream is between its launches
    std::lock_guard<std::mutex> lock(g_ws_mutex);
    Workspace &w = g_ws[std::make_pair(dev, stream)];
    if (w.ptr && w.owned) { (void)hipStreamSynchronize((hipStream_t)stream); (void)hipFree(w.ptr); }
    w = Workspace();
    if (!d_workspace) return 0; // back to the library's own allocation
    const int64_t cap = hybrid_capacity(bytes);
    if (cap <= 0) { snprintf(g_err, sizeof(g_err), "cvxpnpl_set_workspace: %zu bytes hold no problem (cvxpnpl_workspace_bytes)", bytes); g_ws.erase(std::make_pair(dev, stream)); return -1; }
    if (!init_workspace(d_workspace, cap, stream)) { g_ws.erase(std::make_pair(dev, stream)); return set_err("cvxpnpl_set_workspace", hipGetLastError()); }
    w.ptr = d_workspace; w.bytes = bytes; w.cap = cap; w.parked_bytes = bytes - hybrid_queue_bytes(cap); w.owned = false;
    return 0;
}

int cvxpnpl_release_workspace(void *stream, int32_t all_streams)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { snprintf(g_err, sizeof(g_err), "cvxpnpl: hipGetDevice failed"); return -2; }
    std::vector<void *> streams;
    {
        std::lock_guard<std::mutex> lock(g_ws_mutex);
        for (auto &kv : g_ws)
            if (kv.first.first == dev && (all_streams || kv.first.second == stream)) streams.push_back(kv.first.second);
    }
    for (void *st : streams) { // lock order as in a solve: the stream's launch mutex, then the workspace table
        std::lock_guard<std::mutex> launch_lock(launch_mutex(dev, st));
        std::lock_guard<std::mutex> lock(g_ws_mutex);
        auto it = g_ws.find(std::make_pair(dev, st));
        if (it == g_ws.end()) continue;
        if (it->second.ptr && it->second.owned) { (void)hipStreamSynchronize((hipStream_t)st); (void)hipFree(it->second.ptr); }
        g_ws.erase(it);
    }
    return 0;
}

int cvxpnpl_ipm_batch(int64_t batch, const double *d_Qs55, int32_t variant, double *d_Z100, double *d_S100, double *d_gap, int32_t *d_iters, void *stream)
{
    if (batch < 0 || !d_Qs55 || !d_Z100 || !d_S100 || !d_gap || !d_iters || (variant != CVXPNPL_VARIANT_FULL && variant != CVXPNPL_VARIANT_RC) || (batch + 3) / 4 > 0x7fffffffLL) {
        snprintf(g_err, sizeof(g_err), "cvxpnpl_ipm_batch: bad arguments");
        return -1;
    }
    if (batch == 0) return 0;
    cvxi::IpmQuadArgs ia;
    memset(&ia, 0, sizeof(ia));
    ia.batch = batch; ia.rho = 1.0; ia.rho_tail = 1.0;
    ia.qs_in = d_Qs55; ia.z_out = d_Z100; ia.s_out = d_S100; ia.gap_out = d_gap; ia.it_out = d_iters;
    const unsigned grid = (unsigned)((batch + 3) / 4);
    if (variant == CVXPNPL_VARIANT_RC) hipLaunchKernelGGL(cvxi::ipm_quad_kernel<cvx::VAR_RC>, dim3(grid), dim3(64), 0, (hipStream_t)stream, ia);
    else hipLaunchKernelGGL(cvxi::ipm_quad_kernel<cvx::VAR_FULL>, dim3(grid), dim3(64), 0, (hipStream_t)stream, ia);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : set_err("ipm_quad_kernel launch", e);
}

int cvxpnpl_calibration_copy(const void *d_src, void *d_dst, int64_t nbytes, int32_t bytes_per_lane, void *stream)
{
    if (!d_src || !d_dst || nbytes <= 0 || (bytes_per_lane != 4 && bytes_per_lane != 8 && bytes_per_lane != 16) || nbytes % 16) {
        snprintf(g_err, sizeof(g_err), "cvxpnpl_calibration_copy: bad arguments");
        return -1;
    }
    const dim3 grid(4096), block(256);
    if (bytes_per_lane == 4) hipLaunchKernelGGL(cvxd::calibration_copy_kernel<4>, grid, block, 0, (hipStream_t)stream, (const char *)d_src, (char *)d_dst, nbytes);
    if (bytes_per_lane == 8) hipLaunchKernelGGL(cvxd::calibration_copy_kernel<8>, grid, block, 0, (hipStream_t)stream, (const char *)d_src, (char *)d_dst, nbytes);
    if (bytes_per_lane == 16) hipLaunchKernelGGL(cvxd::calibration_copy_kernel<16>, grid, block, 0, (hipStream_t)stream, (const char *)d_src, (char *)d_dst, nbytes);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : set_err("calibration_copy_kernel launch", e);
}

void *cvxpnpl_event_create(void)
{
    hipEvent_t ev;
    if (hipEventCreate(&ev) != hipSuccess) return nullptr;
    return (void *)ev;
}
int cvxpnpl_event_record(void *event, void *stream)
{
    hipError_t e = hipEventRecord((hipEvent_t)event, (hipStream_t)stream);
    return e == hipSuccess ? 0 : set_err("hipEventRecord", e);
}
int cvxpnpl_event_elapsed_ms(void *start, void *stop, float *ms)
{
    hipError_t e = hipEventSynchronize((hipEvent_t)stop);
    if (e != hipSuccess) return set_err("hipEventSynchronize", e);
    e = hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop);
    return e == hipSuccess ? 0 : set_err("hipEventElapsedTime", e);
}
void cvxpnpl_event_destroy(void *event) { if (event) hipEventDestroy((hipEvent_t)event); }

const char *cvxpnpl_last_error(void) { return g_err; }
const char *cvxpnpl_version(void) { return "cvxpnpl_amd 0.1.0 (gfx950)"; }
int cvxpnpl_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

} // extern "C"

#ifdef CVXW_PROFILE
// diagnostics build only (tools/phase_profile.py): read and reset the per-phase cycle counters
extern "C" int cvxpnpl_debug_phase_cycles(unsigned long long *out32, int reset)
{
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(cvxw::g_phase_cycles), 32 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(cvxw::g_phase_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
