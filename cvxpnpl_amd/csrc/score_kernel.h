// score_kernel.h -- consensus scoring of pose hypotheses (RANSAC consumer of the batched solver,
// SURVEY.md section 8(f) row 3; the reference has no RANSAC, BASELINE config 5 does).
//
// One lane per hypothesis; the scene (M correspondences, 40 B each) is staged through LDS a tile at a
// time and broadcast to the lanes (ds_read of a wave-uniform address).  Per hypothesis the kernel reads
// 96 B (R, t) + 4 B (status) and writes 4 B: with the scene in LDS the traffic is the hypotheses, once.
// Correspondence m is an inlier of hypothesis h when it lies in front of the camera and reprojects
// within thresh pixels:  X = R P + t,  (u, v, w) = K X,  X_z > 0,  |(u/w, v/w) - x| < thresh.
// A hypothesis whose pose is not finite, or whose status is not usable, scores 0.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cvxs {

constexpr int SCORE_BLOCK = 256;
constexpr int SCORE_TILE = 512; // correspondences per LDS tile (20 KB)

struct ScoreArgs {
    int64_t n_hyp;
    const double *R, *t;     // [n_hyp][9] row-major, [n_hyp][3]
    const int32_t *status;   // optional: only CERTIFIED (0) and NOT_CONVERGED-free statuses listed in usable_mask count
    uint32_t usable_mask;    // bit s set: status s is scored
    const double *K;         // [9]
    int32_t n_corr;
    const double *p2, *p3;   // [n_corr][2], [n_corr][3]
    double thresh;
    int32_t *count;          // [n_hyp]
    uint8_t *mask;           // optional [n_hyp][n_corr]
};

__global__ void __launch_bounds__(SCORE_BLOCK) score_kernel(ScoreArgs a)
{
    __shared__ double scene[SCORE_TILE * 5];
    const int64_t h = (int64_t)blockIdx.x * SCORE_BLOCK + threadIdx.x;
    const bool live = h < a.n_hyp;
    double M[12]; // K R | K t : pixel-space projection;  r2 / t2: depth row
    double r2[3] = {0, 0, 0}, t2 = 0;
    bool usable = live;
    if (live) {
        double R[9], t[3], K[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) { R[i] = a.R[h * 9 + i]; K[i] = a.K[i]; }
#pragma unroll
        for (int i = 0; i < 3; ++i) t[i] = a.t[h * 3 + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) M[i * 4 + j] = K[i * 3] * R[j] + K[i * 3 + 1] * R[3 + j] + K[i * 3 + 2] * R[6 + j];
            M[i * 4 + 3] = K[i * 3] * t[0] + K[i * 3 + 1] * t[1] + K[i * 3 + 2] * t[2];
        }
        r2[0] = R[6]; r2[1] = R[7]; r2[2] = R[8]; t2 = t[2];
        if (a.status) {
            const int32_t s = a.status[h];
            usable = s >= 0 && s < 32 && ((a.usable_mask >> s) & 1u);
        }
    }
    const double th2 = a.thresh * a.thresh;
    int cnt = 0;
    for (int base = 0; base < a.n_corr; base += SCORE_TILE) {
        const int n = a.n_corr - base < SCORE_TILE ? a.n_corr - base : SCORE_TILE;
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += SCORE_BLOCK) {
            const int m = base + i;
            scene[i * 5 + 0] = a.p3[m * 3 + 0];
            scene[i * 5 + 1] = a.p3[m * 3 + 1];
            scene[i * 5 + 2] = a.p3[m * 3 + 2];
            scene[i * 5 + 3] = a.p2[m * 2 + 0];
            scene[i * 5 + 4] = a.p2[m * 2 + 1];
        }
        __syncthreads();
        if (live) {
            for (int i = 0; i < n; ++i) {
                const double X = scene[i * 5], Y = scene[i * 5 + 1], Z = scene[i * 5 + 2];
                const double u = M[0] * X + M[1] * Y + M[2] * Z + M[3];
                const double v = M[4] * X + M[5] * Y + M[6] * Z + M[7];
                const double w = M[8] * X + M[9] * Y + M[10] * Z + M[11];
                const double depth = r2[0] * X + r2[1] * Y + r2[2] * Z + t2;
                const double du = u / w - scene[i * 5 + 3], dv = v / w - scene[i * 5 + 4];
                const bool in = usable && depth > 0.0 && (du * du + dv * dv < th2); // NaN poses compare false
                cnt += in ? 1 : 0;
                if (a.mask) a.mask[h * a.n_corr + base + i] = in ? 1 : 0;
            }
        }
    }
    if (live) a.count[h] = cnt;
}

// Minimal sets for the hypotheses of a RANSAC frame: K distinct correspondences of the scene per hypothesis, drawn uniformly (partial
// Fisher-Yates on a counter-based stream: Philox4x32-10 keyed by the seed with counter (hypothesis, 0xFFFFFFFE, draw) -- the generator of
// synth_kernel.h), and gathered straight into the [n_hyp][K][2] / [n_hyp][K][3] inputs of the solve.  One lane per hypothesis; the draw
// j picks r = j + floor(u (M - j)) and maps it through the swaps made so far (K <= 8: the swap list lives in registers).
constexpr int SAMPLE_KMAX = 8;
struct SampleArgs {
    int64_t n_hyp;
    int32_t n_corr, k;
    uint64_t seed;
    const double *s2, *s3; // scene [n_corr][2], [n_corr][3]
    int32_t *idx;          // [n_hyp][k] (optional)
    double *p2, *p3;       // [n_hyp][k][2], [n_hyp][k][3]
};
__host__ __device__ inline void sample_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t *out)
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__global__ void __launch_bounds__(256) sample_sets_kernel(SampleArgs a)
{
    const int64_t h = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (h >= a.n_hyp) return;
    int pos[SAMPLE_KMAX], val[SAMPLE_KMAX]; // positions already swapped and what sits there now
    int pick[SAMPLE_KMAX];
    const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < SAMPLE_KMAX; ++j) {
        if (j >= a.k) break;
        if ((j & 3) == 0) sample_philox((uint32_t)h, (uint32_t)((uint64_t)h >> 32), 0xFFFFFFFEu, (uint32_t)(j >> 2), k0, k1, w);
        const uint32_t span = (uint32_t)(a.n_corr - j);
        const int r = j + (int)(((uint64_t)w[j & 3] * span) >> 32); // uniform on j .. n_corr - 1 (bias < 2^-32 n_corr)
        // value at position r and at position j under the swaps so far
        int vr = r, vj = j;
#pragma unroll
        for (int m = 0; m < SAMPLE_KMAX; ++m) {
            if (m >= j) break;
            vr = pos[m] == r ? val[m] : vr;
            vj = pos[m] == j ? val[m] : vj;
        }
        pick[j] = vr;
        pos[j] = r; val[j] = vj; // position r now holds what position j held (position j is never looked at again)
#pragma unroll
        for (int m = 0; m < SAMPLE_KMAX; ++m) { // a later entry for the same position overrides an earlier one: drop the earlier
            if (m >= j) break;
            if (pos[m] == r) pos[m] = -1;
        }
    }
    for (int j = 0; j < a.k; ++j) {
        const int c = pick[j];
        if (a.idx) a.idx[h * a.k + j] = c;
        a.p2[(h * a.k + j) * 2] = a.s2[c * 2]; a.p2[(h * a.k + j) * 2 + 1] = a.s2[c * 2 + 1];
        a.p3[(h * a.k + j) * 3] = a.s3[c * 3]; a.p3[(h * a.k + j) * 3 + 1] = a.s3[c * 3 + 1]; a.p3[(h * a.k + j) * 3 + 2] = a.s3[c * 3 + 2];
    }
}

// ---- the selection of a RANSAC frame (round 6): what ~40 small torch kernels did around the arg-max ----------------------------------
// select_best_kernel: ONE block.  Arg-max of the inlier counts over all hypotheses with a deterministic tie-break (the LOWEST index among
// the best counts -- torch.argmax's choice, whatever the launch geometry), the number of certified hypotheses on the way, then the
// winner's pose, status and -- scored again against the scene by the block's lanes -- its inlier mask and count.
// head: int32[4] = { status of the pose, inliers, index of the winning hypothesis, certified hypotheses } -- the frame's one read-back.
constexpr int SELECT_BLOCK = 1024;
struct SelectArgs {
    int64_t n_hyp;
    const int32_t *count;    // [n_hyp] (score_kernel)
    const double *R, *t;     // [n_hyp][9], [n_hyp][3]
    const int32_t *status;   // [n_hyp]
    const double *K;         // [9]
    int32_t n_corr;
    const double *p2, *p3;   // scene
    double thresh;
    double *out_R, *out_t;   // [9], [3]
    int32_t *head;           // [4]
    uint8_t *mask;           // [n_corr]
};
// inliers of ONE pose over the scene, by the lanes of one block: writes mask (optional) and returns the count to every lane
__device__ inline int block_score_pose(const double *R, const double *t, const double *K, int n_corr, const double *p2, const double *p3, double thresh,
                                       uint8_t *mask, int *red /* LDS, blockDim.x / 64 + 1 ints */)
{
    double M[12];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j) M[i * 4 + j] = K[i * 3] * R[j] + K[i * 3 + 1] * R[3 + j] + K[i * 3 + 2] * R[6 + j];
        M[i * 4 + 3] = K[i * 3] * t[0] + K[i * 3 + 1] * t[1] + K[i * 3 + 2] * t[2];
    }
    const double th2 = thresh * thresh;
    int cnt = 0;
    for (int m = threadIdx.x; m < n_corr; m += blockDim.x) {
        const double X = p3[3 * m], Y = p3[3 * m + 1], Z = p3[3 * m + 2];
        const double u = M[0] * X + M[1] * Y + M[2] * Z + M[3], v = M[4] * X + M[5] * Y + M[6] * Z + M[7], w = M[8] * X + M[9] * Y + M[10] * Z + M[11];
        const double depth = R[6] * X + R[7] * Y + R[8] * Z + t[2];
        const double du = u / w - p2[2 * m], dv = v / w - p2[2 * m + 1];
        const bool in = depth > 0.0 && (du * du + dv * dv < th2); // (the arithmetic of score_kernel: the same mask, bit for bit)
        cnt += in ? 1 : 0;
        if (mask) mask[m] = in ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    int tot = 0;
    for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) tot += red[wv];
    return tot;
}
__global__ void __launch_bounds__(SELECT_BLOCK) select_best_kernel(SelectArgs a)
{
    __shared__ int red[SELECT_BLOCK / 64 + 1];
    __shared__ long long best_w[SELECT_BLOCK / 64];
    __shared__ int cert_w[SELECT_BLOCK / 64];
    // (count, index) packed so that a plain max picks the highest count and, among equals, the LOWEST index
    long long best = -1;
    int cert = 0;
    for (int64_t h = threadIdx.x; h < a.n_hyp; h += SELECT_BLOCK) {
        const long long key = ((long long)a.count[h] << 32) | (long long)(0x7fffffffLL - h);
        best = key > best ? key : best;
        cert += a.status[h] == 0 ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const long long o = __shfl_down(best, off);
        best = o > best ? o : best;
        cert += __shfl_down(cert, off);
    }
    if ((threadIdx.x & 63) == 0) { best_w[threadIdx.x >> 6] = best; cert_w[threadIdx.x >> 6] = cert; }
    __syncthreads();
    best = best_w[0]; cert = cert_w[0];
    for (int wv = 1; wv < SELECT_BLOCK / 64; ++wv) { best = best_w[wv] > best ? best_w[wv] : best; cert += cert_w[wv]; }
    const int64_t hb = a.n_hyp > 0 ? 0x7fffffffLL - (best & 0xffffffffLL) : 0;
    double R[9], t[3], K[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { R[i] = a.R[hb * 9 + i]; K[i] = a.K[i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = a.t[hb * 3 + i];
    const int n_inl = block_score_pose(R, t, K, a.n_corr, a.p2, a.p3, a.thresh, a.mask, red);
    if (threadIdx.x < 9) a.out_R[threadIdx.x] = R[threadIdx.x];
    if (threadIdx.x < 3) a.out_t[threadIdx.x] = t[threadIdx.x];
    if (threadIdx.x == 0) { a.head[0] = a.status[hb]; a.head[1] = n_inl; a.head[2] = (int32_t)hb; a.head[3] = cert; }
}
// refit_update_kernel: ONE block.  The refitted pose of the consensus set (assemble_subsets + solve at the cost seam) is scored against the
// scene and TAKEN -- pose, status, mask and count together -- when it is usable (status 0 or 2, at least four correspondences in its set)
// and keeps at least the consensus it was fitted to; otherwise everything stays.
struct RefitArgs {
    const double *fit_R, *fit_t;   // [9], [3]
    const int32_t *fit_status;     // [1]
    const int32_t *fit_cnt;        // [1] size of the set it was fitted to
    const double *K;
    int32_t n_corr;
    const double *p2, *p3;
    double thresh;
    double *io_R, *io_t;           // [9], [3]
    int32_t *head;                 // [4] (select_best_kernel)
    uint8_t *mask;                 // [n_corr]
};
__global__ void __launch_bounds__(SELECT_BLOCK) refit_update_kernel(RefitArgs a)
{
    __shared__ int red[SELECT_BLOCK / 64 + 1];
    double R[9], t[3], K[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) { R[i] = a.fit_R[i]; K[i] = a.K[i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) t[i] = a.fit_t[i];
    const int st = a.fit_status[0];
    const int n_new = block_score_pose(R, t, K, a.n_corr, a.p2, a.p3, a.thresh, nullptr, red);
    const bool take = (st == 0 || st == 2) && a.fit_cnt[0] >= 4 && n_new >= a.head[1]; // (block-uniform)
    __syncthreads();
    if (!take) return;
    (void)block_score_pose(R, t, K, a.n_corr, a.p2, a.p3, a.thresh, a.mask, red); // pose and mask change together
    if (threadIdx.x < 9) a.io_R[threadIdx.x] = R[threadIdx.x];
    if (threadIdx.x < 3) a.io_t[threadIdx.x] = t[threadIdx.x];
    if (threadIdx.x == 0) { a.head[0] = st; a.head[1] = n_new; }
}

} // namespace cvxs
