// synth_kernel.h -- the reference's benchmark toolkit on the device: synthetic problems, error metrics, pose choice.
//
//   synth_kernel         benchmarks/toolkit/suites/synth.py:27-42 (random_pose: axis U(-.5,.5)^3 normalised, angle 2 pi U,
//                        Rodrigues; t = [U-.5, U-.5, 1.6 U + .6]), :276-346 (3D points 0.6 (U - .5), pixels K (R X + t)
//                        dehomogenised + N(0, sigma^2); lines = consecutive point pairs), suite.py:17-19 (projection).
//   pose_error_kernel    suite.py:8-14 (angle of a matrix after projection onto O(3)), :22-33 (angular error in degrees of
//                        R_gt^-1 R, relative translation error).
//   disambiguate_kernel  suite.py:96-108: among the poses a solve returned, the one whose reprojection of a few support
//                        points is closest to the ground truth's.
// The distributions are the reference's; the random stream is not numpy's MT19937 (that stream is serial): a counter-based
// generator, Philox4x32-10 keyed by the seed with counter (problem, record, draw), so that every record of every problem
// is generated independently and reproducibly -- tests/test_device_toolkit.py restates the generator in numpy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "solver_core.h"

namespace cvxg {

struct U4 { uint32_t x, y, z, w; };

// Philox4x32-10 (Salmon et al., SC'11): counter c, key k
__host__ __device__ inline U4 philox4x32(U4 c, uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c.x, p1 = (uint64_t)0xCD9E8D57u * c.z;
        U4 n;
        n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
        n.y = (uint32_t)p1;
        n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
        n.w = (uint32_t)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}
// uniform in [0, 1) with 53 random bits from two words (the construction numpy's random_sample uses on MT words)
__host__ __device__ inline double u53(uint32_t a, uint32_t b) { return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0); }

struct SynthArgs {
    int64_t batch;
    int n_p, n_l;
    double sigma, length;
    uint64_t seed;
    const double *K; // [3][3]
    double *p2, *p3, *l2, *l3, *R_gt, *t_gt;
};

// pose of problem b: counter (b, 0xFFFFFFFF, 0 | 1)
__device__ __forceinline__ void synth_pose(const SynthArgs &a, int64_t b, double *R, double *t)
{
    const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
    const U4 w0 = philox4x32(U4{(uint32_t)b, (uint32_t)((uint64_t)b >> 32), 0xFFFFFFFFu, 0u}, k0, k1);
    const U4 w1 = philox4x32(U4{(uint32_t)b, (uint32_t)((uint64_t)b >> 32), 0xFFFFFFFFu, 1u}, k0, k1);
    const U4 w2 = philox4x32(U4{(uint32_t)b, (uint32_t)((uint64_t)b >> 32), 0xFFFFFFFFu, 2u}, k0, k1);
    const U4 w3 = philox4x32(U4{(uint32_t)b, (uint32_t)((uint64_t)b >> 32), 0xFFFFFFFFu, 3u}, k0, k1);
    double ax = u53(w0.x, w0.y) - 0.5, ay = u53(w0.z, w0.w) - 0.5, az = u53(w1.x, w1.y) - 0.5; // synth.py:33-34
    const double inv = 1.0 / sqrt(ax * ax + ay * ay + az * az);
    ax *= inv; ay *= inv; az *= inv;
    const double ang = 2.0 * M_PI * u53(w1.z, w1.w);                                          // synth.py:36
    const double s = sin(ang), c1 = 1.0 - cos(ang);
    // R = I + sin K + (1 - cos) K^2, K = [axis]x   (synth.py:20-24)
    const double Kx[9] = {0, -az, ay, az, 0, -ax, -ay, ax, 0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double k2 = 0;
            for (int m = 0; m < 3; ++m) k2 += Kx[i * 3 + m] * Kx[m * 3 + j];
            R[i * 3 + j] = (i == j ? 1.0 : 0.0) + s * Kx[i * 3 + j] + c1 * k2;
        }
    t[0] = u53(w2.x, w2.y) - 0.5; t[1] = u53(w2.z, w2.w) - 0.5; t[2] = 1.6 * u53(w3.x, w3.y) + 0.6; // synth.py:41
}

// one thread per 3D point (record r of problem b: points first, then the 2 n_l line end points)
__global__ void __launch_bounds__(256) synth_kernel(SynthArgs a)
{
    const int nrec = a.n_p + 2 * a.n_l;
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (gid >= a.batch * (nrec > 0 ? nrec : 1)) return;
    const int64_t b = gid / (nrec > 0 ? nrec : 1);
    const int r = (int)(gid - b * (nrec > 0 ? nrec : 1));
    double R[9], t[3];
    synth_pose(a, b, R, t);
    if (r == 0) {
        if (a.R_gt) for (int i = 0; i < 9; ++i) a.R_gt[b * 9 + i] = R[i];
        if (a.t_gt) for (int i = 0; i < 3; ++i) a.t_gt[b * 3 + i] = t[i];
    }
    if (nrec == 0) return;
    const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
    const U4 w0 = philox4x32(U4{(uint32_t)b, (uint32_t)((uint64_t)b >> 32), (uint32_t)r, 0u}, k0, k1);
    const U4 w1 = philox4x32(U4{(uint32_t)b, (uint32_t)((uint64_t)b >> 32), (uint32_t)r, 1u}, k0, k1);
    const U4 w2 = philox4x32(U4{(uint32_t)b, (uint32_t)((uint64_t)b >> 32), (uint32_t)r, 2u}, k0, k1);
    const double X = a.length * (u53(w0.x, w0.y) - 0.5), Y = a.length * (u53(w0.z, w0.w) - 0.5), Z = a.length * (u53(w1.x, w1.y) - 0.5);
    // pixels = K (R X + t), dehomogenised (suite.py:17-19)
    const double xc = R[0] * X + R[1] * Y + R[2] * Z + t[0], yc = R[3] * X + R[4] * Y + R[5] * Z + t[1], zc = R[6] * X + R[7] * Y + R[8] * Z + t[2];
    const double *K = a.K;
    const double uh = K[0] * xc + K[1] * yc + K[2] * zc, vh = K[3] * xc + K[4] * yc + K[5] * zc, wh = K[6] * xc + K[7] * yc + K[8] * zc;
    double u = uh / wh, v = vh / wh;
    if (a.sigma > 0.0) { // N(0, sigma^2) pixel noise (synth.py:283): Box-Muller on two uniforms
        const double rad = sqrt(-2.0 * log(1.0 - u53(w1.z, w1.w))), th = 2.0 * M_PI * u53(w2.x, w2.y);
        u += a.sigma * rad * cos(th);
        v += a.sigma * rad * sin(th);
    }
    if (r < a.n_p) {
        a.p3[(b * a.n_p + r) * 3] = X; a.p3[(b * a.n_p + r) * 3 + 1] = Y; a.p3[(b * a.n_p + r) * 3 + 2] = Z;
        a.p2[(b * a.n_p + r) * 2] = u; a.p2[(b * a.n_p + r) * 2 + 1] = v;
    } else { // line end point e = r - n_p: line e / 2, end e % 2 (synth.py:296-310)
        const int e = r - a.n_p;
        double *q3 = a.l3 + (b * a.n_l * 2 + e) * 3, *q2 = a.l2 + (b * a.n_l * 2 + e) * 2;
        q3[0] = X; q3[1] = Y; q3[2] = Z;
        q2[0] = u; q2[1] = v;
    }
}

// suite.py:8-14, :22-33.  ang [batch] in degrees, trans [batch]; NaN estimates give NaN.
__global__ void __launch_bounds__(256) pose_error_kernel(int64_t batch, const double *R_gt, const double *t_gt, const double *R, const double *t,
                                                         double *ang, double *trans)
{
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= batch) return;
    double G[9], Gi[9], det, E[9], Rb[9];
    bool fin = true;
    for (int i = 0; i < 9; ++i) { G[i] = R_gt[b * 9 + i]; Rb[i] = R[b * 9 + i]; fin = fin && (Rb[i] == Rb[i]); }
    cvx::inv3(G, Gi, det); // R_gt^-1 R (np.linalg.solve, not R_gt^T: R_gt is taken as given, suite.py:28)
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) E[i * 3 + j] = Gi[i * 3] * Rb[j] + Gi[i * 3 + 1] * Rb[3 + j] + Gi[i * 3 + 2] * Rb[6 + j];
    double Q[9];
    cvx::polar3(E, Q, 40); // U Vh of the SVD (suite.py:11-12)
    double c = 0.5 * (Q[0] + Q[4] + Q[8] - 1.0);
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    ang[b] = fin ? acos(c) * (180.0 / M_PI) : NAN;
    double d2 = 0, g2 = 0;
    for (int i = 0; i < 3; ++i) { const double d = t[b * 3 + i] - t_gt[b * 3 + i]; d2 += d * d; g2 += t_gt[b * 3 + i] * t_gt[b * 3 + i]; }
    trans[b] = sqrt(d2) / sqrt(g2);
}

// suite.py:96-108.  R_all [batch][4][9], t_all [batch][4][3], n_poses [batch]; support [n_support][3].  Outputs the chosen
// pose and its index (-1 and NaN when there is none).
__global__ void __launch_bounds__(256) disambiguate_kernel(int64_t batch, const double *R_all, const double *t_all, const int32_t *n_poses, const double *K,
                                                           const double *R_gt, const double *t_gt, const double *support, int n_support,
                                                           double *R_out, double *t_out, int32_t *idx_out)
{
    const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (b >= batch) return;
    const int np = n_poses[b] < 0 ? 0 : (n_poses[b] > 4 ? 4 : n_poses[b]);
    auto project = [&](const double *Rm, const double *tm, const double *P, double &u, double &v) {
        const double xc = Rm[0] * P[0] + Rm[1] * P[1] + Rm[2] * P[2] + tm[0], yc = Rm[3] * P[0] + Rm[4] * P[1] + Rm[5] * P[2] + tm[1],
                     zc = Rm[6] * P[0] + Rm[7] * P[1] + Rm[8] * P[2] + tm[2];
        const double uh = K[0] * xc + K[1] * yc + K[2] * zc, vh = K[3] * xc + K[4] * yc + K[5] * zc, wh = K[6] * xc + K[7] * yc + K[8] * zc;
        u = uh / wh; v = vh / wh;
    };
    int best = -1;
    double best_err = INFINITY;
    for (int i = 0; i < np; ++i) {
        const double *Rm = R_all + (b * 4 + i) * 9, *tm = t_all + (b * 4 + i) * 3;
        double err = 0;
        for (int s = 0; s < n_support; ++s) {
            double ug, vg, ue, ve;
            project(R_gt + b * 9, t_gt + b * 3, support + 3 * s, ug, vg);
            project(Rm, tm, support + 3 * s, ue, ve);
            err += sqrt((ug - ue) * (ug - ue) + (vg - ve) * (vg - ve));
        }
        if (err < best_err) { best_err = err; best = i; } // a NaN error never wins (suite.py:104 compares with <)
    }
    for (int i = 0; i < 9; ++i) R_out[b * 9 + i] = best >= 0 ? R_all[(b * 4 + best) * 9 + i] : NAN;
    for (int i = 0; i < 3; ++i) t_out[b * 3 + i] = best >= 0 ? t_all[(b * 4 + best) * 3 + i] : NAN;
    idx_out[b] = best;
}

} // namespace cvxg
