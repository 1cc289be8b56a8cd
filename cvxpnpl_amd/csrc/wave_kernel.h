// wave_kernel.h -- wave-per-problem layout: one gfx950 wavefront owns one SDP.
//
// The 64 lanes of a wave cooperate on ONE problem; all state lives in registers and in
// a private 6 KB LDS slice (no inter-wave communication, no __syncthreads):
//   * "entry lanes"  e = 0..54   own entry (i <= j) of the symmetric 10x10 iterate W
//                                (vech order of cvxpnpl.py:346-370) -- affine projection,
//                                ADMM update, PSD reconstruction are one entry per lane;
//   * "jacobi lanes" 10 k + i    own row i of the two columns paired at position k of a
//                                round-robin (circle) ordering -- the one-sided Jacobi
//                                rotation of pair k is local to its 10 lanes, the dot
//                                product g_p . g_q is a 10-lane LDS reduction, and the
//                                pairing advances by shifting one value per lane to the
//                                neighbouring group (lane +-10) through LDS;
//   * "accumulator lanes" 0..59  own one of the 60 Gram accumulators of the assembly.
// The certificate (SO(3) Newton polish, dual recovery, LDL^T) is serial 3x3 / 10x10
// algebra: lane 0 runs the scalar routine of solver_core.h on data gathered through LDS.
// Control flow (sweeps, iterations, exit) is wave-uniform: no divergence, per-problem
// early exit frees the SIMD slot for the next wave.
#pragma once
#include <hip/hip_runtime.h>

#include "problem_io.h"
#include "solver_core.h"

namespace cvxw {

constexpr int WPB = 1; // waves (= problems) per block: 1, so a finished problem frees its SIMD slot at once

// LDS slice per wave, in doubles (all offsets even => 16-byte aligned)
constexpr int L_P = 0;      // 64   per-lane products for 10-lane reductions
constexpr int L_EX = 64;    // 256  exchange buffer: lane -> (a, alpha), (b, beta)
constexpr int L_Y = 320;    // 200  (g, w g) per [slot][row]
constexpr int L_X = 520;    // 64   entry scratch (affine projection gathers)
constexpr int L_G = 584;    // 100  full 10x10
constexpr int L_B = 684;    // 28   translation map B (27)
constexpr int L_M = 712;    // 40   misc scalars / results of lane 0
constexpr int LDSW = 752;

struct LaneTab {
    signed char ei[64], ej[64], p1[64], p2[64], s0[64], s1[64], s2[64], diag[64];
};

constexpr LaneTab make_lane_tab()
{
    LaneTab t{};
    int e = 0;
    for (int i = 0; i < 10; ++i)
        for (int j = i; j < 10; ++j) { t.ei[e] = (signed char)i; t.ej[e] = (signed char)j; t.diag[e] = (i == j); ++e; }
    for (int l = 55; l < 64; ++l) { t.ei[l] = t.ei[l - 55]; t.ej[l] = t.ej[l - 55]; t.diag[l] = t.diag[l - 55]; }
    for (int tr = 0; tr < 15; ++tr)
        for (int k = 0; k < 3; ++k) {
            int k1 = (k + 1) % 3, k2 = (k + 2) % 3;
            int ee = cvx::sidx(cvx::tri_i(tr, k), cvx::tri_j(tr, k));
            t.p1[ee] = (signed char)cvx::sidx(cvx::tri_i(tr, k1), cvx::tri_j(tr, k1));
            t.p2[ee] = (signed char)cvx::sidx(cvx::tri_i(tr, k2), cvx::tri_j(tr, k2));
            t.s0[ee] = (signed char)cvx::tri_s(tr, k);
            t.s1[ee] = (signed char)cvx::tri_s(tr, k1);
            t.s2[ee] = (signed char)cvx::tri_s(tr, k2);
        }
    for (int l = 55; l < 64; ++l) { t.p1[l] = t.p1[l - 55]; t.p2[l] = t.p2[l - 55]; t.s0[l] = t.s0[l - 55]; t.s1[l] = t.s1[l - 55]; t.s2[l] = t.s2[l - 55]; }
    return t;
}

__device__ const LaneTab kLaneTab = make_lane_tab();

#define CVXW_SYNC()                                              \
    do {                                                         \
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
        __builtin_amdgcn_wave_barrier();                         \
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
    } while (0)

__device__ __forceinline__ double wave_sum(double x)
{
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64);
    return x;
}

struct WaveArgs {
    int64_t batch;
    int n_p, n_l, K_per_problem;
    const double *p2, *p3, *l2, *l3, *K;
    double *R, *t, *cost, *Z;
    int32_t *status, *iters, *work;
};

__global__ void __launch_bounds__(64 * WPB) solve_wave_kernel(WaveArgs a, cvx::Opts o)
{
    __shared__ __attribute__((aligned(16))) double lds_all[WPB][LDSW];
    const int lane = threadIdx.x & 63, wib = threadIdx.x >> 6;
    const int64_t b = (int64_t)blockIdx.x * WPB + wib;
    if (b >= a.batch) return; // wave-uniform
    double *L = lds_all[wib];
    double2 *L2 = reinterpret_cast<double2 *>(L);

    // ---------------------------------------------------------------- lane roles
    const int ei = kLaneTab.ei[lane], ej = kLaneTab.ej[lane];
    const int el = lane < 55 ? lane : lane - 55;      // entry index (lanes 55..63 alias 0..8, weight 0)
    const double wgt = lane < 55 ? (ei == ej ? 1.0 : 2.0) : 0.0;
    const bool is_diag = kLaneTab.diag[lane] != 0;
    const int p1 = kLaneTab.p1[lane], p2 = kLaneTab.p2[lane];
    const double s0 = kLaneTab.s0[lane], s1 = kLaneTab.s1[lane], s2 = kLaneTab.s2[lane];
    const int jl = lane < 50 ? lane : lane - 50;      // jacobi lane (50..63 alias 0..13)
    const int ji = jl % 10, jk = jl / 10;

    // ---------------------------------------------------------------- assembly
    cvx::ProblemView pv = cvx::make_view(b, a.n_p, a.p2, a.p3, a.n_l, a.l2, a.l3, a.K, a.K_per_problem);
    double Ki[9];
    bool okK;
    {
        double Kc[9], det;
#pragma unroll
        for (int i = 0; i < 9; ++i) Kc[i] = pv.K[i];
        cvx::inv3(Kc, Ki, det);
        okK = (det == det) && det != 0.0;
    }
    // accumulator role of this lane: M0 (6) | M1 (3 x 6) | M2 (6 x 6)
    int acc_type = 0, acc_pa = 0, acc_pb = 0, acc_te = 0;
    {
        const int al = lane < 60 ? lane : 0;
        if (al < 6) { acc_type = 0; acc_te = al; }
        else if (al < 24) { acc_type = 1; acc_pa = (al - 6) / 6; acc_te = (al - 6) % 6; }
        else {
            acc_type = 2;
            int ab = (al - 24) / 6;
            acc_te = (al - 24) % 6;
            acc_pa = ab < 3 ? 0 : (ab < 5 ? 1 : 2);
            acc_pb = ab < 3 ? ab : (ab < 5 ? ab - 2 : 2);
        }
    }
    double acc = 0.0;
    const int nrec = pv.n_p + 2 * pv.n_l;
    constexpr int CHUNK = 32; // records (T[6], P[3]) staged per pass in L_EX.. (32 * 10 doubles)
    for (int base = 0; base < nrec; base += CHUNK) {
        const int cnt = nrec - base < CHUNK ? nrec - base : CHUNK;
        if (lane < cnt) {
            const int r = base + lane;
            double T[6], P[3];
            if (r < pv.n_p) {
                double p[3];
                cvx::bearing(Ki, pv.p2[2 * r], pv.p2[2 * r + 1], p);
                double n2 = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];
                T[0] = n2 - p[0] * p[0]; T[1] = -p[0] * p[1]; T[2] = -p[0] * p[2];
                T[3] = n2 - p[1] * p[1]; T[4] = -p[1] * p[2]; T[5] = n2 - p[2] * p[2];
                P[0] = pv.p3[3 * r]; P[1] = pv.p3[3 * r + 1]; P[2] = pv.p3[3 * r + 2];
            } else {
                const int li = (r - pv.n_p) >> 1, en = (r - pv.n_p) & 1;
                const double *l2 = pv.l2 + 4 * li, *l3 = pv.l3 + 6 * li + 3 * en;
                double u[3], v[3];
                cvx::bearing(Ki, l2[0], l2[1], u);
                cvx::bearing(Ki, l2[2], l2[3], v);
                double n[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
                double inv = 1.0 / sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                n[0] *= inv; n[1] *= inv; n[2] *= inv;
                T[0] = n[0] * n[0]; T[1] = n[0] * n[1]; T[2] = n[0] * n[2]; T[3] = n[1] * n[1]; T[4] = n[1] * n[2]; T[5] = n[2] * n[2];
                P[0] = l3[0]; P[1] = l3[1]; P[2] = l3[2];
            }
            double *rec = L + L_EX + lane * 10;
#pragma unroll
            for (int i = 0; i < 6; ++i) rec[i] = T[i];
            rec[6] = 1.0; rec[7] = P[0]; rec[8] = P[1]; rec[9] = P[2];
        }
        CVXW_SYNC();
        for (int c = 0; c < cnt; ++c) {
            const double *rec = L + L_EX + c * 10;
            double coef = (acc_type == 0 ? 1.0 : rec[7 + acc_pa]) * (acc_type == 2 ? rec[7 + acc_pb] : 1.0);
            acc += coef * rec[acc_te];
        }
        CVXW_SYNC();
    }
    L[L_P + lane] = acc; // ACC[0..59]
    CVXW_SYNC();
    // B = M0^-1 [M1_0 M1_1 M1_2], Q = M2 - M1^T B; every lane inverts M0 redundantly
    double Mi[9];
    bool okG;
    {
        const double *m = L + L_P;
        double M0[9] = {m[0], m[1], m[2], m[1], m[3], m[4], m[2], m[4], m[5]}, det;
        cvx::inv3(M0, Mi, det);
        double sc = m[0] + m[3] + m[5];
        okG = det > 1e-12 * (sc * sc * sc) * (1.0 / 27.0);
    }
    constexpr int psym[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
    if (lane < 27) {
        const int bb = lane / 9, i = (lane % 9) / 3, j = lane % 3;
        const double *m1 = L + L_P + 6 + 6 * bb;
        double v = 0;
#pragma unroll
        for (int k = 0; k < 3; ++k) v += Mi[i * 3 + k] * m1[psym[3 * k + j]];
        L[L_B + i * 9 + 3 * bb + j] = v; // B[i][3 bb + j]
    }
    CVXW_SYNC();
    double Qe = 0.0; // Q9 entry of this entry-lane (0 outside the 9x9 block)
    if (ej < 9) {
        const int qa = ei / 3, qi = ei % 3, qb = ej / 3, qj = ej % 3;
        constexpr int abidx[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
        const double *m1 = L + L_P + 6 + 6 * qa, *m2 = L + L_P + 24 + 6 * abidx[3 * qa + qb];
        double v = m2[psym[3 * qi + qj]];
#pragma unroll
        for (int k = 0; k < 3; ++k) v -= m1[psym[3 * qi + k]] * L[L_B + k * 9 + 3 * qb + qj];
        Qe = v;
    }
    L[L_X + el] = Qe;
    CVXW_SYNC();
    double tr = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) tr += L[L_X + cvx::sidx(k, k)];
    bool finite = okK && okG && (tr == tr) && tr > 0 && tr < 1e300;
    finite = !__any(!(finite && (Qe == Qe)));
    const double itr = finite ? 1.0 / tr : 0.0;
    const double Qs = Qe * itr;

    // ---------------------------------------------------------------- ADMM
    double delta = o.eps / (8.0 * tr);
    delta = delta < 1e-13 ? 1e-13 : delta;
    const double irho = 1.0 / o.rho;
    double W = (lane == 54) ? 1.0 : 0.0, Wp = 0.0;
    int it = 0, total_sweeps = 0, next_check = o.first_check;
    double fp_res = 1e300;
    int status = cvx::ST_NONFINITE, rank_out = 0;
    bool done = !finite;
    bool certified = false;

    while (!done) {
        // ---- eigendecomposition of W: one-sided Jacobi on G = W + sigma I
        const double fro2 = wave_sum(wgt * W * W);
        const double sigma = 1.5 * sqrt(fro2) + 1e-300;
        {
            const double g = W + (is_diag ? sigma : 0.0);
            L[L_G + ei * 10 + ej] = g;
            L[L_G + ej * 10 + ei] = g;
        }
        CVXW_SYNC();
        // columns 2k, 2k+1 of the symmetric G are its rows: contiguous in L_G
        double ca, cb, al, be, gam;
        {
            const double2 *ra = L2 + (L_G + (2 * jk) * 10) / 2, *rb = L2 + (L_G + (2 * jk + 1) * 10) / 2;
            const double2 a0 = ra[0], a1 = ra[1], a2 = ra[2], a3 = ra[3], a4 = ra[4];
            const double2 b0 = rb[0], b1 = rb[1], b2 = rb[2], b3 = rb[3], b4 = rb[4];
            al = ((a0.x * a0.x + a0.y * a0.y) + (a1.x * a1.x + a1.y * a1.y)) + ((a2.x * a2.x + a2.y * a2.y) + (a3.x * a3.x + a3.y * a3.y)) + (a4.x * a4.x + a4.y * a4.y);
            be = ((b0.x * b0.x + b0.y * b0.y) + (b1.x * b1.x + b1.y * b1.y)) + ((b2.x * b2.x + b2.y * b2.y) + (b3.x * b3.x + b3.y * b3.y)) + (b4.x * b4.x + b4.y * b4.y);
            gam = ((a0.x * b0.x + a0.y * b0.y) + (a1.x * b1.x + a1.y * b1.y)) + ((a2.x * b2.x + a2.y * b2.y) + (a3.x * b3.x + a3.y * b3.y)) + (a4.x * b4.x + a4.y * b4.y);
            ca = L[L_G + (2 * jk) * 10 + ji];
            cb = L[L_G + (2 * jk + 1) * 10 + ji];
        }
        // exchange buffers: first / second column of every pair, and their squared norms
        constexpr int CA = L_EX, CB = L_EX + 64, NA = L_EX + 128, NB = L_EX + 136;
        // after the rotation position 0 keeps its first column and every other column moves one
        // place along the ring a1 > a2 > a3 > a4 > b4 > b3 > b2 > b1 > b0 > a1 (circle method)
        const int src_a = jk == 0 ? CA : (jk == 1 ? CB : CA + (jk - 1) * 10);
        const int src_an = jk == 0 ? NA : (jk == 1 ? NB : NA + jk - 1);
        const int src_b = jk == 4 ? CA + 40 : CB + (jk + 1) * 10;
        const int src_bn = jk == 4 ? NA + 4 : NB + jk + 1;
        const double tol2 = o.jacobi_tol * o.jacobi_tol;
        int sweeps = 0;
        bool more;
        do {
            bool coarse = false;
            for (int step = 0; step < 9; ++step) {
                const double g2 = gam * gam, ab = al * be;
                coarse |= g2 > tol2 * ab;
                double c, s, t;
                cvx::jacobi_cs(al, be, gam, g2 > 1e-30 * ab, c, s, t);
                L[CA + jl] = c * ca - s * cb;
                L[CB + jl] = s * ca + c * cb;
                if (ji == 0) { L[NA + jk] = al - t * gam; L[NB + jk] = be + t * gam; }
                CVXW_SYNC();
                const double2 *ra = L2 + src_a / 2, *rb = L2 + src_b / 2;
                const double2 a0 = ra[0], a1 = ra[1], a2 = ra[2], a3 = ra[3], a4 = ra[4];
                const double2 b0 = rb[0], b1 = rb[1], b2 = rb[2], b3 = rb[3], b4 = rb[4];
                ca = L[src_a + ji];
                cb = L[src_b + ji];
                gam = ((a0.x * b0.x + a0.y * b0.y) + (a1.x * b1.x + a1.y * b1.y)) + ((a2.x * b2.x + a2.y * b2.y) + (a3.x * b3.x + a3.y * b3.y)) + (a4.x * b4.x + a4.y * b4.y);
                if (step == 8) { // exact norms once per sweep (the incremental update drifts)
                    al = ((a0.x * a0.x + a0.y * a0.y) + (a1.x * a1.x + a1.y * a1.y)) + ((a2.x * a2.x + a2.y * a2.y) + (a3.x * a3.x + a3.y * a3.y)) + (a4.x * a4.x + a4.y * a4.y);
                    be = ((b0.x * b0.x + b0.y * b0.y) + (b1.x * b1.x + b1.y * b1.y)) + ((b2.x * b2.x + b2.y * b2.y) + (b3.x * b3.x + b3.y * b3.y)) + (b4.x * b4.x + b4.y * b4.y);
                } else {
                    al = L[src_an];
                    be = L[src_bn];
                }
                CVXW_SYNC();
            }
            ++sweeps;
            more = __any(coarse) && sweeps < o.jacobi_sweeps;
        } while (more);
        total_sweeps += sweeps;
        // ---- Wp = sum_{lam > 0} lam v v^T, from (g, w g) with w = lam / lam'^2
        const double lpa = sqrt(al), lpb = sqrt(be);
        const double lama = lpa - sigma, lamb = lpb - sigma;
        const double wa = lama > 0 ? lama / al : 0.0, wb = lamb > 0 ? lamb / be : 0.0;
        L2[(L_Y + 2 * ((2 * jk) * 10 + ji)) / 2] = make_double2(ca, wa * ca);
        L2[(L_Y + 2 * ((2 * jk + 1) * 10 + ji)) / 2] = make_double2(cb, wb * cb);
        if (ji == 0) { // slot eigenvalue data for the top-eigenvector / rank decisions
            L[L_M + 2 * jk] = al;
            L[L_M + 2 * jk + 1] = be;
        }
        const unsigned long long ma = __ballot(wa > 0), mb = __ballot(wb > 0);
        CVXW_SYNC();
        Wp = 0.0;
#pragma unroll
        for (int s = 0; s < 10; ++s) {
            const bool pos = ((s & 1) ? (mb >> (10 * (s >> 1))) : (ma >> (10 * (s >> 1)))) & 1ull;
            if (pos) { // wave-uniform
                const double2 yi = L2[(L_Y + 2 * (s * 10 + ei)) / 2], yj = L2[(L_Y + 2 * (s * 10 + ej)) / 2];
                Wp += yi.y * yj.x;
            }
        }
        ++it;
        const bool check = it >= next_check;
        const bool last = (it >= o.max_iters) || (fp_res < o.res_tol);
        if (check || last) {
            // top eigenvector slot (wave-uniform) and its unit vector
            int smax = 0;
            double best = -1.0;
#pragma unroll
            for (int s = 0; s < 10; ++s) { const double n2 = L[L_M + s]; const bool bt = n2 > best; best = bt ? n2 : best; smax = bt ? s : smax; }
            int rank = 0;
#pragma unroll
            for (int s = 0; s < 10; ++s) rank += (sqrt(L[L_M + s]) - sigma) > 1e-3;
            // gather Qs, W, Wp for lane 0
            L[L_EX + el] = W;
            L[L_EX + 56 + el] = Wp;
            L[L_EX + 112 + el] = Qs;
            CVXW_SYNC();
            if (lane == 0) {
                double q45[45], w55[55], wp55[55], v[10];
                const double il = 1.0 / sqrt(best);
#pragma unroll
                for (int i = 0; i < 55; ++i) { w55[i] = L[L_EX + i]; wp55[i] = L[L_EX + 56 + i]; }
#pragma unroll
                for (int i = 0; i < 9; ++i)
#pragma unroll
                    for (int j = i; j < 9; ++j) q45[cvx::qidx(i, j)] = L[L_EX + 112 + cvx::sidx(i, j)];
#pragma unroll
                for (int i = 0; i < 10; ++i) v[i] = L[L_Y + 2 * (smax * 10 + i)] * il;
                cvx::Cert c;
                cvx::certify(q45, w55, wp55, v, o.rho, delta, c);
                const bool gap_ok = c.ok && (tr * (fabs(c.zSz) + 4.0 * delta) <= (o.eps > 8e-13 * tr ? o.eps : 8e-13 * tr));
                cvx::Solution sol;
                if (gap_ok) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) sol.R[i] = c.R[i];
                    sol.cost = tr * c.pobj;
                    sol.dobj = tr * (c.pobj - c.zSz - 4.0 * delta);
                    sol.status = cvx::ST_CERTIFIED;
                    sol.rank = 1;
                } else if (last) {
                    cvx::fallback_pose(q45, tr, v, rank, sol);
                }
                if (gap_ok || last) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) L[L_M + 16 + i] = sol.R[i];
                    L[L_M + 25] = sol.cost; L[L_M + 26] = sol.dobj;
                    L[L_M + 27] = (double)sol.status; L[L_M + 28] = (double)sol.rank;
                }
                L[L_M + 15] = gap_ok ? 1.0 : 0.0;
            }
            CVXW_SYNC();
            certified = L[L_M + 15] != 0.0;
            next_check = cvx::next_check_after(it, o);
            if (certified || last) {
                status = (int)L[L_M + 27];
                rank_out = (int)L[L_M + 28];
                done = true;
            }
        }
        if (!done) {
            // X = Pi_aff(2 Wp - W - Qs / rho);  W <- W + alpha (X - Wp)
            double X = 2.0 * Wp - W - irho * Qs;
            L[L_X + el] = X;
            CVXW_SYNC();
            double Xn;
            {
                double d[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) d[k] = L[L_X + cvx::sidx(k, k)];
                const double r0 = d[0] + d[3] + d[6] - 1.0, r1 = d[1] + d[4] + d[7] - 1.0, r2 = d[2] + d[5] + d[8] - 1.0;
                const double c0 = d[0] + d[1] + d[2] - 1.0, c1 = d[3] + d[4] + d[5] - 1.0, c2 = d[6] + d[7] + d[8] - 1.0;
                const double tot = r0 + r1 + r2;
                const int ri = ei % 3, ci = ei / 3; // diagonal entry (ei, ei), ei < 9: D[ri][ci]
                const double rr = ri == 0 ? r0 : (ri == 1 ? r1 : r2), cc = ci == 0 ? c0 : (ci == 1 ? c1 : c2);
                const double xdiag = (ei == 9) ? 1.0 : X - (rr + cc) * (1.0 / 3.0) + tot * (1.0 / 9.0);
                const double m = (s0 * X + s1 * L[L_X + p1] + s2 * L[L_X + p2]) * (1.0 / 3.0);
                Xn = is_diag ? xdiag : X - s0 * m;
            }
            const double dd = Xn - Wp;
            W += o.alpha * dd;
            fp_res = sqrt(wave_sum(wgt * dd * dd));
            if (!(fp_res == fp_res)) { status = cvx::ST_NONFINITE; done = true; }
            CVXW_SYNC();
        }
    }

    // ---------------------------------------------------------------- outputs
    const bool have_pose = finite && status != cvx::ST_NONFINITE;
    if (lane < 9) a.R[b * 9 + lane] = have_pose ? L[L_M + 16 + lane] : NAN;
    if (lane < 3) { // t = -B r (cvxpnpl.py:513), r = vec_colmajor(R)
        double tv = 0;
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3)
#pragma unroll
            for (int r3 = 0; r3 < 3; ++r3) tv += L[L_B + lane * 9 + 3 * c3 + r3] * L[L_M + 16 + r3 * 3 + c3];
        a.t[b * 3 + lane] = have_pose ? -tv : NAN;
    }
    if (lane == 0) {
        a.status[b] = status;
        if (a.iters) a.iters[b] = it;
        if (a.cost) { a.cost[2 * b] = have_pose ? L[L_M + 25] : NAN; a.cost[2 * b + 1] = have_pose ? L[L_M + 26] : NAN; }
        if (a.work) { a.work[2 * b] = rank_out; a.work[2 * b + 1] = total_sweeps; }
    }
    if (a.Z && lane < 55) {
        double zv;
        if (!have_pose) zv = NAN;
        else if (certified) { // Z = z z^T with z = [vec_colmajor(R); 1]
            const double zi = ei == 9 ? 1.0 : L[L_M + 16 + (ei % 3) * 3 + ei / 3];
            const double zj = ej == 9 ? 1.0 : L[L_M + 16 + (ej % 3) * 3 + ej / 3];
            zv = zi * zj;
        } else zv = Wp;
        a.Z[b * 55 + lane] = zv;
    }
}

} // namespace cvxw
