// Microbenchmark: one-sided Jacobi eigen-solve of 10x10 symmetric matrices, 16 lanes per matrix
// (4 matrices per wave64), one column per lane, partner columns fetched with ds_bpermute.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o eig16 eig16.hip ; run: ./eig16 [batch] [sweeps]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__host__ __device__ constexpr int rr_col(int step, int pos)
{
    int a[5] = {0, 2, 4, 6, 8}, b[5] = {1, 3, 5, 7, 9};
    for (int s = 0; s < step; ++s) {
        const int na1 = b[0], nb4 = a[4];
        a[4] = a[3]; a[3] = a[2]; a[2] = a[1]; a[1] = na1;
        b[0] = b[1]; b[1] = b[2]; b[2] = b[3]; b[3] = b[4]; b[4] = nb4;
    }
    return pos < 5 ? a[pos] : b[pos - 5];
}
struct PTab { unsigned long long packed[16]; };
constexpr PTab make_ptab()
{
    PTab t{};
    for (int l = 0; l < 16; ++l) {
        unsigned long long w = 0;
        for (int st = 0; st < 9; ++st) {
            int partner = l;
            for (int k = 0; k < 5; ++k) {
                const int p = rr_col(st, k), q = rr_col(st, k + 5);
                if (p == l) partner = q;
                if (q == l) partner = p;
            }
            w |= (unsigned long long)partner << (4 * st);
        }
        t.packed[l] = w;
    }
    return t;
}
__device__ const PTab kPTab = make_ptab();

__device__ __forceinline__ double bperm(int addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ void jacobi_cs(double d, double gam, bool rot, bool tie_neg, double &c, double &s, double &t)
{
    const double g2 = 2.0 * gam;
    const float df = (float)d, gf = (float)g2;
    const float h2 = df * df + gf * gf + 1e-37f;
    const float hf = h2 * __builtin_amdgcn_rsqf(h2);
    float tf = gf * __builtin_amdgcn_rcpf(fabsf(df) + hf);
    const bool neg = d < 0.0 || (d == 0.0 && tie_neg);
    tf = neg ? -tf : tf;
    t = rot ? (double)tf : 0.0;
    const double x = 1.0 + t * t;
    double z = (double)__builtin_amdgcn_rsqf((float)x);
    { double hh = 0.5 * x * z; double e = fma(-hh, z, 0.5); z = fma(z, e, z); }
    c = z;
    s = t * c;
}

// W: [batch][55] packed upper triangle (row-major); out: [batch][10] eigenvalues; sw: sweeps used
template <int MODE>
__global__ __launch_bounds__(64) void eig16_kernel(const double *W, double *lam_out, int *sw_out, double *orth_out, int batch, int max_sweeps, double tol2)
{
    __shared__ double lds[64 * 12];
    const int lane = threadIdx.x & 63;
    const int gl = lane & 15, grp = lane >> 4;
    const long prob = (long)blockIdx.x * 4 + grp;
    const bool valid = prob < batch && gl < 10;
    const long pc = prob < batch ? prob : batch - 1;
    const int col = gl < 10 ? gl : 0;
    // column `col` of W + sigma I
    double g[10];
    double fro = 0;
    {
        const double *w = W + pc * 55;
        int k = 0;
#pragma unroll
        for (int i = 0; i < 10; ++i)
#pragma unroll
            for (int j = i; j < 10; ++j) {
                const double v = w[k++];
                fro += (i == j ? 1.0 : 2.0) * v * v;
#pragma unroll
                for (int r = 0; r < 10; ++r) {
                    if (r == i) g[r] = (col == j) ? v : g[r];
                    if (r == j) g[r] = (col == i) ? v : g[r];
                }
            }
    }
    const double sigma = 1.5 * sqrt(fro) + 1e-300;
#pragma unroll
    for (int r = 0; r < 10; ++r) g[r] += (r == col) ? sigma : 0.0;
    if (gl >= 10) {
#pragma unroll
        for (int r = 0; r < 10; ++r) g[r] = 0.0;
    }
    double al = 0;
#pragma unroll
    for (int r = 0; r < 10; ++r) al += g[r] * g[r];
    const unsigned long long ptab = kPTab.packed[gl];
    const int base = (lane & 48);
    int sweeps = 0;
    bool active = true; // group-uniform
    while (true) {
        bool coarse = false;
#pragma unroll
        for (int st = 0; st < 9; ++st) {
            const int partner = (int)((ptab >> (4 * st)) & 15);
            const int addr = (base + partner) << 2;
            double o[10];
            double be;
            if (MODE == 0) {
#pragma unroll
                for (int r = 0; r < 10; ++r) o[r] = bperm(addr, g[r]);
                be = bperm(addr, al);
            } else if (MODE == 1) { // fake: partner = lane ^ 1 through DPP quad_perm [1,0,3,2]
#pragma unroll
                for (int r = 0; r < 10; ++r) {
                    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(g[r]), 0xB1, 0xF, 0xF, false);
                    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(g[r]), 0xB1, 0xF, 0xF, false);
                    o[r] = __hiloint2double(hi, lo);
                }
                {
                    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(al), 0xB1, 0xF, 0xF, false);
                    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(al), 0xB1, 0xF, 0xF, false);
                    be = __hiloint2double(hi, lo);
                }
            } else { // through LDS memory: write own column (+ norm), read the partner's
                double2 *mine = reinterpret_cast<double2 *>(lds + lane * 12);
                mine[0] = make_double2(g[0], g[1]); mine[1] = make_double2(g[2], g[3]); mine[2] = make_double2(g[4], g[5]);
                mine[3] = make_double2(g[6], g[7]); mine[4] = make_double2(g[8], g[9]); mine[5] = make_double2(al, 0.0);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                const double2 *his = reinterpret_cast<const double2 *>(lds + (base + partner) * 12);
                const double2 h0 = his[0], h1 = his[1], h2 = his[2], h3 = his[3], h4 = his[4], h5 = his[5];
                o[0] = h0.x; o[1] = h0.y; o[2] = h1.x; o[3] = h1.y; o[4] = h2.x; o[5] = h2.y; o[6] = h3.x; o[7] = h3.y; o[8] = h4.x; o[9] = h4.y;
                be = h5.x;
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            const double gam = ((g[0] * o[0] + g[1] * o[1]) + (g[2] * o[2] + g[3] * o[3])) + ((g[4] * o[4] + g[5] * o[5]) + (g[6] * o[6] + g[7] * o[7])) + (g[8] * o[8] + g[9] * o[9]);
            const double g2 = gam * gam, ab = al * be;
            coarse |= g2 > tol2 * ab;
            double c, s, t;
            jacobi_cs(be - al, gam, active && g2 > 1e-30 * ab, gl > partner, c, s, t);
#pragma unroll
            for (int r = 0; r < 10; ++r) g[r] = c * g[r] - s * o[r];
            al = al - t * gam;
        }
        // exact norm once per sweep
        al = 0;
#pragma unroll
        for (int r = 0; r < 10; ++r) al += g[r] * g[r];
        const unsigned long long m = __ballot(coarse && gl < 10 && active);
        const bool grp_more = ((m >> (16 * grp)) & 0xFFFFull) != 0;
        if (active) ++sweeps;
        active = active && grp_more && sweeps < max_sweeps;
        if (!__any(active)) break;
    }
    if (valid) {
        lam_out[prob * 10 + gl] = sqrt(al) - sigma;
        if (gl == 0) sw_out[prob] = sweeps;
    }
    // orthogonality check: max |g_j . g_k| / (|g_j||g_k|) against partner of step 0
    {
        const int partner = (int)(ptab & 15);
        const int addr = (base + partner) << 2;
        double gam = 0;
#pragma unroll
        for (int r = 0; r < 10; ++r) gam += g[r] * bperm(addr, g[r]);
        const double be = bperm(addr, al);
        if (valid) orth_out[prob * 10 + gl] = fabs(gam) / sqrt(al * be);
    }
}

static void launch(int mode, int blocks, const double *dW, double *dl, int *dsw, double *dorth, int batch, int max_sweeps, double tol2)
{
    if (mode == 0) eig16_kernel<0><<<blocks, 64>>>(dW, dl, dsw, dorth, batch, max_sweeps, tol2);
    else if (mode == 1) eig16_kernel<1><<<blocks, 64>>>(dW, dl, dsw, dorth, batch, max_sweeps, tol2);
    else eig16_kernel<2><<<blocks, 64>>>(dW, dl, dsw, dorth, batch, max_sweeps, tol2);
}

int main(int argc, char **argv)
{
    const int mode = argc > 4 ? atoi(argv[4]) : 0;
    const int batch = argc > 1 ? atoi(argv[1]) : 10000;
    const int max_sweeps = argc > 2 ? atoi(argv[2]) : 12;
    const double tol = argc > 3 ? atof(argv[3]) : 6e-2;
    std::vector<double> W((size_t)batch * 55);
    srand(1);
    for (auto &v : W) v = (double)rand() / RAND_MAX - 0.5;
    double *dW, *dl, *dorth;
    int *dsw;
    CHECK(hipMalloc(&dW, W.size() * 8));
    CHECK(hipMalloc(&dl, (size_t)batch * 80));
    CHECK(hipMalloc(&dorth, (size_t)batch * 80));
    CHECK(hipMalloc(&dsw, (size_t)batch * 4));
    CHECK(hipMemcpy(dW, W.data(), W.size() * 8, hipMemcpyHostToDevice));
    const int blocks = (batch + 3) / 4;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) launch(mode, blocks, dW, dl, dsw, dorth, batch, max_sweeps, tol * tol);
    CHECK(hipDeviceSynchronize());
    const int reps = 20;
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch(mode, blocks, dW, dl, dsw, dorth, batch, max_sweeps, tol * tol);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<double> lam((size_t)batch * 10), orth((size_t)batch * 10);
    std::vector<int> sw(batch);
    CHECK(hipMemcpy(lam.data(), dl, lam.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(orth.data(), dorth, orth.size() * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(sw.data(), dsw, sw.size() * 4, hipMemcpyDeviceToHost));
    double tr_err = 0, mo = 0, msw = 0;
    for (int b = 0; b < batch; ++b) {
        double tr = 0, sl = 0;
        int k = 0;
        for (int i = 0; i < 10; ++i) for (int j = i; j < 10; ++j) { if (i == j) tr += W[(size_t)b * 55 + k]; ++k; }
        for (int j = 0; j < 10; ++j) { sl += lam[(size_t)b * 10 + j]; mo = fmax(mo, orth[(size_t)b * 10 + j]); }
        tr_err = fmax(tr_err, fabs(tr - sl));
        msw += sw[b];
    }
    printf("{\"mode\": %d, \"batch\": %d, \"max_sweeps\": %d, \"tol\": %g, \"ms_per_launch\": %.4f, \"eigs_per_s\": %.3e, \"mean_sweeps\": %.2f, \"us_per_sweep_all\": %.3f, \"trace_err\": %.2e, \"max_cos_step0_pairs\": %.2e}\n",
           mode, batch, max_sweeps, tol, ms / reps, batch / (ms / reps * 1e-3), msw / batch, (ms / reps * 1e3) / (msw / batch), tr_err, mo);
    return 0;
}
