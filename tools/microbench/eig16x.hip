// Microbenchmark (round 5): orderings of the column-per-lane one-sided Jacobi eigen-solve of the quad layout (16 lanes per 10x10
// matrix, four matrices per wave64, quad_kernel.h) -- what the column exchange costs and what replaces it.
//   SCHED 0  round-robin (circle method), partner column through ds_bpermute: 11 (float) / 22 (double) per step, 9 steps per sweep
//            -- the shipped ordering
//   SCHED 1  alternating ordering: steps 0, 2, 4, 6, 8 pair lanes (2k, 2k+1) -- the partner's column is a DPP quad_perm [1,0,3,2]
//            operand of the arithmetic itself (float: v_fmac_f32_dpp, no exchange instruction at all; double: 22 v_mov_b32_dpp) --
//            steps 1, 3, 5, 7 are ds_bpermute steps in which a lane may take over the PARTNER's rotated column (a swap costs nothing:
//            both lanes hold both columns).  Still 9 steps and every pair exactly once (the tables below come from a search,
//            tools/microbench/jacobi_orderings.py), LDS round trips per sweep 9 -> 4.
//   SCHED 2  (timing bound only, NOT an ordering: it does not converge) every step a DPP step, same partner every time -- what a sweep
//            without any LDS round trip would cost; no such nine-step ordering exists (jacobi_orderings.py)
//   CS 0     rotation from t = g2 / (|d| + h): v_rsq, v_rcp, v_rsq (pair_cs of quad_kernel.h)
//   CS 1     rotation from the half-angle identities: c^2 = (h + |d|) / 2h, s c = g2 / 2h: two v_rsq, no v_rcp
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o eig16x eig16x.hip ; run: ./eig16x [batch] [max_sweeps] [tol] [waves_per_simd]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

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
// alternating ordering: bits 4s..4s+3 = partner lane of bpermute step s (s = 0..3), bit 16+s = this lane takes over the partner's column
__device__ const unsigned kATab[16] = {0x98692u, 0x67343u, 0xd6960u, 0xe5181u, 0xf9716u, 0xf3878u, 0x92024u, 0x61459u, 0xf0535u, 0xc4207u,
                                       0x0aaaau, 0x0bbbbu, 0x0ccccu, 0x0ddddu, 0x0eeeeu, 0x0ffffu};

template <int CTRL> __device__ __forceinline__ float dppf(float x) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, true)); }
template <int CTRL> __device__ __forceinline__ double dppd(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm(int addr, double v)
{
    const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float bpermf(int addr, float v) { return __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(v))); }


// ---- DPP as an OPERAND of the arithmetic (VOP2 DPP encoding): hipcc 7.2 does not fold v_mov_b32_dpp into the consuming FMA
// (GCNDPPCombine leaves all of them), so the two blocks are written by hand.  s_nop 4 in front: the hazard recogniser does not look
// inside inline asm (VALU write -> DPP read of the same VGPR needs 2 wait states, EXEC write -> DPP 5).
#define DPP_XOR1 "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:0"
// sum_i q_i * q_i[partner]
#define DOT10_DPP(CTRL, g0, g1, e)                                                                                                  \
    asm volatile("s_nop 4\n\t"                                                                                                      \
                 "v_mul_f32_dpp %0, %2, %2 " CTRL "\n\tv_mul_f32_dpp %1, %3, %3 " CTRL "\n\t"                                       \
                 "v_fmac_f32_dpp %0, %4, %4 " CTRL "\n\tv_fmac_f32_dpp %1, %5, %5 " CTRL "\n\t"                                     \
                 "v_fmac_f32_dpp %0, %6, %6 " CTRL "\n\tv_fmac_f32_dpp %1, %7, %7 " CTRL "\n\t"                                     \
                 "v_fmac_f32_dpp %0, %8, %8 " CTRL "\n\tv_fmac_f32_dpp %1, %9, %9 " CTRL "\n\t"                                     \
                 "v_fmac_f32_dpp %0, %10, %10 " CTRL "\n\tv_fmac_f32_dpp %1, %11, %11 " CTRL                                        \
                 : "=&v"(g0), "=&v"(g1)                                                                                             \
                 : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(e[4]), "v"(e[5]), "v"(e[6]), "v"(e[7]), "v"(e[8]), "v"(e[9]))
// n_i += k * q_i[partner]
#define AXPY10_DPP(CTRL, n, e, k)                                                                                                   \
    asm volatile("s_nop 4\n\t"                                                                                                      \
                 "v_fmac_f32_dpp %0, %10, %20 " CTRL "\n\tv_fmac_f32_dpp %1, %11, %20 " CTRL "\n\t"                                 \
                 "v_fmac_f32_dpp %2, %12, %20 " CTRL "\n\tv_fmac_f32_dpp %3, %13, %20 " CTRL "\n\t"                                 \
                 "v_fmac_f32_dpp %4, %14, %20 " CTRL "\n\tv_fmac_f32_dpp %5, %15, %20 " CTRL "\n\t"                                 \
                 "v_fmac_f32_dpp %6, %16, %20 " CTRL "\n\tv_fmac_f32_dpp %7, %17, %20 " CTRL "\n\t"                                 \
                 "v_fmac_f32_dpp %8, %18, %20 " CTRL "\n\tv_fmac_f32_dpp %9, %19, %20 " CTRL                                        \
                 : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]), "+v"(n[4]), "+v"(n[5]), "+v"(n[6]), "+v"(n[7]), "+v"(n[8]), "+v"(n[9]) \
                 : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(e[4]), "v"(e[5]), "v"(e[6]), "v"(e[7]), "v"(e[8]), "v"(e[9]), "v"(k))

__device__ __forceinline__ double rsq64(double x)
{
    double z = (double)__builtin_amdgcn_rsqf((float)x);
    { double hh = 0.5 * x * z; double e = fma(-hh, z, 0.5); z = fma(z, e, z); }
    { double hh = 0.5 * x * z; double e = fma(-hh, z, 0.5); z = fma(z, e, z); }
    return z;
}
__device__ __forceinline__ double rcp64(double x)
{
    double z = (double)__builtin_amdgcn_rcpf((float)x);
    { double e = fma(-x, z, 1.0); z = fma(z, e, z); }
    { double e = fma(-x, z, 1.0); z = fma(z, e, z); }
    return z;
}

// rotation seen from one lane of the pair: d = |other|^2 - |own|^2, gam = own . other; own' = c own - s other
template <int CS> __device__ __forceinline__ void pair_cs(float d, float gam, bool rot, bool tie_neg, float &c, float &s, float &t)
{
    const float g2 = 2.0f * gam;
    const float h2 = d * d + g2 * g2 + 1e-37f;
    const bool neg = d < 0.0f || (d == 0.0f && tie_neg);
    if (CS == 0) {
        const float hf = h2 * __builtin_amdgcn_rsqf(h2);
        float tf = g2 * __builtin_amdgcn_rcpf(fabsf(d) + hf);
        tf = neg ? -tf : tf;
        t = rot ? tf : 0.0f;
        c = __builtin_amdgcn_rsqf(1.0f + t * t);
        s = t * c;
    } else {
        const float rh = __builtin_amdgcn_rsqf(h2);          // 1 / h
        const float c2 = fmaf(0.5f * fabsf(d), rh, 0.5f);     // cos^2 = (h + |d|) / 2h  in [1/2, 1]
        const float rc = __builtin_amdgcn_rsqf(c2);           // 1 / c
        const float sc = 0.5f * g2 * rh;                      // sin cos (sign of gam)
        float sf = sc * rc;                                   // sin
        sf = neg ? -sf : sf;
        c = rot ? c2 * rc : 1.0f;
        s = rot ? sf : 0.0f;
        t = s * rc;
    }
}
template <int CS> __device__ __forceinline__ void pair_cs(double d, double gam, bool rot, bool tie_neg, double &c, double &s, double &t)
{
    const double g2 = 2.0 * gam;
    const double h2 = d * d + g2 * g2 + 1e-290;
    const bool neg = d < 0.0 || (d == 0.0 && tie_neg);
    if (CS == 0) {
        const double h = h2 * rsq64(h2);
        double tf = g2 * rcp64(fabs(d) + h);
        tf = neg ? -tf : tf;
        t = rot ? tf : 0.0;
        c = rsq64(1.0 + t * t);
        s = t * c;
    } else {
        const double rh = rsq64(h2);
        const double c2 = fma(0.5 * fabs(d), rh, 0.5);
        const double rc = rsq64(c2);
        double sf = 0.5 * g2 * rh * rc;
        sf = neg ? -sf : sf;
        c = rot ? c2 * rc : 1.0;
        s = rot ? sf : 0.0;
        t = s * rc;
    }
}

template <class T> struct Col;
template <> struct Col<float> {
    f2 q[5];
    __device__ __forceinline__ float norm2() const { f2 a = q[0] * q[0];
#pragma unroll
        for (int i = 1; i < 5; ++i) a = __builtin_elementwise_fma(q[i], q[i], a);
        return a.x + a.y; }
    __device__ __forceinline__ float get(int i) const { return (i & 1) ? q[i >> 1].y : q[i >> 1].x; }
    __device__ __forceinline__ void set(int i, float v) { if (i & 1) q[i >> 1].y = v; else q[i >> 1].x = v; }
};
template <> struct Col<double> {
    double q[10];
    __device__ __forceinline__ double norm2() const { double a = 0;
#pragma unroll
        for (int i = 0; i < 10; ++i) a = fma(q[i], q[i], a);
        return a; }
    __device__ __forceinline__ double get(int i) const { return q[i]; }
    __device__ __forceinline__ void set(int i, double v) { q[i] = v; }
};

// ---- one ds_bpermute step: partner lane `partner` (row-relative), take = this lane ends up with the partner's rotated column
template <int CS> __device__ __forceinline__ void step_bperm(Col<float> &x, float &al, int addr, bool real, bool tie_neg, bool take, bool active, float tol2, bool &coarse)
{
    f2 o[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) { o[i].x = bpermf(addr, x.q[i].x); o[i].y = bpermf(addr, x.q[i].y); }
    const float be = bpermf(addr, al);
    f2 acc = x.q[0] * o[0];
#pragma unroll
    for (int i = 1; i < 5; ++i) acc = __builtin_elementwise_fma(x.q[i], o[i], acc);
    const float gam = acc.x + acc.y;
    const float g2 = gam * gam, ab = al * be;
    coarse |= real && g2 > tol2 * ab;
    float c, s, t;
    pair_cs<CS>(be - al, gam, active && real && g2 > 1e-30f * ab, tie_neg, c, s, t);
    // own' = c own - s other ; partner' = s own + c other
    const float ka = take ? s : c, kb = take ? c : -s;
    const f2 ka2 = {ka, ka}, kb2 = {kb, kb};
#pragma unroll
    for (int i = 0; i < 5; ++i) x.q[i] = __builtin_elementwise_fma(ka2, x.q[i], kb2 * o[i]);
    al = take ? fmaf(t, gam, be) : fmaf(-t, gam, al);
}
template <int CS> __device__ __forceinline__ void step_bperm(Col<double> &x, double &al, int addr, bool real, bool tie_neg, bool take, bool active, double tol2, bool &coarse)
{
    double o[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) o[i] = bperm(addr, x.q[i]);
    const double be = bperm(addr, al);
    double gam = 0.0;
#pragma unroll
    for (int i = 0; i < 10; ++i) gam = fma(x.q[i], o[i], gam);
    const double g2 = gam * gam, ab = al * be;
    coarse |= real && g2 > tol2 * ab;
    double c, s, t;
    pair_cs<CS>(be - al, gam, active && real && g2 > 1e-30 * ab, tie_neg, c, s, t);
    const double ka = take ? s : c, kb = take ? c : -s;
#pragma unroll
    for (int i = 0; i < 10; ++i) x.q[i] = fma(ka, x.q[i], kb * o[i]);
    al = take ? fma(t, gam, be) : fma(-t, gam, al);
}

// ---- one DPP step on lanes (2k, 2k+1): no exchange instruction in single precision (the partner's element is the DPP operand of the FMA)
template <int CS> __device__ __forceinline__ void step_dpp(Col<float> &x, float &al, bool real, bool tie_neg, bool active, float tol2, bool &coarse)
{
    float e[10], g0, g1;
#pragma unroll
    for (int i = 0; i < 5; ++i) { e[2 * i] = x.q[i].x; e[2 * i + 1] = x.q[i].y; }
    DOT10_DPP(DPP_XOR1, g0, g1, e);
    const float gam = g0 + g1;
    const float be = dppf<0xB1>(al);
    const float g2 = gam * gam, ab = al * be;
    coarse |= real && g2 > tol2 * ab;
    float c, s, t;
    pair_cs<CS>(be - al, gam, active && real && g2 > 1e-30f * ab, tie_neg, c, s, t);
    const f2 cc = {c, c};
    const float ms = -s;
    float n[10];
#pragma unroll
    for (int i = 0; i < 5; ++i) { const f2 p = cc * x.q[i]; n[2 * i] = p.x; n[2 * i + 1] = p.y; }
    AXPY10_DPP(DPP_XOR1, n, e, ms);
#pragma unroll
    for (int i = 0; i < 5; ++i) { x.q[i].x = n[2 * i]; x.q[i].y = n[2 * i + 1]; }
    al = fmaf(-t, gam, al);
}
template <int CS> __device__ __forceinline__ void step_dpp(Col<double> &x, double &al, bool real, bool tie_neg, bool active, double tol2, bool &coarse)
{
    double o[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) o[i] = dppd<0xB1>(x.q[i]);
    const double be = dppd<0xB1>(al);
    double gam = 0.0;
#pragma unroll
    for (int i = 0; i < 10; ++i) gam = fma(x.q[i], o[i], gam);
    const double g2 = gam * gam, ab = al * be;
    coarse |= real && g2 > tol2 * ab;
    double c, s, t;
    pair_cs<CS>(be - al, gam, active && real && g2 > 1e-30 * ab, tie_neg, c, s, t);
#pragma unroll
    for (int i = 0; i < 10; ++i) x.q[i] = fma(c, x.q[i], -s * o[i]);
    al = fma(-t, gam, al);
}

// W: [batch][55] packed upper triangle (row-major); lam_out [batch][10]; g_out [batch][10 lanes][10 rows] (the rotated columns)
template <class T, int SCHED, int CS>
__global__ __launch_bounds__(64) void eigx_kernel(const double *W, double *lam_out, int *sw_out, double *g_out, int batch, int max_sweeps, double tol2d)
{
    extern __shared__ double occupancy_pad[]; // dynamic LDS only limits the wavefronts per CU (see main)
    const int lane = threadIdx.x & 63;
    const int gl = lane & 15, grp = lane >> 4;
    const long prob = (long)blockIdx.x * 4 + grp;
    const bool valid = prob < batch && gl < 10;
    const long pc = prob < batch ? prob : batch - 1;
    const int col = gl < 10 ? gl : 0;
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
    Col<T> x;
#pragma unroll
    for (int r = 0; r < 10; ++r) x.set(r, gl < 10 ? (T)(g[r] + ((r == col) ? sigma : 0.0)) : (T)0);
    T al = x.norm2();
    const unsigned long long ptab = kPTab.packed[gl];
    const unsigned atab = kATab[gl];
    const int base4 = (lane & 48) << 2;
    const T tol2 = (T)tol2d;
    const bool col_lane = gl < 10;
    int sweeps = 0;
    bool active = true; // row-uniform
    while (true) {
        bool coarse = false;
        if (SCHED == 0) {
#pragma unroll
            for (int st = 0; st < 9; ++st) {
                const int partner = (int)((ptab >> (4 * st)) & 15);
                step_bperm<CS>(x, al, base4 + (partner << 2), partner != gl, gl > partner, false, active, tol2, coarse);
            }
        } else if (SCHED == 1) {
#pragma unroll
            for (int st = 0; st < 9; ++st) {
                if ((st & 1) == 0) {
                    step_dpp<CS>(x, al, col_lane, (gl & 1) != 0, active, tol2, coarse);
                } else {
                    const int s = st >> 1;
                    const int partner = (int)((atab >> (4 * s)) & 15);
                    step_bperm<CS>(x, al, base4 + (partner << 2), partner != gl, gl > partner, ((atab >> (16 + s)) & 1) != 0, active, tol2, coarse);
                }
            }
        } else {
#pragma unroll
            for (int st = 0; st < 9; ++st) step_dpp<CS>(x, al, col_lane, (gl & 1) != 0, active, tol2, coarse);
        }
        al = x.norm2(); // exact norm once per sweep
        const unsigned long long m = __ballot(coarse && active);
        const bool grp_more = ((m >> (16 * grp)) & 0xFFFFull) != 0;
        if (active) ++sweeps;
        active = active && grp_more && sweeps < max_sweeps;
        if (!__any(active)) break;
    }
    if (valid) {
        lam_out[prob * 10 + gl] = sqrt((double)al) - sigma;
        if (gl == 0) sw_out[prob] = sweeps;
        if (g_out)
#pragma unroll
            for (int r = 0; r < 10; ++r) g_out[(prob * 10 + gl) * 10 + r] = (double)x.get(r);
    }
}

typedef void (*kern_t)(const double *, double *, int *, double *, int, int, double);
template <class T, int CS> static kern_t pick_sched(int sched)
{
    switch (sched) {
    case 0: return eigx_kernel<T, 0, CS>;
    case 1: return eigx_kernel<T, 1, CS>;
    default: return eigx_kernel<T, 2, CS>;
    }
}

int main(int argc, char **argv)
{
    const int batch = argc > 1 ? atoi(argv[1]) : 10000;
    const int max_sweeps = argc > 2 ? atoi(argv[2]) : 12;
    const double tol = argc > 3 ? atof(argv[3]) : 6e-2;
    const int wps = argc > 4 ? atoi(argv[4]) : 2; // wavefronts per SIMD the dynamic LDS allows
    std::vector<double> W((size_t)batch * 55);
    srand(1);
    for (auto &v : W) v = (double)rand() / RAND_MAX - 0.5;
    double *dW, *dl, *dg;
    int *dsw;
    CHECK(hipMalloc(&dW, W.size() * 8));
    CHECK(hipMalloc(&dl, (size_t)batch * 80));
    CHECK(hipMalloc(&dg, (size_t)batch * 800));
    CHECK(hipMalloc(&dsw, (size_t)batch * 4));
    CHECK(hipMemcpy(dW, W.data(), W.size() * 8, hipMemcpyHostToDevice));
    const int blocks = (batch + 3) / 4;
    const size_t dyn_lds = (size_t)(160 * 1024) / (4 * wps) - 256; // per workgroup of one wavefront: 4 * wps workgroups fit a CU
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int prec = 0; prec < 2; ++prec)
        for (int cs = 0; cs < 2; ++cs)
            for (int sched = 0; sched < 3; ++sched) {
                kern_t k = prec == 0 ? (cs == 0 ? pick_sched<float, 0>(sched) : pick_sched<float, 1>(sched)) : (cs == 0 ? pick_sched<double, 0>(sched) : pick_sched<double, 1>(sched));
                CHECK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn_lds));
                for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), dyn_lds, 0, dW, dl, dsw, dg, batch, max_sweeps, tol * tol);
                CHECK(hipDeviceSynchronize());
                const int reps = 20;
                CHECK(hipEventRecord(e0));
                for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(64), dyn_lds, 0, dW, dl, dsw, (double *)nullptr, batch, max_sweeps, tol * tol);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                std::vector<double> lam((size_t)batch * 10), gg((size_t)batch * 100);
                std::vector<int> sw(batch);
                CHECK(hipMemcpy(lam.data(), dl, lam.size() * 8, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(gg.data(), dg, gg.size() * 8, hipMemcpyDeviceToHost));
                CHECK(hipMemcpy(sw.data(), dsw, sw.size() * 4, hipMemcpyDeviceToHost));
                double tr_err = 0, mo = 0, msw = 0;
                int maxsw = 0;
                for (int b = 0; b < batch; ++b) {
                    double tr = 0, sl = 0;
                    int k2 = 0;
                    for (int i = 0; i < 10; ++i) for (int j = i; j < 10; ++j) { if (i == j) tr += W[(size_t)b * 55 + k2]; ++k2; }
                    for (int j = 0; j < 10; ++j) sl += lam[(size_t)b * 10 + j];
                    tr_err = fmax(tr_err, fabs(tr - sl));
                    msw += sw[b];
                    maxsw = sw[b] > maxsw ? sw[b] : maxsw;
                    if (b < 2000) { // all 45 cosines of the returned columns
                        const double *G = &gg[(size_t)b * 100];
                        double nn[10];
                        for (int j = 0; j < 10; ++j) { nn[j] = 0; for (int r = 0; r < 10; ++r) nn[j] += G[j * 10 + r] * G[j * 10 + r]; }
                        for (int j = 0; j < 10; ++j) for (int k3 = j + 1; k3 < 10; ++k3) {
                            double d = 0; for (int r = 0; r < 10; ++r) d += G[j * 10 + r] * G[k3 * 10 + r];
                            mo = fmax(mo, fabs(d) / sqrt(nn[j] * nn[k3]));
                        }
                    }
                }
                printf("{\"prec\": \"%s\", \"cs\": %d, \"sched\": %d, \"batch\": %d, \"waves_per_simd\": %d, \"max_sweeps\": %d, \"tol\": %g, \"ms_per_launch\": %.4f, \"mean_sweeps\": %.3f, \"max_sweeps_used\": %d, "
                       "\"us_per_sweep_launch\": %.3f, \"trace_err\": %.2e, \"max_cos_all_pairs\": %.2e}\n",
                       prec == 0 ? "f32" : "f64", cs, sched, batch, wps, max_sweeps, tol, ms / reps, msw / batch, maxsw, (ms / reps * 1e3) / (msw / batch), tr_err, mo);
                fflush(stdout);
            }
    return 0;
}
