// rsq_probe.hip -- accuracy of the float64 reciprocal-root sequences of cvx::rsqrt_ (hardware seed + two Newton steps) and of a single third-order step
// (y (1 + r / 2 + 3 r^2 / 8), r = 1 - x y^2), and of the raw seeds v_rsq_f64 / v_rcp_f64, against the correctly rounded values; issue cost of both sequences
// on a dependent chain.  Build: hipcc --offload-arch=gfx950 -O3 -o rsq_probe rsq_probe.hip ; run: ./rsq_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double rsqrt_n2(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    { double h = 0.5 * x * y; double e = fma(-h, y, 0.5); y = fma(y, e, y); }
    { double h = 0.5 * x * y; double e = fma(-h, y, 0.5); y = fma(y, e, y); }
    return y;
}
__device__ __forceinline__ double rsqrt_c3(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double r = fma(-(x * y), y, 1.0);
    return fma(y * r, fma(0.375, r, 0.5), y);
}
__global__ void probe(const double *x, double *seed_rsq, double *seed_rcp, double *n2, double *c3, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    seed_rsq[i] = __builtin_amdgcn_rsq(x[i]);
    seed_rcp[i] = __builtin_amdgcn_rcp(x[i]);
    n2[i] = rsqrt_n2(x[i]);
    c3[i] = rsqrt_c3(x[i]);
}
__global__ void __launch_bounds__(64, 1) chain(double *out, long long *cyc, double seed)
{
    double x = seed + threadIdx.x * 1e-3;
    long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
#pragma unroll 1
    for (int i = 0; i < 256; ++i) { x = rsqrt_n2(x) + 1.5; asm volatile("" : "+v"(x)); }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : : "memory");
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) : : "memory");
#pragma unroll 1
    for (int i = 0; i < 256; ++i) { x = rsqrt_c3(x) + 1.5; asm volatile("" : "+v"(x)); }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) : : "memory");
    if (threadIdx.x == 0) cyc[1] = t1 - t0;
    out[threadIdx.x] = x;
}
int main()
{
    const int n = 1 << 22;
    std::vector<double> hx(n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { // log-uniform over 1e-12 ... 1e12, every mantissa
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double u = (double)(s >> 11) / 9007199254740992.0;
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        const double v = (double)(s >> 11) / 9007199254740992.0;
        hx[i] = std::exp((u * 2.0 - 1.0) * 27.6) * (1.0 + v);
    }
    double *dx, *d0, *d1, *d2, *d3;
    hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8); hipMalloc(&d3, n * 8);
    hipMemcpy(dx, hx.data(), n * 8, hipMemcpyHostToDevice);
    probe<<<n / 256, 256>>>(dx, d0, d1, d2, d3, n);
    std::vector<double> r0(n), r1(n), r2(n), r3(n);
    hipMemcpy(r0.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), d1, n * 8, hipMemcpyDeviceToHost);
    hipMemcpy(r2.data(), d2, n * 8, hipMemcpyDeviceToHost); hipMemcpy(r3.data(), d3, n * 8, hipMemcpyDeviceToHost);
    double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
    for (int i = 0; i < n; ++i) {
        const long double t = 1.0L / sqrtl((long double)hx[i]), q = 1.0L / (long double)hx[i];
        e0 = std::fmax(e0, (double)fabsl((r0[i] - t) / t)); e1 = std::fmax(e1, (double)fabsl((r1[i] - q) / q));
        e2 = std::fmax(e2, (double)fabsl((r2[i] - t) / t)); e3 = std::fmax(e3, (double)fabsl((r3[i] - t) / t));
    }
    printf("max relative error over %d values: v_rsq_f64 seed %.3e (2^%.1f), v_rcp_f64 seed %.3e (2^%.1f), seed + two Newton steps %.3e, seed + one third-order step %.3e (1 ulp = 1.1e-16 ... 2.2e-16)\n",
           n, e0, std::log2(e0), e1, std::log2(e1), e2, e3);
    long long *cyc; double *out;
    hipMalloc(&cyc, 16); hipMalloc(&out, 64 * 8);
    chain<<<1, 64>>>(out, cyc, 2.0);
    long long hc[2];
    hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
    printf("dependent chain, one wavefront: rsq + two Newton steps + add %.1f cycles, rsq + third-order step + add %.1f cycles\n", hc[0] / 256.0, hc[1] / 256.0);
    return 0;
}
