// lat_probe.hip -- instruction latencies / issue rates that bound the dependent chains of the interior-point kernel (gfx950, one wavefront
// per SIMD): float64 FMA (dependent chain vs eight independent chains), v_rcp_f64 + Newton, ds_bpermute round trip, LDS read round trip,
// DPP row reduction.  Build: hipcc --offload-arch=gfx950 -O3 -o lat_probe lat_probe.hip ; run: ./lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N 512
__device__ __forceinline__ long long now_(double &x)
{
    long long t;
    asm volatile("" : "+v"(x));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
    asm volatile("" : "+v"(x));
    return t;
}

__global__ void __launch_bounds__(64, 1) probe(double *out, long long *cyc, double seed)
{
    __shared__ double lds[256];
    const int lane = threadIdx.x;
    lds[lane] = seed + lane; lds[lane + 64] = seed; lds[lane + 128] = seed; lds[lane + 192] = seed;
    __syncthreads();
    double x = seed + lane * 1e-3, y = 1.0 + 1e-9 * lane;
    long long t0, t1;
    // 0: dependent FMA chain
    t0 = now_(x);
#pragma unroll
    for (int i = 0; i < N; ++i) x = fma(x, y, 1e-9);
    t1 = now_(x); if (lane == 0) cyc[0] = t1 - t0;
    // 1: eight independent chains
    double a[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] = x + k;
    t0 = now_(x);
#pragma unroll
    for (int i = 0; i < N / 8; ++i)
#pragma unroll
        for (int k = 0; k < 8; ++k) a[k] = fma(a[k], y, 1e-9);
#pragma unroll
    for (int k = 0; k < 8; ++k) x += a[k];
    t1 = now_(x); if (lane == 0) cyc[1] = t1 - t0;
    // 2: dependent ds_bpermute round trips (one double = two dwords)
    const int addr = ((lane & 48) | ((lane + 1) & 15)) << 2;
    t0 = now_(x);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(x));
        const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(x));
        x = __hiloint2double(hi, lo) + 1e-9;
    }
    t1 = now_(x); if (lane == 0) cyc[2] = t1 - t0;
    // 3: dependent LDS read round trips (address depends on the value read)
    int idx = lane;
    t0 = now_(x);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) { const double v = lds[idx & 255]; idx = (idx + (v > 1e300 ? 1 : 64)) & 255; }
    x += idx;
    t1 = now_(x); if (lane == 0) cyc[3] = t1 - t0;
    // 4: reciprocal (hardware seed + two Newton steps) chain
    t0 = now_(x);
#pragma unroll
    for (int i = 0; i < N / 8; ++i) {
        double r = __builtin_amdgcn_rcp(x);
        r = fma(fma(-x, r, 1.0), r, r);
        r = fma(fma(-x, r, 1.0), r, r);
        x = r + 1.5;
    }
    t1 = now_(x); if (lane == 0) cyc[4] = t1 - t0;
    // 5: DPP row reduction chain (4 steps, two v_mov_dpp + add each)
    t0 = now_(x);
#pragma unroll
    for (int i = 0; i < N / 8; ++i) {
#define DPPD(v, c) __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), c, 0xF, 0xF, true), __builtin_amdgcn_update_dpp(0, __double2loint(v), c, 0xF, 0xF, true))
        x += DPPD(x, 0xB1); x += DPPD(x, 0x4E); x += DPPD(x, 0x141); x += DPPD(x, 0x140);
        x *= 0.0625;
    }
    t1 = now_(x); if (lane == 0) cyc[5] = t1 - t0;
    // 6: float32 dependent FMA chain (for scale)
    float f = (float)x, g = 1.0f + 1e-7f * lane;
    t0 = now_(x);
#pragma unroll
    for (int i = 0; i < N; ++i) f = fmaf(f, g, 1e-9f);
    x += f;
    t1 = now_(x); if (lane == 0) cyc[6] = t1 - t0;
    // 7: float64 multiply-add pairs as the compiler emits them for s += a*b (dependent adds, independent multiplies)
    double s = 0.0;
    t0 = now_(x);
#pragma unroll
    for (int i = 0; i < N / 2; ++i) s += (x + i) * y;
    x += s;
    t1 = now_(x); if (lane == 0) cyc[7] = t1 - t0;
    // 8: dependent row_newbcast broadcasts of a double (two v_mov_b32_dpp, gfx90a+): lane 5 of every row to the whole row
    t0 = now_(x);
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x155, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x155, 0xF, 0xF, true);
        x = __hiloint2double(hi, lo) + 1e-9;
    }
    t1 = now_(x); if (lane == 0) cyc[8] = t1 - t0;
    // semantic check of row_newbcast:3 : every lane must read lane (lane & 48) + 3
    {
        const int got = __builtin_amdgcn_update_dpp(-1, lane, 0x153, 0xF, 0xF, false);
        if (blockIdx.x == 0) { const unsigned long long okm = __ballot(got == ((lane & 48) | 3)); if (lane == 0) cyc[9] = (long long)okm; }
    }
    out[blockIdx.x * 64 + lane] = x + f + s;
}

int main()
{
    double *out; long long *cyc;
    hipMalloc(&out, 1024 * 64 * 8); hipMalloc(&cyc, 128);
    for (int blocks : {1, 1024}) {
        hipLaunchKernelGGL(probe, dim3(blocks), dim3(64), 0, 0, out, cyc, 1.0);
        hipDeviceSynchronize();
        long long h[10];
        hipMemcpy(h, cyc, 80, hipMemcpyDeviceToHost);
        printf("{\"wavefronts\": %d, \"cycles_per\": {\"f64_fma_dependent\": %.1f, \"f64_fma_8_chains\": %.1f, \"bpermute_double_round_trip\": %.1f, \"lds_read_round_trip\": %.1f, \"rcp_newton2_chain\": %.1f, \"dpp_row_sum_f64\": %.1f, \"f32_fma_dependent\": %.1f, \"f64_dot_step\": %.1f, \"row_newbcast_double\": %.1f}, \"row_newbcast_3_ok_mask\": \"%llx\"}\n",
               blocks, h[0] / (double)N, h[1] / (double)N, h[2] / (double)(N / 4), h[3] / (double)(N / 4), h[4] / (double)(N / 8), h[5] / (double)(N / 8), h[6] / (double)N, h[7] / (double)(N / 2), h[8] / (double)(N / 4), (unsigned long long)h[9]);
    }
    return 0;
}
