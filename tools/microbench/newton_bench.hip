// newton_bench.hip -- NEEDS tools/experiments/patches/r06_coop_newton_device.patch applied (git apply): cvxw::coop_newton (wave_kernel.h: the barrier Newton solve of the dual, one wavefront per problem) alone, on failed duals
// recorded by the host build (tools/microbench/newton_records.npy: S1 + delta I packed, R, delta, bottom eigenvalue, iteration), with the
// shader clock around it.  tools/microbench/newton_bench.py drives it and holds the result against cvx::dual_newton (host).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I cvxpnpl_amd/csrc -o libnewton_bench.so newton_bench.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ long long g_nclk[8 * 512];
#define CVXW_NCLK_INIT long long nclk_t = __builtin_readcyclecounter();
#define CVXW_NCLK(i) do { if (lane == 0) { const long long t_ = __builtin_readcyclecounter(); g_nclk[blockIdx.x * 8 + (i)] += t_ - nclk_t; nclk_t = t_; } } while (0)
#define CVXW_NO_KERNELS
#include "problem_io.h"
#include "solver_core.h"
#include "ipm_core.h"
#include "wave_kernel.h"

__global__ void __launch_bounds__(64, 2) newton_bench_kernel(int n, const double *S55, const double *R9, const double *delta_, const double *lam_, double *out)
{
    __shared__ __attribute__((aligned(16))) double L[cvxw::LDSW];
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= n) return;
    const unsigned lw = cvxw::kLanePack.w[lane];
    const int ei = (int)(lw & 15), ej = (int)((lw >> 4) & 15);
    const int el = lane < 55 ? lane : lane - 55;
    const double delta = delta_[b];
    const double S = S55[b * 55 + el] - (ei == ej ? delta : 0.0);
    if (lane < 9) L[cvxw::C_RL + lane] = R9[b * 9 + lane];
    __syncthreads();
    double zSz = 0.0;
    const long long t0 = __builtin_readcyclecounter();
    const double mp = cvxw::coop_newton(L, lane, ei, ej, S, delta, lam_[b], &zSz);
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) { out[b * 4] = mp; out[b * 4 + 1] = zSz; out[b * 4 + 2] = (double)(t1 - t0); out[b * 4 + 3] = 0.0; }
}

extern "C" int newton_bench_run(int n, const double *S55, const double *R9, const double *delta, const double *lam, double *out, int one_at_a_time)
{
    if (one_at_a_time) { // each problem alone on the device: the chain, not the throughput
        for (int b = 0; b < n; ++b) hipLaunchKernelGGL(newton_bench_kernel, dim3(1), dim3(64), 0, 0, 1, S55 + b * 55, R9 + b * 9, delta + b, lam + b, out + b * 4);
    } else hipLaunchKernelGGL(newton_bench_kernel, dim3(n), dim3(64), 0, 0, n, S55, R9, delta, lam, out);
    hipDeviceSynchronize();
    return (int)hipGetLastError();
}
extern "C" void newton_bench_clocks(long long *host, int zero)
{
    static long long z[8 * 512];
    if (zero) hipMemcpyToSymbol(HIP_SYMBOL(g_nclk), z, sizeof(z));
    else hipMemcpyFromSymbol(host, HIP_SYMBOL(g_nclk), sizeof(z));
}
