// prints what each DPP control reads, lane by lane (row 0 of a wavefront): settles shift directions once and for all
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ int dpp(int x) { return __builtin_amdgcn_update_dpp(-1, x, CTRL, 0xF, 0xF, false); }
__global__ void probe(int *out)
{
    const int l = threadIdx.x;
    out[0 * 64 + l] = dpp<0x101>(l); // row_shl:1
    out[1 * 64 + l] = dpp<0x111>(l); // row_shr:1
    out[2 * 64 + l] = dpp<0x121>(l); // row_ror:1
    out[3 * 64 + l] = dpp<0xB1>(l);  // quad_perm [1,0,3,2]
    out[4 * 64 + l] = dpp<0x140>(l); // row_mirror
    out[5 * 64 + l] = dpp<0x141>(l); // row_half_mirror
    out[6 * 64 + l] = dpp<0x128>(l); // row_ror:8
    float a = (float)l, r;
    asm volatile("s_nop 4\n\tv_mul_f32_dpp %0, %1, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=&v"(r) : "v"(a), "v"(1.0f));
    out[7 * 64 + l] = (int)r;
}
int main()
{
    int *d, h[8 * 64];
    hipMalloc(&d, sizeof h);
    probe<<<1, 64>>>(d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[8] = {"row_shl:1", "row_shr:1", "row_ror:1", "quad[1,0,3,2]", "row_mirror", "row_half_mirror", "row_ror:8", "asm mul row_shl:1"};
    for (int k = 0; k < 8; ++k) { printf("%-18s", names[k]); for (int l = 0; l < 20; ++l) printf(" %3d", h[k * 64 + l]); printf("\n"); }
    return 0;
}
