// Probe: which cross-lane primitives work on this GPU (run on the GPU box: hipcc --offload-arch=gfx950 tools/dpp_probe.hip -o /tmp/p && /tmp/p)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out)
{
    const int v = threadIdx.x + 100;
    out[threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x138, 0xF, 0xF, true);          // wave_shr:1
    out[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x130, 0xF, 0xF, true);     // wave_shl:1
    out[128 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);    // row_shr:1
    out[192 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x101, 0xF, 0xF, true);    // row_shl:1
    out[256 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);   // row_bcast15
    out[320 + threadIdx.x] = __shfl_up(v, 1);
    out[384 + threadIdx.x] = __builtin_amdgcn_update_dpp(0, v, 0x13C, 0xF, 0xF, true);    // wave_ror:1
}
int main()
{
    int *d; hipMalloc(&d, 448 * 4); hipMemset(d, 0, 448 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[448]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    const char *names[] = {"wave_shr1", "wave_shl1", "row_shr1", "row_shl1", "row_bcast15", "shfl_up1", "wave_ror1"};
    for (int t = 0; t < 7; ++t) { printf("%-12s", names[t]); for (int i : {0, 1, 2, 15, 16, 17, 31, 32, 47, 48, 62, 63}) printf(" [%d]=%d", i, h[t * 64 + i]); printf("\n"); }
    return 0;
}
