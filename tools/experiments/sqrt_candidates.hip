// Cheaper correctly rounded square roots for the response kernel's lambda_min (round 4 experiment, exhaustive on the GPU):
//   A  y = v_rsq(x), s = x y, r = fma(-s, s, x), s' = fma(r, 0.5 y, s)                       (5 instructions; product fix-up: 8)
//   A2 the same with s' = fma(r y, 0.5, s)
//   C  sidedness of v_sqrt_f32's error: if it never errs upwards (or never downwards) one residual is enough
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/experiments/sqrt_candidates.hip -o build_variants/sqrt_cand && build_variants/sqrt_cand
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ float cand_a(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x * y;
    const float r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r, 0.5f * y, s);
}
__device__ __forceinline__ float cand_a2(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float s = x * y;
    const float r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r * y, 0.5f, s);
}
// two Newton corrections (7 instructions)
__device__ __forceinline__ float cand_a3(float x)
{
    const float y = __builtin_amdgcn_rsqf(x);
    float s = x * y;
    const float h = 0.5f * y;
    float r = __builtin_fmaf(-s, s, x);
    s = __builtin_fmaf(r, h, s);
    r = __builtin_fmaf(-s, s, x);
    return __builtin_fmaf(r, h, s);
}
// v_sqrt + ONE residual at s: result = s + sign-step chosen by comparing the residual with s * ulp(s) ... (sidedness probe only)

__global__ void k_check(unsigned long long *bad, unsigned *first_bad)
{
    const unsigned long long i0 = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 256ull;
    unsigned b[6] = {0, 0, 0, 0, 0, 0};
    for (unsigned k = 0; k < 256; ++k) {
        const unsigned bits = (unsigned)(i0 + k);
        if (bits >= 0x7f000000u) break;
        if (bits < (27u << 23)) continue;                        // below 2^-100: outside the contract (x = 0 is guarded by a clamp)
        const float x = __uint_as_float(bits);
        const unsigned cr = __float_as_uint((float)sqrt((double)x));
        const unsigned hw = __float_as_uint(__builtin_amdgcn_sqrtf(x));
        if (__float_as_uint(cand_a(x)) != cr) { ++b[0]; atomicMin(first_bad + 0, bits); }
        if (__float_as_uint(cand_a2(x)) != cr) { ++b[1]; atomicMin(first_bad + 1, bits); }
        if (__float_as_uint(cand_a3(x)) != cr) { ++b[2]; atomicMin(first_bad + 2, bits); }
        if (hw > cr) ++b[3];
        if (hw < cr) ++b[4];
        if (hw > cr + 1 || hw + 1 < cr) ++b[5];
    }
    for (int j = 0; j < 6; ++j) if (b[j]) atomicAdd(bad + j, (unsigned long long)b[j]);
}

int main()
{
    unsigned long long *d_bad, bad[6] = {0};
    unsigned *d_first, first[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
    hipMalloc(&d_bad, 48); hipMalloc(&d_first, 12);
    hipMemset(d_bad, 0, 48); hipMemcpy(d_first, first, 12, hipMemcpyHostToDevice);
    const unsigned long long n = 0x7f000000ull;
    hipLaunchKernelGGL(k_check, dim3((unsigned)((n / 256 + 255) / 256)), dim3(256), 0, 0, d_bad, d_first);
    hipDeviceSynchronize();
    hipMemcpy(bad, d_bad, 48, hipMemcpyDeviceToHost); hipMemcpy(first, d_first, 12, hipMemcpyDeviceToHost);
    printf("inputs 2^-100 .. 2^127\n");
    printf("A  rsq, mul, fma, mul, fma      : %llu mismatches (first 0x%08x)\n", bad[0], first[0]);
    printf("A2 rsq, mul, fma, mul, fma (r y): %llu mismatches (first 0x%08x)\n", bad[1], first[1]);
    printf("A3 two corrections              : %llu mismatches (first 0x%08x)\n", bad[2], first[2]);
    printf("v_sqrt_f32 above the correctly rounded root: %llu, below: %llu, off by more than one ulp: %llu\n", bad[3], bad[4], bad[5]);
    return 0;
}
