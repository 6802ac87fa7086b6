// Is v_sqrt_f32 on gfx950 correctly rounded?  Exhaustive over every non-negative finite f32 (normal and denormal):
// compares the raw instruction with f32(sqrt(f64(x))) — correctly rounded, because a 53-bit square root rounds to 24 bits
// without double-rounding trouble — and counts the inputs where they differ, per binade.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off tools/sqrt_exhaustive.hip -o /tmp/sqrt_ex && /tmp/sqrt_ex
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_check(unsigned long long *bad_total, unsigned *bad_per_exp, unsigned *first_bad)
{
    const unsigned long long i0 = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 256ull;
    unsigned bad = 0;
    for (unsigned k = 0; k < 256; ++k) {
        const unsigned bits = (unsigned)(i0 + k);
        if (bits >= 0x7f800000u) break;                          // inf / nan
        const float x = __uint_as_float(bits);
        const float hw = __builtin_amdgcn_sqrtf(x);
        const float cr = (float)sqrt((double)x);
        if (__float_as_uint(hw) != __float_as_uint(cr)) {
            ++bad;
            atomicAdd(&bad_per_exp[bits >> 23], 1u);
            atomicMin(first_bad, bits);
        }
    }
    if (bad) atomicAdd(bad_total, (unsigned long long)bad);
}

// The product's square root (k_corners.hip sqrt_rn_normal, round 4: v_rsq_f32 + one Newton correction with fused residual) over every
// input whose root matters: the normal range from 2^-100 up (in the kernel x is zero - clamped to 2^-120, where any root near 2^-60
// serves - or >= 1e-16).  Below 2^-100 the fused residual runs into the denormals and the sequence is NOT exact (16.8 M mismatches
// between 2^-122 and 2^-100): the clamp must stay the only way into that range.
__device__ __forceinline__ float sqrt_rn_normal(float x)
{
    const float xc = fmaxf(x, 0x1p-120f);
    const float y = __builtin_amdgcn_rsqf(xc);
    const float s = xc * y;
    const float r = __builtin_fmaf(-s, s, xc);
    return __builtin_fmaf(r, 0.5f * y, s);
}

__global__ void k_check_fixup(unsigned long long *bad_total, unsigned *first_bad)
{
    const unsigned long long i0 = ((unsigned long long)blockIdx.x * blockDim.x + threadIdx.x) * 256ull;
    unsigned bad = 0;
    for (unsigned k = 0; k < 256; ++k) {
        const unsigned bits = (unsigned)(i0 + k);
        if (bits >= 0x7f000000u) break;
        if (bits < (27u << 23)) continue;                        // below 2^-100: outside the contract (zero is clamped to 2^-120 inside the function)
        const float x = __uint_as_float(bits);
        if (__float_as_uint(sqrt_rn_normal(x)) != __float_as_uint((float)sqrt((double)x))) { ++bad; atomicMin(first_bad, bits); }
    }
    if (bad) atomicAdd(bad_total, (unsigned long long)bad);
}

int main()
{
    unsigned long long *d_total, total = 0;
    unsigned *d_exp, *d_first, per_exp[256], first = 0xffffffffu;
    hipMalloc(&d_total, 8); hipMalloc(&d_exp, 256 * 4); hipMalloc(&d_first, 4);
    hipMemset(d_total, 0, 8); hipMemset(d_exp, 0, 256 * 4); hipMemcpy(d_first, &first, 4, hipMemcpyHostToDevice);
    const unsigned long long n = 0x7f800000ull;                  // all non-negative finite floats
    const unsigned blocks = (unsigned)((n / 256 + 255) / 256);
    hipLaunchKernelGGL(k_check, dim3(blocks), dim3(256), 0, 0, d_total, d_exp, d_first);
    hipDeviceSynchronize();
    hipMemcpy(&total, d_total, 8, hipMemcpyDeviceToHost); hipMemcpy(per_exp, d_exp, 256 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost);
    printf("inputs checked: %llu, v_sqrt_f32 != correctly rounded: %llu (first at bits 0x%08x)\n", n, total, first);
    int shown = 0;
    for (int e = 0; e < 255 && shown < 12; ++e)
        if (per_exp[e]) { printf("  biased exponent %3d: %u mismatches of 8388608\n", e, per_exp[e]); ++shown; }
    total = 0; first = 0xffffffffu;
    hipMemset(d_total, 0, 8); hipMemcpy(d_first, &first, 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_check_fixup, dim3(blocks), dim3(256), 0, 0, d_total, d_first);
    hipDeviceSynchronize();
    hipMemcpy(&total, d_total, 8, hipMemcpyDeviceToHost); hipMemcpy(&first, d_first, 4, hipMemcpyDeviceToHost);
    printf("sqrt_rn_normal (product: rsq + one Newton step), 2^-100 .. 2^127, != correctly rounded: %llu (first at bits 0x%08x)\n", total, first);
    return total ? 1 : 0;
}
