// Issue cost of the instructions the streaming kernels lean on, relative to v_add_u32 (gfx950).
//   hipcc --offload-arch=gfx950 -O2 tools/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
// Each kernel runs ITER x 32 independent instructions of one kind per wave, 8 waves per SIMD on every SIMD of the chip;
// the table prints time per wave-instruction per SIMD in ns and as a multiple of the v_add_u32 time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITER 4096

#define REP8(S, a) S(a##0) S(a##1) S(a##2) S(a##3) S(a##4) S(a##5) S(a##6) S(a##7)
#define KERNEL(NAME, ASM)                                                                                     \
    __global__ __launch_bounds__(256) void NAME(int *out, int seed)                                           \
    {                                                                                                         \
        int r0 = seed + threadIdx.x, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13, r6 = r0 * 17, r7 = r0 * 19; \
        int ad = ((threadIdx.x - 2) & 63) * 4;                                                                \
        asm volatile("s_mov_b64 s[10:11], 0x5555" ::: "s10", "s11");                                          \
        for (int it = 0; it < ITER; ++it) {                                                                   \
            asm volatile(ASM("%0") ASM("%1") ASM("%2") ASM("%3") ASM("%4") ASM("%5") ASM("%6") ASM("%7")          \
                         ASM("%0") ASM("%1") ASM("%2") ASM("%3") ASM("%4") ASM("%5") ASM("%6") ASM("%7")          \
                         ASM("%0") ASM("%1") ASM("%2") ASM("%3") ASM("%4") ASM("%5") ASM("%6") ASM("%7")          \
                         ASM("%0") ASM("%1") ASM("%2") ASM("%3") ASM("%4") ASM("%5") ASM("%6") ASM("%7")          \
                         : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)      \
                         : "v"(ad));                                                                          \
        }                                                                                                     \
        asm volatile("s_waitcnt lgkmcnt(0)");                                                                 \
        out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                          \
    }

#define A_ADD(r) "v_add_u32 " r ", " r ", " r "\n"
#define A_ADD3(r) "v_add3_u32 " r ", " r ", " r ", " r "\n"
#define A_MAX3(r) "v_max3_i32 " r ", " r ", " r ", " r "\n"
#define A_MULLO(r) "v_mul_lo_u32 " r ", " r ", " r "\n"
#define A_MUL24(r) "v_mul_i32_i24 " r ", " r ", " r "\n"
#define A_MAD24(r) "v_mad_i32_i24 " r ", " r ", " r ", " r "\n"
#define A_SQRT(r) "v_sqrt_f32 " r ", " r "\n"
#define A_CVT(r) "v_cvt_f32_i32 " r ", " r "\n"
#define A_FMA(r) "v_fma_f32 " r ", " r ", " r ", " r "\n"
#define A_MULF(r) "v_mul_f32 " r ", " r ", " r "\n"
#define A_CNDMASK(r) "v_cndmask_b32 " r ", " r ", " r ", vcc\n"
#define A_CMP(r) "v_cmp_gt_i32 vcc, " r ", " r "\n"
#define A_ADD_DPP_WSHR(r) "v_add_u32_dpp " r ", " r ", " r " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A_MOV_DPP_WSHR(r) "v_mov_b32_dpp " r ", " r " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A_MOV_DPP_WSHL(r) "v_mov_b32_dpp " r ", " r " wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A_ADD_DPP_RSHR(r) "v_add_u32_dpp " r ", " r ", " r " row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A_ADD_DPP_QUAD(r) "v_add_u32_dpp " r ", " r ", " r " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A_BPERM(r) "ds_bpermute_b32 " r ", %8, " r "\n"
#define A_ALIGNBIT(r) "v_alignbit_b32 " r ", " r ", " r ", 16\n"
#define A_PERM(r) "v_perm_b32 " r ", " r ", " r ", " r "\n"
#define A_LSHLADD(r) "v_lshl_add_u32 " r ", " r ", 2, " r "\n"
#define A_PKMUL(r) "v_pk_mul_lo_u16 " r ", " r ", " r "\n"
#define A_BFE(r) "v_bfe_u32 " r ", " r ", 8, 8\n"
#define A_CNDMASK_S(r) "v_cndmask_b32_e64 " r ", " r ", " r ", s[10:11]\n"
#define A_SUB(r) "v_sub_u32 " r ", " r ", " r "\n"
#define A_LSHL(r) "v_lshlrev_b32 " r ", 1, " r "\n"
#define A_MAXI(r) "v_max_i32 " r ", " r ", " r "\n"
#define A_AND(r) "v_and_b32 " r ", " r ", " r "\n"
#define A_ADDF(r) "v_add_f32 " r ", " r ", " r "\n"
#define A_PKMULF(r) "v_pk_mul_f32 " r ", " r ", " r "\n"
#define A_PKADDU16(r) "v_pk_add_u16 " r ", " r ", " r "\n"
#define A_MOV(r) "v_mov_b32 " r ", " r "\n"
#define A_MAXF(r) "v_max_f32 " r ", " r ", " r "\n"
#define A_MADU24(r) "v_mad_u32_u24 " r ", " r ", " r ", " r "\n"
#define A_MULU24(r) "v_mul_u32_u24 " r ", " r ", " r "\n"
#define A_CMPCND(r) "v_cmp_gt_i32 vcc, " r ", " r "\nv_cndmask_b32 " r ", " r ", " r ", vcc\n"
#define A_SAD(r) "v_sad_u32 " r ", " r ", " r ", " r "\n"

KERNEL(k_add, A_ADD)
KERNEL(k_add3, A_ADD3)
KERNEL(k_max3, A_MAX3)
KERNEL(k_mullo, A_MULLO)
KERNEL(k_mul24, A_MUL24)
KERNEL(k_mad24, A_MAD24)
KERNEL(k_sqrt, A_SQRT)
KERNEL(k_cvt, A_CVT)
KERNEL(k_fma, A_FMA)
KERNEL(k_mulf, A_MULF)
KERNEL(k_cndmask, A_CNDMASK)
KERNEL(k_cmp, A_CMP)
KERNEL(k_add_dpp_wshr, A_ADD_DPP_WSHR)
KERNEL(k_mov_dpp_wshr, A_MOV_DPP_WSHR)
KERNEL(k_mov_dpp_wshl, A_MOV_DPP_WSHL)
KERNEL(k_add_dpp_rshr, A_ADD_DPP_RSHR)
KERNEL(k_add_dpp_quad, A_ADD_DPP_QUAD)
KERNEL(k_bperm, A_BPERM)
KERNEL(k_alignbit, A_ALIGNBIT)
KERNEL(k_perm, A_PERM)
KERNEL(k_lshladd, A_LSHLADD)
KERNEL(k_pkmul, A_PKMUL)
KERNEL(k_bfe, A_BFE)
KERNEL(k_cndmask_s, A_CNDMASK_S)
KERNEL(k_sub, A_SUB)
KERNEL(k_lshl, A_LSHL)
KERNEL(k_maxi, A_MAXI)
KERNEL(k_and, A_AND)
KERNEL(k_addf, A_ADDF)
KERNEL(k_maxf, A_MAXF)
KERNEL(k_mov, A_MOV)
KERNEL(k_madu24, A_MADU24)
KERNEL(k_mulu24, A_MULU24)
KERNEL(k_pkaddu16, A_PKADDU16)
KERNEL(k_cmpcnd, A_CMPCND)
KERNEL(k_sad, A_SAD)

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int simds = p.multiProcessorCount * 4, waves_per_simd = 8;
    const int blocks = simds * waves_per_simd / 4;
    int *out;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    struct { const char *name; void (*fn)(int *, int); } tab[] = {
        {"v_add_u32", k_add}, {"v_add3_u32", k_add3}, {"v_max3_i32", k_max3}, {"v_mul_lo_u32", k_mullo}, {"v_mul_i32_i24", k_mul24},
        {"v_mad_i32_i24", k_mad24}, {"v_sqrt_f32", k_sqrt}, {"v_cvt_f32_i32", k_cvt}, {"v_fma_f32", k_fma}, {"v_mul_f32", k_mulf},
        {"v_cndmask_b32", k_cndmask}, {"v_cmp_gt_i32", k_cmp}, {"v_add_u32_dpp wave_shr", k_add_dpp_wshr},
        {"v_mov_b32_dpp wave_shr", k_mov_dpp_wshr}, {"v_mov_b32_dpp wave_shl", k_mov_dpp_wshl}, {"v_add_u32_dpp row_shr", k_add_dpp_rshr},
        {"v_add_u32_dpp quad_perm", k_add_dpp_quad}, {"ds_bpermute_b32", k_bperm}, {"v_alignbit_b32", k_alignbit}, {"v_perm_b32", k_perm},
        {"v_lshl_add_u32", k_lshladd}, {"v_pk_mul_lo_u16", k_pkmul}, {"v_bfe_u32", k_bfe}, {"v_cndmask_b32 (sgpr mask)", k_cndmask_s},
        {"v_sub_u32", k_sub}, {"v_lshlrev_b32", k_lshl}, {"v_max_i32", k_maxi}, {"v_and_b32", k_and}, {"v_add_f32", k_addf}, {"v_max_f32", k_maxf},
        {"v_mov_b32", k_mov}, {"v_mad_u32_u24", k_madu24}, {"v_mul_u32_u24", k_mulu24}, {"v_pk_add_u16", k_pkaddu16},
        {"v_cmp + v_cndmask (x2 instr)", k_cmpcnd}, {"v_sad_u32", k_sad}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double base = 0;
    printf("%d CUs, clockRate %d kHz, %d blocks of 256\n", p.multiProcessorCount, p.clockRate, blocks);
    for (auto &t : tab) {
        hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(t.fn, dim3(blocks), dim3(256), 0, 0, out, 2);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double per = ms * 1e6 / ((double)ITER * 32 * waves_per_simd);       // ns per wave-instruction per SIMD
        if (base == 0) base = per;
        printf("%-26s %7.3f ms  %6.3f ns/instr/SIMD  x%.2f\n", t.name, ms, per, per / base);
    }
    return 0;
}
