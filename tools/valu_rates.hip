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
        asm volatile("s_mov_b64 s[10:11], 0x5555" ::: "s10", "s11", "s12", "vcc");                                          \
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

#define A_NOT(r) "v_not_b32 " r ", " r "\n"
#define A_XOR(r) "v_xor_b32 " r ", " r ", " r "\n"
#define A_LSHR(r) "v_lshrrev_b32 " r ", 1, " r "\n"
#define A_ASHR(r) "v_ashrrev_i32 " r ", 1, " r "\n"
#define A_SUBF(r) "v_sub_f32 " r ", " r ", " r "\n"
#define A_FMAC(r) "v_fmac_f32 " r ", " r ", " r "\n"
#define A_CMPF(r) "v_cmp_gt_f32 vcc, " r ", " r "\n"
#define A_ADDC(r) "v_addc_co_u32 " r ", vcc, " r ", " r ", vcc\n"
#define A_MAX_DPP(r) "v_max_i32_dpp " r ", " r ", " r " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A_SUB_DPP(r) "v_sub_u32_dpp " r ", " r ", " r " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A_CVTUB(r) "v_cvt_f32_ubyte0 " r ", " r "\n"
#define A_CVTU(r) "v_cvt_f32_u32 " r ", " r "\n"
#define A_CVTRN(r) "v_cvt_i32_f32 " r ", " r "\n"
#define A_DOT2(r) "v_dot2_i32_i16 " r ", " r ", " r ", " r "\n"
#define A_DOT4(r) "v_dot4_u32_u8 " r ", " r ", " r ", " r "\n"
#define A_READLANE(r) "v_readlane_b32 s12, " r ", 5\n"
#define A_MBCNT(r) "v_mbcnt_lo_u32_b32 " r ", " r ", " r "\n"
#define A_LSHLOR(r) "v_lshl_or_b32 " r ", " r ", 3, " r "\n"
#define A_ANDOR(r) "v_and_or_b32 " r ", " r ", " r ", " r "\n"
#define A_MED3(r) "v_med3_i32 " r ", " r ", " r ", " r "\n"
#define A_MINF(r) "v_min_f32 " r ", " r ", " r "\n"
#define A_RSQ(r) "v_rsq_f32 " r ", " r "\n"
#define A_RCP(r) "v_rcp_f32 " r ", " r "\n"
#define A_MULSGPR(r) "v_mul_f32 " r ", s10, " r "\n"
#define A_FLOOR(r) "v_floor_f32 " r ", " r "\n"
#define A_PKFMA16(r) "v_pk_fma_f16 " r ", " r ", " r ", " r "\n"
#define A_PKMAD16(r) "v_pk_mad_u16 " r ", " r ", " r ", " r "\n"
#define A_PKSUB16(r) "v_pk_sub_i16 " r ", " r ", " r "\n"
#define A_ADDF64(r) "v_add_f64 v[20:21], v[20:21], v[22:23]\n"
#define A_CVTF64(r) "v_cvt_f64_i32 v[20:21], " r "\n"
#define A_MADI16(r) "v_mad_i32_i16 " r ", " r ", " r ", " r " op_sel:[0,1,0,0]\n"
#define A_CVTFLR(r) "v_cvt_flr_i32_f32 " r ", " r "\n"
#define A_ADD_DPP_ROR(r) "v_add_u32_dpp " r ", " r ", " r " row_ror:4 row_mask:0xf bank_mask:0xf\n"
#define A_ADD_DPP_RSHL(r) "v_add_u32_dpp " r ", " r ", " r " row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define A_DOT2C(r) "v_dot2c_i32_i16 " r ", " r ", " r "\n"
#define A_MUL24_SDWA(r) "v_mul_i32_i24_sdwa " r ", " r ", " r " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n"
#define A_ADD_SDWA(r) "v_add_u32_sdwa " r ", " r ", " r " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
#define A_DOT2_S(r) "v_dot2_i32_i16 " r ", " r ", " r ", s10\n"
#define A_ALIGNBYTE_V(r) "v_alignbyte_b32 " r ", " r ", " r ", %8\n"

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

KERNEL(k_not, A_NOT)
KERNEL(k_xor, A_XOR)
KERNEL(k_lshr, A_LSHR)
KERNEL(k_ashr, A_ASHR)
KERNEL(k_subf, A_SUBF)
KERNEL(k_fmac, A_FMAC)
KERNEL(k_cmpf, A_CMPF)
KERNEL(k_addc, A_ADDC)
KERNEL(k_max_dpp, A_MAX_DPP)
KERNEL(k_sub_dpp, A_SUB_DPP)
KERNEL(k_cvtub, A_CVTUB)
KERNEL(k_cvtu, A_CVTU)
KERNEL(k_cvtrn, A_CVTRN)
KERNEL(k_dot2, A_DOT2)
KERNEL(k_dot4, A_DOT4)
KERNEL(k_readlane, A_READLANE)
KERNEL(k_mbcnt, A_MBCNT)
KERNEL(k_lshlor, A_LSHLOR)
KERNEL(k_andor, A_ANDOR)
KERNEL(k_med3, A_MED3)
KERNEL(k_minf, A_MINF)
KERNEL(k_rsq, A_RSQ)
KERNEL(k_rcp, A_RCP)
KERNEL(k_mulsgpr, A_MULSGPR)
KERNEL(k_floor, A_FLOOR)
KERNEL(k_pkmad16, A_PKMAD16)
KERNEL(k_pksub16, A_PKSUB16)
KERNEL(k_madi16, A_MADI16)
KERNEL(k_cvtflr, A_CVTFLR)
KERNEL(k_add_dpp_ror, A_ADD_DPP_ROR)
KERNEL(k_add_dpp_rshl, A_ADD_DPP_RSHL)
KERNEL(k_dot2c, A_DOT2C)
KERNEL(k_mul24_sdwa, A_MUL24_SDWA)
KERNEL(k_add_sdwa, A_ADD_SDWA)
KERNEL(k_alignbyte_v, A_ALIGNBYTE_V)
KERNEL(k_dot2_s, A_DOT2_S)

// Which clock does the chip hold while every SIMD issues VALU instructions back to back?  One lane per workgroup reads the shader
// cycle counter (s_memtime) and the constant-rate counter (s_memrealtime) around ITER x 32 x 8 dependent-free v_add_u32 per wave.
__global__ __launch_bounds__(256) void k_clock_probe(long long *ticks, int seed)
{
    int r0 = seed + threadIdx.x, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13, r6 = r0 * 17, r7 = r0 * 19;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < ITER * 8; ++it) {
        asm volatile(A_ADD("%0") A_ADD("%1") A_ADD("%2") A_ADD("%3") A_ADD("%4") A_ADD("%5") A_ADD("%6") A_ADD("%7")
                     A_ADD("%0") A_ADD("%1") A_ADD("%2") A_ADD("%3") A_ADD("%4") A_ADD("%5") A_ADD("%6") A_ADD("%7")
                     A_ADD("%0") A_ADD("%1") A_ADD("%2") A_ADD("%3") A_ADD("%4") A_ADD("%5") A_ADD("%6") A_ADD("%7")
                     A_ADD("%0") A_ADD("%1") A_ADD("%2") A_ADD("%3") A_ADD("%4") A_ADD("%5") A_ADD("%6") A_ADD("%7")
                     : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7));
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = w1 - w0; }
    if (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 == 0x7fffffff) ticks[0] = 0;
}

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
        {"v_cmp + v_cndmask (x2 instr)", k_cmpcnd}, {"v_sad_u32", k_sad},
        {"v_not_b32", k_not}, {"v_xor_b32", k_xor}, {"v_lshrrev_b32", k_lshr}, {"v_ashrrev_i32", k_ashr}, {"v_sub_f32", k_subf}, {"v_fmac_f32", k_fmac},
        {"v_cmp_gt_f32", k_cmpf}, {"v_addc_co_u32", k_addc}, {"v_max_i32_dpp wave_shr", k_max_dpp}, {"v_sub_u32_dpp wave_shr", k_sub_dpp},
        {"v_cvt_f32_ubyte0", k_cvtub}, {"v_cvt_f32_u32", k_cvtu}, {"v_cvt_i32_f32", k_cvtrn}, {"v_dot2_i32_i16", k_dot2}, {"v_dot4_u32_u8", k_dot4},
        {"v_readlane_b32", k_readlane}, {"v_mbcnt_lo_u32_b32", k_mbcnt}, {"v_lshl_or_b32", k_lshlor}, {"v_and_or_b32", k_andor}, {"v_med3_i32", k_med3},
        {"v_min_f32", k_minf}, {"v_rsq_f32", k_rsq}, {"v_rcp_f32", k_rcp}, {"v_mul_f32 (sgpr src)", k_mulsgpr}, {"v_floor_f32", k_floor},
        {"v_pk_mad_u16", k_pkmad16}, {"v_pk_sub_i16", k_pksub16},
        {"v_mad_i32_i16 op_sel", k_madi16}, {"v_cvt_flr_i32_f32", k_cvtflr}, {"v_add_u32_dpp row_ror", k_add_dpp_ror},
        {"v_add_u32_dpp row_shl", k_add_dpp_rshl}, {"v_dot2c_i32_i16 (VOP2)", k_dot2c}, {"v_mul_i32_i24_sdwa", k_mul24_sdwa},
        {"v_add_u32_sdwa", k_add_sdwa}, {"v_alignbyte_b32 (vgpr shift)", k_alignbyte_v},
        {"v_dot2_i32_i16 (sgpr acc)", k_dot2_s}};
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double base = 0;
    printf("%d CUs, clockRate %d kHz, %d blocks of 256\n", p.multiProcessorCount, p.clockRate, blocks);
    {
        long long *ticks, *h = (long long *)malloc((size_t)blocks * 16);
        int wall_khz = 0;
        hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
        hipMalloc(&ticks, (size_t)blocks * 16);
        hipLaunchKernelGGL(k_clock_probe, dim3(blocks), dim3(256), 0, 0, ticks, 1);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_clock_probe, dim3(blocks), dim3(256), 0, 0, ticks, 2);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h, ticks, (size_t)blocks * 16, hipMemcpyDeviceToHost);
        double sc = 0, sw = 0;
        for (int b = 0; b < blocks; ++b) { sc += (double)h[2 * b]; sw += (double)h[2 * b + 1]; }
        const double instr_per_simd = (double)ITER * 8 * 32 * waves_per_simd;     // wave-instructions one SIMD issues in the probe
        const double wall_ms = sw / blocks / wall_khz, ns_each = wall_ms * 1e6 / instr_per_simd;
        printf("clock probe: kernel %.3f ms by events; per workgroup %.0f shader-counter ticks and %.0f ticks of the %d kHz wall counter (%.3f ms)\n"
               "  shader counter / wall time = %.1f MHz\n"
               "  v_add_u32: %.4f ns per wave-instruction per SIMD = %.2f clocks at 2400 MHz, or a sustained clock of %.0f MHz at 2 clocks each\n",
               ms, sc / blocks, sw / blocks, wall_khz, wall_ms, (sc / blocks) / wall_ms / 1e3, ns_each, ns_each * 2.4, 2.0 / ns_each * 1e3);
        hipFree(ticks); free(h);
    }
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
