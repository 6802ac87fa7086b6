// k_corners.hip — Shi-Tomasi corner detection (cv2.goodFeaturesToTrack semantics).  gfx950.
//
//   k_mineig_pair<BS,MASK>   (block 3/5/7, row pitch a dword multiple — the pipeline's kernel)
//       streaming response + 3x3 NMS + candidate keys: one wave per 128-column strip marching down the rows, two columns per
//       lane, no LDS tile, no response map in HBM.  Described at its definition.
//   k_mineig_stream<BS,MASK> (block 3/5/7/12, any width)
//       the same march with one column per lane and 64-column strips (byte loads on border strips).
//   k_mineig<BS,FUSED>       (any block size; the f32 map of ofk_mineig_response)
//       One 256-thread block computes the min-eigenvalue response on a 64 x OH region:
//         1. the gray tile (+ halo) is staged into LDS re-aligned with v_alignbyte (dword global loads);
//         2. one thread per (row, 8-column segment) reads its gray bytes as dwords, forms the Sobel column sums
//            s = r0+2r1+r2, t = r2-r0 (dx = s[j+2]-s[j], dy = t[j]+2t[j+1]+t[j+2]), the three products and their
//            horizontal box sums by a sliding window, all in registers (fully unrolled for BS = 3, 7, 12), and
//            writes eight int4 {xx,xy,yy} entries to LDS (XOR-swizzled: conflict-free ds_write_b128);
//         3. one thread per (column, row segment) slides the vertical box sum over ds_read_b128 entries and
//            evaluates lambda_min in f32 with a fixed operation order (-ffp-contract=off).
//       All window sums are integers, so the result does not depend on the summation order (bit-exact vs oracle).
//       FUSED = false: the f32 map is written to HBM (ofk_mineig_response).
//       FUSED = true : no map.  The region overlaps its neighbours by one pixel, the block applies the 3x3
//         local-max test itself and appends the survivors as 64-bit keys (~bits(value) << 32 | ~linear index: ascending key = value desc, index desc, OpenCV's greaterThanPtr order) with
//         ONE global atomic per block.  The quality threshold needs the image-wide maximum, which is not known yet;
//         the block prunes with the running maximum (atomicMax so far), a valid lower bound, and k_select_prep applies the
//         exact threshold.  HBM traffic: P read + O(candidates) written, instead of P + 4P + 4P.
//   k_nms    : the same threshold / 3x3 test / key list for a response map supplied by the caller
//              (ofk_select_corners), block-aggregated appends.
//   k_select_prep / k_select_pick / k_select_greedy : exact threshold + key histogram, first cut, then sort + greedy
//              minimum-distance pass in a 256-thread workgroup per image (described in front of them).
//              Result identical to a full sort (value desc, index desc) followed by the serial greedy pass.
#include "ofk_internal.h"

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

// ------------------------------------------------------------------------------------------------ min-eigenvalue
#define ME_OW 64                                              // region width computed by one block
#define ME_CAND_MAX 1920                                      // >= 62 x 30 interior pixels: the block buffer cannot overflow

__device__ __forceinline__ int hs_phys(int x) { return (x & ~7) | ((x & 7) ^ ((x >> 3) & 7)); }

template <int BS, bool FUSED>
__global__ __launch_bounds__(256) void k_mineig(const uint8_t *__restrict__ gray, size_t gray_stride, int h, int w,
                                                int bs_rt, int OH, float kd, float ko, float *__restrict__ eig,
                                                size_t eig_stride, unsigned int *__restrict__ maxbits,
                                                const uint8_t *__restrict__ mask, size_t mask_stride, double quality,
                                                unsigned long long *__restrict__ cand, int cand_cap,
                                                int *__restrict__ cand_count, int *__restrict__ flags)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int bs = BS ? BS : bs_rt;
    const int an = bs / 2;
    const int PH = OH + bs - 1;                               // product rows
    const int GH = PH + 2;                                    // gray rows
    const int GWp = 56 + 4 * ((9 + bs + 3) / 4);              // gray row pitch (bytes), covers every item's dword reads
    // LDS carve: gray | hs (PH x 64 int4) | eig (OH x 64 f32) | misc
    uint8_t *s_g = smem;
    int4 *s_h = reinterpret_cast<int4 *>(smem + ((GH * GWp + 15) & ~15));
    float *s_e = reinterpret_cast<float *>(s_h + PH * ME_OW);
    int *s_misc = reinterpret_cast<int *>(s_e + OH * ME_OW);   // [0..3] wave maxima, [4] thr bits, [5] cand count, [6] base
    unsigned long long *s_cand = reinterpret_cast<unsigned long long *>(s_h);   // aliases hs after the vertical pass

    const int b = blockIdx.z, tid = threadIdx.x;
    const uint8_t *img = gray + (size_t)b * gray_stride;
    const int IH = FUSED ? OH - 2 : OH;
    const int ex0 = FUSED ? (int)blockIdx.x * (ME_OW - 2) - 1 : (int)blockIdx.x * ME_OW;     // eig region origin
    const int ey0 = FUSED ? (int)blockIdx.y * IH - 1 : (int)blockIdx.y * IH;
    const int px0 = ex0 - an, py0 = ey0 - an;                 // product region origin
    const int gx0 = px0 - 1, gy0 = py0 - 1;                   // gray region origin
    const int GWd = GWp / 4;

    // ---- 1. stage gray
    const bool interior = gx0 >= 0 && gy0 >= 0 && gx0 + GWp + 4 <= w && gy0 + GH <= h && (w & 3) == 0;
    if (interior) {
        const int sh = gx0 & 3;
        const uint8_t *base = img + (size_t)gy0 * w + (gx0 & ~3);
        for (int i = tid; i < GH * GWd; i += 256) {
            const int r = i / GWd, c = i - r * GWd;
            const unsigned *p = reinterpret_cast<const unsigned *>(base + (size_t)r * w) + c;
            const unsigned lo = p[0], hi = p[1];
            unsigned v;
            switch (sh) {                                       // block-uniform
                case 0: v = lo; break;
                case 1: v = __builtin_amdgcn_alignbyte(hi, lo, 1); break;
                case 2: v = __builtin_amdgcn_alignbyte(hi, lo, 2); break;
                default: v = __builtin_amdgcn_alignbyte(hi, lo, 3); break;
            }
            reinterpret_cast<unsigned *>(s_g)[r * GWd + c] = v;
        }
    } else {
        for (int i = tid; i < GH * GWp; i += 256) {
            const int r = i / GWp, c = i - r * GWp;
            s_g[i] = img[(size_t)reflect101(gy0 + r, h) * w + reflect101(gx0 + c, w)];
        }
    }
    __syncthreads();

    // ---- 2. Sobel + products + horizontal box sums, item = (product row, 8-column segment)
    for (int it = tid; it < PH * 8; it += 256) {
        const int r = it >> 3, seg = it & 7;
        const int Y = py0 + r;
        const bool flipy = Y < 0 || Y >= h;                   // a mirrored row/column flips the sign of dy/dx: undo it in dx*dy
        int4 *orow = s_h + r * ME_OW + seg * 8;
        if (BS > 0) {
            constexpr int B_ = BS > 0 ? BS : 1;
            constexpr int NCOL = 8 + B_ + 1, ND = (NCOL + 3) / 4, NS = 8 + B_ - 1;
            unsigned a[3][ND];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const unsigned *g = reinterpret_cast<const unsigned *>(s_g + (r + q) * GWp + seg * 8);
#pragma unroll
                for (int i = 0; i < ND; ++i) a[q][i] = g[i];
            }
            int s[NCOL], t[NCOL];
#pragma unroll
            for (int j = 0; j < NCOL; ++j) {
                const int b0 = (a[0][j >> 2] >> (8 * (j & 3))) & 255, b1 = (a[1][j >> 2] >> (8 * (j & 3))) & 255,
                          b2 = (a[2][j >> 2] >> (8 * (j & 3))) & 255;
                s[j] = b0 + 2 * b1 + b2; t[j] = b2 - b0;
            }
            int pxx[NS], pxy[NS], pyy[NS];
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                const int dx = s[j + 2] - s[j], dy = t[j] + 2 * t[j + 1] + t[j + 2];
                const int X = px0 + seg * 8 + j;
                const bool flip = (X < 0 || X >= w) != flipy;
                pxx[j] = dx * dx; pyy[j] = dy * dy; pxy[j] = flip ? -(dx * dy) : dx * dy;
            }
            int sxx = 0, sxy = 0, syy = 0;
#pragma unroll
            for (int j = 0; j < B_; ++j) { sxx += pxx[j]; sxy += pxy[j]; syy += pyy[j]; }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (k) { sxx += pxx[k + B_ - 1] - pxx[k - 1]; sxy += pxy[k + B_ - 1] - pxy[k - 1]; syy += pyy[k + B_ - 1] - pyy[k - 1]; }
                orow[k ^ seg] = make_int4(sxx, sxy, syy, 0);
            }
        } else {
            // generic block size: byte reads, trailing products recomputed
            const uint8_t *g0 = s_g + r * GWp + seg * 8, *g1 = g0 + GWp, *g2 = g1 + GWp;
            int sxx = 0, sxy = 0, syy = 0;
            for (int j = 0; j < 8 + bs - 1; ++j) {
                {
                    const int dx = (g0[j + 2] - g0[j]) + 2 * (g1[j + 2] - g1[j]) + (g2[j + 2] - g2[j]);
                    const int dy = (g2[j] - g0[j]) + 2 * (g2[j + 1] - g0[j + 1]) + (g2[j + 2] - g0[j + 2]);
                    const int X = px0 + seg * 8 + j;
                    const bool flip = (X < 0 || X >= w) != flipy;
                    sxx += dx * dx; syy += dy * dy; sxy += flip ? -(dx * dy) : dx * dy;
                }
                if (j >= bs) {
                    const int jo = j - bs;
                    const int dx = (g0[jo + 2] - g0[jo]) + 2 * (g1[jo + 2] - g1[jo]) + (g2[jo + 2] - g2[jo]);
                    const int dy = (g2[jo] - g0[jo]) + 2 * (g2[jo + 1] - g0[jo + 1]) + (g2[jo + 2] - g0[jo + 2]);
                    const int X = px0 + seg * 8 + jo;
                    const bool flip = (X < 0 || X >= w) != flipy;
                    sxx -= dx * dx; syy -= dy * dy; sxy -= flip ? -(dx * dy) : dx * dy;
                }
                if (j >= bs - 1) orow[(j - bs + 1) ^ seg] = make_int4(sxx, sxy, syy, 0);
            }
        }
    }
    __syncthreads();

    // ---- 3. vertical box sums + lambda_min; thread = (column, row segment)
    float lmax = 0.f;
    {
        const int x = tid & 63, q = tid >> 6;
        const int len = (OH + 3) >> 2;
        const int r0 = q * len;
        const int n = min(len, OH - r0);
        const int xp = hs_phys(x);
        if (n > 0) {
            int sxx = 0, sxy = 0, syy = 0;
            for (int j = 0; j < bs; ++j) { const int4 v = s_h[(r0 + j) * ME_OW + xp]; sxx += v.x; sxy += v.y; syy += v.z; }
            const uint8_t *mk = mask ? mask + (size_t)b * mask_stride : nullptr;
            const int gx = ex0 + x;
            for (int k = 0; k < n; ++k) {
                if (k) {
                    const int4 vo = s_h[(r0 + k - 1) * ME_OW + xp], vn = s_h[(r0 + k - 1 + bs) * ME_OW + xp];
                    sxx += vn.x - vo.x; sxy += vn.y - vo.y; syy += vn.z - vo.z;
                }
                const float a = (float)sxx * kd, bb = (float)sxy * ko, c = (float)syy * kd;
                const float amc = a - c;
                const float v = (a + c) - sqrtf(amc * amc + bb * bb);
                const int gy = ey0 + r0 + k;
                const bool inimg = gx >= 0 && gx < w && gy >= 0 && gy < h;
                if (FUSED) {
                    s_e[(r0 + k) * ME_OW + x] = v;
                    const bool own = x >= 1 && x <= ME_OW - 2 && r0 + k >= 1 && r0 + k <= OH - 2;     // interior: counted once
                    if (inimg && own && (!mk || mk[(size_t)gy * w + gx])) lmax = fmaxf(lmax, v);
                } else if (inimg) {
                    eig[(size_t)b * eig_stride + (size_t)gy * w + gx] = v;
                    if (!mk || mk[(size_t)gy * w + gx]) lmax = fmaxf(lmax, v);
                }
            }
        }
    }
    if (!maxbits) return;                                      // map only (uniform)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
    if ((tid & 63) == 0) s_misc[tid >> 6] = __float_as_int(lmax);
    if (tid == 0) s_misc[5] = 0;
    __syncthreads();
    if (tid == 0) {
        const float m = fmaxf(fmaxf(__int_as_float(s_misc[0]), __int_as_float(s_misc[1])),
                              fmaxf(__int_as_float(s_misc[2]), __int_as_float(s_misc[3])));
        unsigned cur = __float_as_uint(m);                    // m >= 0: bit patterns order like the values
        const unsigned old = m > 0.f ? atomicMax(maxbits + b * OFK_MAX_STRIDE, cur)
                                     : __hip_atomic_load(maxbits + b * OFK_MAX_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old > cur) cur = old;
        s_misc[4] = __float_as_int((float)((double)__uint_as_float(cur) * quality));   // running threshold <= final threshold
    }
    if (!FUSED) return;
    __syncthreads();

    // ---- 4. 3x3 local maxima above the running threshold -> keys
    {
        const float thr = __int_as_float(s_misc[4]);
        const uint8_t *mk = mask ? mask + (size_t)b * mask_stride : nullptr;
        const int x = tid & 63;
        for (int y = 1 + (tid >> 6); y <= OH - 2; y += 4) {
            if (x < 1 || x > ME_OW - 2) continue;
            const int gx = ex0 + x, gy = ey0 + y;
            if (gx < 1 || gx >= w - 1 || gy < 1 || gy >= h - 1) continue;
            const float *c = s_e + y * ME_OW + x;
            const float v = c[0];
            if (!(v > thr) || !(v > 0.f)) continue;
            if (mk && !mk[(size_t)gy * w + gx]) continue;
            const float m = fmaxf(fmaxf(fmaxf(c[-ME_OW - 1], c[-ME_OW]), fmaxf(c[-ME_OW + 1], c[-1])),
                                  fmaxf(fmaxf(c[1], c[ME_OW - 1]), fmaxf(c[ME_OW], c[ME_OW + 1])));
            if (m > v) continue;
            const int slot = atomicAdd(&s_misc[5], 1);
            if (slot < ME_CAND_MAX) s_cand[slot] = ((unsigned long long)(~__float_as_uint(v)) << 32) | (unsigned)~(unsigned)(gy * w + gx);
        }
    }
    __syncthreads();
    const int nc = min(s_misc[5], ME_CAND_MAX);
    if (nc == 0) return;
    if (tid == 0) s_misc[6] = atomicAdd(cand_count + b * OFK_CNT_STRIDE, nc);
    __syncthreads();
    const int base = s_misc[6];
    for (int i = tid; i < nc; i += 256) {
        if (base + i < cand_cap) cand[(size_t)b * cand_cap + base + i] = s_cand[i];
        else if (i == 0 || base + i == cand_cap) atomicOr(flags, 1);
    }
}

static int mineig_oh(int bs) { int oh = 33 - bs; if (oh < 8) oh = 8; if (oh > 32) oh = 32; return oh; }

template <int BS, bool FUSED>
static int launch_mineig_t(hipStream_t s, const uint8_t *gray, size_t gray_stride, int h, int w, int block, float *eig,
                           size_t eig_stride, unsigned int *maxbits, const uint8_t *mask, size_t mask_stride,
                           double quality, unsigned long long *cand, int cand_cap, int *cand_count, int *flags, int batch)
{
    const int OH = mineig_oh(block), PH = OH + block - 1, GH = PH + 2, GWp = 56 + 4 * ((9 + block + 3) / 4);
    size_t lds = ((size_t)(GH * GWp + 15) & ~(size_t)15) + (size_t)PH * ME_OW * 16 + (size_t)OH * ME_OW * 4 + 32;
    if (lds > 160 * 1024) return -1;
    static size_t attr_lds = 64 * 1024;                       // dynamic LDS above 64 KiB has to be opted into
    if (lds > attr_lds) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_mineig<BS, FUSED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        attr_lds = lds;
    }
    const double scale = 1.0 / (4.0 * block * 255.0);
    const float kd = (float)(0.5 * scale * scale), ko = (float)(scale * scale);
    const int IH = FUSED ? OH - 2 : OH, IW = FUSED ? ME_OW - 2 : ME_OW;
    dim3 grid((w + IW - 1) / IW, (h + IH - 1) / IH, batch);
    hipLaunchKernelGGL((k_mineig<BS, FUSED>), grid, dim3(256), lds, s, gray, gray_stride, h, w, block, OH, kd, ko, eig, eig_stride,
                       maxbits, mask, mask_stride, quality, cand, cand_cap, cand_count, flags);
    return 0;
}

// Writes the f32 response map (maxbits optional).
int ofk_launch_mineig(hipStream_t s, const uint8_t *gray, size_t gray_stride, int h, int w, int block, float *eig,
                      size_t eig_stride, unsigned int *maxbits, const uint8_t *mask, size_t mask_stride, int batch)
{
#define ME_ARGS s, gray, gray_stride, h, w, block, eig, eig_stride, maxbits, mask, mask_stride, 0.0, nullptr, 0, nullptr, nullptr, batch
    switch (block) {
        case 3: return launch_mineig_t<3, false>(ME_ARGS);
        case 7: return launch_mineig_t<7, false>(ME_ARGS);
        case 12: return launch_mineig_t<12, false>(ME_ARGS);
        default: return launch_mineig_t<0, false>(ME_ARGS);
    }
#undef ME_ARGS
}

// ------------------------------------------------------------------------------------------------ streaming response + NMS
// k_mineig_stream<BS>: one WAVE per column strip, marching down the rows — no LDS tile, no barriers.
//   lane = gray/product column.  Per row: 1 byte load per lane (prefetched a block of BS rows ahead), Sobel column
//   sums in registers, x-neighbours through DPP wave shifts, the horizontal box sum as a Horner chain of BS-1
//   v_add_u32_dpp (acc = shr1(acc) + p), the vertical box sum as a register ring of BS rows, lambda_min in f32,
//   3x3 NMS on the last three response rows, candidate keys buffered per wave in LDS and flushed 64 at a time with one
//   global atomic.  A strip of 64 lanes yields 61-BS candidate columns (BS=7: 54, i.e. 84 % useful lanes; the LDS-tile
//   kernel's halo redundancy is 2.4x).  Integer sums are identical to the oracle's, the f32 formula is the same.
#define DPP_SHR1(v) __builtin_amdgcn_update_dpp(0, (v), 0x138, 0xF, 0xF, true)      /* lane l <- lane l-1 (column x-1) */
#define DPP_SHL1(v) __builtin_amdgcn_update_dpp(0, (v), 0x130, 0xF, 0xF, true)      /* lane l <- lane l+1 (column x+1) */

__device__ __forceinline__ float wave_max_f32(float v)
{
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true)));
    v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, true)));
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16)),
                c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return fmaxf(fmaxf(a, b), fmaxf(c, d));
}

// Correctly rounded sqrt(x) for the x the response meets: x = (a-c)^2 + b^2 with a, b, c integer multiples of ~1e-8, i.e. zero or
// >= 1e-16.  v_sqrt_f32 is NOT correctly rounded on gfx950 (288 M of the 2.1 G non-negative floats come out one ulp low, 0.1 M one ulp
// high: tools/sqrt_exhaustive.hip), and the oracle's sqrtf is.  Round 4: reciprocal square root + ONE Newton correction with fused
// residual - y = v_rsq(x), s = x y, r = fma(-s, s, x), s' = fma(r, y / 2, s) - is correctly rounded for EVERY float from 2^-100 to
// 2^127 (exhaustive on the device against f32(sqrt(f64)), 0 mismatches: tools/sqrt_exhaustive.hip; below 2^-100 the residual goes denormal), in 5 instructions where the
// v_sqrt-based fix-up of rounds 1-3 (two neighbours, two fused residuals, two clamps, one add) took 8.  x = 0 (flat regions; v_rsq
// gives inf) is clamped to 2^-120: the root, 2^-60, is far below half an ulp of the smallest a + c > 0 there is (2 kd ~ 2e-8), so
// (a + c) - root rounds to a + c exactly as with root 0; where a + c = 0 as well the response becomes -2^-60 instead of +0 - a value
// below every threshold (>= +0) and below every candidate, which no maximum, key or count can tell from +0.
__device__ __forceinline__ float sqrt_rn_normal(float x)
{
    const float xc = fmaxf(x, 0x1p-120f);
    const float y = __builtin_amdgcn_rsqf(xc);
    const float s = xc * y;
    const float r = __builtin_fmaf(-s, s, xc);
    return __builtin_fmaf(r, 0.5f * y, s);
}

// Horizontal box sum of BS consecutive lanes (x-BS+1 .. x).  A Horner chain of wave shifts costs BS-1 VALU slots, and the
// kernel is bound by VALU issue while the LDS pipe idles — so beyond BS = 3 only the pair sum uses a DPP add and the pairs
// (quads for BS = 12) are gathered with ds_bpermute, which runs on the LDS pipe: 3 VALU + 3 LDS instead of 6 (11) VALU.
// Lanes below BS-1 receive garbage (bpermute wraps around), exactly the lanes that are never used.
template <int BS>
__device__ __forceinline__ int box_row(int a, int ad2, int ad4, int adl)
{
    if constexpr (BS == 5 || BS == 7 || BS == 12) {
        const int p = DPP_SHR1(a) + a;                                           // x-1 .. x
        if constexpr (BS == 5) return p + __builtin_amdgcn_ds_bpermute(ad2, p) + __builtin_amdgcn_ds_bpermute(ad4, a);
        if constexpr (BS == 7) return p + __builtin_amdgcn_ds_bpermute(ad2, p) + (__builtin_amdgcn_ds_bpermute(ad4, p) + __builtin_amdgcn_ds_bpermute(adl, a));
        if constexpr (BS == 12) {
            const int q = p + __builtin_amdgcn_ds_bpermute(ad2, p);             // x-3 .. x
            return q + __builtin_amdgcn_ds_bpermute(ad4, q) + __builtin_amdgcn_ds_bpermute(adl, q);
        }
    } else {
        int hsum = a;
#pragma unroll
        for (int k = 1; k < BS; ++k) hsum = DPP_SHR1(hsum) + a;
        return hsum;
    }
}

// One block of BS rows of k_mineig_stream (expanded twice inside the kernel: IN = true for interior blocks).
//  * dx is written as an ADD of the negated source: ROCm 7.2 folds `a - dpp(b)` into v_subrev_u32_dpp, which on gfx950
//    returned shr(shl(s)) - s here (measured); v_add_u32_dpp is exact.
//  * |dx|, |dy| <= 4080, so the products are exact in v_mul_i32_i24 (full rate; v_mul_lo_u32 is a quarter-rate
//    instruction and three of them cost as much as twelve adds).  The empty asm keeps instruction selection from folding
//    a product into v_mad_i32_i24 in front of the box sum, which would undo the v_add_u32_dpp pair sum.
//  * A mirrored product row (column) has its y (x) derivative negated, which only flips the sign of dx*dy (border blocks).
//  * Maxima run on the BIT PATTERNS as signed integers: the image maximum and every candidate are positive, and for a
//    positive e1 "some neighbour is greater" is the same statement on floats and on their bits (negative floats are
//    negative integers; there are no NaNs).  The row maxima hm include the centre, so max(hm0, hm1, hm2) == e1 iff none of
//    the eight neighbours is greater — two wave shifts and two v_max3 per row.  thri >= 0, so e1i > thri implies e1 > 0.
//  * Interior blocks take the running maximum over every lane; the columns that do not count are dropped when the maximum
//    is published.  yo = response row completed by the step (garbage while r <= BS), yn = row whose 3x3 neighbourhood
//    is complete.
// Keys inside a strip's SEGMENT are stored raw — (response bits << 32) | linear index — and complemented by k_select_prep when it
// compacts them into the flat list (ascending key = value desc, index desc): two v_not per candidate row and slot leave the
// response kernels' row loop, which is bound by VALU issue, for one 64-bit NOT per key in a kernel that waits for memory.
#define OFK_SEG_KEY(bits, idx) (((unsigned long long)(unsigned)(bits) << 32) | (unsigned)(idx))
#define OFK_SEG_KEY_DECODE(k) (~(k))
// A lane holds candidates in both of its columns only on a plateau (two neighbours that are both 3x3 maxima are equal): nearly every row
// has candidates in one column per lane at most, and then ONE store block serves both columns - one rank, one address, the value and
// the column picked by the column's mask - instead of two blocks of which almost every row executed both (round 4: -0.7 % alone,
// -1.2 % on top of the cheaper square root).  The order of keys inside a segment is irrelevant (the selection sorts).
#define OFK_PAIR_KEY_STORE()                                                                                           \
        if ((bale & balo) == 0) {                                                                                      \
            const unsigned long long bal_ = bale | balo;                                                               \
            if (ise | iso) buf[cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal_ >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal_, 0u))] = \
                OFK_SEG_KEY(ise ? e1e : e1o, yn * w + (ise ? xo_e : xo_o));                                            \
        } else {                                                                                                       \
            if (ise) buf[cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bale >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bale, 0u))] =   \
                OFK_SEG_KEY(e1e, yn * w + xo_e);                                                                       \
            if (iso) buf[cnt + ne + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(balo >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)balo, 0u))] = \
                OFK_SEG_KEY(e1o, yn * w + xo_o);                                                                       \
        }
#define OFK_EIG_ROWS(IN)                                                                                               \
    _Pragma("unroll") for (int i = 0; i < BS; ++i) {                                                                   \
        const int r = base + i;                                                                                        \
        const int g2 = curg[i];                                                                                        \
        const int s = g0 + 2 * g1 + g2, t = g2 - g0;                                                                   \
        const int ns = -s;                                                                                             \
        const int dx = DPP_SHL1(s) + DPP_SHR1(ns);                                                                     \
        const int dy = (DPP_SHR1(t) + t) + (DPP_SHL1(t) + t);                                                          \
        int pxx = __mul24(dx, dx), pyy = __mul24(dy, dy), pxy = __mul24(dx, dy);                                       \
        asm("" : "+v"(pxx), "+v"(pxy), "+v"(pyy));                                                                     \
        if (!(IN)) {                                                                                                   \
            const int Y = Yp0 + r - 2;                                                                                 \
            pxy = (((Y < 0) | (Y >= h)) != flipx) ? -pxy : pxy;                                                        \
        }                                                                                                              \
        const int hxx = box_row<BS>(pxx, ad2, ad4, adl), hxy = box_row<BS>(pxy, ad2, ad4, adl),                        \
                  hyy = box_row<BS>(pyy, ad2, ad4, adl);                                                               \
        vxx += hxx - ringxx[i]; vxy += hxy - ringxy[i]; vyy += hyy - ringyy[i];                                        \
        ringxx[i] = hxx; ringxy[i] = hxy; ringyy[i] = hyy;                                                             \
        const int yo = ya - 2 + r - BS;                                                                                \
        const float a = (float)vxx * kd, bb = (float)vxy * ko, c = (float)vyy * kd;                                    \
        const float amc = a - c;                                                                                       \
        const int e2i = __float_as_int((a + c) - sqrt_rn_normal(amc * amc + bb * bb));                                 \
        if ((IN) && !MASK) {                                                                                           \
            lmaxi = max(lmaxi, e2i);                                                                                   \
        } else {                                                                                                       \
            bool cm = own_col & (yo >= ya) & (yo < yb);                                                                \
            if (MASK) cm = cm & (mk[(size_t)min(max(yo, 0), h - 1) * w + xoc] != 0);                                   \
            lmaxi = max(lmaxi, cm ? e2i : 0);                                                                          \
        }                                                                                                              \
        const int hm2 = max(max(DPP_SHR1(e2i), e2i), DPP_SHL1(e2i));                                                   \
        const int yn = yo - 1;                                                                                         \
        const int m = max(max(hm0, hm1), hm2);                                                                         \
        bool is = e1i >= max(m, thr1);                                                                                 \
        if (!(IN)) is = is & ((yn >= ya) & (yn < yb) & (yn >= 1) & (yn < h - 1));                                      \
        if (MASK) is = is & (mk[(size_t)min(max(yn, 0), h - 1) * w + xoc] != 0);                                       \
        const unsigned long long bal = __builtin_amdgcn_ballot_w64(is);                                                \
        if (is) buf[cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u))] =   \
            OFK_SEG_KEY(e1i, yn * w + xo);                                                                              \
        cnt += __popcll(bal);                                                                                          \
        e1i = e2i; hm0 = hm1; hm1 = hm2;                                                                               \
        g0 = g1; g1 = g2;                                                                                              \
    }

template <int BS, bool MASK>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void k_mineig_stream(const uint8_t *__restrict__ gray, size_t gray_stride, int h, int w,
                                                       int rows_per_strip, float kd, float ko,
                                                       unsigned int *__restrict__ maxbits, const uint8_t *__restrict__ mask,
                                                       size_t mask_stride, double quality,
                                                       unsigned long long *__restrict__ seg, int seg_cap,
                                                       int *__restrict__ seg_count, int *__restrict__ flags)
{
    constexpr int AN = BS / 2, SW = 61 - BS;
    constexpr int NBUF = 64 + BS * SW;                          // keys a wave can hold between two flush points
    __shared__ unsigned long long s_buf[NBUF + 64];             // + 64: a flush reads one whole 64-key chunk past the count
    const int lane = threadIdx.x;                               // one wave per workgroup (see k_mineig_pair): no half-empty workgroups
    const int sx = blockIdx.x;                                  // wave-uniform: strip geometry stays in SGPRs
    if (sx * SW >= w) return;                                   // whole wave
    const int b = blockIdx.z;
    const int ya = blockIdx.y * rows_per_strip, yb = min(h, ya + rows_per_strip);
    const uint8_t *img = gray + (size_t)b * gray_stride;
    const uint8_t *mk = MASK ? mask + (size_t)b * mask_stride : nullptr;
    unsigned long long *buf = s_buf;                            // private to this wave: LDS executes a wave's accesses in order

    const int gx = sx * SW - 2 - AN + lane;                     // gray / product column of this lane
    const int gxr = reflect101(gx, w);
    const bool flipx = gx < 0 || gx >= w;
    const bool edge_strip = sx * SW - 2 - AN < 0 || sx * SW - 2 - AN + 63 >= w;      // some lane is mirrored (wave-uniform)
    const int xo = gx - (BS - 1) + AN;                          // column of this lane's box sums / response
    const bool lane_ok = lane >= BS + 1 && lane <= 61;
    const bool own_col = lane_ok && xo >= 0 && xo < w;          // counted for the maximum (every pixel exactly once)
    const bool nms_col = lane_ok && xo >= 1 && xo < w - 1;      // may be a corner
    const int xoc = min(max(xo, 0), w - 1);                     // clamped: safe address for predicated mask reads

    const int Yp0 = ya - 1 - AN;                                // first product row
    const int nsteps = (yb - ya) + BS + 3;                      // gray rows consumed (padded up to a multiple of BS below)
    // Every strip owns a fixed segment of the key buffer: keys are appended with plain stores, the count is written
    // once at the end.  (A returning global atomic per flush stalled the whole CU's memory pipeline: 0.26 -> 1.1 ms.)
    const int nstrips = (w + SW - 1) / SW;
    const int nseg = nstrips * (int)gridDim.y, segid = (int)blockIdx.y * nstrips + sx;
    unsigned long long *myseg = seg + ((size_t)b * nseg + segid) * seg_cap;
    int written = 0;                                            // keys already in the segment (uniform)
    int cnt = 0;                                                // keys buffered in LDS by this wave (uniform)
    int lmaxi = 0;                                              // bits of the running maximum (>= +0.0f)
    float published = 0.f;
    unsigned mb_seen = __hip_atomic_load(maxbits + b * OFK_MAX_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // running maximum, refreshed per block
    float thr = (float)((double)__uint_as_float(mb_seen) * quality);

    // Row r of the march reads gray row reflect101(Yp0 - 1 + r).  The overhang is at most BS + 3 < h (checked by the
    // host), so one branch-free reflection suffices; rows past the strip are clamped (their results are never used).
    // Interior strips fetch DWORDS: one wave-instruction brings 3 rows x 17 dwords (68 B cover the strip's 64 columns at
    // any alignment) and ds_bpermute hands every lane its byte — byte-wide global loads cost a TA cycle per lane quad
    // and were the bottleneck (measured: 66 % of the wave time parked on vmcnt).  Edge strips (mirrored columns) keep
    // the per-lane byte loads.
    const unsigned ugx = (unsigned)gxr;
    const int gx0 = sx * SW - 2 - AN;                           // column of lane 0
    const bool fast = gx0 >= 0 && (gx0 & ~3) + 68 <= w && (w & 3) == 0;       // wave-uniform
    constexpr int NL = (BS + 2) / 3;                            // dword loads per block of BS rows
    const int sh = gx0 & 3;
    const int lrow = min(lane / 17, 2), lk = lane - 17 * (lane / 17);          // loader role of this lane (lanes >= 51: dummy)
    const unsigned ldoff = (unsigned)((gx0 & ~3) + 4 * min(lk, 16));
    const int srcsel = ((lane + sh) >> 2) * 4, bytesh = ((lane + sh) & 3) * 8;  // consumer role: source dword lane, byte
    const int ad2 = ((lane - 2) & 63) * 4, ad4 = ((lane - 4) & 63) * 4, adl = ((lane - (BS == 12 ? 8 : 6)) & 63) * 4;   // box_row gathers
    auto row_of = [&](int r) -> int {
        int gy = Yp0 - 1 + min(r, nsteps - 1);
        gy = gy < 0 ? -gy : gy;
        return gy >= h ? 2 * (h - 1) - gy : gy;
    };
#define OFK_LOAD_BLOCK(R0, OUT)                                                                                        \
    do {                                                                                                               \
        if (fast) {                                                                                                    \
            _Pragma("unroll") for (int q_ = 0; q_ < NL; ++q_) {                                                        \
                const int a_ = row_of((R0) + 3 * q_), b_ = row_of((R0) + 3 * q_ + 1), c_ = row_of((R0) + 3 * q_ + 2); \
                const int gy_ = lrow == 0 ? a_ : (lrow == 1 ? b_ : c_);                                                \
                OUT[q_] = (int)*reinterpret_cast<const unsigned *>(img + (size_t)gy_ * w + ldoff);                     \
            }                                                                                                          \
        } else {                                                                                                       \
            _Pragma("unroll") for (int i_ = 0; i_ < BS; ++i_) OUT[i_] = (img + (size_t)row_of((R0) + i_) * w)[ugx];    \
        }                                                                                                              \
    } while (0)

    // Warm-up needs no special case: the first two (incomplete) Sobel rows enter the vertical ring as garbage and
    // leave it again (exact integer subtraction) before the first response row that is used.
    int g0 = 0, g1 = 0;
    int ringxx[BS], ringxy[BS], ringyy[BS];
#pragma unroll
    for (int i = 0; i < BS; ++i) { ringxx[i] = 0; ringxy[i] = 0; ringyy[i] = 0; }
    int vxx = 0, vxy = 0, vyy = 0;
    int e1i = 0, hm0 = 0, hm1 = 0;                              // bit patterns of f32 responses (see OFK_EIG_ROWS)
    int nextg[BS];
#pragma unroll
    for (int i = 0; i < BS; ++i) nextg[i] = 0;
    OFK_LOAD_BLOCK(0, nextg);

    for (int base = 0; base < nsteps; base += BS) {
        int curg[BS];
        if (fast) {
#pragma unroll
            for (int i = 0; i < BS; ++i) {
                const int v = __builtin_amdgcn_ds_bpermute(srcsel + 68 * (i % 3), nextg[i / 3]);
                curg[i] = (int)(((unsigned)v >> bytesh) & 255u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < BS; ++i) curg[i] = nextg[i];
        }
        // ---- the only branchy part of the loop sits here: move full 64-key chunks from LDS to this strip's segment
        //      (plain stores) and publish the strip maximum (non-returning atomic)
        {
            unsigned cur_seen = (unsigned)__builtin_amdgcn_readfirstlane((int)mb_seen);
            if (__float_as_uint(published) > cur_seen) cur_seen = __float_as_uint(published);
            thr = (float)((double)__uint_as_float(cur_seen) * quality);
        }
        if (cnt >= 64) {
            const int nchunk = cnt >> 6;
            for (int q = 0; q < nchunk; ++q) {
                const unsigned long long key = buf[q * 64 + lane];
                if (written + q * 64 + lane < seg_cap) myseg[written + q * 64 + lane] = key;
            }
            written += nchunk * 64;
            const unsigned long long rest = buf[nchunk * 64 + lane];   // < 64 keys remain
            __builtin_amdgcn_wave_barrier();
            buf[lane] = rest;
            cnt -= nchunk * 64;
            const float mw = wave_max_f32(own_col ? __int_as_float(lmaxi) : 0.f);
            if (mw > published) {                               // publish only when this strip raised its maximum
                if (lane == 0) (void)atomicMax(maxbits + b * OFK_MAX_STRIDE, __float_as_uint(mw));
                published = mw;
            }
            mb_seen = __hip_atomic_load(maxbits + b * OFK_MAX_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // consumed at the next block
        }
        OFK_LOAD_BLOCK(base + BS, nextg);                                      // prefetch the next block of rows
        // ---- BS rows of straight-line code, in two versions: most blocks are INTERIOR (no mirrored product row or
        //      column, every response row owned by this strip, every NMS row valid) and skip the border bookkeeping
        const int r1 = base + BS - 1;
        const bool interior = !edge_strip && Yp0 + base - 2 >= 0 && Yp0 + r1 - 2 < h && base >= BS + 3 && ya - 3 + base - BS >= 1 &&
                              ya - 2 + r1 - BS < yb && ya - 3 + r1 - BS < h - 1;
        // candidate <=> e1 is a strict-or-equal 3x3 maximum (m == e1i, and m >= e1i always) above the threshold, in a
        // column that may hold a corner: ONE compare, e1i >= max(m, thr1), with thr1 = threshold bits + 1 (never for the
        // other columns).  A single compare feeds the ballot directly; an AND of compares costs two extra VALU slots.
        const int thr1 = nms_col ? __float_as_int(fmaxf(thr, 0.f)) + 1 : 0x7fffffff;
        if (interior) { OFK_EIG_ROWS(true) } else { OFK_EIG_ROWS(false) }
    }
    // ---- tail: remaining keys, the segment's count and the strip maximum
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < cnt; i += 64)
        if (written + i < seg_cap) myseg[written + i] = buf[i];
    if (lane == 0) {
        seg_count[(size_t)b * nseg + segid] = min(written + cnt, seg_cap);
        if (written + cnt > seg_cap) (void)atomicOr(flags, 1);
    }
    const float mw = wave_max_f32(own_col ? __int_as_float(lmaxi) : 0.f);
    if (lane == 0 && mw > published) (void)atomicMax(maxbits + b * OFK_MAX_STRIDE, __float_as_uint(mw));
}

// ------------------------------------------------------------------------------------------------ streaming response, 2 columns / lane
// k_mineig_pair<BS>: the same march as k_mineig_stream with every lane owning TWO adjacent columns (even slot 2l, odd slot
// 2l+1 of a 128-column strip).  The kernel is bound by VALU issue (SQ counters: ~1 VALU instruction per 4 clocks per SIMD,
// 98 % of the time), so what counts is instructions per pixel:
//   * the halo (BS + 3 columns) is paid once per 128 columns instead of once per 64: 116 of 128 columns are outputs for
//     BS = 7 (54 of 64 before);
//   * one of the two x-neighbours of a column lives in the same lane, so Sobel and the 3x3 maximum need half the wave
//     shifts per pixel;
//   * the box sums work on pair sums q = a_e + a_o: for BS = 2K+1, s_o(l) = a_o(l-K) + q(l-K+1..l) and
//     s_e(l) = a_e(l) + q(l-K..l-1) share their gathers (ds_bpermute on the otherwise idle LDS pipe);
//   * gray rows arrive as aligned dwords, two rows per wave-load, for EVERY strip: the strip origin is a multiple of 4
//     and mirrored border columns are patched from the lanes that hold their sources (two gathers per row, border strips
//     only) — no byte-load path.
// Integer sums and the f32 formula are those of k_mineig_stream (and of the oracle), so results are bit-identical.
template <int K>
__device__ __forceinline__ void box_pair_q(int ae, int ao, int q, int ad2, int ad3, int &he, int &ho);
template <int K>
__device__ __forceinline__ void box_pair(int ae, int ao, int ad2, int ad3, int &he, int &ho) { box_pair_q<K>(ae, ao, ae + ao, ad2, ad3, he, ho); }
template <int K>
__device__ __forceinline__ void box_pair_q(int ae, int ao, int q, int ad2, int ad3, int &he, int &ho)
{
    if constexpr (K == 1) {
        ho = DPP_SHR1(ao) + q;
        he = ae + DPP_SHR1(q);
    } else {
        const int P = DPP_SHR1(q) + q;                                           // q(l-1) + q(l)
        const int g2q = __builtin_amdgcn_ds_bpermute(ad2, q);                    // q(l-2)
        if constexpr (K == 2) {
            ho = P + __builtin_amdgcn_ds_bpermute(ad2, ao);
            he = (P - ao) + g2q;
        } else {
            ho = P + g2q + __builtin_amdgcn_ds_bpermute(ad3, ao);
            he = (P - ao) + g2q + __builtin_amdgcn_ds_bpermute(ad3, q);
        }
    }
}

// lambda_min = (a + c) - sqrt((a - c)^2 + b^2) with a = f32(Sxx) kd, c = f32(Syy) kd, b = f32(Sxy) ko and ko = 2 kd exactly
// (both are one f64 value scaled by a power of two, then rounded).  Scaling by a power of two commutes with rounding, so
// b = 2 y with y = f32(Sxy) kd, b*b = 4 (y*y) and fl(amc*amc + b*b) = fma(4, y*y, amc*amc) bit for bit — one constant
// (kept in a VGPR: an SGPR operand halves the issue rate of v_mul_f32, tools/valu_rates.hip) instead of two.
__device__ __forceinline__ int lambda_min_bits(int vxx, int vxy, int vyy, float kd)
{
    const float a = (float)vxx * kd, y = (float)vxy * kd, c = (float)vyy * kd;
    const float amc = a - c;
    const float y2 = y * y, amc2 = amc * amc;
    return __float_as_int((a + c) - sqrt_rn_normal(__builtin_fmaf(4.f, y2, amc2)));
}

// (Measured and dropped: the same formula on float PAIRS — v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 — executes 11 fewer
//  instructions per row and is 10 % SLOWER: plain f32 add/mul/fma issue at twice the rate of the packed and VOP3 forms,
//  tools/valu_rates.hip.)
// Full 64-key chunks of the wave's key buffer to its segment.  At the top of every block of rows and in front of every second row
// inside one: a row adds at most SW keys, so with fewer than 64 keys held two rows later there are fewer than 64 + 2 SW = NBUF.
#define OFK_PAIR_SPILL_KEYS()                                                                                          \
    {                                                                                                                  \
        const int nchunk = cnt >> 6;                                                                                   \
        for (int q = 0; q < nchunk; ++q) {                                                                             \
            const unsigned long long key = buf[q * 64 + lane];                                                         \
            if (written + q * 64 + lane < seg_cap) myseg[written + q * 64 + lane] = key;                               \
        }                                                                                                              \
        written += nchunk * 64;                                                                                        \
        const unsigned long long rest = buf[nchunk * 64 + lane];                                                       \
        __builtin_amdgcn_wave_barrier();                                                                               \
        buf[lane] = rest;                                                                                              \
        cnt -= nchunk * 64;                                                                                            \
    }
#define OFK_RING_READ(i)                                                                                               \
        const int oyye = s_ring[(4 * (i)) * 64 + lane], oyyo = s_ring[(4 * (i) + 1) * 64 + lane];                      \
        const int oxye = s_ring[(4 * (i) + 2) * 64 + lane], oxyo = s_ring[(4 * (i) + 3) * 64 + lane];
// One block of BS rows (expanded twice: IN = true for interior blocks); see OFK_EIG_ROWS for the conventions.
#define OFK_PAIR_ROWS(IN)                                                                                             \
    _Pragma("unroll") for (int i = 0; i < BS; ++i) {                                                                   \
        const int r = base + i;                                                                                        \
        int ge2 = (int)__builtin_amdgcn_ubfe((unsigned)curv[i], bsh, 8u), go2 = (int)__builtin_amdgcn_ubfe((unsigned)curv[i], bsh + 8u, 8u); \
        if (!(IN)) {                                                                                                   \
            if (edge_strip) { ge2 = __builtin_amdgcn_ds_bpermute(mir_e, ge2); go2 = __builtin_amdgcn_ds_bpermute(mir_o, go2); } \
        }                                                                                                              \
        if (i > 0 && (i & 1) == 0 && cnt >= 64) OFK_PAIR_SPILL_KEYS()                                                  \
        OFK_RING_READ(i)                                           /* the yy and xy box sums of BS rows ago (LDS ring) */ \
        const int r12e = g1e + ge2, r12o = g1o + go2;              /* Sobel column sums from row-pair sums: s = (g0+g1) + (g1+g2), */ \
        const int se = r01e + r12e, so = r01o + r12o, te = r12e - r01e, to = r12o - r01o;   /* t = g2 - g0 = (g1+g2) - (g0+g1): three full-rate adds */ \
        const int nso = -so, nse = -se, tt = te + to;                                                                  \
        const int dxe = DPP_SHR1(nso) + so;                                                                            \
        const int dxo = DPP_SHL1(se) + nse;                                                                            \
        int uye = DPP_SHR1(to) + te, uyo = DPP_SHL1(te) + to;   /* kept apart from "+ tt": v_add_u32_dpp + v_add_u32 (6 issue clocks) */ \
        asm("" : "+v"(uye), "+v"(uyo));                               /* where the compiler forms v_mov_b32_dpp + v_add3_u32 (8) */ \
        const int dye = uye + tt;                                                                                      \
        const int dyo = uyo + tt;                                                                                      \
        int hxxe, hxxo, hxye, hxyo, hyye, hyyo;                                                                        \
        int pxxe = __mul24(dxe, dxe), pyye = __mul24(dye, dye), pxye = __mul24(dxe, dye);                          \
        int pxxo = __mul24(dxo, dxo), pyyo = __mul24(dyo, dyo), pxyo = __mul24(dxo, dyo);                          \
        asm("" : "+v"(pxxe), "+v"(pxye), "+v"(pyye), "+v"(pxxo), "+v"(pxyo), "+v"(pyyo));                          \
        if (!(IN)) {                                                                                               \
            const int Y = Yp0 + r - 2;                                                                             \
            const bool rowflip = (Y < 0) | (Y >= h);                                                               \
            pxye = (rowflip != flip_e) ? -pxye : pxye;                                                             \
            pxyo = (rowflip != flip_o) ? -pxyo : pxyo;                                                             \
        }                                                                                                          \
        box_pair<BS / 2>(pxxe, pxxo, ad2, ad3, hxxe, hxxo);                                                        \
        box_pair<BS / 2>(pxye, pxyo, ad2, ad3, hxye, hxyo);                                                        \
        box_pair<BS / 2>(pyye, pyyo, ad2, ad3, hyye, hyyo);                                                        \
        vxxe += hxxe - rxxe[i]; vxye += hxye - oxye; vyye += hyye - oyye;                                              \
        vxxo += hxxo - rxxo[i]; vxyo += hxyo - oxyo; vyyo += hyyo - oyyo;                                              \
        rxxe[i] = hxxe; rxxo[i] = hxxo;                                                                                \
        s_ring[(4 * i) * 64 + lane] = hyye; s_ring[(4 * i + 1) * 64 + lane] = hyyo;                                    \
        s_ring[(4 * i + 2) * 64 + lane] = hxye; s_ring[(4 * i + 3) * 64 + lane] = hxyo;                                \
        const int yo = ya - 2 + r - BS;                                                                                \
        const int e2e = lambda_min_bits(vxxe, vxye, vyye, kdv), e2o = lambda_min_bits(vxxo, vxyo, vyyo, kdv);    \
        if ((IN) && !MASK) {                                                                                           \
            lmaxi = max(max(lmaxi, e2e), e2o);                                                                         \
        } else {                                                                                                       \
            const bool rown = (yo >= ya) & (yo < yb);                                                                  \
            bool cme = own_e & rown, cmo = own_o & rown;                                                               \
            if (MASK) {                                                                                                \
                const uint8_t *mrow = mk + (size_t)min(max(yo, 0), h - 1) * w;                                         \
                cme = cme & (mrow[xoc_e] != 0); cmo = cmo & (mrow[xoc_o] != 0);                                        \
            }                                                                                                          \
            lmaxi = max(max(lmaxi, cme ? e2e : 0), cmo ? e2o : 0);                                                     \
        }                                                                                                              \
        const int hm2e = max(max(DPP_SHR1(e2o), e2e), e2o);                                                         \
        const int hm2o = max(max(DPP_SHL1(e2e), e2o), e2e);                                                         \
        const int yn = yo - 1;                                                                                         \
        const int me = max(max(hm0e, hm1e), hm2e), mo = max(max(hm0o, hm1o), hm2o);                                    \
        bool ise = e1e >= max(me, thr1e), iso = e1o >= max(mo, thr1o);                                                 \
        if (!(IN)) {                                                                                                   \
            const bool rnms = (yn >= ya) & (yn < yb) & (yn >= 1) & (yn < h - 1);                                       \
            ise = ise & rnms; iso = iso & rnms;                                                                        \
        }                                                                                                              \
        if (MASK) {                                                                                                    \
            const uint8_t *mrow = mk + (size_t)min(max(yn, 0), h - 1) * w;                                             \
            ise = ise & (mrow[xoc_e] != 0); iso = iso & (mrow[xoc_o] != 0);                                            \
        }                                                                                                              \
        const unsigned long long bale = __builtin_amdgcn_ballot_w64(ise), balo = __builtin_amdgcn_ballot_w64(iso);     \
        const int ne = (int)__popcll(bale);                                                                            \
        OFK_PAIR_KEY_STORE()                                                                                           \
        cnt += ne + (int)__popcll(balo);                                                                               \
        e1e = e2e; e1o = e2o; hm0e = hm1e; hm0o = hm1o; hm1e = hm2e; hm1o = hm2o;                                      \
        r01e = r12e; r01o = r12o; g1e = ge2; g1o = go2;                                                                  \
    }

template <int BS> struct pair_geom {
    static constexpr int AN = BS / 2, PAD = (2 + AN + 3) & ~3, D = PAD - 2 - AN, SW = (125 - BS) & ~3;
};

// BS = 7 needs 42 ring values per lane; 14 stay in registers, 28 live in LDS (see s_ring)
// One WAVE per workgroup.  With four strips per 256-thread workgroup a 1080p frame (17 strips) left every fifth workgroup
// with one live wave: its three dead waves' register slots could not host another workgroup (4 x 168 VGPRs) until the live wave
// had marched down its 540 rows, and the 2560 workgroups of a 256-frame launch made 3.33 rounds over the chip's 768 slots.
// Wave-granular workgroups fill every slot: SQ counters put the wave-slot utilisation of the old launch at 66 %
// (profiles/r02a_valu_pmc.json: 6.1 G live wave-cycles in 9.3 G slot-cycles).
template <int BS, bool MASK>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(3, 8))) void k_mineig_pair(
    const uint8_t *__restrict__ gray, size_t gray_stride, int h, int w, int rows_per_strip, float kd, float ko,
    unsigned int *__restrict__ maxbits, const uint8_t *__restrict__ mask, size_t mask_stride, double quality,
    unsigned long long *__restrict__ seg, int seg_cap, int *__restrict__ seg_count, int *__restrict__ flags)
{
    static_assert(BS == 3 || BS == 5 || BS == 7, "pair sums are written for odd boxes up to 7");
    constexpr int AN = pair_geom<BS>::AN, PAD = pair_geom<BS>::PAD, D = pair_geom<BS>::D, SW = pair_geom<BS>::SW;
    constexpr int L0 = (BS + 1) / 2, L1 = (BS + 1 + SW) / 2;    // lanes [L0, L1) hold output columns (both slots)
    constexpr int NBUF = 64 + 2 * SW;                           // keys a wave can hold: fewer than 64 after a spill check + two rows (OFK_PAIR_SPILL_KEYS)
    __shared__ unsigned long long s_buf[NBUF + 64];             // + 64: a flush reads one whole 64-key chunk past the count
    // The vertical rings of the yy and xy box-row sums live in LDS (the xx ring stays in registers) and the key buffer holds two rows'
    // worth of keys instead of a block's: 125 VGPRs and 10 KB of LDS per wave = FOUR waves per SIMD (all 42 ring values in
    // registers: 168 VGPRs, three waves).  Alone on the chip 1.03 -> 0.96 ms; with the yy ring only (150 VGPRs, three waves, room
    // for a gray wave beside them) the kernel alone was no faster but the step 2.4 % shorter — four waves are another 1 % on top.
    __shared__ int s_ring[4 * BS * 64];
    const int lane = threadIdx.x;
    // XCD-aware block -> (image, chunk, strip block) map: workgroups are dealt round-robin over the 8 XCDs (blocks n and
    // n + 8 share one L2); an image's strips and chunks go to ONE XCD, so the halo columns and rows they share are fetched once.
    int bxi = blockIdx.x, byi = blockIdx.y, b = blockIdx.z;
    if ((gridDim.z & 7) == 0) {
        const unsigned per = gridDim.x * gridDim.y, n = blockIdx.z * per + blockIdx.y * gridDim.x + blockIdx.x, k = n >> 3, rem = k % per;
        b = 8 * (int)(k / per) + (int)(n & 7);
        byi = (int)(rem / gridDim.x); bxi = (int)(rem % gridDim.x);
    }
    const int sx = bxi;
    if (sx * SW - D >= w) return;                               // whole wave
    const int ya = byi * rows_per_strip, yb = min(h, ya + rows_per_strip);
    const uint8_t *img = gray + (size_t)b * gray_stride;
    const uint8_t *mk = MASK ? mask + (size_t)b * mask_stride : nullptr;
    unsigned long long *buf = s_buf;                           // one wave per workgroup: LDS executes its accesses in order, no barriers

    const int G0 = sx * SW - PAD;                              // gray column of lane 0's even slot: a multiple of 4
    const int ce = G0 + 2 * lane;                               // this lane's even gray / product column (odd: ce + 1)
    const bool edge_strip = G0 < 0 || G0 + 128 > w;             // some column is mirrored (wave-uniform)
    const bool flip_e = ce < 0 || ce >= w, flip_o = ce + 1 < 0 || ce + 1 >= w;
    const int xo_e = ce - (BS - 1) + AN, xo_o = xo_e + 1;       // columns of this lane's box sums / responses
    const bool lane_ok = lane >= L0 && lane < L1;
    const bool own_e = lane_ok && xo_e >= 0 && xo_e < w, own_o = lane_ok && xo_o >= 0 && xo_o < w;
    const bool nms_e = lane_ok && xo_e >= 1 && xo_e < w - 1, nms_o = lane_ok && xo_o >= 1 && xo_o < w - 1;
    const int xoc_e = min(max(xo_e, 0), w - 1), xoc_o = min(max(xo_o, 0), w - 1);

    const int Yp0 = ya - 1 - AN;                                // first product row
    const int nsteps = (yb - ya) + BS + 3;
    const int nstrips = (w + D + SW - 1) / SW;
    const int nseg = nstrips * (int)gridDim.y, segid = byi * nstrips + sx;
    unsigned long long *myseg = seg + ((size_t)b * nseg + segid) * seg_cap;
    int written = 0, cnt = 0;
    int lmaxi = 0;
    float published = 0.f;
    unsigned mb_seen = __hip_atomic_load(maxbits + b * OFK_MAX_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    float thr = (float)((double)__uint_as_float(mb_seen) * quality);

    // loader role: lane (lrow, lk) fetches dword lk of the 128-byte row segment of row lrow of a row pair (clamped into the
    // image: the clamped dwords only ever feed mirrored columns, which are patched below).  consumer role: the pair of
    // bytes (2l, 2l+1) sits in dword l/2, bytes (l&1)*2 and +1.
    const int lrow = lane >> 5, lk = lane & 31;
    const unsigned ldoff = (unsigned)min(max(G0 + 4 * lk, 0), w - 4);
    const int srcsel = (lane >> 1) * 4;
    const unsigned bsh = (unsigned)(lane & 1) * 16u;
    auto mirror = [&](int c) -> int { c = c < 0 ? -c : c; c = c >= w ? 2 * (w - 1) - c : c; return min(max((c - G0) >> 1, 0), 63) * 4; };
    const int mir_e = mirror(ce), mir_o = mirror(ce + 1);       // lane that holds the source of a mirrored column (else: itself)
    const int ad2 = ((lane - 2) & 63) * 4, ad3 = ((lane - 3) & 63) * 4;
    constexpr int NL = (BS + 1) / 2;                            // dword loads per block of BS rows (two rows each)
    auto row_of = [&](int r) -> int {
        int gy = Yp0 - 1 + min(r, nsteps - 1);
        gy = gy < 0 ? -gy : gy;
        return gy >= h ? 2 * (h - 1) - gy : gy;
    };
    // 32-bit byte offsets from the image's (wave-uniform) base: the row products are scalar, the lane adds its own part - a 64-bit
    // per-lane address would cost two quarter-rate v_mad_u64_u32 per load
#define OFK_PAIR_LOAD(R0, OUT)                                                                                         \
    _Pragma("unroll") for (int q_ = 0; q_ < NL; ++q_) {                                                                \
        const unsigned a_ = (unsigned)row_of((R0) + 2 * q_) * (unsigned)w, b_ = (unsigned)row_of((R0) + 2 * q_ + 1) * (unsigned)w; \
        OUT[q_] = (int)*reinterpret_cast<const unsigned *>(img + ((lrow ? b_ : a_) + ldoff));                          \
    }

    float kdv = kd;
    asm volatile("" : "+v"(kdv));                               // keep the scale factor in a VGPR (see lambda_min_bits)
    int r01e = 0, r01o = 0, g1e = 0, g1o = 0;                   // previous gray row and the sum of the two before it (per slot)
    int rxxe[BS], rxxo[BS];
#pragma unroll
    for (int i = 0; i < BS; ++i) { rxxe[i] = rxxo[i] = 0; }
#pragma unroll
    for (int i = 0; i < 4 * BS; ++i) s_ring[i * 64 + lane] = 0;
    int vxxe = 0, vxye = 0, vyye = 0, vxxo = 0, vxyo = 0, vyyo = 0;
    int e1e = 0, e1o = 0, hm0e = 0, hm0o = 0, hm1e = 0, hm1o = 0;
    int nextg[NL];
    OFK_PAIR_LOAD(0, nextg)

    for (int base = 0; base < nsteps; base += BS) {
        int curv[BS];
#pragma unroll
        for (int i = 0; i < BS; ++i) curv[i] = __builtin_amdgcn_ds_bpermute(srcsel + 128 * (i & 1), nextg[i >> 1]);
        {
            unsigned cur_seen = (unsigned)__builtin_amdgcn_readfirstlane((int)mb_seen);
            if (__float_as_uint(published) > cur_seen) cur_seen = __float_as_uint(published);
            thr = (float)((double)__uint_as_float(cur_seen) * quality);
        }
        if (cnt >= 64) {                                        // move full 64-key chunks to this strip's segment, publish the maximum
            OFK_PAIR_SPILL_KEYS()
            const float mw = wave_max_f32(lane_ok ? __int_as_float(lmaxi) : 0.f);
            if (mw > published) {
                if (lane == 0) (void)atomicMax(maxbits + b * OFK_MAX_STRIDE, __float_as_uint(mw));
                published = mw;
            }
            mb_seen = __hip_atomic_load(maxbits + b * OFK_MAX_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        OFK_PAIR_LOAD(base + BS, nextg)                         // prefetch the next block of rows
        const int r1 = base + BS - 1;
        const bool interior = !edge_strip && Yp0 + base - 2 >= 0 && Yp0 + r1 - 2 < h && base >= BS + 3 && ya - 3 + base - BS >= 1 &&
                              ya - 2 + r1 - BS < yb && ya - 3 + r1 - BS < h - 1;
        const int thrb = __float_as_int(fmaxf(thr, 0.f)) + 1;
        const int thr1e = nms_e ? thrb : 0x7fffffff, thr1o = nms_o ? thrb : 0x7fffffff;
        if (interior) { OFK_PAIR_ROWS(true) } else { OFK_PAIR_ROWS(false) }
    }
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < cnt; i += 64)
        if (written + i < seg_cap) myseg[written + i] = buf[i];
    if (lane == 0) {
        seg_count[(size_t)b * nseg + segid] = min(written + cnt, seg_cap);
        if (written + cnt > seg_cap) (void)atomicOr(flags, 1);
    }
    const float mw = wave_max_f32(lane_ok ? __int_as_float(lmaxi) : 0.f);
    if (lane == 0 && mw > published) (void)atomicMax(maxbits + b * OFK_MAX_STRIDE, __float_as_uint(mw));
}

#undef OFK_PAIR_LOAD
#undef OFK_PAIR_ROWS
#undef OFK_PAIR_SPILL_KEYS
#undef OFK_LOAD_BLOCK
#undef OFK_EIG_ROWS

// The pair kernel takes odd boxes up to 7 on images whose rows are dword multiples and wide enough for one mirror fold.
static bool pair_ok(int w, int block)
{
    return !g_ofk_tuning.no_pair && (block == 3 || block == 5 || block == 7) && (w & 3) == 0 && w >= 64;
}
static void pair_sw(int block, int *sw, int *d)
{
    switch (block) {
        case 3: *sw = pair_geom<3>::SW; *d = pair_geom<3>::D; break;
        case 5: *sw = pair_geom<5>::SW; *d = pair_geom<5>::D; break;
        default: *sw = pair_geom<7>::SW; *d = pair_geom<7>::D; break;
    }
}

// Geometry of the streaming kernels' segments for an image of h x w (shared by launcher and host-side sizing).
void ofk_stream_geometry(int h, int w, int block, int batch, int *rows, int *nseg, int *seg_cap)
{
    int SW = 61 - block, d = 0;
    if (pair_ok(w, block)) pair_sw(block, &SW, &d);
    const int strips = (w + d + SW - 1) / SW;
    // strip length when the batch fills the chip: every strip pays BS + 3 warm-up rows (270 rows: 3.7 %), but a launch makes
    // only a few rounds over the chip's 3072 wave slots (1080p, B = 256: 17 strips x chunks x 256 waves = 2.83 rounds at 540 rows,
    // 5.67 at 270), and the last round's tail costs more than the warm-up: isolated 1.126 ms at 540 rows, 1.071 at 270 or 360,
    // 1.08 at 135-216 (profiles/r02_eig_rows_sweep.txt)
    int r = batch >= 64 ? (h + (h + 269) / 270 - 1) / ((h + 269) / 270) : (batch >= 16 ? 128 : 32);
    if (g_ofk_tuning.eig_rows >= 8) r = g_ofk_tuning.eig_rows;  // ofk_set_tuning("eig_rows")
    while (strips * ((h + r - 1) / r) > 2048) r *= 2;           // the selection walks at most 2048 segments per image
    *rows = r; *nseg = strips * ((h + r - 1) / r);
    *seg_cap = ((SW * r / 4 + 64 + 63) / 64) * 64;              // strict local maxima fill at most a quarter of the strip
}

template <int BS>
static int launch_mineig_stream(hipStream_t s, const uint8_t *gray, size_t gray_stride, int h, int w, unsigned int *maxbits,
                                const uint8_t *mask, size_t mask_stride, double quality, unsigned long long *seg,
                                size_t seg_keys_per_image, int *seg_count, int seg_count_cap, int *flags, int batch, int *nseg_out,
                                int *segcap_out)
{
    constexpr int SW = 61 - BS;
    int rows, nseg, seg_cap;
    ofk_stream_geometry(h, w, BS, batch, &rows, &nseg, &seg_cap);
    if ((size_t)nseg * seg_cap > seg_keys_per_image || nseg > seg_count_cap) return -1;
    const int strips = (w + SW - 1) / SW;
    const double scale = 1.0 / (4.0 * BS * 255.0);
    const float kd = (float)(0.5 * scale * scale), ko = (float)(scale * scale);
    dim3 grid(strips, (h + rows - 1) / rows, batch);
    if (mask)
        hipLaunchKernelGGL((k_mineig_stream<BS, true>), grid, dim3(256 / 4), 0, s, gray, gray_stride, h, w, rows, kd, ko, maxbits, mask,
                           mask_stride, quality, seg, seg_cap, seg_count, flags);
    else
        hipLaunchKernelGGL((k_mineig_stream<BS, false>), grid, dim3(256 / 4), 0, s, gray, gray_stride, h, w, rows, kd, ko, maxbits, mask,
                           mask_stride, quality, seg, seg_cap, seg_count, flags);
    *nseg_out = nseg; *segcap_out = seg_cap;
    return 0;
}

template <int BS>
static int launch_mineig_pair(hipStream_t s, const uint8_t *gray, size_t gray_stride, int h, int w, unsigned int *maxbits,
                              const uint8_t *mask, size_t mask_stride, double quality, unsigned long long *seg,
                              size_t seg_keys_per_image, int *seg_count, int seg_count_cap, int *flags, int batch, int *nseg_out,
                              int *segcap_out)
{
    constexpr int SW = pair_geom<BS>::SW, D = pair_geom<BS>::D;
    int rows, nseg, seg_cap;
    ofk_stream_geometry(h, w, BS, batch, &rows, &nseg, &seg_cap);
    if ((size_t)nseg * seg_cap > seg_keys_per_image || nseg > seg_count_cap) return -1;
    const int strips = (w + D + SW - 1) / SW;
    const double scale = 1.0 / (4.0 * BS * 255.0);
    const float kd = (float)(0.5 * scale * scale), ko = (float)(scale * scale);
    if (ko != 2.f * kd) return -1;                              // lambda_min_bits relies on it (always true: power-of-two scaling)
    dim3 grid(strips, (h + rows - 1) / rows, batch);
    if (mask)
        hipLaunchKernelGGL((k_mineig_pair<BS, true>), grid, dim3(64), 0, s, gray, gray_stride, h, w, rows, kd, ko, maxbits, mask,
                           mask_stride, quality, seg, seg_cap, seg_count, flags);
    else
        hipLaunchKernelGGL((k_mineig_pair<BS, false>), grid, dim3(64), 0, s, gray, gray_stride, h, w, rows, kd, ko, maxbits, mask,
                           mask_stride, quality, seg, seg_cap, seg_count, flags);
    *nseg_out = nseg; *segcap_out = seg_cap;
    return 0;
}

// Response + 3x3 NMS + candidate keys + image maximum; no map.  Block sizes 3/5/7/12 run the streaming kernel, which
// writes per-strip SEGMENTS (*nseg_out > 0); the others run the LDS-tile kernel, which appends to the flat list.
int ofk_launch_mineig_cand(hipStream_t s, const uint8_t *gray, size_t gray_stride, int h, int w, int block,
                           unsigned int *maxbits, const uint8_t *mask, size_t mask_stride, double quality,
                           unsigned long long *cand, int cand_cap, int *cand_count, unsigned long long *seg,
                           size_t seg_keys_per_image, int *seg_count, int seg_count_cap, int *flags, int batch, int *nseg_out,
                           int *segcap_out)
{
    *nseg_out = 0; *segcap_out = 0;
#define ME_ARGS s, gray, gray_stride, h, w, block, nullptr, 0, maxbits, mask, mask_stride, quality, cand, cand_cap, cand_count, flags, batch
#define ST_ARGS s, gray, gray_stride, h, w, maxbits, mask, mask_stride, quality, seg, seg_keys_per_image, seg_count, seg_count_cap, flags, batch, nseg_out, segcap_out
    if (pair_ok(w, block)) {
        switch (block) {
            case 3: return launch_mineig_pair<3>(ST_ARGS);
            case 5: return launch_mineig_pair<5>(ST_ARGS);
            default: return launch_mineig_pair<7>(ST_ARGS);
        }
    }
    switch (block) {
        case 3: return launch_mineig_stream<3>(ST_ARGS);
        case 5: return launch_mineig_stream<5>(ST_ARGS);
        case 7: return launch_mineig_stream<7>(ST_ARGS);
        case 12: return launch_mineig_stream<12>(ST_ARGS);
        default: return launch_mineig_t<0, true>(ME_ARGS);       // LDS-tile kernel for the remaining block sizes
    }
#undef ST_ARGS
#undef ME_ARGS
}

// max over (mask != 0) of a response map that was supplied by the caller (ofk_select_corners)
__global__ __launch_bounds__(256) void k_maxbits(const float *__restrict__ eig, size_t eig_stride,
                                                 const uint8_t *__restrict__ mask, size_t mask_stride, int npx,
                                                 unsigned int *__restrict__ maxbits)
{
    const int b = blockIdx.y;
    const float *e = eig + (size_t)b * eig_stride;
    const uint8_t *mk = mask ? mask + (size_t)b * mask_stride : nullptr;
    float m = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npx; i += gridDim.x * 256)
        if (!mk || mk[i]) m = fmaxf(m, e[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float s_max[4];
    if ((threadIdx.x & 63) == 0) s_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
        if (m > 0.f) atomicMax(maxbits + b * OFK_MAX_STRIDE, __float_as_uint(m));
    }
}

void ofk_launch_maxbits(hipStream_t s, const float *eig, size_t eig_stride, const uint8_t *mask, size_t mask_stride,
                        int h, int w, unsigned int *maxbits, int batch)
{
    const int npx = h * w;
    int blocks = (npx + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_maxbits, dim3(blocks, batch), dim3(256), 0, s, eig, eig_stride, mask, mask_stride, npx, maxbits);
}

// ------------------------------------------------------------------------------------------------ threshold + NMS on a given map
__global__ __launch_bounds__(256) void k_nms(const float *__restrict__ eig, size_t eig_stride,
                                             const uint8_t *__restrict__ mask, size_t mask_stride, int h, int w,
                                             const unsigned int *__restrict__ maxbits, double quality,
                                             unsigned long long *__restrict__ cand, int cand_cap,
                                             int *__restrict__ cand_count, int *__restrict__ flags)
{
    __shared__ unsigned long long s_cand[1024];
    __shared__ int s_n, s_base;
    const int b = blockIdx.z, tid = threadIdx.x;
    const unsigned mb = maxbits[b * OFK_MAX_STRIDE];
    if (mb == 0) return;                                       // max <= 0: no corners
    if (tid == 0) s_n = 0;
    __syncthreads();
    const float thr = (float)((double)__uint_as_float(mb) * quality);
    const float *e = eig + (size_t)b * eig_stride;
    const int x = blockIdx.x * 64 + (tid & 63);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int y = blockIdx.y * 16 + (tid >> 6) * 4 + k;
        if (x >= 1 && x < w - 1 && y >= 1 && y < h - 1) {
            const size_t i = (size_t)y * w + x;
            const float v = e[i];
            if (v > thr && (!mask || mask[(size_t)b * mask_stride + i])) {
                const float *r0 = e + i - w, *r2 = e + i + w;
                const float m = fmaxf(fmaxf(fmaxf(r0[-1], r0[0]), fmaxf(r0[1], e[i - 1])),
                                      fmaxf(fmaxf(e[i + 1], r2[-1]), fmaxf(r2[0], r2[1])));
                if (!(m > v)) s_cand[atomicAdd(&s_n, 1)] = ((unsigned long long)(~__float_as_uint(v)) << 32) | (unsigned)~(unsigned)(y * w + x);
            }
        }
    }
    __syncthreads();
    const int n = s_n;
    if (n == 0) return;
    if (tid == 0) s_base = atomicAdd(cand_count + b * OFK_CNT_STRIDE, n);
    __syncthreads();
    const int base = s_base;
    for (int i = tid; i < n; i += 256) {
        if (base + i < cand_cap) cand[(size_t)b * cand_cap + base + i] = s_cand[i];
        else if (i == 0 || base + i == cand_cap) atomicOr(flags, 1);
    }
}

void ofk_launch_nms(hipStream_t s, const float *eig, size_t eig_stride, const uint8_t *mask, size_t mask_stride, int h,
                    int w, const unsigned int *maxbits, double quality, unsigned long long *cand, int cand_cap,
                    int *cand_count, int *flags, int batch)
{
    dim3 grid((w + 63) / 64, (h + 15) / 16, batch);
    hipLaunchKernelGGL(k_nms, grid, dim3(256), 0, s, eig, eig_stride, mask, mask_stride, h, w, maxbits, quality, cand,
                       cand_cap, cand_count, flags);
}

// ------------------------------------------------------------------------------------------------ sort + greedy min-distance
#define SEL_T 256                                             // threads of the sort + greedy workgroup (k_select_greedy)
#define SEL_NB 512                                            // histogram bins per refinement level of its generic path

// Selection = three short kernels, none of which needs more of a CU than one retiring wave of the response kernel frees
// (round 2: ONE 1024-thread workgroup per image with 57 KB of LDS — beside the other slice's response kernel, which rents every
// VGPR and all 160 KB of LDS of every CU in 128-VGPR / 10 KB wave-sized pieces, it waited for a CU to drain: 61 us alone, 459 us mean /
// 1.4 ms worst under the schedule, profiles/r02_kernel_stats_slices2.csv):
//   k_select_prep   SEL_G workgroups per image: keys above the exact quality threshold -> flat list, 1024-bin histogram of their
//                   bit patterns over [key of the maximum, threshold key);
//   k_select_pick   SEL_G workgroups per image: the leading histogram bins that hold at most `tgt` ~ 2 maxCorners keys give a cut
//                   T; every workgroup moves its share of the keys below T into the image's pick list (<= tgt keys);
//   k_select_greedy one 256-thread workgroup per image, 9 KB of LDS at 500 corners: bitonic sort of the pick list, then the
//                   greedy minimum-distance pass against a coarse GRID of the accepted corners (cells >= minDistance wide, 3 x 3
//                   cells probed per candidate — what OpenCV does; round 2 compared every candidate with every accepted corner);
//                   when the picked keys do not fill the budget (heavy ties, tight distances) it goes on with the generic loop:
//                   next cut by multi-level histogram over the flat list, gather, sort, greedy.
// Result identical to a full sort (value desc, index desc) followed by the serial greedy pass.
#define SEL_G 16
#define SEL_HB 1024

__host__ __device__ __forceinline__ int sel_tgt(int max_corners)
{
    int t = 512;                                                // keys handled per round: about twice the corners wanted
    while (t < 2 * max_corners && t < OFK_CHUNK) t <<= 1;
    return t;
}
// per-image counter line (OFK_CNT_STRIDE ints): [0] keys in the flat list, [1] keys in the pick list, [2] histogram bins below the cut
#define SEL_CNT_PICK 1
#define SEL_CNT_BINS 2

__global__ __launch_bounds__(256) void k_select_prep(unsigned long long *__restrict__ cand_all, int cand_cap, int *__restrict__ cand_count,
                                                     const unsigned long long *__restrict__ seg, int seg_cap,
                                                     const int *__restrict__ seg_count, int nseg,
                                                     const unsigned int *__restrict__ maxbits, double quality,
                                                     unsigned *__restrict__ hist, const int *__restrict__ limit)
{
    __shared__ unsigned s_h[SEL_HB];
    const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    if (limit && limit[b] <= 0) return;
    const unsigned mb = maxbits[b * OFK_MAX_STRIDE];
    if (mb == 0) return;
    const float thr = (float)((double)__uint_as_float(mb) * quality);
    const unsigned a_hi = ~mb, kend_hi = ~__float_as_uint(thr);                    // key < kend  <=>  key_hi < kend_hi (kend's low word is 0)
    if (!(a_hi < kend_hi)) return;                               // nothing is strictly above the threshold
    const unsigned width = kend_hi - a_hi;
    const int shift = width <= SEL_HB ? 0 : 32 - __clz((int)(width - 1)) - 10;
    for (int i = tid; i < SEL_HB; i += 256) s_h[i] = 0;
    __syncthreads();
    unsigned long long *cand = cand_all + (size_t)b * cand_cap;
    if (nseg > 0) {
        const unsigned long long *sbase = seg + (size_t)b * nseg * seg_cap;
        for (int sg = g; sg < nseg; sg += (int)gridDim.x) {
            const int n = min(seg_count[(size_t)b * nseg + sg], seg_cap);
            const unsigned long long *sp = sbase + (size_t)sg * seg_cap;
            for (int i0 = 0; i0 < n; i0 += 1024) {                   // uniform trip count per wave: ballots below need every lane
                // four keys per thread and trip: the loads are in flight together and one atomic reserves the list slots of all four
                // (the loop is a chain of load and atomic round trips: ~680 keys per segment at 1080p are one trip now, not three)
                unsigned long long key[4], bal[4];
                int cnt4 = 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) { const int i = i0 + 256 * q + tid; key[q] = i < n ? OFK_SEG_KEY_DECODE(sp[i]) : ~0ull; }
#pragma unroll
                for (int q = 0; q < 4; ++q) { bal[q] = __ballot((unsigned)(key[q] >> 32) < kend_hi); cnt4 += __popcll(bal[q]); }
                if (cnt4) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(cand_count + b * OFK_CNT_STRIDE, cnt4);
                    base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned hi = (unsigned)(key[q] >> 32);
                        if (hi < kend_hi) {
                            const int pos = base + __popcll(bal[q] & ((1ull << lane) - 1));
                            if (pos < cand_cap) cand[pos] = key[q];
                            atomicAdd(&s_h[(hi - a_hi) >> shift], 1u);
                        }
                        base += __popcll(bal[q]);
                    }
                }
            }
        }
    } else {
        // the flat list exists already (k_nms, or the LDS-tile response kernel with its running threshold): histogram only
        const int C = min(cand_count[b * OFK_CNT_STRIDE], cand_cap);
        for (int i0 = g * 1024; i0 < C; i0 += (int)gridDim.x * 1024) {
            unsigned long long key[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int i = i0 + 256 * q + tid; key[q] = i < C ? cand[i] : ~0ull; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned hi = (unsigned)(key[q] >> 32);
                if (hi < kend_hi && hi >= a_hi) atomicAdd(&s_h[(hi - a_hi) >> shift], 1u);
            }
        }
    }
    __syncthreads();
    unsigned *hb = hist + (size_t)b * SEL_HB;
    for (int i = tid; i < SEL_HB; i += 256) if (s_h[i]) atomicAdd(hb + i, s_h[i]);
}

// Inclusive prefix over NPT consecutive values per thread of a 256-thread workgroup; returns the inclusive prefix of the thread's
// LAST value.  s_w: 4 words.  Two barriers.
__device__ __forceinline__ unsigned block_scan256(unsigned local_sum, unsigned *s_w, unsigned &total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned incl = local_sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned n_ = __shfl_up(incl, o); if (lane >= o) incl += n_; }
    __syncthreads();
    if (lane == 63) s_w[wave] = incl;
    __syncthreads();
    unsigned woff = 0; total = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const unsigned ws = s_w[q]; if (q < wave) woff += ws; total += ws; }
    return woff + incl;
}

__global__ __launch_bounds__(256) void k_select_pick(const unsigned long long *__restrict__ cand_all, int cand_cap, int *__restrict__ cand_count,
                                                     const unsigned int *__restrict__ maxbits, double quality, const unsigned *__restrict__ hist,
                                                     const int *__restrict__ limit, int max_corners_all,
                                                     unsigned long long *__restrict__ sel_keys, int sel_stride)
{
    __shared__ unsigned s_w[4];
    __shared__ int s_D;
    const int b = blockIdx.y, g = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int mc = limit ? min(max_corners_all, limit[b]) : max_corners_all;
    if (mc <= 0) return;
    const unsigned mb = maxbits[b * OFK_MAX_STRIDE];
    if (mb == 0) return;
    const float thr = (float)((double)__uint_as_float(mb) * quality);
    const unsigned a_hi = ~mb, kend_hi = ~__float_as_uint(thr);
    if (!(a_hi < kend_hi)) return;
    int *cnt = cand_count + b * OFK_CNT_STRIDE;
    const int total_keys = cnt[0];
    if (total_keys > cand_cap || total_keys <= 0) return;        // overflow is reported by k_select_greedy
    const int tgt = sel_tgt(mc);
    const unsigned width = kend_hi - a_hi;
    const int hshift = width <= SEL_HB ? 0 : 32 - __clz((int)(width - 1)) - 10;
    // the cut: D = number of leading bins whose inclusive prefix fits the budget (prefixes are non-decreasing: a leading run)
    if (tid == 0) s_D = 0;
    const uint4 h4 = reinterpret_cast<const uint4 *>(hist + (size_t)b * SEL_HB)[tid];       // bins 4 tid .. 4 tid + 3
    unsigned tot;
    const unsigned i3 = block_scan256(h4.x + h4.y + h4.z + h4.w, s_w, tot);
    const unsigned i2 = i3 - h4.w, i1 = i2 - h4.z, i0 = i1 - h4.y;
    const int fit = ((int)i0 <= tgt) + ((int)i1 <= tgt) + ((int)i2 <= tgt) + ((int)i3 <= tgt);
    if (fit) atomicAdd(&s_D, fit);
    __syncthreads();
    const int D = s_D;
    if (g == 0 && tid == 0) cnt[SEL_CNT_BINS] = D;
    if (D == 0) return;                                          // the first bin alone exceeds the budget (ties): generic path
    const unsigned long long t64 = (unsigned long long)a_hi + ((unsigned long long)D << hshift);
    const unsigned T_hi = t64 < (unsigned long long)kend_hi ? (unsigned)t64 : kend_hi;
    const unsigned long long *cand = cand_all + (size_t)b * cand_cap;
    unsigned long long *dst = sel_keys + (size_t)b * sel_stride;
    const int per = ((total_keys + SEL_G - 1) / SEL_G + 255) & ~255;
    const int ibeg = min(total_keys, g * per), iend = min(total_keys, ibeg + per);
    for (int i0_ = ibeg; i0_ < iend; i0_ += 8 * 256) {           // eight loads in flight per thread: one trip at 1080p (2.9 k keys per workgroup)
        unsigned long long key[8], bal[8];
        int c8 = 0;
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int i = i0_ + 256 * q + tid; key[q] = i < iend ? cand[i] : ~0ull; }
#pragma unroll
        for (int q = 0; q < 8; ++q) { const unsigned hi = (unsigned)(key[q] >> 32); bal[q] = __ballot(hi < T_hi && hi >= a_hi); c8 += __popcll(bal[q]); }
        if (c8) {
            int base = 0;
            if (lane == 0) base = atomicAdd(cnt + SEL_CNT_PICK, c8);
            base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if ((bal[q] >> lane) & 1ull) {
                    const int pos = base + __popcll(bal[q] & ((1ull << lane) - 1));
                    if (pos < sel_stride) dst[pos] = key[q];
                }
                base += __popcll(bal[q]);
            }
        }
    }
}

extern __shared__ __attribute__((aligned(16))) uint8_t sel_smem[];

// LDS map (bytes, tgtA = sel_tgt(max_corners_all)):  [0, 8 tgtA) sort buffer of the first round | [0, 4 tgtA) sorted indices,
// sort buffer of later rounds (tgtA / 2 keys), histogram of the cut search | [4 tgtA, 6 tgtA) grid heads u16 | [6 tgtA, + 4 mc)
// accepted corners x | y << 16 | then mc u16 chain links.  Total max(8 tgtA, 6 tgtA + 6 mc): 9.1 KB at 500 corners.
// NT = threads of the workgroup: 256 (the pipeline's: fits beside the response kernel) or 1024 (small batches of many corners - 32 4K
// pairs with 2000 corners each leave the chip nearly empty, and a bitonic sort of 4096 keys is four times shorter on sixteen waves).
template <int NT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT == 256 ? 8 : 4, 8))) void k_select_greedy(const unsigned long long *__restrict__ cand_all, int cand_cap,
                                                         const int *__restrict__ cand_count, const unsigned int *__restrict__ maxbits,
                                                         double quality, int w, int h, int max_corners_all, float min_distance,
                                                         float *__restrict__ pts, int pts_stride, int *__restrict__ counts,
                                                         const int *__restrict__ limit, const unsigned long long *__restrict__ sel_keys,
                                                         int sel_stride, int tgtA, int cs)
{
    constexpr int NW = NT / 64;
    constexpr int NB = NT > SEL_NB ? NT : SEL_NB;               // histogram bins per refinement level of the generic path (>= one per thread)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int max_corners = limit ? min(max_corners_all, limit[b]) : max_corners_all;
    if (max_corners <= 0) { if (tid == 0) counts[b] = 0; return; }
    unsigned long long *s_key = reinterpret_cast<unsigned long long *>(sel_smem);
    unsigned *s_idx = reinterpret_cast<unsigned *>(sel_smem);
    unsigned *s_hist = reinterpret_cast<unsigned *>(sel_smem);
    volatile unsigned short *s_grid = reinterpret_cast<volatile unsigned short *>(sel_smem + 4 * (size_t)tgtA);
    int *s_acc = reinterpret_cast<int *>(sel_smem + 6 * (size_t)tgtA);
    volatile unsigned short *s_next = reinterpret_cast<volatile unsigned short *>(sel_smem + 6 * (size_t)tgtA + 4 * (size_t)max_corners_all);
    __shared__ unsigned long long s_conf[128];                  // conflict matrices of the current and the next greedy round
    __shared__ unsigned s_w[NW];
    __shared__ int s_n, s_nacc, s_D, s_cum;

    const unsigned long long *cand = cand_all + (size_t)b * cand_cap;
    const unsigned mb = maxbits[b * OFK_MAX_STRIDE];
    if (tid == 0) { s_nacc = 0; s_n = 0; counts[b] = 0; }
    if (mb == 0) return;
    const float thr = (float)((double)__uint_as_float(mb) * quality);
    // keys of interest: [a, kend);  v > thr  <=>  key < (~bits(thr)) << 32
    const unsigned long long kend = (unsigned long long)(~__float_as_uint(thr)) << 32;
    unsigned long long a = (unsigned long long)(~mb) << 32;     // smallest possible key (value == max)
    const float md2 = min_distance * min_distance;
    const bool use_dist = min_distance >= 1.f;
    const int total_keys = cand_count[b * OFK_CNT_STRIDE];
    if (total_keys > cand_cap) {                                // flat list overflow (uniform): the host reports OFK_E_CAPACITY
        if (tid == 0) counts[b] = -1;
        return;
    }
    const int C = total_keys;
    if (C <= 0 || !(thr < __uint_as_float(mb))) return;         // nothing is strictly above the threshold
    const int tgt = sel_tgt(max_corners);
    const int gw = (w + cs - 1) / cs, gh = (h + cs - 1) / cs;
    const int nsel = min(cand_count[b * OFK_CNT_STRIDE + SEL_CNT_PICK], min(tgt, sel_stride));
    const int Dbins = cand_count[b * OFK_CNT_STRIDE + SEL_CNT_BINS];
    bool fast = Dbins > 0;                                      // k_select_pick made the first cut
    bool grid_ready = false;
    __syncthreads();
    while (true) {
        unsigned long long T = kend;
        int n = 0;
        if (fast) {
            const unsigned a_hi = ~mb, width = (unsigned)(kend >> 32) - a_hi;
            const int hshift = width <= SEL_HB ? 0 : 32 - __clz((int)(width - 1)) - 10;
            const unsigned long long cut = ((unsigned long long)a_hi + ((unsigned long long)Dbins << hshift)) << 32;
            T = cut < kend ? cut : kend;
            n = nsel;
            const unsigned long long *src = sel_keys + (size_t)b * sel_stride;
            for (int i = tid; i < n; i += NT) s_key[i] = src[i];
        } else {
            if (!(a < kend)) break;
            // ---- choose T in (a, kend] so that 1 <= #{a <= key < T} <= cap (or detect that none is left)
            const int cap = tgtA / 2;                           // later rounds sort inside the index region: the grid and the accepted set live behind it
            unsigned long long curA = a, curB = kend;
            int taken = 0;
            bool have_cut = false;
            for (int level = 0; level < 9; ++level) {
                const unsigned long long width = curB - curA;
                const int shift = width <= NB ? 0 : 64 - __clzll((long long)(width - 1)) - (NB == 512 ? 9 : 10);
                const int nb = (int)((width - 1) >> shift) + 1;
                for (int i = tid; i < NB; i += NT) s_hist[i] = 0;
                if (tid == 0) { s_D = 0; s_cum = 0; }
                __syncthreads();
                for (int i0 = tid; i0 < C; i0 += 4 * NT) {
                    unsigned long long key[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { const int i = i0 + q * NT; key[q] = i < C ? cand[i] : ~0ull; }
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (key[q] >= curA && key[q] < curB) atomicAdd(&s_hist[(unsigned)((key[q] - curA) >> shift)], 1u);
                }
                __syncthreads();
                // inclusive prefix over the bins (NB / NT per thread); D = #bins whose inclusive prefix fits the budget
                constexpr int BPT = NB / NT;
                unsigned hv[BPT], loc = 0;
#pragma unroll
                for (int q = 0; q < BPT; ++q) { hv[q] = s_hist[BPT * tid + q]; loc += hv[q]; }
                unsigned incl = loc;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const unsigned n_ = __shfl_up(incl, o); if (lane >= o) incl += n_; }
                if (lane == 63) s_w[wave] = incl;
                __syncthreads();
                unsigned woff = 0, total = 0;
#pragma unroll
                for (int q = 0; q < NW; ++q) { const unsigned ws = s_w[q]; if (q < wave) woff += ws; total += ws; }
                const int budget = cap - taken;
                unsigned pre = woff + incl - loc;               // exclusive prefix of this thread's first bin
                int fit = 0;
#pragma unroll
                for (int q = 0; q < BPT; ++q) { pre += hv[q]; fit += (int)pre <= budget; }
                if (fit) atomicAdd(&s_D, fit);
                __syncthreads();
                const int D = min(s_D, nb);
                // the thread that owns bin D - 1 knows the keys below the cut (prefixes are non-decreasing: the fitting bins are a leading run)
                if (D > 0 && (D - 1) / BPT == tid) {
                    unsigned p2 = woff + incl - loc;
#pragma unroll
                    for (int q = 0; q < BPT; ++q) { p2 += hv[q]; if (BPT * tid + q == D - 1) s_cum = (int)p2; }
                }
                __syncthreads();
                taken += s_cum;
                if (level == 0 && total == 0) break;            // no key left in [a, kend)
                if (D >= nb) { T = curB; have_cut = true; break; }             // everything in [curA, curB) fits
                const unsigned long long newA = curA + ((unsigned long long)D << shift);
                if (taken >= cap / 4) { T = newA; have_cut = true; break; }
                curA = newA;                                    // descend into the first bin that did not fit
                const unsigned long long bin_end = newA + (1ull << shift);
                if (bin_end < curB) curB = bin_end;
                __syncthreads();                                // s_D / s_cum are reset at the top of the next level
            }
            if (!have_cut) break;                               // (nine levels of nine bits always reach single keys: a cut exists unless nothing is left)
            // ---- gather keys in [a, T)
            __syncthreads();
            if (tid == 0) s_n = 0;
            __syncthreads();
            for (int i0 = tid; i0 < C; i0 += 4 * NT) {
                unsigned long long key[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { const int i = i0 + q * NT; key[q] = i < C ? cand[i] : ~0ull; }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (key[q] >= a && key[q] < T) { const int slot = atomicAdd(&s_n, 1); if (slot < cap) s_key[slot] = key[q]; }
            }
            __syncthreads();
            n = min(s_n, cap);
        }
        fast = false;
        int npad = 64;
        while (npad < n) npad <<= 1;
        for (int i = n + tid; i < npad; i += NT) s_key[i] = ~0ull;
        __syncthreads();
        // ---- bitonic sort ascending.  Element i is handled by thread i % NT: for j >= NT both partners of a compare-exchange
        // belong to the same thread, for j < 64 to the same wave (LDS executes a wave's accesses in order) — only the steps with
        // 64 <= j < NT exchange between waves and need the workgroup barrier, before and after.
        for (int kk = 2; kk <= npad; kk <<= 1)
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < npad; i += NT) {
                    const int p = i ^ j;
                    if (p > i) {
                        const unsigned long long x0 = s_key[i], x1 = s_key[p];
                        const bool up = (i & kk) == 0;
                        if ((x0 > x1) == up) { s_key[i] = x1; s_key[p] = x0; }
                    }
                }
                const int nj = j > 1 ? (j >> 1) : kk;           // the step that follows (first step of the next stage: j = kk)
                if ((j >= 64 && j < NT) || (nj >= 64 && nj < NT)) __syncthreads();
                else __builtin_amdgcn_wave_barrier();
            }
        __syncthreads();
        // ---- sorted keys -> sorted pixel positions x | y << 16, in place (the key holds ~index: equal responses sort by DESCENDING
        //      index); the two divisions per candidate happen here, once, spread over the whole workgroup
        {
            unsigned idxv[OFK_CHUNK / NT];
#pragma unroll
            for (int q = 0; q < OFK_CHUNK / NT; ++q) {
                const int i = tid + q * NT;
                const unsigned idx = i < n ? ~(unsigned)(s_key[i] & 0xffffffffu) : 0u;
                const unsigned y = idx / (unsigned)w;
                idxv[q] = (idx - y * (unsigned)w) | (y << 16);
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < OFK_CHUNK / NT; ++q) { const int i = tid + q * NT; if (i < n) s_idx[i] = idxv[q]; }
        }
        if (!grid_ready) {                                      // the first sort may have run over the grid's place
            unsigned *g32 = reinterpret_cast<unsigned *>(sel_smem + 4 * (size_t)tgtA);
            for (int i = tid; i < tgtA / 2; i += NT) g32[i] = 0xffffffffu;
            grid_ready = true;
        }
        __syncthreads();
        // ---- greedy over the sorted chunk, 64 candidates per round, software-pipelined over the waves: while wave 0 resolves round r
        //      (accepted-set test through the grid, acceptance sweeps, stores), the other waves build the 64 x 64 conflict matrix of
        //      round r + 1 into the other half of s_conf - one workgroup barrier per round, the matrix off wave 0's path
        auto conflict_rows = [&](int base_, unsigned long long *conf) {
            // rows of the matrix of the round starting at base_, shared by the waves 1 .. NW-1 (all of it by wave 0 when alone)
            const int ci_ = base_ + lane;
            const unsigned xy_ = ci_ < n ? s_idx[ci_] : 0u;
            const int x_ = (int)(xy_ & 0xffffu), y_ = (int)(xy_ >> 16);
            const int w0 = NW > 1 ? wave - 1 : 0, nw = NW > 1 ? NW - 1 : 1;
            for (int j = __builtin_amdgcn_readfirstlane(w0); j < 64; j += nw) {
                const int jx = __builtin_amdgcn_readlane(x_, j), jy = __builtin_amdgcn_readlane(y_, j);
                const int dx = x_ - jx, dy = y_ - jy;
                const unsigned long long bj = __ballot((float)(dx * dx + dy * dy) < md2);
                if (lane == 0) conf[j] = bj;
            }
        };
        if (use_dist && n > 0 && (NW == 1 || wave > 0)) conflict_rows(0, s_conf);
        __syncthreads();
        const float inv_cs = 1.f / (float)cs;                   // floor((c + 0.5) / cs) exactly: |error| <= 2.4e-7 * 16384 / cs < 0.5 / cs
        int round = 0;
        for (int base = 0; base < n; base += 64, ++round) {
            const int nacc = s_nacc;
            if (nacc >= max_corners) break;
            if (wave > 0 || NW == 1) {
                if (use_dist && base + 64 < n) conflict_rows(base + 64, s_conf + 64 * ((round + 1) & 1));
            }
            if (wave == 0) {
                const int ci = base + lane;
                const bool live = ci < n;
                const unsigned xy = live ? s_idx[ci] : 0u;
                const int cx = (int)(xy & 0xffffu), cy = (int)(xy >> 16);
                // the accepted set through the grid: corners closer than minDistance sit in the 3 x 3 cells around the candidate's
                const int gx = (int)(((float)cx + 0.5f) * inv_cs), gy = (int)(((float)cy + 0.5f) * inv_cs);
                bool rej = false;
                if (use_dist && live) {
                    unsigned head[9];                           // the nine cell heads first (independent LDS reads), then the short chains
#pragma unroll
                    for (int q = 0; q < 9; ++q) {
                        const int yy = gy + q / 3 - 1, xx = gx + q % 3 - 1;
                        const bool in = yy >= 0 && yy < gh && xx >= 0 && xx < gw;
                        head[q] = in ? (unsigned)s_grid[in ? yy * gw + xx : 0] : 0xffffu;
                    }
#pragma unroll
                    for (int q = 0; q < 9; ++q) {
                        unsigned j = head[q];
                        while (j != 0xffffu) {
                            const int aj = s_acc[j];
                            const int dx = cx - (aj & 0xffff), dy = cy - (aj >> 16);
                            rej = rej || (float)(dx * dx + dy * dy) < md2;
                            j = s_next[j];
                        }
                    }
                }
                const unsigned long long alive = __ballot(live && !rej);   // survivors of the accepted-set test, best first
                const unsigned long long myconf = use_dist ? s_conf[64 * (round & 1) + lane] : 0ull;   // lanes clashing with candidate `lane`
                // Greedy acceptance in rank order, a few parallel sweeps instead of one scalar step per candidate: U = candidates not
                // decided yet, with everything that clashes with an accepted one already removed.  A lane whose earlier clashing
                // lanes are all decided is accepted in this sweep (the lowest undecided lane always is); the accepted lanes and
                // whatever clashes with them (the matrix is symmetric and has its diagonal set) leave U.  Decisions only depend on
                // earlier lanes, so the result is that of the sequential pass, and its first `room` members are what the sequential
                // pass would have accepted before running out of room.
                const unsigned long long lower = (1ull << lane) - 1ull;
                unsigned long long U = alive, A = 0;
                while (U) {
                    const bool join = ((U >> lane) & 1ull) && (myconf & U & lower) == 0ull;
                    const unsigned long long J = __ballot(join);
                    A |= J;
                    U &= ~(__ballot((myconf & J) != 0ull) | J);
                }
                const int room = max_corners - nacc;
                const unsigned long long acc = __ballot(((A >> lane) & 1ull) && __popcll(A & lower) < room);
                const bool mine = (acc >> lane) & 1ull;
                const int pos = nacc + __popcll(acc & lower);   // accepted candidates store in parallel, in rank order
                if (mine) {
                    s_acc[pos] = cx | (cy << 16);
                    pts[((size_t)b * pts_stride + pos) * 2] = (float)cx;
                    pts[((size_t)b * pts_stride + pos) * 2 + 1] = (float)cy;
                }
                if (use_dist) {
                    // link the new corners into their cells: all lanes store their index as the cell's head, the lane that reads its
                    // own index back won and chains to the old head; the others of that cell go again (order inside a cell is free)
                    bool pend = mine;
                    const int cell = gy * gw + gx;
                    while (__ballot(pend)) {
                        unsigned old = 0xffffu;
                        if (pend) old = s_grid[cell];
                        __builtin_amdgcn_wave_barrier();
                        if (pend) s_grid[cell] = (unsigned short)pos;
                        __builtin_amdgcn_wave_barrier();
                        if (pend && s_grid[cell] == (unsigned short)pos) { s_next[pos] = (unsigned short)old; pend = false; }
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                if (lane == 0) s_nacc = nacc + __popcll(acc);
            }
            __syncthreads();
        }
        if (s_nacc >= max_corners) break;
        a = T;
        __syncthreads();
    }
    if (tid == 0) counts[b] = s_nacc;
}

// Zeroes the three per-image arrays a detection step starts from (response maxima, candidate counters, key histogram) in ONE launch:
// three hipMemsetAsync calls are three fill kernels with a dependency gap each on the critical chain of a step.
__global__ __launch_bounds__(256) void k_zero_detect_state(unsigned *__restrict__ a, int na, unsigned *__restrict__ b, int nb, unsigned *__restrict__ c, int nc)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < na) { a[i] = 0u; return; }
    i -= na;
    if (i < nb) { b[i] = 0u; return; }
    i -= nb;
    if (i < nc) c[i] = 0u;
}
void ofk_launch_zero_detect_state(hipStream_t s, unsigned int *maxbits, int *cand_count, unsigned *sel_hist, int batch)
{
    const int na = batch * OFK_MAX_STRIDE, nb = batch * OFK_CNT_STRIDE, nc = sel_hist ? batch * SEL_HB : 0;
    hipLaunchKernelGGL(k_zero_detect_state, dim3((na + nb + nc + 255) / 256), dim3(256), 0, s, maxbits, na, reinterpret_cast<unsigned *>(cand_count), nb, sel_hist, nc);
}

void ofk_launch_select(hipStream_t s, unsigned long long *cand, int cand_cap, int *cand_count, const unsigned long long *seg,
                       int seg_cap, const int *seg_count, int nseg, const unsigned int *maxbits, double quality, int h, int w,
                       int max_corners, float min_distance, float *pts, int pts_stride, int *counts, const int *limit, int batch,
                       unsigned *sel_hist, unsigned long long *sel_keys, bool hist_is_zero)
{
    if (!hist_is_zero) (void)hipMemsetAsync(sel_hist, 0, (size_t)batch * SEL_HB * sizeof(unsigned), s);
    // (SEL_G = 16 workgroups per image walk ~4 segments each; one segment per workgroup - 64 per image - was measured: 44 -> 67 us, the
    //  histogram flush of four times as many workgroups costs more than the shorter chains save)
    hipLaunchKernelGGL(k_select_prep, dim3(SEL_G, batch), dim3(256), 0, s, cand, cand_cap, cand_count, seg, seg_cap, seg_count, nseg, maxbits,
                       quality, sel_hist, limit);
    hipLaunchKernelGGL(k_select_pick, dim3(SEL_G, batch), dim3(256), 0, s, cand, cand_cap, cand_count, maxbits, quality, sel_hist, limit,
                       max_corners, sel_keys, OFK_CHUNK);
    const int tgtA = sel_tgt(max_corners);
    // grid of the accepted corners: cells at least minDistance wide, coarsened until the image has at most tgtA of them
    const int cs0 = min_distance >= 1.f ? (int)ceilf(min_distance) : 1;
    int cs = cs0;
    while ((long long)((w + cs - 1) / cs) * ((h + cs - 1) / cs) > tgtA) cs += cs0;
    const size_t lds = (size_t)8 * tgtA > (size_t)6 * tgtA + (size_t)6 * max_corners + 16 ? (size_t)8 * tgtA : (size_t)6 * tgtA + (size_t)6 * max_corners + 16;
    if (batch <= 64 && tgtA >= 2048)
        hipLaunchKernelGGL(k_select_greedy<1024>, dim3(batch), dim3(1024), lds, s, cand, cand_cap, cand_count, maxbits, quality, w, h, max_corners,
                           min_distance, pts, pts_stride, counts, limit, sel_keys, OFK_CHUNK, tgtA, cs);
    else
        hipLaunchKernelGGL(k_select_greedy<SEL_T>, dim3(batch), dim3(SEL_T), lds, s, cand, cand_cap, cand_count, maxbits, quality, w, h, max_corners,
                           min_distance, pts, pts_stride, counts, limit, sel_keys, OFK_CHUNK, tgtA, cs);
}
