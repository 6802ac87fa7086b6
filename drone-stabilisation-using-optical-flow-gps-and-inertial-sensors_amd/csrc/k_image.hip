// k_image.hip — dense u8 stages: BGR->gray, pyrDown, Scharr.  gfx950.
//
// All three are HBM-streaming integer kernels (no MFMA).  Layout: images of a batch live at
// base + b*stride (stride 256-B aligned), rows tight (pitch = w).
//   gray    : 16 px / thread, 3x dwordx4 loads + 1x dwordx4 store, two v_dot4_u32_u8 per pixel.
//   pyrDown : k_pyr_down_stream (widths divisible by 8): one wave per 496-column strip marching down the rows,
//             packed-u16 5-tap filter in registers, no LDS;
//             k_pyr_down (other widths): 64x16 output tile / 256-thread block; source tile (136x35) staged into LDS
//             with dword loads, separable 5-tap pass through a u16 LDS intermediate.
//   scharr  : 64x16 output tile, 68x18 source tile in LDS, int16x2 (4 B/px) coalesced stores.
#include "ofk_internal.h"

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

// ------------------------------------------------------------------------------------------------ gray
__device__ __forceinline__ unsigned gray1(unsigned b, unsigned g, unsigned r)
{
    return (b * 3735u + g * 19235u + r * 9798u + 16384u) >> 15;
}

__global__ __launch_bounds__(256) void k_gray_bgr8(const uint8_t *__restrict__ bgr, size_t bgr_stride,
                                                   uint8_t *__restrict__ gray, size_t gray_stride, int npx)
{
    const int b = blockIdx.y;
    const uint8_t *src = bgr + (size_t)b * bgr_stride;
    uint8_t *dst = gray + (size_t)b * gray_stride;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;       // group of 16 pixels
    const int p0 = t * 16;
    if (p0 >= npx) return;
    if (p0 + 16 <= npx) {
        // 2*(3735 b + 19235 g + 9798 r) + 32768 = 256 * dot(px, HI) + dot(px, LO) + 32768 with byte-sized coefficients
        // (7470 = 29*256 + 46, 38470 = 150*256 + 70, 19596 = 76*256 + 140): two v_dot4_u32_u8 per pixel, and the gray value
        // (sum >> 16; the sum stays below 2^24) is byte 2 of the result — v_perm gathers four of them into the output dword.
        // One v_alignbit brings the pixel's three bytes to the bottom of a dword (the fourth byte meets coefficient 0); a pixel that
        // sits in bytes 1..3 of one dword (every fourth) meets coefficients shifted by one byte instead.
        constexpr unsigned HI = 29u | (150u << 8) | (76u << 16), LO = 46u | (70u << 8) | (140u << 16);
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src + (size_t)p0 * 3);
        const uint4 a = s4[0], c = s4[1], d = s4[2];
        const unsigned wv[13] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w, 0u};
        unsigned t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int byte = 3 * k;
            if ((byte & 3) == 1) {                              // bytes 1..3 of one dword: the coefficients move, not the pixel
                const unsigned px = wv[byte >> 2];
                t[k] = (__builtin_amdgcn_udot4(px, HI << 8, 0u, false) << 8) + __builtin_amdgcn_udot4(px, LO << 8, 32768u, false);
            } else {
                const unsigned px = (byte & 3) ? __builtin_amdgcn_alignbit(wv[(byte >> 2) + 1], wv[byte >> 2], (byte & 3) * 8) : wv[byte >> 2];
                t[k] = (__builtin_amdgcn_udot4(px, HI, 0u, false) << 8) + __builtin_amdgcn_udot4(px, LO, 32768u, false);
            }
        }
        unsigned out[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned p01 = __builtin_amdgcn_perm(t[4 * q + 1], t[4 * q], 0x0c0c0602u);      // (t0.b2, t1.b2, 0, 0)
            const unsigned p23 = __builtin_amdgcn_perm(t[4 * q + 3], t[4 * q + 2], 0x06020c0cu);  // (0, 0, t2.b2, t3.b2)
            out[q] = p01 | p23;
        }
        *reinterpret_cast<uint4 *>(dst + p0) = make_uint4(out[0], out[1], out[2], out[3]);
    } else {
        for (int p = p0; p < npx; ++p) dst[p] = (uint8_t)gray1(src[3 * (size_t)p], src[3 * (size_t)p + 1], src[3 * (size_t)p + 2]);
    }
}

// experiment (ofk_set_tuning "gray_px" = 16 / 32 / 64 -> one-WAVE workgroups of NPX pixels per thread, all loads of the thread in
// flight before the first use): a single wave fits the hole one retiring response wave leaves, whatever its register count
template <int NPX>
__global__ __launch_bounds__(64) void k_gray_bgr8_wide(const uint8_t *__restrict__ bgr, size_t bgr_stride, uint8_t *__restrict__ gray, size_t gray_stride, int npx)
{
    constexpr unsigned HI = 29u | (150u << 8) | (76u << 16), LO = 46u | (70u << 8) | (140u << 16);
    const int b = blockIdx.y;
    const uint8_t *src = bgr + (size_t)b * bgr_stride;
    uint8_t *dst = gray + (size_t)b * gray_stride;
    const int p0 = min((int)(blockIdx.x * blockDim.x + threadIdx.x) * NPX, npx - NPX);    // the last thread overlaps its neighbour (npx % 16 == 0, npx >= NPX)
    const uint4 *s4 = reinterpret_cast<const uint4 *>(src + (size_t)p0 * 3);
    uint4 v[NPX * 3 / 16];
#pragma unroll
    for (int i = 0; i < NPX * 3 / 16; ++i) v[i] = s4[i];
#pragma unroll
    for (int g = 0; g < NPX / 16; ++g) {
        const uint4 a = v[3 * g], c = v[3 * g + 1], d = v[3 * g + 2];
        const unsigned wv[13] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w, 0u};
        unsigned t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int byte = 3 * k;
            if ((byte & 3) == 1) {                              // bytes 1..3 of one dword: the coefficients move, not the pixel
                const unsigned px = wv[byte >> 2];
                t[k] = (__builtin_amdgcn_udot4(px, HI << 8, 0u, false) << 8) + __builtin_amdgcn_udot4(px, LO << 8, 32768u, false);
            } else {
                const unsigned px = (byte & 3) ? __builtin_amdgcn_alignbit(wv[(byte >> 2) + 1], wv[byte >> 2], (byte & 3) * 8) : wv[byte >> 2];
                t[k] = (__builtin_amdgcn_udot4(px, HI, 0u, false) << 8) + __builtin_amdgcn_udot4(px, LO, 32768u, false);
            }
        }
        unsigned out[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
            out[q] = __builtin_amdgcn_perm(t[4 * q + 1], t[4 * q], 0x0c0c0602u) | __builtin_amdgcn_perm(t[4 * q + 3], t[4 * q + 2], 0x06020c0cu);
        *reinterpret_cast<uint4 *>(dst + p0 + 16 * g) = make_uint4(out[0], out[1], out[2], out[3]);
    }
}

void ofk_launch_gray(hipStream_t s, const uint8_t *bgr, size_t bgr_stride, uint8_t *gray, size_t gray_stride, int batch,
                     int h, int w)
{
    const int npx = h * w;
    const int wide = g_ofk_tuning.gray_px;
    if ((wide == 16 || wide == 32 || wide == 64) && (npx & 15) == 0 && npx >= 64 && (bgr_stride & 15) == 0 && (gray_stride & 15) == 0) {
        const int threads = (npx + wide - 1) / wide;
        dim3 grid((threads + 63) / 64, batch);
        if (wide == 16) hipLaunchKernelGGL((k_gray_bgr8_wide<16>), grid, dim3(64), 0, s, bgr, bgr_stride, gray, gray_stride, npx);
        else if (wide == 32) hipLaunchKernelGGL((k_gray_bgr8_wide<32>), grid, dim3(64), 0, s, bgr, bgr_stride, gray, gray_stride, npx);
        else hipLaunchKernelGGL((k_gray_bgr8_wide<64>), grid, dim3(64), 0, s, bgr, bgr_stride, gray, gray_stride, npx);
        return;
    }
    const int groups = (npx + 15) / 16;
    dim3 grid((groups + 255) / 256, batch);
    hipLaunchKernelGGL(k_gray_bgr8, grid, dim3(256), 0, s, bgr, bgr_stride, gray, gray_stride, npx);
}

// ------------------------------------------------------------------------------------------------ pyrDown
#define PD_TW 64
#define PD_TH 16
#define PD_SW 136                       // staged source columns: [2*ox0-4, 2*ox0+132)
#define PD_SH (2 * PD_TH + 3)           // staged source rows:    [2*oy0-2, 2*oy0+2*TH+1)

// z < batch: image z of (src0 -> dst0); z >= batch: image z - batch of (src1 -> dst1) — both frames of a pair in one launch.
__global__ __launch_bounds__(256) void k_pyr_down(const uint8_t *__restrict__ src0, const uint8_t *__restrict__ src1,
                                                  size_t src_stride, int h, int w, uint8_t *__restrict__ dst0,
                                                  uint8_t *__restrict__ dst1, size_t dst_stride, int dh, int dw, int batch)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_src[PD_SH][PD_SW];
    __shared__ __attribute__((aligned(16))) uint16_t s_h[PD_SH][PD_TW];
    const int z = blockIdx.z;
    const uint8_t *img = z < batch ? src0 + (size_t)z * src_stride : src1 + (size_t)(z - batch) * src_stride;
    uint8_t *out = z < batch ? dst0 + (size_t)z * dst_stride : dst1 + (size_t)(z - batch) * dst_stride;
    const int ox0 = blockIdx.x * PD_TW, oy0 = blockIdx.y * PD_TH;
    const int sx0 = 2 * ox0 - 4, sy0 = 2 * oy0 - 2;
    const int tid = threadIdx.x;
    if ((w & 3) == 0 && w >= 8) {
        // Dword staging for every tile: rows are mirrored per row, out-of-range dwords are clamped (never faulting) and
        // the two mirrored columns a border tile really needs (-2,-1 / w,w+1) are patched from LDS afterwards.
        // All of this thread's loads are issued before the first LDS store (one memory latency per block, not five).
        constexpr int ND = PD_SH * (PD_SW / 4), NIT = (ND + 255) / 256;
        unsigned v[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = min(tid + 256 * k, ND - 1);
            const int r = i / (PD_SW / 4), c4 = i % (PD_SW / 4);
            const int sc = min(max(sx0 + 4 * c4, 0), w - 4);
            v[k] = *reinterpret_cast<const unsigned *>(img + (size_t)reflect101(sy0 + r, h) * w + sc);
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int i = tid + 256 * k;
            if (i < ND) reinterpret_cast<unsigned *>(&s_src[0][0])[i] = v[k];
        }
        if (sx0 < 0 || sx0 + PD_SW > w) {                       // block-uniform
            __syncthreads();
            if (tid < PD_SH) {
                if (sx0 < 0) { s_src[tid][2] = s_src[tid][6]; s_src[tid][3] = s_src[tid][5]; }          // cols -2,-1 <- 2,1
                const int c = w - sx0;                          // staged index of source column w
                if (c >= 3 && c < PD_SW) s_src[tid][c] = s_src[tid][c - 2];                              // w   <- w-2
                if (c >= 3 && c + 1 < PD_SW) s_src[tid][c + 1] = s_src[tid][c - 3];                      // w+1 <- w-3
            }
        }
    } else {
        for (int i = tid; i < PD_SH * PD_SW; i += 256) {
            const int r = i / PD_SW, c = i % PD_SW;
            s_src[r][c] = img[(size_t)reflect101(sy0 + r, h) * w + reflect101(sx0 + c, w)];
        }
    }
    __syncthreads();
    // horizontal [1 4 6 4 1]: item = (staged row, 4 outputs); 16 source bytes come in as two ds_read_b64, 4 u16 go out as one
    for (int i = tid; i < PD_SH * (PD_TW / 4); i += 256) {
        const int r = i >> 4, q = i & 15;
        const uint2 lo = *reinterpret_cast<const uint2 *>(&s_src[r][8 * q]);
        const uint2 hi = *reinterpret_cast<const uint2 *>(&s_src[r][8 * q + 8]);
        const unsigned d[4] = {lo.x, lo.y, hi.x, hi.y};
        unsigned o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {                       // output 4q+j reads staged bytes 8q + 2j + 2 .. + 6
            unsigned acc = 0;
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const int byte = 2 * j + 2 + k;
                const unsigned v = (d[byte >> 2] >> (8 * (byte & 3))) & 255u;
                acc += (k == 0 || k == 4) ? v : (k == 2 ? 6u * v : 4u * v);
            }
            o[j] = acc;
        }
        *reinterpret_cast<uint2 *>(&s_h[r][4 * q]) = make_uint2(o[0] | (o[1] << 16), o[2] | (o[3] << 16));
    }
    __syncthreads();
    // vertical: item = (output row, 4 outputs): five ds_read_b64, one dword store
    {
        const int y = tid >> 4, q = tid & 15;
        const int oy = oy0 + y, ox = ox0 + 4 * q;
        if (oy < dh && ox < dw) {
            unsigned acc[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 5; ++k) {
                const uint2 v = *reinterpret_cast<const uint2 *>(&s_h[2 * y + k][4 * q]);
                const unsigned wgt = (k == 0 || k == 4) ? 1u : (k == 2 ? 6u : 4u);
                acc[0] += wgt * (v.x & 0xffffu); acc[1] += wgt * (v.x >> 16);
                acc[2] += wgt * (v.y & 0xffffu); acc[3] += wgt * (v.y >> 16);
            }
            const unsigned pk = ((acc[0] + 128) >> 8) | (((acc[1] + 128) >> 8) << 8) | (((acc[2] + 128) >> 8) << 16) |
                                (((acc[3] + 128) >> 8) << 24);
            uint8_t *dp = out + (size_t)oy * dw + ox;
            if (ox + 3 < dw && (dw & 3) == 0) *reinterpret_cast<unsigned *>(dp) = pk;
            else
                for (int j = 0; j < 4 && ox + j < dw; ++j) dp[j] = (uint8_t)(pk >> (8 * j));
        }
    }
}

// Streaming pyrDown (widths that are multiples of 8): one WAVE per strip of 496 source columns, marching down the rows —
// no LDS, no barriers, every source byte fetched once per strip.
//   lane = 8 source columns (one dwordx2 load per source row) = 4 output columns (one dword store per output row); lanes
//   0 and 63 only feed their neighbours (the 2-column halo travels through DPP wave shifts), so a strip yields 248 outputs.
//   The horizontal [1 4 6 4 1] runs on PAIRS of u16 packed in a dword (even bytes / odd bytes of the loaded dwords, the
//   shifted pairs through v_alignbit): a horizontal sum is <= 4080 and a vertical one <= 65280 + 128, so the two halves
//   never carry into each other and plain 32-bit adds are exact.  The vertical pass keeps the last three horizontal rows in
//   registers and consumes two new ones per output row; v_perm picks the rounded high bytes into the output dword.
#define PDS_COLS 496                    // source columns per strip (62 lanes x 8)
__device__ __forceinline__ void pds_hrow(uint2 d, bool first, bool last, unsigned &h01, unsigned &h23)
{
    const unsigned A = d.x & 0x00ff00ffu, C = (d.x >> 8) & 0x00ff00ffu;      // (b0,b2) (b1,b3)
    const unsigned B = d.y & 0x00ff00ffu, D = (d.y >> 8) & 0x00ff00ffu;      // (b4,b6) (b5,b7)
    unsigned pB = (unsigned)__builtin_amdgcn_update_dpp(0, (int)B, 0x138, 0xF, 0xF, true);   // lane-1: (.., b6) of the 8 columns to the left
    unsigned pD = (unsigned)__builtin_amdgcn_update_dpp(0, (int)D, 0x138, 0xF, 0xF, true);   //         (.., b7)
    unsigned nA = (unsigned)__builtin_amdgcn_update_dpp(0, (int)A, 0x130, 0xF, 0xF, true);   // lane+1: (b0, ..) of the 8 columns to the right
    if (first) { pB = A; pD = C << 16; }                        // columns -2,-1 mirror 2,1
    if (last) nA = B >> 16;                                     // column w mirrors w-2
    const unsigned le01 = __builtin_amdgcn_alignbit(A, pB, 16);  // (p6,b0)
    const unsigned lo01 = __builtin_amdgcn_alignbit(C, pD, 16);  // (p7,b1)
    const unsigned re01 = __builtin_amdgcn_alignbit(B, A, 16);   // (b2,b4)
    const unsigned lo23 = __builtin_amdgcn_alignbit(D, C, 16);   // (b3,b5)
    const unsigned re23 = __builtin_amdgcn_alignbit(nA, B, 16);  // (b6,n0)
    h01 = 6u * A + 4u * (lo01 + C) + le01 + re01;               // centres b0, b2
    h23 = 6u * B + 4u * (lo23 + D) + re01 + re23;               // centres b4, b6
}

__global__ __launch_bounds__(256) void k_pyr_down_stream(const uint8_t *__restrict__ src0, const uint8_t *__restrict__ src1,
                                                         size_t src_stride, int h, int w, uint8_t *__restrict__ dst0,
                                                         uint8_t *__restrict__ dst1, size_t dst_stride, int dh, int dw, int batch,
                                                         int rows, int nstrips, int nchunks)
{
    const int lane = threadIdx.x & 63;
    // XCD-aware block -> (image, blocks of the image) map: workgroups are dealt round-robin over the 8 XCDs (blocks n and
    // n + 8 share one L2), so an image's row chunks go to ONE XCD and the halo rows two chunks share are fetched once.
    int bx = blockIdx.x, z = blockIdx.y;
    if ((gridDim.y & 7) == 0) {
        const unsigned n = blockIdx.y * gridDim.x + blockIdx.x, k = n >> 3;
        z = 8 * (int)(k / gridDim.x) + (int)(n & 7);
        bx = (int)(k % gridDim.x);
    }
    const int wid = __builtin_amdgcn_readfirstlane(bx * 4 + (threadIdx.x >> 6));   // wave-uniform: row math stays scalar
    if (wid >= nstrips * nchunks) return;                       // whole wave
    const int strip = wid % nstrips, chunk = wid / nstrips;
    const uint8_t *img = z < batch ? src0 + (size_t)z * src_stride : src1 + (size_t)(z - batch) * src_stride;
    uint8_t *out = z < batch ? dst0 + (size_t)z * dst_stride : dst1 + (size_t)(z - batch) * dst_stride;
    const int c0 = strip * PDS_COLS - 8 + 8 * lane;             // first source column of this lane
    const int sc = min(max(c0, 0), w - 8);                      // clamped: lanes outside the image load something valid
    const bool first = c0 == 0, last = c0 == w - 8;
    const int ox = strip * (PDS_COLS / 2) - 4 + 4 * lane;
    const bool store_ok = lane >= 1 && lane <= 62 && ox < dw;
    const int oy0 = chunk * rows, oy1 = min(dh, oy0 + rows);
    // one branch-free reflection covers every row that is used (h >= 4: -2 -> 2, h+1 -> h-3); the clamp only catches the
    // prefetch running past the chunk, whose values are never consumed
    auto ld = [&](int sy) -> uint2 {
        sy = sy < 0 ? -sy : sy;
        sy = sy >= h ? 2 * (h - 1) - sy : sy;
        const uint8_t *rowp = img + (size_t)__builtin_amdgcn_readfirstlane(max(sy, 0)) * (unsigned)w;   // scalar base + lane offset
        return *reinterpret_cast<const uint2 *>(rowp + (unsigned)sc);
    };
    unsigned a01, a23, b01, b23, c01, c23;
    {
        const uint2 r0 = ld(2 * oy0 - 2), r1 = ld(2 * oy0 - 1), r2 = ld(2 * oy0);
        pds_hrow(r0, first, last, a01, a23);
        pds_hrow(r1, first, last, b01, b23);
        pds_hrow(r2, first, last, c01, c23);
    }
    uint2 q0 = ld(2 * oy0 + 1), q1 = ld(2 * oy0 + 2), q2 = ld(2 * oy0 + 3), q3 = ld(2 * oy0 + 4);
    uint2 q4 = ld(2 * oy0 + 5), q5 = ld(2 * oy0 + 6);
    auto step = [&](uint2 u0, uint2 u1, int oy) {
        unsigned d01, d23, e01, e23;
        pds_hrow(u0, first, last, d01, d23);
        pds_hrow(u1, first, last, e01, e23);
        // a + e + 4(b + d) + 6c + 128 as shifts and adds (a 32-bit multiply by 6 would not fit v_mul_u32_u24)
        const unsigned v01 = ((b01 + d01 + c01) << 2) + a01 + ((c01 << 1) + e01) + 0x00800080u;
        const unsigned v23 = ((b23 + d23 + c23) << 2) + a23 + ((c23 << 1) + e23) + 0x00800080u;
        const unsigned pk = __builtin_amdgcn_perm(v23, v01, 0x07050301u);      // (v >> 8) of the four u16 halves
        if (store_ok && oy < oy1) *reinterpret_cast<unsigned *>(out + (size_t)oy * dw + ox) = pk;
        a01 = c01; a23 = c23; b01 = d01; b23 = d23; c01 = e01; c23 = e23;
    };
    for (int oy = oy0; oy < oy1; oy += 3) {                     // six rows in flight per wave; unrolled so the queue renames
        uint2 u0 = q0, u1 = q1;
        q0 = ld(2 * oy + 7); q1 = ld(2 * oy + 8);
        step(u0, u1, oy);
        u0 = q2; u1 = q3;
        q2 = ld(2 * oy + 9); q3 = ld(2 * oy + 10);
        step(u0, u1, oy + 1);
        u0 = q4; u1 = q5;
        q4 = ld(2 * oy + 11); q5 = ld(2 * oy + 12);
        step(u0, u1, oy + 2);
    }
}

// ------------------------------------------------------------------------------------------------ three pyramid levels in one pass
// k_pyr3_stream: levels 1, 2 and 3 from one read of level 0 (widths divisible by 16, heights by 8).  The per-level kernels read
// level 1 and level 2 back from HBM (1.3 of the 6.8 MB per pair); here a wave marches down a strip of level 0 once and every
// level's rows are built from the rows of the level above while those are still in registers.
//   lane = 16 source columns (one dwordx4 load per row) = 8 / 4 / 2 columns of levels 1 / 2 / 3.  Every level's horizontal
//   [1 4 6 4 1] needs one value pair from each neighbouring lane (DPP wave shifts), so three lanes on either side of a strip
//   only feed their neighbours: 58 payload lanes between strips, 61 in a strip that touches an image border (the mirrored
//   columns come from the lane itself).  1920 columns = 120 lanes = two strips.
//   Vertically, step t consumes level-0 rows 2t+1, 2t+2 and yields level-1 row t; at even t level-2 row (t-2)/2 follows from
//   the last five horizontally filtered level-1 rows, at t = 2 (mod 4) level-3 row (t-6)/4 from the last five level-2 rows.
//   A chunk of steps starts 12 steps early (its windows fill with the rows of the chunk above - or, at the top of the image,
//   with rows of negative index, which equal their mirror images at every level because filter and extension are symmetric);
//   at the bottom (even heights at every level) row n of a level mirrors row n-2, the centre row of the last window.
//   Arithmetic per level is exactly that of k_pyr_down_stream: packed u16 pairs, (sum + 128) >> 8.
#define P3_M 0x00ff00ffu
#define P3_SHR(x) ((unsigned)__builtin_amdgcn_update_dpp(0, (int)(x), 0x138, 0xF, 0xF, true))   // value of lane-1
#define P3_SHL(x) ((unsigned)__builtin_amdgcn_update_dpp(0, (int)(x), 0x130, 0xF, 0xF, true))   // value of lane+1
#define P3_AB(hi, lo) __builtin_amdgcn_alignbit((hi), (lo), 16)                                   // (lo.hi16, hi.lo16)
#define P3_EVEN(t1, t0) __builtin_amdgcn_perm((t1), (t0), 0x05040100u)                            // (t0.lo16, t1.lo16)
#define P3_ODD(t1, t0) __builtin_amdgcn_perm((t1), (t0), 0x07060302u)                             // (t0.hi16, t1.hi16)
#define P3_BYTES(v1, v0) __builtin_amdgcn_perm((v1), (v0), 0x07050301u)                           // (v >> 8) of the four u16 halves
#define P3_ODDB(d) __builtin_amdgcn_perm(0u, (d), 0x0c030c01u)                                    // (d >> 8) & 0x00ff00ff in one instruction
// value of lane-1; lane 0 of the wave, which has no lane-1, keeps `old` (bound_ctrl off).  In the strip that starts at the image's left
// edge lane 0 IS the lane whose columns -2, -1 mirror 2, 1, so the mirrored pair goes in as `old` and no select is needed; in the other
// strips lane 0 is a halo lane whose values never reach a stored column.
#define P3_SHR_OLD(old, x) ((unsigned)__builtin_amdgcn_update_dpp((int)(old), (int)(x), 0x138, 0xF, 0xF, false))

__device__ __forceinline__ unsigned p3_vert(unsigned a, unsigned b, unsigned c, unsigned d, unsigned e)
{
    return ((b + d + c) << 2) + a + ((c << 1) + e) + 0x00800080u;       // a + e + 4(b + d) + 6c + 128 in both halves
}

// 16 bytes of a level-0 row -> the four pairs of horizontal sums centred on bytes (0,2) (4,6) (8,10) (12,14)
__device__ __forceinline__ void p3_h16(uint4 d, bool first, bool last, unsigned *h)
{
    const unsigned E0 = d.x & P3_M, O0 = P3_ODDB(d.x), E1 = d.y & P3_M, O1 = P3_ODDB(d.y);
    const unsigned E2 = d.z & P3_M, O2 = P3_ODDB(d.z), E3 = d.w & P3_M, O3 = P3_ODDB(d.w);
    const unsigned pE = P3_SHR_OLD(E0, E3), pO = P3_SHR_OLD(O0 << 16, O3);      // columns -2,-1 mirror 2,1 (lane 0 of the first strip)
    unsigned nE = P3_SHL(E0);
    (void)first;
    if (last) nE = E3 >> 16;                                    // column w mirrors w-2
    const unsigned l20 = P3_AB(E0, pE), l21 = P3_AB(E1, E0), l22 = P3_AB(E2, E1), l23 = P3_AB(E3, E2), l24 = P3_AB(nE, E3);
    const unsigned l10 = P3_AB(O0, pO), l11 = P3_AB(O1, O0), l12 = P3_AB(O2, O1), l13 = P3_AB(O3, O2);
    h[0] = 6u * E0 + 4u * (l10 + O0) + l20 + l21;
    h[1] = 6u * E1 + 4u * (l11 + O1) + l21 + l22;
    h[2] = 6u * E2 + 4u * (l12 + O2) + l22 + l23;
    h[3] = 6u * E3 + 4u * (l13 + O3) + l23 + l24;
}

// 8 values of a level-1 row as even/odd pairs (E0 = (o0,o2), O0 = (o1,o3), E1 = (o4,o6), O1 = (o5,o7)) -> two pairs of sums
__device__ __forceinline__ void p3_h8(unsigned E0, unsigned O0, unsigned E1, unsigned O1, bool first, bool last, unsigned &g0, unsigned &g1)
{
    const unsigned pE = P3_SHR_OLD(E0, E1), pO = P3_SHR_OLD(O0 << 16, O1);
    unsigned nE = P3_SHL(E0);
    (void)first;
    if (last) nE = E1 >> 16;
    const unsigned l20 = P3_AB(E0, pE), l21 = P3_AB(E1, E0), l22 = P3_AB(nE, E1), l10 = P3_AB(O0, pO), l11 = P3_AB(O1, O0);
    g0 = 6u * E0 + 4u * (l10 + O0) + l20 + l21;
    g1 = 6u * E1 + 4u * (l11 + O1) + l21 + l22;
}

// 4 values of a level-2 row (E = (t0,t2), O = (t1,t3)) -> one pair of sums
__device__ __forceinline__ unsigned p3_h4(unsigned E, unsigned O, bool first, bool last)
{
    const unsigned pE = P3_SHR_OLD(E, E), pO = P3_SHR_OLD(O << 16, O);
    unsigned nE = P3_SHL(E);
    (void)first;
    if (last) nE = E >> 16;
    return 6u * E + 4u * (P3_AB(O, pO) + O) + P3_AB(E, pE) + P3_AB(nE, E);
}

struct p3_args {
    const uint8_t *base0, *base1;       // pyramid slabs of the two frame sets (level 0 at offset 0)
    size_t stride, off1, off2, off3;    // bytes between images; offsets of levels 1..3 inside a slab
    int h, w, batch, steps, nstrips, nchunks;
};

// One WAVE per workgroup: beside the response kernel (3 x 168 VGPRs per SIMD) a 256-thread workgroup has to wait until all four
// SIMDs of a CU have 96 registers free at the same moment; a single wave moves in as soon as one response wave retires
// (2.28 -> 2.22 ms per step at B = 256; alone on the chip the pass is 10 % slower — tools/experiments/hbm_wg.sh).
__global__ __launch_bounds__(64) void k_pyr3_stream(p3_args A)
{
    const int lane = threadIdx.x;
    int bx = blockIdx.x, z = blockIdx.y;
    if ((gridDim.y & 7) == 0) {                                 // XCD-aware: an image's chunks on one XCD (see k_pyr_down_stream)
        const unsigned n = blockIdx.y * gridDim.x + blockIdx.x, k = n >> 3;
        z = 8 * (int)(k / gridDim.x) + (int)(n & 7);
        bx = (int)(k % gridDim.x);
    }
    const int wid = bx;
    const int strip = wid % A.nstrips, chunk = wid / A.nstrips;
    const int h = A.h, w = A.w, n1 = h >> 1, n2 = h >> 2, n3 = h >> 3, w1 = w >> 1, w2 = w >> 2, w3 = w >> 3;
    const uint8_t *img = (z < A.batch ? A.base0 + (size_t)z * A.stride : A.base1 + (size_t)(z - A.batch) * A.stride);
    uint8_t *slab = const_cast<uint8_t *>(img);
    const int hl = strip == 0 ? 0 : 3, hr = strip == A.nstrips - 1 ? 0 : 3;
    const int cs = strip == 0 ? 0 : 16 * (61 + 58 * (strip - 1));           // first payload column of the strip
    const int c0 = cs - 16 * hl + 16 * lane;
    const int sc = min(max(c0, 0), w - 16);
    const bool first = c0 == 0, last = c0 == w - 16;
    const bool st = lane >= hl && lane <= 63 - hr && c0 < w;
    const int T0 = chunk * A.steps, T1 = T0 + A.steps;
    auto ld = [&](int sy) -> uint4 {
        sy = sy < 0 ? -sy : sy;
        sy = sy >= h ? 2 * (h - 1) - sy : sy;
        const uint8_t *rowp = img + (size_t)__builtin_amdgcn_readfirstlane(max(sy, 0)) * (unsigned)w;
        return *reinterpret_cast<const uint4 *>(rowp + (unsigned)sc);
    };
    const int ts = T0 - 12;
    unsigned a[4], b[4], c[4];
    p3_h16(ld(2 * ts - 2), first, last, a);
    p3_h16(ld(2 * ts - 1), first, last, b);
    p3_h16(ld(2 * ts), first, last, c);
    uint4 q[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) q[k] = ld(2 * ts + 1 + k);
    unsigned r1x[5] = {0, 0, 0, 0, 0}, r1y[5] = {0, 0, 0, 0, 0}, r2[5] = {0, 0, 0, 0, 0};
    for (int t = ts; t < T1; t += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int tt = t + k;
            const bool own = tt >= T0;                          // warm-up steps store nothing
            unsigned d[4], e[4], v[4];
            p3_h16(q[2 * k], first, last, d);
            p3_h16(q[2 * k + 1], first, last, e);
            q[2 * k] = ld(2 * tt + 9);
            q[2 * k + 1] = ld(2 * tt + 10);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = p3_vert(a[j], b[j], c[j], d[j], e[j]); a[j] = c[j]; b[j] = d[j]; c[j] = e[j]; }
            if (st && own && tt < n1)
                *reinterpret_cast<uint2 *>(slab + A.off1 + (size_t)tt * w1 + (c0 >> 1)) = make_uint2(P3_BYTES(v[1], v[0]), P3_BYTES(v[3], v[2]));
            // level 1 -> horizontally filtered row of level 1
            const unsigned t0 = P3_ODDB(v[0]), t1 = P3_ODDB(v[1]), t2 = P3_ODDB(v[2]), t3 = P3_ODDB(v[3]);
            unsigned g0, g1;
            p3_h8(P3_EVEN(t1, t0), P3_ODD(t1, t0), P3_EVEN(t3, t2), P3_ODD(t3, t2), first, last, g0, g1);
#pragma unroll
            for (int j = 0; j < 4; ++j) { r1x[j] = r1x[j + 1]; r1y[j] = r1y[j + 1]; }
            r1x[4] = g0; r1y[4] = g1;
            if ((k & 1) == 0) {                                 // even step: level-2 row (tt - 2) / 2 from level-1 rows tt-4 .. tt
                const bool m1 = tt == n1;                       // row n1 mirrors row n1 - 2, the centre of this window
                const unsigned w0 = p3_vert(r1x[0], r1x[1], r1x[2], r1x[3], m1 ? r1x[2] : r1x[4]);
                const unsigned w1v = p3_vert(r1y[0], r1y[1], r1y[2], r1y[3], m1 ? r1y[2] : r1y[4]);
                const int qrow = (tt - 2) >> 1;
                if (st && own && tt >= 2 && qrow < n2)
                    *reinterpret_cast<unsigned *>(slab + A.off2 + (size_t)qrow * w2 + (c0 >> 2)) = P3_BYTES(w1v, w0);
                const unsigned s0 = P3_ODDB(w0), s1 = P3_ODDB(w1v);
                const unsigned k2 = p3_h4(P3_EVEN(s1, s0), P3_ODD(s1, s0), first, last);
#pragma unroll
                for (int j = 0; j < 4; ++j) r2[j] = r2[j + 1];
                r2[4] = k2;
                if (k == 2) {                                   // step 2 (mod 4): level-3 row (tt - 6) / 4 from level-2 rows q-4 .. q
                    const unsigned zv = p3_vert(r2[0], r2[1], r2[2], r2[3], qrow == n2 ? r2[2] : r2[4]);
                    const int rrow = (tt - 6) >> 2;
                    if (st && own && tt >= 6 && rrow < n3)
                        *reinterpret_cast<unsigned short *>(slab + A.off3 + (size_t)rrow * w3 + (c0 >> 3)) =
                            (unsigned short)(((zv >> 8) & 0xffu) | ((zv >> 16) & 0xff00u));
                }
            }
        }
    }
}

static bool pyr3_ok(int h, int w) { return (w & 15) == 0 && (h & 7) == 0 && w >= 64 && h >= 32 && !g_ofk_tuning.no_pyr3; }

// levels 1..3 of both frame sets from level 0; false if the geometry does not fit (the caller then goes level by level)
bool ofk_launch_pyr3(hipStream_t s, uint8_t *pyr0, uint8_t *pyr1, size_t stride, const ofk_levels &lv, int batch, int images)
{
    const int h = lv.h[0], w = lv.w[0];
    if (lv.n < 3 || !pyr3_ok(h, w)) return false;
    const int lanes = w / 16;
    int nstrips = 1;
    if (lanes > 64) { nstrips = 2; while (61 * 2 + 58 * (nstrips - 2) < lanes) ++nstrips; }
    const int total = ((h >> 1) + 3 + 3) / 4 * 4;               // steps 0 .. n1 + 2, rounded up to whole groups of four
    // >= 2048 waves x 8 KB of loads in flight cover the HBM latency and more chunks only add warm-up rows ALONE on the chip (B = 256:
    // 2 chunks 71 %, 4 chunks 69 %, 8 chunks 61 % of the HBM roof) - but beside the response kernel shorter-lived waves find room more
    // often: B = 512 pairs, 1 / 4 / 8 / 16 chunks -> 123.5-125.4 / 126.8-128.0 / 126.4-127.5 / 124.4 k pairs/s (round 3).  Target 8192 waves.
    int nchunks = (8192 + images * nstrips - 1) / (images * nstrips);
    const int maxchunks = total / 32 > 1 ? total / 32 : 1;                 // chunks of at least 32 steps (12 warm-up steps each)
    nchunks = nchunks < 1 ? 1 : nchunks > maxchunks ? maxchunks : nchunks;
    if (g_ofk_tuning.pyr3_chunks >= 1 && g_ofk_tuning.pyr3_chunks <= maxchunks) nchunks = g_ofk_tuning.pyr3_chunks;   // ofk_set_tuning("pyr3_chunks")
    int steps = ((total + nchunks - 1) / nchunks + 3) / 4 * 4;
    nchunks = (total + steps - 1) / steps;
    p3_args A = {pyr0, pyr1 ? pyr1 : pyr0, stride, lv.off[1], lv.off[2], lv.off[3], h, w, batch, steps, nstrips, nchunks};
    dim3 grid(nstrips * nchunks, images);
    hipLaunchKernelGGL(k_pyr3_stream, grid, dim3(64), 0, s, A);
    return true;
}

static bool pyr_stream_ok(int h, int w) { return (w & 7) == 0 && w >= 16 && h >= 4; }

static void launch_pyr_stream(hipStream_t s, const uint8_t *src0, const uint8_t *src1, size_t src_stride, int h, int w,
                              uint8_t *dst0, uint8_t *dst1, size_t dst_stride, int batch, int images)
{
    const int dh = (h + 1) / 2, dw = (w + 1) / 2;
    const int nstrips = (w + PDS_COLS - 1) / PDS_COLS;
    int rows = 32;                                              // long strips re-read fewer halo rows; shorten them until the
    while (rows > 4 && (long long)nstrips * ((dh + rows - 1) / rows) * images < 8192) rows >>= 1;   // launch has >= 8 waves per SIMD
    if (g_ofk_tuning.pyr_rows >= 1) rows = g_ofk_tuning.pyr_rows;   // ofk_set_tuning("pyr_rows")
    const int nchunks = (dh + rows - 1) / rows;
    dim3 grid((nstrips * nchunks + 3) / 4, images);
    hipLaunchKernelGGL(k_pyr_down_stream, grid, dim3(256), 0, s, src0, src1, src_stride, h, w, dst0, dst1, dst_stride, dh, dw, batch,
                       rows, nstrips, nchunks);
}

void ofk_launch_pyr_down(hipStream_t s, const uint8_t *src, size_t src_stride, int h, int w, uint8_t *dst,
                         size_t dst_stride, int batch)
{
    if (pyr_stream_ok(h, w)) { launch_pyr_stream(s, src, nullptr, src_stride, h, w, dst, nullptr, dst_stride, batch, batch); return; }
    const int dh = (h + 1) / 2, dw = (w + 1) / 2;
    dim3 grid((dw + PD_TW - 1) / PD_TW, (dh + PD_TH - 1) / PD_TH, batch);
    hipLaunchKernelGGL(k_pyr_down, grid, dim3(256), 0, s, src, (const uint8_t *)nullptr, src_stride, h, w, dst, (uint8_t *)nullptr,
                       dst_stride, dh, dw, batch);
}

// the same level of both frames of every pair in one launch
void ofk_launch_pyr_down2(hipStream_t s, const uint8_t *src0, const uint8_t *src1, size_t src_stride, int h, int w,
                          uint8_t *dst0, uint8_t *dst1, size_t dst_stride, int batch)
{
    if (pyr_stream_ok(h, w)) { launch_pyr_stream(s, src0, src1, src_stride, h, w, dst0, dst1, dst_stride, batch, 2 * batch); return; }
    const int dh = (h + 1) / 2, dw = (w + 1) / 2;
    dim3 grid((dw + PD_TW - 1) / PD_TW, (dh + PD_TH - 1) / PD_TH, 2 * batch);
    hipLaunchKernelGGL(k_pyr_down, grid, dim3(256), 0, s, src0, src1, src_stride, h, w, dst0, dst1, dst_stride, dh, dw, batch);
}

// ------------------------------------------------------------------------------------------------ Scharr
#define SC_TW 64
#define SC_TH 16
#define SC_SW 72                        // staged columns [x0-4, x0+68)
#define SC_SH (SC_TH + 2)

__global__ __launch_bounds__(256) void k_scharr(const uint8_t *__restrict__ src, size_t src_stride, int h, int w,
                                                int16_t *__restrict__ dxdy, size_t dst_stride_elems)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_src[SC_SH][SC_SW];
    const int b = blockIdx.z;
    const uint8_t *img = src + (size_t)b * src_stride;
    int16_t *out = dxdy + (size_t)b * dst_stride_elems * 2;
    const int x0 = blockIdx.x * SC_TW, y0 = blockIdx.y * SC_TH;
    const int sx0 = x0 - 4, sy0 = y0 - 1;
    const int tid = threadIdx.x;
    const bool interior = sx0 >= 0 && sy0 >= 0 && sx0 + SC_SW <= w && sy0 + SC_SH <= h && (w & 3) == 0;
    if (interior) {
        for (int i = tid; i < SC_SH * (SC_SW / 4); i += 256) {
            const int r = i / (SC_SW / 4), c4 = i % (SC_SW / 4);
            *reinterpret_cast<unsigned *>(&s_src[r][c4 * 4]) =
                *reinterpret_cast<const unsigned *>(img + (size_t)(sy0 + r) * w + sx0 + c4 * 4);
        }
    } else {
        for (int i = tid; i < SC_SH * SC_SW; i += 256) {
            const int r = i / SC_SW, c = i % SC_SW;
            s_src[r][c] = img[(size_t)reflect101(sy0 + r, h) * w + reflect101(sx0 + c, w)];
        }
    }
    __syncthreads();
    for (int i = tid; i < SC_TH * SC_TW; i += 256) {
        const int y = i / SC_TW, x = i % SC_TW;
        const int gy = y0 + y, gx = x0 + x;
        if (gy < h && gx < w) {
            const uint8_t *r0 = &s_src[y][x + 3], *r1 = &s_src[y + 1][x + 3], *r2 = &s_src[y + 2][x + 3];
            const int dx = 3 * (r0[2] - r0[0]) + 10 * (r1[2] - r1[0]) + 3 * (r2[2] - r2[0]);
            const int dy = 3 * (r2[0] - r0[0]) + 10 * (r2[1] - r0[1]) + 3 * (r2[2] - r0[2]);
            const unsigned pk = ((unsigned)dx & 0xffffu) | ((unsigned)dy << 16);
            *reinterpret_cast<unsigned *>(out + 2 * ((size_t)gy * w + gx)) = pk;
        }
    }
}

void ofk_launch_scharr(hipStream_t s, const uint8_t *src, size_t src_stride, int h, int w, int16_t *dxdy,
                       size_t dst_stride_elems, int batch)
{
    dim3 grid((w + SC_TW - 1) / SC_TW, (h + SC_TH - 1) / SC_TH, batch);
    hipLaunchKernelGGL(k_scharr, grid, dim3(256), 0, s, src, src_stride, h, w, dxdy, dst_stride_elems);
}
