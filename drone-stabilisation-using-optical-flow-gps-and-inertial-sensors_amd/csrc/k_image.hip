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
        // One v_alignbit brings the pixel's three bytes to the bottom of a dword (the fourth byte meets coefficient 0).
        constexpr unsigned HI = 29u | (150u << 8) | (76u << 16), LO = 46u | (70u << 8) | (140u << 16);
        const uint4 *s4 = reinterpret_cast<const uint4 *>(src + (size_t)p0 * 3);
        const uint4 a = s4[0], c = s4[1], d = s4[2];
        const unsigned wv[13] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w, 0u};
        unsigned t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int byte = 3 * k;
            const unsigned px = (byte & 3) ? __builtin_amdgcn_alignbit(wv[(byte >> 2) + 1], wv[byte >> 2], (byte & 3) * 8) : wv[byte >> 2];
            t[k] = (__builtin_amdgcn_udot4(px, HI, 0u, false) << 8) + __builtin_amdgcn_udot4(px, LO, 32768u, false);
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

void ofk_launch_gray(hipStream_t s, const uint8_t *bgr, size_t bgr_stride, uint8_t *gray, size_t gray_stride, int batch,
                     int h, int w)
{
    const int npx = h * w;
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

static bool pyr_stream_ok(int h, int w) { return (w & 7) == 0 && w >= 16 && h >= 4; }

static void launch_pyr_stream(hipStream_t s, const uint8_t *src0, const uint8_t *src1, size_t src_stride, int h, int w,
                              uint8_t *dst0, uint8_t *dst1, size_t dst_stride, int batch, int images)
{
    const int dh = (h + 1) / 2, dw = (w + 1) / 2;
    const int nstrips = (w + PDS_COLS - 1) / PDS_COLS;
    int rows = 32;                                              // long strips re-read fewer halo rows; shorten them until the
    while (rows > 4 && (long long)nstrips * ((dh + rows - 1) / rows) * images < 8192) rows >>= 1;   // launch has >= 8 waves per SIMD
    if (const char *e = getenv("OFK_PYR_ROWS")) { const int v = atoi(e); if (v >= 1 && v <= 4096) rows = v; }   // tuning knob
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
