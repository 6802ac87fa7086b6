// k_corners.hip — Shi-Tomasi corner detection (cv2.goodFeaturesToTrack semantics).  gfx950.
//
//   k_mineig : gray tile (+halo) -> LDS; Sobel-3 -> int16 (dx,dy) in LDS; separable box sums of the three
//              products through an int32 LDS intermediate (sliding windows, all integer = exact);
//              lambda_min in f32 with a fixed operation order (-ffp-contract=off); coalesced f32 store and a
//              per-image max via one atomicMax per block.  HBM: reads P, writes 4P.
//   k_nms    : threshold (> quality*max), 3x3 local max, mask; survivors appended to a per-image list as
//              unique 64-bit keys  (~bits(value) << 32 | linear index)  so ascending key order is
//              (value descending, index ascending).
//   k_select : one 1024-thread workgroup per image.  Repeats { radix-select the next <=4096 smallest keys
//              (8 passes x 8 bit), bitonic sort in LDS, greedy min-distance over the sorted chunk } until
//              max_corners are accepted or the candidates are exhausted.  Result identical to a full sort
//              followed by the serial greedy pass.
#include "ofk_internal.h"

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

// ------------------------------------------------------------------------------------------------ min-eigenvalue map
#define ME_TW 64
#define ME_TH 32

// dynamic LDS layout: gray (GH x GWp) u8 | d (PH x PW) int16x2 | hs[3] (PH x TW) int32
__global__ __launch_bounds__(256) void k_mineig(const uint8_t *__restrict__ gray, size_t gray_stride, int h, int w,
                                                int bs, float kd, float ko, float *__restrict__ eig, size_t eig_stride,
                                                unsigned int *__restrict__ maxbits, const uint8_t *__restrict__ mask,
                                                size_t mask_stride)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int an = bs / 2;
    const int PW = ME_TW + bs - 1, PH = ME_TH + bs - 1;       // product region
    const int GW = PW + 2, GH = PH + 2;                       // gray region
    const int GWp = (GW + 3) & ~3;
    uint8_t *s_g = smem;
    unsigned *s_d = reinterpret_cast<unsigned *>(smem + ((GH * GWp + 15) & ~15));
    int *s_hxx = reinterpret_cast<int *>(s_d + PH * PW);
    int *s_hxy = s_hxx + PH * ME_TW;
    int *s_hyy = s_hxy + PH * ME_TW;
    float *s_max = reinterpret_cast<float *>(s_hyy + PH * ME_TW);

    const int b = blockIdx.z;
    const uint8_t *img = gray + (size_t)b * gray_stride;
    const int x0 = blockIdx.x * ME_TW, y0 = blockIdx.y * ME_TH;
    const int px0 = x0 - an, py0 = y0 - an;                   // product region origin
    const int gx0 = px0 - 1, gy0 = py0 - 1;                   // gray region origin
    const int tid = threadIdx.x;

    for (int i = tid; i < GH * GW; i += 256) {
        const int r = i / GW, c = i - r * GW;
        s_g[r * GWp + c] = img[(size_t)reflect101(gy0 + r, h) * w + reflect101(gx0 + c, w)];
    }
    __syncthreads();
    // Sobel at every product position.  A position mirrored across an image edge reads mirrored gray, which
    // flips the sign of the derivative along that axis; undo it so that the PRODUCT image is what gets reflected.
    for (int i = tid; i < PH * PW; i += 256) {
        const int r = i / PW, c = i - r * PW;
        const uint8_t *r0 = s_g + r * GWp + c, *r1 = r0 + GWp, *r2 = r1 + GWp;
        int dx = (r0[2] - r0[0]) + 2 * (r1[2] - r1[0]) + (r2[2] - r2[0]);
        int dy = (r2[0] - r0[0]) + 2 * (r2[1] - r0[1]) + (r2[2] - r0[2]);
        const int X = px0 + c, Y = py0 + r;
        if (X < 0 || X >= w) dx = -dx;
        if (Y < 0 || Y >= h) dy = -dy;
        s_d[i] = ((unsigned)dx & 0xffffu) | ((unsigned)dy << 16);
    }
    __syncthreads();
    // horizontal box sums: item = (row, 8-column segment), sliding window
    for (int it = tid; it < PH * (ME_TW / 8); it += 256) {
        const int r = it / (ME_TW / 8), seg = it - r * (ME_TW / 8);
        const unsigned *row = s_d + r * PW + seg * 8;
        int sxx = 0, sxy = 0, syy = 0;
        for (int i = 0; i < bs; ++i) {
            const unsigned v = row[i];
            const int dx = (int)(short)(v & 0xffffu), dy = (int)v >> 16;
            sxx += dx * dx; sxy += dx * dy; syy += dy * dy;
        }
        int *oxx = s_hxx + r * ME_TW + seg * 8, *oxy = s_hxy + r * ME_TW + seg * 8, *oyy = s_hyy + r * ME_TW + seg * 8;
        oxx[0] = sxx; oxy[0] = sxy; oyy[0] = syy;
#pragma unroll
        for (int k = 1; k < 8; ++k) {
            const unsigned vo = row[k - 1], vn = row[k - 1 + bs];
            const int dxo = (int)(short)(vo & 0xffffu), dyo = (int)vo >> 16;
            const int dxn = (int)(short)(vn & 0xffffu), dyn = (int)vn >> 16;
            sxx += dxn * dxn - dxo * dxo; sxy += dxn * dyn - dxo * dyo; syy += dyn * dyn - dyo * dyo;
            oxx[k] = sxx; oxy[k] = sxy; oyy[k] = syy;
        }
    }
    __syncthreads();
    // vertical box sums: thread = (column, 8-row segment); lambda_min; store
    float lmax = 0.f;
    {
        const int x = tid & 63, ys = (tid >> 6) * 8;
        int sxx = 0, sxy = 0, syy = 0;
        for (int j = 0; j < bs; ++j) {
            const int q = (ys + j) * ME_TW + x;
            sxx += s_hxx[q]; sxy += s_hxy[q]; syy += s_hyy[q];
        }
        const uint8_t *mk = mask ? mask + (size_t)b * mask_stride : nullptr;
        float *out = eig + (size_t)b * eig_stride;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k) {
                const int qo = (ys + k - 1) * ME_TW + x, qn = (ys + k - 1 + bs) * ME_TW + x;
                sxx += s_hxx[qn] - s_hxx[qo]; sxy += s_hxy[qn] - s_hxy[qo]; syy += s_hyy[qn] - s_hyy[qo];
            }
            const int gy = y0 + ys + k, gx = x0 + x;
            if (gy < h && gx < w) {
                const float a = (float)sxx * kd, bb = (float)sxy * ko, c = (float)syy * kd;
                const float amc = a - c;
                const float v = (a + c) - sqrtf(amc * amc + bb * bb);
                out[(size_t)gy * w + gx] = v;
                if (!mk || mk[(size_t)gy * w + gx]) lmax = fmaxf(lmax, v);
            }
        }
    }
    if (maxbits) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o));
        if ((tid & 63) == 0) s_max[tid >> 6] = lmax;
        __syncthreads();
        if (tid == 0) {
            const float m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
            if (m > 0.f) atomicMax(maxbits + b, __float_as_uint(m));
        }
    }
}

int ofk_launch_mineig(hipStream_t s, const uint8_t *gray, size_t gray_stride, int h, int w, int block, float *eig,
                      size_t eig_stride, unsigned int *maxbits, const uint8_t *mask, size_t mask_stride, int batch)
{
    const int PW = ME_TW + block - 1, PH = ME_TH + block - 1, GW = PW + 2, GH = PH + 2, GWp = (GW + 3) & ~3;
    const size_t lds = ((size_t)(GH * GWp + 15) & ~(size_t)15) + (size_t)PH * PW * 4 + (size_t)3 * PH * ME_TW * 4 + 16;
    if (lds > 160 * 1024) return -1;
    const double scale = 1.0 / (4.0 * block * 255.0);
    const float kd = (float)(0.5 * scale * scale), ko = (float)(scale * scale);
    static size_t attr_lds = 64 * 1024;                       // dynamic LDS above 64 KiB has to be opted into
    if (lds > attr_lds) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_mineig), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return -1;
        }
        attr_lds = lds;
    }
    dim3 grid((w + ME_TW - 1) / ME_TW, (h + ME_TH - 1) / ME_TH, batch);
    hipLaunchKernelGGL(k_mineig, grid, dim3(256), lds, s, gray, gray_stride, h, w, block, kd, ko, eig, eig_stride, maxbits,
                       mask, mask_stride);
    return 0;
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
        if (m > 0.f) atomicMax(maxbits + b, __float_as_uint(m));
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

// ------------------------------------------------------------------------------------------------ threshold + NMS + compaction
__global__ __launch_bounds__(256) void k_nms(const float *__restrict__ eig, size_t eig_stride,
                                             const uint8_t *__restrict__ mask, size_t mask_stride, int h, int w,
                                             const unsigned int *__restrict__ maxbits, double quality,
                                             unsigned long long *__restrict__ cand, int cand_cap,
                                             int *__restrict__ cand_count, int *__restrict__ flags)
{
    const int b = blockIdx.z;
    const unsigned mb = maxbits[b];
    if (mb == 0) return;                                       // max <= 0: no corners
    const float thr = (float)((double)__uint_as_float(mb) * quality);
    const float *e = eig + (size_t)b * eig_stride;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    bool keep = false;
    float v = 0.f;
    if (x >= 1 && x < w - 1 && y >= 1 && y < h - 1) {
        const size_t i = (size_t)y * w + x;
        v = e[i];
        if (v > thr && (!mask || mask[(size_t)b * mask_stride + i])) {
            const float *r0 = e + i - w, *r2 = e + i + w;
            const float m = fmaxf(fmaxf(fmaxf(r0[-1], r0[0]), fmaxf(r0[1], e[i - 1])),
                                  fmaxf(fmaxf(e[i + 1], r2[-1]), fmaxf(r2[0], r2[1])));
            keep = !(m > v);
        }
    }
    if (keep) {
        const int slot = atomicAdd(cand_count + b, 1);          // hipcc aggregates this per wave
        if (slot < cand_cap)
            cand[(size_t)b * cand_cap + slot] =
                ((unsigned long long)(~__float_as_uint(v)) << 32) | (unsigned)(y * w + x);
        else
            atomicOr(flags, 1);
    }
}

void ofk_launch_nms(hipStream_t s, const float *eig, size_t eig_stride, const uint8_t *mask, size_t mask_stride, int h,
                    int w, const unsigned int *maxbits, double quality, unsigned long long *cand, int cand_cap,
                    int *cand_count, int *flags, int batch)
{
    dim3 grid((w + 63) / 64, (h + 3) / 4, batch);
    hipLaunchKernelGGL(k_nms, grid, dim3(256), 0, s, eig, eig_stride, mask, mask_stride, h, w, maxbits, quality, cand,
                       cand_cap, cand_count, flags);
}

// ------------------------------------------------------------------------------------------------ sort + greedy min-distance
#define SEL_T 1024

__global__ __launch_bounds__(SEL_T) void k_select(unsigned long long *__restrict__ cand_all, int cand_cap,
                                                  const int *__restrict__ cand_count, int w, int max_corners,
                                                  float min_distance, float *__restrict__ pts, int pts_stride,
                                                  int *__restrict__ counts)
{
    __shared__ unsigned long long s_key[OFK_CHUNK];
    __shared__ unsigned s_hist[256];
    __shared__ int s_acc_xy[4096];                              // accepted corners, x | y<<16 (max_corners <= 4096)
    __shared__ unsigned long long s_rej[16];                    // per-wave reject ballots of the current 64-candidate round
    __shared__ int s_k, s_n, s_nacc, s_digit;

    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long *cand = cand_all + (size_t)b * cand_cap;
    int C = cand_count[b];
    if (C > cand_cap) C = cand_cap;
    const float md2 = min_distance * min_distance;
    const bool use_dist = min_distance >= 1.f;
    unsigned long long lo = 0;                                  // keys <= lo were handled by earlier chunks
    int remaining = C;
    if (tid == 0) s_nacc = 0;
    __syncthreads();

    while (remaining > 0) {
        int n;                                                  // keys in this chunk
        unsigned long long hi;                                  // largest key of this chunk
        if (remaining <= OFK_CHUNK) {
            n = remaining; hi = ~0ull;
        } else {
            // radix select: the OFK_CHUNK-th smallest key among keys > lo
            unsigned long long prefix = 0; int k = OFK_CHUNK;
            for (int pass = 7; pass >= 0; --pass) {
                if (tid < 256) s_hist[tid] = 0;
                __syncthreads();
                const int sh = pass * 8;
                const unsigned long long himask = pass == 7 ? 0ull : (~0ull << (sh + 8));
                for (int i = tid; i < C; i += SEL_T) {
                    const unsigned long long key = cand[i];
                    if (key > lo && (key & himask) == prefix) atomicAdd(&s_hist[(unsigned)(key >> sh) & 255u], 1u);
                }
                __syncthreads();
                if (tid == 0) {
                    int acc = 0, d = 0;
                    for (; d < 256; ++d) { if (acc + (int)s_hist[d] >= k) break; acc += (int)s_hist[d]; }
                    s_digit = d; s_k = k - acc;
                }
                __syncthreads();
                prefix |= (unsigned long long)s_digit << sh; k = s_k;
                __syncthreads();
            }
            hi = prefix; n = OFK_CHUNK;
        }
        // gather keys in (lo, hi] into LDS (unordered), pad with ~0
        if (tid == 0) s_n = 0;
        __syncthreads();
        for (int i = tid; i < C; i += SEL_T) {
            const unsigned long long key = cand[i];
            if (key > lo && key <= hi) s_key[atomicAdd(&s_n, 1)] = key;
        }
        __syncthreads();
        int npad = 64;
        while (npad < n) npad <<= 1;
        for (int i = n + tid; i < npad; i += SEL_T) s_key[i] = ~0ull;
        __syncthreads();
        // bitonic sort ascending
        for (int kk = 2; kk <= npad; kk <<= 1)
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < npad; i += SEL_T) {
                    const int p = i ^ j;
                    if (p > i) {
                        const unsigned long long a = s_key[i], c = s_key[p];
                        const bool up = (i & kk) == 0;
                        if ((a > c) == up) { s_key[i] = c; s_key[p] = a; }
                    }
                }
                __syncthreads();
            }
        // greedy over the sorted chunk, 64 candidates per round
        for (int base = 0; base < n; base += 64) {
            const int nacc = s_nacc;
            if (max_corners > 0 && nacc >= max_corners) break;
            const int ci = base + lane;
            const bool live = ci < n;
            const unsigned idx = live ? (unsigned)(s_key[ci] & 0xffffffffu) : 0u;
            const int cx = (int)(idx % (unsigned)w), cy = (int)(idx / (unsigned)w);
            bool rej = false;
            if (use_dist && live)
                for (int j = wave; j < nacc; j += SEL_T / 64) {
                    const int a = s_acc_xy[j];
                    const int dx = cx - (a & 0xffff), dy = cy - (a >> 16);
                    if ((float)(dx * dx + dy * dy) < md2) { rej = true; break; }
                }
            const unsigned long long bal = __ballot(rej);
            if (lane == 0) s_rej[wave] = bal;
            __syncthreads();
            if (wave == 0) {
                unsigned long long r = 0;
#pragma unroll
                for (int q = 0; q < SEL_T / 64; ++q) r |= s_rej[q];
                unsigned long long m = __ballot(live) & ~r;    // survivors of the accepted-set test, best first
                int na = nacc;
                while (m && (max_corners <= 0 || na < max_corners)) {
                    const int win = __ffsll((long long)m) - 1;
                    const int wx = __shfl(cx, win), wy = __shfl(cy, win);
                    if (lane == 0) {
                        s_acc_xy[na] = wx | (wy << 16);
                        pts[((size_t)b * pts_stride + na) * 2] = (float)wx;
                        pts[((size_t)b * pts_stride + na) * 2 + 1] = (float)wy;
                    }
                    ++na;
                    const int dx = cx - wx, dy = cy - wy;
                    const bool clash = use_dist && (float)(dx * dx + dy * dy) < md2;
                    m &= ~__ballot(clash);
                    m &= ~(1ull << win);
                }
                if (lane == 0) s_nacc = na;
            }
            __syncthreads();
        }
        if (max_corners > 0 && s_nacc >= max_corners) break;
        remaining -= n;
        lo = hi;
        __syncthreads();
    }
    if (tid == 0) counts[b] = s_nacc;
}

void ofk_launch_select(hipStream_t s, unsigned long long *cand, int cand_cap, const int *cand_count, int w,
                       int max_corners, float min_distance, float *pts, int pts_stride, int *counts, int batch)
{
    hipLaunchKernelGGL(k_select, dim3(batch), dim3(SEL_T), 0, s, cand, cand_cap, cand_count, w, max_corners, min_distance,
                       pts, pts_stride, counts);
}
