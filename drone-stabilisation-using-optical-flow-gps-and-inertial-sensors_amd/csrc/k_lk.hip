// k_lk.hip — pyramidal Lucas-Kanade (cv2.calcOpticalFlowPyrLK semantics).  gfx950.
//   k_lk15q   the pipeline's kernel: 15 x 15 window, FOUR points per wave (a 16-lane DPP row per point) — see its own header below;
//   k_lk15    windows <= 15 on levels too small or too oddly sized for k_lk15q: one wave per point, lane = (row, 4-pixel segment);
//   k_lk<W>   windows up to 21 / 31: one wave per point, generic.
// Common to all three: a 64-thread workgroup (= one wave) tracks its point(s) through all pyramid levels, coarse to fine:
//   * the (win+3)^2 neighbourhood of the previous-frame level is staged into LDS, the Scharr derivatives of its
//     (win+1)^2 core are computed there (never materialised in HBM), and each lane keeps its <= NPL window
//     pixels (I, Ix, Iy as int16) in registers;
//   * a (win+1+2M)^2 region of the next-frame level is staged into LDS once per level and re-staged only when the
//     window walks out of it, so Newton iterations touch LDS only;
//   * the 2x2 normal matrix and the mismatch vector are exact integer sums reduced across the wave with
//     shuffles (order-independent), converted to f32 once; the 2x2 solve is f32 with a fixed operation order.
// Parity target: bit-identical to oracle/image_oracle.c:orc_lk_pyr.
#include "ofk_internal.h"
#include <float.h>

#define LK_M 8                                   // margin of the staged next-frame region (pixels each side)

__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}
__device__ __forceinline__ int descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }

__device__ __forceinline__ long long wave_sum_i64(long long v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_sum_i32(int v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// cvRound of a value in [0, 2^22): adding 1.5 * 2^23 leaves the integer, rounded half to even by the addition itself, in the low
// mantissa bits — one full-rate f32 add and one integer subtract instead of v_rndne_f32 + v_cvt_i32_f32 (4 clocks each).
__device__ __forceinline__ int rn_small(float x) { return __float_as_int(x + 12582912.f) - 0x4B400000; }

__device__ __forceinline__ void lk_weights(float a, float b, int &w00, int &w01, int &w10, int &w11)
{
    // fl(fl((1-a)(1-b)) * 2^14) = fl((1-a) * ((1-b) * 2^14)): scaling by a power of two commutes with rounding
    const float a1 = 1.f - a, b1 = (1.f - b) * 16384.f, b0 = b * 16384.f;
    w00 = rn_small(a1 * b1);
    w01 = rn_small(a * b1);
    w10 = rn_small(a1 * b0);
    w11 = 16384 - w00 - w01 - w10;
}

template <int WMAX>
__global__ __launch_bounds__(64) void k_lk(const uint8_t *__restrict__ prev, const uint8_t *__restrict__ next,
                                           size_t pyr_stride, ofk_levels lv, const float *__restrict__ prev_pts,
                                           const int *__restrict__ counts, int pts_stride, int win, int max_count,
                                           double eps2, double min_eig_thr, float *__restrict__ next_pts,
                                           uint8_t *__restrict__ status, float *__restrict__ err)
{
    constexpr int NPL = (WMAX * WMAX + 63) / 64;               // window pixels per lane
    constexpr int IW = WMAX + 3;                               // staged prev neighbourhood
    constexpr int DW = WMAX + 1;                               // derivative core
    constexpr int JW = WMAX + 1 + 2 * LK_M;                    // staged next region
    __shared__ uint8_t s_I[IW * IW];
    __shared__ unsigned s_D[DW * DW];
    __shared__ uint8_t s_J[JW * JW];

    const int b = blockIdx.y, p = blockIdx.x, lane = threadIdx.x;
    if (p >= counts[b]) return;
    const size_t pi = (size_t)b * pts_stride + p;
    const float ptx = prev_pts[2 * pi], pty = prev_pts[2 * pi + 1];
    const uint8_t *Pb = prev + (size_t)b * pyr_stride, *Nb = next + (size_t)b * pyr_stride;
    const float half = (float)(win - 1) * 0.5f;
    const float eps2_lo = (float)(eps2 * (1.0 - 1e-5)), eps2_hi = (float)(eps2 * (1.0 + 1e-5));
    const int ww = win * win;
    const int iw_ = win + 3, dw_ = win + 1, jw_ = win + 1 + 2 * LK_M;

    int st = 1;
    float errv = 0.f, nx = 0.f, ny = 0.f;
    short pI[NPL], pIx[NPL], pIy[NPL];

    for (int l = lv.n; l >= 0; --l) {
        const int lh = lv.h[l], lw = lv.w[l];
        const uint8_t *I = Pb + lv.off[l], *J = Nb + lv.off[l];
        const float sc = __int_as_float((127 - l) << 23);           // 2^-l, exactly what (float)(1.0 / (double)(1 << l)) is
        float px = ptx * sc, py = pty * sc, qx, qy;
        if (l == lv.n) { qx = px; qy = py; } else { qx = nx * 2.f; qy = ny * 2.f; }
        nx = qx; ny = qy;
        px -= half; py -= half;
        const int ipx = __builtin_amdgcn_readfirstlane((int)floorf(px)), ipy = __builtin_amdgcn_readfirstlane((int)floorf(py));
        if (ipx < -win || ipx >= lw || ipy < -win || ipy >= lh) {
            if (l == 0) { st = 0; errv = 0.f; }
            continue;
        }
        // stage prev neighbourhood (origin ipx-1, ipy-1) and its Scharr derivatives
        __syncthreads();
        for (int i = lane; i < iw_ * iw_; i += 64) {
            const int r = i / iw_, c = i - r * iw_;
            s_I[i] = I[(size_t)reflect101(ipy - 1 + r, lh) * lw + reflect101(ipx - 1 + c, lw)];
        }
        __syncthreads();
        for (int i = lane; i < dw_ * dw_; i += 64) {
            const int r = i / dw_, c = i - r * dw_;
            const int X = ipx + c, Y = ipy + r;
            unsigned pk = 0;
            if (X >= 0 && X < lw && Y >= 0 && Y < lh) {       // derivative image has a constant-0 border
                const uint8_t *r0 = s_I + r * iw_ + c, *r1 = r0 + iw_, *r2 = r1 + iw_;
                const int dx = 3 * (r0[2] - r0[0]) + 10 * (r1[2] - r1[0]) + 3 * (r2[2] - r2[0]);
                const int dy = 3 * (r2[0] - r0[0]) + 10 * (r2[1] - r0[1]) + 3 * (r2[2] - r0[2]);
                pk = ((unsigned)dx & 0xffffu) | ((unsigned)dy << 16);
            }
            s_D[i] = pk;
        }
        __syncthreads();
        int w00, w01, w10, w11;
        lk_weights(px - (float)ipx, py - (float)ipy, w00, w01, w10, w11);
        int a11 = 0, a12 = 0, a22 = 0;
        long long A11s = 0, A12s = 0, A22s = 0;
#pragma unroll
        for (int j = 0; j < NPL; ++j) {
            const int k = lane + 64 * j;
            pI[j] = 0; pIx[j] = 0; pIy[j] = 0;
            if (k < ww) {
                const int y = k / win, x = k - y * win;
                const uint8_t *i0 = s_I + (y + 1) * iw_ + x + 1, *i1 = i0 + iw_;
                const int iv = descale(i0[0] * w00 + i0[1] * w01 + i1[0] * w10 + i1[1] * w11, 9);
                const unsigned d00 = s_D[y * dw_ + x], d01 = s_D[y * dw_ + x + 1], d10 = s_D[(y + 1) * dw_ + x],
                               d11 = s_D[(y + 1) * dw_ + x + 1];
                const int ix = descale((int)(short)(d00 & 0xffffu) * w00 + (int)(short)(d01 & 0xffffu) * w01 +
                                           (int)(short)(d10 & 0xffffu) * w10 + (int)(short)(d11 & 0xffffu) * w11, 14);
                const int iy = descale(((int)d00 >> 16) * w00 + ((int)d01 >> 16) * w01 + ((int)d10 >> 16) * w10 +
                                           ((int)d11 >> 16) * w11, 14);
                pI[j] = (short)iv; pIx[j] = (short)ix; pIy[j] = (short)iy;
                if (NPL <= 8) { a11 += ix * ix; a12 += ix * iy; a22 += iy * iy; }
                else { A11s += (long long)ix * ix; A12s += (long long)ix * iy; A22s += (long long)iy * iy; }
            }
        }
        if (NPL <= 8) { A11s = a11; A12s = a12; A22s = a22; }
        A11s = wave_sum_i64(A11s); A12s = wave_sum_i64(A12s); A22s = wave_sum_i64(A22s);
        const float A11 = (float)((double)A11s * 0x1p-20), A12 = (float)((double)A12s * 0x1p-20),
                    A22 = (float)((double)A22s * 0x1p-20);
        float D = A11 * A22 - A12 * A12;
        const float dd = A11 - A22;
        const float minEig = (A22 + A11 - sqrtf(dd * dd + 4.f * A12 * A12)) / (float)(2 * ww);
        if ((double)minEig < min_eig_thr || D < FLT_EPSILON) {
            if (l == 0) st = 0;
            continue;
        }
        D = 1.f / D;
        qx -= half; qy -= half;
        float pdx = 0.f, pdy = 0.f;
        int jx0 = 0, jy0 = 0;
        bool jvalid = false;
        int jwx = 0, jwy = 0;                                  // window origin inside the staged region
        auto stage_J = [&](int iqx, int iqy) {
            jx0 = iqx - LK_M; jy0 = iqy - LK_M;
            __syncthreads();
            for (int i = lane; i < jw_ * jw_; i += 64) {
                const int r = i / jw_, c = i - r * jw_;
                s_J[i] = J[(size_t)reflect101(jy0 + r, lh) * lw + reflect101(jx0 + c, lw)];
            }
            __syncthreads();
            jvalid = true;
        };
        for (int j = 0; j < max_count; ++j) {
            // every lane holds the same position: move the integer part to the scalar unit (bounds tests, LDS offsets)
            const int iqx = __builtin_amdgcn_readfirstlane((int)floorf(qx)), iqy = __builtin_amdgcn_readfirstlane((int)floorf(qy));
            if (iqx < -win || iqx >= lw || iqy < -win || iqy >= lh) {
                if (l == 0) st = 0;
                break;
            }
            if (!jvalid || iqx < jx0 || iqx > jx0 + 2 * LK_M || iqy < jy0 || iqy > jy0 + 2 * LK_M) stage_J(iqx, iqy);
            jwx = iqx - jx0; jwy = iqy - jy0;
            lk_weights(qx - (float)iqx, qy - (float)iqy, w00, w01, w10, w11);
            int b1 = 0, b2 = 0;
            long long B1 = 0, B2 = 0;
#pragma unroll
            for (int jj = 0; jj < NPL; ++jj) {
                const int k = lane + 64 * jj;
                if (k < ww) {
                    const int y = k / win, x = k - y * win;
                    const uint8_t *j0 = s_J + (jwy + y) * jw_ + jwx + x, *j1 = j0 + jw_;
                    const int diff = descale(j0[0] * w00 + j0[1] * w01 + j1[0] * w10 + j1[1] * w11, 9) - pI[jj];
                    if (NPL <= 8) { b1 += diff * pIx[jj]; b2 += diff * pIy[jj]; }
                    else { B1 += (long long)diff * pIx[jj]; B2 += (long long)diff * pIy[jj]; }
                }
            }
            if (NPL <= 8) { B1 = b1; B2 = b2; }
            B1 = wave_sum_i64(B1); B2 = wave_sum_i64(B2);
            const float fb1 = (float)((double)B1 * 0x1p-20), fb2 = (float)((double)B2 * 0x1p-20);
            const float dx = (A12 * fb2 - A22 * fb1) * D, dy = (A12 * fb1 - A11 * fb2) * D;
            qx += dx; qy += dy;
            nx = qx + half; ny = qy + half;
            // |delta|^2 <= eps^2 is defined in f64; the f32 value decides it unless it falls within 1e-5 of the threshold
            // (its own error is 2e-7), so the half-rate f64 instructions only run in that band
            const float d2 = dx * dx + dy * dy;
            if (d2 < eps2_lo) break;
            if (d2 <= eps2_hi && (double)dx * (double)dx + (double)dy * (double)dy <= eps2) break;
            // an f32 x satisfies |x| < 0.01 (the f64 constant) iff |x| <= 0.01f: 0.01f is the largest f32 below 0.01
            if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
                nx -= dx * 0.5f; ny -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (st && l == 0) {
            const float ex = nx - half, ey = ny - half;
            const int iex = __builtin_amdgcn_readfirstlane((int)floorf(ex)), iey = __builtin_amdgcn_readfirstlane((int)floorf(ey));
            if (iex < -win || iex >= lw || iey < -win || iey >= lh) { st = 0; continue; }
            if (!jvalid || iex < jx0 || iex > jx0 + 2 * LK_M || iey < jy0 || iey > jy0 + 2 * LK_M) stage_J(iex, iey);
            jwx = iex - jx0; jwy = iey - jy0;
            lk_weights(ex - (float)iex, ey - (float)iey, w00, w01, w10, w11);
            int se = 0;
#pragma unroll
            for (int jj = 0; jj < NPL; ++jj) {
                const int k = lane + 64 * jj;
                if (k < ww) {
                    const int y = k / win, x = k - y * win;
                    const uint8_t *j0 = s_J + (jwy + y) * jw_ + jwx + x, *j1 = j0 + jw_;
                    const int diff = descale(j0[0] * w00 + j0[1] * w01 + j1[0] * w10 + j1[1] * w11, 9) - pI[jj];
                    se += diff < 0 ? -diff : diff;
                }
            }
            se = wave_sum_i32(se);
            errv = (float)se / (float)(32 * ww);
        }
    }
    if (lane == 0) {
        next_pts[2 * pi] = nx; next_pts[2 * pi + 1] = ny;
        status[pi] = (uint8_t)st;
        err[pi] = st ? errv : 0.f;
    }
}


// ================================================================================================
// Fast path, win <= 15: lane = (window row, 4-pixel segment).
//   * staging: all global loads of a level (prev neighbourhood + next region) are issued before the first wait;
//     interior regions come in as aligned dwords re-aligned with v_alignbyte and land in LDS with 16-byte stores;
//   * every tap read is a pair of aligned dwords (ds_read2_b32) + v_alignbyte: 2 LDS reads per lane per iteration;
//   * the Scharr derivatives of the lane's own 2x5 taps are computed from its 4x7 neighbourhood in registers;
//   * reductions: 4 DPP adds inside each 16-lane row (int32 is exact: 16 lanes x 4 px x 8160 x 4080 < 2^31),
//     4 v_readlane, int64 scalar adds — no LDS traffic, no ds_bpermute.
// compiler-level ordering of LDS accesses inside the single wave of a workgroup (see the staging code of k_lk15)
#define LDS_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#define LKF_IP 24                                  // pitch of the staged prev neighbourhood (18 x 18 used)
#define LKF_JW 32                                  // staged next region: 32 x 32 = win + 1 + 2*LK_M at win 15
#define LKF_JP 40                                  // its LDS pitch: 10 banks per row, so the 16 rows a wave reads together hit 16 different bank groups (pitch 32: 4-way conflicts)

__device__ __forceinline__ int row_sum16(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);      // quad_perm(1,0,3,2)
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);      // quad_perm(2,3,0,1)
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);     // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);     // row_mirror
    return v;
}
__device__ __forceinline__ long long wave_sum_rows(int v)
{
    v = row_sum16(v);
    return (long long)__builtin_amdgcn_readlane(v, 0) + (long long)__builtin_amdgcn_readlane(v, 16) +
           (long long)__builtin_amdgcn_readlane(v, 32) + (long long)__builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ float wave_sum_rows_scaled(int v)
{
    v = row_sum16(v);
    // the four row sums are added as 64-bit integers on the scalar unit (they sit in SGPRs after v_readlane); the int64 is
    // below 2^33, so its f64 image and the scaling are exact and one rounding to f32 remains — as in the oracle
    const long long S = ((long long)__builtin_amdgcn_readlane(v, 0) + (long long)__builtin_amdgcn_readlane(v, 16)) +
                        ((long long)__builtin_amdgcn_readlane(v, 32) + (long long)__builtin_amdgcn_readlane(v, 48));
    // Nearly every sum fits 32 bits (the mismatch sums always do in practice): v_cvt_f32_i32 rounds to nearest even exactly as
    // f64 -> f32 does, and the scaling is a power of two, so one conversion + one multiply replace ten half-rate f64
    // instructions.  S is wave-uniform: the test and the branch run on the scalar unit.
    const int lo = (int)S, hi = (int)(S >> 32);
    int sx;                                                      // sign extension of lo, hidden from the optimiser: it would otherwise
    asm("s_ashr_i32 %0, %1, 31" : "=s"(sx) : "s"(lo));           // fold the test back into a 64-bit range check on the VALU
    if (__builtin_expect(hi == sx, 1)) return (float)lo * 0x1p-20f;
    return (float)((double)S * 0x1p-20);
}

// 5 consecutive bytes starting at byte offset `off` of an LDS byte array (4-byte aligned base)
__device__ __forceinline__ void lds_read5(const uint8_t *base, int off, int t[5])
{
    const unsigned *p = reinterpret_cast<const unsigned *>(base + (off & ~3));
    const unsigned d0 = p[0], d1 = p[1];
    const unsigned sh = (unsigned)off & 3u;
    const unsigned v = __builtin_amdgcn_alignbyte(d1, d0, sh);                 // bytes off .. off+3
    t[0] = v & 255; t[1] = (v >> 8) & 255; t[2] = (v >> 16) & 255; t[3] = v >> 24; t[4] = (d1 >> (8 * sh)) & 255;
}
// 7 consecutive bytes starting at `off`
__device__ __forceinline__ void lds_read7(const uint8_t *base, int off, int t[7])
{
    const unsigned *p = reinterpret_cast<const unsigned *>(base + (off & ~3));
    const unsigned d0 = p[0], d1 = p[1], d2 = p[2];
    const unsigned sh = (unsigned)off & 3u;
    const unsigned lo = __builtin_amdgcn_alignbyte(d1, d0, sh), hi = __builtin_amdgcn_alignbyte(d2, d1, sh);
    t[0] = lo & 255; t[1] = (lo >> 8) & 255; t[2] = (lo >> 16) & 255; t[3] = lo >> 24;
    t[4] = hi & 255; t[5] = (hi >> 8) & 255; t[6] = (hi >> 16) & 255;
}

// Packed 16-bit arithmetic of k_lk15.  Pixels, Scharr sums (<= 16 * 255) and the 14-bit interpolation weights all fit 16 bits,
// so two adjacent columns travel in one register: v_perm expands byte pairs (b_k, b_k+1) to u16 pairs, v_pk_* instructions run
// the separable Scharr on two columns at once, and ONE v_dot2_i32_i16 evaluates a row of the bilinear interpolation
// (a*w0 + b*w1 + acc) — every sum is an exact integer, so the results are those of the scalar formulation bit for bit.
typedef short lk_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lk_s2 lk_as_s2(unsigned v) { return __builtin_bit_cast(lk_s2, v); }
__device__ __forceinline__ unsigned lk_as_u(lk_s2 v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ int lk_dot2(unsigned a, unsigned w, int acc) { return __builtin_amdgcn_sdot2(lk_as_s2(a), lk_as_s2(w), acc, false); }
#define LK_PAIR_SEL(k) ((unsigned)(k) | 0x0c00u | ((unsigned)((k) + 1) << 16) | 0x0c000000u)   /* v_perm selector: (byte k, 0, byte k+1, 0) */

__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_lk15(const uint8_t *__restrict__ prev, const uint8_t *__restrict__ next,
                                             size_t pyr_stride, ofk_levels lv, const float *__restrict__ prev_pts,
                                             const int *__restrict__ counts, int pts_stride, int win, int max_count,
                                             double eps2, float eps2_lo, float eps2_hi, double min_eig_thr,
                                             float *__restrict__ next_pts, uint8_t *__restrict__ status, float *__restrict__ err)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_I[20 * LKF_IP];
    __shared__ __attribute__((aligned(16))) uint8_t s_J[(LKF_JW + 1) * LKF_JP];

    // XCD-aware block -> (image, point) map.  Workgroups are dealt round-robin over the 8 XCDs (blocks n and n + 8 share one),
    // each with its own L2.  The windows of an image's points overlap heavily (at the coarse levels every point reads most
    // of the level), so all points of an image go to ONE XCD — image b to the XCD of blocks n = b (mod 8) — and its pyramid
    // lines are fetched from HBM once instead of once per XCD.  Speed only; any placement gives the same results.
    const int lane = threadIdx.x;
    int b = blockIdx.y, p = blockIdx.x;
    if ((gridDim.y & 7) == 0) {
        const unsigned n = blockIdx.y * gridDim.x + blockIdx.x, k = n >> 3;
        b = 8 * (int)(k / gridDim.x) + (int)(n & 7);
        p = (int)(k % gridDim.x);
    }
    if (p >= counts[b]) return;
    const size_t pi = (size_t)b * pts_stride + p;
    const float ptx = prev_pts[2 * pi], pty = prev_pts[2 * pi + 1];
    const uint8_t *Pb = prev + (size_t)b * pyr_stride, *Nb = next + (size_t)b * pyr_stride;
    const float half = (float)(win - 1) * 0.5f;
    const int ww = win * win;
    const int iw_ = win + 3, jw_ = win + 1 + 2 * LK_M;
    const int wy = lane >> 2, wx0 = (lane & 3) * 4;            // this lane's window row and first column
    const int npx = wy < win ? min(4, max(0, win - wx0)) : 0;  // pixels owned by the lane

    int st = 1;
    float errv = 0.f, nx = 0.f, ny = 0.f;
    int pI[4], pIx[4], pIy[4];

    for (int l = lv.n; l >= 0; --l) {
        const int lh = lv.h[l], lw = lv.w[l];
        const uint8_t *I = Pb + lv.off[l], *J = Nb + lv.off[l];
        const float sc = __int_as_float((127 - l) << 23);           // 2^-l, exactly what (float)(1.0 / (double)(1 << l)) is
        float px = ptx * sc, py = pty * sc, qx, qy;
        if (l == lv.n) { qx = px; qy = py; } else { qx = nx * 2.f; qy = ny * 2.f; }
        nx = qx; ny = qy;
        px -= half; py -= half;
        const int ipx = __builtin_amdgcn_readfirstlane((int)floorf(px)), ipy = __builtin_amdgcn_readfirstlane((int)floorf(py));
        if (ipx < -win || ipx >= lw || ipy < -win || ipy >= lh) {
            if (l == 0) { st = 0; errv = 0.f; }
            continue;
        }
        qx -= half; qy -= half;
        int jx0 = 0, jy0 = 0;
        bool jvalid = false;
        // ---- staging.  One wave per workgroup: the LDS operations of a wave execute in order, so staging and reading need no
        //      s_barrier and, unlike __syncthreads(), no wait for outstanding GLOBAL loads — LDS_FENCE only keeps the compiler
        //      from moving LDS accesses across it.  The loads of the previous-frame neighbourhood and of the next-frame region
        //      are issued back to back (one memory round trip per level instead of two), then both are written to LDS.
        unsigned jd0 = 0, jd1 = 0, jd2 = 0, jd3 = 0, jd4 = 0, jsh = 0;
        bool jinner = false;
        auto J_issue = [&](int iqx, int iqy) {
            jx0 = iqx - LK_M; jy0 = iqy - LK_M;
            // dword path whenever the COLUMNS lie inside the image; rows are mirrored per lane (at the coarse levels a third
            // of the regions cross the top or bottom border, and the byte-wise path costs ~400 VALU instructions)
            jinner = jx0 >= 4 && jx0 + jw_ + 8 <= lw && (lw & 3) == 0 && jw_ == LKF_JW;
            if (jinner) {
                const int r = lane >> 1, hf = lane & 1;                   // 32 rows x 2 halves of 16 bytes
                const size_t addr = (size_t)reflect101(jy0 + r, lh) * lw + jx0 + 16 * hf;
                jsh = (unsigned)addr & 3u;
                const unsigned *g = reinterpret_cast<const unsigned *>(J + (addr & ~(size_t)3));
                jd0 = g[0]; jd1 = g[1]; jd2 = g[2]; jd3 = g[3]; jd4 = g[4];
            }
        };
        auto J_commit = [&]() {
            LDS_FENCE();                                                  // earlier readers of s_J are done
            if (jinner) {
                const int r = lane >> 1, hf = lane & 1;
                uint2 *dstp = reinterpret_cast<uint2 *>(s_J + r * LKF_JP + 16 * hf);      // the pitch keeps 8-byte alignment only
                dstp[0] = make_uint2(__builtin_amdgcn_alignbyte(jd1, jd0, jsh), __builtin_amdgcn_alignbyte(jd2, jd1, jsh));
                dstp[1] = make_uint2(__builtin_amdgcn_alignbyte(jd3, jd2, jsh), __builtin_amdgcn_alignbyte(jd4, jd3, jsh));
            } else {
                for (int i = lane; i < jw_ * jw_; i += 64) {
                    const int r = i / jw_, c = i - r * jw_;
                    s_J[r * LKF_JP + c] = J[(size_t)reflect101(jy0 + r, lh) * lw + reflect101(jx0 + c, lw)];
                }
            }
            LDS_FENCE();
            jvalid = true;
        };
        auto stage_J = [&](int iqx, int iqy) { J_issue(iqx, iqy); J_commit(); };
        {
            // the prev neighbourhood (origin ipx-1, ipy-1), (win+3)^2: one row of <= 18 bytes per lane, 6 dwords in, 5 out
            const bool inner = ipx >= 5 && ipx - 1 + 28 <= lw && (lw & 3) == 0;         // columns inside; rows mirrored per lane
            unsigned d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0, d5 = 0, sh = 0;
            if (inner && lane < iw_) {
                const size_t addr = (size_t)reflect101(ipy - 1 + lane, lh) * lw + (ipx - 1);
                sh = (unsigned)addr & 3u;
                const unsigned *g = reinterpret_cast<const unsigned *>(I + (addr & ~(size_t)3));
                d0 = g[0]; d1 = g[1]; d2 = g[2]; d3 = g[3]; d4 = g[4]; d5 = g[5];
            }
            // every lane holds the same position: move the integer part to the scalar unit (bounds tests, LDS offsets)
            const int iqx = __builtin_amdgcn_readfirstlane((int)floorf(qx)), iqy = __builtin_amdgcn_readfirstlane((int)floorf(qy));
            const bool doJ = !(iqx < -win || iqx >= lw || iqy < -win || iqy >= lh);
            if (doJ) J_issue(iqx, iqy);
            LDS_FENCE();                                                  // the previous level's readers of s_I are done
            if (inner) {
                if (lane < iw_) {
                    unsigned *o = reinterpret_cast<unsigned *>(s_I + lane * LKF_IP);
                    o[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
                    o[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
                    o[2] = __builtin_amdgcn_alignbyte(d3, d2, sh);
                    o[3] = __builtin_amdgcn_alignbyte(d4, d3, sh);
                    o[4] = __builtin_amdgcn_alignbyte(d5, d4, sh);
                }
            } else {
                for (int i = lane; i < iw_ * iw_; i += 64) {
                    const int r = i / iw_, c = i - r * iw_;
                    s_I[r * LKF_IP + c] = I[(size_t)reflect101(ipy - 1 + r, lh) * lw + reflect101(ipx - 1 + c, lw)];
                }
            }
            if (doJ) J_commit();
            else LDS_FENCE();
        }
        // ---- patch: I (5 fractional bits), Ix, Iy of the lane's pixels; exact integer normal matrix
        int w00, w01, w10, w11;
        lk_weights(px - (float)ipx, py - (float)ipy, w00, w01, w10, w11);
        int a11 = 0, a12 = 0, a22 = 0;
        pI[0] = pI[1] = pI[2] = pI[3] = 0; pIx[0] = pIx[1] = pIx[2] = pIx[3] = 0; pIy[0] = pIy[1] = pIy[2] = pIy[3] = 0;
        if (npx > 0) {
            // rows wy..wy+3 of s_I, columns wx0..wx0+7 (the staged rows are dword aligned): P[r][k] = (n[r][k], n[r][k+1])
            unsigned P[4][7];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned *pr = reinterpret_cast<const unsigned *>(s_I + (wy + r) * LKF_IP + wx0);
                const unsigned d0 = pr[0], d1 = pr[1];
#pragma unroll
                for (int k = 0; k < 7; ++k) P[r][k] = __builtin_amdgcn_perm(d1, d0, LK_PAIR_SEL(k));
            }
            // Scharr at the window taps (rows wy, wy+1; columns wx0..wx0+4), separable and two columns per instruction:
            // slot s = columns (2s, 2s+1).  hd = n[c+2] - n[c], hs = 3 (n[c] + n[c+2]) + 10 n[c+1] per neighbourhood row,
            // dx = 3 (hd_r + hd_{r+2}) + 10 hd_{r+1}, dy = hs_{r+2} - hs_r.  (The upper half of slot 2 is column 5: unused.)
            unsigned DX[2][3], DY[2][3];
#pragma unroll
            for (int sl = 0; sl < 3; ++sl) {
                lk_s2 hd[4], hs[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const lk_s2 a = lk_as_s2(P[r][2 * sl]), m = lk_as_s2(P[r][2 * sl + 1]), c2 = lk_as_s2(P[r][2 * sl + 2]);
                    hd[r] = c2 - a;
                    hs[r] = (a + c2) * (short)3 + m * (short)10;
                }
                DX[0][sl] = lk_as_u((hd[0] + hd[2]) * (short)3 + hd[1] * (short)10); DX[1][sl] = lk_as_u((hd[1] + hd[3]) * (short)3 + hd[2] * (short)10);
                DY[0][sl] = lk_as_u(hs[2] - hs[0]); DY[1][sl] = lk_as_u(hs[3] - hs[1]);
            }
            if (!(ipx >= 0 && ipx + win < lw && ipy >= 0 && ipy + win < lh)) {      // wave-uniform: the window touches the border
#pragma unroll
                for (int r = 0; r < 2; ++r)
#pragma unroll
                    for (int sl = 0; sl < 3; ++sl) {
                        const int X = ipx + wx0 + 2 * sl, Y = ipy + wy + r;
                        const bool rowok = Y >= 0 && Y < lh;
                        const unsigned keep = ((rowok && X >= 0 && X < lw) ? 0x0000ffffu : 0u) | ((rowok && X + 1 >= 0 && X + 1 < lw) ? 0xffff0000u : 0u);
                        DX[r][sl] &= keep; DY[r][sl] &= keep;                   // constant-0 derivative border
                    }
            }
            // pairs starting at column k = 0..3: (k, k+1)
            unsigned QX[2][4], QY[2][4];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                QX[r][0] = DX[r][0]; QX[r][1] = __builtin_amdgcn_alignbit(DX[r][1], DX[r][0], 16); QX[r][2] = DX[r][1];
                QX[r][3] = __builtin_amdgcn_alignbit(DX[r][2], DX[r][1], 16);
                QY[r][0] = DY[r][0]; QY[r][1] = __builtin_amdgcn_alignbit(DY[r][1], DY[r][0], 16); QY[r][2] = DY[r][1];
                QY[r][3] = __builtin_amdgcn_alignbit(DY[r][2], DY[r][1], 16);
            }
            const unsigned W0 = (unsigned)w00 | ((unsigned)w01 << 16), W1 = (unsigned)w10 | ((unsigned)w11 << 16);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (k < npx) {
                    // descale(a w00 + b w01 + c w10 + d w11, n) = (two dot products + 2^(n-1)) >> n
                    const int iv = lk_dot2(P[1][k + 1], W0, lk_dot2(P[2][k + 1], W1, 1 << 8)) >> 9;
                    const int ix = lk_dot2(QX[0][k], W0, lk_dot2(QX[1][k], W1, 1 << 13)) >> 14;
                    const int iy = lk_dot2(QY[0][k], W0, lk_dot2(QY[1][k], W1, 1 << 13)) >> 14;
                    pI[k] = iv; pIx[k] = ix; pIy[k] = iy;
                    a11 += __mul24(ix, ix); a12 += __mul24(ix, iy); a22 += __mul24(iy, iy);
                }
        }
        const float A11 = wave_sum_rows_scaled(a11), A12 = wave_sum_rows_scaled(a12), A22 = wave_sum_rows_scaled(a22);
        float D = A11 * A22 - A12 * A12;
        const float dd = A11 - A22;
        const float minEig = (A22 + A11 - sqrtf(dd * dd + 4.f * A12 * A12)) / (float)(2 * ww);
        if ((double)minEig < min_eig_thr || D < FLT_EPSILON) {
            if (l == 0) st = 0;
            continue;
        }
        D = 1.f / D;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < max_count; ++j) {
            // every lane holds the same position: move the integer part to the scalar unit (bounds tests, LDS offsets)
            const int iqx = __builtin_amdgcn_readfirstlane((int)floorf(qx)), iqy = __builtin_amdgcn_readfirstlane((int)floorf(qy));
            if (iqx < -win || iqx >= lw || iqy < -win || iqy >= lh) {
                if (l == 0) st = 0;
                break;
            }
            if (!jvalid || iqx < jx0 || iqx > jx0 + 2 * LK_M || iqy < jy0 || iqy > jy0 + 2 * LK_M) stage_J(iqx, iqy);
            lk_weights(qx - (float)iqx, qy - (float)iqy, w00, w01, w10, w11);
            int b1 = 0, b2 = 0;
            if (npx > 0) {
                // two rows of 5 bytes at byte offset sh = (iqx - jx0) & 3 of two dwords each; the offset is wave-uniform, so the
                // v_perm selectors that expand the byte pairs (t[k], t[k+1]) come from the scalar unit
                const int off = (iqy - jy0 + wy) * LKF_JP + (iqx - jx0) + wx0;
                const unsigned *r0 = reinterpret_cast<const unsigned *>(s_J + (off & ~3)), *r1 = reinterpret_cast<const unsigned *>(s_J + ((off + LKF_JP) & ~3));
                const unsigned d0 = r0[0], d1 = r0[1], e0 = r1[0], e1 = r1[1];
                const unsigned shs = (unsigned)__builtin_amdgcn_readfirstlane((iqx - jx0) & 3) * 0x00010001u;
                const unsigned W0 = (unsigned)w00 | ((unsigned)w01 << 16), W1 = (unsigned)w10 | ((unsigned)w11 << 16);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < npx) {
                        const unsigned sel = LK_PAIR_SEL(k) + shs;
                        const int diff = (lk_dot2(__builtin_amdgcn_perm(d1, d0, sel), W0, lk_dot2(__builtin_amdgcn_perm(e1, e0, sel), W1, 1 << 8)) >> 9) - pI[k];
                        b1 += __mul24(diff, pIx[k]); b2 += __mul24(diff, pIy[k]);
                    }
            }
            const float fb1 = wave_sum_rows_scaled(b1), fb2 = wave_sum_rows_scaled(b2);
            const float dx = (A12 * fb2 - A22 * fb1) * D, dy = (A12 * fb1 - A11 * fb2) * D;
            qx += dx; qy += dy;
            nx = qx + half; ny = qy + half;
            // |delta|^2 <= eps^2 is defined in f64; the f32 value decides it unless it falls within 1e-5 of the threshold
            // (its own error is 2e-7), so the half-rate f64 instructions only run in that band
            const float d2 = dx * dx + dy * dy;
            if (d2 < eps2_lo) break;
            if (d2 <= eps2_hi && (double)dx * (double)dx + (double)dy * (double)dy <= eps2) break;
            // an f32 x satisfies |x| < 0.01 (the f64 constant) iff |x| <= 0.01f: 0.01f is the largest f32 below 0.01
            if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
                nx -= dx * 0.5f; ny -= dy * 0.5f;
                break;
            }
            pdx = dx; pdy = dy;
        }
        if (st && l == 0) {
            const float ex = nx - half, ey = ny - half;
            const int iex = __builtin_amdgcn_readfirstlane((int)floorf(ex)), iey = __builtin_amdgcn_readfirstlane((int)floorf(ey));
            if (iex < -win || iex >= lw || iey < -win || iey >= lh) { st = 0; continue; }
            if (!jvalid || iex < jx0 || iex > jx0 + 2 * LK_M || iey < jy0 || iey > jy0 + 2 * LK_M) stage_J(iex, iey);
            lk_weights(ex - (float)iex, ey - (float)iey, w00, w01, w10, w11);
            int se = 0;
            if (npx > 0) {
                const int off = (iey - jy0 + wy) * LKF_JP + (iex - jx0) + wx0;
                const unsigned *r0 = reinterpret_cast<const unsigned *>(s_J + (off & ~3)), *r1 = reinterpret_cast<const unsigned *>(s_J + ((off + LKF_JP) & ~3));
                const unsigned d0 = r0[0], d1 = r0[1], e0 = r1[0], e1 = r1[1];
                const unsigned shs = (unsigned)__builtin_amdgcn_readfirstlane((iex - jx0) & 3) * 0x00010001u;
                const unsigned W0 = (unsigned)w00 | ((unsigned)w01 << 16), W1 = (unsigned)w10 | ((unsigned)w11 << 16);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < npx) {
                        const unsigned sel = LK_PAIR_SEL(k) + shs;
                        const int diff = (lk_dot2(__builtin_amdgcn_perm(d1, d0, sel), W0, lk_dot2(__builtin_amdgcn_perm(e1, e0, sel), W1, 1 << 8)) >> 9) - pI[k];
                        se += diff < 0 ? -diff : diff;
                    }
            }
            const long long SE = wave_sum_rows(se);
            errv = (float)(int)SE / (float)(32 * ww);
        }
    }
    if (lane == 0) {
        next_pts[2 * pi] = nx; next_pts[2 * pi + 1] = ny;
        status[pi] = (uint8_t)st;
        err[pi] = st ? errv : 0.f;
    }
}

// ================================================================================================
// k_lk15q — win == 15, FOUR points per wave: lane = (point g = lane / 16, window row r = lane % 16).
//
// k_lk15 spends most of its issue slots on work that is the same for all 64 lanes of a point: interpolation weights, the 2x2
// solve, convergence tests, 12 cross-lane instructions per window sum.  Here a 16-lane DPP row owns a point and a lane owns one
// window row of 15 pixels, so that per-point work is shared by four points, a window sum is 4 DPP adds that leave the total in
// every lane of the row (no v_readlane, no scalar round trip), and the Scharr pass runs on full rows: vertical pass first on
// 9 packed column pairs, the horizontal pass is then one packed subtract (dx) or one v_alignbit + 3 packed ops (dy) per pair.
//   * staging: LDS keeps RAW aligned dwords of each staged row (prev: 18 rows x 6 dwords, next: 32 rows x 9 dwords per point);
//     the byte offset of a row's first pixel goes into the v_perm selectors that expand byte pairs, so nothing is re-aligned.
//     Columns outside the image: image widths are multiples of 4, so an aligned dword is wholly inside or wholly outside; an
//     outside dword is ONE v_perm of two inside dwords (reflect-101 reverses bytes), loaded from the mirrored address.
//   * a lane interpolates its derivative row with both weight rows; the half that belongs to the window row above travels there
//     inside the add (v_add_u32_dpp row_shl:1).
//   * the mismatch sums use b = sum(J * Ixy) - sum(I * Ixy): the second term is a per-lane constant of the level; (Ix, Iy) sit
//     packed in one register per pixel and feed v_mad_i32_i16 (op_sel picks the half).
//   * exactness: a lane's 15 products and a quad's 60 fit int32 always; a whole window may not (|sum| < 2^33).  With
//     A11, A22 < 3e8 Cauchy-Schwarz bounds every partial sum of diff * Ix below 2^31 and the plain 32-bit DPP tree is exact;
//     windows above that (full-contrast noise) reduce the 16-bit halves of the lane sums separately.  A11, A22 < 2^32 always
//     (225 * 4080^2) and reduce as unsigned; A12 fits int32 when max(A11, A22) < 2^31, else it takes the split path too.
// The four points of a wave iterate until the last one has converged; finished rows idle.  Results are those of k_lk15 bit for
// bit (integer sums are order-free, the float operations and their order are the same).
#define LKQ_IP 28                                  // LDS pitch of a staged prev row: 6 dwords + 1 (odd dword pitch: no bank conflicts between rows)
#define LKQ_JP 36                                  // LDS pitch of a staged next row: 9 dwords
#define LKQ_ISZ (18 * LKQ_IP)
#define LKQ_JSZ (32 * LKQ_JP)
#define LKQ_SAFE_LIM 300000000u                    // 8160 * sqrt(225 * A) < 2^31  <=>  A < 3.078e8

__device__ __forceinline__ int lkq_reflect(int i, int n)               // one reflection: -n < i < 2n - 1
{
    i = max(i, -i);
    return i >= n ? 2 * (n - 1) - i : i;
}
__device__ __forceinline__ int lkq_floor_i(float x)
{
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));               // floor + convert in one instruction
    return r;
}
// a.lo * w.lo + a.hi * w.hi + c with the rounding constant c in an SGPR: the VOP2 form the compiler picks (v_dot2c) accumulates in
// place and needs a v_mov of the constant first
__device__ __forceinline__ int lkq_dot2_k(unsigned a, unsigned w, int c)
{
    int d;
    asm("v_dot2_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(w), "s"(c));
    return d;
}
__device__ __forceinline__ int lkq_mad_lo(int a, unsigned packed, int acc)     // acc + a.lo16 * packed.lo16
{
    int d;
    asm("v_mad_i32_i16 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(packed), "v"(acc));
    return d;
}
__device__ __forceinline__ int lkq_mad_hi(int a, unsigned packed, int acc)     // acc + a.lo16 * packed.hi16
{
    int d;
    asm("v_mad_i32_i16 %0, %1, %2, %3 op_sel:[0,1,0,0]" : "=v"(d) : "v"(a), "v"(packed), "v"(acc));
    return d;
}
// sum over the 16 lanes of a DPP row, total in every lane
__device__ __forceinline__ int lkq_row_sum(int v)
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);      // quad_perm(1,0,3,2)
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);      // quad_perm(2,3,0,1)
    v += __builtin_amdgcn_update_dpp(0, v, 0x124, 0xF, 0xF, true);     // row_ror:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true);     // row_ror:8
    return v;
}
// exact sum of 16 int32 lane values as f32(sum * 2^-20): halves reduced separately (|sum of halves| < 2^21)
__device__ __forceinline__ float lkq_row_sum_split(int v)
{
    const int lo = lkq_row_sum(v & 0xffff), hi = lkq_row_sum(v >> 16);
    return (float)((double)(((long long)hi << 16) + lo) * 0x1p-20);
}
// Staging of one image row into LDS as ND raw aligned dwords from column A (multiple of 4, any sign), reflect-101 columns.
// The loads always come from inside the row: the strip of ND dwords at Sb = clamp(A, 0, lw - 4 ND).  Interior rows (Sb == A)
// store dword j at slot j.  A row over a border stores dword j at slot j + (Sb - A) / 4 when that is inside the staged row, and
// fills the outside slots with mirrored dwords: the dword at virtual column X < 0 is bytes n[-X], n[-X-1], n[-X-2], n[-X-3] — one
// v_perm of strip dwords j = -X/4 and j-1 — and X >= lw likewise from the strip at the right edge.  Slot positions are per-lane LDS
// addresses, so no register is indexed by a lane-varying amount; writes that fall outside go to a dump slot.
template <int ND>
__device__ __forceinline__ void lkq_load_row(const uint8_t *rowp, int lw, int A, unsigned (&d)[ND])
{
    const unsigned *g = reinterpret_cast<const unsigned *>(rowp + min(max(A, 0), lw - 4 * ND));
#pragma unroll
    for (int i = 0; i < ND; ++i) d[i] = g[i];
}
template <int ND>
__device__ __forceinline__ void lkq_store_row(unsigned *o, unsigned *dump, int lw, int A, bool border, const unsigned (&d)[ND])
{
    if (!border) {                                                      // wave-uniform: no lane of the wave is over a border
#pragma unroll
        for (int i = 0; i < ND; ++i) o[i] = d[i];
        return;
    }
    const bool left = A < 0, right = A + 4 * ND > lw;
    const int dpos = min(max(A, 0), lw - 4 * ND) - A;                   // byte position of strip dword 0 in the staged row
    const unsigned msel = left ? 0x01020304u : 0x03040506u;
    const int mbase = left ? -A : right ? lw + 4 * (ND - 1) - A : -1;   // mirrored dword of strip pair (j, j-1) sits at mbase - 4 j
#pragma unroll
    for (int j = 0; j < ND; ++j) {
        const int pos = dpos + 4 * j;
        *((unsigned)pos < 4u * ND ? o + (pos >> 2) : dump) = d[j];
        if (j > 0) {
            const int mpos = mbase - 4 * j;
            *((unsigned)mpos < 4u * ND ? o + (mpos >> 2) : dump) = __builtin_amdgcn_perm(d[j], d[j - 1], msel);
        }
    }
}

// 120 VGPRs (amdgpu_num_vgpr counts register PAIRS of the unified file: 60): four waves per SIMD leave 32 registers, i.e. room for one
// 32-register gray wave beside them - the compiler's own allocation after the pixel-pair packing is 124, which takes that room away
#define OFK_LKQ_ATTR __attribute__((amdgpu_num_vgpr(60)))
__global__ __launch_bounds__(64) OFK_LKQ_ATTR void k_lk15q(const uint8_t *__restrict__ prev, const uint8_t *__restrict__ next, size_t pyr_stride,
                                              ofk_levels lv, const float *__restrict__ prev_pts, const int *__restrict__ counts,
                                              int pts_stride, int max_count, double eps2, float eps2_lo, float eps2_hi,
                                              double min_eig_thr, float *__restrict__ next_pts, uint8_t *__restrict__ status,
                                              float *__restrict__ err)
{
    constexpr int win = 15, ww = 225;
    constexpr float half = 7.f;
    __shared__ __attribute__((aligned(16))) uint8_t s_I[4 * LKQ_ISZ];
    __shared__ __attribute__((aligned(16))) uint8_t s_J[4 * LKQ_JSZ];
    __shared__ unsigned s_dump[64];                                // where the staging writes of slots outside a row go (one word per lane)

    const int lane = threadIdx.x, g = lane >> 4, r = lane & 15;
    unsigned *dump = s_dump + lane;
    int b = blockIdx.y, chunk = blockIdx.x;                        // XCD-aware map as in k_lk15: the points of an image stay on one XCD
    if ((gridDim.y & 7) == 0) {
        const unsigned n = blockIdx.y * gridDim.x + blockIdx.x, k = n >> 3;
        b = 8 * (int)(k / gridDim.x) + (int)(n & 7);
        chunk = (int)(k % gridDim.x);
    }
    const int cnt = counts[b];
    if (chunk * 4 >= cnt) return;
    const int p = chunk * 4 + g;
    const bool live = p < cnt;
    const size_t pi = (size_t)b * pts_stride + min(p, cnt - 1);
    const float ptx = prev_pts[2 * pi], pty = prev_pts[2 * pi + 1];
    const uint8_t *Pb = prev + (size_t)b * pyr_stride, *Nb = next + (size_t)b * pyr_stride;
    uint8_t *sI = s_I + g * LKQ_ISZ, *sJ = s_J + g * LKQ_JSZ;
    const bool rowact = r < win;                                   // lane 15 of a row only stages and computes derivative row 15

    int st = 1;
    float errv = 0.f, nx = 0.f, ny = 0.f;
    // Ix and Iy of the lane's 15 window pixels as int16 PAIRS OF NEIGHBOURING PIXELS: pxx[j] = (Ix[2j], Ix[2j+1]), pyy likewise (the last
    // pair's upper half is zero).  Every sum over the row is then a v_dot2_i32_i16 per pixel pair: the normal matrix (3 per pair where
    // round 2 spent 3 multiplies + 3 adds per pixel) and the mismatch vector (pack the two interpolated values with one v_lshl_or, two
    // dot2: 1.5 per pixel where two v_mad_i32_i16 stood) - 7 % fewer instructions per point, sums bit for bit the same integers.
    unsigned pxx[8], pyy[8];

    for (int l = lv.n; l >= 0; --l) {
        const int lh = lv.h[l], lw = lv.w[l];
        const uint8_t *I = Pb + lv.off[l], *J = Nb + lv.off[l];
        const float sc = __int_as_float((127 - l) << 23);
        float px = ptx * sc, py = pty * sc, qx, qy;
        if (l == lv.n) { qx = px; qy = py; } else { qx = nx * 2.f; qy = ny * 2.f; }
        nx = qx; ny = qy;
        px -= half; py -= half;
        const int ipx = lkq_floor_i(px), ipy = lkq_floor_i(py);
        const bool lev = live && !(ipx < -win || ipx >= lw || ipy < -win || ipy >= lh);
        if (live && !lev && l == 0) { st = 0; errv = 0.f; }
        if (__builtin_amdgcn_ballot_w64(lev) == 0) continue;
        qx -= half; qy -= half;

        int jx0 = 0, jy0 = 0, jA = 0;
        bool jvalid = false;
        // The next-frame region of the lanes that are `on`: rows jy0 + r and jy0 + r + 16 from column jA.  At the start of a level
        // both rows are loaded before the first wait (J_issue / J_commit around the prev-frame staging); a re-staging inside the
        // Newton loop (the window walked out of the region: rare) goes row by row to keep 9 registers fewer alive in the loop.
        unsigned jd[2][9];
        auto J_place = [&](int iqx, int iqy, bool on) {
            if (on) { jx0 = iqx - LK_M; jy0 = iqy - LK_M; jA = jx0 & ~3; jvalid = true; }
        };
        auto J_load = [&](int h, bool on) {
            if (on) lkq_load_row<9>(J + (size_t)lkq_reflect(jy0 + r + 16 * h, lh) * lw, lw, jA, jd[h]);
        };
        auto J_store = [&](int h, bool on, bool border) {
            if (on) lkq_store_row<9>(reinterpret_cast<unsigned *>(sJ + (r + 16 * h) * LKQ_JP), dump, lw, jA, border, jd[h]);
        };
        auto J_border = [&](bool on) { return __builtin_amdgcn_ballot_w64(on && (jA < 0 || jA + 36 > lw)) != 0; };
        auto J_restage = [&](int iqx, int iqy, bool on) {
            J_place(iqx, iqy, on);
            const bool border = J_border(on);
            LDS_FENCE();                                                  // earlier readers of s_J are done
            J_load(0, on); J_store(0, on, border);
            LDS_FENCE();
            J_load(1, on); J_store(1, on, border);
            LDS_FENCE();
        };
        // ---- staging: prev neighbourhood rows ipy-1 .. ipy+16 (lane r: row r; lanes 0, 1 also rows 16, 17), columns from
        //      iA = (ipx-1) & ~3; then the next-frame region around the start position.  All loads are issued before the first wait.
        const int iA = (ipx - 1) & ~3;
        {
            const bool iborder = __builtin_amdgcn_ballot_w64(lev && (iA < 0 || iA + 24 > lw)) != 0;
            unsigned id0[6], id1[6];
            if (lev) {
                lkq_load_row<6>(I + (size_t)lkq_reflect(ipy - 1 + r, lh) * lw, lw, iA, id0);
                if (r < 2) lkq_load_row<6>(I + (size_t)lkq_reflect(ipy + 15 + r, lh) * lw, lw, iA, id1);
            }
            const int iqx = lkq_floor_i(qx), iqy = lkq_floor_i(qy);
            const bool doJ = lev && !(iqx < -win || iqx >= lw || iqy < -win || iqy >= lh);
            J_place(iqx, iqy, doJ);
            J_load(0, doJ); J_load(1, doJ);
            LDS_FENCE();                                                  // the previous level's readers of s_I and s_J are done
            if (lev) {
                lkq_store_row<6>(reinterpret_cast<unsigned *>(sI + r * LKQ_IP), dump, lw, iA, iborder, id0);
                if (r < 2) lkq_store_row<6>(reinterpret_cast<unsigned *>(sI + (r + 16) * LKQ_IP), dump, lw, iA, iborder, id1);
            }
            const bool jborder = J_border(doJ);
            J_store(0, doJ, jborder); J_store(1, doJ, jborder);
            LDS_FENCE();
        }
        // taps of the lane's window row: two LDS rows (pitch `pitch`) from byte offset `off` of `base`, 5 raw dwords each, and the
        // byte-pair selectors; value k = interpolated pixel k with 5 fractional bits (descale by 9 of the four weighted taps)
        unsigned jr0[5], jr1[5], jsel[4];
        auto taps_read = [&](const uint8_t *base, unsigned off, int pitch) {
            const unsigned *q = reinterpret_cast<const unsigned *>(base + (off & ~3u));
#pragma unroll
            for (int i = 0; i < 5; ++i) { jr0[i] = q[i]; jr1[i] = q[i + pitch / 4]; }
            const unsigned shs = (off & 3u) | ((off & 3u) << 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) jsel[i] = LK_PAIR_SEL(i) + shs;
        };
        auto tap_value = [&](int k, unsigned W0_, unsigned W1_) {
            return lk_dot2(__builtin_amdgcn_perm(jr0[(k >> 2) + 1], jr0[k >> 2], jsel[k & 3]), W0_,
                           lkq_dot2_k(__builtin_amdgcn_perm(jr1[(k >> 2) + 1], jr1[k >> 2], jsel[k & 3]), W1_, 1 << 8)) >> 9;
        };
        auto J_read = [&](int ix_, int iy_, bool on) {
            taps_read(sJ, on && rowact ? __umul24((unsigned)(iy_ - jy0 + r), LKQ_JP) + (unsigned)(ix_ - jA) : 0u, LKQ_JP);
        };
        // previous-frame window row r: neighbourhood rows r+1, r+2 from column 1
        auto I_read = [&]() { taps_read(sI, (unsigned)((r + 1) * LKQ_IP + ((ipx - 1) & 3) + 1), LKQ_IP); };
        // ---- patch.  Neighbourhood rows r, r+1, r+2 -> derivative row r (16 columns); window row r = derivative rows r, r+1.
        int w00, w01, w10, w11;
        lk_weights(px - (float)ipx, py - (float)ipy, w00, w01, w10, w11);
        const unsigned W0 = (unsigned)w00 | ((unsigned)w01 << 16), W1 = (unsigned)w10 | ((unsigned)w11 << 16);
        unsigned a11 = 0, a22 = 0;
        int a12 = 0, c1 = 0, c2 = 0;
        {
            const unsigned ishs = (unsigned)((ipx - 1) & 3) * 0x00010001u;
            const unsigned sel0 = LK_PAIR_SEL(0) + ishs, sel1 = LK_PAIR_SEL(1) + ishs, sel2 = LK_PAIR_SEL(2) + ishs, sel3 = LK_PAIR_SEL(3) + ishs;
            unsigned R[3][6];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const unsigned *q = reinterpret_cast<const unsigned *>(sI + (r + j) * LKQ_IP);
#pragma unroll
                for (int i = 0; i < 6; ++i) R[j][i] = q[i];
            }
            // pair (n[c], n[c+1]) of neighbourhood row j: bytes c + ish, c + 1 + ish of the row's dwords
            auto pair = [&](int j, int c) {
                const unsigned sel = (c & 3) == 0 ? sel0 : (c & 3) == 1 ? sel1 : (c & 3) == 2 ? sel2 : sel3;
                return __builtin_amdgcn_perm(R[j][(c >> 2) + 1], R[j][c >> 2], sel);
            };
            lk_s2 VS[9], VD[9];
#pragma unroll
            for (int s = 0; s < 9; ++s) {
                const lk_s2 e0 = lk_as_s2(pair(0, 2 * s)), e1 = lk_as_s2(pair(1, 2 * s)), e2 = lk_as_s2(pair(2, 2 * s));
                VS[s] = (e0 + e2) * (short)3 + e1 * (short)10;
                VD[s] = e2 - e0;
            }
            unsigned DX[8], DY[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                DX[s] = lk_as_u(VS[s + 1] - VS[s]);
                const lk_s2 m = lk_as_s2(__builtin_amdgcn_alignbit(lk_as_u(VD[s + 1]), lk_as_u(VD[s]), 16));
                DY[s] = lk_as_u((VD[s] + VD[s + 1]) * (short)3 + m * (short)10);
            }
            // constant-0 border of the derivative image: taps (ipx + x, ipy + r) outside the level are zero
            if (__builtin_amdgcn_ballot_w64(lev && !(ipx >= 0 && ipx + win < lw && ipy >= 0 && ipy + win < lh)) != 0) {
                const int Y = ipy + r;
                const bool rowok = Y >= 0 && Y < lh;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int X = ipx + 2 * s;
                    const unsigned keep = ((rowok && X >= 0 && X < lw) ? 0x0000ffffu : 0u) | ((rowok && X + 1 >= 0 && X + 1 < lw) ? 0xffff0000u : 0u);
                    DX[s] &= keep; DY[s] &= keep;
                }
            }
            int ixp = 0, iyp = 0;
#pragma unroll
            for (int k = 0; k < 15; ++k) {
                const unsigned qx_ = (k & 1) ? __builtin_amdgcn_alignbit(DX[(k + 1) >> 1], DX[k >> 1], 16) : DX[k >> 1];
                const unsigned qy_ = (k & 1) ? __builtin_amdgcn_alignbit(DY[(k + 1) >> 1], DY[k >> 1], 16) : DY[k >> 1];
                // own derivative row with the upper weights; the lower-weight half comes from the lane below (row r + 1)
                const int hx = lkq_dot2_k(qx_, W0, 1 << 13), gx = lk_dot2(qx_, W1, 0);
                const int hy = lkq_dot2_k(qy_, W0, 1 << 13), gy = lk_dot2(qy_, W1, 0);
                const int ix = (hx + __builtin_amdgcn_update_dpp(0, gx, 0x101, 0xF, 0xF, true)) >> 14;     // row_shl:1
                const int iy = (hy + __builtin_amdgcn_update_dpp(0, gy, 0x101, 0xF, 0xF, true)) >> 14;
                if (k & 1) {
                    pxx[k >> 1] = __builtin_amdgcn_perm((unsigned)ix, (unsigned)ixp, 0x05040100u);
                    pyy[k >> 1] = __builtin_amdgcn_perm((unsigned)iy, (unsigned)iyp, 0x05040100u);
                } else if (k == 14) {
                    pxx[7] = (unsigned)ix & 0xffffu; pyy[7] = (unsigned)iy & 0xffffu;
                }
                ixp = ix; iyp = iy;
            }
            if (!(lev && rowact)) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { pxx[j] = 0u; pyy[j] = 0u; }
            }
            // |Ix|, |Iy| <= 4080: a lane's 15 squares sum below 2.5e8, so the signed dot products are exact
            int s11 = 0, s22 = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) { s11 = lk_dot2(pxx[j], pxx[j], s11); s22 = lk_dot2(pyy[j], pyy[j], s22); a12 = lk_dot2(pxx[j], pyy[j], a12); }
            a11 = (unsigned)s11; a22 = (unsigned)s22;
        }
        // c = sum(I * Ix), sum(I * Iy) over the lane's row: the constant part of the mismatch sums (pxy is zero on idle lanes)
        I_read();
#pragma unroll
        for (int j = 0; j < 8; ++j) {                            // interpolated values are 13-bit and non-negative: two per register
            const unsigned ivp = j < 7 ? (unsigned)tap_value(2 * j, W0, W1) | ((unsigned)tap_value(2 * j + 1, W0, W1) << 16) : (unsigned)tap_value(14, W0, W1);
            c1 = lk_dot2(ivp, pxx[j], c1); c2 = lk_dot2(ivp, pyy[j], c2);
        }
        const unsigned A11u = (unsigned)lkq_row_sum((int)a11), A22u = (unsigned)lkq_row_sum((int)a22);
        float A12 = (float)lkq_row_sum(a12) * 0x1p-20f;
        if (__builtin_amdgcn_ballot_w64(lev && max(A11u, A22u) >= 0x80000000u) != 0) {
            const float A12x = lkq_row_sum_split(a12);
            if (max(A11u, A22u) >= 0x80000000u) A12 = A12x;
        }
        const float A11 = (float)A11u * 0x1p-20f, A22 = (float)A22u * 0x1p-20f;
        const bool safe = A11u < LKQ_SAFE_LIM && A22u < LKQ_SAFE_LIM;
        float D = A11 * A22 - A12 * A12;
        const float dd = A11 - A22;
        const float minEig = (A22 + A11 - sqrtf(dd * dd + 4.f * A12 * A12)) / (float)(2 * ww);
        const bool solv = lev && !((double)minEig < min_eig_thr || D < FLT_EPSILON);
        if (lev && !solv && l == 0) st = 0;
        D = 1.f / D;

        bool act = solv;
        float pdx = 0.f, pdy = 0.f;
        for (int j = 0; j < max_count; ++j) {
            if (__builtin_amdgcn_ballot_w64(act) == 0) break;
            const int iqx = lkq_floor_i(qx), iqy = lkq_floor_i(qy);
            if (act && (iqx < -win || iqx >= lw || iqy < -win || iqy >= lh)) {
                if (l == 0) st = 0;
                act = false;
            }
            const bool need = act && (!jvalid || (unsigned)(iqx - jx0) > 2u * LK_M || (unsigned)(iqy - jy0) > 2u * LK_M);
            if (__builtin_amdgcn_ballot_w64(need) != 0) J_restage(iqx, iqy, need);
            lk_weights(qx - (float)iqx, qy - (float)iqy, w00, w01, w10, w11);
            const unsigned V0 = (unsigned)w00 | ((unsigned)w01 << 16), V1 = (unsigned)w10 | ((unsigned)w11 << 16);
            J_read(iqx, iqy, act);
            int b1 = -c1, b2 = -c2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned jvp = j < 7 ? (unsigned)tap_value(2 * j, V0, V1) | ((unsigned)tap_value(2 * j + 1, V0, V1) << 16) : (unsigned)tap_value(14, V0, V1);
                b1 = lk_dot2(jvp, pxx[j], b1); b2 = lk_dot2(jvp, pyy[j], b2);
            }
            float fb1 = (float)lkq_row_sum(b1) * 0x1p-20f, fb2 = (float)lkq_row_sum(b2) * 0x1p-20f;
            if (__builtin_amdgcn_ballot_w64(act && !safe) != 0) {
                const float x1 = lkq_row_sum_split(b1), x2 = lkq_row_sum_split(b2);
                if (!safe) { fb1 = x1; fb2 = x2; }
            }
            const float dx = (A12 * fb2 - A22 * fb1) * D, dy = (A12 * fb1 - A11 * fb2) * D;
            if (act) {
                qx += dx; qy += dy;
                nx = qx + half; ny = qy + half;
                const float d2 = dx * dx + dy * dy;
                bool done = d2 < eps2_lo;
                if (!done && d2 <= eps2_hi && (double)dx * (double)dx + (double)dy * (double)dy <= eps2) done = true;
                if (!done && j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
                    nx -= dx * 0.5f; ny -= dy * 0.5f;
                    done = true;
                }
                act = !done;
                pdx = dx; pdy = dy;
            }
        }
        if (l == 0) {
            bool eact = solv && st != 0;
            const float ex = nx - half, ey = ny - half;
            const int iex = lkq_floor_i(ex), iey = lkq_floor_i(ey);
            if (eact && (iex < -win || iex >= lw || iey < -win || iey >= lh)) { st = 0; eact = false; }
            if (__builtin_amdgcn_ballot_w64(eact) != 0) {
                const bool need = eact && (!jvalid || (unsigned)(iex - jx0) > 2u * LK_M || (unsigned)(iey - jy0) > 2u * LK_M);
                if (__builtin_amdgcn_ballot_w64(need) != 0) J_restage(iex, iey, need);
                lk_weights(ex - (float)iex, ey - (float)iey, w00, w01, w10, w11);
                const unsigned V0 = (unsigned)w00 | ((unsigned)w01 << 16), V1 = (unsigned)w10 | ((unsigned)w11 << 16);
                // the previous-frame values are recomputed from the staged neighbourhood (level 0 is still in s_I): keeping them in
                // registers through the Newton loop would cost 15 VGPRs for one use per point
                int pv[15];
                I_read();
#pragma unroll
                for (int k = 0; k < 15; ++k) pv[k] = tap_value(k, W0, W1);
                J_read(iex, iey, eact);
                int se = 0;
#pragma unroll
                for (int k = 0; k < 15; ++k) {
                    const int diff = tap_value(k, V0, V1) - pv[k];
                    se += diff < 0 ? -diff : diff;
                }
                if (!(eact && rowact)) se = 0;
                se = lkq_row_sum(se);
                if (eact) errv = (float)se / (float)(32 * ww);
            }
        }
    }
    if (live && r == 0) {
        next_pts[2 * pi] = nx; next_pts[2 * pi + 1] = ny;
        status[pi] = (uint8_t)st;
        err[pi] = st ? errv : 0.f;
    }
}


void ofk_launch_lk(hipStream_t s, const uint8_t *prev, const uint8_t *next, size_t pyr_stride, const ofk_levels &lv,
                   const float *prev_pts, const int *counts, int pts_stride, int win, int max_count, double eps,
                   double min_eig_thr, float *next_pts, uint8_t *status, float *err, int batch)
{
    if (max_count < 0) max_count = 0;
    if (max_count > 100) max_count = 100;
    if (eps < 0) eps = 0;
    if (eps > 10) eps = 10;
    const double eps2 = eps * eps;
    dim3 grid(pts_stride, batch);
    // four points per wave when the window is the reference's 15 x 15 and every level allows dword rows with ONE reflection
    // (staged columns reach 27 past a border, staged rows 23, and the 36-byte strip of a row has to fit: levels of at least 40 x 32)
    bool quad = win == 15 && (pyr_stride & 3) == 0;
    for (int l = 0; l <= lv.n; ++l) quad = quad && (lv.w[l] & 3) == 0 && lv.w[l] >= 40 && lv.h[l] >= 32 && (lv.off[l] & 3) == 0;
    if (quad)
        hipLaunchKernelGGL(k_lk15q, dim3((pts_stride + 3) / 4, batch), dim3(64), 0, s, prev, next, pyr_stride, lv, prev_pts, counts, pts_stride,
                           max_count, eps2, (float)(eps2 * (1.0 - 1e-5)), (float)(eps2 * (1.0 + 1e-5)), min_eig_thr, next_pts, status, err);
    else if (win <= 15)
        hipLaunchKernelGGL(k_lk15, grid, dim3(64), 0, s, prev, next, pyr_stride, lv, prev_pts, counts, pts_stride, win,
                           max_count, eps2, (float)(eps2 * (1.0 - 1e-5)), (float)(eps2 * (1.0 + 1e-5)), min_eig_thr, next_pts, status, err);
    else if (win <= 21)
        hipLaunchKernelGGL(k_lk<21>, grid, dim3(64), 0, s, prev, next, pyr_stride, lv, prev_pts, counts, pts_stride, win,
                           max_count, eps2, min_eig_thr, next_pts, status, err);
    else
        hipLaunchKernelGGL(k_lk<31>, grid, dim3(64), 0, s, prev, next, pyr_stride, lv, prev_pts, counts, pts_stride, win,
                           max_count, eps2, min_eig_thr, next_pts, status, err);
}
