// k_lk_quad.hip — pyramidal Lucas-Kanade, FOUR points per wavefront (win <= 15).  gfx950.
//
// EXPERIMENT, NOT PART OF libofk.so.  Measured on MI355X (1080p, 500 corners, B = 128): bit-identical to k_lk15 on the whole
// GPU test suite, 21 % fewer VALU instructions per point (2806 vs 3564, SQ_INSTS_VALU) — and 30 % SLOWER (0.96 vs 0.74 ms per
// 256 pairs): the 45 patch registers per lane push the kernel to 159 VGPRs = 3 waves per SIMD (2.7 resident on average),
// and at that occupancy a VALU instruction takes 5.9 clocks instead of 3.7 (the 2-clock issue of simple instructions needs
// several ready waves).  Capping the registers for a 4th wave spills 26 dwords.  Kept as a record of the design; to try it,
// add it to csrc/Makefile's SRCS and call ofk_launch_lk_quad from ofk_launch_lk (csrc/k_lk.hip) for win <= 15.
//
// k_lk15 (k_lk.hip) gives a point all 64 lanes; two thirds of its instructions are work that is the same in every lane
// (positions, weights, the 2x2 solve, reductions across the four 16-lane rows) or redundant (each lane rebuilds Scharr values
// its neighbours also build).  Here a point owns one 16-lane DPP row: lane r of the row is window row r (and derivative row
// r), its 15 pixels are a serial loop, and everything that is "per point" is computed once per row — four points share
// every instruction.  Reductions stay inside the row (row_sum16, no v_readlane), the derivative row r+1 a window row needs
// comes from the next lane by DPP row_shl:1, and the per-pixel arithmetic is the packed one of k_lk15 (v_perm byte pairs,
// v_pk_* Scharr, v_dot2_i32_i16 interpolation).  Every sum is an exact integer and every f32 expression is evaluated in the
// order of k_lk15 / oracle/image_oracle.c:orc_lk_pyr, so the results are bit-identical.
//
// Points of a wave converge after different numbers of Newton steps and may skip different pyramid levels: lanes of a
// finished point idle (predicates), the wave leaves a loop when none of its four points needs it.
#include "ofk_internal.h"
#include <float.h>

#define LQ_M 8                                     // margin of the staged next-frame region (pixels each side)
#define LQ_IP 24                                   // LDS pitch of the staged prev neighbourhood (18 x 18 used)
#define LQ_IROWS 20
#define LQ_JW 32                                   // staged next region: 32 x 32 = win + 1 + 2*LQ_M at win 15
#define LQ_JP 40                                   // its LDS pitch
#define LQ_PAIR_SEL(k) ((unsigned)(k) | 0x0c00u | ((unsigned)((k) + 1) << 16) | 0x0c000000u)   /* v_perm: (byte k, 0, byte k+1, 0) */
#define LQ_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

typedef short lq_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ lq_s2 lq_s(unsigned v) { return __builtin_bit_cast(lq_s2, v); }
__device__ __forceinline__ unsigned lq_u(lq_s2 v) { return __builtin_bit_cast(unsigned, v); }
__device__ __forceinline__ int lq_dot2(unsigned a, unsigned w, int acc) { return __builtin_amdgcn_sdot2(lq_s(a), lq_s(w), acc, false); }
__device__ __forceinline__ unsigned lq_next_lane(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xF, 0xF, true); }   // row_shl:1

__device__ __forceinline__ int lq_reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

__device__ __forceinline__ int lq_row_sum16(int v)     // every lane of a 16-lane row receives the row's sum
{
    v += __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true);      // quad_perm(1,0,3,2)
    v += __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true);      // quad_perm(2,3,0,1)
    v += __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true);     // row_half_mirror
    v += __builtin_amdgcn_update_dpp(0, v, 0x140, 0xF, 0xF, true);     // row_mirror
    return v;
}
// f32(S * 2^-20) of the exact row sum S of per-lane int32 partials (|S| may exceed int32: 16-bit halves are summed
// separately, S = HI * 65536 + LO exactly in f64) — the value k_lk15 forms from its int64 sum
__device__ __forceinline__ float lq_row_sum_scaled(int v)
{
    const int hi = lq_row_sum16(v >> 16), lo = lq_row_sum16(v & 0xffff);
    return (float)(((double)hi * 65536.0 + (double)lo) * 0x1p-20);
}
__device__ __forceinline__ long long lq_row_sum_i64(int v)
{
    const int hi = lq_row_sum16(v >> 16), lo = lq_row_sum16(v & 0xffff);
    return (long long)hi * 65536 + lo;
}

__device__ __forceinline__ void lq_weights(float a, float b, int &w00, int &w01, int &w10, int &w11)
{
    w00 = __float2int_rn((1.f - a) * (1.f - b) * 16384.f);
    w01 = __float2int_rn(a * (1.f - b) * 16384.f);
    w10 = __float2int_rn((1.f - a) * b * 16384.f);
    w11 = 16384 - w00 - w01 - w10;
}

// 16 consecutive bytes starting at byte offset `off` of an LDS byte array (4-byte aligned base) as the 8 even pairs
// E[s] = (t[2s], t[2s+1]) and the 7 odd pairs O[s] = (t[2s+1], t[2s+2])
__device__ __forceinline__ void lq_row_pairs16(const uint8_t *base, int off, unsigned E[8], unsigned O[7])
{
    const unsigned *p = reinterpret_cast<const unsigned *>(base + (off & ~3));
    const unsigned d0 = p[0], d1 = p[1], d2 = p[2], d3 = p[3], d4 = p[4];
    const unsigned sh = (unsigned)off & 3u;
    const unsigned a0 = __builtin_amdgcn_alignbyte(d1, d0, sh), a1 = __builtin_amdgcn_alignbyte(d2, d1, sh),
                   a2 = __builtin_amdgcn_alignbyte(d3, d2, sh), a3 = __builtin_amdgcn_alignbyte(d4, d3, sh);
    const unsigned a[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        E[2 * j] = __builtin_amdgcn_perm(0u, a[j], LQ_PAIR_SEL(0));
        E[2 * j + 1] = __builtin_amdgcn_perm(0u, a[j], LQ_PAIR_SEL(2));
    }
#pragma unroll
    for (int s = 0; s < 7; ++s) O[s] = __builtin_amdgcn_alignbit(E[s + 1], E[s], 16);
}

__global__ __launch_bounds__(64) void k_lk15q(const uint8_t *__restrict__ prev, const uint8_t *__restrict__ next, size_t pyr_stride,
                                              ofk_levels lv, const float *__restrict__ prev_pts, const int *__restrict__ counts,
                                              int pts_stride, int win, int max_count, double eps2, float eps2_lo, float eps2_hi,
                                              double min_eig_thr, float *__restrict__ next_pts, uint8_t *__restrict__ status,
                                              float *__restrict__ err)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_Iall[4][LQ_IROWS * LQ_IP];
    __shared__ __attribute__((aligned(16))) uint8_t s_Jall[4][(LQ_JW + 1) * LQ_JP];

    const int b = blockIdx.y, lane = threadIdx.x, q = lane >> 4, r = lane & 15;
    const int cnt = counts[b];
    if (4 * (int)blockIdx.x >= cnt) return;                     // whole wave
    const int p = 4 * blockIdx.x + q;
    const bool act = p < cnt;
    const size_t pi = (size_t)b * pts_stride + min(p, cnt - 1);
    const float ptx = prev_pts[2 * pi], pty = prev_pts[2 * pi + 1];
    const uint8_t *Pb = prev + (size_t)b * pyr_stride, *Nb = next + (size_t)b * pyr_stride;
    uint8_t *s_I = s_Iall[q], *s_J = s_Jall[q];
    const float half = (float)(win - 1) * 0.5f;
    const int ww = win * win;
    const int iw_ = win + 3, jw_ = win + 1 + 2 * LQ_M;
    const bool wrow = r < win;                                  // this lane is a window row

    int st = 1;
    float errv = 0.f, nx = 0.f, ny = 0.f;
    int pI[15], pIx[15], pIy[15];

    for (int l = lv.n; l >= 0; --l) {
        const int lh = lv.h[l], lw = lv.w[l];
        const uint8_t *I = Pb + lv.off[l], *J = Nb + lv.off[l];
        const float sc = __int_as_float((127 - l) << 23);       // 2^-l
        float px = ptx * sc, py = pty * sc, qx, qy;
        if (l == lv.n) { qx = px; qy = py; } else { qx = nx * 2.f; qy = ny * 2.f; }
        nx = qx; ny = qy;
        px -= half; py -= half;
        const int ipx = (int)floorf(px), ipy = (int)floorf(py);
        bool run = act && !(ipx < -win || ipx >= lw || ipy < -win || ipy >= lh);
        if (act && !run && l == 0) { st = 0; errv = 0.f; }
        if (!__any(run)) continue;
        qx -= half; qy -= half;
        int jx0 = 0, jy0 = 0;
        bool jvalid = false;

        // ---- staging of the next-frame region around (iqx, iqy) for the lanes with `need`: rows r and r + 16 of 32
        auto stage_J = [&](bool need, int iqx, int iqy) {
            if (need) { jx0 = iqx - LQ_M; jy0 = iqy - LQ_M; }
            const bool fastp = need && jx0 >= 4 && jx0 + jw_ + 8 <= lw && (lw & 3) == 0 && jw_ == LQ_JW;
            unsigned d[2][9];
            unsigned sh = 0;
            if (fastp) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const size_t addr = (size_t)lq_reflect101(jy0 + r + 16 * h2, lh) * lw + jx0;
                    sh = (unsigned)addr & 3u;                   // the same for both rows: lw is a multiple of 4
                    const unsigned *g = reinterpret_cast<const unsigned *>(J + (addr & ~(size_t)3));
#pragma unroll
                    for (int k = 0; k < 9; ++k) d[h2][k] = g[k];
                }
            }
            LQ_FENCE();                                         // earlier readers of s_J are done
            if (fastp) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    uint2 *dst = reinterpret_cast<uint2 *>(s_J + (r + 16 * h2) * LQ_JP);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        dst[k] = make_uint2(__builtin_amdgcn_alignbyte(d[h2][2 * k + 1], d[h2][2 * k], sh),
                                            __builtin_amdgcn_alignbyte(d[h2][2 * k + 2], d[h2][2 * k + 1], sh));
                }
            }
            if (__any(need && !fastp)) {                        // byte path: regions that cross the left/right border, other window sizes
                for (int i = r; i < jw_ * jw_; i += 16) {
                    const int rr = i / jw_, cc = i - rr * jw_;
                    if (need && !fastp) s_J[rr * LQ_JP + cc] = J[(size_t)lq_reflect101(jy0 + rr, lh) * lw + lq_reflect101(jx0 + cc, lw)];
                }
            }
            LQ_FENCE();
            if (need) jvalid = true;
        };

        // ---- staging of the prev neighbourhood (origin ipx-1, ipy-1), (win+3)^2, and of the next-frame region; the global
        //      loads of both are in flight together
        const int iqx0 = (int)floorf(qx), iqy0 = (int)floorf(qy);
        const bool doJ = run && !(iqx0 < -win || iqx0 >= lw || iqy0 < -win || iqy0 >= lh);
        {
            const bool fastI = run && ipx >= 5 && ipx - 1 + 28 <= lw && (lw & 3) == 0;
            unsigned d[2][6];
            unsigned sh = 0;
            if (fastI) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int row = r + 16 * h2;                // rows 16, 17 by lanes 0, 1 (clamped for the others: not stored)
                    const size_t addr = (size_t)lq_reflect101(ipy - 1 + min(row, iw_ - 1), lh) * lw + (ipx - 1);
                    sh = (unsigned)addr & 3u;
                    const unsigned *g = reinterpret_cast<const unsigned *>(I + (addr & ~(size_t)3));
#pragma unroll
                    for (int k = 0; k < 6; ++k) d[h2][k] = g[k];
                }
            }
            LQ_FENCE();                                         // the previous level's readers of s_I are done
            if (fastI) {
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int row = r + 16 * h2;
                    if (row < iw_) {
                        unsigned *o = reinterpret_cast<unsigned *>(s_I + row * LQ_IP);
#pragma unroll
                        for (int k = 0; k < 5; ++k) o[k] = __builtin_amdgcn_alignbyte(d[h2][k + 1], d[h2][k], sh);
                    }
                }
            }
            if (__any(run && !fastI)) {
                for (int i = r; i < iw_ * iw_; i += 16) {
                    const int rr = i / iw_, cc = i - rr * iw_;
                    if (run && !fastI) s_I[rr * LQ_IP + cc] = I[(size_t)lq_reflect101(ipy - 1 + rr, lh) * lw + lq_reflect101(ipx - 1 + cc, lw)];
                }
            }
            stage_J(doJ, iqx0, iqy0);                           // ends with a fence: s_I and s_J are readable
        }

        // ---- patch: I (5 fractional bits), Ix, Iy of the lane's window row; exact integer normal matrix
        int w00, w01, w10, w11;
        lq_weights(px - (float)ipx, py - (float)ipy, w00, w01, w10, w11);
        int a11 = 0, a12 = 0, a22 = 0;
        if (run) {
            // rows r, r+1, r+2 of the neighbourhood as even pairs E[s] = (b[2s], b[2s+1]), s = 0..8, and odd pairs O[s] = (b[2s+1], b[2s+2])
            unsigned E[3][9], O[3][8];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const unsigned *pr = reinterpret_cast<const unsigned *>(s_I + min(r + k, LQ_IROWS - 1) * LQ_IP);
                unsigned dw[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) dw[j] = pr[j];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    E[k][2 * j] = __builtin_amdgcn_perm(0u, dw[j], LQ_PAIR_SEL(0));
                    E[k][2 * j + 1] = __builtin_amdgcn_perm(0u, dw[j], LQ_PAIR_SEL(2));
                }
                E[k][8] = __builtin_amdgcn_perm(0u, dw[4], LQ_PAIR_SEL(0));
#pragma unroll
                for (int s = 0; s < 8; ++s) O[k][s] = __builtin_amdgcn_alignbit(E[k][s + 1], E[k][s], 16);
            }
            // Scharr of derivative row r, two columns per instruction: slot s = columns (2s, 2s+1), s = 0..7
            unsigned DX[8], DY[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                lq_s2 hd[3], hs[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const lq_s2 a = lq_s(E[k][s]), c2 = lq_s(E[k][s + 1]), m = lq_s(O[k][s]);
                    hd[k] = c2 - a;
                    hs[k] = (a + c2) * (short)3 + m * (short)10;
                }
                DX[s] = lq_u((hd[0] + hd[2]) * (short)3 + hd[1] * (short)10);
                DY[s] = lq_u(hs[2] - hs[0]);
            }
            const bool touch = !(ipx >= 0 && ipx + win < lw && ipy >= 0 && ipy + win < lh);     // the window touches the border
            if (__any(run && touch)) {
                const int Y = ipy + r;
                const bool rowok = Y >= 0 && Y < lh;
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    const int X = ipx + 2 * s;
                    const unsigned keep = ((rowok && X >= 0 && X < lw) ? 0x0000ffffu : 0u) | ((rowok && X + 1 >= 0 && X + 1 < lw) ? 0xffff0000u : 0u);
                    if (touch) { DX[s] &= keep; DY[s] &= keep; }               // constant-0 derivative border
                }
            }
            // derivative row r+1 from the next lane of the row
            unsigned DXn[8], DYn[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) { DXn[s] = lq_next_lane(DX[s]); DYn[s] = lq_next_lane(DY[s]); }
            const unsigned W0 = (unsigned)w00 | ((unsigned)w01 << 16), W1 = (unsigned)w10 | ((unsigned)w11 << 16);
            if (wrow) {
#pragma unroll
                for (int x = 0; x < 15; ++x)
                    if (x < win) {
                        // pairs starting at column x: derivative (x, x+1); intensity (x+1, x+2) of rows r+1, r+2
                        const unsigned qx0 = (x & 1) ? __builtin_amdgcn_alignbit(DX[(x + 1) / 2], DX[x / 2], 16) : DX[x / 2];
                        const unsigned qx1 = (x & 1) ? __builtin_amdgcn_alignbit(DXn[(x + 1) / 2], DXn[x / 2], 16) : DXn[x / 2];
                        const unsigned qy0 = (x & 1) ? __builtin_amdgcn_alignbit(DY[(x + 1) / 2], DY[x / 2], 16) : DY[x / 2];
                        const unsigned qy1 = (x & 1) ? __builtin_amdgcn_alignbit(DYn[(x + 1) / 2], DYn[x / 2], 16) : DYn[x / 2];
                        const unsigned i1 = ((x + 1) & 1) ? O[1][x / 2] : E[1][(x + 1) / 2], i2 = ((x + 1) & 1) ? O[2][x / 2] : E[2][(x + 1) / 2];
                        const int iv = lq_dot2(i1, W0, lq_dot2(i2, W1, 1 << 8)) >> 9;
                        const int ix = lq_dot2(qx0, W0, lq_dot2(qx1, W1, 1 << 13)) >> 14;
                        const int iy = lq_dot2(qy0, W0, lq_dot2(qy1, W1, 1 << 13)) >> 14;
                        pI[x] = iv; pIx[x] = ix; pIy[x] = iy;
                        a11 += __mul24(ix, ix); a12 += __mul24(ix, iy); a22 += __mul24(iy, iy);
                    }
            }
        }
        const float A11 = lq_row_sum_scaled(a11), A12 = lq_row_sum_scaled(a12), A22 = lq_row_sum_scaled(a22);
        float D = A11 * A22 - A12 * A12;
        const float dd = A11 - A22;
        const float minEig = (A22 + A11 - sqrtf(dd * dd + 4.f * A12 * A12)) / (float)(2 * ww);
        bool it = run && !((double)minEig < min_eig_thr || D < FLT_EPSILON);
        if (run && !it && l == 0) st = 0;
        D = 1.f / D;
        float pdx = 0.f, pdy = 0.f;
        bool on = it;                                           // this point still iterates
        for (int j = 0; j < max_count; ++j) {
            if (!__any(on)) break;
            const int iqx = (int)floorf(qx), iqy = (int)floorf(qy);
            if (on && (iqx < -win || iqx >= lw || iqy < -win || iqy >= lh)) {
                if (l == 0) st = 0;
                on = false;
            }
            const bool need = on && (!jvalid || iqx < jx0 || iqx > jx0 + 2 * LQ_M || iqy < jy0 || iqy > jy0 + 2 * LQ_M);
            if (__any(need)) stage_J(need, iqx, iqy);
            int v00, v01, v10, v11;
            lq_weights(qx - (float)iqx, qy - (float)iqy, v00, v01, v10, v11);
            int b1 = 0, b2 = 0;
            if (on && wrow) {
                unsigned E0[8], O0[7], E1[8], O1[7];
                const int off = (iqy - jy0 + r) * LQ_JP + (iqx - jx0);
                lq_row_pairs16(s_J, off, E0, O0); lq_row_pairs16(s_J, off + LQ_JP, E1, O1);
                const unsigned W0 = (unsigned)v00 | ((unsigned)v01 << 16), W1 = (unsigned)v10 | ((unsigned)v11 << 16);
#pragma unroll
                for (int x = 0; x < 15; ++x)
                    if (x < win) {
                        const unsigned t0 = (x & 1) ? O0[x / 2] : E0[x / 2], t1 = (x & 1) ? O1[x / 2] : E1[x / 2];
                        const int diff = (lq_dot2(t0, W0, lq_dot2(t1, W1, 1 << 8)) >> 9) - pI[x];
                        b1 += __mul24(diff, pIx[x]); b2 += __mul24(diff, pIy[x]);
                    }
            }
            const float fb1 = lq_row_sum_scaled(b1), fb2 = lq_row_sum_scaled(b2);
            const float dx = (A12 * fb2 - A22 * fb1) * D, dy = (A12 * fb1 - A11 * fb2) * D;
            if (on) {
                qx += dx; qy += dy;
                nx = qx + half; ny = qy + half;
                const float d2 = dx * dx + dy * dy;
                if (d2 < eps2_lo || (d2 <= eps2_hi && (double)dx * (double)dx + (double)dy * (double)dy <= eps2)) on = false;
                else if (j > 0 && fabsf(dx + pdx) <= 0.01f && fabsf(dy + pdy) <= 0.01f) {
                    nx -= dx * 0.5f; ny -= dy * 0.5f;
                    on = false;
                }
                pdx = dx; pdy = dy;
            }
        }
        if (l == 0) {
            bool ev = it && st != 0;
            const float ex = nx - half, ey = ny - half;
            const int iex = (int)floorf(ex), iey = (int)floorf(ey);
            if (ev && (iex < -win || iex >= lw || iey < -win || iey >= lh)) { st = 0; ev = false; }
            const bool need = ev && (!jvalid || iex < jx0 || iex > jx0 + 2 * LQ_M || iey < jy0 || iey > jy0 + 2 * LQ_M);
            if (__any(need)) stage_J(need, iex, iey);
            if (__any(ev)) {
                int v00, v01, v10, v11;
                lq_weights(ex - (float)iex, ey - (float)iey, v00, v01, v10, v11);
                int se = 0;
                if (ev && wrow) {
                    unsigned E0[8], O0[7], E1[8], O1[7];
                    const int off = (iey - jy0 + r) * LQ_JP + (iex - jx0);
                    lq_row_pairs16(s_J, off, E0, O0); lq_row_pairs16(s_J, off + LQ_JP, E1, O1);
                    const unsigned W0 = (unsigned)v00 | ((unsigned)v01 << 16), W1 = (unsigned)v10 | ((unsigned)v11 << 16);
#pragma unroll
                    for (int x = 0; x < 15; ++x)
                        if (x < win) {
                            const unsigned t0 = (x & 1) ? O0[x / 2] : E0[x / 2], t1 = (x & 1) ? O1[x / 2] : E1[x / 2];
                            const int diff = (lq_dot2(t0, W0, lq_dot2(t1, W1, 1 << 8)) >> 9) - pI[x];
                            se += diff < 0 ? -diff : diff;
                        }
                }
                const long long SE = lq_row_sum_i64(se);
                if (ev) errv = (float)(int)SE / (float)(32 * ww);
            }
        }
    }
    if (act && r == 0) {
        next_pts[2 * pi] = nx; next_pts[2 * pi + 1] = ny;
        status[pi] = (uint8_t)st;
        err[pi] = st ? errv : 0.f;
    }
}

void ofk_launch_lk_quad(hipStream_t s, const uint8_t *prev, const uint8_t *next, size_t pyr_stride, const ofk_levels &lv,
                        const float *prev_pts, const int *counts, int pts_stride, int win, int max_count, double eps2,
                        double min_eig_thr, float *next_pts, uint8_t *status, float *err, int batch)
{
    dim3 grid((pts_stride + 3) / 4, batch);
    hipLaunchKernelGGL(k_lk15q, grid, dim3(64), 0, s, prev, next, pyr_stride, lv, prev_pts, counts, pts_stride, win, max_count, eps2,
                       (float)(eps2 * (1.0 - 1e-5)), (float)(eps2 * (1.0 + 1e-5)), min_eig_thr, next_pts, status, err);
}
