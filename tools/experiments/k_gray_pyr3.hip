// k_gray_pyr3.hip — EXPERIMENT, NOT PART OF libofk.so: gray conversion fused into the three-level pyramid pass (k_pyr3_stream fed
// from the BGR frames), so that level 0 is never read back.  Bit-identical to k_gray_bgr8 + k_pyr3_stream (resident pyramids
// compared byte for byte, tests/test_gpu_pipeline.py::test_resident_pyramids_after_a_pairs_run, all GPU tests green), and measured
// on MI355X at B = 256 x 1080p pairs:
//   alone on the chip   0.984 ms  (k_gray_bgr8 0.766 + k_pyr3_stream 0.312 = 1.078 ms): 5.1 TB/s against gray's 5.6 — the march
//                       re-reads 9-14 % of the BGR bytes (strip halo lanes, chunk warm-up rows) and keeps 8 rows x 48 B per lane
//                       in flight at 166 VGPRs;
//   in the pipeline     2.296 ms per step (111.5 k pairs/s) against 2.249 ms (113.8 k) with the separate kernels; with 4 rows in
//                       flight (118 VGPRs) 2.491 ms.  The 130 000 tiny workgroups of k_gray_bgr8 fill the register-file gaps the
//                       VALU-bound kernels leave; 2048 long-lived 166-register waves do not.
// To try it again: paste into csrc/k_image.hip behind k_pyr3_stream (it uses its p3_* helpers), declare ofk_launch_gray_pyr3 in
// ofk_internal.h and call it from ofk_pairs_run / stream_ingest in place of ofk_launch_gray + ofk_launch_pyr3.

// ------------------------------------------------------------------------------------------------ gray + three pyramid levels in one pass
// k_gray_pyr3: k_pyr3_stream fed from the BGR frames.  The march is the same (lane = 16 columns, a wave walks down a strip, levels
// 1..3 from register rings); a level-0 row arrives as 48 BGR bytes per lane (three dwordx4), is converted with k_gray_bgr8's
// arithmetic and stored once — by the chunk whose own steps consume it — and the 16 gray bytes go straight into the level-1 filter.
// Level 0 is then never read back: 3 (+ 9-14 % strip halo and chunk warm-up) + 1 + 0.33 bytes per pixel instead of 3 + 1 + 1 + 0.33.
struct gp3_args {
    const uint8_t *bgr0, *bgr1;         // BGR frames of the two frame sets
    size_t bgr_stride;
    uint8_t *base0, *base1;             // pyramid slabs (level 0 = gray at offset 0)
    size_t stride, off1, off2, off3;
    int h, w, batch, steps, nstrips, nchunks;
};
struct gp3_px16 { uint4 a, c, d; };     // 16 BGR pixels

__device__ __forceinline__ uint4 gp3_gray16(const gp3_px16 &v)
{
    constexpr unsigned HI = 29u | (150u << 8) | (76u << 16), LO = 46u | (70u << 8) | (140u << 16);       // see k_gray_bgr8
    const unsigned wv[13] = {v.a.x, v.a.y, v.a.z, v.a.w, v.c.x, v.c.y, v.c.z, v.c.w, v.d.x, v.d.y, v.d.z, v.d.w, 0u};
    unsigned t[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int byte = 3 * k;
        const unsigned px = (byte & 3) ? __builtin_amdgcn_alignbit(wv[(byte >> 2) + 1], wv[byte >> 2], (byte & 3) * 8) : wv[byte >> 2];
        t[k] = (__builtin_amdgcn_udot4(px, HI, 0u, false) << 8) + __builtin_amdgcn_udot4(px, LO, 32768u, false);
    }
    unsigned out[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        out[q] = __builtin_amdgcn_perm(t[4 * q + 1], t[4 * q], 0x0c0c0602u) | __builtin_amdgcn_perm(t[4 * q + 3], t[4 * q + 2], 0x06020c0cu);
    return make_uint4(out[0], out[1], out[2], out[3]);
}

template <int NQ>                                               // level-0 rows in flight per wave (4 or 8)
__global__ __launch_bounds__(256) void k_gray_pyr3(gp3_args A)
{
    const int lane = threadIdx.x & 63;
    int bx = blockIdx.x, z = blockIdx.y;
    if ((gridDim.y & 7) == 0) {                                 // XCD-aware: an image's chunks on one XCD (see k_pyr_down_stream)
        const unsigned n = blockIdx.y * gridDim.x + blockIdx.x, k = n >> 3;
        z = 8 * (int)(k / gridDim.x) + (int)(n & 7);
        bx = (int)(k % gridDim.x);
    }
    const int wid = __builtin_amdgcn_readfirstlane(bx * 4 + (threadIdx.x >> 6));
    if (wid >= A.nstrips * A.nchunks) return;
    const int strip = wid % A.nstrips, chunk = wid / A.nstrips;
    const int h = A.h, w = A.w, n1 = h >> 1, n2 = h >> 2, n3 = h >> 3, w1 = w >> 1, w2 = w >> 2, w3 = w >> 3;
    const uint8_t *src = z < A.batch ? A.bgr0 + (size_t)z * A.bgr_stride : A.bgr1 + (size_t)(z - A.batch) * A.bgr_stride;
    uint8_t *slab = z < A.batch ? A.base0 + (size_t)z * A.stride : A.base1 + (size_t)(z - A.batch) * A.stride;
    const int hl = strip == 0 ? 0 : 3, hr = strip == A.nstrips - 1 ? 0 : 3;
    const int cs = strip == 0 ? 0 : 16 * (61 + 58 * (strip - 1));           // first payload column of the strip
    const int c0 = cs - 16 * hl + 16 * lane;
    const int sc = min(max(c0, 0), w - 16);
    const bool first = c0 == 0, last = c0 == w - 16;
    const bool st = lane >= hl && lane <= 63 - hr && c0 < w;
    const int T0 = chunk * A.steps, T1 = T0 + A.steps;
    auto ld = [&](int sy) -> gp3_px16 {
        sy = sy < 0 ? -sy : sy;
        sy = sy >= h ? 2 * (h - 1) - sy : sy;
        const uint4 *p = reinterpret_cast<const uint4 *>(src + (size_t)__builtin_amdgcn_readfirstlane(max(sy, 0)) * (3u * (unsigned)w) + 3u * (unsigned)sc);
        return gp3_px16{p[0], p[1], p[2]};
    };
    // gray row sy from its BGR bytes; stored when this chunk owns the row (its own steps consume it; row 0 belongs to chunk 0's warm-up)
    auto use = [&](const gp3_px16 &raw, int sy, bool own) -> uint4 {
        const uint4 g = gp3_gray16(raw);
        if (st && own && sy >= 0 && sy < h) *reinterpret_cast<uint4 *>(slab + (size_t)sy * w + c0) = g;
        return g;
    };
    const int ts = T0 - 12;
    unsigned a[4], b[4], c[4];
    p3_h16(use(ld(2 * ts - 2), 2 * ts - 2, false), first, last, a);
    p3_h16(use(ld(2 * ts - 1), 2 * ts - 1, false), first, last, b);
    p3_h16(use(ld(2 * ts), 2 * ts, false), first, last, c);
    gp3_px16 q[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) q[k] = ld(2 * ts + 1 + k);
    unsigned r1x[5] = {0, 0, 0, 0, 0}, r1y[5] = {0, 0, 0, 0, 0}, r2[5] = {0, 0, 0, 0, 0};
    for (int t = ts; t < T1; t += 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int tt = t + k;
            const bool own = tt >= T0;                          // warm-up steps store nothing (but level-0 row 0, see below)
            unsigned d[4], e[4], v[4];
            p3_h16(use(q[(2 * k) % NQ], 2 * tt + 1, own), first, last, d);
            p3_h16(use(q[(2 * k + 1) % NQ], 2 * tt + 2, own || (T0 == 0 && tt == -1)), first, last, e);
            q[(2 * k) % NQ] = ld(2 * tt + 1 + NQ);
            q[(2 * k + 1) % NQ] = ld(2 * tt + 2 + NQ);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = p3_vert(a[j], b[j], c[j], d[j], e[j]); a[j] = c[j]; b[j] = d[j]; c[j] = e[j]; }
            if (st && own && tt < n1)
                *reinterpret_cast<uint2 *>(slab + A.off1 + (size_t)tt * w1 + (c0 >> 1)) = make_uint2(P3_BYTES(v[1], v[0]), P3_BYTES(v[3], v[2]));
            const unsigned t0 = (v[0] >> 8) & P3_M, t1 = (v[1] >> 8) & P3_M, t2 = (v[2] >> 8) & P3_M, t3 = (v[3] >> 8) & P3_M;
            unsigned g0, g1;
            p3_h8(P3_EVEN(t1, t0), P3_ODD(t1, t0), P3_EVEN(t3, t2), P3_ODD(t3, t2), first, last, g0, g1);
#pragma unroll
            for (int j = 0; j < 4; ++j) { r1x[j] = r1x[j + 1]; r1y[j] = r1y[j + 1]; }
            r1x[4] = g0; r1y[4] = g1;
            if ((k & 1) == 0) {                                 // even step: level-2 row (tt - 2) / 2 from level-1 rows tt-4 .. tt
                const bool m1 = tt == n1;
                const unsigned w0 = p3_vert(r1x[0], r1x[1], r1x[2], r1x[3], m1 ? r1x[2] : r1x[4]);
                const unsigned w1v = p3_vert(r1y[0], r1y[1], r1y[2], r1y[3], m1 ? r1y[2] : r1y[4]);
                const int qrow = (tt - 2) >> 1;
                if (st && own && tt >= 2 && qrow < n2)
                    *reinterpret_cast<unsigned *>(slab + A.off2 + (size_t)qrow * w2 + (c0 >> 2)) = P3_BYTES(w1v, w0);
                const unsigned s0 = (w0 >> 8) & P3_M, s1 = (w1v >> 8) & P3_M;
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

// gray level + levels 1..3 of both frame sets from the BGR frames in one launch; false if the geometry does not fit
bool ofk_launch_gray_pyr3(hipStream_t s, const uint8_t *bgr0, const uint8_t *bgr1, size_t bgr_stride, uint8_t *pyr0, uint8_t *pyr1,
                          size_t stride, const ofk_levels &lv, int batch, int images)
{
    const int h = lv.h[0], w = lv.w[0];
    if (lv.n < 3 || !pyr3_ok(h, w) || (bgr_stride & 15) != 0 || getenv("OFK_NO_GRAY_PYR3") != nullptr) return false;
    const int lanes = w / 16;
    int nstrips = 1;
    if (lanes > 64) { nstrips = 2; while (61 * 2 + 58 * (nstrips - 2) < lanes) ++nstrips; }
    const int total = ((h >> 1) + 3 + 3) / 4 * 4;
    int nchunks = (2048 + images * nstrips - 1) / (images * nstrips);
    const int maxchunks = total / 32 > 1 ? total / 32 : 1;
    nchunks = nchunks < 1 ? 1 : nchunks > maxchunks ? maxchunks : nchunks;
    if (const char *e = getenv("OFK_PYR3_CHUNKS")) { const int v = atoi(e); if (v >= 1 && v <= maxchunks) nchunks = v; }   // tuning knob
    int steps = ((total + nchunks - 1) / nchunks + 3) / 4 * 4;
    nchunks = (total + steps - 1) / steps;
    gp3_args A = {bgr0, bgr1 ? bgr1 : bgr0, bgr_stride, pyr0, pyr1 ? pyr1 : pyr0, stride, lv.off[1], lv.off[2], lv.off[3], h, w, batch, steps, nstrips, nchunks};
    dim3 grid((nstrips * nchunks + 3) / 4, images);
    if (getenv("OFK_GP3_NQ4")) hipLaunchKernelGGL(k_gray_pyr3<4>, grid, dim3(256), 0, s, A);
    else hipLaunchKernelGGL(k_gray_pyr3<8>, grid, dim3(256), 0, s, A);
    return true;
}
