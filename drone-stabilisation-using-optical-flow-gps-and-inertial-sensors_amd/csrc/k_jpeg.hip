// k_jpeg.hip — baseline JPEG -> BGR8 on the device: the ingest in front of the hot path (the reference decodes
// every sensor_msgs/CompressedImage with cv_bridge.compressed_imgmsg_to_cv2 = cv::imdecode = libjpeg,
// velocity_measurment_node.py:112).  Output is bit-identical to libjpeg's default decompressor (ISLOW integer IDCT, "fancy"
// triangle chroma upsampling, 16-bit fixed-point YCbCr->RGB).  The host parses the marker segments, packs the tables and takes the
// byte stuffing out of the entropy segments while it copies them (jdestuff: memchr + memcpy); entropy decoding, IDCT, upsampling
// and colour conversion run on the GPU.
//
// Huffman decoding is a serial bit-by-bit process per image.  It is parallelised by *self-synchronisation* (Klein & Wiseman 2003;
// Weissenberger & Schmidt, "Massively parallel Huffman decoding on GPUs", ICPP 2018, and their 2021 JPEG follow-up): the entropy
// segment is cut into chunks of `jch` bytes, one decoder thread per chunk.  A decoder that starts at a wrong bit position / block
// position produces garbage for a while but, because Huffman codes are prefix codes, falls into step with the true symbol sequence
// after a few dozen symbols with overwhelming probability.
//   1. k_jpeg_sync, iteration 0: every thread decodes its chunk from a guessed state (chunk start, block start) and publishes the
//      state in which it crossed into the next chunk; iteration n > 0: every thread whose predecessor published a different state
//      than the one it started from last time decodes its chunk again from that state.  Chunk 0 starts from the true state, so the
//      truth advances at least one chunk per iteration; in practice half of the chunks agree after two iterations and the rest
//      after four to six.  An iteration in which no thread changed its published state is a fixed point, and a fixed point that
//      starts from the true state is the true decode (induction over the chunks) — the result never depends on the probabilistic
//      argument, only the run time does.
//   2. k_jpeg_scan: exclusive prefix sum of the blocks completed per chunk = index of the block a chunk starts in.
//   3. k_jpeg_write: every thread decodes its chunk once more from its (now true) entry state and stores the coefficients
//      (k_jpeg_zero_heads zeroes the few blocks that more than one thread writes).
//   4. k_jpeg_dc: DC prediction = prefix sum of the DC differences per component in decode order.
//   5. k_jpeg_idct<false>: dequantisation + ISLOW IDCT of the chroma blocks, 8 lanes per block (columns, then rows through LDS) -> planes.
//   6. k_jpeg_idct<true>: the same for the luma blocks, into an LDS tile, then fancy upsampling + colour conversion -> BGR8 (or straight
//      to gray for the pipeline).
// The decoder state at a symbol boundary is (bit position in the destuffed segment, zigzag index, block-in-MCU): a function of
// the true bit position only, so two decoders that agree on a symbol boundary agree on everything that follows.  A symbol belongs
// to the chunk its first bit lies in.  Restart markers are taken out by the host and listed as boundaries, see jrun.
#include "ofk_internal.h"
#include <string.h>
#include <stdlib.h>
#include <atomic>
#include <thread>
#include <vector>

#define JCH 256                 // entropy bytes per decoder thread when there is enough data; halved down to JCH_MIN for small
#define JCH_MIN 64              // batches (a single 1080p frame: 7000 threads of 64 bytes), where latency counts, not throughput
#define JCH_MAX 1024            // largest chunk ofk_set_tuning("jpeg_chunk") may ask for: the padding is sized for it
#define JTPB 256                // decoder threads per workgroup (chunks of ONE image: the tables live in LDS)
#define JTPW 256                // ... of the coefficient-writing pass (its LDS rows bound the occupancy)
#define JMAX_ITERS 64           // flag slots; more iterations than this are read back one by one
#define JNSUB 12                // second-level tables an image's Huffman tables may use (the standard tables of T.81 Annex K need 11)

// Look-up tables of the entropy decoder.  First level: the next 9 bits of the stream; second level (codes of 10..16 bits): the 7
// bits behind them.  The same tables in two encodings, one per kind of pass, so that a symbol step is one LDS read and two field
// extractions (before: code length and symbol, then run / size / EOB / ZRL told apart by three branches):
//   S (synchronisation passes):  (len + s) << 8 | dk        bits the symbol takes with its extra bits; advance of the zigzag index
//   W (write pass):              (len - 1) << 11 | s << 7 | dk
// dk = run + 1 for a coefficient, 16 for ZRL, 64 for EOB (any k + 64 ends the block), 1 for a DC symbol; a prefix that is no code
// reads as a 16-bit EOB (DC: a 16-bit zero difference), which is what jdhuff.c makes of it.  0x8000 | n = second-level table n;
// 0xFFFF = the image's tables need more than JNSUB second-level tables: canonical search (jslow) for this prefix.
struct jfast { uint16_t lut[6][512]; uint16_t sub[JNSUB][128]; };    // slot = 2 * component + (AC ? 1 : 0)

struct jpeg_tab {               // per image
    jfast S, W;
    int32_t maxcode[6][18];     // largest code of length l (-1: none), [17] = INT_MAX        (canonical tables: the slow path)
    int32_t valoff[6][18];      // vals index of the first code of length l minus that code
    uint8_t vals[6][256];
    uint16_t q[3][64];          // natural order
    uint32_t ent_off, ent_len;  // entropy segment (destuffed by the host) inside the batch's entropy buffer
    int32_t nch, ri;            // chunks of this image; restart interval in MCUs (0 = none)
    uint32_t rst_off, nrst;     // restart boundaries: nrst byte offsets into the destuffed segment, from rst[rst_off] on
};

struct jpeg_geom {
    int w, h, ncomp, hmax, vmax, mcux, mcuy, bpm, nblk, jch;  // jch: bytes per decoder thread of this batch
    int blk_comp[10];           // component of block j of an MCU
    int comp_off[3], comp_nb[3];
    int pw[3], ph[3];           // plane sizes (whole MCUs)
    size_t plane_off[3], plane_stride;   // bytes between images in the plane buffer
};

__constant__ uint8_t c_izz[64] = {0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53, 10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};      // natural position -> zigzag index
static const uint8_t h_zz[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                                 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// the entry of a symbol whose code has `len` bits (both encodings; the host builds the tables with it, the device its slow path)
__host__ __device__ inline uint32_t jentry(bool write, bool ac, int len, int sym)
{
    const int s = sym & 15, run = sym >> 4;
    const int dk = !ac ? 1 : s ? run + 1 : run == 15 ? 16 : 64;
    return write ? (uint32_t)(((len - 1) << 11) | (s << 7) | dk) : (uint32_t)(((len + s) << 8) | dk);
}

// ------------------------------------------------------------------------------------------------ device: entropy decoder
// The host hands over DESTUFFED entropy segments (FF00 -> FF, RSTn markers taken out and listed), so the decoder's position is a
// plain bit offset and its state at a symbol boundary (bit position, zigzag index, block in MCU) is canonical by construction.
// Every segment starts 256-aligned and is zero-padded up to JPAD(len): no bounds checks.
#define JPAD(len) (((size_t)(len) + 2 * JCH_MAX + 19) / JCH * JCH)

// The reader: a ring of JRING stream dwords per thread in LDS, refilled at wave-uniform times.  A symbol step needs the 64 stream
// bits at the cursor (big-endian dwords w, w + 1: the next 32 bits are one 64-bit shift away - enough for the longest code, 16, plus
// the longest run of extra bits, 15).  History: a byte load per symbol put a memory round trip into every step (83 % of the wave
// cycles waiting); a register window (hi:lo + two read-ahead dwords, the next dword requested at every slide) took the thread's own
// loads off its chain but not the WAVE's - loads and waits are counted per wave, so every slide of any lane waits for the youngest
// load of all 64, some lane slides in almost every step, and each lane walks through memory of its own: an L2 or HBM round trip in
// nearly every symbol step (2300 clocks per step at 8 waves per SIMD; the pass took as long with 35 instructions per symbol as with
// 90).  Here every JREFILL steps ALL lanes ask for the 16 bytes behind their ring and take in, if there is room, what they asked
// for JREFILL steps ago: one wait per JREFILL steps, for a load that old.  A symbol step reads its 64 bits with one ds_read2_b32
// (dword 0 is kept a second time behind dword 15: no wrap) - LDS latency instead of memory latency.  A lane that outruns its ring (a
// burst of > 2 bytes per step) skips steps until the next refill.  Staging whole chunks in LDS costs two thirds of the occupancy.
#define JRING 16
#define JREFILL 8
struct jring {
    const uint8_t *d; uint32_t *rg;                              // rg: this thread's JRING + 1 dwords
    uint32_t bp, fill;                                           // bit position; the ring holds the stream up to this byte offset (16-aligned)
    uint4 in;                                                    // the 16 bytes at `fill`, requested at the last refill
    int k, blk;
    __device__ void put(const uint4 v, uint32_t at)
    {
        // the ring runs BACKWARDS (stream dword i at slot JRING - (i mod JRING), dword 0 mod JRING also at slot 0): the pair (w + 1, w)
        // then comes out of one ds_read2_b32 as the low and the high half of a 64-bit register pair, no swap
        const uint32_t s = JRING - 3 - ((at >> 2) & (JRING - 1));          // slot of dword at + 12
        const uint32_t a = __builtin_bswap32(v.x);
        rg[s] = __builtin_bswap32(v.w); rg[s + 1] = __builtin_bswap32(v.z); rg[s + 2] = __builtin_bswap32(v.y); rg[s + 3] = a;
        if (s == JRING - 3) rg[0] = a;
    }
    __device__ void seek(uint32_t b)
    {
        bp = b;
        const uint32_t f0 = (b >> 7) * 16u;
        const uint4 v0 = *reinterpret_cast<const uint4 *>(d + f0), v1 = *reinterpret_cast<const uint4 *>(d + f0 + 16), v2 = *reinterpret_cast<const uint4 *>(d + f0 + 32);
        put(v0, f0); put(v1, f0 + 16); put(v2, f0 + 32);
        fill = f0 + 48;
        in = *reinterpret_cast<const uint4 *>(d + fill);
    }
    __device__ uint32_t pos() const { return bp; }
    // every JREFILL steps: true = the lane can decode JREFILL symbols (< 4 bytes each) without looking at its ring's fill level
    __device__ bool refill()
    {
        const uint32_t wb = (bp >> 5) * 4u;
        if (fill + 16u - wb <= 4u * JRING) { put(in, fill); fill += 16; }
        in = *reinterpret_cast<const uint4 *>(d + fill);
        return fill - wb >= 8u + 4u * JREFILL;
    }
    __device__ uint32_t top() const
    {
        const uint32_t *p = rg + (~(bp >> 5) & (JRING - 1));                // slots of dwords w + 1, w
        return (uint32_t)((((uint64_t)p[1] << 32 | p[0]) << (bp & 31u)) >> 32);
    }
    __device__ void advance(int tot) { bp += (uint32_t)tot; }
};

__device__ inline void jload_tables(jfast &T, const jfast *t)
{
    const uint4 *src = (const uint4 *)t;
    uint4 *dst = (uint4 *)&T;
    for (int i = threadIdx.x; i < (int)(sizeof(jfast) / 16); i += blockDim.x) dst[i] = src[i];
    __syncthreads();
}

__device__ inline int jextend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

__device__ inline uint64_t jpack(uint32_t bp, int k, int blk) { return ((uint64_t)bp << 32) | ((uint64_t)k << 8) | (uint64_t)blk; }

// A code longer than the look-ahead whose prefix got no second-level table: the first length whose largest code is not below the
// prefix (jdhuff.c's loop), the seven comparisons side by side; canonical tables from global memory (rare: no LDS spent on them).
__device__ inline uint32_t jslow(const jpeg_tab *t, int slot, uint32_t top, bool write)
{
    const int c16 = (int)(top >> 16);
    unsigned below = 0;
#pragma unroll
    for (int j = 0; j < 7; ++j) below |= (unsigned)((c16 >> (6 - j)) > t->maxcode[slot][10 + j]) << j;
    const int l = 10 + __builtin_ctz(~below | 0x80u);
    if (l > 16) return jentry(write, slot & 1, 16, 0);
    return jentry(write, slot & 1, l, t->vals[slot][((c16 >> (16 - l)) + t->valoff[slot][l]) & 255]);
}

// Decodes symbols from the reader's position while they START in front of bit `bend` (and, WRITE, fewer than max_blocks blocks are
// complete).  emit.coef(zigzag index, value) for every non-zero coefficient of the block in progress (DC as its difference),
// emit.block(i) when the i-th block of this run is complete.
// RST (streams with restart intervals): rl = bits up to the next restart boundary.  A symbol that would run across it is the
// interval's padding (1-bits: a proper prefix of every table's longest codes, never a code - T.81 Annex C), so the interval is
// complete (jdhuff.c process_restart): on to the boundary, next MCU.  A decoder in step is at k = 0, blk = 0 there anyway; one that
// is out of step is in step from there on.
template <bool WRITE, bool RST, class Emit>
__device__ inline int jrun(const jfast &T, const jpeg_tab *t, const uint32_t *__restrict__ rst, jring &r, uint32_t bend, const jpeg_geom &g, int max_blocks, Emit &emit)
{
    int done = 0;
    const int ny = g.comp_nb[0], bpm = g.bpm;
    const uint16_t *lut = &T.lut[0][0];
    int sb = (r.blk < ny ? 0 : r.blk - ny + 1) * 2048;          // byte offset of the DC table of the block's component (AC: + 1024)
    const uint32_t bp0 = r.pos();
    int rem = (int)(bend - bp0);                                  // bits up to the end of the chunk
    uint32_t ridx = 0, nrst = 0;
    int rl = 0x7fffffff;
    if (RST) {
        nrst = t->nrst;
        uint32_t lo = 0, hi = nrst;                               // first boundary behind bp0
        while (lo < hi) { const uint32_t m = (lo + hi) >> 1; if (rst[m] * 8u > bp0) hi = m; else lo = m + 1; }
        ridx = lo;
        if (ridx < nrst) rl = (int)(rst[ridx] * 8u - bp0);
    }
    // The loop is the WAVE's - every lane stays until the last one is done (a finished or stalled lane skips the body), so that the
    // refill points are the same for all lanes: JREFILL symbol steps, refill, ...
    for (;;) {
        if (!__any(rem > 0 && (!WRITE || done < max_blocks))) break;
        const bool go = r.refill();
#pragma unroll 1
        for (int u = 0; u < JREFILL; ++u) {
            if (!(go && rem > 0 && (!WRITE || done < max_blocks))) continue;
            const uint32_t top = r.top();
            uint32_t e = *reinterpret_cast<const uint16_t *>(reinterpret_cast<const char *>(lut) + sb + (r.k ? 1024 : 0) + ((top >> 22) & 0x3FEu));
            if (e & 0x8000u) {
                if (e != 0xFFFFu) e = T.sub[e & 0x7FFFu][(top >> 16) & 127u];
                else e = jslow(t, (sb >> 10) + (r.k ? 1 : 0), top, WRITE);
            }
            int tot, dk, len = 0, s = 0;
            if (WRITE) { len = (int)(e >> 11) + 1; s = (int)(e >> 7) & 15; dk = (int)(e & 127u); tot = len + s; }
            else { tot = (int)(e >> 8); dk = (int)(e & 255u); }
            if (RST && tot > rl) {
                const uint32_t lim = rst[ridx] * 8u;
                r.seek(lim);
                r.k = 0; r.blk = 0; sb = 0;
                rem = (int)(bend - lim);
                ++ridx;
                rl = ridx < nrst ? (int)(rst[ridx] * 8u - lim) : 0x7fffffff;
                continue;
            }
            if (WRITE && s) {
                const int kk = r.k + dk - 1;
                if (kk < 64) emit.coef(kk, jextend((int)((top << len) >> (32 - s)), s));
            }
            r.advance(tot); rem -= tot;
            if (RST) rl -= tot;
            r.k += dk;
            if (r.k >= 64) {
                r.k = 0; r.blk = r.blk + 1 == bpm ? 0 : r.blk + 1;
                sb = (r.blk < ny ? 0 : r.blk - ny + 1) * 2048;
                emit.block(done); ++done;
            }
        }
    }
    return done;
}

struct jemit_none { __device__ void coef(int, int) const {} __device__ void block(int) const {} };

#define JSYNC_LDS_PAD 2560
template <bool RST>
__global__ __launch_bounds__(JTPB) void k_jpeg_sync(const jpeg_tab *__restrict__ tabs, const uint8_t *__restrict__ ent, const uint32_t *__restrict__ rst,
                                                    jpeg_geom g, int nch_max, unsigned long long *__restrict__ state, unsigned long long *__restrict__ used,
                                                    int *__restrict__ count, int iter, int *__restrict__ flags)
{
    __shared__ jfast T;
    __shared__ uint32_t ring[JTPB][JRING + 1];
    // (26.9 KB of LDS + JSYNC_LDS_PAD bytes at the launch: five workgroups per CU = five waves per SIMD, the optimum - six take 6 % longer, four 7 %)
    const int b = blockIdx.y;
    const jpeg_tab *t = tabs + b;
    if ((int)(blockIdx.x * JTPB) >= t->nch) return;
    const int i = blockIdx.x * JTPB + threadIdx.x;
    const size_t o = (size_t)b * nch_max + i;
    // who has work?  iteration 0: everybody; iteration 1: the chunks whose predecessor published a state they have not started from
    // (chunk 0 started from the truth and never has).  The later iterations are k_jpeg_sync_tail's.
    unsigned long long e = 0;
    bool work = i < t->nch;
    if (work && iter) {
        work = i > 0;
        if (work) { e = state[o - 1]; work = e != used[o]; }
    }
    if (!__syncthreads_or(work)) return;
    jload_tables(T, &t->S);
    if (!work) return;
    jring r;
    r.d = ent + t->ent_off; r.rg = ring[threadIdx.x];
    const uint32_t cbits = (uint32_t)g.jch * 8u;
    // chunk 0 starts at the true start of the scan; the guess of every other chunk: a block of the MCU's first component starts here
    if (iter == 0) e = jpack((uint32_t)i * cbits, 0, 0);
    r.k = (int)((e >> 8) & 63); r.blk = (int)(e & 15);
    r.seek((uint32_t)(e >> 32));
    used[o] = e;
    jemit_none em;
    const int n = jrun<false, RST>(T, t, rst + t->rst_off, r, (uint32_t)(i + 1) * cbits, g, 0x7fffffff, em);
    count[o] = n;
    const unsigned long long x = jpack(r.pos(), r.k, r.blk);
    if (iter == 0) state[o] = x;
    else if (x != state[o]) { state[o] = x; flags[iter < JMAX_ITERS ? iter : JMAX_ITERS - 1] = 1; }
}

// The passes from the third on: a few per cent of the chunks still have work (14 % of the chunks do not fall into step inside their
// 256 bytes, so 2 %, 0.3 %, ... are still wrong), and the pass costs what ONE chunk's decode takes at whatever occupancy it runs.  Per
// 256 chunks (k_jpeg_sync) that left every workgroup with one wave of a few lanes - 3584 sparse waves per pass of 512 frames.  Here a
// workgroup looks at JTAIL chunks of an image and packs those with work into full waves: 4 to 30 times fewer waves, each pass closer
// to the latency of a single chunk.
#define JTAIL 2048
template <bool RST>
__global__ __launch_bounds__(JTPB) void k_jpeg_sync_tail(const jpeg_tab *__restrict__ tabs, const uint8_t *__restrict__ ent, const uint32_t *__restrict__ rst,
                                                         jpeg_geom g, int nch_max, unsigned long long *__restrict__ state, unsigned long long *__restrict__ used,
                                                         int *__restrict__ count, int iter, int *__restrict__ flags)
{
    __shared__ jfast T;
    __shared__ uint32_t ring[JTPB][JRING + 1];
    __shared__ uint16_t alist[JTAIL];
    __shared__ int acount;
    const int b = blockIdx.y, c0 = blockIdx.x * JTAIL;
    const jpeg_tab *t = tabs + b;
    if (c0 >= t->nch) return;
    if (threadIdx.x == 0) acount = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < JTAIL; j += JTPB) {
        const int i = c0 + j;
        if (i > 0 && i < t->nch && state[(size_t)b * nch_max + i - 1] != used[(size_t)b * nch_max + i]) alist[atomicAdd(&acount, 1)] = (uint16_t)j;
    }
    __syncthreads();
    const int total = acount;
    if (total == 0) return;
    jload_tables(T, &t->S);
    const uint32_t cbits = (uint32_t)g.jch * 8u;
    for (int a0 = 0; a0 < total; a0 += JTPB) {
        if (a0 + (int)threadIdx.x >= total) continue;
        const int i = c0 + alist[a0 + threadIdx.x];
        const size_t o = (size_t)b * nch_max + i;
        const unsigned long long e = state[o - 1];                // (may be newer than what the list was built from: whatever it is, `used` records it)
        jring r;
        r.d = ent + t->ent_off; r.rg = ring[threadIdx.x];
        r.k = (int)((e >> 8) & 63); r.blk = (int)(e & 15);
        r.seek((uint32_t)(e >> 32));
        used[o] = e;
        jemit_none em;
        const int n = jrun<false, RST>(T, t, rst + t->rst_off, r, (uint32_t)(i + 1) * cbits, g, 0x7fffffff, em);
        count[o] = n;
        const unsigned long long x = jpack(r.pos(), r.k, r.blk);
        if (x != state[o]) { state[o] = x; flags[iter < JMAX_ITERS ? iter : JMAX_ITERS - 1] = 1; }
    }
}

// exclusive prefix sum of count[b][0..nch) -> base[b][...]; total in base[b][nch_max]
__global__ __launch_bounds__(1024) void k_jpeg_scan(const jpeg_tab *__restrict__ tabs, int nch_max, const int *__restrict__ count, int *__restrict__ base)
{
    __shared__ int wsum[16];
    __shared__ int carry;
    const int b = blockIdx.x, nch = tabs[b].nch;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int i0 = 0; i0 < nch; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        const int v = i < nch ? count[(size_t)b * nch_max + i] : 0;
        int s = v;
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(s, d, 64); if (lane >= d) s += t; }
        if (lane == 63) wsum[wv] = s;
        __syncthreads();
        int pre = carry;
        for (int k = 0; k < wv; ++k) pre += wsum[k];
        if (i < nch) base[(size_t)b * (nch_max + 1) + i] = pre + s - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = pre + s;
        __syncthreads();
    }
    if (threadIdx.x == 0) base[(size_t)b * (nch_max + 1) + nch_max] = carry;
}

// Coefficients of the block in progress are collected in the thread's LDS row and leave as one wide store when the block is
// complete (2-byte stores scattered over HBM - partial-line writes - made this kernel 3x slower than the counting pass).  The row
// holds the first JROW_K zigzag positions only (20 KB per 256 threads beside the 17 KB of the readers' rings and 9 KB of tables:
// three waves per SIMD; all 64 positions would leave two).  The coefficients behind position JROW_K (the high-frequency half: a few
// per cent at camera qualities) go straight to the block's upper half - scattered 2-byte stores, which is why shorter rows lose:
// 1.82 ms per 512 frames with 32 positions, 2.38 with 16, 3.59 with 8 (profiles/r04_exp_jpeg_rows.txt).  A block that straddles a
// chunk boundary is shared with the neighbouring thread: its parts are scattered element-wise onto a zeroed block instead.
#define JROW_K 32
#define JBLK_PITCH (JROW_K + 8)                                   // int16 per LDS row: 16-byte aligned (the row moves as b128 reads / writes), 4 dwords of padding spread the lanes over the banks
struct jemit_store {
    int16_t *row;                                                 // this thread's LDS row (zigzag order, zero between blocks)
    int16_t *out; int n0, nblk, nfin; bool head_partial;          // nfin: blocks this chunk completes (the one after them straddles)
    int16_t *dc; int cur;                                         // DC differences go to a dense array of their own (k_jpeg_dc scans it)
    int curdc; bool up;                                           // of the block in progress: DC difference; it has coefficients behind the row
    // DC entry: bits 0..14 the difference (12 significant bits), bit 15 = the block's upper half holds data (else it is never read)
    __device__ static int16_t dc_entry(int diff, bool upper) { return (int16_t)((diff & 0x7FFF) | (upper ? 0x8000 : 0)); }
    __device__ void coef(int k, int v)
    {
        if (k < JROW_K) { row[k] = (int16_t)v; if (k == 0) curdc = v; }
        else if (n0 + cur < nblk) {
            int16_t *blk = out + (size_t)(n0 + cur) * 64;
            // the first coefficient behind the row: the block's upper half becomes data, on a zero background the thread lays itself -
            // unless the block straddles a chunk boundary (two threads write it: k_jpeg_zero_heads has zeroed it)
            if (!up && !((cur == 0 && head_partial) || cur >= nfin)) {
#pragma unroll
                for (int q = JROW_K / 8; q < 8; ++q) reinterpret_cast<uint4 *>(blk)[q] = make_uint4(0, 0, 0, 0);
            }
            up = true;
            blk[k] = (int16_t)v;
        }
    }
    __device__ void scatter(int n)
    {
        int16_t *dst = out + (size_t)n * 64;
        for (int k = 0; k < JROW_K; ++k) { const int16_t v = row[k]; if (v) { dst[k] = v; row[k] = 0; } }
    }
    // A complete block is read out of the row at once (the row is needed for the next block) but stored when the thread's NEXT
    // block is complete: the store then finds its data long arrived, where waiting for the LDS reads in place put their latency
    // into the dependency chain of almost every step (some lane of a wave completes a block in almost every step): 1.81 -> x ms
    uint32_t pend[JROW_K / 2]; uint4 *pend_dst;                   // (scalars: a uint4 member sends the whole struct to the stack)
    __device__ void drain()
    {
        if (!pend_dst) return;
#pragma unroll
        for (int q = 0; q < JROW_K / 8; ++q) pend_dst[q] = make_uint4(pend[4 * q], pend[4 * q + 1], pend[4 * q + 2], pend[4 * q + 3]);
        pend_dst = nullptr;
    }
    __device__ void block(int done)
    {
        const int n = n0 + done;
        const int diff = curdc; const bool upper = up;
        cur = done + 1; curdc = 0; up = false;
        if (n >= nblk) return;
        if (done == 0 && head_partial) { scatter(n); return; }       // (its DC entry is the starting thread's)
        dc[n] = dc_entry(diff, upper);
        drain();
        uint4 *src = reinterpret_cast<uint4 *>(row);
        pend_dst = (uint4 *)(out + (size_t)n * 64);
#pragma unroll
        for (int q = 0; q < JROW_K / 8; ++q) {
            const uint4 v = src[q];
            pend[4 * q] = v.x; pend[4 * q + 1] = v.y; pend[4 * q + 2] = v.z; pend[4 * q + 3] = v.w;
            src[q] = make_uint4(0, 0, 0, 0);
        }
    }
};

template <bool RST>
__global__ __launch_bounds__(JTPW) void k_jpeg_write(const jpeg_tab *__restrict__ tabs, const uint8_t *__restrict__ ent, const uint32_t *__restrict__ rst,
                                                     jpeg_geom g, int nch_max, const unsigned long long *__restrict__ state, const int *__restrict__ base,
                                                     int16_t *__restrict__ coef, int16_t *__restrict__ dcarr, int *__restrict__ endinfo)
{
    __shared__ jfast T;
    __shared__ __attribute__((aligned(16))) int16_t rows[JTPW][JBLK_PITCH];
    __shared__ uint32_t wring[JTPW][JRING + 1];
    const int b = blockIdx.y;
    const jpeg_tab *t = tabs + b;
    if ((int)(blockIdx.x * JTPW) >= t->nch) return;
    jload_tables(T, &t->W);
    const int i = blockIdx.x * JTPW + threadIdx.x;
    if (i >= t->nch) return;
    const int n0 = base[(size_t)b * (nch_max + 1) + i];
    if (n0 >= g.nblk) return;
    jring r;                                                      // the ring's 17 KB leave three waves per SIMD where the register window had five: still 15 % faster
    r.d = ent + t->ent_off; r.rg = wring[threadIdx.x];
    const unsigned long long e = i ? state[(size_t)b * nch_max + i - 1] : 0ull;
    r.k = (int)((e >> 8) & 63); r.blk = (int)(e & 15);
    r.seek((uint32_t)(e >> 32));
    for (int q = 0; q < JBLK_PITCH / 2; ++q) ((uint32_t *)rows[threadIdx.x])[q] = 0;
    jemit_store em;
    em.row = rows[threadIdx.x]; em.out = coef + (size_t)b * g.nblk * 64; em.n0 = n0; em.nblk = g.nblk; em.head_partial = r.k != 0;
    em.dc = dcarr + (size_t)b * g.nblk; em.cur = 0; em.pend_dst = nullptr; em.curdc = 0; em.up = false;
    em.nfin = base[(size_t)b * (nch_max + 1) + (i + 1 < t->nch ? i + 1 : nch_max)] - n0;
    const int n = jrun<true, RST>(T, t, rst + t->rst_off, r, (uint32_t)(i + 1) * (uint32_t)g.jch * 8u, g, g.nblk - n0, em);
    em.drain();
    if (r.k != 0 && n0 + n < g.nblk) {                            // the block still in progress continues in the next chunk
        em.scatter(n0 + n);
        if (!(n == 0 && em.head_partial)) em.dc[n0 + n] = jemit_store::dc_entry(em.curdc, true);   // it started here: its DC entry, upper half "data" (zeroed)
    }
    if (n0 + n == g.nblk && n > 0) {                              // this thread finished the last block: where the scan ended
        endinfo[b * 2] = (int)((r.pos() + 7u) >> 3);
        endinfo[b * 2 + 1] = 1;
    }
}

// The write pass stores the first JROW_K zigzag positions of every block it completes as one 64-byte row; the upper half of a block is
// written - zeros first - only when the block has a coefficient there, and the block's DC entry says so (the IDCT does not read the
// others).  Blocks that straddle a chunk boundary are scattered element-wise by two or more threads: k_jpeg_zero_heads zeroes those (a
// few thousand per image, between the scan and the write pass) and they count as "upper half holds data".  No fill of the 3.3 GB
// coefficient buffer: beside the first synchronisation pass and the previous batch's pipeline run even half of it (the upper halves)
// cost the double-buffered loop 7 % of its period.
__global__ __launch_bounds__(256) void k_jpeg_zero_heads(const jpeg_tab *__restrict__ tabs, jpeg_geom g, int nch_max, const unsigned long long *__restrict__ state,
                                                         const int *__restrict__ base, int16_t *__restrict__ coef)
{
    const int b = blockIdx.y, i = blockIdx.x * 32 + (threadIdx.x >> 3);
    if (i < 1 || i >= tabs[b].nch) return;
    if (((state[(size_t)b * nch_max + i - 1] >> 8) & 63) == 0) return;        // a block starts with the chunk
    const int n = base[(size_t)b * (nch_max + 1) + i];
    if (n >= g.nblk) return;
    reinterpret_cast<uint4 *>(coef + ((size_t)b * g.nblk + n) * 64)[threadIdx.x & 7] = make_uint4(0, 0, 0, 0);
}

// DC prediction: inclusive prefix sum of the DC differences of one component in decode order (one workgroup per image x component),
// restarted at every restart interval (segmented scan: flag = first block of an interval)
__global__ __launch_bounds__(1024) void k_jpeg_dc(const jpeg_tab *__restrict__ tabs, jpeg_geom g, int16_t *__restrict__ dcarr)
{
    __shared__ int wsum[16], wflag[16];
    __shared__ int carry;
    const int b = blockIdx.x, c = blockIdx.y;
    const int nbc = g.comp_nb[c], off = g.comp_off[c];
    const int total = g.mcux * g.mcuy * nbc;
    const int seg = tabs[b].ri * nbc;                           // blocks of this component per restart interval (0: one interval)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int16_t *dc = dcarr + (size_t)b * g.nblk;                   // dense: the strided DC slots of the coefficient blocks cost 0.29 ms, this 0.0x
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int j0 = 0; j0 < total; j0 += 4096) {                  // four consecutive blocks per thread: a quarter of the barriers
        const int jb = j0 + 4 * threadIdx.x;
        size_t n[4];
        int v[4], f[4], up[4] = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = jb + i;
            n[i] = 0; v[i] = 0;
            f[i] = seg > 0 && j % seg == 0;
            if (j < total) { n[i] = (size_t)(j / nbc) * g.bpm + off + j % nbc; const int x = (uint16_t)dc[n[i]]; up[i] = x & 0x8000; v[i] = (int)((unsigned)x << 17) >> 17; }   // bit 15: the upper-half flag, carried through
        }
        int sl[4], gl[4];                                       // sums since the last interval start inside the thread / "one was seen"
        sl[0] = v[0]; gl[0] = f[0];
#pragma unroll
        for (int i = 1; i < 4; ++i) { sl[i] = f[i] ? v[i] : sl[i - 1] + v[i]; gl[i] = gl[i - 1] | f[i]; }
        int s = sl[3], fl = gl[3];
        for (int d = 1; d < 64; d <<= 1) {
            const int ts = __shfl_up(s, d, 64), tf = __shfl_up(fl, d, 64);
            if (lane >= d) { if (!fl) s += ts; fl |= tf; }
        }
        if (lane == 63) { wsum[wv] = s; wflag[wv] = fl; }
        int ex = __shfl_up(s, 1, 64), exf = __shfl_up(fl, 1, 64);   // the same, up to the thread in front of this one
        if (lane == 0) { ex = 0; exf = 0; }
        __syncthreads();
        int pre = carry;                                        // sum since the last interval start in front of this wave
        for (int k = 0; k < wv; ++k) pre = wflag[k] ? wsum[k] : pre + wsum[k];
        const int pt = exf ? ex : pre + ex;                     // ... in front of this thread
        int r = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            r = gl[i] ? sl[i] : sl[i] + pt;
            if (jb + i < total) dc[n[i]] = (int16_t)((r & 0x7FFF) | up[i]);
        }
        __syncthreads();
        if (threadIdx.x == 1023) carry = r;
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ device: IDCT, upsampling, colour
#define JFIX_0_298631336 2446
#define JFIX_0_390180644 3196
#define JFIX_0_541196100 4433
#define JFIX_0_765366865 6270
#define JFIX_0_899976223 7373
#define JFIX_1_175875602 9633
#define JFIX_1_501321110 12299
#define JFIX_1_847759065 15137
#define JFIX_1_961570560 16069
#define JFIX_2_053119869 16819
#define JFIX_2_562915447 20995
#define JFIX_3_072711026 25172

// one 8-point pass of libjpeg's jidctint.c (before the descale)
__device__ inline void jidct8(const int *in, int *o)
{
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * JFIX_0_541196100;
    int tmp2 = z1 + z3 * (-JFIX_1_847759065);
    int tmp3 = z1 + z2 * JFIX_0_765366865;
    z2 = in[0]; z3 = in[4];
    int tmp0 = (int)((unsigned)(z2 + z3) << 13);
    int tmp1 = (int)((unsigned)(z2 - z3) << 13);
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * JFIX_1_175875602;
    tmp0 *= JFIX_0_298631336; tmp1 *= JFIX_2_053119869; tmp2 *= JFIX_3_072711026; tmp3 *= JFIX_1_501321110;
    z1 *= -JFIX_0_899976223; z2 *= -JFIX_2_562915447; z3 *= -JFIX_1_961570560; z4 *= -JFIX_0_390180644;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    o[0] = tmp10 + tmp3; o[7] = tmp10 - tmp3;
    o[1] = tmp11 + tmp2; o[6] = tmp11 - tmp2;
    o[2] = tmp12 + tmp1; o[5] = tmp12 - tmp1;
    o[3] = tmp13 + tmp0; o[4] = tmp13 - tmp0;
}

__device__ inline uint32_t jrange_limit(int x)
{   // jdmaster.c prepare_range_limit_table, post-IDCT half, index i = x & 1023: i < 128 -> i + 128, < 512 -> 255, < 896 -> 0, else i - 896.
    // With t = the index read as a signed 10-bit number that is clamp(t + 128, 0, 255) in every one of the four ranges: one
    // sign-extending field extract, one add, one v_med3_i32 instead of three compares and three selects (eight times per lane).
    const int v = ((x << 22) >> 22) + 128;
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(v), "v"(255));
    return (uint32_t)r;
}

__device__ inline int jclamp8(int x) { return x < 0 ? 0 : x > 255 ? 255 : x; }

// Chroma of the 8 output pixels x0 .. x0+7 (x0 a multiple of 8) of row y with libjpeg's "fancy" upsampling
// (jdsample.c h2v1_fancy_upsample / h2v2_fancy_upsample; context rows replicated at the top and bottom edge as jdmainct.c does).
// The four chroma columns under the pixels come in as one dword per row, their two neighbours as bytes.
// libjpeg's edge rules - the first output of column 0 and the second of column cw - 1 are the column itself, (4 v + r) >> s - are the
// general formula (3 v + neighbour + r) >> s with the neighbour replaced by the column: the left neighbour of column 0 and the right
// one of the group's last column come in replicated by the clamped loads (cl, cr); only where the image's last column sits INSIDE
// a group of four (cw not a multiple of four: one thread per row) the columns behind it are patched.  No select per output.
#define JCHROMA_EDGE() do { if (cx0 + 3 > cw - 1) { for (int k = 1; k < 4; ++k) if (cx0 + k > cw - 1) t[k + 1] = t[k]; } } while (0)
__device__ inline void jchroma8(const uint8_t *__restrict__ pl, int pitch, int cw, int ch, int hmax, int vmax, int x0, int y, int *out)
{   // (offsets inside a plane are 32-bit: the plane base is uniform, so the loads take it from scalar registers)
    if (hmax == 1) {
        const uint2 v = *(const uint2 *)(pl + (unsigned)(y * pitch + x0));
#pragma unroll
        for (int k = 0; k < 4; ++k) { out[k] = (v.x >> (8 * k)) & 255; out[k + 4] = (v.y >> (8 * k)) & 255; }
        return;
    }
    const int cx0 = x0 >> 1;
    const int cl = cx0 > 0 ? cx0 - 1 : 0, cr = cx0 + 4 < cw ? cx0 + 4 : cw - 1;
    int t[6];
    if (vmax == 1) {
        const unsigned ro = (unsigned)(y * pitch);
        const uint32_t m = *(const uint32_t *)(pl + (ro + cx0));
        t[0] = pl[ro + cl]; t[5] = pl[ro + cr];
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k + 1] = (m >> (8 * k)) & 255;
        JCHROMA_EDGE();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int v = t[k + 1];
            out[2 * k] = (v * 3 + t[k] + 1) >> 2;
            out[2 * k + 1] = (v * 3 + t[k + 2] + 2) >> 2;
        }
        return;
    }
    const int r0 = y >> 1;
    int r1 = (y & 1) ? r0 + 1 : r0 - 1;
    r1 = r1 < 0 ? 0 : r1 > ch - 1 ? ch - 1 : r1;
    const unsigned o0 = (unsigned)(r0 * pitch), o1 = (unsigned)(r1 * pitch);
    const uint32_t m0 = *(const uint32_t *)(pl + (o0 + cx0)), m1 = *(const uint32_t *)(pl + (o1 + cx0));
    t[0] = pl[o0 + cl] * 3 + pl[o1 + cl]; t[5] = pl[o0 + cr] * 3 + pl[o1 + cr];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k + 1] = (int)((m0 >> (8 * k)) & 255) * 3 + (int)((m1 >> (8 * k)) & 255);
    JCHROMA_EDGE();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int v = t[k + 1];
        out[2 * k] = (v * 3 + t[k] + 8) >> 4;
        out[2 * k + 1] = (v * 3 + t[k + 2] + 7) >> 4;
    }
}

// Eight pixels of row y from x0 on: luma y8, chroma from the planes with fancy upsampling, fixed-point YCbCr -> BGR (jdcolor.c), 24
// bytes of BGR out as dwords (bytes when the row pitch is not a dword multiple).
// as_gray: bgr / bgr2 are gray planes (rows of g.w bytes) and receive what k_gray_bgr8 (k_image.hip: cv2.cvtColor BGR2GRAY) makes of the
// pixel - the pipeline's first stage fused into the decoder's last, the BGR frame (three times the bytes, written here and read there)
// never exists.
__device__ inline void jcolor8(const jpeg_geom &g, const uint8_t *__restrict__ pl, uint2 y8, int b, int x0, int y, uint8_t *__restrict__ bgr,
                               uint8_t *__restrict__ bgr2, int split, size_t bgr_stride, int as_gray)
{
    uint8_t *o = (b < split ? bgr + (size_t)b * bgr_stride : bgr2 + (size_t)(b - split) * bgr_stride) + ((size_t)y * g.w + x0) * 3;   // pairs: previous frames | next frames
    int pb[8], pg[8], pr[8];                                      // clamped to 0..255
    if (g.ncomp == 1) {
#pragma unroll
        for (int k = 0; k < 8; ++k) pb[k] = pg[k] = pr[k] = (int)(((k < 4 ? y8.x : y8.y) >> (8 * (k & 3))) & 255);
    } else {
        const int cw = (g.w + g.hmax - 1) >> (g.hmax - 1), ch = (g.h + g.vmax - 1) >> (g.vmax - 1);    // sampling factors are 1 or 2
        int cb[8], cr[8];
        jchroma8(pl + g.plane_off[1], g.pw[1], cw, ch, g.hmax, g.vmax, x0, y, cb);
        jchroma8(pl + g.plane_off[2], g.pw[2], cw, ch, g.hmax, g.vmax, x0, y, cr);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int Y = (int)(((k < 4 ? y8.x : y8.y) >> (8 * (k & 3))) & 255);
            // jdcolor.c build_ycc_rgb_table: SCALEBITS 16, FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554,
            // applied to cb - 128, cr - 128 and rounded with ONE_HALF: the - 128 and the rounding are folded into the constants (17-bit
            // constants x 8-bit chroma: exact in the 24-bit multiplier, half the issue cost of v_mul_lo_u32)
            pb[k] = jclamp8(Y + ((__mul24(116130, cb[k]) + (32768 - 128 * 116130)) >> 16));
            pg[k] = jclamp8(Y + ((__mul24(-22554, cb[k]) + (32768 + 128 * 22554 + 128 * 46802) + __mul24(-46802, cr[k])) >> 16));
            pr[k] = jclamp8(Y + ((__mul24(91881, cr[k]) + (32768 - 128 * 91881)) >> 16));
        }
    }
    if (as_gray) {
        uint8_t *og = (b < split ? bgr + (size_t)b * bgr_stride : bgr2 + (size_t)(b - split) * bgr_stride) + (size_t)y * g.w + x0;
        uint32_t lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {                               // k_image.hip gray1(): (3735 b + 19235 g + 9798 r + 16384) >> 15
            lo |= (((uint32_t)pb[k] * 3735u + (uint32_t)pg[k] * 19235u + (uint32_t)pr[k] * 9798u + 16384u) >> 15) << (8 * k);
            hi |= (((uint32_t)pb[k + 4] * 3735u + (uint32_t)pg[k + 4] * 19235u + (uint32_t)pr[k + 4] * 9798u + 16384u) >> 15) << (8 * k);
        }
        if ((g.w & 7) == 0 && (bgr_stride & 7) == 0) *(uint2 *)og = make_uint2(lo, hi);
        else {
            const int n = g.w - x0 < 8 ? g.w - x0 : 8;
            for (int k = 0; k < n; ++k) og[k] = (uint8_t)((k < 4 ? lo : hi) >> (8 * (k & 3)));
        }
        return;
    }
    uint32_t px[24];
#pragma unroll
    for (int k = 0; k < 8; ++k) { px[3 * k] = (uint32_t)pb[k]; px[3 * k + 1] = (uint32_t)pg[k]; px[3 * k + 2] = (uint32_t)pr[k]; }
    if ((g.w & 7) == 0 && (bgr_stride & 3) == 0) {
        uint32_t *o4 = (uint32_t *)o;
#pragma unroll
        for (int k = 0; k < 6; ++k) o4[k] = px[4 * k] | (px[4 * k + 1] << 8) | (px[4 * k + 2] << 16) | (px[4 * k + 3] << 24);
    } else {
        const int n = (g.w - x0 < 8 ? g.w - x0 : 8) * 3;
        for (int k = 0; k < n; ++k) o[k] = (uint8_t)px[k];
    }
}

// IDCT: 8 lanes per block, 32 blocks per step; a workgroup walks along ONE MCU row of one image (block positions from shifts - the
// blocks of a kind per MCU are 1, 2 or 4 -, quantiser tables loaded once per row, the next step's coefficients in flight during both
// passes).  Two launches per decode:
//   LUMA = false: the chroma blocks -> the Cb / Cr planes (fancy upsampling needs the neighbours of a sample, so chroma goes through HBM);
//   LUMA = true:  the luma blocks of 32 / (hmax vmax) MCUs -> a tile of 8 vmax rows x 256 / vmax pixels in LDS, then the same 256 threads
//                 take eight pixels each: chroma from the planes, colour conversion, BGR or gray out (jcolor8).  The Y plane never
//                 exists: its 8-byte row stores from lanes that sit in different image rows were a quarter of the IDCT's time, and the
//                 colour pass read it back (k_jpeg_idct + k_jpeg_color 2.65 -> 0.39 + 1.71 ms per 512 frames).
template <bool LUMA>
__global__ __launch_bounds__(256) void k_jpeg_idct(const jpeg_tab *__restrict__ tabs, jpeg_geom g, const int16_t *__restrict__ coef,
                                                   const int16_t *__restrict__ dcarr, uint8_t *__restrict__ planes, uint8_t *__restrict__ bgr,
                                                   uint8_t *__restrict__ bgr2, int split, size_t bgr_stride, int as_gray)
{
    __shared__ int ws[32][72];
    __shared__ int16_t cz[32][72];                                // the blocks as stored (zigzag order), one 16-byte load per lane
    __shared__ uint16_t qs[3][64];
    __shared__ uint2 ytile[LUMA ? 16 : 1][LUMA ? 33 : 1];         // [row][8-pixel group] (+ 1: rows on different banks)
    const int b = blockIdx.y, mrow = blockIdx.x;
    const int lb = threadIdx.x >> 3, c = threadIdx.x & 7;
    if (threadIdx.x < 192) qs[threadIdx.x >> 6][threadIdx.x & 63] = tabs[b].q[threadIdx.x >> 6][threadIdx.x & 63];
    unsigned zlo = 0, zhi = 0;                                    // zigzag positions of this lane's column, rows 0..3 / 4..7
#pragma unroll
    for (int r = 0; r < 4; ++r) { zlo |= (unsigned)c_izz[r * 8 + c] << (8 * r); zhi |= (unsigned)c_izz[(r + 4) * 8 + c] << (8 * r); }
    const int ny = g.comp_nb[0];
    const int per = LUMA ? ny : g.bpm - ny, sh = per == 4 ? 2 : per == 2 ? 1 : 0, j0 = LUMA ? 0 : ny;   // blocks of this kind per MCU
    const int nsel = g.mcux * per;                                // ... in the MCU row
    const size_t nbase = (size_t)b * g.nblk + (size_t)mrow * g.mcux * g.bpm;
    auto block_of = [&](int s) { return nbase + (size_t)((s >> sh) * g.bpm + j0 + (s & (per - 1))); };
    const int hs = g.hmax, vsl = g.vmax == 2 ? 1 : 0;              // tile: 8 << vsl rows of 32 >> vsl groups
    const uint8_t *pl = planes + (size_t)b * g.plane_stride;
    // DC entries (predicted DC in bits 0..14, bit 15: the block's upper half holds data) are read two steps ahead, the coefficients
    // one step ahead - the lanes of the upper half (c >= 4) only where the entry says so
    uint4 nv = make_uint4(0, 0, 0, 0);
    int d0 = lb < nsel ? (int)(uint16_t)dcarr[block_of(lb)] : 0, d1 = lb + 32 < nsel ? (int)(uint16_t)dcarr[block_of(lb + 32)] : 0;
    if (lb < nsel && (c < 4 || (d0 & 0x8000))) nv = reinterpret_cast<const uint4 *>(coef + block_of(lb) * 64)[c];
    for (int l0 = 0; l0 < nsel; l0 += 32) {
        const int local = l0 + lb;
        const bool live = local < nsel;
        *reinterpret_cast<uint4 *>(&cz[lb][8 * c]) = nv;
        if (c == 0) cz[lb][0] = (int16_t)((int)((unsigned)d0 << 17) >> 17);      // the predicted DC (the block holds the difference)
        __syncthreads();
        nv = make_uint4(0, 0, 0, 0);
        if (local + 32 < nsel && (c < 4 || (d1 & 0x8000))) nv = reinterpret_cast<const uint4 *>(coef + block_of(local + 32) * 64)[c];
        d0 = d1;
        d1 = local + 64 < nsel ? (int)(uint16_t)dcarr[block_of(local + 64)] : 0;
        const int mx = local >> sh, jj = local & (per - 1);
        const int comp = LUMA ? 0 : 1 + jj;
        if (live) {
            int in[8], o[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int zp = (int)(((r < 4 ? zlo : zhi) >> (8 * (r & 3))) & 255u);
                in[r] = __mul24((int)cz[lb][zp], (int)qs[comp][r * 8 + c]);   // 16 x 16 bits: exact in the 24-bit multiplier
            }
            jidct8(in, o);
#pragma unroll
            for (int r = 0; r < 8; ++r) ws[lb][r * 9 + c] = (o[r] + (1 << 10)) >> 11;
        }
        __syncthreads();                                           // (a wave-level fence would do - the eight lanes of a block are lanes of one wave - and measured 6 % slower)
        if (live) {
            int in[8], o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) in[k] = ws[lb][c * 9 + k];                // this lane's row = c
            jidct8(in, o);
            uint32_t lo = 0, hi = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lo |= jrange_limit((o[k] + (1 << 17)) >> 18) << (8 * k);
                hi |= jrange_limit((o[k + 4] + (1 << 17)) >> 18) << (8 * k);
            }
            if (LUMA) ytile[((jj >> (hs - 1)) << 3) + c][((mx - (l0 >> sh)) << (hs - 1)) + (jj & (hs - 1))] = make_uint2(lo, hi);
            else *(uint2 *)(planes + (size_t)b * g.plane_stride + g.plane_off[comp] + (size_t)(mrow * 8 + c) * g.pw[comp] + mx * 8) = make_uint2(lo, hi);
        }
        if (LUMA) {
            __syncthreads();
            const int r = threadIdx.x >> (5 - vsl), xg = threadIdx.x & ((32 >> vsl) - 1);
            const int y = (mrow << (3 + vsl)) + r, x0 = ((l0 >> sh) * hs + xg) * 8;
            if (x0 < g.w && y < g.h) jcolor8(g, pl, ytile[r][xg], b, x0, y, bgr, bgr2, split, bgr_stride, as_gray);
        }
    }
}

// ------------------------------------------------------------------------------------------------ host: marker segments
struct jhost {
    int w, h, ncomp, hs[3], vs[3], tq[3], td[3], ta[3], ri;
    uint16_t q[4][64]; int qok[4];
    uint8_t bits[2][4][17], vals[2][4][256]; int hok[2][4];
    const uint8_t *ent; size_t ent_len;
};

static const char *jparse(const uint8_t *d, size_t n, jhost *j)
{
    memset(j, 0, sizeof *j);
    if (!d || n < 4 || d[0] != 0xFF || d[1] != 0xD8) return "not a JPEG stream (no SOI)";
    size_t i = 2;
    bool sof = false;
    while (i + 4 <= n) {
        if (d[i] != 0xFF) return "marker expected";
        while (i < n && d[i] == 0xFF) ++i;
        if (i >= n) break;
        const int m = d[i++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return "EOI before the scan";
        if (i + 2 > n) break;
        const size_t L = ((size_t)d[i] << 8) | d[i + 1];
        if (L < 2 || i + L > n) return "truncated marker segment";
        const uint8_t *s = d + i + 2;
        const size_t sl = L - 2;
        if (m == 0xDB) {
            size_t k = 0;
            while (k < sl) {
                const int pq = s[k] >> 4, tq = s[k] & 15;
                ++k;
                if (tq > 3 || pq > 1 || k + (pq ? 128 : 64) > sl) return "bad DQT";
                for (int z = 0; z < 64; ++z) { j->q[tq][h_zz[z]] = pq ? (uint16_t)((s[k] << 8) | s[k + 1]) : s[k]; k += pq ? 2 : 1; }
                j->qok[tq] = 1;
            }
        } else if (m == 0xC4) {
            size_t k = 0;
            while (k + 17 <= sl) {
                const int tc = s[k] >> 4, th = s[k] & 15;
                ++k;
                if (tc > 1 || th > 3) return "bad DHT";
                int cnt = 0;
                j->bits[tc][th][0] = 0;
                for (int l = 1; l <= 16; ++l) { j->bits[tc][th][l] = s[k++]; cnt += j->bits[tc][th][l]; }
                if (cnt > 256 || k + cnt > sl) return "bad DHT";
                memcpy(j->vals[tc][th], s + k, cnt);
                k += cnt;
                j->hok[tc][th] = 1;
            }
        } else if (m == 0xC0 || m == 0xC1) {
            if (sl < 6 || s[0] != 8) return "only 8-bit samples are supported";
            j->h = (s[1] << 8) | s[2]; j->w = (s[3] << 8) | s[4]; j->ncomp = s[5];
            if ((j->ncomp != 1 && j->ncomp != 3) || sl < 6 + 3 * (size_t)j->ncomp || j->h < 1 || j->w < 1) return "unsupported frame header";
            for (int c = 0; c < j->ncomp; ++c) {
                j->hs[c] = s[7 + 3 * c] >> 4; j->vs[c] = s[7 + 3 * c] & 15; j->tq[c] = s[8 + 3 * c];
                if (j->tq[c] > 3) return "bad quantisation table index";
            }
            sof = true;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC8 && m != 0xCC) {
            return "only baseline / extended-sequential Huffman JPEG is supported (progressive, lossless and arithmetic coding are not)";
        } else if (m == 0xDD) {
            if (sl < 2) return "bad DRI";
            j->ri = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {
            if (!sof || sl < 1 || s[0] != j->ncomp || sl < 4 + 2 * (size_t)j->ncomp) return "unsupported scan header (one interleaved scan expected)";
            for (int c = 0; c < j->ncomp; ++c) {
                j->td[c] = s[2 + 2 * c] >> 4; j->ta[c] = s[2 + 2 * c] & 15;
                if (j->td[c] > 3 || j->ta[c] > 3) return "bad Huffman table index";
            }
            const size_t e0 = i + L;
            size_t e = e0;
            if (n >= e0 + 2 && d[n - 2] == 0xFF && d[n - 1] == 0xD9) e = n - 2;   // ends in EOI (every encoder's output): no need to read it all
            else while (e < n) {                                       // up to the next marker (FF followed by anything but a stuffed zero)
                const uint8_t *f = (const uint8_t *)memchr(d + e, 0xFF, n - e);
                if (!f) { e = n; break; }
                e = (size_t)(f - d);
                if (e + 1 >= n) { e = n; break; }
                if (d[e + 1] != 0x00 && !(j->ri && (d[e + 1] & 0xF8) == 0xD0)) break;    // RSTn markers belong to the scan
                e += 2;
            }
            j->ent = d + e0; j->ent_len = e - e0;
            break;
        }
        i += L;
    }
    if (!j->ent) return "no scan found";
    if (j->ncomp == 1) j->hs[0] = j->vs[0] = 1;
    else {
        if (j->hs[1] != 1 || j->vs[1] != 1 || j->hs[2] != 1 || j->vs[2] != 1) return "unsupported chroma sampling";
        if (!((j->hs[0] == 1 && j->vs[0] == 1) || (j->hs[0] == 2 && j->vs[0] == 1) || (j->hs[0] == 2 && j->vs[0] == 2)))
            return "unsupported luma sampling (4:4:4, 4:2:2 and 4:2:0 are supported)";
    }
    for (int c = 0; c < j->ncomp; ++c)
        if (!j->qok[j->tq[c]] || !j->hok[0][j->td[c]] || !j->hok[1][j->ta[c]]) return "a table the scan refers to is missing";
    if (j->ent_len >= (1ull << 28)) return "entropy segment too long";      // bit positions are 32-bit
    return nullptr;
}

// T.81 Annex C code assignment into the look-up tables of one slot (jfast: both encodings) and the canonical tables of the slow path
static void jbuild_slot(jpeg_tab *t, int slot, const uint8_t *bits, const uint8_t *vals, int *nsub)
{
    const bool ac = slot & 1;
    const uint16_t invS = (uint16_t)jentry(false, ac, 16, 0), invW = (uint16_t)jentry(true, ac, 16, 0);
    for (int i = 0; i < 512; ++i) { t->S.lut[slot][i] = invS; t->W.lut[slot][i] = invW; }
    memcpy(t->vals[slot], vals, 256);
    int code = 0, p = 0;
    for (int l = 1; l <= 16; ++l) {
        t->valoff[slot][l] = p - code;
        for (int k = 0; k < bits[l]; ++k, ++p, ++code) {
            const int sym = vals[p & 255];
            const uint16_t eS = (uint16_t)jentry(false, ac, l, sym), eW = (uint16_t)jentry(true, ac, l, sym);
            if (l <= 9) {
                const int lo = code << (9 - l);
                for (int f = 0; f < (1 << (9 - l)) && lo + f < 512; ++f) { t->S.lut[slot][lo + f] = eS; t->W.lut[slot][lo + f] = eW; }
                continue;
            }
            const int pre = code >> (l - 9);
            if (pre >= 512) continue;                            // an over-subscribed (invalid) table: no bit pattern reaches this code
            uint16_t head = t->S.lut[slot][pre];
            if (!(head & 0x8000u)) {                             // the first long code under this prefix: a second-level table, if one is left
                head = 0xFFFFu;
                if (*nsub < (g_ofk_tuning.jpeg_sub ? g_ofk_tuning.jpeg_sub - 1 : JNSUB)) {     // ofk_set_tuning("jpeg_sub")
                    head = (uint16_t)(0x8000u | (unsigned)*nsub);
                    for (int f = 0; f < 128; ++f) { t->S.sub[*nsub][f] = invS; t->W.sub[*nsub][f] = invW; }
                    ++*nsub;
                }
                t->S.lut[slot][pre] = t->W.lut[slot][pre] = head;
            }
            if (head != 0xFFFFu) {
                const int n = head & 0x7FFF, lo = (code & ((1 << (l - 9)) - 1)) << (16 - l);
                for (int f = 0; f < (1 << (16 - l)); ++f) { t->S.sub[n][lo + f] = eS; t->W.sub[n][lo + f] = eW; }
            }
        }
        t->maxcode[slot][l] = bits[l] ? code - 1 : -1;
        code <<= 1;
    }
    t->maxcode[slot][0] = -1; t->maxcode[slot][17] = 0x7fffffff;
    t->valoff[slot][0] = t->valoff[slot][17] = 0;
}

// all tables of one image; components that name the same Huffman table share its second-level tables
static void jbuild_tables(jpeg_tab *t, const jhost &j)
{
    int nsub = 0;
    for (int f = 0; f < JNSUB; ++f) { memset(t->S.sub[f], 0, sizeof t->S.sub[f]); memset(t->W.sub[f], 0, sizeof t->W.sub[f]); }
    for (int cc = 0; cc < 3; ++cc) {
        const int cs = cc < j.ncomp ? cc : 0;
        for (int ac = 0; ac < 2; ++ac) {
            const int slot = 2 * cc + ac, th = ac ? j.ta[cs] : j.td[cs];
            int same = -1;
            for (int pc = 0; pc < cc && same < 0; ++pc) { const int ps = pc < j.ncomp ? pc : 0; if ((ac ? j.ta[ps] : j.td[ps]) == th) same = 2 * pc + ac; }
            if (same < 0) { jbuild_slot(t, slot, j.bits[ac][th], j.vals[ac][th], &nsub); continue; }
            memcpy(t->S.lut[slot], t->S.lut[same], sizeof t->S.lut[slot]); memcpy(t->W.lut[slot], t->W.lut[same], sizeof t->W.lut[slot]);
            memcpy(t->maxcode[slot], t->maxcode[same], sizeof t->maxcode[slot]); memcpy(t->valoff[slot], t->valoff[same], sizeof t->valoff[slot]);
            memcpy(t->vals[slot], t->vals[same], 256);
        }
        memcpy(t->q[cc], j.q[j.tq[cs]], 128);
    }
}

// Entropy segment -> staging buffer without its byte stuffing: FF00 -> FF; RSTn markers (streams with a restart interval) are taken
// out and the offsets behind them listed; any other marker ends the data (libjpeg stops reading there too).  memchr + memcpy of the
// runs between the FFs (one FF per ~256 bytes of entropy-coded data).  Returns the destuffed length; *nrst > maxr: too many markers.
static size_t jdestuff(uint8_t *dst, const uint8_t *src, size_t n, bool restarts, uint32_t *rst, uint32_t maxr, uint32_t *nrst)
{
    size_t o = 0;
    *nrst = 0;
    while (n) {
        const uint8_t *f = (const uint8_t *)memchr(src, 0xFF, n);
        if (!f) { memcpy(dst + o, src, n); o += n; break; }
        const size_t run = (size_t)(f - src);
        memcpy(dst + o, src, run); o += run; src = f; n -= run;
        if (n < 2) { dst[o++] = 0xFF; break; }                  // an FF in front of the zero padding counts as a stuffed one
        const uint8_t nx = src[1];
        if (nx == 0) dst[o++] = 0xFF;
        else if (restarts && (nx & 0xF8u) == 0xD0u) { if (*nrst < maxr) rst[*nrst] = (uint32_t)o; ++*nrst; }
        else break;
        src += 2; n -= 2;
    }
    return o;
}

static jpeg_geom jgeom(const jhost &j)
{
    jpeg_geom g;
    memset(&g, 0, sizeof g);
    g.w = j.w; g.h = j.h; g.ncomp = j.ncomp; g.hmax = j.hs[0]; g.vmax = j.vs[0];
    g.mcux = (j.w + 8 * g.hmax - 1) / (8 * g.hmax); g.mcuy = (j.h + 8 * g.vmax - 1) / (8 * g.vmax);
    int off = 0;
    size_t po = 0;
    for (int c = 0; c < j.ncomp; ++c) {
        g.comp_off[c] = off; g.comp_nb[c] = j.hs[c] * j.vs[c];
        for (int k = 0; k < g.comp_nb[c]; ++k) g.blk_comp[off + k] = c;
        off += g.comp_nb[c];
        g.pw[c] = g.mcux * j.hs[c] * 8; g.ph[c] = g.mcuy * j.vs[c] * 8;
        g.plane_off[c] = po;
        po += ((size_t)g.pw[c] * g.ph[c] + 255) & ~(size_t)255;
    }
    g.bpm = off; g.nblk = g.mcux * g.mcuy * g.bpm;
    g.plane_stride = po;
    return g;
}

extern "C" int ofk_jpeg_info(const uint8_t *jpeg, size_t nbytes, int *h, int *w, int *components)
{
    jhost j;
    if (jparse(jpeg, nbytes, &j)) return OFK_E_INVALID;
    if (h) *h = j.h;
    if (w) *w = j.w;
    if (components) *components = j.ncomp;
    return OFK_OK;
}

extern "C" int ofk_jpeg_destuff(const uint8_t *jpeg, size_t nbytes, uint8_t *out, size_t out_capacity, size_t *out_len, uint32_t *rst, int rst_capacity, int *nrst)
{
    jhost j;
    if (!out || !out_len || !nrst || (rst_capacity > 0 && !rst) || jparse(jpeg, nbytes, &j)) return OFK_E_INVALID;
    if (out_capacity < j.ent_len) return OFK_E_INVALID;
    uint32_t n = 0;
    *out_len = jdestuff(out, j.ent, j.ent_len, j.ri > 0, rst, rst_capacity > 0 ? (uint32_t)rst_capacity : 0u, &n);
    *nrst = (int)n;
    return n > (uint32_t)(rst_capacity > 0 ? rst_capacity : 0) ? OFK_E_INVALID : OFK_OK;
}

static size_t jup(size_t v, size_t a) { return (v + a - 1) / a * a; }
#define TRY_J(expr) do { int rc_ = (expr); if (rc_ != OFK_OK) return rc_; } while (0)

// ---- two-phase ingest.  Phase 1 (jstage_fill, host + copy engine): marker parse, tables and entropy segments into a pinned
// staging slot, one asynchronous H2D copy on the context's copy stream.  Phase 2 (jdecode_staged, device): the decoder passes on
// the context's stream, behind the copy.  Two slots: a caller thread stages batch k + 1 (ofk_jpeg_stage) while the main thread is
// inside the decode of batch k (ofk_pairs_upload_staged) - host parse, staging copy and PCIe transfer then hide behind the GPU's work.
struct jstage {
    void *host; size_t host_bytes;          // pinned
    void *dev; size_t dev_bytes;            // device copy of the slot
    size_t stage_bytes, tab_bytes, rst_bytes;
    jpeg_geom g;
    int batch, nch_max, valid, restarts;     // restarts: some stream has a restart interval (the RST instantiation of the decoder passes)
    hipEvent_t copied;                      // H2D of this slot complete
    char err[256];                          // last staging error of THIS slot: ofk_jpeg_stage may run on a helper thread while the owner
                                            // thread writes the context's message (ofk_jpeg_stage_error reads this one)
};
struct jstages {
    jstage slot[2]; hipStream_t copy;
    hipStream_t dec;                          // the decoder passes of a double-buffered frame-pair ingest: beside the pipeline run of the batch before
    int *hmap, *hmap_dev; size_t hmap_ints;   // host memory the device writes its convergence flags / end-of-stream records into (pinned,
                                              // mapped): the host reads them after a stream wait, no copy engine in the round trip - a D2H
                                              // copy queues behind the other slot's 200 MB H2D transfer and stalls the decoder for its length
};

__global__ void k_jpeg_ints_to_host(const int *__restrict__ src, int *__restrict__ dst, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

static int jhmap(ofk_ctx *c, jstages *js, size_t ints)
{
    if (js->hmap_ints >= ints) return OFK_OK;
    if (js->hmap) { hipHostFree(js->hmap); js->hmap = nullptr; js->hmap_ints = 0; }
    ints = (ints + 4095) / 4096 * 4096;
    OFK_HIP(c, hipHostMalloc((void **)&js->hmap, ints * 4, hipHostMallocMapped));
    OFK_HIP(c, hipHostGetDevicePointer((void **)&js->hmap_dev, js->hmap, 0));
    js->hmap_ints = ints;
    return OFK_OK;
}

static void jstages_free(jstages *js)
{
    for (int k = 0; k < 2; ++k) {
        if (js->slot[k].host) hipHostFree(js->slot[k].host);
        if (js->slot[k].dev) hipFree(js->slot[k].dev);
        if (js->slot[k].copied) hipEventDestroy(js->slot[k].copied);
    }
    if (js->copy) hipStreamDestroy(js->copy);
    if (js->dec) hipStreamDestroy(js->dec);
    if (js->hmap) hipHostFree(js->hmap);
    free(js);
}

static jstages *jstages_of(ofk_ctx *c)
{
    if (!c->jstage) {
        jstages *js = (jstages *)calloc(1, sizeof(jstages));
        if (!js) return nullptr;
        bool ok = hipStreamCreateWithFlags(&js->copy, hipStreamNonBlocking) == hipSuccess &&
                  hipStreamCreateWithFlags(&js->dec, hipStreamNonBlocking) == hipSuccess &&
                  true;
        for (int k = 0; k < 2 && ok; ++k) ok = hipEventCreateWithFlags(&js->slot[k].copied, hipEventDisableTiming) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); jstages_free(js); return nullptr; }      // whatever was created so far is released
        c->jstage = js;
    }
    return (jstages *)c->jstage;
}

void ofk_jpeg_release(ofk_ctx *c)
{
    if (!c->jstage) return;
    jstages_free((jstages *)c->jstage);
    c->jstage = nullptr;
}

// Phase 1.  Touches only the slot, the copy stream and the host: safe to call from a second thread while the context's owner is
// inside a decode of the OTHER slot.
static int jstage_fill(ofk_ctx *c, int slot, const uint8_t *const *jpeg, const size_t *nbytes, int batch)
{
    if (!jpeg || !nbytes || batch < 1 || slot < 0 || slot > 1) return ofk_fail(c, OFK_E_INVALID, "ofk_jpeg: bad argument");
    jstages *js = jstages_of(c);
    if (!js) return ofk_fail(c, OFK_E_HIP, "ofk_jpeg: staging set-up failed");
    jstage &J = js->slot[slot];
    J.valid = 0;
    J.err[0] = 0;
    // errors go to the SLOT's message (this function may run beside the owner thread, which owns the context's message)
#define fail(code, ...) (snprintf(J.err, sizeof(J.err), __VA_ARGS__), (code))
    jhost *jh = (jhost *)malloc(sizeof(jhost) * (size_t)batch);
    if (!jh) return fail(OFK_E_INVALID, "ofk_jpeg: out of host memory");
    size_t ent_total = 0;
    for (int b = 0; b < batch; ++b) {
        const char *err = jparse(jpeg[b], nbytes[b], &jh[b]);
        if (err) { const int rc = fail(OFK_E_INVALID, "ofk_jpeg: stream %d: %s", b, err); free(jh); return rc; }
        if (b && (jh[b].w != jh[0].w || jh[b].h != jh[0].h || jh[b].ncomp != jh[0].ncomp || jh[b].hs[0] != jh[0].hs[0] || jh[b].vs[0] != jh[0].vs[0])) {
            const int rc = fail(OFK_E_INVALID, "ofk_jpeg: stream %d differs in size or sampling from stream 0 (one geometry per batch)", b);
            free(jh); return rc;
        }
        ent_total += JPAD(jh[b].ent_len);
    }
    if (ent_total >= (1ull << 32)) { free(jh); return fail(OFK_E_INVALID, "ofk_jpeg: more than 4 GiB of entropy data in one batch"); }
    jpeg_geom g = jgeom(jh[0]);
    g.jch = JCH;
    while (g.jch > JCH_MIN && ent_total / (size_t)g.jch < 131072) g.jch >>= 1;   // keep >= 128 k decoder threads if the data allows
    if (g_ofk_tuning.jpeg_chunk) g.jch = g_ofk_tuning.jpeg_chunk;   // ofk_set_tuning("jpeg_chunk")
    // restart boundaries: at most one in front of every interval but the first
    uint32_t maxr = 0;
    for (int b = 0; b < batch; ++b)
        if (jh[b].ri > 0) { const uint32_t iv = (uint32_t)((g.mcux * g.mcuy + jh[b].ri - 1) / jh[b].ri); if (iv > maxr) maxr = iv; }
    // pinned staging: tables, restart boundaries, entropy segments - one H2D copy
    const size_t tab_bytes = jup(sizeof(jpeg_tab) * (size_t)batch, 256), rst_bytes = jup((size_t)maxr * 4 * (size_t)batch, 256);
    const size_t stage = tab_bytes + rst_bytes + ent_total;
    if (J.copied) (void)hipEventSynchronize(J.copied);           // the slot's previous transfer has left the pinned buffer
    if (J.host_bytes < stage) {
        if (J.host) { hipHostFree(J.host); J.host = nullptr; J.host_bytes = 0; }
        if (hipHostMalloc(&J.host, jup(stage, 1 << 20), hipHostMallocDefault) != hipSuccess) { free(jh); return fail(OFK_E_HIP, "ofk_jpeg: pinned staging allocation failed"); }
        J.host_bytes = jup(stage, 1 << 20);
    }
    if (J.dev_bytes < stage) {
        if (J.dev) { hipFree(J.dev); J.dev = nullptr; J.dev_bytes = 0; }
        if (hipMalloc(&J.dev, jup(stage, 1 << 20)) != hipSuccess) { free(jh); return fail(OFK_E_HIP, "ofk_jpeg: device staging allocation failed"); }
        J.dev_bytes = jup(stage, 1 << 20);
    }
    jpeg_tab *ht = (jpeg_tab *)J.host;
    uint32_t *hrst = (uint32_t *)((uint8_t *)J.host + tab_bytes);
    uint8_t *hent = (uint8_t *)J.host + tab_bytes + rst_bytes;
    // tables and entropy segments into the staging buffer: ~0.4 MB per 1080p frame, destuffed on the way (jdestuff), spread over a
    // few host threads (one thread moves ~10 GB/s, which would make this copy the slowest stage of the ingest)
    size_t *eoff = (size_t *)malloc(sizeof(size_t) * (size_t)batch);
    if (!eoff) { free(jh); return fail(OFK_E_INVALID, "ofk_jpeg: out of host memory"); }
    size_t eo = 0;
    for (int b = 0; b < batch; ++b) { eoff[b] = eo; eo += JPAD(jh[b].ent_len); }
    // Every worker takes a contiguous range of images and hands it to the copy engine in a few pieces as it goes (the entropy bytes of
    // consecutive images are contiguous in the staging buffer): the H2D transfer - 8 ms for the 460 MB of 1024 1080p frames - runs
    // beside the destuffing instead of behind it.  One copy at the end put staging + transfer (15-21 ms) on the critical path of a
    // double-buffered ingest whose decode takes 13.5 ms.
    std::atomic<int> bad_rst(-1), copy_failed(0);
    uint8_t *dent_dst = (uint8_t *)J.dev + tab_bytes + rst_bytes;
    auto stage_range = [&](int b0, int b1, int pieces) {
        if (pieces > 1 && hipSetDevice(c->device) != hipSuccess) copy_failed = 1;
        const int per = (b1 - b0 + pieces - 1) / pieces;
        for (int p0 = b0; p0 < b1; p0 += per) {
            const int p1 = p0 + per < b1 ? p0 + per : b1;
            for (int b = p0; b < p1; ++b) {
                const jhost &j = jh[b];
                jpeg_tab *t = ht + b;
                jbuild_tables(t, j);
                t->rst_off = (uint32_t)((size_t)b * maxr);
                const size_t len = jdestuff(hent + eoff[b], j.ent, j.ent_len, j.ri > 0, hrst + t->rst_off, maxr, &t->nrst);
                if (t->nrst > maxr) { t->nrst = maxr; bad_rst = b; }
                memset(hent + eoff[b] + len, 0, JPAD(j.ent_len) - len);
                t->ent_off = (uint32_t)eoff[b]; t->ent_len = (uint32_t)len;
                t->nch = (int)(len / (size_t)g.jch) + 1; t->ri = j.ri;
            }
            if (pieces > 1) {
                const size_t lo = eoff[p0], hi = eoff[p1 - 1] + JPAD(jh[p1 - 1].ent_len);
                if (hipMemcpyAsync(dent_dst + lo, hent + lo, hi - lo, hipMemcpyHostToDevice, js->copy) != hipSuccess) copy_failed = 1;
            }
        }
    };
    unsigned nthr = std::thread::hardware_concurrency() / 2;
    nthr = nthr < 1 ? 1 : nthr > 8 ? 8 : nthr;
    if ((unsigned)batch < nthr) nthr = (unsigned)batch;
    const bool pieces = !(ent_total < (4u << 20) || nthr == 1);
    if (!pieces) stage_range(0, batch, 1);
    else {
        std::vector<std::thread> pool;
        auto first = [&](unsigned k) { return (int)((size_t)batch * k / nthr); };
        for (unsigned k = 1; k < nthr; ++k) pool.emplace_back(stage_range, first(k), first(k + 1), 4);
        stage_range(0, first(1), 4);
        for (auto &th : pool) th.join();
    }
    int nch_max = 1, any_rst = 0;
    for (int b = 0; b < batch; ++b) { if (ht[b].nch > nch_max) nch_max = ht[b].nch; any_rst |= jh[b].ri > 0; }
    if (bad_rst >= 0) { const int rc = fail(OFK_E_INVALID, "ofk_jpeg: stream %d: more restart markers than restart intervals (corrupt)", (int)bad_rst); free(eoff); free(jh); return rc; }
    free(eoff);
    free(jh);
    // tables and restart boundaries (and, for a small batch, everything) in one copy behind the pieces
    if (copy_failed || hipMemcpyAsync(J.dev, J.host, pieces ? tab_bytes + rst_bytes : stage, hipMemcpyHostToDevice, js->copy) != hipSuccess ||
        hipEventRecord(J.copied, js->copy) != hipSuccess) {
        (void)hipGetLastError();
        return fail(OFK_E_HIP, "ofk_jpeg: staging copy to the device failed");
    }
    J.stage_bytes = stage; J.tab_bytes = tab_bytes; J.rst_bytes = rst_bytes; J.g = g; J.batch = batch; J.nch_max = nch_max; J.restarts = any_rst; J.valid = 1;
    return OFK_OK;
}

#undef fail

// Phase 2: the staged streams [0, split) go to dst ([.][dst_stride] BGR8), the streams [split, batch) to dst2 (split >= batch: all
// to dst); with dst == NULL into the context's scratch (*out / *out_stride tell where).  Synchronous on the context's stream.
static int jdecode_staged(ofk_ctx *c, int slot, uint8_t *dst, uint8_t *dst2, int split, size_t dst_stride, size_t dst_capacity_px,
                          int *h_out, int *w_out, uint8_t **out, size_t *out_stride, bool own_stream = false, const hipEvent_t *wait_ev = nullptr, int nwait = 0,
                          int as_gray = 0)
{
    jstages *js = (jstages *)c->jstage;
    if (!js || slot < 0 || slot > 1 || !js->slot[slot].valid) return ofk_fail(c, OFK_E_INVALID, "ofk_jpeg: slot %d holds no staged streams (ofk_jpeg_stage)", slot);
    jstage &J = js->slot[slot];
    J.valid = 0;                                                 // a slot is decoded once
    const jpeg_geom g = J.g;
    const int batch = J.batch, nch_max = J.nch_max;
    if (dst && (size_t)g.w * g.h > dst_capacity_px) return ofk_fail(c, OFK_E_INVALID, "ofk_jpeg: %dx%d frames exceed the destination", g.w, g.h);
    // a caller-supplied destination is one of the context's resident buffers: max_batch images each (checked BEFORE any kernel is queued -
    // the colour pass would write past them)
    if (dst && ((split < batch ? split : batch) > c->max_batch || (dst2 && split < batch && batch - split > c->max_batch)))
        return ofk_fail(c, OFK_E_INVALID, "ofk_jpeg: %d staged streams exceed the context's %d images per frame set", batch, c->max_batch);
    // device scratch
    const size_t B = (size_t)batch;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += jup(bytes, 256); return o; };
    const size_t o_state = take(B * nch_max * 8), o_used = take(B * nch_max * 8), o_count = take(B * nch_max * 4),
                 o_base = take(B * (nch_max + 1) * 4), o_flags = take(JMAX_ITERS * 4 + B * 8), o_coef = take(B * g.nblk * 130),
                 o_planes = take(B * g.plane_stride);
    const size_t own_stride = jup((size_t)g.w * g.h * 3, 256);
    const size_t o_out = dst ? 0 : take(B * own_stride);
    const int rc = ofk_need_scratch(c, off);
    if (rc != OFK_OK) return rc;
    char *S = (char *)c->scratch;
    if (!dst) { dst = (uint8_t *)(S + o_out); dst_stride = own_stride; dst2 = nullptr; split = batch; }
    if (!dst2 || split >= batch) { dst2 = dst; split = batch; }
    if (out) *out = dst;
    if (out_stride) *out_stride = dst_stride;
    const jpeg_tab *dt = (const jpeg_tab *)J.dev;
    const uint32_t *drst = (const uint32_t *)((const uint8_t *)J.dev + J.tab_bytes);
    const uint8_t *dent = (const uint8_t *)J.dev + J.tab_bytes + J.rst_bytes;
    auto sync_pass = J.restarts ? k_jpeg_sync<true> : k_jpeg_sync<false>;
    auto sync_tail = J.restarts ? k_jpeg_sync_tail<true> : k_jpeg_sync_tail<false>;
    const dim3 tgrid((nch_max + JTAIL - 1) / JTAIL, batch);
    auto write_pass = J.restarts ? k_jpeg_write<true> : k_jpeg_write<false>;
    unsigned long long *state = (unsigned long long *)(S + o_state), *used = (unsigned long long *)(S + o_used);
    int *count = (int *)(S + o_count), *base = (int *)(S + o_base), *flags = (int *)(S + o_flags), *endinfo = flags + JMAX_ITERS;
    int16_t *coef = (int16_t *)(S + o_coef), *dcarr = coef + B * g.nblk * 64;     // coefficient blocks, then the dense DC array
    uint8_t *planes = (uint8_t *)(S + o_planes);
    // own_stream: nothing the context's streams hold touches the destination (a frame-pair set of its own) or this scratch, so the passes
    // run on the ingest stream, beside the pipeline run of the batch before; wait_ev = the last reader of the destination
    hipStream_t st = own_stream ? js->dec : c->stream;
    for (int k = 0; wait_ev && k < nwait; ++k) if (wait_ev[k]) OFK_HIP(c, hipStreamWaitEvent(st, wait_ev[k], 0));
    OFK_HIP(c, hipMemsetAsync(flags, 0, JMAX_ITERS * 4 + B * 8, st));
    OFK_HIP(c, hipStreamWaitEvent(st, J.copied, 0));             // tables and entropy data are on the device from here on
    const dim3 dgrid((nch_max + JTPB - 1) / JTPB, batch);
    TRY_J(jhmap(c, js, JMAX_ITERS + 2 * B));
    volatile int *hflags = js->hmap;
    int iter = 0;
    hipLaunchKernelGGL(sync_pass, dgrid, dim3(JTPB), JSYNC_LDS_PAD, st, dt, dent, drst, g, nch_max, state, used, count, iter, flags);
    bool converged = nch_max == 1;
    while (!converged) {
        const int first = iter + 1;
        // Two iterations settle most chunks, the stragglers take five to eleven (1080p, quality 80); an iteration past the fixed
        // point costs ~10 us (its workgroups leave at once), a host round trip far more: seven iterations go out before the first look.
        const int burst = iter == 0 ? 7 : 4;
        for (int k = 0; k < burst; ++k) {
            ++iter;
            if (iter >= JMAX_ITERS) OFK_HIP(c, hipMemsetAsync(flags + JMAX_ITERS - 1, 0, 4, st));
            if (iter >= 2) hipLaunchKernelGGL(sync_tail, tgrid, dim3(JTPB), 0, st, dt, dent, drst, g, nch_max, state, used, count, iter, flags);
            else hipLaunchKernelGGL(sync_pass, dgrid, dim3(JTPB), JSYNC_LDS_PAD, st, dt, dent, drst, g, nch_max, state, used, count, iter, flags);
        }
        hipLaunchKernelGGL(k_jpeg_ints_to_host, dim3(1), dim3(64), 0, st, flags, js->hmap_dev, JMAX_ITERS);
        OFK_HIP(c, hipStreamSynchronize(st));
        for (int k = first; k <= iter; ++k)
            if (!hflags[k < JMAX_ITERS ? k : JMAX_ITERS - 1]) converged = true;
        if (!converged && iter > nch_max + 2) return ofk_fail(c, OFK_E_INVALID, "ofk_jpeg: entropy decoders did not converge");
    }
    hipLaunchKernelGGL(k_jpeg_scan, dim3(batch), dim3(1024), 0, st, dt, nch_max, count, base);
    hipLaunchKernelGGL(k_jpeg_zero_heads, dim3((nch_max + 31) / 32, batch), dim3(256), 0, st, dt, g, nch_max, state, base, coef);
    hipLaunchKernelGGL(write_pass, dim3((nch_max + JTPW - 1) / JTPW, batch), dim3(JTPW), 0, st, dt, dent, drst, g, nch_max, state, base, coef, dcarr, endinfo);
    hipLaunchKernelGGL(k_jpeg_dc, dim3(batch, g.ncomp), dim3(1024), 0, st, dt, g, dcarr);
    if (g.ncomp > 1) hipLaunchKernelGGL(k_jpeg_idct<false>, dim3(g.mcuy, batch), dim3(256), 0, st, dt, g, coef, dcarr, planes, dst, dst2, split, dst_stride, as_gray);
    hipLaunchKernelGGL(k_jpeg_idct<true>, dim3(g.mcuy, batch), dim3(256), 0, st, dt, g, coef, dcarr, planes, dst, dst2, split, dst_stride, as_gray);
    hipLaunchKernelGGL(k_jpeg_ints_to_host, dim3(4), dim3(256), 0, st, endinfo, js->hmap_dev + JMAX_ITERS, 2 * batch);
    hipError_t e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return ofk_fail(c, OFK_E_HIP, "ofk_jpeg: %s", hipGetErrorString(e));
    const volatile int *hend = js->hmap + JMAX_ITERS;
    for (int b = 0; b < batch; ++b)
        if (!hend[2 * b + 1]) return ofk_fail(c, OFK_E_INVALID, "ofk_jpeg: stream %d: entropy data ends before the last block (truncated or corrupt)", b);
    if (h_out) *h_out = g.h;
    if (w_out) *w_out = g.w;
    return OFK_OK;
}

// Decodes `batch` streams of one geometry into dst (device, [batch][dst_stride] BGR8), or, with dst == NULL, into the context's
// scratch (*out / *out_stride tell where).  Synchronous on the context's stream.  (Both phases, staging slot 0.)
int ofk_jpeg_decode_device(ofk_ctx *c, const uint8_t *const *jpeg, const size_t *nbytes, int batch, uint8_t *dst, size_t dst_stride,
                           size_t dst_capacity_px, int *h_out, int *w_out, uint8_t **out, size_t *out_stride)
{
    int rc = jstage_fill(c, 0, jpeg, nbytes, batch);
    if (rc != OFK_OK) return c->jstage && ((jstages *)c->jstage)->slot[0].err[0] ? ofk_fail(c, rc, "%s", ((jstages *)c->jstage)->slot[0].err) : rc;   // owner thread: its message
    return jdecode_staged(c, 0, dst, nullptr, batch, dst_stride, dst_capacity_px, h_out, w_out, out, out_stride);
}

int ofk_jpeg_stage_streams(ofk_ctx *c, int slot, const uint8_t *const *jpeg, const size_t *nbytes, int count) { return jstage_fill(c, slot, jpeg, nbytes, count); }

// the slot's own message ("" when its last staging succeeded)
const char *ofk_jpeg_slot_error(const ofk_ctx *c, int slot)
{
    const jstages *js = c ? (const jstages *)c->jstage : nullptr;
    return (js && slot >= 0 && slot <= 1) ? js->slot[slot].err : "";
}

int ofk_jpeg_decode_staged_pairs(ofk_ctx *c, int slot, uint8_t *dst_prev, uint8_t *dst_next, size_t dst_stride, size_t dst_capacity_px, int *batch_out,
                                 int *h_out, int *w_out, const hipEvent_t *wait_before_writing, int nwait, int as_gray)
{
    jstages *js = (jstages *)c->jstage;
    if (!js || slot < 0 || slot > 1 || !js->slot[slot].valid) return ofk_fail(c, OFK_E_INVALID, "ofk_pairs_upload_staged: slot %d holds no staged streams (ofk_jpeg_stage)", slot);
    const int count = js->slot[slot].batch;
    if (count & 1) { js->slot[slot].valid = 0; return ofk_fail(c, OFK_E_INVALID, "ofk_pairs_upload_staged: %d staged streams are not pairs (previous frames first, then the next frames)", count); }
    if (count / 2 > c->max_batch) { js->slot[slot].valid = 0; return ofk_fail(c, OFK_E_INVALID, "ofk_pairs_upload_staged: %d pairs exceed the context (%d)", count / 2, c->max_batch); }
    if (batch_out) *batch_out = count / 2;
    return jdecode_staged(c, slot, dst_prev, dst_next, count / 2, dst_stride, dst_capacity_px, h_out, w_out, nullptr, nullptr, wait_before_writing != nullptr,
                          wait_before_writing, nwait, as_gray);
}
