/* jpeg_oracle.c — CPU restatement of baseline JPEG decoding as OpenCV's imdecode performs it for the reference's
 * CompressedImage ingest (velocity_measurment_node.py:112, cv_bridge.compressed_imgmsg_to_cv2 -> cv::imdecode -> libjpeg).
 *
 * TEST INFRASTRUCTURE ONLY: linked into oracle/_build/liboracle.so, never into the product library.
 *
 * The algorithm lives in a third-party dependency of the reference (libjpeg / libjpeg-turbo behind OpenCV); it is restated
 * here from its published form with libjpeg's default decompression parameters (the ones cv::imdecode uses):
 *   - sequential Huffman entropy decoding              (ITU-T T.81 Annex F.2.2; libjpeg jdhuff.c decode_mcu)
 *   - dequantisation + "ISLOW" integer inverse DCT      (libjpeg jidctint.c jpeg_idct_islow: CONST_BITS 13, PASS1_BITS 2)
 *   - "fancy" (triangle filter) chroma upsampling       (libjpeg jdsample.c h2v1_fancy_upsample / h2v2_fancy_upsample)
 *   - YCbCr -> RGB with 16-bit fixed-point tables       (libjpeg jdcolor.c build_ycc_rgb_table / ycc_rgb_convert)
 * Pinned by tests/golden/jpeg_golden.npz: JPEG byte streams with the pixels libjpeg-turbo returned for them (generated through
 * Pillow, which drives the same library with the same defaults; tests/golden/make_golden_jpeg.py).
 *
 * Scope (everything else is refused with a negative return code): 8-bit baseline / extended-sequential Huffman (SOF0, SOF1), one
 * interleaved scan, 1 component (gray) or 3 components YCbCr with luma sampling 1x1 (4:4:4), 2x1 (4:2:2) or 2x2 (4:2:0) and 1x1
 * chroma.  Restart intervals (DRI / RSTn, T.81 F.2.2.4 + E.2.4; jdhuff.c process_restart): the bit buffer is dropped, the marker
 * skipped and the DC predictors reset every `ri` MCUs. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int present;
    uint8_t bits[17], vals[256];
    int mincode[17], maxcode[18], valptr[17];
} orc_htab;

typedef struct {
    int w, h, ncomp;
    int hs[3], vs[3], tq[3], td[3], ta[3];
    int hmax, vmax, mcux, mcuy, bpm;            /* MCUs per row / column, blocks per MCU */
    int ri;                                     /* restart interval in MCUs (0 = none) */
    uint16_t q[4][64];                          /* natural order */
    int qok[4];
    orc_htab dc[4], ac[4];
    const uint8_t *ent; size_t ent_len;         /* entropy-coded segment (still byte-stuffed) */
} orc_jpeg;

static const uint8_t ZZ[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                               35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static void derive(orc_htab *t)
{   /* T.81 Annex C / jdhuff.c jpeg_make_d_derived_tbl */
    int code = 0, p = 0;
    for (int l = 1; l <= 16; ++l) {
        t->valptr[l] = p;
        t->mincode[l] = code;
        p += t->bits[l];
        code += t->bits[l];
        t->maxcode[l] = t->bits[l] ? code - 1 : -1;
        code <<= 1;
    }
    t->maxcode[17] = 0x7fffffff;
}

static int parse(const uint8_t *d, size_t n, orc_jpeg *j)
{
    memset(j, 0, sizeof *j);
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return -1;
    size_t i = 2;
    int sof = 0;
    while (i + 4 <= n) {
        if (d[i] != 0xFF) return -2;
        while (i < n && d[i] == 0xFF) ++i;                      /* fill bytes */
        if (i >= n) return -2;
        const int m = d[i++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) return -3;                               /* EOI before a scan */
        if (i + 2 > n) return -2;
        const size_t L = ((size_t)d[i] << 8) | d[i + 1];
        if (L < 2 || i + L > n) return -2;
        const uint8_t *s = d + i + 2; const size_t sl = L - 2;
        if (m == 0xDB) {                                        /* DQT */
            size_t k = 0;
            while (k < sl) {
                const int pq = s[k] >> 4, tq = s[k] & 15; ++k;
                if (tq > 3 || pq > 1 || k + (pq ? 128 : 64) > sl) return -4;
                for (int z = 0; z < 64; ++z) { j->q[tq][ZZ[z]] = pq ? (uint16_t)((s[k] << 8) | s[k + 1]) : s[k]; k += pq ? 2 : 1; }
                j->qok[tq] = 1;
            }
        } else if (m == 0xC4) {                                 /* DHT */
            size_t k = 0;
            while (k + 17 <= sl) {
                const int tc = s[k] >> 4, th = s[k] & 15; ++k;
                if (tc > 1 || th > 3) return -5;
                orc_htab *t = tc ? &j->ac[th] : &j->dc[th];
                int cnt = 0;
                t->bits[0] = 0;
                for (int l = 1; l <= 16; ++l) { t->bits[l] = s[k++]; cnt += t->bits[l]; }
                if (cnt > 256 || k + cnt > sl) return -5;
                memcpy(t->vals, s + k, cnt); k += cnt;
                derive(t); t->present = 1;
            }
        } else if (m == 0xC0 || m == 0xC1) {                    /* SOF0 / SOF1 */
            if (sl < 6 || s[0] != 8) return -6;
            j->h = (s[1] << 8) | s[2]; j->w = (s[3] << 8) | s[4]; j->ncomp = s[5];
            if ((j->ncomp != 1 && j->ncomp != 3) || sl < 6 + 3 * (size_t)j->ncomp || j->h < 1 || j->w < 1) return -6;
            for (int c = 0; c < j->ncomp; ++c) { j->hs[c] = s[7 + 3 * c] >> 4; j->vs[c] = s[7 + 3 * c] & 15; j->tq[c] = s[8 + 3 * c]; if (j->tq[c] > 3) return -6; }
            sof = 1;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return -7;                                          /* progressive, lossless, arithmetic ... */
        } else if (m == 0xDD) {
            if (sl < 2) return -8;
            j->ri = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {                                 /* SOS */
            if (!sof || sl < 1 || s[0] != j->ncomp || sl < 4 + 2 * (size_t)j->ncomp) return -9;
            for (int c = 0; c < j->ncomp; ++c) { j->td[c] = s[2 + 2 * c] >> 4; j->ta[c] = s[2 + 2 * c] & 15; if (j->td[c] > 3 || j->ta[c] > 3) return -9; }
            j->ent = d + i + L;
            size_t e = i + L;
            while (e + 1 < n && !(d[e] == 0xFF && d[e + 1] != 0x00 && !(d[e + 1] >= 0xD0 && d[e + 1] <= 0xD7))) ++e;   /* up to the next marker that is not RSTn */
            if (e + 1 >= n) e = n;
            j->ent_len = e - (i + L);
            break;
        }
        i += L;
    }
    if (!j->ent) return -10;
    if (j->ncomp == 1) { j->hs[0] = j->vs[0] = 1; }
    else {
        if (j->hs[1] != 1 || j->vs[1] != 1 || j->hs[2] != 1 || j->vs[2] != 1) return -11;
        if (!((j->hs[0] == 1 && j->vs[0] == 1) || (j->hs[0] == 2 && j->vs[0] == 1) || (j->hs[0] == 2 && j->vs[0] == 2))) return -11;
    }
    j->hmax = j->hs[0]; j->vmax = j->vs[0];
    j->mcux = (j->w + 8 * j->hmax - 1) / (8 * j->hmax); j->mcuy = (j->h + 8 * j->vmax - 1) / (8 * j->vmax);
    j->bpm = 0;
    for (int c = 0; c < j->ncomp; ++c) {
        j->bpm += j->hs[c] * j->vs[c];
        if (!j->qok[j->tq[c]] || !j->dc[j->td[c]].present || !j->ac[j->ta[c]].present) return -12;
    }
    return 0;
}

typedef struct { const uint8_t *p; size_t n, pos; uint64_t buf; int nb; int hit_marker; } bitrd;

static void fill(bitrd *b)
{
    while (b->nb <= 48) {
        int c = 0;
        if (!b->hit_marker && b->pos < b->n) {
            c = b->p[b->pos++];
            if (c == 0xFF) {
                if (b->pos < b->n && b->p[b->pos] == 0) ++b->pos;                 /* stuffed zero */
                else { b->hit_marker = 1; c = 0; }                               /* libjpeg feeds zeros past a marker */
            }
        }
        b->buf = (b->buf << 8) | (uint64_t)c; b->nb += 8;
    }
}
static int getbits(bitrd *b, int s) { if (!s) return 0; fill(b); b->nb -= s; return (int)((b->buf >> b->nb) & ((1u << s) - 1)); }
static int decode(bitrd *b, const orc_htab *t)
{
    int code = getbits(b, 1), l = 1;
    while (l <= 16 && code > t->maxcode[l]) { code = (code << 1) | getbits(b, 1); ++l; }
    if (l > 16) return 0;                                                         /* jdhuff.c: bad code -> symbol 0 */
    return t->vals[(t->valptr[l] + code - t->mincode[l]) & 255];
}
static int extend(int r, int s) { return r < (1 << (s - 1)) ? r + (int)(~0u << s) + 1 : r; }

/* coefficient blocks in decode order, natural positions, DC already predicted: coef[nblocks][64] */
static int entropy(const orc_jpeg *j, int16_t *coef)
{
    bitrd b = {j->ent, j->ent_len, 0, 0, 0, 0};
    int pred[3] = {0, 0, 0};
    const size_t nmcu = (size_t)j->mcux * j->mcuy;
    memset(coef, 0, nmcu * j->bpm * 64 * sizeof(int16_t));
    int16_t *blk = coef;
    for (size_t m = 0; m < nmcu; ++m) {
        if (j->ri && m && m % (size_t)j->ri == 0) {             /* process_restart */
            b.nb = 0; b.buf = 0;
            if (b.hit_marker && b.pos < b.n && b.p[b.pos] >= 0xD0 && b.p[b.pos] <= 0xD7) { ++b.pos; b.hit_marker = 0; }
            pred[0] = pred[1] = pred[2] = 0;
        }
        for (int c = 0; c < j->ncomp; ++c)
            for (int k2 = 0; k2 < j->hs[c] * j->vs[c]; ++k2, blk += 64) {
                int s = decode(&b, &j->dc[j->td[c]]);
                if (s) { const int r = getbits(&b, s & 15); s = extend(r, s & 15); }
                pred[c] += s;
                blk[0] = (int16_t)pred[c];
                for (int k = 1; k < 64; ++k) {
                    const int rs = decode(&b, &j->ac[j->ta[c]]);
                    const int r = rs >> 4; s = rs & 15;
                    if (s) {
                        k += r;
                        const int v = extend(getbits(&b, s), s);
                        blk[ZZ[k > 63 ? 63 : k]] = (int16_t)v;
                    } else {
                        if (r != 15) break;
                        k += 15;
                    }
                }
            }
    }
    return 0;
}

#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172
#define DESCALE(x, n) (((x) + (1 << ((n) - 1))) >> (n))

static uint8_t range_limit(int x)
{   /* jdmaster.c prepare_range_limit_table, post-IDCT half: index (x & 1023) */
    const int i = x & 1023;
    return (uint8_t)(i < 128 ? i + 128 : i < 512 ? 255 : i < 896 ? 0 : i - 896);
}

static void idct1d(const int *in, int stride, int *o)
{   /* one 8-point pass of jidctint.c, results before the descale: o[0..3] = even+odd, o[4..7] = even-odd for outputs 7..4 reversed */
    int z2 = in[2 * stride], z3 = in[6 * stride];
    int z1 = (z2 + z3) * FIX_0_541196100;
    int tmp2 = z1 + z3 * (-FIX_1_847759065);
    int tmp3 = z1 + z2 * FIX_0_765366865;
    z2 = in[0]; z3 = in[4 * stride];
    int tmp0 = (int)((unsigned)(z2 + z3) << 13);
    int tmp1 = (int)((unsigned)(z2 - z3) << 13);
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7 * stride]; tmp1 = in[5 * stride]; tmp2 = in[3 * stride]; tmp3 = in[stride];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * FIX_1_175875602;
    tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869; tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    o[0] = tmp10 + tmp3; o[7] = tmp10 - tmp3;
    o[1] = tmp11 + tmp2; o[6] = tmp11 - tmp2;
    o[2] = tmp12 + tmp1; o[5] = tmp12 - tmp1;
    o[3] = tmp13 + tmp0; o[4] = tmp13 - tmp0;
}

static void idct_islow(const int16_t *coef, const uint16_t *q, uint8_t *out, int pitch)
{
    int deq[64], ws[64], o[8];
    for (int i = 0; i < 64; ++i) deq[i] = coef[i] * q[i];
    for (int c = 0; c < 8; ++c) {
        idct1d(deq + c, 8, o);
        for (int r = 0; r < 8; ++r) ws[r * 8 + c] = DESCALE(o[r], 13 - 2);
    }
    for (int r = 0; r < 8; ++r) {
        idct1d(ws + r * 8, 1, o);
        for (int c = 0; c < 8; ++c) out[r * pitch + c] = range_limit(DESCALE(o[c], 13 + 2 + 3));
    }
}

static uint8_t clamp8(int x) { return (uint8_t)(x < 0 ? 0 : x > 255 ? 255 : x); }

int orc_jpeg_info(const uint8_t *d, size_t n, int *info)
{
    orc_jpeg j;
    const int rc = parse(d, n, &j);
    if (rc) return rc;
    info[0] = j.h; info[1] = j.w; info[2] = j.ncomp; info[3] = j.hmax; info[4] = j.vmax; info[5] = j.mcux * j.mcuy * j.bpm;
    return 0;
}

/* coef: [nblocks][64] int16 (decode order, natural positions, absolute DC), may be NULL; bgr: [h][w][3] (gray replicated) */
int orc_jpeg_decode(const uint8_t *d, size_t n, int16_t *coef_out, uint8_t *bgr)
{
    orc_jpeg j;
    int rc = parse(d, n, &j);
    if (rc) return rc;
    const size_t nblk = (size_t)j.mcux * j.mcuy * j.bpm;
    int16_t *coef = coef_out ? coef_out : (int16_t *)malloc(nblk * 64 * sizeof(int16_t));
    if (!coef) return -20;
    entropy(&j, coef);
    /* component planes, padded to whole MCUs */
    uint8_t *pl[3] = {0, 0, 0}; int pw[3], ph[3];
    for (int c = 0; c < j.ncomp; ++c) {
        pw[c] = j.mcux * j.hs[c] * 8; ph[c] = j.mcuy * j.vs[c] * 8;
        pl[c] = (uint8_t *)malloc((size_t)pw[c] * ph[c]);
    }
    const int16_t *blk = coef;
    for (int my = 0; my < j.mcuy; ++my)
        for (int mx = 0; mx < j.mcux; ++mx)
            for (int c = 0; c < j.ncomp; ++c)
                for (int by = 0; by < j.vs[c]; ++by)
                    for (int bx = 0; bx < j.hs[c]; ++bx, blk += 64)
                        idct_islow(blk, j.q[j.tq[c]], pl[c] + (size_t)((my * j.vs[c] + by) * 8) * pw[c] + (mx * j.hs[c] + bx) * 8, pw[c]);
    if (bgr) {
        if (j.ncomp == 1) {
            for (int y = 0; y < j.h; ++y)
                for (int x = 0; x < j.w; ++x) { const uint8_t v = pl[0][(size_t)y * pw[0] + x]; uint8_t *o = bgr + ((size_t)y * j.w + x) * 3; o[0] = o[1] = o[2] = v; }
        } else {
            const int cw = (j.w + j.hmax - 1) / j.hmax, ch = (j.h + j.vmax - 1) / j.vmax;     /* downsampled_width / height */
            uint8_t *up[2];
            for (int c = 1; c < 3; ++c) {
                up[c - 1] = (uint8_t *)malloc((size_t)(2 * cw + 2) * (2 * ch + 2));
                const int uw = 2 * cw + 2;
                const uint8_t *src = pl[c]; const int sp = pw[c];
                if (j.hmax == 1) {
                    for (int y = 0; y < j.h; ++y) memcpy(up[c - 1] + (size_t)y * uw, src + (size_t)y * sp, j.w);
                } else if (j.vmax == 1) {                      /* h2v1_fancy_upsample */
                    for (int y = 0; y < j.h; ++y) {
                        const uint8_t *in = src + (size_t)y * sp; uint8_t *o = up[c - 1] + (size_t)y * uw;
                        for (int x = 0; x < cw; ++x) {
                            const int v = in[x];
                            o[2 * x] = (uint8_t)(x == 0 ? v : (v * 3 + in[x - 1] + 1) >> 2);
                            o[2 * x + 1] = (uint8_t)(x == cw - 1 ? v : (v * 3 + in[x + 1] + 2) >> 2);
                        }
                    }
                } else {                                       /* h2v2_fancy_upsample; context rows replicated at top and bottom */
                    for (int y = 0; y < 2 * ch; ++y) {
                        const int r0 = y >> 1;
                        int r1 = (y & 1) ? r0 + 1 : r0 - 1;
                        if (r1 < 0) r1 = 0;
                        if (r1 > ch - 1) r1 = ch - 1;
                        const uint8_t *i0 = src + (size_t)r0 * sp, *i1 = src + (size_t)r1 * sp; uint8_t *o = up[c - 1] + (size_t)y * uw;
                        for (int x = 0; x < cw; ++x) {
                            const int t = i0[x] * 3 + i1[x];
                            const int l = x > 0 ? i0[x - 1] * 3 + i1[x - 1] : 0, nx = x < cw - 1 ? i0[x + 1] * 3 + i1[x + 1] : 0;
                            o[2 * x] = (uint8_t)(x == 0 ? (t * 4 + 8) >> 4 : (t * 3 + l + 8) >> 4);
                            o[2 * x + 1] = (uint8_t)(x == cw - 1 ? (t * 4 + 7) >> 4 : (t * 3 + nx + 7) >> 4);
                        }
                    }
                }
            }
            const int uw = 2 * cw + 2;
            for (int y = 0; y < j.h; ++y)
                for (int x = 0; x < j.w; ++x) {
                    const int Y = pl[0][(size_t)y * pw[0] + x], cb = up[0][(size_t)y * uw + x] - 128, cr = up[1][(size_t)y * uw + x] - 128;
                    /* jdcolor.c: SCALEBITS 16, FIX(x) = (int)(x * 65536 + 0.5), arithmetic right shifts */
                    const int r = Y + ((91881 * cr + 32768) >> 16);
                    const int g = Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
                    const int b = Y + ((116130 * cb + 32768) >> 16);
                    uint8_t *o = bgr + ((size_t)y * j.w + x) * 3;
                    o[0] = clamp8(b); o[1] = clamp8(g); o[2] = clamp8(r);
                }
            free(up[0]); free(up[1]);
        }
    }
    for (int c = 0; c < j.ncomp; ++c) free(pl[c]);
    if (!coef_out) free(coef);
    return 0;
}
