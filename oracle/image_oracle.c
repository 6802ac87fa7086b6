/*
 * oracle/image_oracle.c — CPU restatement of the image stages of the optical-flow hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call this file.  The product (libofk.so) never
 * links or falls back to it.
 *
 * PARITY UNPINNED for everything in this file: the reference performs these stages by
 * calling OpenCV (cv2.cvtColor / goodFeaturesToTrack / calcOpticalFlowPyrLK), which is
 * neither vendored, pinned nor installed here, and the reference holds no input->output
 * pair for them.  This file therefore DEFINES the semantics (SURVEY.md Appendix B,
 * restating OpenCV's published algorithms) and the HIP kernels are checked bit-exactly
 * against it.  Reference call sites each function stands for:
 *
 *   orc_gray_bgr8      cv2.cvtColor(.., COLOR_BGR2GRAY)   of_module.py:40,80; velocity_measurment_node:113;
 *                                                         evaluate_exp.py:65,85; of_library.py:236,248
 *   orc_mineig / orc_select_corners / orc_good_features
 *                      cv2.goodFeaturesToTrack            of_module.py:44,86; velocity_measurment_node:120,163;
 *                                                         evaluate_exp.py:66,106; of_library.py:238
 *   orc_pyr_down / orc_scharr / orc_lk_pyr
 *                      cv2.calcOpticalFlowPyrLK           of_module.py:88; velocity_measurment_node:133;
 *                                                         evaluate_exp.py:98; of_library.py:249
 *
 * Definitional choices where OpenCV's float summation order is unspecified: all window
 * sums (structure tensor, LK normal matrix, LK mismatch vector) are accumulated EXACTLY in
 * integers and converted to float once, so any reduction order gives the same bits.
 * Compile with -ffp-contract=off (no FMA contraction), default rounding mode.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

static inline int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        else i = 2 * (n - 1) - i;
    }
    return i;
}

/* ---- S1: BGR -> gray, fixed point (OpenCV 4 coefficients, 15-bit shift) ---- */
void orc_gray_bgr8(const uint8_t *bgr, int h, int w, uint8_t *gray)
{
    const long n = (long)h * w;
    for (long i = 0; i < n; ++i) {
        const int b = bgr[3 * i], g = bgr[3 * i + 1], r = bgr[3 * i + 2];
        gray[i] = (uint8_t)((b * 3735 + g * 19235 + r * 9798 + 16384) >> 15);
    }
}

/* ---- S3: pyrDown, separable [1 4 6 4 1], REFLECT_101, (sum+128)>>8 ---- */
void orc_pyr_down(const uint8_t *src, int h, int w, uint8_t *dst)
{
    const int dh = (h + 1) / 2, dw = (w + 1) / 2;
    static const int k[5] = {1, 4, 6, 4, 1};
    for (int y = 0; y < dh; ++y)
        for (int x = 0; x < dw; ++x) {
            int s = 0;
            for (int j = 0; j < 5; ++j) {
                const uint8_t *row = src + (long)reflect101(2 * y + j - 2, h) * w;
                int rs = 0;
                for (int i = 0; i < 5; ++i) rs += k[i] * row[reflect101(2 * x + i - 2, w)];
                s += k[j] * rs;
            }
            dst[(long)y * dw + x] = (uint8_t)((s + 128) >> 8);
        }
}

/* Number of pyramid levels actually used by LK: level l>=1 exists while its size is > win in both dims. */
int orc_lk_levels(int h, int w, int win, int max_level)
{
    int l = 0;
    while (l < max_level) {
        h = (h + 1) / 2; w = (w + 1) / 2;
        if (w <= win || h <= win) break;
        ++l;
    }
    return l;
}

/* ---- S3: Scharr derivatives, int16 interleaved (dx,dy), REFLECT_101 inside the image ---- */
void orc_scharr(const uint8_t *src, int h, int w, int16_t *dxdy)
{
    for (int y = 0; y < h; ++y) {
        const uint8_t *r0 = src + (long)reflect101(y - 1, h) * w;
        const uint8_t *r1 = src + (long)y * w;
        const uint8_t *r2 = src + (long)reflect101(y + 1, h) * w;
        for (int x = 0; x < w; ++x) {
            const int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
            const int dx = 3 * (r0[xr] - r0[xl]) + 10 * (r1[xr] - r1[xl]) + 3 * (r2[xr] - r2[xl]);
            const int dy = 3 * (r2[xl] - r0[xl]) + 10 * (r2[x] - r0[x]) + 3 * (r2[xr] - r0[xr]);
            dxdy[2 * ((long)y * w + x)] = (int16_t)dx;
            dxdy[2 * ((long)y * w + x) + 1] = (int16_t)dy;
        }
    }
}

/* ---- S2a: min-eigenvalue response (Sobel-3 -> products -> box(block) -> lambda_min), f32 map ---- */
int orc_mineig(const uint8_t *gray, int h, int w, int block, float *eig)
{
    if (block < 1 || block > 45) return -1;      /* int32 box sums are exact up to 45x45 */
    const long n = (long)h * w;
    int32_t *pxx = (int32_t *)malloc(n * 4), *pxy = (int32_t *)malloc(n * 4), *pyy = (int32_t *)malloc(n * 4);
    if (!pxx || !pxy || !pyy) { free(pxx); free(pxy); free(pyy); return -2; }
    for (int y = 0; y < h; ++y) {
        const uint8_t *r0 = gray + (long)reflect101(y - 1, h) * w;
        const uint8_t *r1 = gray + (long)y * w;
        const uint8_t *r2 = gray + (long)reflect101(y + 1, h) * w;
        for (int x = 0; x < w; ++x) {
            const int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
            const int dx = (r0[xr] - r0[xl]) + 2 * (r1[xr] - r1[xl]) + (r2[xr] - r2[xl]);
            const int dy = (r2[xl] - r0[xl]) + 2 * (r2[x] - r0[x]) + (r2[xr] - r0[xr]);
            pxx[(long)y * w + x] = dx * dx;
            pxy[(long)y * w + x] = dx * dy;
            pyy[(long)y * w + x] = dy * dy;
        }
    }
    const int an = block / 2;                    /* window [p-an, p-an+block-1], REFLECT_101 of the product image */
    const double scale = 1.0 / (4.0 * block * 255.0);
    const float kd = (float)(0.5 * scale * scale), ko = (float)(scale * scale);
    /* horizontal pass into temporaries, then vertical */
    int32_t *hxx = (int32_t *)malloc(n * 4), *hxy = (int32_t *)malloc(n * 4), *hyy = (int32_t *)malloc(n * 4);
    if (!hxx || !hxy || !hyy) { free(pxx); free(pxy); free(pyy); free(hxx); free(hxy); free(hyy); return -2; }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int32_t a = 0, b = 0, c = 0;
            for (int i = 0; i < block; ++i) {
                const long q = (long)y * w + reflect101(x - an + i, w);
                a += pxx[q]; b += pxy[q]; c += pyy[q];
            }
            hxx[(long)y * w + x] = a; hxy[(long)y * w + x] = b; hyy[(long)y * w + x] = c;
        }
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) {
            int32_t sa = 0, sb = 0, sc = 0;
            for (int j = 0; j < block; ++j) {
                const long q = (long)reflect101(y - an + j, h) * w + x;
                sa += hxx[q]; sb += hxy[q]; sc += hyy[q];
            }
            const float a = (float)sa * kd, b = (float)sb * ko, c = (float)sc * kd;
            const float amc = a - c;
            eig[(long)y * w + x] = (a + c) - sqrtf(amc * amc + b * b);
        }
    free(pxx); free(pxy); free(pyy); free(hxx); free(hxy); free(hyy);
    return 0;
}

typedef struct { float v; int32_t idx; } cand_t;
static int cand_cmp(const void *pa, const void *pb)
{
    const cand_t *a = (const cand_t *)pa, *b = (const cand_t *)pb;
    if (a->v > b->v) return -1;
    if (a->v < b->v) return 1;
    return (a->idx < b->idx) - (a->idx > b->idx);   /* tie: HIGHER linear index first - OpenCV's greaterThanPtr (featureselect.cpp) orders equal values by descending address */
}

/* ---- S2b: threshold vs q*max, 3x3 local max, sort, greedy min-distance, top-K ----
 * Returns the number of corners written to pts_xy (x,y float pairs), or <0 on error
 * (-3: more corners than cap).  n_cand_out (optional) receives the candidate count. */
int orc_select_corners(const float *eig, const uint8_t *mask, int h, int w, int max_corners, double quality,
                       double min_distance, float *pts_xy, int cap, int *n_cand_out)
{
    const long n = (long)h * w;
    float maxv = -INFINITY;
    for (long i = 0; i < n; ++i)
        if ((!mask || mask[i]) && eig[i] > maxv) maxv = eig[i];
    if (n_cand_out) *n_cand_out = 0;
    if (!(maxv > 0.f)) return 0;
    const float thr = (float)((double)maxv * quality);
    long nc = 0, capc = 1024;
    cand_t *c = (cand_t *)malloc(capc * sizeof(cand_t));
    for (int y = 1; y < h - 1; ++y)
        for (int x = 1; x < w - 1; ++x) {
            const long i = (long)y * w + x;
            const float v = eig[i];
            if (!(v > thr) || (mask && !mask[i])) continue;
            int ismax = 1;
            for (int dy = -1; dy <= 1 && ismax; ++dy)
                for (int dx = -1; dx <= 1; ++dx)
                    if (eig[i + (long)dy * w + dx] > v) { ismax = 0; break; }
            if (!ismax) continue;
            if (nc == capc) { capc *= 2; c = (cand_t *)realloc(c, capc * sizeof(cand_t)); }
            c[nc].v = v; c[nc].idx = (int32_t)i; ++nc;
        }
    if (n_cand_out) *n_cand_out = (int)nc;
    qsort(c, nc, sizeof(cand_t), cand_cmp);
    int nout = 0;
    const float md = (float)min_distance;
    const float md2 = md * md;
    for (long k = 0; k < nc; ++k) {
        const int x = c[k].idx % w, y = c[k].idx / w;
        int ok = 1;
        if (md >= 1.f)
            for (int j = 0; j < nout; ++j) {
                const int dx = x - (int)pts_xy[2 * j], dy = y - (int)pts_xy[2 * j + 1];
                if ((float)(dx * dx + dy * dy) < md2) { ok = 0; break; }
            }
        if (!ok) continue;
        if (nout >= cap) { free(c); return -3; }
        pts_xy[2 * nout] = (float)x; pts_xy[2 * nout + 1] = (float)y; ++nout;
        if (max_corners > 0 && nout == max_corners) break;
    }
    free(c);
    return nout;
}

int orc_good_features(const uint8_t *gray, const uint8_t *mask, int h, int w, int max_corners, double quality,
                      double min_distance, int block, float *pts_xy, int cap)
{
    float *eig = (float *)malloc((long)h * w * 4);
    if (!eig) return -2;
    int rc = orc_mineig(gray, h, w, block, eig);
    if (rc == 0) rc = orc_select_corners(eig, mask, h, w, max_corners, quality, min_distance, pts_xy, cap, 0);
    free(eig);
    return rc;
}

/* ---- S4: pyramidal Lucas-Kanade ---- */
static inline int descale(int v, int n) { return (v + (1 << (n - 1))) >> n; }
static inline int round_half_even(float v) { return (int)lrintf(v); }

typedef struct { const uint8_t *img; const int16_t *der; int h, w; } level_t;

static inline int pix(const level_t *L, int y, int x) { return L->img[(long)reflect101(y, L->h) * L->w + reflect101(x, L->w)]; }
static inline void der(const level_t *L, int y, int x, int *dx, int *dy)
{
    if (x < 0 || y < 0 || x >= L->w || y >= L->h) { *dx = 0; *dy = 0; return; }   /* constant-0 border */
    *dx = L->der[2 * ((long)y * L->w + x)]; *dy = L->der[2 * ((long)y * L->w + x) + 1];
}

static void weights(float a, float b, int iw[4])
{
    iw[0] = round_half_even((1.f - a) * (1.f - b) * 16384.f);
    iw[1] = round_half_even(a * (1.f - b) * 16384.f);
    iw[2] = round_half_even((1.f - a) * b * 16384.f);
    iw[3] = 16384 - iw[0] - iw[1] - iw[2];
}

/* Instrumentation (tools/experiments/probe_lk_divergence.py): when set, orc_lk_pyr stores the Newton steps point p spent on level l
 * at orc_lk_iter_stats[p * 9 + l] (the device kernel runs four points per wave until the last one has converged). */
int *orc_lk_iter_stats = 0;

int orc_lk_pyr(const uint8_t *prev, const uint8_t *next, int h, int w, const float *prev_pts, int n, int win,
               int max_level, int max_count, double eps, double min_eig_thr, float *next_pts, uint8_t *status,
               float *err)
{
    if (win < 3 || win > 31 || (win & 1) == 0 || max_level < 0 || max_level > 8) return -1;
    if (max_count < 0) max_count = 0;
    if (max_count > 100) max_count = 100;
    if (eps < 0) eps = 0;
    if (eps > 10) eps = 10;
    const double eps2 = eps * eps;
    const int L = orc_lk_levels(h, w, win, max_level);
    level_t P[9], Q[9];
    uint8_t *own_p[9] = {0}, *own_q[9] = {0};
    int16_t *own_d[9] = {0};
    P[0].img = prev; Q[0].img = next; P[0].h = Q[0].h = h; P[0].w = Q[0].w = w;
    for (int l = 1; l <= L; ++l) {
        const int ph = P[l - 1].h, pw = P[l - 1].w, nh = (ph + 1) / 2, nw = (pw + 1) / 2;
        own_p[l] = (uint8_t *)malloc((long)nh * nw); own_q[l] = (uint8_t *)malloc((long)nh * nw);
        orc_pyr_down(P[l - 1].img, ph, pw, own_p[l]); orc_pyr_down(Q[l - 1].img, ph, pw, own_q[l]);
        P[l].img = own_p[l]; Q[l].img = own_q[l]; P[l].h = Q[l].h = nh; P[l].w = Q[l].w = nw;
    }
    for (int l = 0; l <= L; ++l) {
        own_d[l] = (int16_t *)malloc((long)P[l].h * P[l].w * 4);
        orc_scharr(P[l].img, P[l].h, P[l].w, own_d[l]);
        P[l].der = own_d[l]; Q[l].der = 0;
    }
    const float half = (float)(win - 1) * 0.5f;
    const int ww = win * win;
    int16_t *Ip = (int16_t *)malloc(ww * 2), *Ixp = (int16_t *)malloc(ww * 2), *Iyp = (int16_t *)malloc(ww * 2);

    for (int p = 0; p < n; ++p) {
        status[p] = 1;
        if (err) err[p] = 0.f;
        float nx = 0.f, ny = 0.f;      /* nextPts[p], carried between levels */
        for (int l = L; l >= 0; --l) {
            const level_t *I = &P[l], *J = &Q[l];
            const float sc = (float)(1.0 / (double)(1 << l));
            float px = prev_pts[2 * p] * sc, py = prev_pts[2 * p + 1] * sc;
            float qx, qy;
            if (l == L) { qx = px; qy = py; } else { qx = nx * 2.f; qy = ny * 2.f; }
            nx = qx; ny = qy;
            px -= half; py -= half;
            const int ipx = (int)floorf(px), ipy = (int)floorf(py);
            if (ipx < -win || ipx >= I->w || ipy < -win || ipy >= I->h) {
                if (l == 0) { status[p] = 0; if (err) err[p] = 0.f; }
                continue;
            }
            int iw[4];
            weights(px - (float)ipx, py - (float)ipy, iw);
            int64_t sA11 = 0, sA12 = 0, sA22 = 0;
            for (int y = 0; y < win; ++y)
                for (int x = 0; x < win; ++x) {
                    const int Y = ipy + y, X = ipx + x;
                    const int iv = descale(pix(I, Y, X) * iw[0] + pix(I, Y, X + 1) * iw[1] + pix(I, Y + 1, X) * iw[2] +
                                               pix(I, Y + 1, X + 1) * iw[3], 14 - 5);
                    int d00x, d00y, d01x, d01y, d10x, d10y, d11x, d11y;
                    der(I, Y, X, &d00x, &d00y); der(I, Y, X + 1, &d01x, &d01y);
                    der(I, Y + 1, X, &d10x, &d10y); der(I, Y + 1, X + 1, &d11x, &d11y);
                    const int ix = descale(d00x * iw[0] + d01x * iw[1] + d10x * iw[2] + d11x * iw[3], 14);
                    const int iy = descale(d00y * iw[0] + d01y * iw[1] + d10y * iw[2] + d11y * iw[3], 14);
                    Ip[y * win + x] = (int16_t)iv; Ixp[y * win + x] = (int16_t)ix; Iyp[y * win + x] = (int16_t)iy;
                    sA11 += (int64_t)ix * ix; sA12 += (int64_t)ix * iy; sA22 += (int64_t)iy * iy;
                }
            const float A11 = (float)((double)sA11 * 0x1p-20), A12 = (float)((double)sA12 * 0x1p-20),
                        A22 = (float)((double)sA22 * 0x1p-20);
            float D = A11 * A22 - A12 * A12;
            const float dd = A11 - A22;
            const float minEig = (A22 + A11 - sqrtf(dd * dd + 4.f * A12 * A12)) / (float)(2 * ww);
            if ((double)minEig < min_eig_thr || D < FLT_EPSILON) {
                if (l == 0) status[p] = 0;
                continue;
            }
            D = 1.f / D;
            qx -= half; qy -= half;
            float pdx = 0.f, pdy = 0.f;
            for (int j = 0; j < max_count; ++j) {
                const int iqx = (int)floorf(qx), iqy = (int)floorf(qy);
                if (iqx < -win || iqx >= J->w || iqy < -win || iqy >= J->h) {
                    if (l == 0) status[p] = 0;
                    break;
                }
                if (orc_lk_iter_stats) orc_lk_iter_stats[p * 9 + l] = j + 1;
                weights(qx - (float)iqx, qy - (float)iqy, iw);
                int64_t sb1 = 0, sb2 = 0;
                for (int y = 0; y < win; ++y)
                    for (int x = 0; x < win; ++x) {
                        const int Y = iqy + y, X = iqx + x;
                        const int diff = descale(pix(J, Y, X) * iw[0] + pix(J, Y, X + 1) * iw[1] + pix(J, Y + 1, X) * iw[2] +
                                                     pix(J, Y + 1, X + 1) * iw[3], 14 - 5) - Ip[y * win + x];
                        sb1 += (int64_t)diff * Ixp[y * win + x]; sb2 += (int64_t)diff * Iyp[y * win + x];
                    }
                const float b1 = (float)((double)sb1 * 0x1p-20), b2 = (float)((double)sb2 * 0x1p-20);
                const float dx = (A12 * b2 - A22 * b1) * D, dy = (A12 * b1 - A11 * b2) * D;
                qx += dx; qy += dy;
                nx = qx + half; ny = qy + half;
                if ((double)dx * dx + (double)dy * dy <= eps2) break;
                if (j > 0 && fabs((double)(dx + pdx)) < 0.01 && fabs((double)(dy + pdy)) < 0.01) {
                    nx -= dx * 0.5f; ny -= dy * 0.5f;
                    break;
                }
                pdx = dx; pdy = dy;
            }
            if (status[p] && err && l == 0) {
                const float ex = nx - half, ey = ny - half;
                const int iex = (int)floorf(ex), iey = (int)floorf(ey);
                if (iex < -win || iex >= J->w || iey < -win || iey >= J->h) { status[p] = 0; continue; }
                weights(ex - (float)iex, ey - (float)iey, iw);
                int64_t se = 0;
                for (int y = 0; y < win; ++y)
                    for (int x = 0; x < win; ++x) {
                        const int Y = iey + y, X = iex + x;
                        const int diff = descale(pix(J, Y, X) * iw[0] + pix(J, Y, X + 1) * iw[1] + pix(J, Y + 1, X) * iw[2] +
                                                     pix(J, Y + 1, X + 1) * iw[3], 14 - 5) - Ip[y * win + x];
                        se += diff < 0 ? -diff : diff;
                    }
                err[p] = (float)se / (float)(32 * ww);
            }
        }
        next_pts[2 * p] = nx; next_pts[2 * p + 1] = ny;
    }
    free(Ip); free(Ixp); free(Iyp);
    for (int l = 0; l <= L; ++l) { free(own_p[l]); free(own_q[l]); free(own_d[l]); }
    return 0;
}
