// ofk_api.hip — the C ABI of libofk.so (include/ofk.h): context lifecycle, host<->HBM plumbing, stage entry
// points and the resident frame-pair pipeline.  No CPU fallback anywhere: every entry point launches HIP
// kernels on the context's stream or returns an error.
#include "ofk_internal.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static char g_create_err[512] = "";

int ofk_fail(ofk_ctx *ctx, int code, const char *fmt, ...)
{
    char *dst = ctx ? ctx->errmsg : g_create_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    return code;
}

#define TRY(expr) do { int rc_ = (expr); if (rc_ != OFK_OK) return rc_; } while (0)
static size_t up(size_t v, size_t a) { return (v + a - 1) / a * a; }

ofk_levels ofk_make_levels(int h, int w, int win, int max_level)
{
    ofk_levels lv;
    memset(&lv, 0, sizeof lv);
    lv.h[0] = h; lv.w[0] = w; lv.off[0] = 0;
    size_t off = up((size_t)h * w, 256);
    int l = 0;
    while (l < max_level && l < OFK_MAX_LEVELS - 1) {
        const int nh = (lv.h[l] + 1) / 2, nw = (lv.w[l] + 1) / 2;
        if (win > 0 && (nw <= win || nh <= win)) break;
        ++l;
        lv.h[l] = nh; lv.w[l] = nw; lv.off[l] = off;
        off += up((size_t)nh * nw, 256);
    }
    lv.n = l;
    return lv;
}

static size_t levels_bytes(int h, int w, int max_level)
{
    const ofk_levels lv = ofk_make_levels(h, w, 0, max_level);
    return lv.off[lv.n] + up((size_t)lv.h[lv.n] * lv.w[lv.n], 256);
}

// Hardware queues: the free-running slices keep four HIP streams busy (six to eight with RCCL's) and the runtime multiplexes
// streams onto GPU_MAX_HW_QUEUES hardware queues, 4 by default; two streams of one queue run in order (-10 % measured).  The
// variable is read when the HIP runtime initialises, so it is the HOST's to set before its first HIP call (the Python binding and
// bench.py do; INTEGRATION.md tells C callers): a library that edits its host's environment from a load-time constructor races with
// getenv on other threads and changes every other HIP user of the process (rounds 1-2 did that).  ofk_set_streams warns when the
// value in effect is too small.

ofk_tuning g_ofk_tuning = {0, 0, 0, 0, 0, 0, 0};

static int *tuning_slot(const char *knob, int *lo, int *hi)
{
    struct { const char *name; int *slot; int lo, hi; } tab[] = {
        {"eig_rows", &g_ofk_tuning.eig_rows, 8, 4096},        // rows per strip of the streaming response kernels
        {"no_pair", &g_ofk_tuning.no_pair, 0, 1},             // 1: one column per lane everywhere (k_mineig_stream instead of k_mineig_pair)
        {"no_pyr3", &g_ofk_tuning.no_pyr3, 0, 1},             // 1: pyramid level by level instead of the three-level pass
        {"pyr3_chunks", &g_ofk_tuning.pyr3_chunks, 1, 4096},  // row chunks per strip of k_pyr3_stream
        {"pyr_rows", &g_ofk_tuning.pyr_rows, 1, 4096},        // rows per strip of k_pyr_down_stream
        {"jpeg_chunk", &g_ofk_tuning.jpeg_chunk, 64, 1024},   // bytes of entropy data per decoder thread (a power of two)
        {"jpeg_sub", &g_ofk_tuning.jpeg_sub, 1, 13},          // second-level Huffman look-up tables per image + 1 (1: none - every long code takes the canonical search)
        {"gray_px", &g_ofk_tuning.gray_px, 16, 64},           // experiment: one-wave workgroups of 16 / 32 / 64 pixels per thread in the BGR -> gray conversion
    };
    for (auto &t : tab)
        if (knob && strcmp(knob, t.name) == 0) { *lo = t.lo; *hi = t.hi; return t.slot; }
    return nullptr;
}

extern "C" int ofk_set_tuning(const char *knob, int value)
{
    int lo, hi;
    int *slot = tuning_slot(knob, &lo, &hi);
    if (!slot) return ofk_fail(nullptr, OFK_E_INVALID, "ofk_set_tuning: unknown knob '%s'", knob ? knob : "(null)");
    const bool jpeg_ok = slot != &g_ofk_tuning.jpeg_chunk || (value & (value - 1)) == 0;
    if (value != 0 && (value < lo || value > hi || !jpeg_ok)) return ofk_fail(nullptr, OFK_E_INVALID, "ofk_set_tuning: %s takes 0 (default) or %d..%d", knob, lo, hi);
    *slot = value;
    return OFK_OK;
}

extern "C" int ofk_get_tuning(const char *knob, int *value)
{
    int lo, hi;
    int *slot = tuning_slot(knob, &lo, &hi);
    if (!slot || !value) return ofk_fail(nullptr, OFK_E_INVALID, "ofk_get_tuning: unknown knob '%s'", knob ? knob : "(null)");
    *value = *slot;
    return OFK_OK;
}

extern "C" int ofk_version(void) { return OFK_VERSION; }

extern "C" const char *ofk_last_error(const ofk_ctx *ctx) { return ctx ? ctx->errmsg : g_create_err; }

extern "C" int ofk_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int ofk_device_sync(void) { return hipDeviceSynchronize() == hipSuccess ? OFK_OK : OFK_E_HIP; }

#define ALLOC(ptr, bytes)                                                                                     \
    do {                                                                                                      \
        hipError_t e_ = hipMalloc((void **)&(ptr), (bytes));                                                  \
        if (e_ != hipSuccess) {                                                                               \
            ofk_fail(nullptr, OFK_E_HIP, "hipMalloc(%zu bytes) for %s: %s", (size_t)(bytes), #ptr, hipGetErrorString(e_)); \
            ofk_destroy(c);                                                                                   \
            return OFK_E_HIP;                                                                                 \
        }                                                                                                     \
    } while (0)

extern "C" int ofk_create(int device, int max_w, int max_h, int max_batch, int max_pts, int max_level, ofk_ctx **out)
{
    if (!out) return ofk_fail(nullptr, OFK_E_INVALID, "out is NULL");
    *out = nullptr;
    if (max_w < 16 || max_h < 16 || max_w > 16384 || max_h > 16384 || max_batch < 1 || max_pts < 1 || max_pts > 4096 ||
        max_level < 0 || max_level > 8)
        return ofk_fail(nullptr, OFK_E_INVALID, "ofk_create: bad limits (w %d h %d batch %d pts %d level %d)", max_w, max_h,
                        max_batch, max_pts, max_level);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return ofk_fail(nullptr, OFK_E_NOGPU, "no HIP device visible");
    if (device < 0 || device >= ndev) return ofk_fail(nullptr, OFK_E_INVALID, "device %d out of range (%d visible)", device, ndev);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ofk_fail(nullptr, OFK_E_HIP, "hipGetDeviceProperties failed");
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return ofk_fail(nullptr, OFK_E_NOGPU, "device %d is %s; libofk.so is built for gfx950 only", device, prop.gcnArchName);
    if (hipSetDevice(device) != hipSuccess) return ofk_fail(nullptr, OFK_E_HIP, "hipSetDevice(%d) failed", device);

    ofk_ctx *c = (ofk_ctx *)calloc(1, sizeof(ofk_ctx));
    if (!c) return ofk_fail(nullptr, OFK_E_INVALID, "out of host memory");
    c->device = device; c->max_w = max_w; c->max_h = max_h; c->max_batch = max_batch; c->max_pts = max_pts; c->max_level = max_level;
    c->P = (size_t)max_w * max_h;
    c->bgr_stride = up(c->P * 3, 256);
    c->pyr_stride = levels_bytes(max_h, max_w, max_level);
    c->img_stride = up(c->P, 64);
    c->cand_cap = (int)(c->P / 4 < 4096 ? 4096 : c->P / 4);
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { free(c); return ofk_fail(nullptr, OFK_E_HIP, "hipStreamCreate failed"); }
    c->nstreams = 1;
    c->streams[0] = c->stream;
    c->overlap = 1;
    c->gray_direct_set = -1;
    // Slice and auxiliary streams are created when a call first needs them (need_streams): the runtime multiplexes HIP streams
    // onto a few hardware queues (4 by default), and two streams of one queue run in order - an idle stream would cost a real one
    // its concurrency.
    bool ok = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&c->ev_x, hipEventDisableTiming) == hipSuccess;
    for (int k = 0; k < OFK_MAX_STREAMS && ok; ++k)
        ok = hipEventCreateWithFlags(&c->ev_g0[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_aux[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_lkdone[0][k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_lkdone[1][k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_end[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c->ev_stagger[k], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        ofk_fail(nullptr, OFK_E_HIP, "hipEventCreate failed");
        ofk_destroy(c);
        return OFK_E_HIP;
    }
    const size_t B = (size_t)max_batch;
    for (int k = 0; k < 2; ++k) { ALLOC(c->bgr[k], B * c->bgr_stride); ALLOC(c->pyr[k], B * c->pyr_stride); }
    ALLOC(c->eig, B * c->img_stride * sizeof(float));
    ALLOC(c->cand, B * (size_t)c->cand_cap * 8);
    c->seg_keys = (size_t)(c->P / 4) + (size_t)(c->P / 4) / 2 + 64 * OFK_SEG_MAX;   // segment rounding slack
    ALLOC(c->cand_seg, B * c->seg_keys * 8);
    ALLOC(c->seg_count, B * OFK_SEG_MAX * 4);
    ALLOC(c->cand_count, B * OFK_CNT_STRIDE * 4); ALLOC(c->maxbits, B * OFK_MAX_STRIDE * 4);
    ALLOC(c->sel_hist, B * 1024 * 4);
    ALLOC(c->sel_keys, B * OFK_CHUNK * 8);
    ALLOC(c->pts_prev, B * max_pts * 8); ALLOC(c->pts_next, B * max_pts * 8);
    ALLOC(c->status, B * max_pts); ALLOC(c->err, B * max_pts * 4); ALLOC(c->counts, B * 4);
    ALLOC(c->sensors, B * OFK_SENSOR_DOUBLES * 8); ALLOC(c->records, B * OFK_RECORD_DOUBLES * 8);
    ALLOC(c->dev_flags, 16);
    hipMemsetAsync(c->dev_flags, 0, 16, c->stream);
    hipMemsetAsync(c->counts, 0, B * 4, c->stream);
    hipMemsetAsync(c->sensors, 0, B * OFK_SENSOR_DOUBLES * 8, c->stream);
    c->ev_cap = 32768;                                           // stage timers between two ofk_profile_read calls (28 per step with two slices);
                                                                 // the events themselves are created on first use
    c->ev = (hipEvent_t *)calloc(c->ev_cap, sizeof(hipEvent_t));
    c->ev_stage = (int *)calloc(c->ev_cap / 2, sizeof(int));
    hipStreamSynchronize(c->stream);
    *out = c;
    return OFK_OK;
}

extern "C" int ofk_destroy(ofk_ctx *c)
{
    if (!c) return OFK_OK;
    hipSetDevice(c->device);
    if (c->stream) hipDeviceSynchronize();
    ofk_comm_destroy(c);
    for (int k = 0; k < 2; ++k) { if (c->bgr[k]) hipFree(c->bgr[k]); if (c->pyr[k]) hipFree(c->pyr[k]); }
    void *ptrs[] = {c->eig, c->mask, c->deriv, c->cand, c->cand_seg, c->seg_count, c->cand_count, c->sel_hist, c->sel_keys, c->maxbits, c->pts_prev, c->pts_next, c->status, c->err,
                    c->counts, c->sensors, c->records, c->dev_flags, c->scratch, c->pts_new, c->new_counts, c->limit,
                    c->imu_state, c->imu_dv, c->kf_mats, c->kf_x, c->kf_P, c->fused, c->imu_msgs, c->imu_counts};
    for (void *p : ptrs) if (p) hipFree(p);
    if (c->hstage) hipHostFree(c->hstage);
    ofk_jpeg_release(c);
    if (c->ev) { for (int i = 0; i < c->ev_cap; ++i) if (c->ev[i]) hipEventDestroy(c->ev[i]); free(c->ev); }
    free(c->ev_stage);
    free(c->h_counts);
    for (int k = 1; k < OFK_MAX_STREAMS; ++k) if (c->streams[k]) hipStreamDestroy(c->streams[k]);
    for (int k = 0; k < OFK_MAX_STREAMS; ++k) {
        if (c->aux[k]) hipStreamDestroy(c->aux[k]);
        if (c->ev_g0[k]) hipEventDestroy(c->ev_g0[k]);
        if (c->ev_aux[k]) hipEventDestroy(c->ev_aux[k]);
        for (int s = 0; s < 2; ++s) if (c->ev_lkdone[s][k]) hipEventDestroy(c->ev_lkdone[s][k]);
    }
    for (int k = 0; k < 2; ++k) if (c->pyr_alt[k]) hipFree(c->pyr_alt[k]);
    for (int k = 0; k < 8; ++k) if (c->marks[k]) hipEventDestroy(c->marks[k]);
    for (int k = 0; k < OFK_MAX_STREAMS; ++k) {
        if (c->ev_end[k]) hipEventDestroy(c->ev_end[k]);
        if (c->ev_stagger[k]) hipEventDestroy(c->ev_stagger[k]);
    }
    if (c->ev_x) hipEventDestroy(c->ev_x);
    if (c->ev_fork) hipEventDestroy(c->ev_fork);
    if (c->stream) hipStreamDestroy(c->stream);
    free(c);
    return OFK_OK;
}

// Slice streams free-run across consecutive ofk_pairs_run calls.  Every other entry point works on the context's stream and
// on buffers the slices use, so it first makes that stream wait for whatever the slices (and the export stream) still hold.
static int need_streams(ofk_ctx *c, int slices, bool overlap)
{
    for (int k = 1; k < slices; ++k)
        if (!c->streams[k]) OFK_HIP(c, hipStreamCreateWithFlags(&c->streams[k], hipStreamNonBlocking));
    for (int k = 0; k < slices && overlap; ++k)
        if (!c->aux[k]) OFK_HIP(c, hipStreamCreateWithFlags(&c->aux[k], hipStreamNonBlocking));
    return OFK_OK;
}

// The last slice's stream once every other slice of the latest call has finished: where record export and marks are queued while
// the slices are open.  The slices are offset in time and the last one finishes last, so the waits cost it nothing; a stream of
// their own would have to share a hardware queue with one of the slices.
static int tail_stream(ofk_ctx *c, hipStream_t *out)
{
    *out = c->stream;
    if (!c->slices_open) return OFK_OK;
    hipStream_t s = c->streams[c->open_slices - 1];
    for (int k = 0; k + 1 < c->open_slices; ++k) OFK_HIP(c, hipStreamWaitEvent(s, c->ev_end[k], 0));
    *out = s;
    return OFK_OK;
}

static int join_slices(ofk_ctx *c)
{
    if (c->slices_open) {
        for (int k = 1; k < c->open_slices; ++k) OFK_HIP(c, hipStreamWaitEvent(c->stream, c->ev_end[k], 0));
        c->slices_open = 0;
    }
    if (c->x_pending) {
        OFK_HIP(c, hipStreamWaitEvent(c->stream, c->ev_x, 0));
        c->x_pending = 0;
    }
    return OFK_OK;
}

int ofk_join_slices(ofk_ctx *c) { return join_slices(c); }
int ofk_prepare_streams(ofk_ctx *c) { return need_streams(c, c->nstreams < 1 ? 1 : c->nstreams, c->overlap != 0); }

extern "C" int ofk_sync(ofk_ctx *c)
{
    if (!c) return OFK_E_INVALID;
    TRY(join_slices(c));
    OFK_HIP(c, hipStreamSynchronize(c->stream));
    return OFK_OK;
}

// every stream of the context idle (slice, auxiliary and context stream)
static int drain_all(ofk_ctx *c)
{
    OFK_HIP(c, hipSetDevice(c->device));
    TRY(join_slices(c));
    OFK_HIP(c, hipStreamSynchronize(c->stream));
    for (int k = 0; k < OFK_MAX_STREAMS; ++k) {
        if (k && c->streams[k]) OFK_HIP(c, hipStreamSynchronize(c->streams[k]));
        if (c->aux[k]) OFK_HIP(c, hipStreamSynchronize(c->aux[k]));
    }
    return OFK_OK;
}

int ofk_resident_pyramid(ofk_ctx *c, int frame_set, int image, uint8_t *out, size_t bytes)
{
    if (!c || !out || frame_set < 0 || frame_set > 1 || image < 0 || image >= c->max_batch || bytes > c->pyr_stride)
        return ofk_fail(c, OFK_E_INVALID, "resident pyramid: frame set 0/1, image < max_batch, at most one slab (%zu bytes)", c ? c->pyr_stride : (size_t)0);
    TRY(drain_all(c));
    const uint8_t *base = (c->pyr_last && c->pyr_alt[frame_set]) ? c->pyr_alt[frame_set] : c->pyr[frame_set];
    OFK_HIP(c, hipMemcpy(out, base + (size_t)image * c->pyr_stride, bytes, hipMemcpyDeviceToHost));
    return OFK_OK;
}

int ofk_need_scratch(ofk_ctx *c, size_t bytes)
{
    if (bytes <= c->scratch_bytes) return OFK_OK;
    if (c->scratch) { hipStreamSynchronize(c->stream); hipFree(c->scratch); c->scratch = nullptr; c->scratch_bytes = 0; }
    bytes = up(bytes, 1 << 20);
    OFK_HIP(c, hipMalloc(&c->scratch, bytes));
    c->scratch_bytes = bytes;
    return OFK_OK;
}

static int check_geom(ofk_ctx *c, int batch, int h, int w, const char *who)
{
    if (!c) return OFK_E_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) return ofk_fail(c, OFK_E_HIP, "hipSetDevice failed");
    if (batch < 1 || batch > c->max_batch || h < 1 || w < 1 || (size_t)h * w > c->P || h > 16384 || w > 16384)
        return ofk_fail(c, OFK_E_INVALID, "%s: batch %d / %dx%d exceeds the context (batch %d, %zu px)", who, batch, w, h, c->max_batch, c->P);
    TRY(join_slices(c));
    return OFK_OK;
}

// host [batch][bytes_per] (tight) <-> device base + b*stride
static int h2d(ofk_ctx *c, void *dev, size_t stride, const void *host, size_t bytes_per, int batch)
{
    OFK_HIP(c, hipMemcpy2DAsync(dev, stride, host, bytes_per, bytes_per, batch, hipMemcpyHostToDevice, c->stream));
    return OFK_OK;
}
static int d2h(ofk_ctx *c, void *host, const void *dev, size_t stride, size_t bytes_per, int batch)
{
    OFK_HIP(c, hipMemcpy2DAsync(host, bytes_per, dev, stride, bytes_per, batch, hipMemcpyDeviceToHost, c->stream));
    OFK_HIP(c, hipStreamSynchronize(c->stream));
    return OFK_OK;
}
static int check_launch(ofk_ctx *c, const char *who)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return ofk_fail(c, OFK_E_HIP, "%s: kernel launch failed: %s", who, hipGetErrorString(e));
    return OFK_OK;
}

static int lazy_mask(ofk_ctx *c)
{
    if (!c->mask) OFK_HIP(c, hipMalloc((void **)&c->mask, (size_t)c->max_batch * c->img_stride));
    return OFK_OK;
}

// ------------------------------------------------------------------------------------------------ stage entry points
extern "C" int ofk_gray_bgr8(ofk_ctx *c, const uint8_t *bgr, int batch, int h, int w, uint8_t *gray)
{
    TRY(check_geom(c, batch, h, w, "ofk_gray_bgr8"));
    if (!bgr || !gray) return ofk_fail(c, OFK_E_INVALID, "ofk_gray_bgr8: NULL buffer");
    const size_t px = (size_t)h * w;
    TRY(h2d(c, c->bgr[0], c->bgr_stride, bgr, px * 3, batch));
    ofk_launch_gray(c->stream, c->bgr[0], c->bgr_stride, c->pyr[0], c->pyr_stride, batch, h, w);
    TRY(check_launch(c, "k_gray_bgr8"));
    return d2h(c, gray, c->pyr[0], c->pyr_stride, px, batch);
}

extern "C" int ofk_pyr_down_u8(ofk_ctx *c, const uint8_t *src, int batch, int h, int w, uint8_t *dst)
{
    TRY(check_geom(c, batch, h, w, "ofk_pyr_down_u8"));
    if (!src || !dst) return ofk_fail(c, OFK_E_INVALID, "ofk_pyr_down_u8: NULL buffer");
    if (h < 2 || w < 2) return ofk_fail(c, OFK_E_INVALID, "ofk_pyr_down_u8: image smaller than 2x2");
    const size_t px = (size_t)h * w, dpx = (size_t)((h + 1) / 2) * ((w + 1) / 2);
    TRY(h2d(c, c->pyr[0], c->pyr_stride, src, px, batch));
    ofk_launch_pyr_down(c->stream, c->pyr[0], c->pyr_stride, h, w, c->pyr[1], c->pyr_stride, batch);
    TRY(check_launch(c, "k_pyr_down"));
    return d2h(c, dst, c->pyr[1], c->pyr_stride, dpx, batch);
}

extern "C" int ofk_scharr_s16(ofk_ctx *c, const uint8_t *gray, int batch, int h, int w, int16_t *dxdy)
{
    TRY(check_geom(c, batch, h, w, "ofk_scharr_s16"));
    if (!gray || !dxdy) return ofk_fail(c, OFK_E_INVALID, "ofk_scharr_s16: NULL buffer");
    if (h < 2 || w < 2) return ofk_fail(c, OFK_E_INVALID, "ofk_scharr_s16: image smaller than 2x2");
    if (!c->deriv) OFK_HIP(c, hipMalloc((void **)&c->deriv, (size_t)c->max_batch * c->img_stride * 4));
    const size_t px = (size_t)h * w;
    TRY(h2d(c, c->pyr[0], c->pyr_stride, gray, px, batch));
    ofk_launch_scharr(c->stream, c->pyr[0], c->pyr_stride, h, w, c->deriv, c->img_stride, batch);
    TRY(check_launch(c, "k_scharr"));
    return d2h(c, dxdy, c->deriv, c->img_stride * 4, px * 4, batch);
}

static int check_block(ofk_ctx *c, int h, int w, int block)
{
    if (block < 1 || block > 45) return ofk_fail(c, OFK_E_INVALID, "block_size %d outside 1..45", block);
    if (h < block + 4 || w < block + 4) return ofk_fail(c, OFK_E_INVALID, "image %dx%d too small for block_size %d", w, h, block);
    return OFK_OK;
}

extern "C" int ofk_mineig_response(ofk_ctx *c, const uint8_t *gray, int batch, int h, int w, int block, float *eig)
{
    TRY(check_geom(c, batch, h, w, "ofk_mineig_response"));
    if (!gray || !eig) return ofk_fail(c, OFK_E_INVALID, "ofk_mineig_response: NULL buffer");
    TRY(check_block(c, h, w, block));
    const size_t px = (size_t)h * w;
    TRY(h2d(c, c->pyr[0], c->pyr_stride, gray, px, batch));
    if (ofk_launch_mineig(c->stream, c->pyr[0], c->pyr_stride, h, w, block, c->eig, c->img_stride, nullptr, nullptr, 0, batch))
        return ofk_fail(c, OFK_E_INVALID, "k_mineig: LDS tile too large for block_size %d", block);
    TRY(check_launch(c, "k_mineig"));
    return d2h(c, eig, c->eig, c->img_stride * 4, px * 4, batch);
}

static int check_select(ofk_ctx *c, int max_corners, double quality, double min_distance)
{
    if (max_corners < 1 || max_corners > c->max_pts) return ofk_fail(c, OFK_E_INVALID, "max_corners %d outside 1..%d", max_corners, c->max_pts);
    if (!(quality > 0.0) || !(min_distance >= 0.0)) return ofk_fail(c, OFK_E_INVALID, "quality must be > 0 and min_distance >= 0");
    return OFK_OK;
}

// device-side: eig (+mask) resident in ctx -> corners in pts_prev / counts
static int run_select(ofk_ctx *c, bool have_max, const uint8_t *dmask, int batch, int h, int w, int max_corners, double quality,
                      double min_distance)
{
    if (!have_max) {
        OFK_HIP(c, hipMemsetAsync(c->maxbits, 0, (size_t)batch * OFK_MAX_STRIDE * 4, c->stream));
        ofk_launch_maxbits(c->stream, c->eig, c->img_stride, dmask, c->img_stride, h, w, c->maxbits, batch);
    }
    OFK_HIP(c, hipMemsetAsync(c->cand_count, 0, (size_t)batch * OFK_CNT_STRIDE * 4, c->stream));
    ofk_launch_nms(c->stream, c->eig, c->img_stride, dmask, c->img_stride, h, w, c->maxbits, quality, c->cand, c->cand_cap,
                   c->cand_count, c->dev_flags, batch);
    ofk_launch_select(c->stream, c->cand, c->cand_cap, c->cand_count, nullptr, 0, nullptr, 0, c->maxbits, quality, h, w, max_corners,
                      (float)min_distance, c->pts_prev, c->max_pts, c->counts, nullptr, batch, c->sel_hist, c->sel_keys);
    return check_launch(c, "corner selection");
}

static int fetch_corners(ofk_ctx *c, int batch, int max_corners, float *pts, int *counts)
{
    int flags[4];
    OFK_HIP(c, hipMemcpyAsync(flags, c->dev_flags, 16, hipMemcpyDeviceToHost, c->stream));
    OFK_HIP(c, hipMemcpyAsync(counts, c->counts, (size_t)batch * 4, hipMemcpyDeviceToHost, c->stream));
    TRY(d2h(c, pts, c->pts_prev, (size_t)c->max_pts * 8, (size_t)max_corners * 8, batch));
    bool over = (flags[0] & 1) != 0;
    for (int b = 0; b < batch; ++b) if (counts[b] < 0) { over = true; counts[b] = 0; }
    if (over) {
        hipMemsetAsync(c->dev_flags, 0, 16, c->stream);
        return ofk_fail(c, OFK_E_CAPACITY, "corner candidates exceeded the per-image capacity (%d)", c->cand_cap);
    }
    return OFK_OK;
}

extern "C" int ofk_select_corners(ofk_ctx *c, const float *eig, const uint8_t *mask, int batch, int h, int w, int max_corners,
                                  double quality, double min_distance, float *pts, int *counts)
{
    TRY(check_geom(c, batch, h, w, "ofk_select_corners"));
    if (!eig || !pts || !counts) return ofk_fail(c, OFK_E_INVALID, "ofk_select_corners: NULL buffer");
    if (h < 3 || w < 3) return ofk_fail(c, OFK_E_INVALID, "ofk_select_corners: image smaller than 3x3");
    TRY(check_select(c, max_corners, quality, min_distance));
    const size_t px = (size_t)h * w;
    TRY(h2d(c, c->eig, c->img_stride * 4, eig, px * 4, batch));
    const uint8_t *dmask = nullptr;
    if (mask) { TRY(lazy_mask(c)); TRY(h2d(c, c->mask, c->img_stride, mask, px, batch)); dmask = c->mask; }
    TRY(run_select(c, false, dmask, batch, h, w, max_corners, quality, min_distance));
    return fetch_corners(c, batch, max_corners, pts, counts);
}

extern "C" int ofk_good_features(ofk_ctx *c, const uint8_t *gray, const uint8_t *mask, int batch, int h, int w, int max_corners,
                                 double quality, double min_distance, int block, float *pts, int *counts)
{
    TRY(check_geom(c, batch, h, w, "ofk_good_features"));
    if (!gray || !pts || !counts) return ofk_fail(c, OFK_E_INVALID, "ofk_good_features: NULL buffer");
    TRY(check_block(c, h, w, block));
    TRY(check_select(c, max_corners, quality, min_distance));
    const size_t px = (size_t)h * w;
    TRY(h2d(c, c->pyr[0], c->pyr_stride, gray, px, batch));
    const uint8_t *dmask = nullptr;
    if (mask) { TRY(lazy_mask(c)); TRY(h2d(c, c->mask, c->img_stride, mask, px, batch)); dmask = c->mask; }
    OFK_HIP(c, hipMemsetAsync(c->maxbits, 0, (size_t)batch * OFK_MAX_STRIDE * 4, c->stream));
    OFK_HIP(c, hipMemsetAsync(c->cand_count, 0, (size_t)batch * OFK_CNT_STRIDE * 4, c->stream));
    int nseg = 0, segcap = 0;
    if (ofk_launch_mineig_cand(c->stream, c->pyr[0], c->pyr_stride, h, w, block, c->maxbits, dmask, c->img_stride, quality, c->cand,
                               c->cand_cap, c->cand_count, c->cand_seg, c->seg_keys, c->seg_count, OFK_SEG_MAX, c->dev_flags, batch,
                               &nseg, &segcap))
        return ofk_fail(c, OFK_E_INVALID, "corner response: block_size %d does not fit (LDS tile / key segments)", block);
    ofk_launch_select(c->stream, c->cand, c->cand_cap, c->cand_count, c->cand_seg, segcap, c->seg_count, nseg, c->maxbits, quality, h, w,
                      max_corners, (float)min_distance, c->pts_prev, c->max_pts, c->counts, nullptr, batch, c->sel_hist, c->sel_keys);
    TRY(check_launch(c, "corner detection"));
    return fetch_corners(c, batch, max_corners, pts, counts);
}

static int check_lk(ofk_ctx *c, int h, int w, int win, int max_level)
{
    if (win < 3 || win > 31 || (win & 1) == 0) return ofk_fail(c, OFK_E_INVALID, "win %d must be odd and in 3..31", win);
    if (max_level < 0 || max_level > c->max_level) return ofk_fail(c, OFK_E_INVALID, "max_level %d outside 0..%d", max_level, c->max_level);
    if (h <= win || w <= win) return ofk_fail(c, OFK_E_INVALID, "image %dx%d not larger than the LK window %d", w, h, win);
    return OFK_OK;
}

// builds levels 1..L of both resident pyramids
static void build_pyramids(ofk_ctx *c, const ofk_levels &lv, int batch, int which_mask)
{
    int l0 = 1;                                                  // levels 1..3 in one pass where the geometry allows it
    if (which_mask == 3 ? ofk_launch_pyr3(c->stream, c->pyr[0], c->pyr[1], c->pyr_stride, lv, batch, 2 * batch)
                        : ofk_launch_pyr3(c->stream, c->pyr[which_mask == 1 ? 0 : 1], nullptr, c->pyr_stride, lv, batch, batch)) l0 = 4;
    for (int l = l0; l <= lv.n; ++l) {
        if (which_mask == 3)
            ofk_launch_pyr_down2(c->stream, c->pyr[0] + lv.off[l - 1], c->pyr[1] + lv.off[l - 1], c->pyr_stride, lv.h[l - 1], lv.w[l - 1],
                                 c->pyr[0] + lv.off[l], c->pyr[1] + lv.off[l], c->pyr_stride, batch);
        else {
            const int k = which_mask == 1 ? 0 : 1;
            ofk_launch_pyr_down(c->stream, c->pyr[k] + lv.off[l - 1], c->pyr_stride, lv.h[l - 1], lv.w[l - 1], c->pyr[k] + lv.off[l],
                                c->pyr_stride, batch);
        }
    }
}

extern "C" int ofk_lk_pyr(ofk_ctx *c, const uint8_t *prev, const uint8_t *next, int batch, int h, int w, const float *prev_pts,
                          const int *counts, int pts_stride, int win, int max_level, int max_count, double eps, double min_eig_thr,
                          float *next_pts, uint8_t *status, float *err)
{
    TRY(check_geom(c, batch, h, w, "ofk_lk_pyr"));
    if (!prev || !next || !prev_pts || !counts || !next_pts || !status || !err) return ofk_fail(c, OFK_E_INVALID, "ofk_lk_pyr: NULL buffer");
    TRY(check_lk(c, h, w, win, max_level));
    if (pts_stride < 1 || pts_stride > c->max_pts) return ofk_fail(c, OFK_E_INVALID, "pts_stride %d outside 1..%d", pts_stride, c->max_pts);
    for (int b = 0; b < batch; ++b)
        if (counts[b] < 0 || counts[b] > pts_stride) return ofk_fail(c, OFK_E_INVALID, "counts[%d]=%d outside 0..%d", b, counts[b], pts_stride);
    const size_t px = (size_t)h * w;
    const ofk_levels lv = ofk_make_levels(h, w, win, max_level);
    TRY(h2d(c, c->pyr[0], c->pyr_stride, prev, px, batch));
    TRY(h2d(c, c->pyr[1], c->pyr_stride, next, px, batch));
    TRY(h2d(c, c->pts_prev, (size_t)c->max_pts * 8, prev_pts, (size_t)pts_stride * 8, batch));
    OFK_HIP(c, hipMemcpyAsync(c->counts, counts, (size_t)batch * 4, hipMemcpyHostToDevice, c->stream));
    build_pyramids(c, lv, batch, 3);
    ofk_launch_lk(c->stream, c->pyr[0], c->pyr[1], c->pyr_stride, lv, c->pts_prev, c->counts, c->max_pts, win, max_count, eps,
                  min_eig_thr, c->pts_next, c->status, c->err, batch);
    TRY(check_launch(c, "k_lk"));
    OFK_HIP(c, hipMemcpy2DAsync(next_pts, (size_t)pts_stride * 8, c->pts_next, (size_t)c->max_pts * 8, (size_t)pts_stride * 8, batch,
                                hipMemcpyDeviceToHost, c->stream));
    OFK_HIP(c, hipMemcpy2DAsync(status, pts_stride, c->status, c->max_pts, pts_stride, batch, hipMemcpyDeviceToHost, c->stream));
    return d2h(c, err, c->err, (size_t)c->max_pts * 4, (size_t)pts_stride * 4, batch);
}

// levels 1..L of a batch of gray images, built exactly as ofk_lk_pyr / ofk_pairs_run build them (three levels per pass where the
// geometry allows it); out = per image the levels 1..L back to back, tightly packed
extern "C" int ofk_pyramid_u8(ofk_ctx *c, const uint8_t *gray, int batch, int h, int w, int max_level, uint8_t *out, int *levels_built)
{
    TRY(check_geom(c, batch, h, w, "ofk_pyramid_u8"));
    if (!gray || !out) return ofk_fail(c, OFK_E_INVALID, "ofk_pyramid_u8: NULL buffer");
    if (max_level < 0 || max_level > c->max_level) return ofk_fail(c, OFK_E_INVALID, "max_level %d outside 0..%d", max_level, c->max_level);
    const ofk_levels lv = ofk_make_levels(h, w, 0, max_level);
    TRY(h2d(c, c->pyr[0], c->pyr_stride, gray, (size_t)h * w, batch));
    build_pyramids(c, lv, batch, 1);
    TRY(check_launch(c, "pyramid"));
    size_t total = 0, o = 0;
    for (int l = 1; l <= lv.n; ++l) total += (size_t)lv.h[l] * lv.w[l];
    for (int l = 1; l <= lv.n; ++l) {
        const size_t px = (size_t)lv.h[l] * lv.w[l];
        OFK_HIP(c, hipMemcpy2DAsync(out + o, total, c->pyr[0] + lv.off[l], c->pyr_stride, px, batch, hipMemcpyDeviceToHost, c->stream));
        o += px;
    }
    OFK_HIP(c, hipStreamSynchronize(c->stream));
    if (levels_built) *levels_built = lv.n;
    return OFK_OK;
}

// ------------------------------------------------------------------------------------------------ estimation entry points
struct Bump {                                               // carves device scratch and uploads host arrays
    ofk_ctx *c; char *base; size_t off; int rc;
    double *put(const void *host, size_t bytes)
    {
        if (!host) return nullptr;
        char *p = base + off;
        off += up(bytes, 256);
        if (rc == OFK_OK && hipMemcpyAsync(p, host, bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) rc = OFK_E_HIP;
        return (double *)p;
    }
    double *take(size_t bytes) { char *p = base + off; off += up(bytes, 256); return (double *)p; }
};
static int get(ofk_ctx *c, void *host, const void *dev, size_t bytes)
{
    OFK_HIP(c, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    OFK_HIP(c, hipStreamSynchronize(c->stream));
    return OFK_OK;
}
static int est_begin(ofk_ctx *c, size_t bytes, Bump &bp)
{
    if (!c) return OFK_E_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) return ofk_fail(c, OFK_E_HIP, "hipSetDevice failed");
    TRY(join_slices(c));
    TRY(ofk_need_scratch(c, bytes + 64 * 256));
    bp.c = c; bp.base = (char *)c->scratch; bp.off = 0; bp.rc = OFK_OK;
    return OFK_OK;
}

extern "C" int ofk_flow_model(ofk_ctx *c, const double *x, int batch, int n, const double *v, const double *omega, const double *d,
                              const double *nrm, const double *t, double *flow)
{
    if (!c || !x || !v || !omega || !d || !nrm || !flow || batch < 1 || n < 1) return ofk_fail(c, OFK_E_INVALID, "ofk_flow_model: bad argument");
    const size_t pb = (size_t)batch * n * 16;
    Bump bp;
    TRY(est_begin(c, 2 * pb + (size_t)batch * 256 * 5, bp));
    double *dx = bp.put(x, pb), *dv = bp.put(v, batch * 24), *dom = bp.put(omega, batch * 24), *dd = bp.put(d, batch * 8),
           *dn = bp.put(nrm, batch * 24), *dt = bp.put(t, batch * 24), *df = bp.take(pb);
    if (bp.rc) return ofk_fail(c, OFK_E_HIP, "ofk_flow_model: upload failed");
    ofk_launch_flow_model(c->stream, dx, batch, n, dv, dom, dd, dn, dt, df);
    TRY(check_launch(c, "k_flow_model"));
    return get(c, flow, df, pb);
}

extern "C" int ofk_feasibility(ofk_ctx *c, int variant, const double *x, const double *u, int batch, int n, const double *nrm,
                               const double *v, const double *dist, const double *omega, const double *t, double *r, double *dd)
{
    if (!c || !x || !u || !nrm || !v || !r || !dd || batch < 1 || n < 1 || variant < 0 || variant > 2)
        return ofk_fail(c, OFK_E_INVALID, "ofk_feasibility: bad argument");
    if (variant == OFK_FEAS_RTILDE && !dist) return ofk_fail(c, OFK_E_INVALID, "ofk_feasibility: dist required");
    if (variant == OFK_FEAS_SIM && (!omega || !t)) return ofk_fail(c, OFK_E_INVALID, "ofk_feasibility: omega and t required");
    const size_t pb = (size_t)batch * n * 16, rb = (size_t)batch * n * 8;
    Bump bp;
    TRY(est_begin(c, 2 * pb + 2 * rb + (size_t)batch * 256 * 5, bp));
    double *dx = bp.put(x, pb), *du = bp.put(u, pb), *dn = bp.put(nrm, batch * 24), *dv = bp.put(v, batch * 24),
           *ddist = bp.put(dist, batch * 8), *dom = bp.put(omega, batch * 24), *dt = bp.put(t, batch * 24), *dr = bp.take(rb),
           *ddd = bp.take(rb);
    if (bp.rc) return ofk_fail(c, OFK_E_HIP, "ofk_feasibility: upload failed");
    ofk_launch_feasibility(c->stream, variant, dx, du, batch, n, dn, dv, ddist, dom, dt, dr, ddd);
    TRY(check_launch(c, "k_feasibility"));
    TRY(get(c, r, dr, rb));
    return get(c, dd, ddd, rb);
}

extern "C" int ofk_velocity_solve(ofk_ctx *c, int variant, const double *x, const double *u, const uint8_t *valid, int batch, int n,
                                  const double *d, const double *nrm, const double *omega, const double *t, const double *wgt,
                                  double *out)
{
    if (!c || !x || !u || !nrm || !out || batch < 1 || n < 1 || variant < 0 || variant > 2)
        return ofk_fail(c, OFK_E_INVALID, "ofk_velocity_solve: bad argument");
    if (variant == OFK_SOLVE_OFMODULE ? !wgt : (!d || !omega)) return ofk_fail(c, OFK_E_INVALID, "ofk_velocity_solve: missing input for variant %d", variant);
    const size_t pb = (size_t)batch * n * 16;
    Bump bp;
    TRY(est_begin(c, 2 * pb + (size_t)batch * n * 9 + (size_t)batch * 256 * 6, bp));
    double *dx = bp.put(x, pb), *du = bp.put(u, pb);
    uint8_t *dval = (uint8_t *)bp.put(valid, (size_t)batch * n);
    double *dd = bp.put(d, batch * 8), *dn = bp.put(nrm, batch * 24), *dom = bp.put(omega, batch * 24), *dt = bp.put(t, batch * 24),
           *dw = bp.put(wgt, (size_t)batch * n * 8), *dout = bp.take((size_t)batch * OFK_SOLVE_DOUBLES * 8);
    if (bp.rc) return ofk_fail(c, OFK_E_HIP, "ofk_velocity_solve: upload failed");
    ofk_launch_solve(c->stream, variant, dx, du, dval, batch, n, dd, dn, dom, dt, dw, dout);
    TRY(check_launch(c, "k_solve"));
    return get(c, out, dout, (size_t)batch * OFK_SOLVE_DOUBLES * 8);
}

extern "C" int ofk_imu_propagate(ofk_ctx *c, double *state, const double *msg, int batch)
{
    if (!c || !state || !msg || batch < 1) return ofk_fail(c, OFK_E_INVALID, "ofk_imu_propagate: bad argument");
    Bump bp;
    TRY(est_begin(c, (size_t)batch * (OFK_IMU_STATE + OFK_IMU_MSG) * 8, bp));
    double *ds = bp.put(state, (size_t)batch * OFK_IMU_STATE * 8), *dm = bp.put(msg, (size_t)batch * OFK_IMU_MSG * 8);
    if (bp.rc) return ofk_fail(c, OFK_E_HIP, "ofk_imu_propagate: upload failed");
    ofk_launch_imu(c->stream, ds, dm, batch);
    TRY(check_launch(c, "k_imu"));
    return get(c, state, ds, (size_t)batch * OFK_IMU_STATE * 8);
}

extern "C" int ofk_post_solve(ofk_ctx *c, const double *v_obs, const double *rotation, const double *ang, const double *offset,
                              int batch, double *v_uav)
{
    if (!c || !v_obs || !rotation || !ang || !offset || !v_uav || batch < 1) return ofk_fail(c, OFK_E_INVALID, "ofk_post_solve: bad argument");
    Bump bp;
    TRY(est_begin(c, (size_t)batch * 24 * 8, bp));
    double *dv = bp.put(v_obs, batch * 24), *dr = bp.put(rotation, batch * 72), *da = bp.put(ang, batch * 24),
           *dof = bp.put(offset, batch * 24), *dout = bp.take(batch * 24);
    if (bp.rc) return ofk_fail(c, OFK_E_HIP, "ofk_post_solve: upload failed");
    ofk_launch_post_solve(c->stream, dv, dr, da, dof, batch, dout);
    TRY(check_launch(c, "k_post_solve"));
    return get(c, v_uav, dout, batch * 24);
}

extern "C" int ofk_associate_sensors(ofk_ctx *c, const double *t_img, int n_img, const double *imu_t, const double *imu_quat,
                                     const double *imu_omega, int n_imu, const double *hgt_t, const double *hgt_range, int n_hgt,
                                     double *sensors, int *imu_index, int *hgt_index)
{
    if (!c || !t_img || !imu_t || !imu_quat || !imu_omega || !hgt_t || !hgt_range || !sensors || n_img < 1 || n_imu < 1 || n_hgt < 1)
        return ofk_fail(c, OFK_E_INVALID, "ofk_associate_sensors: bad argument (every log needs at least one sample)");
    Bump bp;
    TRY(est_begin(c, ((size_t)n_img * (1 + OFK_SENSOR_DOUBLES + 1) + (size_t)n_imu * 8 + (size_t)n_hgt * 2) * 8, bp));
    double *dt = bp.put(t_img, (size_t)n_img * 8), *dit = bp.put(imu_t, (size_t)n_imu * 8), *diq = bp.put(imu_quat, (size_t)n_imu * 32),
           *diw = bp.put(imu_omega, (size_t)n_imu * 24), *dht = bp.put(hgt_t, (size_t)n_hgt * 8), *dhr = bp.put(hgt_range, (size_t)n_hgt * 8),
           *ds = bp.put(sensors, (size_t)n_img * OFK_SENSOR_DOUBLES * 8);
    int *dii = (int *)bp.take((size_t)n_img * 4), *dhi = (int *)bp.take((size_t)n_img * 4);
    if (bp.rc) return ofk_fail(c, OFK_E_HIP, "ofk_associate_sensors: upload failed");
    ofk_launch_associate(c->stream, dt, n_img, n_imu, dit, diq, diw, n_hgt, dht, dhr, ds, dii, dhi);
    TRY(check_launch(c, "k_associate"));
    if (imu_index) TRY(get(c, imu_index, dii, (size_t)n_img * 4));
    if (hgt_index) TRY(get(c, hgt_index, dhi, (size_t)n_img * 4));
    return get(c, sensors, ds, (size_t)n_img * OFK_SENSOR_DOUBLES * 8);
}

extern "C" int ofk_d_split(ofk_ctx *c, const double *d, const int *counts, int batch, int stride, double d_exp_err, double *sorted,
                           double *diff, int *nsplit)
{
    if (!c || !d || !counts || !sorted || !diff || !nsplit || batch < 1 || stride < 1 || stride > 4096)
        return ofk_fail(c, OFK_E_INVALID, "ofk_d_split: bad argument (stride 1..4096)");
    const size_t n = (size_t)batch * stride;
    Bump bp;
    TRY(est_begin(c, n * 24 + (size_t)batch * 8 + 1024, bp));
    double *dd = bp.put(d, n * 8);
    int *dcn = (int *)bp.put(counts, (size_t)batch * 4);
    double *ds = bp.take(n * 8), *dg = bp.take(n * 8);
    int *dn = (int *)bp.take((size_t)batch * 4);
    if (bp.rc) return ofk_fail(c, OFK_E_HIP, "ofk_d_split: upload failed");
    OFK_HIP(c, hipMemsetAsync(ds, 0, n * 8, c->stream)); OFK_HIP(c, hipMemsetAsync(dg, 0, n * 8, c->stream));
    ofk_launch_d_split(c->stream, dd, dcn, batch, stride, d_exp_err, ds, dg, dn);
    TRY(check_launch(c, "k_d_split"));
    TRY(get(c, sorted, ds, n * 8)); TRY(get(c, diff, dg, n * 8));
    return get(c, nsplit, dn, (size_t)batch * 4);
}

extern "C" int ofk_feature_eval(ofk_ctx *c, const double *pos, const double *pos_err, const double *oldpos, const double *oldpos_err,
                                const int *counts, int batch, int stride, const double *vel, const double *vel_err, double focal_len,
                                double dummy_value, int img_w, int img_h, const double *weight, double *height, double *height_err,
                                uint8_t *immobile, double *score, int *order, int *bad_height)
{
    if (!c || !pos || !pos_err || !oldpos || !oldpos_err || !counts || !vel || !vel_err || !weight || !height || !height_err || !immobile ||
        !score || !order || batch < 1 || stride < 1)
        return ofk_fail(c, OFK_E_INVALID, "ofk_feature_eval: bad argument");
    const size_t n = (size_t)batch * stride;
    Bump bp;
    TRY(est_begin(c, n * (2 + 1 + 2 + 1 + 1 + 1 + 1 + 1) * 8 + (size_t)batch * 64 + 1024, bp));
    double *dp = bp.put(pos, n * 16), *dpe = bp.put(pos_err, n * 8), *dop = bp.put(oldpos, n * 16), *doe = bp.put(oldpos_err, n * 8);
    int *dcn = (int *)bp.put(counts, (size_t)batch * 4);
    double *dv = bp.put(vel, (size_t)batch * 24), *dve = bp.put(vel_err, (size_t)batch * 24), *dw = bp.put(weight, 32);
    double *dh = bp.take(n * 8), *dhe = bp.take(n * 8), *dsc = bp.take(n * 8);
    uint8_t *dim = (uint8_t *)bp.take(n);
    int *dord = (int *)bp.take(n * 4), *dfl = (int *)bp.take(16);
    if (bp.rc) return ofk_fail(c, OFK_E_HIP, "ofk_feature_eval: upload failed");
    OFK_HIP(c, hipMemsetAsync(dfl, 0, 16, c->stream));
    OFK_HIP(c, hipMemsetAsync(dim, 0, n, c->stream));
    OFK_HIP(c, hipMemsetAsync(dh, 0, n * 8, c->stream)); OFK_HIP(c, hipMemsetAsync(dhe, 0, n * 8, c->stream));
    OFK_HIP(c, hipMemsetAsync(dsc, 0, n * 8, c->stream));
    OFK_HIP(c, hipMemsetAsync(dord, 0xff, n * 4, c->stream));                   // slots past a set's count read -1
    // of.pix_trans (of_library.py:31-43): d/2 if even else (d+1)/2
    const double tx = (img_w % 2 == 0) ? img_w / 2.0 : (img_w + 1) / 2.0, ty = (img_h % 2 == 0) ? img_h / 2.0 : (img_h + 1) / 2.0;
    ofk_launch_feature_eval(c->stream, dp, dpe, dop, doe, dcn, batch, stride, dv, dve, focal_len, dummy_value, tx, ty, dw, dh, dhe, dim, dsc,
                            dord, dfl);
    TRY(check_launch(c, "k_feature_eval"));
    int fl[4] = {0, 0, 0, 0};
    TRY(get(c, fl, dfl, 16));
    if (bad_height) *bad_height = fl[1];
    TRY(get(c, height, dh, n * 8)); TRY(get(c, height_err, dhe, n * 8)); TRY(get(c, score, dsc, n * 8));
    TRY(get(c, immobile, dim, n));
    return get(c, order, dord, n * 4);
}

extern "C" int ofk_kf_predict_update(ofk_ctx *c, int ns, int nm, int nc, const double *F, const double *Bm, const double *H,
                                     const double *Q, const double *Rm, double *x, double *P, const double *u, const double *z,
                                     int batch, int do_predict)
{
    if (!c || ns < 1 || ns > 6 || nm < 1 || nm > 6 || nc < 0 || nc > 6 || !F || !H || !Q || !Rm || !x || !P || batch < 1)
        return ofk_fail(c, OFK_E_INVALID, "ofk_kf_predict_update: bad argument");
    Bump bp;
    TRY(est_begin(c, (size_t)batch * (ns + ns * ns + nc + nm) * 8 + 8 * 256 * 8, bp));
    double *dF = bp.put(F, ns * ns * 8), *dB = (Bm && nc) ? bp.put(Bm, ns * nc * 8) : nullptr, *dH = bp.put(H, nm * ns * 8),
           *dQ = bp.put(Q, ns * ns * 8), *dR = bp.put(Rm, nm * nm * 8), *dx = bp.put(x, (size_t)batch * ns * 8),
           *dP = bp.put(P, (size_t)batch * ns * ns * 8), *du = (u && nc) ? bp.put(u, (size_t)batch * nc * 8) : nullptr,
           *dz = bp.put(z, (size_t)batch * nm * 8);
    if (bp.rc) return ofk_fail(c, OFK_E_HIP, "ofk_kf_predict_update: upload failed");
    ofk_launch_kf(c->stream, ns, nm, nc, dF, dB, dH, dQ, dR, dx, dP, du, dz, batch, do_predict);
    TRY(check_launch(c, "k_kf"));
    TRY(get(c, x, dx, (size_t)batch * ns * 8));
    return get(c, P, dP, (size_t)batch * ns * ns * 8);
}

extern "C" int ofk_of_simulation(ofk_ctx *c, const double *truth, const double *sig, const double *pos, const double *true_flow,
                                 int n, const double *z, int trials, double *v_obs, double *bound)
{
    if (!c || !truth || !sig || !pos || !true_flow || !z || !v_obs || !bound || n < 1 || trials < 1)
        return ofk_fail(c, OFK_E_INVALID, "ofk_of_simulation: bad argument");
    const size_t zb = (size_t)trials * (10 + 4 * (size_t)n) * 8;
    Bump bp;
    TRY(est_begin(c, zb + (size_t)n * 32 + (size_t)trials * 32 + 8 * 256, bp));
    double *dt = bp.put(truth, 13 * 8), *ds = bp.put(sig, 6 * 8), *dp = bp.put(pos, (size_t)n * 16), *df = bp.put(true_flow, (size_t)n * 16),
           *dz = bp.put(z, zb), *dv = bp.take((size_t)trials * 24), *db = bp.take((size_t)trials * 8);
    if (bp.rc) return ofk_fail(c, OFK_E_HIP, "ofk_of_simulation: upload failed");
    ofk_launch_of_simulation(c->stream, dt, ds, dp, df, n, dz, trials, dv, db);
    TRY(check_launch(c, "k_of_simulation"));
    TRY(get(c, v_obs, dv, (size_t)trials * 24));
    return get(c, bound, db, (size_t)trials * 8);
}

// of_simulation with the normals drawn on the device (counter-based: k_estimate.hip ofk_noise_normal); nothing but 13 + 6 + 4 n doubles
// goes up and 4 doubles per trial come back
extern "C" int ofk_of_simulation_rng(ofk_ctx *c, const double *truth, const double *sig, const double *pos, const double *true_flow, int n,
                                     unsigned long long seed, unsigned step, unsigned trial0, int trials, double *v_obs, double *bound)
{
    if (!c || !truth || !sig || !pos || !true_flow || !v_obs || !bound || n < 1 || trials < 1 || (unsigned long long)n * 4 + 10 > 0x7fffffffull)
        return ofk_fail(c, OFK_E_INVALID, "ofk_of_simulation_rng: bad argument");
    Bump bp;
    TRY(est_begin(c, (size_t)n * 32 + (size_t)trials * 32 + 8 * 256, bp));
    double *dt = bp.put(truth, 13 * 8), *ds = bp.put(sig, 6 * 8), *dp = bp.put(pos, (size_t)n * 16), *df = bp.put(true_flow, (size_t)n * 16),
           *dv = bp.take((size_t)trials * 24), *db = bp.take((size_t)trials * 8);
    if (bp.rc) return ofk_fail(c, OFK_E_HIP, "ofk_of_simulation_rng: upload failed");
    ofk_launch_of_simulation_rng(c->stream, dt, ds, dp, df, n, seed, step, trial0, trials, dv, db);
    TRY(check_launch(c, "k_of_simulation<rng>"));
    TRY(get(c, v_obs, dv, (size_t)trials * 24));
    return get(c, bound, db, (size_t)trials * 8);
}

// elements 0 .. count - 1 of the noise row of (seed, step, trial): the generator itself, for tests and for callers that want the draws
extern "C" int ofk_noise_normals(ofk_ctx *c, unsigned long long seed, unsigned step, unsigned trial, int count, double *out)
{
    if (!c || !out || count < 1) return ofk_fail(c, OFK_E_INVALID, "ofk_noise_normals: bad argument");
    Bump bp;
    TRY(est_begin(c, (size_t)count * 8 + 256, bp));
    double *d = bp.take((size_t)count * 8);
    if (bp.rc) return ofk_fail(c, OFK_E_HIP, "ofk_noise_normals: scratch");
    ofk_launch_noise_normals(c->stream, (unsigned)seed, (unsigned)(seed >> 32), step, trial, count, d);
    TRY(check_launch(c, "k_noise_normals"));
    return get(c, out, d, (size_t)count * 8);
}

extern "C" int ofk_feas_simulation(ofk_ctx *c, const double *truth, const double *sig, const double *pos, const double *true_flow,
                                   int n, const double *z, int trials, double *mean, double *per_trial, double *v_obs)
{
    if (!c || !truth || !sig || !pos || !true_flow || !z || !mean || n < 1 || trials < 1)
        return ofk_fail(c, OFK_E_INVALID, "ofk_feas_simulation: bad argument");
    const size_t zb = (size_t)trials * (12 + 4 * (size_t)n) * 8, pb = (size_t)trials * 6 * n * 8;
    Bump bp;
    TRY(est_begin(c, zb + pb + (size_t)n * 32 + (size_t)n * 48 + (size_t)trials * 24 + 8 * 256, bp));
    double *dt = bp.put(truth, OFK_FEAS_SIM_TRUTH * 8), *ds = bp.put(sig, OFK_FEAS_SIM_SIG * 8), *dp = bp.put(pos, (size_t)n * 16),
           *df = bp.put(true_flow, (size_t)n * 16), *dz = bp.put(z, zb), *dper = bp.take(pb), *dmean = bp.take((size_t)n * 48),
           *dv = bp.take((size_t)trials * 24);
    if (bp.rc) return ofk_fail(c, OFK_E_HIP, "ofk_feas_simulation: upload failed");
    ofk_launch_feas_simulation(c->stream, dt, ds, dp, df, n, dz, trials, dper, dmean, dv);
    TRY(check_launch(c, "k_feas_simulation"));
    if (per_trial) TRY(get(c, per_trial, dper, pb));
    if (v_obs) TRY(get(c, v_obs, dv, (size_t)trials * 24));
    return get(c, mean, dmean, (size_t)n * 48);
}

extern "C" int ofk_hist_overlap(ofk_ctx *c, const double *data1, int n1, const double *data2, int n2, int bins, int *overlap)
{
    if (!c || !data1 || !data2 || !overlap || n1 < 1 || n2 < 1 || bins < 1 || bins > 1024)
        return ofk_fail(c, OFK_E_INVALID, "ofk_hist_overlap: bad argument (bins 1..1024, both samples non-empty)");
    Bump bp;
    TRY(est_begin(c, ((size_t)n1 + n2) * 8 + 1024, bp));
    double *d1 = bp.put(data1, (size_t)n1 * 8), *d2 = bp.put(data2, (size_t)n2 * 8);
    int *dout = (int *)bp.take(16);
    if (bp.rc) return ofk_fail(c, OFK_E_HIP, "ofk_hist_overlap: upload failed");
    ofk_launch_hist_overlap(c->stream, d1, n1, d2, n2, bins, dout);
    TRY(check_launch(c, "k_hist_overlap"));
    return get(c, overlap, dout, 4);
}

// ------------------------------------------------------------------------------------------------ resident pipeline
extern "C" int ofk_pairs_upload(ofk_ctx *c, const uint8_t *prev_bgr, const uint8_t *next_bgr, int batch, int h, int w)
{
    TRY(check_geom(c, batch, h, w, "ofk_pairs_upload"));
    if (!prev_bgr || !next_bgr) return ofk_fail(c, OFK_E_INVALID, "ofk_pairs_upload: NULL buffer");
    const size_t px = (size_t)h * w;
    TRY(h2d(c, c->bgr[0], c->bgr_stride, prev_bgr, px * 3, batch));
    TRY(h2d(c, c->bgr[1], c->bgr_stride, next_bgr, px * 3, batch));
    OFK_HIP(c, hipStreamSynchronize(c->stream));
    c->cur_batch = batch; c->cur_h = h; c->cur_w = w;
    c->gray_direct_set = -1;
    return OFK_OK;
}

// Compressed ingest: the frames arrive as baseline JPEG streams (sensor_msgs/CompressedImage payloads) and are decoded on the
// device straight into the resident BGR buffers (k_jpeg.hip) - 15-20x fewer bytes over PCIe than ofk_pairs_upload.
extern "C" int ofk_pairs_upload_jpeg(ofk_ctx *c, const uint8_t *const *prev_jpeg, const size_t *prev_bytes, const uint8_t *const *next_jpeg,
                                     const size_t *next_bytes, int batch)
{
    if (!c) return OFK_E_INVALID;
    if (!prev_jpeg || !prev_bytes || !next_jpeg || !next_bytes) return ofk_fail(c, OFK_E_INVALID, "ofk_pairs_upload_jpeg: NULL argument");
    if (batch < 1 || batch > c->max_batch) return ofk_fail(c, OFK_E_INVALID, "ofk_pairs_upload_jpeg: batch %d exceeds the context (%d)", batch, c->max_batch);
    // both frames of every pair in ONE decoder batch (previous frames first): half the launches and host round trips of two batches
    const uint8_t **all = (const uint8_t **)malloc(sizeof(void *) * 2 * (size_t)batch);
    size_t *len = (size_t *)malloc(sizeof(size_t) * 2 * (size_t)batch);
    if (!all || !len) { free(all); free(len); return ofk_fail(c, OFK_E_INVALID, "out of host memory"); }
    for (int b = 0; b < batch; ++b) { all[b] = prev_jpeg[b]; len[b] = prev_bytes[b]; all[batch + b] = next_jpeg[b]; len[batch + b] = next_bytes[b]; }
    int rc = ofk_jpeg_stage(c, 0, all, len, 2 * batch);
    free(all); free(len);
    if (rc != OFK_OK) return ofk_fail(c, rc, "%s", ofk_jpeg_slot_error(c, 0));      // synchronous caller = the owner thread: its message
    return ofk_pairs_upload_staged(c, 0);
}

// Phase 1 of the compressed ingest, callable from a second thread while the context's owner decodes the other slot (include/ofk.h).
extern "C" int ofk_jpeg_stage(ofk_ctx *c, int slot, const uint8_t *const *jpeg, const size_t *nbytes, int count)
{
    if (!c) return OFK_E_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) return OFK_E_HIP;
    return ofk_jpeg_stage_streams(c, slot, jpeg, nbytes, count);
}

// Message of the slot's last ofk_jpeg_stage ("" after a success).  ofk_jpeg_stage does not touch ofk_last_error: it may run on a helper
// thread while the owner thread is inside another entry point.
extern "C" const char *ofk_jpeg_stage_error(const ofk_ctx *c, int slot) { return ofk_jpeg_slot_error(c, slot); }

// Phase 2: the 2 B streams staged in `slot` (B previous frames, then B next frames) decoded into the resident frame-pair buffers.
extern "C" int ofk_pairs_upload_staged(ofk_ctx *c, int slot)
{
    if (!c) return OFK_E_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) return ofk_fail(c, OFK_E_HIP, "hipSetDevice failed");
    // With one slice and the double-buffered pyramid sets of the overlapped schedule the decoder's colour kernel writes GRAY straight
    // into level 0 of the set the next ofk_pairs_run will take (the one the run in flight is NOT reading), on the ingest stream: the
    // decoder passes of batch k + 1 go on beside the run of batch k, and that run's first stage is already done (gray_direct_set).  The
    // set was last read by the LK of the run before the latest one (ev_lkdone).  With several slices, or without the second set, the
    // old way: BGR into bgr[] on the context's stream, behind everything.
    bool direct = c->nstreams <= 1 && c->overlap != 0;
    for (int k = 0; k < 2 && direct; ++k)
        if (!c->pyr_alt[k] && hipMalloc((void **)&c->pyr_alt[k], (size_t)c->max_batch * c->pyr_stride) != hipSuccess) { (void)hipGetLastError(); direct = false; }
    int h = 0, w = 0, batch = 0;
    c->cur_batch = 0;
    c->gray_direct_set = -1;
    if (direct) {
        const int set = c->pyr_set;
        uint8_t *const P0 = set ? c->pyr_alt[0] : c->pyr[0], *const P1 = set ? c->pyr_alt[1] : c->pyr[1];
        TRY(ofk_jpeg_decode_staged_pairs(c, slot, P0, P1, c->pyr_stride, c->P, &batch, &h, &w, c->ev_lkdone[set], OFK_MAX_STREAMS, 1));
        c->gray_direct_set = set;
    } else {
        TRY(join_slices(c));
        TRY(ofk_jpeg_decode_staged_pairs(c, slot, c->bgr[0], c->bgr[1], c->bgr_stride, c->P, &batch, &h, &w, nullptr, 0, 0));
    }
    if (batch > c->max_batch) return ofk_fail(c, OFK_E_INVALID, "ofk_pairs_upload_staged: %d pairs exceed the context (%d)", batch, c->max_batch);
    c->cur_batch = batch; c->cur_h = h; c->cur_w = w;
    return OFK_OK;
}

extern "C" int ofk_jpeg_decode_bgr8(ofk_ctx *c, const uint8_t *const *jpeg, const size_t *nbytes, int batch, uint8_t *bgr)
{
    if (!c) return OFK_E_INVALID;
    if (hipSetDevice(c->device) != hipSuccess) return ofk_fail(c, OFK_E_HIP, "hipSetDevice failed");
    if (!bgr) return ofk_fail(c, OFK_E_INVALID, "ofk_jpeg_decode_bgr8: NULL output");
    TRY(join_slices(c));
    int h = 0, w = 0;
    uint8_t *dev = nullptr;
    size_t stride = 0;
    TRY(ofk_jpeg_decode_device(c, jpeg, nbytes, batch, nullptr, 0, 0, &h, &w, &dev, &stride));
    return d2h(c, bgr, dev, stride, (size_t)h * w * 3, batch);
}

extern "C" int ofk_pairs_set_sensors(ofk_ctx *c, const double *sensors, int batch)
{
    if (!c || !sensors || batch < 1 || batch > c->max_batch) return ofk_fail(c, OFK_E_INVALID, "ofk_pairs_set_sensors: bad argument");
    TRY(join_slices(c));
    OFK_HIP(c, hipMemcpyAsync(c->sensors, sensors, (size_t)batch * OFK_SENSOR_DOUBLES * 8, hipMemcpyHostToDevice, c->stream));
    OFK_HIP(c, hipStreamSynchronize(c->stream));
    return OFK_OK;
}

struct StageTimer {
    ofk_ctx *c; int stage; bool on; hipStream_t st; int slot;
    StageTimer(ofk_ctx *c_, int stg, hipStream_t s) : c(c_), stage(stg), on(((c_->prof_mask >> stg) & 1) && c_->ev_n + 2 <= c_->ev_cap), st(s), slot(c_->ev_n)
    {
        if (!on) return;
        c->ev_n += 2;
        if (!c->ev[slot]) hipEventCreate(&c->ev[slot]);
        if (!c->ev[slot + 1]) hipEventCreate(&c->ev[slot + 1]);
        hipEventRecord(c->ev[slot], st);
    }
    ~StageTimer()
    {
        if (!on) return;
        hipEventRecord(c->ev[slot + 1], st);
        c->ev_stage[slot / 2] = stage;
    }
};

extern "C" int ofk_pairs_run(ofk_ctx *c, const ofk_params *p)
{
    if (!c || !p) return OFK_E_INVALID;
    if (c->cur_batch < 1) return ofk_fail(c, OFK_E_INVALID, "ofk_pairs_run: no resident frame pairs (call ofk_pairs_upload)");
    if (hipSetDevice(c->device) != hipSuccess) return ofk_fail(c, OFK_E_HIP, "hipSetDevice failed");
    const int B = c->cur_batch, h = c->cur_h, w = c->cur_w;
    TRY(check_block(c, h, w, p->block_size));
    TRY(check_select(c, p->max_corners, p->quality, p->min_distance));
    TRY(check_lk(c, h, w, p->win, p->max_level));
    if (p->solve_variant != OFK_SOLVE_NODE && p->solve_variant != OFK_SOLVE_SIM) return ofk_fail(c, OFK_E_INVALID, "solve_variant must be NODE or SIM");
    const ofk_levels lv = ofk_make_levels(h, w, p->win, p->max_level);
    // The batch is cut into `nstreams` contiguous slices, each running the whole stage chain on its own stream: the
    // latency-bound stages of one slice (corner selection: one workgroup per image; the per-pair solve) overlap with the
    // streaming stages of the others.  Pairs are independent, so slices share nothing.
    const int S = c->nstreams < 1 ? 1 : (c->nstreams > B ? B : c->nstreams);
    // Inside a slice the chain forks once more: the response kernel and LK are VALU-bound, the gray conversions and the
    // pyramids HBM-bound, so the two kinds run beside each other.  The gray conversions and the pyramids (HBM-bound) run on an auxiliary stream and may even run AHEAD of the context's
    // stream: they write one of two pyramid buffer sets, alternating per call, so the next call's conversions overlap this
    // call's LK.  The only hazard is the set itself, still read by the LK of the call that used it last (ev_lkdone).
    bool overlap = c->overlap != 0;
    if (overlap && !c->pyr_alt[0]) {
        for (int k = 0; k < 2 && overlap; ++k)
            if (hipMalloc((void **)&c->pyr_alt[k], (size_t)c->max_batch * c->pyr_stride) != hipSuccess) { (void)hipGetLastError(); overlap = false; }
        if (!overlap) { for (int k = 0; k < 2; ++k) if (c->pyr_alt[k]) { hipFree(c->pyr_alt[k]); c->pyr_alt[k] = nullptr; } c->overlap = 0; }
    }
    int set = 0;
    if (overlap) { set = c->pyr_set; c->pyr_set ^= 1; }
    const bool have_gray = c->gray_direct_set >= 0;               // compressed ingest: the frames came in as gray level 0 of that set (and only there)
    if (have_gray) { set = c->gray_direct_set; c->pyr_set = set ^ 1; }
    c->pyr_last = set;
    uint8_t *const P0 = set ? c->pyr_alt[0] : c->pyr[0], *const P1 = set ? c->pyr_alt[1] : c->pyr[1];
    // Slices are forked off the context's stream once and then free-run over consecutive calls: nothing joins them until an
    // entry point needs the context's stream to see their results (join_slices).  On the fork the response kernels are chained
    // slice after slice, which offsets the slices by one response kernel for as long as they run.
    const bool fork = S > 1 && !c->slices_open;
    TRY(need_streams(c, S, overlap));
    if (fork) {
        OFK_HIP(c, hipEventRecord(c->ev_fork, c->stream));
        for (int k = 1; k < S; ++k) OFK_HIP(c, hipStreamWaitEvent(c->streams[k], c->ev_fork, 0));
    }
    for (int k = 0; k < S; ++k) {
        const int b0 = (int)((long long)B * k / S), nb = (int)((long long)B * (k + 1) / S) - b0;
        if (nb <= 0) continue;
        hipStream_t st = k == 0 ? c->stream : c->streams[k];
        hipStream_t sa = overlap ? c->aux[k] : st;
        if (overlap) OFK_HIP(c, hipStreamWaitEvent(sa, c->ev_lkdone[set][k], 0));
        uint8_t *bgr0 = c->bgr[0] + (size_t)b0 * c->bgr_stride, *bgr1 = c->bgr[1] + (size_t)b0 * c->bgr_stride;
        uint8_t *pyr0 = P0 + (size_t)b0 * c->pyr_stride, *pyr1 = P1 + (size_t)b0 * c->pyr_stride;
        unsigned int *maxbits = c->maxbits + (size_t)b0 * OFK_MAX_STRIDE;
        int *cand_count = c->cand_count + (size_t)b0 * OFK_CNT_STRIDE;
        unsigned long long *cand = c->cand + (size_t)b0 * c->cand_cap, *cand_seg = c->cand_seg + (size_t)b0 * c->seg_keys;
        int *seg_count = c->seg_count + (size_t)b0 * OFK_SEG_MAX;
        float *pts_prev = c->pts_prev + (size_t)b0 * c->max_pts * 2, *pts_next = c->pts_next + (size_t)b0 * c->max_pts * 2;
        uint8_t *status = c->status + (size_t)b0 * c->max_pts;
        float *err = c->err + (size_t)b0 * c->max_pts;
        int *counts = c->counts + b0;
        int nseg = 0, segcap = 0;
        if (!have_gray) {
            StageTimer t(c, OFK_STAGE_GRAY, sa);
            ofk_launch_gray(sa, bgr0, c->bgr_stride, pyr0, c->pyr_stride, nb, h, w);
        }
        if (overlap) OFK_HIP(c, hipEventRecord(c->ev_g0[k], sa));
        if (!have_gray) {
            StageTimer t(c, OFK_STAGE_GRAY, sa);
            ofk_launch_gray(sa, bgr1, c->bgr_stride, pyr1, c->pyr_stride, nb, h, w);
        }
        {
            StageTimer t(c, OFK_STAGE_PYR, sa);
            for (int l = ofk_launch_pyr3(sa, pyr0, pyr1, c->pyr_stride, lv, nb, 2 * nb) ? 4 : 1; l <= lv.n; ++l)
                ofk_launch_pyr_down2(sa, pyr0 + lv.off[l - 1], pyr1 + lv.off[l - 1], c->pyr_stride, lv.h[l - 1], lv.w[l - 1], pyr0 + lv.off[l],
                                     pyr1 + lv.off[l], c->pyr_stride, nb);
        }
        if (overlap) {
            OFK_HIP(c, hipEventRecord(c->ev_aux[k], sa));
            OFK_HIP(c, hipStreamWaitEvent(st, c->ev_g0[k], 0));              // the response kernel needs the previous frame's gray level
        }
        if (fork && k > 0) OFK_HIP(c, hipStreamWaitEvent(st, c->ev_stagger[k - 1], 0));
        {
            StageTimer t(c, OFK_STAGE_EIG, st);                  // response + 3x3 NMS + candidate keys, no map in HBM
            ofk_launch_zero_detect_state(st, maxbits, cand_count, c->sel_hist + (size_t)b0 * 1024, nb);   // one launch, histogram of the selection included
            if (ofk_launch_mineig_cand(st, pyr0, c->pyr_stride, h, w, p->block_size, maxbits, nullptr, 0, p->quality, cand, c->cand_cap,
                                       cand_count, cand_seg, c->seg_keys, seg_count, OFK_SEG_MAX, c->dev_flags, nb, &nseg, &segcap))
                return ofk_fail(c, OFK_E_INVALID, "corner response: block_size %d does not fit (LDS tile / key segments)", p->block_size);
        }
        if (fork) OFK_HIP(c, hipEventRecord(c->ev_stagger[k], st));
        {
            StageTimer t(c, OFK_STAGE_SELECT, st);
            ofk_launch_select(st, cand, c->cand_cap, cand_count, cand_seg, segcap, seg_count, nseg, maxbits, p->quality, h, w, p->max_corners,
                              (float)p->min_distance, pts_prev, c->max_pts, counts, nullptr, nb, c->sel_hist + (size_t)b0 * 1024,
                              c->sel_keys + (size_t)b0 * OFK_CHUNK, true);
        }
        if (overlap) OFK_HIP(c, hipStreamWaitEvent(st, c->ev_aux[k], 0));    // LK needs both pyramids
        {
            StageTimer t(c, OFK_STAGE_LK, st);
            ofk_launch_lk(st, pyr0, pyr1, c->pyr_stride, lv, pts_prev, counts, c->max_pts, p->win, p->max_count, p->eps, p->min_eig_thr,
                          pts_next, status, err, nb);
        }
        if (overlap) OFK_HIP(c, hipEventRecord(c->ev_lkdone[set][k], st));   // this pyramid set may be rewritten from here on
        if (c->x_pending) OFK_HIP(c, hipStreamWaitEvent(st, c->ev_x, 0));    // the previous call's records are still being exported
        {
            StageTimer t(c, OFK_STAGE_SOLVE, st);
            ofk_launch_pairs_solve(st, pts_prev, pts_next, status, counts, c->max_pts, c->sensors + (size_t)b0 * OFK_SENSOR_DOUBLES,
                                   p->solve_variant, p->use_feasibility, p->feas_T, cand_count, c->records + (size_t)b0 * OFK_RECORD_DOUBLES, nb);
        }
        if (S > 1) OFK_HIP(c, hipEventRecord(c->ev_end[k], st));             // joined lazily (join_slices), not here
    }
    if (S > 1) { c->slices_open = 1; c->open_slices = S; }
    return check_launch(c, "ofk_pairs_run");
}

extern "C" int ofk_pairs_download(ofk_ctx *c, double *records, float *prev_pts, float *next_pts, uint8_t *status, float *err,
                                  int *counts)
{
    if (!c || c->cur_batch < 1) return ofk_fail(c, OFK_E_INVALID, "ofk_pairs_download: nothing resident");
    if (hipSetDevice(c->device) != hipSuccess) return ofk_fail(c, OFK_E_HIP, "hipSetDevice failed");
    const int B = c->cur_batch;
    const size_t np = (size_t)B * c->max_pts;
    int flags[4];
    TRY(join_slices(c));
    OFK_HIP(c, hipMemcpyAsync(flags, c->dev_flags, 16, hipMemcpyDeviceToHost, c->stream));
    if (records) OFK_HIP(c, hipMemcpyAsync(records, c->records, (size_t)B * OFK_RECORD_DOUBLES * 8, hipMemcpyDeviceToHost, c->stream));
    if (prev_pts) OFK_HIP(c, hipMemcpyAsync(prev_pts, c->pts_prev, np * 8, hipMemcpyDeviceToHost, c->stream));
    if (next_pts) OFK_HIP(c, hipMemcpyAsync(next_pts, c->pts_next, np * 8, hipMemcpyDeviceToHost, c->stream));
    if (status) OFK_HIP(c, hipMemcpyAsync(status, c->status, np, hipMemcpyDeviceToHost, c->stream));
    if (err) OFK_HIP(c, hipMemcpyAsync(err, c->err, np * 4, hipMemcpyDeviceToHost, c->stream));
    if (counts) OFK_HIP(c, hipMemcpyAsync(counts, c->counts, (size_t)B * 4, hipMemcpyDeviceToHost, c->stream));
    OFK_HIP(c, hipStreamSynchronize(c->stream));
    bool over = (flags[0] & 1) != 0;
    if (counts)
        for (int b = 0; b < B; ++b) if (counts[b] < 0) { over = true; counts[b] = 0; }
    if (over) {
        hipMemsetAsync(c->dev_flags, 0, 16, c->stream);
        return ofk_fail(c, OFK_E_CAPACITY, "corner candidates exceeded the per-image capacity (%d)", c->cand_cap);
    }
    return OFK_OK;
}

// k_records_f32 of the latest ofk_pairs_run on the stream that ends the step (the last slice's, behind the other slices' end
// events); *stream_out lets a caller queue more work behind it (the RCCL gather of ofk_comm.hip)
int ofk_export_records_stream(ofk_ctx *c, float *device_dst, int batch, hipStream_t *stream_out)
{
    hipStream_t s;
    TRY(tail_stream(c, &s));
    ofk_launch_records_f32(s, c->records, device_dst, batch);
    if (c->slices_open) {                                        // the other slices' next solve must not overtake the export
        OFK_HIP(c, hipEventRecord(c->ev_x, s));
        c->x_pending = 1;
    }
    if (stream_out) *stream_out = s;
    return check_launch(c, "k_records_f32");
}

extern "C" int ofk_pairs_export_records_f32(ofk_ctx *c, void *device_dst, int batch)
{
    if (!c || !device_dst || batch < 1 || batch > c->cur_batch) return ofk_fail(c, OFK_E_INVALID, "ofk_pairs_export_records_f32: bad argument");
    return ofk_export_records_stream(c, (float *)device_dst, batch, nullptr);
}

// ------------------------------------------------------------------------------------------------ video streams
static int stream_alloc(ofk_ctx *c)
{
    if (!c->pts_new) {
        OFK_HIP(c, hipMalloc((void **)&c->pts_new, (size_t)c->max_batch * c->max_pts * 8));
        OFK_HIP(c, hipMalloc((void **)&c->new_counts, (size_t)c->max_batch * 4));
        OFK_HIP(c, hipMalloc((void **)&c->limit, (size_t)c->max_batch * 4));
    }
    return lazy_mask(c);
}

// gray + pyramid of a batch of BGR frames (host) into pyramid slot k
static int stream_ingest(ofk_ctx *c, int k, const uint8_t *bgr, int batch, int h, int w, const ofk_levels &lv)
{
    if (bgr) TRY(h2d(c, c->bgr[k], c->bgr_stride, bgr, (size_t)h * w * 3, batch));      // NULL: the frames are in bgr[k] already (JPEG ingest)
    c->pyr_last = 0;
    ofk_launch_gray(c->stream, c->bgr[k], c->bgr_stride, c->pyr[k], c->pyr_stride, batch, h, w);
    for (int l = ofk_launch_pyr3(c->stream, c->pyr[k], nullptr, c->pyr_stride, lv, batch, batch) ? 4 : 1; l <= lv.n; ++l)
        ofk_launch_pyr_down(c->stream, c->pyr[k] + lv.off[l - 1], c->pyr_stride, lv.h[l - 1], lv.w[l - 1], c->pyr[k] + lv.off[l],
                            c->pyr_stride, batch);
    return OFK_OK;
}

// corners of pyramid slot k (level 0) -> dst/dst_counts, optional mask and per-stream budget
static int stream_detect(ofk_ctx *c, int k, const uint8_t *dmask, const int *limit, int batch, int h, int w, const ofk_params *p,
                         float *dst, int *dst_counts)
{
    int nseg = 0, segcap = 0;
    OFK_HIP(c, hipMemsetAsync(c->maxbits, 0, (size_t)batch * OFK_MAX_STRIDE * 4, c->stream));
    OFK_HIP(c, hipMemsetAsync(c->cand_count, 0, (size_t)batch * OFK_CNT_STRIDE * 4, c->stream));
    if (ofk_launch_mineig_cand(c->stream, c->pyr[k], c->pyr_stride, h, w, p->block_size, c->maxbits, dmask, c->img_stride, p->quality, c->cand,
                               c->cand_cap, c->cand_count, c->cand_seg, c->seg_keys, c->seg_count, OFK_SEG_MAX, c->dev_flags, batch, &nseg,
                               &segcap))
        return ofk_fail(c, OFK_E_INVALID, "corner response: block_size %d does not fit (LDS tile / key segments)", p->block_size);
    ofk_launch_select(c->stream, c->cand, c->cand_cap, c->cand_count, c->cand_seg, segcap, c->seg_count, nseg, c->maxbits, p->quality, h, w,
                      p->max_corners, (float)p->min_distance, dst, c->max_pts, dst_counts, limit, batch, c->sel_hist, c->sel_keys);
    return check_launch(c, "stream corner detection");
}

static int stream_fetch_tracks(ofk_ctx *c, int batch, int max_corners, float *tracks, int *counts)
{
    int flags[4];
    if (!c->h_counts) c->h_counts = (int *)calloc(c->max_batch, sizeof(int));
    OFK_HIP(c, hipMemcpyAsync(flags, c->dev_flags, 16, hipMemcpyDeviceToHost, c->stream));
    OFK_HIP(c, hipMemcpyAsync(c->h_counts, c->counts, (size_t)batch * 4, hipMemcpyDeviceToHost, c->stream));   // also tells the next step who re-detects
    if (tracks) OFK_HIP(c, hipMemcpy2DAsync(tracks, (size_t)max_corners * 8, c->pts_prev, (size_t)c->max_pts * 8, (size_t)max_corners * 8, batch,
                                            hipMemcpyDeviceToHost, c->stream));
    OFK_HIP(c, hipStreamSynchronize(c->stream));
    bool over = (flags[0] & 1) != 0;
    for (int b = 0; b < batch; ++b) if (c->h_counts[b] < 0) { over = true; c->h_counts[b] = 0; }
    if (counts) memcpy(counts, c->h_counts, (size_t)batch * 4);
    if (over) {
        hipMemsetAsync(c->dev_flags, 0, 16, c->stream);
        return ofk_fail(c, OFK_E_CAPACITY, "corner candidates exceeded the per-image capacity (%d)", c->cand_cap);
    }
    return OFK_OK;
}

// first_bgr == NULL: the first frames have been decoded into bgr[0] already (ofk_stream_begin_jpeg)
static int stream_begin_impl(ofk_ctx *c, const uint8_t *first_bgr, int batch, int h, int w, const ofk_params *p, float *tracks, int *counts)
{
    TRY(check_block(c, h, w, p->block_size));
    TRY(check_select(c, p->max_corners, p->quality, p->min_distance));
    TRY(check_lk(c, h, w, p->win, p->max_level));
    TRY(stream_alloc(c));
    const ofk_levels lv = ofk_make_levels(h, w, p->win, p->max_level);
    TRY(stream_ingest(c, 0, first_bgr, batch, h, w, lv));
    TRY(stream_detect(c, 0, nullptr, nullptr, batch, h, w, p, c->pts_prev, c->counts));
    c->stream_h = h; c->stream_w = w; c->stream_batch = batch;
    return stream_fetch_tracks(c, batch, p->max_corners, tracks, counts);
}

extern "C" int ofk_stream_begin(ofk_ctx *c, const uint8_t *first_bgr, int batch, int h, int w, const ofk_params *p, float *tracks,
                                int *counts)
{
    TRY(check_geom(c, batch, h, w, "ofk_stream_begin"));
    if (!first_bgr || !p) return ofk_fail(c, OFK_E_INVALID, "ofk_stream_begin: NULL argument");
    return stream_begin_impl(c, first_bgr, batch, h, w, p, tracks, counts);
}

// The same with the frames arriving as baseline JPEG streams (CompressedImage payloads, node:112): decoded on the device straight
// into the stream's frame buffer, no decoded frame ever crosses PCIe.
extern "C" int ofk_stream_begin_jpeg(ofk_ctx *c, const uint8_t *const *jpeg, const size_t *nbytes, int batch, const ofk_params *p,
                                     float *tracks, int *counts)
{
    if (!c || !jpeg || !nbytes || !p || batch < 1) return ofk_fail(c, OFK_E_INVALID, "ofk_stream_begin_jpeg: bad argument");
    int h = 0, w = 0;
    if (ofk_jpeg_info(jpeg[0], nbytes[0], &h, &w, nullptr) != OFK_OK) return ofk_fail(c, OFK_E_INVALID, "ofk_stream_begin_jpeg: stream 0 is not a supported JPEG stream");
    TRY(check_geom(c, batch, h, w, "ofk_stream_begin_jpeg"));
    c->stream_batch = 0;
    TRY(ofk_jpeg_decode_device(c, jpeg, nbytes, batch, c->bgr[0], c->bgr_stride, c->P, &h, &w, nullptr, nullptr));
    return stream_begin_impl(c, nullptr, batch, h, w, p, tracks, counts);
}

// next_bgr == NULL: the new frames are in bgr[1] already (ofk_stream_step_jpeg)
static int filters_alloc(ofk_ctx *c)
{
    const size_t B = (size_t)c->max_batch;
    if (!c->imu_state) {
        OFK_HIP(c, hipMalloc((void **)&c->imu_state, B * OFK_IMU_STATE * 8)); OFK_HIP(c, hipMalloc((void **)&c->imu_dv, B * 24));
        OFK_HIP(c, hipMalloc((void **)&c->kf_mats, 5 * 36 * 8)); OFK_HIP(c, hipMalloc((void **)&c->kf_x, B * 6 * 8));
        OFK_HIP(c, hipMalloc((void **)&c->kf_P, B * 36 * 8)); OFK_HIP(c, hipMalloc((void **)&c->fused, B * 8 * 8));
        OFK_HIP(c, hipMemsetAsync(c->imu_dv, 0, B * 24, c->stream)); OFK_HIP(c, hipMemsetAsync(c->fused, 0, B * 64, c->stream));
        OFK_HIP(c, hipMemsetAsync(c->kf_mats, 0, 5 * 36 * 8, c->stream));
        // the node's initial state (node:182-217): vel 0.1, first message pending, rotation I, normal e_z
        double init[OFK_IMU_STATE] = {0.1, 0.1, 0.1, 0, 0, 1, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0};
        for (size_t b = 0; b < B; ++b) OFK_HIP(c, hipMemcpyAsync(c->imu_state + b * OFK_IMU_STATE, init, sizeof init, hipMemcpyHostToDevice, c->stream));
        OFK_HIP(c, hipStreamSynchronize(c->stream));
    }
    return OFK_OK;
}

extern "C" int ofk_imu_reset(ofk_ctx *c, const double *state0, int batch)
{
    if (!c || batch < 1 || batch > c->max_batch) return ofk_fail(c, OFK_E_INVALID, "ofk_imu_reset: bad argument");
    OFK_HIP(c, hipSetDevice(c->device));
    TRY(join_slices(c));
    TRY(filters_alloc(c));
    double init[OFK_IMU_STATE] = {0.1, 0.1, 0.1, 0, 0, 1, 1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 1, 0, 0, 0, 0, 0, 0};
    const double *src = state0 ? state0 : init;
    for (int b = 0; b < batch; ++b) OFK_HIP(c, hipMemcpyAsync(c->imu_state + (size_t)b * OFK_IMU_STATE, src, sizeof init, hipMemcpyHostToDevice, c->stream));
    OFK_HIP(c, hipMemsetAsync(c->imu_dv, 0, (size_t)batch * 24, c->stream));
    OFK_HIP(c, hipStreamSynchronize(c->stream));
    return OFK_OK;
}

extern "C" int ofk_imu_push(ofk_ctx *c, const double *msgs, const int *counts, int max_msgs, int batch)
{
    if (!c || !msgs || !counts || max_msgs < 1 || batch < 1 || batch > c->max_batch) return ofk_fail(c, OFK_E_INVALID, "ofk_imu_push: bad argument");
    OFK_HIP(c, hipSetDevice(c->device));
    TRY(join_slices(c));
    TRY(filters_alloc(c));
    const size_t mb = (size_t)batch * max_msgs * OFK_IMU_MSG * 8;
    if (mb > c->imu_msgs_bytes) {
        if (c->imu_msgs) { OFK_HIP(c, hipStreamSynchronize(c->stream)); hipFree(c->imu_msgs); c->imu_msgs = nullptr; c->imu_msgs_bytes = 0; }
        OFK_HIP(c, hipMalloc((void **)&c->imu_msgs, mb));
        c->imu_msgs_bytes = mb;
    }
    if (!c->imu_counts) OFK_HIP(c, hipMalloc((void **)&c->imu_counts, (size_t)c->max_batch * 4));
    OFK_HIP(c, hipMemcpyAsync(c->imu_msgs, msgs, mb, hipMemcpyHostToDevice, c->stream));
    OFK_HIP(c, hipMemcpyAsync(c->imu_counts, counts, (size_t)batch * 4, hipMemcpyHostToDevice, c->stream));
    ofk_launch_imu_seq(c->stream, c->imu_state, c->imu_dv, c->imu_msgs, c->imu_counts, max_msgs, batch);
    TRY(check_launch(c, "k_imu_seq"));
    OFK_HIP(c, hipStreamSynchronize(c->stream));                  // the host buffers are the caller's again
    return OFK_OK;
}

extern "C" int ofk_imu_state(ofk_ctx *c, double *state, double *dv, int batch)
{
    if (!c || !state || batch < 1 || batch > c->max_batch) return ofk_fail(c, OFK_E_INVALID, "ofk_imu_state: bad argument");
    OFK_HIP(c, hipSetDevice(c->device));
    TRY(join_slices(c));
    TRY(filters_alloc(c));
    if (dv) OFK_HIP(c, hipMemcpyAsync(dv, c->imu_dv, (size_t)batch * 24, hipMemcpyDeviceToHost, c->stream));
    return get(c, state, c->imu_state, (size_t)batch * OFK_IMU_STATE * 8);
}

extern "C" int ofk_filter_configure(ofk_ctx *c, int ns, int nm, int nc, const double *F, const double *Bm, const double *H, const double *Q,
                                    const double *Rm, const double *x0, const double *P0, int batch)
{
    if (!c || ns < 1 || ns > 6 || nm < 1 || nm > 6 || nc < 0 || nc > 6 || !F || !H || !Q || !Rm || !x0 || !P0 || (nc && !Bm) || batch < 1 ||
        batch > c->max_batch)
        return ofk_fail(c, OFK_E_INVALID, "ofk_filter_configure: bad argument");
    OFK_HIP(c, hipSetDevice(c->device));
    TRY(join_slices(c));
    TRY(filters_alloc(c));
    double mats[5 * 36] = {0};
    memcpy(mats, F, (size_t)ns * ns * 8); if (nc) memcpy(mats + 36, Bm, (size_t)ns * nc * 8);
    memcpy(mats + 72, H, (size_t)nm * ns * 8); memcpy(mats + 108, Q, (size_t)ns * ns * 8); memcpy(mats + 144, Rm, (size_t)nm * nm * 8);
    OFK_HIP(c, hipMemcpyAsync(c->kf_mats, mats, sizeof mats, hipMemcpyHostToDevice, c->stream));
    for (int b = 0; b < batch; ++b) {
        OFK_HIP(c, hipMemcpyAsync(c->kf_x + (size_t)b * ns, x0, (size_t)ns * 8, hipMemcpyHostToDevice, c->stream));
        OFK_HIP(c, hipMemcpyAsync(c->kf_P + (size_t)b * ns * ns, P0, (size_t)ns * ns * 8, hipMemcpyHostToDevice, c->stream));
    }
    OFK_HIP(c, hipStreamSynchronize(c->stream));
    c->kf_ns = ns; c->kf_nm = nm; c->kf_nc = nc;
    return OFK_OK;
}

extern "C" int ofk_filter_state(ofk_ctx *c, double *x, double *P, int batch)
{
    if (!c || !x || !P || batch < 1 || batch > c->max_batch || c->kf_ns < 1) return ofk_fail(c, OFK_E_INVALID, "ofk_filter_state: no filter configured / bad argument");
    OFK_HIP(c, hipSetDevice(c->device));
    TRY(join_slices(c));
    OFK_HIP(c, hipMemcpyAsync(x, c->kf_x, (size_t)batch * c->kf_ns * 8, hipMemcpyDeviceToHost, c->stream));
    return get(c, P, c->kf_P, (size_t)batch * c->kf_ns * c->kf_ns * 8);
}

// Per-pair filter update of the resident batch behind the latest ofk_pairs_run (asynchronous, on the stream that ends the step):
// predict + correct(z_sign * v) for every pair from its own record; the filter states (ofk_filter_configure) never leave the device.
extern "C" int ofk_pairs_filter_step(ofk_ctx *c, double z_sign, int z_source, int batch)
{
    if (!c || batch < 1 || batch > c->cur_batch || c->kf_ns < 1) return ofk_fail(c, OFK_E_INVALID, "ofk_pairs_filter_step: no filter configured / bad batch");
    OFK_HIP(c, hipSetDevice(c->device));
    hipStream_t s;
    TRY(tail_stream(c, &s));
    ofk_launch_kf_records(s, c->kf_ns, c->kf_nm, c->kf_mats, c->kf_x, c->kf_P, c->records, z_sign, z_source, batch);
    if (c->slices_open) {                                        // the next solves must not overwrite the records the filter still reads
        OFK_HIP(c, hipEventRecord(c->ev_x, s));
        c->x_pending = 1;
    }
    return check_launch(c, "k_kf_records");
}

// fu == NULL: the plain step (status filter, node-style solve, no resident filters)
static int stream_step_impl(ofk_ctx *c, const uint8_t *next_bgr, const double *sensors, const ofk_params *p, const ofk_fusion *fu,
                            int min_features, int mask_radius, double *records, double *fused, float *tracks, int *counts)
{
    const int B = c->stream_batch, h = c->stream_h, w = c->stream_w;
    TRY(check_block(c, h, w, p->block_size));
    TRY(check_select(c, p->max_corners, p->quality, p->min_distance));
    TRY(check_lk(c, h, w, p->win, p->max_level));
    if (mask_radius < 0 || mask_radius > 255) return ofk_fail(c, OFK_E_INVALID, "mask_radius outside 0..255");
    if (fu) {
        if (p->solve_variant < OFK_SOLVE_NODE || p->solve_variant > OFK_SOLVE_OFMODULE) return ofk_fail(c, OFK_E_INVALID, "solve_variant must be NODE, SIM or OFMODULE");
        if (p->solve_variant == OFK_SOLVE_OFMODULE && fu->keep != OFK_KEEP_LEGACY)
            return ofk_fail(c, OFK_E_INVALID, "OFK_SOLVE_OFMODULE weights its rows with the legacy r_tilde distances: it needs keep = OFK_KEEP_LEGACY");
        if (fu->filter && c->kf_ns < 1) return ofk_fail(c, OFK_E_INVALID, "ofk_stream_step_fused: filter requested but ofk_filter_configure was not called");
        if (fu->flow < 0 || fu->flow > 1 || fu->keep < 0 || fu->keep > 1 || fu->control < 0 || fu->control > 1 || fu->min_solve < 0)
            return ofk_fail(c, OFK_E_INVALID, "ofk_stream_step_fused: bad ofk_fusion field");
        if (fu->hold_on_skip && B != 1) return ofk_fail(c, OFK_E_INVALID, "ofk_fusion.hold_on_skip needs a context with one stream (%d here): a batch shares one frame swap", B);
        TRY(filters_alloc(c));
    } else if (p->solve_variant != OFK_SOLVE_NODE && p->solve_variant != OFK_SOLVE_SIM) return ofk_fail(c, OFK_E_INVALID, "solve_variant must be NODE or SIM");
    const ofk_levels lv = ofk_make_levels(h, w, p->win, p->max_level);
    OFK_HIP(c, hipMemcpyAsync(c->sensors, sensors, (size_t)B * OFK_SENSOR_DOUBLES * 8, hipMemcpyHostToDevice, c->stream));
    bool few = false;                                            // the host knows the track counts from the previous call
    for (int b = 0; b < B; ++b) few = few || (c->h_counts && c->h_counts[b] <= min_features);
    if (fu && fu->redetect_replace && few) {
        // of_module.py:83-86: before tracking, streams with few tracks replace them by fresh corners of the PREVIOUS frame (no mask)
        ofk_launch_redetect_limits(c->stream, c->counts, min_features, p->max_corners, c->limit, B);
        TRY(stream_detect(c, 0, nullptr, c->limit, B, h, w, p, c->pts_new, c->new_counts));
        ofk_launch_replace_tracks(c->stream, c->limit, c->pts_new, c->new_counts, c->max_pts, c->pts_prev, c->counts, B);
    }
    TRY(stream_ingest(c, 1, next_bgr, B, h, w, lv));
    // track (node:133), solve on the tracked points (node:229-258)
    ofk_launch_lk(c->stream, c->pyr[0], c->pyr[1], c->pyr_stride, lv, c->pts_prev, c->counts, c->max_pts, p->win, p->max_count, p->eps,
                  p->min_eig_thr, c->pts_next, c->status, c->err, B);
    if (fu)
        ofk_launch_stream_fuse(c->stream, c->pts_prev, c->pts_next, c->status, c->counts, c->max_pts, c->sensors, c->imu_state, c->imu_dv,
                               c->kf_ns, c->kf_nm, c->kf_nc, c->kf_mats, c->kf_x, c->kf_P, fu, p->solve_variant, p->use_feasibility, p->feas_T,
                               c->records, c->fused, B);
    else
        ofk_launch_pairs_solve(c->stream, c->pts_prev, c->pts_next, c->status, c->counts, c->max_pts, c->sensors, p->solve_variant,
                               p->use_feasibility, p->feas_T, nullptr, c->records, B);
    // re-detection for the streams that had few features (node:157-166): mask = discs around the OLD positions, image = OLD frame.
    // The host knows the track counts from the previous call, so the whole branch is skipped when no stream needs it.
    // of_module.py:138 `continue`: a step of the ONE stream that did not solve leaves old_gray / old_pos as they were.  The host has to
    // know before it queues the track update, hence one wait (hold_on_skip is opt-in).
    bool hold = false;
    if (fu && fu->hold_on_skip) {
        double solved = 1.0;
        OFK_HIP(c, hipMemcpyAsync(&solved, c->records + 15, 8, hipMemcpyDeviceToHost, c->stream));
        OFK_HIP(c, hipStreamSynchronize(c->stream));
        hold = solved == 0.0;
    }
    const bool any = few && !(fu && fu->redetect_replace) && !hold;
    if (any) {
        ofk_launch_redetect_limits(c->stream, c->counts, min_features, p->max_corners, c->limit, B);
        OFK_HIP(c, hipMemsetAsync(c->mask, 1, (size_t)B * c->img_stride, c->stream));
        ofk_launch_disc_mask(c->stream, c->mask, c->img_stride, h, w, c->pts_prev, c->counts, c->max_pts, mask_radius, c->limit, B);
        TRY(stream_detect(c, 0, c->mask, c->limit, B, h, w, p, c->pts_new, c->new_counts));
    }
    // tracks := new[status == 1] ++ re-detected (node:134,166); the new frame becomes the previous one (node:175)
    if (!hold)
        ofk_launch_update_tracks(c->stream, c->pts_next, c->status, c->counts, c->max_pts, any ? c->pts_new : nullptr, any ? c->new_counts : nullptr,
                                 c->pts_prev, c->counts, p->max_corners, B);
    TRY(check_launch(c, "ofk_stream_step"));
    if (records) OFK_HIP(c, hipMemcpyAsync(records, c->records, (size_t)B * OFK_RECORD_DOUBLES * 8, hipMemcpyDeviceToHost, c->stream));
    if (fu && fused) OFK_HIP(c, hipMemcpyAsync(fused, c->fused, (size_t)B * 64, hipMemcpyDeviceToHost, c->stream));
    // The tracks on the device already belong to the new frame (k_update_tracks), so the frame swap happens whatever the fetch
    // reports (OFK_E_CAPACITY from a re-detection, a failed copy): the next step must track against THIS frame's pyramid.
    const int rc = stream_fetch_tracks(c, B, p->max_corners, tracks, counts);
    if (hold) return rc;                                         // old frame and old tracks stay; the next frame overwrites slot 1
    uint8_t *t = c->pyr[0]; c->pyr[0] = c->pyr[1]; c->pyr[1] = t;
    t = c->bgr[0]; c->bgr[0] = c->bgr[1]; c->bgr[1] = t;
    return rc;
}

extern "C" int ofk_stream_step(ofk_ctx *c, const uint8_t *next_bgr, const double *sensors, const ofk_params *p, int min_features,
                               int mask_radius, double *records, float *tracks, int *counts)
{
    if (!c || !next_bgr || !sensors || !p) return ofk_fail(c, OFK_E_INVALID, "ofk_stream_step: NULL argument");
    if (c->stream_batch < 1) return ofk_fail(c, OFK_E_INVALID, "ofk_stream_step: call ofk_stream_begin first");
    if (hipSetDevice(c->device) != hipSuccess) return ofk_fail(c, OFK_E_HIP, "hipSetDevice failed");
    TRY(join_slices(c));
    return stream_step_impl(c, next_bgr, sensors, p, nullptr, min_features, mask_radius, records, nullptr, tracks, counts);
}

extern "C" int ofk_stream_step_jpeg(ofk_ctx *c, const uint8_t *const *jpeg, const size_t *nbytes, const double *sensors, const ofk_params *p,
                                    int min_features, int mask_radius, double *records, float *tracks, int *counts)
{
    if (!c || !jpeg || !nbytes || !sensors || !p) return ofk_fail(c, OFK_E_INVALID, "ofk_stream_step_jpeg: NULL argument");
    if (c->stream_batch < 1) return ofk_fail(c, OFK_E_INVALID, "ofk_stream_step_jpeg: call ofk_stream_begin / ofk_stream_begin_jpeg first");
    if (hipSetDevice(c->device) != hipSuccess) return ofk_fail(c, OFK_E_HIP, "hipSetDevice failed");
    TRY(join_slices(c));
    int h = 0, w = 0;
    TRY(ofk_jpeg_decode_device(c, jpeg, nbytes, c->stream_batch, c->bgr[1], c->bgr_stride, c->P, &h, &w, nullptr, nullptr));
    if (h != c->stream_h || w != c->stream_w)
        return ofk_fail(c, OFK_E_INVALID, "ofk_stream_step_jpeg: frames are %dx%d, the streams were begun with %dx%d", w, h, c->stream_w, c->stream_h);
    return stream_step_impl(c, nullptr, sensors, p, nullptr, min_features, mask_radius, records, nullptr, tracks, counts);
}

// next positions and keep flags of the LATEST stream step (they stay in place until the next step): what a caller needs to form
// the flow of the kept points, new - old, against the tracks it received before the step (node:134-136)
extern "C" int ofk_stream_last_points(ofk_ctx *c, float *next_pts, uint8_t *keep, int stride)
{
    if (!c || !next_pts || !keep || stride < 1 || stride > c->max_pts) return ofk_fail(c, OFK_E_INVALID, "ofk_stream_last_points: bad argument");
    if (c->stream_batch < 1) return ofk_fail(c, OFK_E_INVALID, "ofk_stream_last_points: no active streams");
    OFK_HIP(c, hipSetDevice(c->device));
    TRY(join_slices(c));
    OFK_HIP(c, hipMemcpy2DAsync(next_pts, (size_t)stride * 8, c->pts_next, (size_t)c->max_pts * 8, (size_t)stride * 8, c->stream_batch, hipMemcpyDeviceToHost, c->stream));
    OFK_HIP(c, hipMemcpy2DAsync(keep, stride, c->status, c->max_pts, stride, c->stream_batch, hipMemcpyDeviceToHost, c->stream));
    OFK_HIP(c, hipStreamSynchronize(c->stream));
    return OFK_OK;
}

extern "C" int ofk_stream_step_fused(ofk_ctx *c, const uint8_t *next_bgr, const double *sensors, const ofk_params *p, const ofk_fusion *f,
                                     int min_features, int mask_radius, double *records, double *fused, float *tracks, int *counts)
{
    if (!c || !next_bgr || !sensors || !p || !f) return ofk_fail(c, OFK_E_INVALID, "ofk_stream_step_fused: NULL argument");
    if (c->stream_batch < 1) return ofk_fail(c, OFK_E_INVALID, "ofk_stream_step_fused: call ofk_stream_begin first");
    if (hipSetDevice(c->device) != hipSuccess) return ofk_fail(c, OFK_E_HIP, "hipSetDevice failed");
    TRY(join_slices(c));
    return stream_step_impl(c, next_bgr, sensors, p, f, min_features, mask_radius, records, fused, tracks, counts);
}

extern "C" int ofk_stream_step_fused_jpeg(ofk_ctx *c, const uint8_t *const *jpeg, const size_t *nbytes, const double *sensors, const ofk_params *p,
                                          const ofk_fusion *f, int min_features, int mask_radius, double *records, double *fused, float *tracks,
                                          int *counts)
{
    if (!c || !jpeg || !nbytes || !sensors || !p || !f) return ofk_fail(c, OFK_E_INVALID, "ofk_stream_step_fused_jpeg: NULL argument");
    if (c->stream_batch < 1) return ofk_fail(c, OFK_E_INVALID, "ofk_stream_step_fused_jpeg: call ofk_stream_begin / ofk_stream_begin_jpeg first");
    if (hipSetDevice(c->device) != hipSuccess) return ofk_fail(c, OFK_E_HIP, "hipSetDevice failed");
    TRY(join_slices(c));
    int h = 0, w = 0;
    TRY(ofk_jpeg_decode_device(c, jpeg, nbytes, c->stream_batch, c->bgr[1], c->bgr_stride, c->P, &h, &w, nullptr, nullptr));
    if (h != c->stream_h || w != c->stream_w)
        return ofk_fail(c, OFK_E_INVALID, "ofk_stream_step_fused_jpeg: frames are %dx%d, the streams were begun with %dx%d", w, h, c->stream_w, c->stream_h);
    return stream_step_impl(c, nullptr, sensors, p, f, min_features, mask_radius, records, fused, tracks, counts);
}

extern "C" int ofk_set_streams(ofk_ctx *c, int nstreams)
{
    if (!c || nstreams < 1 || nstreams > OFK_MAX_STREAMS) return ofk_fail(c, OFK_E_INVALID, "ofk_set_streams: 1..%d", OFK_MAX_STREAMS);
    // A different slice count moves the slice boundaries: a new slice's auxiliary stream would only wait for the LK of the OLD
    // slice with the same index (ev_lkdone[set][k]) while another old slice may still read the pyramid rows it is about to
    // rewrite.  Changing the schedule is rare (set-up time), so drain everything.
    TRY(drain_all(c));
    c->nstreams = nstreams;
    if (nstreams > 1) {
        const char *q = getenv("GPU_MAX_HW_QUEUES");
        static bool warned = false;
        if (!warned && q && atoi(q) > 0 && atoi(q) < 2 * nstreams) {
            fprintf(stderr, "libofk: GPU_MAX_HW_QUEUES=%s, but %d slices keep %d streams busy: streams that share a hardware queue run in order "
                            "(set GPU_MAX_HW_QUEUES >= %d before the first HIP call)\n", q, nstreams, 2 * nstreams, 2 * nstreams);
            warned = true;
        }
    }
    // Create the slice and auxiliary streams NOW: the runtime deals streams onto its hardware queues in creation order, and a
    // library that creates streams of its own later (RCCL does at ncclCommInitRank) must not get in between - with the
    // pipeline's streams created lazily behind RCCL's, two of them shared a queue and the rate fell by 10 % (99 k -> 87 k).
    return need_streams(c, nstreams, c->overlap != 0);
}

extern "C" int ofk_set_overlap(ofk_ctx *c, int on)
{
    if (!c) return OFK_E_INVALID;
    TRY(drain_all(c));                                           // same hazard as ofk_set_streams
    c->overlap = on ? 1 : 0;
    return OFK_OK;
}

extern "C" int ofk_mark(ofk_ctx *c, int slot)
{
    if (!c || slot < 0 || slot >= 8) return ofk_fail(c, OFK_E_INVALID, "ofk_mark: slot 0..7");
    OFK_HIP(c, hipSetDevice(c->device));
    if (!c->marks[slot]) OFK_HIP(c, hipEventCreateWithFlags(&c->marks[slot], hipEventDisableTiming));
    hipStream_t s;
    TRY(tail_stream(c, &s));                                     // "everything enqueued so far" includes every slice
    OFK_HIP(c, hipEventRecord(c->marks[slot], s));
    return OFK_OK;
}

extern "C" int ofk_mark_wait(ofk_ctx *c, int slot)
{
    if (!c || slot < 0 || slot >= 8) return ofk_fail(c, OFK_E_INVALID, "ofk_mark_wait: slot 0..7");
    if (!c->marks[slot]) return OFK_OK;
    OFK_HIP(c, hipEventSynchronize(c->marks[slot]));
    return OFK_OK;
}

extern "C" int ofk_profile_enable(ofk_ctx *c, int stage_mask)
{
    if (!c) return OFK_E_INVALID;
    c->prof_mask = stage_mask;
    return OFK_OK;
}

extern "C" int ofk_profile_read(ofk_ctx *c, double *ms_total, int *launches)
{
    if (!c || !ms_total || !launches) return OFK_E_INVALID;
    TRY(join_slices(c));
    OFK_HIP(c, hipStreamSynchronize(c->stream));
    for (int s = 0; s < OFK_N_STAGES; ++s) { ms_total[s] = 0.0; launches[s] = 0; }
    for (int i = 0; i + 1 < c->ev_n; i += 2) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, c->ev[i], c->ev[i + 1]) == hipSuccess) { ms_total[c->ev_stage[i / 2]] += ms; launches[c->ev_stage[i / 2]] += 1; }
    }
    c->ev_n = 0;
    return OFK_OK;
}
