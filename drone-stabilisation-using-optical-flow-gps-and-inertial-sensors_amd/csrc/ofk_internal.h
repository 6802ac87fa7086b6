// ofk_internal.h — context layout and kernel-launcher prototypes shared by the libofk.so sources.
// gfx950 only (wave64, 160 KiB LDS/CU, 256 CUs in 8 XCDs).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "ofk.h"

#define OFK_MAX_LEVELS 9            // level 0 .. 8
#define OFK_MAX_STREAMS 8           // slices of a batch that run their stage chains concurrently
#define OFK_CHUNK 4096              // candidates sorted per selection round (LDS resident)
#define OFK_SEG_MAX 2048            // segments (strips) per image the selection kernel walks
#define OFK_MAX_STRIDE 128          // ints between per-image response maxima: one L2 channel each (a packed array was a hot spot)
#define OFK_CNT_STRIDE 32           // ints between per-image candidate counters (one 128-B line each: no atomic contention)

struct ofk_levels {
    int n;                          // deepest level index in use (0..8)
    int h[OFK_MAX_LEVELS], w[OFK_MAX_LEVELS];
    size_t off[OFK_MAX_LEVELS];     // byte offset of level l inside one image's pyramid slab (256-B aligned)
};

struct ofk_comm;
struct ofk_ctx {
    int device;
    ofk_comm *comm;                                       // RCCL communicator + gather buffers (ofk_comm.hip), NULL until ofk_comm_init
    hipStream_t stream;                           // the context's stream (slice 0); entry points synchronise on it
    hipStream_t streams[OFK_MAX_STREAMS]; int nstreams;   // extra slice streams (created on first use), joined back into `stream` by events
    hipEvent_t ev_fork;
    hipStream_t aux[OFK_MAX_STREAMS]; int overlap;        // per-slice auxiliary stream: next-frame gray + pyramids beside the response kernel
    hipEvent_t ev_g0[OFK_MAX_STREAMS], ev_aux[OFK_MAX_STREAMS];
    uint8_t *pyr_alt[2]; int pyr_set;                     // second pyramid set: the auxiliary stream runs one call ahead
    int pyr_last;                                         // the set the latest ofk_pairs_run built its pyramids in (ofk_resident_pyramid)
    hipEvent_t ev_lkdone[2][OFK_MAX_STREAMS];             // LK of the call that last read a set has finished
    hipEvent_t marks[8];                                  // ofk_mark / ofk_mark_wait
    // Slices free-run across consecutive ofk_pairs_run calls (no fork/join per call) and are offset by one response kernel, so
    // one slice's latency-bound stages (selection, solve) always run beside another slice's response kernel or LK.
    int slices_open, open_slices;                         // slice streams hold work the context's stream has not joined yet / how many
    hipEvent_t ev_end[OFK_MAX_STREAMS];                   // end of a slice's chain of the latest call (slice 0 included)
    hipEvent_t ev_stagger[OFK_MAX_STREAMS];               // response kernel of slice k launched (first call after a join)
    hipEvent_t ev_x; int x_pending;                       // record export queued behind the last slice, not yet seen by `stream`
    int max_w, max_h, max_batch, max_pts, max_level;
    size_t P;                       // max_w * max_h
    size_t bgr_stride;              // bytes between images in bgr[], 256-B aligned
    size_t pyr_stride;              // bytes between images in pyr[], 256-B aligned (all levels of one image)
    size_t img_stride;              // elements between images in eig/mask (P rounded up to 64)
    int cand_cap;                   // candidate keys per image

    uint8_t *bgr[2];                // [B][bgr_stride]            prev / next BGR frames
    int gray_direct_set;            // >= 0: the resident frame pairs exist as gray level 0 of this pyramid set only (compressed ingest,
                                    // ofk_pairs_upload_staged): ofk_pairs_run takes that set and skips its BGR -> gray conversions; -1: bgr[] holds them
    uint8_t *pyr[2];                // [B][pyr_stride]            gray pyramids (level 0 = gray frame)
    float *eig;                     // [B][img_stride]            Shi-Tomasi response
    uint8_t *mask;                  // [B][img_stride]            optional detection mask (lazily allocated)
    int16_t *deriv;                 // [B][img_stride][2]         only for ofk_scharr_s16 (lazily allocated)
    unsigned long long *cand;       // [B][cand_cap]              candidate keys (~value bits << 32 | linear index), flat list
    unsigned long long *cand_seg;   // [B][seg_keys]              per-strip key segments written by the streaming response kernel
    int *seg_count;                 // [B][OFK_SEG_MAX]           keys per segment
    size_t seg_keys;                // keys per image in cand_seg
    int *cand_count;                // [B][OFK_CNT_STRIDE]
    unsigned *sel_hist;             // [B][1024]                  key histogram of k_select_prep
    unsigned long long *sel_keys;   // [B][OFK_CHUNK]             keys below the first cut, picked by k_select_pick for k_select_greedy
    unsigned int *maxbits;          // [B][OFK_MAX_STRIDE]                       bit pattern of max positive response
    float *pts_prev, *pts_next;     // [B][max_pts][2]
    uint8_t *status;                // [B][max_pts]
    float *err;                     // [B][max_pts]
    int *counts;                    // [B]                        corners per image
    float *pts_new;                 // [B][max_pts][2]            re-detected corners of a stream step (lazily allocated)
    int *new_counts, *limit;        // [B]                        their counts / per-stream re-detection budget
    int stream_h, stream_w, stream_batch;         // geometry of the active video streams (ofk_stream_begin), 0 = none
    int *h_counts;                                // host copy of the streams' track counts after the last call
    double *sensors;                // [B][OFK_SENSOR_DOUBLES]
    double *records;                // [B][OFK_RECORD_DOUBLES]
    int *dev_flags;                 // [4]                        device-side error flags (bit 0: candidate overflow)
    void *scratch; size_t scratch_bytes;          // device scratch for the estimation entry points
    void *hstage; size_t hstage_bytes;            // pinned host staging
    void *jstage;                                 // JPEG staging slots + copy stream (k_jpeg.hip), NULL until the first compressed frame

    // per-stream filters resident on the device (ofk_imu_*, ofk_filter_*, ofk_stream_step_fused); lazily allocated
    double *imu_state, *imu_dv;     // [B][OFK_IMU_STATE], [B][3]
    double *kf_mats, *kf_x, *kf_P;  // 5 x 36 doubles (F, B, H, Q, R), [B][6], [B][36]
    double *fused;                  // [B][8]
    double *imu_msgs; int *imu_counts; size_t imu_msgs_bytes;     // upload staging for ofk_imu_push
    int kf_ns, kf_nm, kf_nc;
    int cur_batch, cur_h, cur_w;    // resident pair geometry (ofk_pairs_upload)
    int prof_mask;
    hipEvent_t *ev; int ev_cap, ev_n; int *ev_stage;   // pairs of events: start/stop
    char errmsg[512];
};

// Launch-geometry knobs of ofk_set_tuning (ofk.h): process-wide, 0 = the built-in choice.  Results never depend on them.
struct ofk_tuning { int eig_rows, no_pair, no_pyr3, pyr3_chunks, pyr_rows, jpeg_chunk, gray_px, jpeg_sub; };
extern ofk_tuning g_ofk_tuning;

// --- helpers (host)
int ofk_fail(ofk_ctx *ctx, int code, const char *fmt, ...);
#define OFK_HIP(ctx, call)                                                                       \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) return ofk_fail(ctx, OFK_E_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
    } while (0)
ofk_levels ofk_make_levels(int h, int w, int win, int max_level);   // win <= 0: ignore the winSize stop rule
int ofk_need_scratch(ofk_ctx *ctx, size_t bytes);
int ofk_join_slices(ofk_ctx *ctx);
int ofk_prepare_streams(ofk_ctx *ctx);                                 // create the slice / auxiliary streams of the current schedule
int ofk_export_records_stream(ofk_ctx *c, float *device_dst, int batch, hipStream_t *stream_out);   // k_records_f32 on the stream that ends the step
                                   // before touching the context's stream / shared buffers
void ofk_jpeg_release(ofk_ctx *c);
int ofk_jpeg_stage_streams(ofk_ctx *c, int slot, const uint8_t *const *jpeg, const size_t *nbytes, int count);
const char *ofk_jpeg_slot_error(const ofk_ctx *c, int slot);       // the slot's own message: staging may run beside the owner thread
int ofk_jpeg_decode_staged_pairs(ofk_ctx *c, int slot, uint8_t *dst_prev, uint8_t *dst_next, size_t dst_stride, size_t dst_capacity_px, int *batch_out,
                                 int *h_out, int *w_out, const hipEvent_t *wait_before_writing, int nwait, int as_gray);
                                 // wait_before_writing != NULL: decode on the ingest stream, beside whatever runs on the context's, behind those events;
                                 // as_gray: dst_* are gray planes (rows of w bytes; what k_gray_bgr8 would make of the decoded BGR frame), no BGR is written
int ofk_jpeg_decode_device(ofk_ctx *c, const uint8_t *const *jpeg, const size_t *nbytes, int batch, uint8_t *dst, size_t dst_stride,
                           size_t dst_capacity_px, int *h_out, int *w_out, uint8_t **out, size_t *out_stride);   // k_jpeg.hip

// --- launchers (all asynchronous on `s`; pointers are device pointers)
void ofk_launch_gray(hipStream_t s, const uint8_t *bgr, size_t bgr_stride, uint8_t *gray, size_t gray_stride, int batch,
                     int h, int w);
void ofk_launch_pyr_down(hipStream_t s, const uint8_t *src, size_t src_stride, int h, int w, uint8_t *dst,
                         size_t dst_stride, int batch);
void ofk_launch_pyr_down2(hipStream_t s, const uint8_t *src0, const uint8_t *src1, size_t src_stride, int h, int w,
                          uint8_t *dst0, uint8_t *dst1, size_t dst_stride, int batch);
bool ofk_launch_pyr3(hipStream_t s, uint8_t *pyr0, uint8_t *pyr1, size_t stride, const ofk_levels &lv, int batch, int images);
void ofk_launch_scharr(hipStream_t s, const uint8_t *src, size_t src_stride, int h, int w, int16_t *dxdy,
                       size_t dst_stride_elems, int batch);
int  ofk_launch_mineig(hipStream_t s, const uint8_t *gray, size_t gray_stride, int h, int w, int block, float *eig,
                       size_t eig_stride, unsigned int *maxbits, const uint8_t *mask, size_t mask_stride, int batch);
void ofk_launch_maxbits(hipStream_t s, const float *eig, size_t eig_stride, const uint8_t *mask, size_t mask_stride,
                        int h, int w, unsigned int *maxbits, int batch);
void ofk_launch_nms(hipStream_t s, const float *eig, size_t eig_stride, const uint8_t *mask, size_t mask_stride, int h,
                    int w, const unsigned int *maxbits, double quality, unsigned long long *cand, int cand_cap,
                    int *cand_count, int *flags, int batch);
int  ofk_launch_mineig_cand(hipStream_t s, const uint8_t *gray, size_t gray_stride, int h, int w, int block,
                            unsigned int *maxbits, const uint8_t *mask, size_t mask_stride, double quality,
                            unsigned long long *cand, int cand_cap, int *cand_count, unsigned long long *seg,
                            size_t seg_keys_per_image, int *seg_count, int seg_count_cap, int *flags, int batch, int *nseg_out,
                            int *segcap_out);
void ofk_stream_geometry(int h, int w, int block, int batch, int *rows, int *nseg, int *seg_cap);
void ofk_launch_select(hipStream_t s, unsigned long long *cand, int cand_cap, int *cand_count, const unsigned long long *seg,
                       int seg_cap, const int *seg_count, int nseg, const unsigned int *maxbits, double quality, int h, int w,
                       int max_corners, float min_distance, float *pts, int pts_stride, int *counts, const int *limit, int batch,
                       unsigned *sel_hist, unsigned long long *sel_keys, bool hist_is_zero = false);   // nseg == 0: the flat list holds the candidates already
void ofk_launch_zero_detect_state(hipStream_t s, unsigned int *maxbits, int *cand_count, unsigned *sel_hist, int batch);                     // [batch][1024] scratch of the multi-workgroup prefilter (NULL: single-workgroup path)
void ofk_launch_disc_mask(hipStream_t s, uint8_t *mask, size_t mask_stride, int h, int w, const float *pts, const int *counts,
                          int pts_stride, int radius, const int *limit, int batch);
void ofk_launch_redetect_limits(hipStream_t s, const int *counts, int min_feat, int max_feat, int *limit, int batch);
void ofk_launch_update_tracks(hipStream_t s, const float *next_pts, const uint8_t *status, const int *counts_in, int pts_stride,
                              const float *new_pts, const int *new_counts, float *tracks, int *counts_out, int max_total, int batch);
void ofk_launch_lk(hipStream_t s, const uint8_t *prev, const uint8_t *next, size_t pyr_stride, const ofk_levels &lv,
                   const float *prev_pts, const int *counts, int pts_stride, int win, int max_count, double eps,
                   double min_eig_thr, float *next_pts, uint8_t *status, float *err, int batch);
void ofk_launch_pairs_solve(hipStream_t s, const float *prev_pts, const float *next_pts, const uint8_t *status,
                            const int *counts, int pts_stride, const double *sensors, int variant, int use_feas,
                            double feas_T, const int *cand_count, double *records, int batch);
void ofk_launch_records_f32(hipStream_t s, const double *records, float *dst, int batch);

void ofk_launch_flow_model(hipStream_t s, const double *x, int batch, int n, const double *v, const double *omega,
                           const double *d, const double *nrm, const double *t, double *flow);
void ofk_launch_feasibility(hipStream_t s, int variant, const double *x, const double *u, int batch, int n,
                            const double *nrm, const double *v, const double *dist, const double *omega,
                            const double *t, double *r, double *dd);
void ofk_launch_solve(hipStream_t s, int variant, const double *x, const double *u, const uint8_t *valid, int batch,
                      int n, const double *d, const double *nrm, const double *omega, const double *t,
                      const double *wgt, double *out);
void ofk_launch_imu(hipStream_t s, double *state, const double *msg, int batch);
void ofk_launch_kf_records(hipStream_t s, int ns, int nm, const double *mats, double *x, double *P, const double *records, double z_sign,
                           int z_source, int batch);
void ofk_launch_imu_seq(hipStream_t s, double *state, double *dv, const double *msgs, const int *counts, int max_msgs, int batch);
void ofk_launch_stream_fuse(hipStream_t s, const float *prev_pts, const float *next_pts, uint8_t *status, const int *counts, int pts_stride,
                            const double *sensors, double *imu_state, double *imu_dv, int ns, int nm, int nc, const double *kf_mats,
                            double *kf_x, double *kf_P, const ofk_fusion *f, int variant, int use_feas, double feas_T, double *records,
                            double *fused, int batch);
void ofk_launch_replace_tracks(hipStream_t s, const int *limit, const float *new_pts, const int *new_counts, int pts_stride, float *tracks,
                               int *counts, int batch);
void ofk_launch_post_solve(hipStream_t s, const double *v_obs, const double *rot, const double *ang,
                           const double *offset, int batch, double *v_uav);
void ofk_launch_kf(hipStream_t s, int ns, int nm, int nc, const double *F, const double *Bm, const double *H,
                   const double *Q, const double *Rm, double *x, double *P, const double *u, const double *z, int batch,
                   int do_predict);
void ofk_launch_of_simulation(hipStream_t s, const double *truth, const double *sig, const double *pos,
                              const double *true_flow, int n, const double *z, int trials, double *v_obs,
                              double *bound);
void ofk_launch_of_simulation_rng(hipStream_t s, const double *truth, const double *sig, const double *pos, const double *true_flow, int n,
                                  unsigned long long seed, unsigned step, unsigned trial0, int trials, double *v_obs, double *bound);
void ofk_launch_noise_normals(hipStream_t s, unsigned k0, unsigned k1, unsigned step, unsigned trial, int count, double *out);
void ofk_launch_feas_simulation(hipStream_t s, const double *truth, const double *sig, const double *pos, const double *true_flow,
                                int n, const double *z, int trials, double *per, double *mean, double *v_obs);
void ofk_launch_hist_overlap(hipStream_t s, const double *d1, int n1, const double *d2, int n2, int bins, int *out);
void ofk_launch_feature_eval(hipStream_t s, const double *pos, const double *pos_err, const double *oldpos, const double *oldpos_err,
                             const int *counts, int batch, int stride, const double *vel, const double *vel_err, double focal,
                             double dummy, double tx, double ty, const double *weight, double *height, double *height_err,
                             uint8_t *immobile, double *score, int *order, int *flags);
void ofk_launch_d_split(hipStream_t s, const double *d, const int *counts, int batch, int stride, double d_exp_err, double *sorted,
                        double *diff, int *nsplit);
void ofk_launch_associate(hipStream_t s, const double *t_img, int n_img, int n_imu, const double *imu_t, const double *imu_q,
                          const double *imu_w, int n_hgt, const double *hgt_t, const double *hgt_r, double *sensors, int *imu_idx,
                          int *hgt_idx);
