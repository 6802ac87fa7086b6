/*
 * include/ofk.h — C ABI of libofk.so, the MI355X (gfx950) implementation of the
 * optical-flow -> ego-velocity hot path of
 * liquidcronos/Drone-stabilisation-using-Optical-Flow-Gps-and-Inertial-Sensors.
 *
 * The reference has no FFI for this path: it is Python calling cv2 / numpy.  The entry
 * points below are therefore what a ctypes binding in the reference's own files would
 * call in place of each cv2 / numpy call (file:line of the call being replaced is given
 * per function; paths relative to the reference root; "node" = velocity_measurment_node).
 * INTEGRATION.md shows the binding.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++ or torch types, no exceptions.
 *   - Every function returns 0 on success or a negative OFK_E_* code; the text of the
 *     last error is available from ofk_last_error().
 *   - Image buffers are C-contiguous: gray [batch][h][w] u8, BGR [batch][h][w][3] u8,
 *     points [batch][stride][2] f32 as (x = column, y = row) — the layout of OpenCV's
 *     (N,1,2) float32 arrays.
 *   - Unless a parameter is documented as a DEVICE pointer, buffers are caller-owned
 *     HOST memory; the library copies to/from device memory it owns inside the context.
 *   - One context = one device + one HIP stream.  Not thread-safe: the caller serialises
 *     calls on a context (the Python facade holds a lock).
 *   - There is no CPU fallback: every entry point runs HIP kernels or fails.
 */
#ifndef OFK_H
#define OFK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OFK_VERSION 100          /* 0.1.0 */

#define OFK_OK            0
#define OFK_E_INVALID    -1      /* bad argument / exceeds what the context was created for */
#define OFK_E_HIP        -2      /* a HIP runtime call failed (see ofk_last_error) */
#define OFK_E_CAPACITY   -3      /* a device-side list overflowed its capacity */
#define OFK_E_NOGPU      -4      /* no usable gfx950 device */

typedef struct ofk_ctx ofk_ctx;

/* ---------------------------------------------------------------- lifecycle */
int         ofk_version(void);
const char *ofk_last_error(const ofk_ctx *ctx);      /* ctx may be NULL: error of the last failed ofk_create */
int         ofk_device_count(void);
/* max_w/max_h: largest frame; max_batch: frame pairs per call; max_pts: corners per frame;
 * max_level: deepest LK pyramid level (0..8). */
int         ofk_create(int device, int max_w, int max_h, int max_batch, int max_pts, int max_level, ofk_ctx **out);
int         ofk_destroy(ofk_ctx *ctx);
int         ofk_sync(ofk_ctx *ctx);                   /* hipStreamSynchronize on the context's stream */
int         ofk_device_sync(void);                    /* hipDeviceSynchronize */

/* ------------------------------------------------- stage entry points (host buffers, synchronous) */

/* cv2.cvtColor(img, cv2.COLOR_BGR2GRAY) — of_module.py:40,80; node:113; evaluate_exp.py:65,85; of_library.py:236,248.
 * Y = (3735 B + 19235 G + 9798 R + 16384) >> 15. */
int ofk_gray_bgr8(ofk_ctx *ctx, const uint8_t *bgr, int batch, int h, int w, uint8_t *gray);

/* One pyrDown step of the pyramid cv2.calcOpticalFlowPyrLK builds (of_module.py:88; node:133; evaluate_exp.py:98):
 * separable [1 4 6 4 1]/16, REFLECT_101, dst = ((h+1)/2, (w+1)/2). */
int ofk_pyr_down_u8(ofk_ctx *ctx, const uint8_t *src, int batch, int h, int w, uint8_t *dst);

/* Gaussian pyramid (cv2.buildOpticalFlowPyramid without the derivative planes; what calcOpticalFlowPyrLK builds internally,
 * of_module.py:88): levels 1..max_level of every image by repeated pyrDown, stopping early when a level would be empty.
 * out: per image the levels back to back, tightly packed ((h+1)/2 x (w+1)/2, ...); *levels_built = number of levels written. */
int ofk_pyramid_u8(ofk_ctx *ctx, const uint8_t *gray, int batch, int h, int w, int max_level, uint8_t *out, int *levels_built);

/* Scharr derivatives of one pyramid level as calcOpticalFlowPyrLK computes them (same call sites):
 * dxdy [batch][h][w][2] int16 = (dx, dy), REFLECT_101 at the image border. */
int ofk_scharr_s16(ofk_ctx *ctx, const uint8_t *gray, int batch, int h, int w, int16_t *dxdy);

/* Shi-Tomasi response inside cv2.goodFeaturesToTrack (of_module.py:44,86; node:120,163; evaluate_exp.py:66,106;
 * of_library.py:238): Sobel-3 -> structure tensor box(block_size) -> min eigenvalue, f32 map [batch][h][w].
 * block_size 1..45. */
int ofk_mineig_response(ofk_ctx *ctx, const uint8_t *gray, int batch, int h, int w, int block_size, float *eig);

/* Selection half of cv2.goodFeaturesToTrack: > quality*max, 3x3 local max, sort (value desc, linear index desc: OpenCV's greaterThanPtr),
 * greedy min-distance, top max_corners.  mask (nullable) [batch][h][w] u8, 0 = excluded.
 * pts [batch][max_corners][2] f32, counts [batch]. max_corners in 1..ctx max_pts. */
int ofk_select_corners(ofk_ctx *ctx, const float *eig, const uint8_t *mask, int batch, int h, int w, int max_corners,
                       double quality, double min_distance, float *pts, int *counts);

/* cv2.goodFeaturesToTrack(gray, mask=mask, maxCorners, qualityLevel, minDistance, blockSize) — same call sites. */
int ofk_good_features(ofk_ctx *ctx, const uint8_t *gray, const uint8_t *mask, int batch, int h, int w, int max_corners,
                      double quality, double min_distance, int block_size, float *pts, int *counts);

/* cv2.calcOpticalFlowPyrLK(prev, next, prevPts, None, winSize=(win,win), maxLevel, criteria=(EPS|COUNT, max_count, eps))
 * — of_module.py:88; node:133; evaluate_exp.py:98; of_library.py:249.
 * prev_pts/next_pts [batch][pts_stride][2] f32, counts [batch] (points used per image, <= pts_stride),
 * status [batch][pts_stride] u8, err [batch][pts_stride] f32.  win odd, 3..31. */
int ofk_lk_pyr(ofk_ctx *ctx, const uint8_t *prev, const uint8_t *next, int batch, int h, int w, const float *prev_pts,
               const int *counts, int pts_stride, int win, int max_level, int max_count, double eps, double min_eig_thr,
               float *next_pts, uint8_t *status, float *err);

/* ------------------------------------------------- estimation (float64, batched over `batch` independent problems) */

/* generate_test_data(x, v, omega, d, n[, t]) — node:25-29; simulation.py:7-12.
 * x [batch][n][2]; v, omega, nrm, t [batch][3] (t nullable = no lever arm); d [batch]; flow [batch][n][2]. */
int ofk_flow_model(ofk_ctx *ctx, const double *x, int batch, int n, const double *v, const double *omega, const double *d,
                   const double *nrm, const double *t, double *flow);

#define OFK_FEAS_RTILDE  0   /* of.r_tilde(x,u,n,v,dist)            of_library.py:365-386 (node:238) */
#define OFK_FEAS_LEGACY  1   /* 4-arg r_tilde, no guard, no /dist    sensor_precision_experiments/pixhawk_pure_IMU/of_library.py:365-380 (of_module.py:125) */
#define OFK_FEAS_SIM     2   /* feasibility(pos,v,flow,omega,t,n)    simulation.py:108-120 */
/* x,u [batch][n][2]; nrm,v [batch][3]; dist [batch] (RTILDE only); omega,t [batch][3] (SIM only, else nullable);
 * r,dd [batch][n]. */
int ofk_feasibility(ofk_ctx *ctx, int variant, const double *x, const double *u, int batch, int n, const double *nrm,
                    const double *v, const double *dist, const double *omega, const double *t, double *r, double *dd);

#define OFK_SOLVE_NODE      0   /* node:30-42 / evaluate_exp.py:18-31: A_i=[p]x,        b_i=[p]x(u+[p]x w)/(n.p) */
#define OFK_SOLVE_SIM       1   /* simulation.py:15-30:               A_i=[p]x (n.p),  b_i=[p]x(u+[p]x w)        */
#define OFK_SOLVE_OFMODULE  2   /* of_module.py:139-146:              A_i=[p]x/w_i,    b_i=A_i u_i/(n.p), no omega, no d */
#define OFK_SOLVE_DOUBLES   8   /* out per problem: v[3], residual SS, rank, s[3] (singular values, descending) */
/* x,u [batch][n][2]; valid (nullable) [batch][n] u8, 0 = skip the point; d [batch]; nrm, omega [batch][3];
 * t (nullable) [batch][3]: subtract omega x t from v (simulation.py:28, evaluate_exp.py:29);
 * wgt (OFMODULE only) [batch][n] per-point distance.  Fewer than 1 valid point -> rank 0, v = 0. */
int ofk_velocity_solve(ofk_ctx *ctx, int variant, const double *x, const double *u, const uint8_t *valid, int batch, int n,
                       const double *d, const double *nrm, const double *omega, const double *t, const double *wgt,
                       double *out);

/* optical_fusion.call_imu — node:61-89, batched over independent IMU streams, one message each.
 * state [batch][OFK_IMU_STATE]: vel[3], old_time, time_zero, first(0/1), rotation[9], normal[3], ang[3], ang_err[3]
 * msg   [batch][OFK_IMU_MSG]  : secs, nsecs, qx,qy,qz,qw, wx,wy,wz, cov0,cov4,cov8, ax,ay,az */
#define OFK_IMU_STATE 24
#define OFK_IMU_MSG   15
int ofk_imu_propagate(ofk_ctx *ctx, double *state, const double *msg, int batch);

/* of_library.py:270-286 (calc_height), :53-75 + :100-114 (convert_to_of inside dynamic_immobile), :291-317 (eval_ft) — the
 * per-feature estimators around the path, for a batch of track sets.  Per set b with counts[b] features (arrays are
 * [batch][stride]...): observed flow = pos - oldpos; height and its variance from the pin-hole model with the set's velocity
 * vel[b] +- vel_err[b]; immobile[i] = (observed - expected flow)^2 < var(observed) + var(expected) on both axes and neither old
 * coordinate equals dummy_value; score = w0 (1 - height_n) + w1 height_err_n + w2 (1 - centre_dist_n) + w3 pos_err_n with the
 * min/max normalisations of eval_ft; order = feature indices by ascending score (ties by index, NaN last; -1 past the count).
 * *bad_height is set non-zero when a height is not positive (where the reference raises ValueError).  pos, oldpos
 * [batch][stride][2] in pixels; pos_err, oldpos_err [batch][stride]; vel, vel_err [batch][3]; weight [4]. */
int ofk_feature_eval(ofk_ctx *ctx, const double *pos, const double *pos_err, const double *oldpos, const double *oldpos_err,
                     const int *counts, int batch, int stride, const double *vel, const double *vel_err, double focal_len,
                     double dummy_value, int img_w, int img_h, const double *weight, double *height, double *height_err,
                     uint8_t *immobile, double *score, int *order, int *bad_height);

/* velocity_measurment_node:249-252 — statistics of the per-feature plane distances d_i that r_tilde returns (ofk_feasibility):
 * sorted[b] = np.sort(d[b][:counts[b]]), diff[b][i] = sorted[i+1] - sorted[i], nsplit[b] = number of gaps >= d_exp_err (the
 * split the node's commented line describes: several ground planes in view).  Arrays [batch][stride], stride <= 4096. */
int ofk_d_split(ofk_ctx *ctx, const double *d, const int *counts, int batch, int stride, double d_exp_err, double *sorted,
                double *diff, int *nsplit);

/* evaluate_exp.py:68-95 — sensor association for replayed logs: for every image time t_img[k] (seconds, as the script forms
 * them: float(secs - secs0) + float(nsecs)/1e9) the nearest IMU and range samples (np.argmin(np.abs(values - t)): the first
 * minimum), then d = range, R from the IMU quaternion (x,y,z,w), normal = R e_z, omega = angular velocity.  Fills fields
 * 0-15 of sensors[k] (see OFK_SENSOR_DOUBLES below; the other fields keep their values) and, when not NULL, the chosen
 * indices.  imu_quat [n_imu][4], imu_omega [n_imu][3]. */
int ofk_associate_sensors(ofk_ctx *ctx, const double *t_img, int n_img, const double *imu_t, const double *imu_quat,
                          const double *imu_omega, int n_imu, const double *hgt_t, const double *hgt_range, int n_hgt,
                          double *sensors, int *imu_index, int *hgt_index);

/* node:258 — v_uav = R (v_obs - [w]x offset).  v_obs, ang, offset [batch][3]; rotation [batch][9]; v_uav [batch][3]. */
int ofk_post_solve(ofk_ctx *ctx, const double *v_obs, const double *rotation, const double *ang, const double *offset,
                   int batch, double *v_uav);

/* cv2.KalmanFilter(ns, nm, nc).predict(control) then .correct(measurement) — of_module.py:63-76,122,152
 * (the reference uses ns=nm=nc=3 with F=B=H=I; ns<=6, nm<=6, nc<=6 supported, row-major matrices shared by the batch).
 * x [batch][ns], P [batch][ns][ns] are updated in place; B/u nullable (no control); z nullable (predict only);
 * do_predict = 0 skips the predict step. */
int ofk_kf_predict_update(ofk_ctx *ctx, int ns, int nm, int nc, const double *F, const double *Bm, const double *H,
                          const double *Q, const double *Rm, double *x, double *P, const double *u, const double *z,
                          int batch, int do_predict);

/* of_simulation(...) — simulation.py:36-66 with the np.random.normal draws supplied by the caller:
 * z [trials][10+4n] standard normals in the reference's draw order (omega 3, t 3, height 1, flow 2n, position 2n,
 * normal 3 — drawn and discarded, simulation.py:45-46).  truth = v[3], omega[3], height, normal[3], t[3] (13);
 * sig = ang_vel, translation, height, flow, position, normal (6).  pos, true_flow [n][2].
 * v_obs [trials][3], bound [trials] (analytic error bound, simulation.py:56-64). */
int ofk_of_simulation(ofk_ctx *ctx, const double *truth, const double *sig, const double *pos, const double *true_flow,
                      int n, const double *z, int trials, double *v_obs, double *bound);
/* The same with the normals drawn ON THE DEVICE by a counter-based generator, for the 4096-wide batches of the Monte-Carlo sweeps
 * (simulation.py:183-461 at BASELINE configs[4]'s size: 2000 points x 4096 trials = 262 MB of host noise per step otherwise):
 * element e of the row of trial t of sweep step `step` = output (e & 1) of the Box-Muller transform of
 * Philox4x32-10(counter (e >> 1, t, step, 0), key (seed & 0xffffffff, seed >> 32)); uniforms ((x0 >> 5) 2^26 + (x1 >> 6) + 0.5) 2^-53.
 * Trials trial0 .. trial0 + trials - 1 (GLOBAL indices: ranks that shard the trials produce the same rows as one rank would).
 * ofk_noise_normals returns elements 0 .. count - 1 of one row - the generator itself (oracle: estimation_oracle.noise_normals). */
int ofk_of_simulation_rng(ofk_ctx *ctx, const double *truth, const double *sig, const double *pos, const double *true_flow, int n,
                          unsigned long long seed, unsigned step, unsigned trial0, int trials, double *v_obs, double *bound);
int ofk_noise_normals(ofk_ctx *ctx, unsigned long long seed, unsigned step, unsigned trial, int count, double *out);

/* feas_simulation(...) - simulation.py:70-104 (driven by the live experiment simulation.py:753-812: three ground planes, the
 * second one with randomly rotated flow), with the np.random.normal draws supplied by the caller:
 * z [trials][12+4n] standard normals in the reference's draw order (omega 3, t 3, height 1, flow 2n, position 2n,
 * velocity 3, orient 1, orient2 1).  truth = v[3] (unused by the reference's function body), omega[3], height, normal[3],
 * t[3], true_vel[3] (16); sig = ang_vel, translation, height, flow, position, normal, velocity (7; the reference reads the
 * last two from module globals).  pos, true_flow [n][2].  Per trial: perturbed inputs -> solve_lgs -> feasibility with the
 * solved velocity ("backward") and with the noisy prior velocity ("forward") -> per-point residual norms of both.
 * mean [6][n] = np.mean over the trials in the reference's return order: backward_para, backward_dist, forward_para,
 * forward_dist, backward_res, forward_res; per_trial (nullable) [trials][6][n]; v_obs (nullable) [trials][3]. */
#define OFK_FEAS_SIM_TRUTH 16
#define OFK_FEAS_SIM_SIG    7
int ofk_feas_simulation(ofk_ctx *ctx, const double *truth, const double *sig, const double *pos, const double *true_flow, int n,
                        const double *z, int trials, double *mean, double *per_trial, double *v_obs);

/* overlap(data1, data2) - simulation.py:124-136: histogram both samples over the `bins` (reference: 100, at most 1024) equal
 * bins spanning their joint range (np.histogram's edges and bin rule) and sum the bin-wise minima. */
int ofk_hist_overlap(ofk_ctx *ctx, const double *data1, int n1, const double *data2, int n2, int bins, int *overlap);

/* ------------------------------------------------- resident frame-pair pipeline (the benchmarked path) */

typedef struct ofk_params {
    int    max_corners;      /* goodFeaturesToTrack maxCorners (<= ctx max_pts) */
    double quality;          /* qualityLevel */
    double min_distance;     /* minDistance */
    int    block_size;       /* blockSize */
    int    win;              /* LK winSize (square) */
    int    max_level;        /* LK maxLevel */
    int    max_count;        /* LK criteria COUNT */
    double eps;              /* LK criteria EPS */
    double min_eig_thr;      /* LK minEigThreshold (1e-4) */
    int    solve_variant;    /* OFK_SOLVE_NODE / OFK_SOLVE_SIM (ofk_pairs_run, ofk_stream_step); also OFK_SOLVE_OFMODULE in ofk_stream_step_fused */
    int    use_feasibility;  /* 1: keep points with r_tilde <= feas_T (node:238-245) using sensors' prior velocity */
    double feas_T;
} ofk_params;

/* Per-pair sensor record, [batch][OFK_SENSOR_DOUBLES] doubles:
 * 0 d (plane distance)  1-3 normal  4-6 omega  7-15 rotation (row-major)  16-18 offset (lever arm, node:204)
 * 19 scaling (node:182)  20 cx  21 cy (of.pix_trans, node:229)  22-24 prior velocity (feasibility; second measurement of a
 * 6-row filter)  25-27 filter control input (ofk_stream_step_fused, OFK_CONTROL_SENSORS) */
#define OFK_SENSOR_DOUBLES 28
/* Per-pair result record, [batch][OFK_RECORD_DOUBLES] doubles:
 * 0-2 v_obs  3 residual SS  4 rank  5-7 singular values  8-10 v_uav (node:258)  11 points used in the solve
 * 12 corners detected  13 points tracked (status==1)  14 corner candidates (after threshold + NMS)  15 reserved */
#define OFK_RECORD_DOUBLES 16

/* Copies a batch of BGR frame pairs into the context's device buffers (host -> HBM). */
int ofk_pairs_upload(ofk_ctx *ctx, const uint8_t *prev_bgr, const uint8_t *next_bgr, int batch, int h, int w);
/* The same from compressed frames: jpeg[i] / nbytes[i] = one baseline JPEG stream per frame (the payload of a
 * sensor_msgs/CompressedImage; the reference decodes it with cv_bridge.compressed_imgmsg_to_cv2 = cv::imdecode,
 * velocity_measurment_node.py:112).  All frames of a call must share size and chroma sampling.  Decoded on the device; pixels
 * identical to libjpeg's default decompressor (see ofk_jpeg_decode_bgr8).  On the default schedule (one slice, overlap on) the
 * decoder's colour kernel applies the pipeline's BGR -> gray conversion (node:113 cv2.cvtColor) to every pixel itself and writes the
 * gray frame straight into the pyramid set the next ofk_pairs_run takes: the BGR frame is never stored and that run skips its
 * first stage - same bytes in the gray level as after ofk_pairs_upload of the decoded frames. */
int ofk_pairs_upload_jpeg(ofk_ctx *ctx, const uint8_t *const *prev_jpeg, const size_t *prev_bytes, const uint8_t *const *next_jpeg,
                          const size_t *next_bytes, int batch);
/* The same in two phases, so that a camera loop can hide the host's share of the ingest (marker parse, staging copy, PCIe) behind
 * the GPU's work on the batch before:
 *   ofk_jpeg_stage(ctx, slot, jpeg, nbytes, count)   host + copy engine: parses `count` streams, packs tables and the entropy segments
 *       (without their byte stuffing, see ofk_jpeg_destuff) into pinned staging slot 0 or 1 and queues their asynchronous H2D copies
 *       on the context's copy stream as its worker threads finish their shares; returns when the host part is done.  It touches nothing but its slot, so it may run on a SECOND THREAD while the context's owner is inside any
 *       other entry point - as long as that is not the decode of the same slot.
 *   ofk_pairs_upload_staged(ctx, slot)               device: decodes the 2 B streams staged in `slot` - the B previous frames
 *       first, then the B next frames - into the resident frame-pair buffers (what ofk_pairs_upload_jpeg does after staging
 *       slot 0 itself).  A slot is decoded once.  On the default schedule the decoder runs on a stream of its own and writes the
 *       pyramid set the ofk_pairs_run in flight is not using, so it overlaps that run.
 * Loop: stage(0, batch 0); for k: { stage(k+1 & 1, batch k+1) on the helper thread; upload_staged(k & 1); ofk_pairs_run; }.
 * pipeline.FlowPipeline.run_jpeg_batches does exactly that; bench.py reports its rate as ingest_inclusive.jpeg_double_buffered. */
int ofk_jpeg_stage(ofk_ctx *ctx, int slot, const uint8_t *const *jpeg, const size_t *nbytes, int count);
/* Message of the slot's last ofk_jpeg_stage ("" after a success).  ofk_jpeg_stage never writes ofk_last_error: it may run on a helper
 * thread while the owner thread is inside another entry point, and the context holds ONE message. */
const char *ofk_jpeg_stage_error(const ofk_ctx *ctx, int slot);
int ofk_pairs_upload_staged(ofk_ctx *ctx, int slot);
int ofk_pairs_set_sensors(ofk_ctx *ctx, const double *sensors, int batch);
/* Runs gray -> pyramids -> corners -> LK -> centre/scale -> (feasibility) -> solve -> post-solve for every resident
 * pair.  Asynchronous on the context's stream; call ofk_sync / ofk_pairs_download to wait. */
int ofk_pairs_run(ofk_ctx *ctx, const ofk_params *p);
/* Any output pointer may be NULL.  records [batch][16] f64; prev_pts/next_pts [batch][max_corners][2] f32;
 * status [batch][max_corners] u8; err [batch][max_corners] f32; counts [batch]. */
int ofk_pairs_download(ofk_ctx *ctx, double *records, float *prev_pts, float *next_pts, uint8_t *status, float *err,
                       int *counts);
/* Writes the batch's velocity records as float32 [batch][8] = {vx,vy,vz,residual,n_used,s_min,rank,corners} to a
 * DEVICE pointer owned by the caller (the buffer an RCCL all_gather sends), asynchronously behind the latest ofk_pairs_run
 * (wait with ofk_mark + ofk_mark_wait, or ofk_sync). */
int ofk_pairs_export_records_f32(ofk_ctx *ctx, void *device_dst, int batch);

/* ------------------------------------------------- video streams: persistent tracks on the device (feature lifecycle)
 * velocity_measurment_node:92-177 with its commented-out blocks restored (of_module.py:78-167 and evaluate_exp.py:77-121
 * follow the same loop): `batch` independent streams advance one frame per call.
 *   begin : gray + pyramid of the first frames, goodFeaturesToTrack -> tracks                       (node:117-128, :120)
 *   step  : gray + pyramid of the new frames; LK from the tracks (node:133); velocity solve on the tracked points with
 *           x = new position, u = new - old (node:134-136, :229-258); tracks := new[status == 1];
 *           streams that had <= min_features tracks re-detect on the PREVIOUS frame with discs of mask_radius around
 *           the old positions masked out, maxCorners = p->max_corners - (old count), and append (node:157-166);
 *           the new frame becomes the previous one (node:175).
 * sensors as in ofk_pairs_set_sensors.  records [batch][16] as in ofk_pairs_download (slot 12 = tracks before the step,
 * 13 = tracked), tracks [batch][p->max_corners][2] f32 and counts [batch] = the tracks AFTER the step; any may be NULL. */
int ofk_stream_begin(ofk_ctx *ctx, const uint8_t *first_bgr, int batch, int h, int w, const ofk_params *p, float *tracks,
                     int *counts);
int ofk_stream_step(ofk_ctx *ctx, const uint8_t *next_bgr, const double *sensors, const ofk_params *p, int min_features,
                    int mask_radius, double *records, float *tracks, int *counts);
/* The same with the frames as the node receives them (node:112, 215): one baseline JPEG stream per camera (the payload of a
 * sensor_msgs/CompressedImage), decoded on the device into the stream's frame buffer - no decoded frame crosses PCIe.  Frame size
 * comes from the streams (all of one size and sampling; later frames must match the first). */
int ofk_stream_begin_jpeg(ofk_ctx *ctx, const uint8_t *const *jpeg, const size_t *nbytes, int batch, const ofk_params *p, float *tracks,
                          int *counts);
int ofk_stream_step_jpeg(ofk_ctx *ctx, const uint8_t *const *jpeg, const size_t *nbytes, const double *sensors, const ofk_params *p,
                         int min_features, int mask_radius, double *records, float *tracks, int *counts);

/* ---- the per-stream filters, resident on the device ("the only cross-pair state is the per-stream filter,
 * which stays on the GPU that owns the stream")
 * IMU dead-reckoning state (velocity_measurment_node:61-89), layout OFK_IMU_STATE as in ofk_imu_propagate:
 *   ofk_imu_reset   state of `batch` streams := state0 [OFK_IMU_STATE] (NULL: the node's initial values, node:182-217:
 *                   vel = 0.1, rotation = I, normal = e_z, first message pending)
 *   ofk_imu_push    the messages each stream received since its last frame, applied in order (k_imu_seq): msgs
 *                   [batch][max_msgs][OFK_IMU_MSG], counts [batch]; also accumulates the velocity increments (the filter's control)
 *   ofk_imu_state   download: state [batch][OFK_IMU_STATE], dv (nullable) [batch][3] = increments not yet consumed by a step
 * Kalman filter (cv2.KalmanFilter of of_module.py:63-76; ns, nm <= 6, nc <= 6; row-major matrices shared by the streams):
 *   ofk_filter_configure  matrices + every stream's state := x0 [ns], P0 [ns][ns]
 *   ofk_filter_state      download x [batch][ns], P [batch][ns][ns] */
int ofk_imu_reset(ofk_ctx *ctx, const double *state0, int batch);
int ofk_imu_push(ofk_ctx *ctx, const double *msgs, const int *counts, int max_msgs, int batch);
int ofk_imu_state(ofk_ctx *ctx, double *state, double *dv, int batch);
int ofk_filter_configure(ofk_ctx *ctx, int ns, int nm, int nc, const double *F, const double *Bm, const double *H, const double *Q,
                         const double *Rm, const double *x0, const double *P0, int batch);
int ofk_filter_state(ofk_ctx *ctx, double *x, double *P, int batch);

/* Per-pair filter update of the resident batch, queued behind the latest ofk_pairs_run (asynchronous): every pair's filter
 * (state from ofk_filter_configure, resident) predicts and corrects with z = z_sign * (z_source ? v_uav : v_obs) of its own record
 * when the solve had full rank - BASELINE configs[2] "batch of 1024 independent frame pairs + per-frame EKF update". */
int ofk_pairs_filter_step(ofk_ctx *ctx, double z_sign, int z_source, int batch);

/* What happens between calcOpticalFlowPyrLK and the next frame (ofk_stream_step_fused). */
#define OFK_FLOW_LK          0   /* u = new - old (node:235; of_module.py:108) */
#define OFK_FLOW_ROTATIONAL  1   /* of_module.py:113-114: the flow is overwritten by the rotational field of the sensors' omega */
#define OFK_KEEP_STATUS      0   /* status == 1, and r_tilde <= feas_T when p->use_feasibility (node:238-245) */
#define OFK_KEEP_LEGACY      1   /* legacy 4-arg r_tilde with the filter's predicted velocity, keep r - (uint8)(status - 1) >= feas_T: tracked points with r >= feas_T, a lost point's status-1 wraps to 255 (of_module.py:93,125-131) */
#define OFK_CONTROL_SENSORS  0   /* filter control = sensors[25..27] (of_module.py:122 draws it at random) */
#define OFK_CONTROL_IMU      1   /* filter control = velocity increments accumulated by ofk_imu_push since the last step */
typedef struct ofk_fusion {
    int    use_imu;          /* 1: normal, omega, rotation and the prior velocity come from the resident IMU state, not from `sensors` */
    int    flow;             /* OFK_FLOW_* */
    int    keep;             /* OFK_KEEP_*; the kept points become the stream's tracks (of_module.py:166) */
    int    filter;           /* 1: resident Kalman filter: predict(control) before the feasibility test, correct() after the solve */
    int    control;          /* OFK_CONTROL_* */
    double z_sign;           /* measurement = z_sign * velocity (of_module.py:152 corrects with -v_obs) */
    int    z_source;         /* 0: v_obs, 1: v_uav (lever arm + rotation applied, node:258) */
    int    vel_overwrite;    /* 1: the IMU state's velocity := v_uav after a solve (node:261) */
    int    redetect_replace; /* 1: streams with <= min_features tracks REPLACE them by maxCorners - count fresh corners of the previous
                                frame, no mask, before tracking (of_module.py:83-86); 0: append with a disc mask after tracking (node:157-166) */
    int    min_solve;        /* solve only with MORE than this many kept points (of_module.py:138: 3; node:256: 2) */
    int    hold_on_skip;     /* 1 (one stream per context only): a step that does not solve leaves the previous frame and the tracks as they
                                were, like the `continue` of of_module.py:138 - the next frame is tracked from the OLD one; the filter keeps
                                its prediction.  Costs one host wait per step.  0: the frame always advances (a batch shares one frame swap) */
} ofk_fusion;
/* ofk_stream_step with the filters in the loop: (redetect_replace) -> gray + pyramid of the new frames -> LK -> k_stream_fuse
 * (centre/scale, flow, filter predict, feasibility, solve with p->solve_variant incl. OFK_SOLVE_OFMODULE, lever arm + rotation,
 * filter correct, velocity overwrite) -> tracks := kept points -> (masked re-detection) -> frame swap.  records as in
 * ofk_stream_step (slot 15: 1 if the system was solved); fused [batch][8] = filter state x[0..5] (zero padded), trace(P), solved —
 * without a filter: v_uav (or the dead-reckoned velocity when nothing was solved).  of_module.py:138 `continue`s on <= 3 feasible points
 * WITHOUT advancing the frame: f->hold_on_skip = 1 reproduces that for a context with ONE stream (what the script is); a batch of
 * streams shares one frame swap, so there the frame always advances, the filter keeps its prediction and the record reports rank 0. */
int ofk_stream_step_fused(ofk_ctx *ctx, const uint8_t *next_bgr, const double *sensors, const ofk_params *p, const ofk_fusion *f,
                          int min_features, int mask_radius, double *records, double *fused, float *tracks, int *counts);
/* Next positions [batch][stride][2] and keep flags [batch][stride] of the latest step (valid until the next one): with the tracks
 * the caller held before the step they give the flow of the kept points, new - old (node:134-136). */
int ofk_stream_last_points(ofk_ctx *ctx, float *next_pts, uint8_t *keep, int stride);
int ofk_stream_step_fused_jpeg(ofk_ctx *ctx, const uint8_t *const *jpeg, const size_t *nbytes, const double *sensors, const ofk_params *p,
                               const ofk_fusion *f, int min_features, int mask_radius, double *records, double *fused, float *tracks,
                               int *counts);

/* ---- compressed-image ingest (cv2.imdecode of the reference's CompressedImage callback, velocity_measurment_node.py:112) ----
 * ofk_jpeg_info: header fields of a JPEG stream (host only; no context, no GPU).  OFK_E_INVALID if the stream is not one the decoder
 * accepts: 8-bit baseline / extended-sequential Huffman, one interleaved scan, gray or YCbCr 4:4:4 / 4:2:2 / 4:2:0, with or
 * without restart intervals.
 * ofk_jpeg_decode_bgr8: decodes `batch` streams of equal size and sampling on the device into bgr [batch][h][w][3] (host; gray
 * streams are replicated over the three channels like cv2.IMREAD_COLOR).  Bit-identical to libjpeg's default decompressor (ISLOW
 * IDCT, fancy upsampling) - what cv::imdecode returns.  Entropy decoding runs on the GPU too (self-synchronising chunked Huffman
 * decoders); truncated or corrupt entropy data is an error, not a partially grey picture. */
int ofk_jpeg_info(const uint8_t *jpeg, size_t nbytes, int *h, int *w, int *components);
/* ofk_jpeg_destuff (host only; no context, no GPU): the entropy-coded segment of the stream's scan as the device decoders read it -
 * byte stuffing removed (FF00 -> FF, what jdhuff.c's fill_bit_buffer does on the fly behind cv::imdecode), the RSTn markers of a
 * stream with a restart interval taken out and the offsets behind them written to rst[0 .. *nrst) (offsets into `out`), the data
 * ending at the first other marker.  *out_len = bytes written; OFK_E_INVALID if the stream is not decodable (ofk_jpeg_info), if
 * out_capacity is smaller than the stuffed segment or rst_capacity smaller than the number of markers.  The staging of the ingest
 * (ofk_jpeg_stage, ofk_pairs_upload_jpeg ...) runs the same routine; this entry exists so that it can be tested without a GPU. */
int ofk_jpeg_destuff(const uint8_t *jpeg, size_t nbytes, uint8_t *out, size_t out_capacity, size_t *out_len, uint32_t *rst, int rst_capacity, int *nrst);
int ofk_jpeg_decode_bgr8(ofk_ctx *ctx, const uint8_t *const *jpeg, const size_t *nbytes, int batch, uint8_t *bgr);

/* ------------------------------------------------- multi-GPU exchange: RCCL over xGMI, no PyTorch
 * The reference has no distributed code; frame pairs (and Monte-Carlo trials) are independent, so each rank (one process per
 * GPU) owns its own pairs and the only exchange is an all-gather of the per-pair velocity records.  librccl.so is bound at run
 * time (dlopen) by the first ofk_comm_* call; a single-GPU program never needs it.
 *   ofk_comm_unique_id     rank 0: n_ids x ncclGetUniqueId -> n_ids x 128 bytes the caller hands to every rank (file, socket ...)
 *   ofk_comm_init          ncclCommInitRank of communicator 0 on the context's device + the gather buffers.  With n_ids > 1 (one
 *                          communicator per free-running slice of ofk_set_streams, so that every slice gathers its own records on its
 *                          own stream; one is enough for correctness) the ranks first agree on the smallest n_ids any of them passed
 *                          (an all-reduce over communicator 0) and then create exactly that many.  ncclCommInitRank is collective, so
 *                          a failure behind that agreement is an error return on the rank that sees it - fatal for the job, never a
 *                          per-rank fallback that would leave the peers waiting inside the call
 *   ofk_comm_add           one more communicator from a 128-byte id; collective (every rank, same order); what ofk_comm_init does
 *                          n_ids - 1 times, exported for callers that negotiate the count themselves (sharding.Comm)
 *   ofk_comm_gather_records  k_records_f32 of the latest ofk_pairs_run + ncclAllGather of [batch][8] f32 {vx,vy,vz,residual,
 *                          n_used,s_min,rank,corners} into the context's receive buffer `slot` (0/1), stream-ordered behind the step
 *                          on the library's own EXCHANGE stream (one slice; with free-running slices: on the slices' streams): no
 *                          host wait, and step k+1's kernels do not queue behind the collective - its solve only waits for the
 *                          export kernel that reads the records
 *   ofk_comm_fetch_records waits for the gather of `slot`, copies [world][batch][8] f32 (rank-major) to host_out
 *   ofk_comm_allreduce_f64 in-place all-reduce of n <= 64 doubles (op 0 sum, 1 max, 2 min), synchronous: barriers, max-over-ranks
 *                          timing, and the Monte-Carlo sweep's per-step (sum v, sum v^2, count) statistics */
int ofk_comm_unique_id(uint8_t *ids, int n_ids);
int ofk_comm_init(ofk_ctx *ctx, const uint8_t *ids, int n_ids, int rank, int world);
int ofk_comm_add(ofk_ctx *ctx, const uint8_t *id);
int ofk_comm_destroy(ofk_ctx *ctx);
int ofk_comm_rank(const ofk_ctx *ctx);
int ofk_comm_world(const ofk_ctx *ctx);
int ofk_comm_gather_records(ofk_ctx *ctx, int batch, int slot);
int ofk_comm_fetch_records(ofk_ctx *ctx, int slot, int batch, float *host_out);
int ofk_comm_allreduce_f64(ofk_ctx *ctx, double *inout, int n, int op);
/* Communicators this rank holds = what the ranks agreed on (1: the step's records travel in one gather behind the last slice). */
int ofk_comm_count(const ofk_ctx *ctx);
/* Non-blocking watchdog query: bit k set = the gather of slice k of `slot` has not completed (0 = done / nothing queued). */
int ofk_comm_pending(ofk_ctx *ctx, int slot);
/* Host-only helper (needs neither a device nor a communicator): receive-buffer order of a step gathered per slice
 * ([slice][world][pairs of the slice][8] f32) -> rank-major [world][batch][8]; what ofk_comm_fetch_records applies. */
int ofk_comm_reorder_records(const float *recv, int world, int batch, int slices, float *out);

/* Number of concurrent slices ofk_pairs_run cuts the batch into (1..8, default 1): each slice runs the whole stage chain on
 * its own HIP stream so that latency-bound stages overlap with streaming ones; results do not depend on it.  With more than
 * one slice, consecutive ofk_pairs_run calls do not join the slices: they free-run, offset by one response kernel, until any
 * other entry point (download, sync, upload, set_sensors ...) needs their results and joins them.  ofk_pairs_export_records_f32
 * and ofk_mark queue behind the slices without joining them.  (The schedule needs 2 x nstreams hardware queues; the HIP
 * runtime's default is 4 in total - GPU_MAX_HW_QUEUES.) */
int ofk_set_streams(ofk_ctx *ctx, int nstreams);
/* ofk_pairs_run scheduling (default on): the HBM-bound gray conversions and pyramids run on an auxiliary stream into one
 * of two pyramid buffer sets, alternating per call, so that they overlap the VALU-bound response kernel and LK — of this call
 * and, when calls are queued back to back, of the previous one.  Results identical. */
int ofk_set_overlap(ofk_ctx *ctx, int on);
/* Launch-geometry knobs for measurements (process-wide; value 0 restores the built-in choice).  Results are bit-identical for
 * every setting - the knobs move strip lengths and pick between kernels that compute the same thing (tests/test_gpu_image_parity.py
 * runs the parity cases under them).  Knobs: "eig_rows" 8..4096 rows per strip of the streaming response kernels; "no_pair" 1 = one
 * column per lane (k_mineig_stream) where k_mineig_pair would run; "no_pyr3" 1 = pyramid level by level; "pyr3_chunks" row chunks per
 * strip of the three-level pyramid pass; "pyr_rows" rows per strip of the one-level pass; "jpeg_chunk" 64/128/256/512/1024 bytes of entropy
 * data per decoder thread; "jpeg_sub" 1..13 = second-level Huffman look-up tables per image + 1 (1: every code longer than 9 bits takes
 * the canonical search - a test hook for that path); "gray_px" 16/32/64 = the BGR -> gray conversion as one-wave workgroups of that many pixels per thread (an
 * experiment of DESIGN.md section 8: slower in the pipeline).  (Rounds 1-2 read OFK_* environment variables in the launch code instead.) */
int ofk_set_tuning(const char *knob, int value);
int ofk_get_tuning(const char *knob, int *value);
/* Completion marks (slots 0..7): ofk_mark records one behind everything queued so far on every slice, ofk_mark_wait
 * blocks the host until it has been reached (returns at once for a slot never marked).  They let a caller hand step k's
 * records to another library (an RCCL gather) while step k+1 is already queued, without draining the stream. */
int ofk_mark(ofk_ctx *ctx, int slot);
int ofk_mark_wait(ofk_ctx *ctx, int slot);

/* Per-stage HIP-event timing on the context's stream. */
#define OFK_STAGE_GRAY    0
#define OFK_STAGE_PYR     1
#define OFK_STAGE_EIG     2
#define OFK_STAGE_NMS     3
#define OFK_STAGE_SELECT  4
#define OFK_STAGE_LK      5
#define OFK_STAGE_SOLVE   6
#define OFK_N_STAGES      7
int ofk_profile_enable(ofk_ctx *ctx, int stage_mask);   /* bit s set: bracket stage s with hipEvents in ofk_pairs_run */
/* Sums the events recorded since the last call (synchronises the stream). ms_total and launches have OFK_N_STAGES entries. */
int ofk_profile_read(ofk_ctx *ctx, double *ms_total, int *launches);

/* Inspection: the resident pyramid slab of one image — frame set 0 (previous frames) or 1 (next frames) of the latest
 * ofk_pairs_run / ofk_pairs_upload batch, or pyramid slot 0/1 of the stream loop — copied to the host: level 0 (the gray image
 * cvtColor would return, of_module.py:86) at offset 0, level l at the offset ofk_pyramid_u8 reports (256-byte aligned levels,
 * tight rows), `bytes` bytes from the start of the slab.  Synchronises every stream of the context. */
int ofk_resident_pyramid(ofk_ctx *ctx, int frame_set, int image, uint8_t *out, size_t bytes);

#ifdef __cplusplus
}
#endif
#endif /* OFK_H */
