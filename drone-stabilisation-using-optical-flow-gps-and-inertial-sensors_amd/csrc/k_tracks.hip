// k_tracks.hip — feature lifecycle of a video stream on the device:
// velocity_measurment_node:131-173 with its ''' blocks restored — track the previous positions, keep status==1, and when
// few features were left re-detect with a mask of discs around the old positions and append.  gfx950.
#include "ofk_internal.h"
#include <stdlib.h>

// mask := 1, then zero a disc of `radius` around every track (centre = truncated position, dx^2+dy^2 <= r^2).
// grid (pts_stride, batch); block 256.  Streams whose limit is <= 0 (no re-detection this step) are skipped.
__global__ __launch_bounds__(256) void k_disc_mask(uint8_t *__restrict__ mask, size_t mask_stride, int h, int w,
                                                   const float *__restrict__ pts, const int *__restrict__ counts,
                                                   int pts_stride, int radius, const int *__restrict__ limit)
{
    const int b = blockIdx.y, p = blockIdx.x;
    if (limit[b] <= 0 || p >= counts[b]) return;
    const float *q = pts + ((size_t)b * pts_stride + p) * 2;
    const int cx = (int)q[0], cy = (int)q[1];
    const int side = 2 * radius + 1;
    uint8_t *m = mask + (size_t)b * mask_stride;
    for (int i = threadIdx.x; i < side * side; i += 256) {
        const int dy = i / side - radius, dx = i - (i / side) * side - radius;
        const int x = cx + dx, y = cy + dy;
        if (dx * dx + dy * dy <= radius * radius && x >= 0 && x < w && y >= 0 && y < h) m[(size_t)y * w + x] = 0;
    }
}

void ofk_launch_disc_mask(hipStream_t s, uint8_t *mask, size_t mask_stride, int h, int w, const float *pts, const int *counts,
                          int pts_stride, int radius, const int *limit, int batch)
{
    hipLaunchKernelGGL(k_disc_mask, dim3(pts_stride, batch), dim3(256), 0, s, mask, mask_stride, h, w, pts, counts, pts_stride, radius, limit);
}

// limit[b] = max_feat - count[b] if count[b] <= min_feat else 0   (node:157-163)
__global__ void k_redetect_limits(const int *__restrict__ counts, int min_feat, int max_feat, int *__restrict__ limit, int batch)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < batch) limit[b] = counts[b] <= min_feat ? max(0, max_feat - counts[b]) : 0;
}

void ofk_launch_redetect_limits(hipStream_t s, const int *counts, int min_feat, int max_feat, int *limit, int batch)
{
    hipLaunchKernelGGL(k_redetect_limits, dim3((batch + 63) / 64), dim3(64), 0, s, counts, min_feat, max_feat, limit, batch);
}

// tracks := next_pts[status == 1] in order (node:134), then the re-detected corners appended (node:166).  One block per stream.
__global__ __launch_bounds__(256) void k_update_tracks(const float *__restrict__ next_pts, const uint8_t *__restrict__ status,
                                                       const int *__restrict__ counts_in, int pts_stride,
                                                       const float *__restrict__ new_pts, const int *__restrict__ new_counts,
                                                       float *__restrict__ tracks, int *__restrict__ counts_out, int max_total)
{
    __shared__ int s_wave[4];
    __shared__ int s_base;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = counts_in[b];
    const float *src = next_pts + (size_t)b * pts_stride * 2;
    const uint8_t *st = status + (size_t)b * pts_stride;
    float *dst = tracks + (size_t)b * pts_stride * 2;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 256) {
        const int i = i0 + tid;
        const bool keep = i < n && st[i] != 0;
        const float x = keep ? src[2 * i] : 0.f, y = keep ? src[2 * i + 1] : 0.f;
        const unsigned long long bal = __ballot(keep);
        if (lane == 0) s_wave[wave] = __popcll(bal);
        __syncthreads();
        int off = s_base;
        for (int q = 0; q < wave; ++q) off += s_wave[q];
        const int pos = off + __popcll(bal & ((1ull << lane) - 1));
        __syncthreads();                                        // every thread has read the old tracks of this chunk: safe to overwrite
        if (keep) { dst[2 * pos] = x; dst[2 * pos + 1] = y; }
        if (tid == 0) s_base += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    int total = s_base;
    const int extra = new_counts ? min(new_counts[b], max_total - total) : 0;
    for (int i = tid; i < extra; i += 256) {
        dst[2 * (total + i)] = new_pts[((size_t)b * pts_stride + i) * 2];
        dst[2 * (total + i) + 1] = new_pts[((size_t)b * pts_stride + i) * 2 + 1];
    }
    if (tid == 0) counts_out[b] = total + max(extra, 0);
}

void ofk_launch_update_tracks(hipStream_t s, const float *next_pts, const uint8_t *status, const int *counts_in, int pts_stride,
                              const float *new_pts, const int *new_counts, float *tracks, int *counts_out, int max_total, int batch)
{
    hipLaunchKernelGGL(k_update_tracks, dim3(batch), dim3(256), 0, s, next_pts, status, counts_in, pts_stride, new_pts, new_counts, tracks,
                       counts_out, max_total);
}

// of_module.py:83-86: a stream that was left with <= min_feat tracks REPLACES them by a fresh detection (maxCorners = max_feat -
// count, no mask) on its previous frame.  One block per stream; streams whose budget (limit) is 0 keep their tracks.
__global__ __launch_bounds__(256) void k_replace_tracks(const int *__restrict__ limit, const float *__restrict__ new_pts,
                                                        const int *__restrict__ new_counts, int pts_stride, float *__restrict__ tracks,
                                                        int *__restrict__ counts)
{
    const int b = blockIdx.x;
    if (limit[b] <= 0) return;
    const int n = max(new_counts[b], 0);
    for (int i = threadIdx.x; i < 2 * n; i += 256) tracks[(size_t)b * pts_stride * 2 + i] = new_pts[(size_t)b * pts_stride * 2 + i];
    if (threadIdx.x == 0) counts[b] = n;
}

void ofk_launch_replace_tracks(hipStream_t s, const int *limit, const float *new_pts, const int *new_counts, int pts_stride, float *tracks,
                               int *counts, int batch)
{
    hipLaunchKernelGGL(k_replace_tracks, dim3(batch), dim3(256), 0, s, limit, new_pts, new_counts, pts_stride, tracks, counts);
}
