// fetch_calib.hip — what does rocprofv3's FETCH_SIZE count for the access patterns of this pipeline?  (gfx950)
//
//   hipcc --offload-arch=gfx950 -O2 tools/fetch_calib.hip -o build_variants/fetch_calib
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_calib -- build_variants/fetch_calib > gpurun_out/calib_truth.txt
//   python tools/make_fetch_calibration.py profiles/r03_fetch_calibration.json
//
// MI355X_MICROARCH.md (HBM section): FETCH_SIZE = TCC_EA0_RDREQ x 64 B and reports exactly half of the bytes of a wide coalesced
// read; "other access widths are uncalibrated".  Every kernel below reads a buffer far larger than L2 + Infinity Cache (2 GiB
// against 32 + 256 MiB) and touches every cache line it touches exactly ONCE, so the bytes that must cross the fabric are known
// on the host at both candidate request sizes (unique 64-B lines x 64, unique 128-B lines x 128).  The program prints those
// truths; the FETCH_SIZE of each kernel comes from the profiler; their ratio is the factor to apply to a pipeline kernel with the
// same access pattern:
//   c_wide16   16 B per lane, coalesced           (k_gray_bgr8, k_pyr3_stream: dwordx4 per lane)
//   c_qword8    8 B per lane, coalesced           (k_select_prep / k_select_pick: 64-bit keys)
//   c_dword4    4 B per lane, coalesced           (k_pyr_down_stream)
//   c_rows128   2 rows x 128 B per wave-load at dword alignment, strips overlapping by 12 B, marching down the rows
//                                                   (k_mineig_pair's gray rows)
//   c_lkrows<9,32> / <6,18>   a lane reads ND consecutive dwords of "its" row, 16 rows per 16-lane group, four groups per wave,
//                             windows scattered at dword alignment   (k_lk15q's staging of the next / previous level)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <unordered_set>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void c_wide16(const uint4 *__restrict__ p, size_t n, unsigned *out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void c_qword8(const uint2 *__restrict__ p, size_t n, unsigned *out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) { const uint2 v = p[i]; acc ^= v.x ^ v.y; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ __launch_bounds__(256) void c_dword4(const unsigned *__restrict__ p, size_t n, unsigned *out)
{
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
// images of H x W bytes, strips of 116 columns starting at 116 sx - 8 (clamped into the row), chunks of `rows` rows; the blocks of
// an image go to ONE XCD (blocks n and n + 8 share an L2), as k_mineig_pair maps them: neighbouring strips share the 128-B lines
// their 116-column pitch straddles, and only a common L2 fetches such a line once
__global__ __launch_bounds__(64) void c_rows128(const uint8_t *__restrict__ p, int H, int W, int rows, int nimg, unsigned *out)
{
    const int lane = threadIdx.x;
    const unsigned per = 17 * 4, n = blockIdx.x, k = n >> 3, rem = k % per;
    const int b = 8 * (int)(k / per) + (int)(n & 7);
    if (b >= nimg) return;
    const int chunk = (int)(rem / 17), sx = (int)(rem % 17);
    const uint8_t *img = p + (size_t)b * H * W;
    const int G0 = sx * 116 - 8;
    const int lrow = lane >> 5, lk = lane & 31;
    int off = G0 + 4 * lk; off = off < 0 ? 0 : off > W - 4 ? W - 4 : off;
    const int y0 = chunk * rows - 5, y1 = min(H, chunk * rows + rows + 5);
    unsigned acc = 0;
    for (int y = y0; y < y1; y += 2) {
        int gy = y + lrow; gy = gy < 0 ? -gy : gy; gy = gy >= H ? 2 * (H - 1) - gy : gy;
        acc ^= *reinterpret_cast<const unsigned *>(img + (size_t)gy * W + off);
    }
    if (acc == 0x12345678u) out[0] = acc;
}
// window w: private slot of 256 B x NR rows in an image of pitch 4096; first byte at slot + xoff[w] (dword aligned, < 256 - 4 ND)
template <int ND, int NR>
__global__ __launch_bounds__(64) void c_lkrows(const uint8_t *__restrict__ p, const int *__restrict__ xoff, int nwin, unsigned *out)
{
    const int lane = threadIdx.x, g = lane >> 4, r = lane & 15;
    const int w = blockIdx.x * 4 + g;
    if (w >= nwin) return;
    const size_t base = (size_t)(w >> 4) * NR * 4096 + (size_t)(w & 15) * 256 + xoff[w];
    unsigned acc = 0;
#pragma unroll
    for (int rr = r; rr < NR; rr += 16) {
        const unsigned *q = reinterpret_cast<const unsigned *>(p + base + (size_t)rr * 4096);
        unsigned d[ND];
#pragma unroll
        for (int i = 0; i < ND; ++i) d[i] = q[i];
#pragma unroll
        for (int i = 0; i < ND; ++i) acc ^= d[i];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int ND, int NR>
static void run_lkrows(const uint8_t *buf, size_t bytes, unsigned *out, const char *name)
{
    const int nwin = (int)(bytes / ((size_t)NR * 4096)) * 16;
    std::vector<int> xoff(nwin);
    unsigned s = 12345u + ND;
    size_t u64 = 0, u128 = 0;
    for (int w = 0; w < nwin; ++w) {
        s = s * 1664525u + 1013904223u;
        const int x = (int)((s >> 8) % (unsigned)((256 - 4 * ND) / 4 + 1)) * 4;
        xoff[w] = x;
        const int a = (w & 15) * 256 + x, b = a + 4 * ND - 1;          // byte range of every row of the window inside the 4096-B pitch
        u64 += (size_t)(b / 64 - a / 64 + 1) * NR; u128 += (size_t)(b / 128 - a / 128 + 1) * NR;
    }
    int *dx;
    CK(hipMalloc(&dx, (size_t)nwin * 4));
    CK(hipMemcpy(dx, xoff.data(), (size_t)nwin * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL((c_lkrows<ND, NR>), dim3((nwin + 3) / 4), dim3(64), 0, 0, buf, dx, nwin, out);
    CK(hipDeviceSynchronize());
    printf("%s requested %zu unique64 %zu unique128 %zu\n", name, (size_t)nwin * NR * ND * 4, u64 * 64, u128 * 128);
    CK(hipFree(dx));
}

int main()
{
    const size_t bytes = (size_t)2 << 30;
    uint8_t *buf; unsigned *out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, bytes)); CK(hipMemset(out, 0, 64));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(c_wide16, dim3(8192), dim3(256), 0, 0, reinterpret_cast<const uint4 *>(buf), bytes / 16, out);
    CK(hipDeviceSynchronize());
    printf("c_wide16 requested %zu unique64 %zu unique128 %zu\n", bytes, bytes, bytes);
    hipLaunchKernelGGL(c_qword8, dim3(8192), dim3(256), 0, 0, reinterpret_cast<const uint2 *>(buf), bytes / 8, out);
    CK(hipDeviceSynchronize());
    printf("c_qword8 requested %zu unique64 %zu unique128 %zu\n", bytes, bytes, bytes);
    hipLaunchKernelGGL(c_dword4, dim3(8192), dim3(256), 0, 0, reinterpret_cast<const unsigned *>(buf), bytes / 4, out);
    CK(hipDeviceSynchronize());
    printf("c_dword4 requested %zu unique64 %zu unique128 %zu\n", bytes, bytes, bytes);
    {
        const int H = 1080, W = 1920, rows = 270, nimg = (int)(bytes / ((size_t)H * W)) & ~7;
        hipLaunchKernelGGL(c_rows128, dim3(17 * 4 * nimg), dim3(64), 0, 0, buf, H, W, rows, nimg, out);
        CK(hipDeviceSynchronize());
        const size_t img_bytes = (size_t)nimg * H * W;                // every byte of every image is inside some strip: all lines, once
        printf("c_rows128 requested %zu unique64 %zu unique128 %zu\n", (size_t)nimg * 17 * 4 * (rows + 10) * 128, img_bytes, img_bytes);
    }
    run_lkrows<9, 32>(buf, bytes, out, "c_lkrows<9,32>");
    run_lkrows<6, 18>(buf, bytes, out, "c_lkrows<6,18>");
    CK(hipFree(buf)); CK(hipFree(out));
    return 0;
}
