// Ablation timing of k_mineig_stream<7> (GPU box): compile with -DABLATE_LOAD / -DABLATE_HORNER / -DABLATE_EIG / -DABLATE_NMS
#include "k_corners.hip"
#include <cstdio>
#include <vector>
int main()
{
    const int h = 1080, w = 1920, B = 32;
    std::vector<uint8_t> img((size_t)h * w * B);
    unsigned s = 12345;
    for (auto &p : img) { s = s * 1664525u + 1013904223u; p = (uint8_t)(s >> 24); }
    // smooth it a little so that the candidate density resembles the benchmark texture
    for (size_t i = 2; i < img.size(); ++i) img[i] = (uint8_t)((img[i] + img[i - 1] + img[i - 2]) / 3);
    uint8_t *d_img; hipMalloc(&d_img, img.size()); hipMemcpy(d_img, img.data(), img.size(), hipMemcpyHostToDevice);
    unsigned *d_max; hipMalloc(&d_max, 4 * B * OFK_MAX_STRIDE);
    const int cap = h * w / 4;
    unsigned long long *d_c; hipMalloc(&d_c, 8ull * cap * B);
    int *d_cnt; hipMalloc(&d_cnt, 4 * 32 * B);
    int *d_fl; hipMalloc(&d_fl, 16); hipMemset(d_fl, 0, 16);
    const size_t seg_keys = (size_t)cap * 2 + 64 * 2048;
    unsigned long long *d_seg; hipMalloc(&d_seg, 8ull * seg_keys * B);
    int *d_sc; hipMalloc(&d_sc, 4 * 2048 * B);
    int nseg = 0, segcap = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
        hipMemset(d_max, 0, 4 * B * OFK_MAX_STRIDE); hipMemset(d_cnt, 0, 4 * 32 * B);
        hipEventRecord(e0);
        ofk_launch_mineig_cand(0, d_img, (size_t)h * w, h, w, 7, d_max, nullptr, 0, 0.01, d_c, cap, d_cnt, d_seg, seg_keys, d_sc, 2048, d_fl, B, &nseg, &segcap);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    std::vector<int> sc(nseg); hipMemcpy(sc.data(), d_sc, 4 * nseg, hipMemcpyDeviceToHost);
    int cnt = 0; for (int v : sc) cnt += v;
    printf("%-16s %.3f ms per %d images (%.2f us/image), candidates[0] = %d\n", VARIANT, best, B, best * 1000 / B, cnt);
    return 0;
}
