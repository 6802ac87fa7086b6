// Phase timing of k_select (GPU box):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -Iinclude -I<pkg>/csrc -DOFK_SEL_STAMPS tools/bench_select.hip -o /tmp/bsel && /tmp/bsel
#include "k_corners.hip"
#include <cstdio>
#include <vector>
int main()
{
    const int h = 1080, w = 1920, B = 128;
    std::vector<uint8_t> img((size_t)h * w);
    unsigned s = 12345;
    for (auto &p : img) { s = s * 1664525u + 1013904223u; p = (uint8_t)(s >> 24); }
    for (int pass = 0; pass < 3; ++pass)                         // smooth: candidate density like the benchmark texture
        for (size_t i = 2; i < img.size(); ++i) img[i] = (uint8_t)((img[i] + img[i - 1] + img[i - 2]) / 3);
    uint8_t *d_img; hipMalloc(&d_img, img.size() * B);
    for (int b = 0; b < B; ++b) hipMemcpy(d_img + (size_t)b * h * w, img.data(), img.size(), hipMemcpyHostToDevice);
    unsigned *d_max; hipMalloc(&d_max, 4 * B * OFK_MAX_STRIDE);
    const int cap = h * w / 4;
    unsigned long long *d_c; hipMalloc(&d_c, 8ull * cap * B);
    int *d_cnt; hipMalloc(&d_cnt, 4 * OFK_CNT_STRIDE * B);
    int *d_fl; hipMalloc(&d_fl, 16); hipMemset(d_fl, 0, 16);
    const size_t seg_keys = (size_t)cap * 2 + 64 * 2048;
    unsigned long long *d_seg; hipMalloc(&d_seg, 8ull * seg_keys * B);
    int *d_sc; hipMalloc(&d_sc, 4 * 2048 * B);
    float *d_pts; hipMalloc(&d_pts, 8 * 512 * B);
    int *d_counts; hipMalloc(&d_counts, 4 * B);
    long long *d_st; hipMalloc(&d_st, 8 * 16); hipMemset(d_st, 0, 8 * 16);
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(long long *));
    int nseg = 0, segcap = 0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) {
        hipMemset(d_max, 0, 4 * B * OFK_MAX_STRIDE); hipMemset(d_cnt, 0, 4 * OFK_CNT_STRIDE * B);
        ofk_launch_mineig_cand(0, d_img, (size_t)h * w, h, w, 7, d_max, nullptr, 0, 0.01, d_c, cap, d_cnt, d_seg, seg_keys, d_sc, 2048, d_fl, B, &nseg, &segcap);
        hipEventRecord(e0);
        ofk_launch_select(0, d_c, cap, d_cnt, d_seg, segcap, d_sc, nseg, d_max, 0.01, w, 500, 10.f, d_pts, 512, d_counts, nullptr, B);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long st[8]; hipMemcpy(st, d_st, sizeof st, hipMemcpyDeviceToHost);
        int cnt, nc; hipMemcpy(&cnt, d_cnt, 4, hipMemcpyDeviceToHost); hipMemcpy(&nc, d_counts, 4, hipMemcpyDeviceToHost);
        printf("select %.3f ms for %d images; block 0: candidates %d corners %d; cycles: compact %lld, pick %lld, gather %lld, sort %lld, greedy %lld\n", ms, B,
               cnt, nc, st[1] - st[0], st[2] - st[1], st[3] - st[2], st[4] - st[3], st[5] - st[4]);
    }
    return 0;
}
