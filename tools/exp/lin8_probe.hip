// Diagnostic: per-tile cycle breakdown of the lin8 GEMM (top wait | main loop | next-prologue issue | rank ext |
// epilogue passes).  Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only
//   -I vision_ft_amd/csrc tools/exp/lin8_probe.hip -o tools/exp/lin8_probe ;  run:  lin8_probe M N K cfg [reps]
#define L8_STAMPS 1
#include "../../vision_ft_amd/csrc/linear8.hip"

#include <algorithm>
#include <cstdlib>
#include <vector>

void vft_set_error(const char*, ...) {}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 98304, N = argc > 2 ? atoi(argv[2]) : 5120, K = argc > 3 ? atoi(argv[3]) : 640;
    const int cfg = argc > 4 ? atoi(argv[4]) : 1, reps = argc > 5 ? atoi(argv[5]) : 5;
    bf16_t *x, *w, *y;
    hipMalloc(&x, (size_t)M * K * 2), hipMalloc(&w, (size_t)N * K * 2), hipMalloc(&y, (size_t)M * N * 2);
    std::vector<unsigned short> hx((size_t)M * K), hw((size_t)N * K);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (unsigned short)(0x3c00 + ((s >> 9) & 0x3ff) - ((s >> 20) & 1) * 0x8000u * 0 + ((s >> 8) & 0x8000)); };
    for (auto& v : hx) v = rnd();
    for (auto& v : hw) v = rnd();
    hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    Lin8Params p{};
    p.x = x, p.w = w, p.y = y, p.M = M, p.N = N, p.K = K, p.ldx = K, p.ldw = K, p.ldy = N;
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) vft_lin8_launch(p, cfg, false, 0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) vft_lin8_launch(p, cfg, false, 0);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("M=%d N=%d K=%d cfg=%d: %.1f us/launch, %.1f TF/s\n", M, N, K, cfg, us, 2.0 * M * N * K / us / 1e6);
    static unsigned long long st[512][8][12];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_l8_stamps), sizeof(st));
    const int nwg = 256;
    const char* names[] = {"top wait", "main loop", "next prologue issue", "(tail setup)", "rank ext", "pass 0", "pass 1", "pass 2", "pass 3"};
    for (int it = 0; it < 6; ++it) {
        printf("tile %d:", it);
        for (int k = 0; k < 8; ++k) {
            std::vector<double> d;
            for (int b = 0; b < nwg; ++b)
                if (st[b][it][k + 1] > st[b][it][k]) d.push_back((double)(st[b][it][k + 1] - st[b][it][k]));
            if (d.empty()) continue;
            std::sort(d.begin(), d.end());
            printf("  %s %.0f", names[k], d[d.size() / 2]);
        }
        std::vector<double> tot, gap;
        for (int b = 0; b < nwg; ++b) {
            if (st[b][it][8] > st[b][it][0]) tot.push_back((double)(st[b][it][8] - st[b][it][0]));
            if (it < 7 && st[b][it + 1][0] > st[b][it][8]) gap.push_back((double)(st[b][it + 1][0] - st[b][it][8]));
        }
        std::sort(tot.begin(), tot.end());
        std::sort(gap.begin(), gap.end());
        if (!tot.empty()) printf("  | total %.0f", tot[tot.size() / 2]);
        if (!gap.empty()) printf("  gap->next %.0f", gap[gap.size() / 2]);
        printf("  (cycles, median over workgroups)\n");
    }
    return 0;
}
