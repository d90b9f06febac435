// Diagnostic: time the lin8 GEMM built from the source file given with -DL8_SRC (A/B runs of main-loop variants in one
// gpurun call).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffast-math -fno-finite-math-only -I vision_ft_amd/csrc
//   -DL8_SRC='"<path>/linear8.hip"' tools/exp/lin8_time.hip -o tools/exp/lin8_time_<tag>;  run: lin8_time_<tag> cfg M N K [M N K ...]
#include L8_SRC

#include <cstdlib>
#include <vector>

void vft_set_error(const char*, ...) {}

int main(int argc, char** argv) {
    const int cfg = atoi(argv[1]);
    for (int a = 2; a + 2 < argc; a += 3) {
        const int M = atoi(argv[a]), N = atoi(argv[a + 1]), K = atoi(argv[a + 2]);
        bf16_t *x, *w, *y;
        hipMalloc(&x, (size_t)M * K * 2), hipMalloc(&w, (size_t)N * K * 2), hipMalloc(&y, (size_t)M * N * 2);
        std::vector<unsigned short> hx((size_t)M * K), hw((size_t)N * K);
        unsigned s = 12345;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (unsigned short)(0x3c00 + ((s >> 9) & 0x3ff) + ((s >> 8) & 0x8000)); };
        for (auto& v : hx) v = rnd();
        for (auto& v : hw) v = rnd();
        hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
        Lin8Params p{};
        p.x = x, p.w = w, p.y = y, p.M = M, p.N = N, p.K = K, p.ldx = K, p.ldw = K, p.ldy = N;
        hipEvent_t e0, e1;
        hipEventCreate(&e0), hipEventCreate(&e1);
        double best = 1e30;
        for (int round = 0; round < 3; ++round) {
            for (int i = 0; i < 3; ++i) vft_lin8_launch(p, cfg, false, 0);
            hipEventRecord(e0);
            for (int i = 0; i < 10; ++i) vft_lin8_launch(p, cfg, false, 0);
            hipEventRecord(e1);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (ms * 100 < best) best = ms * 100;
        }
        printf("cfg=%d M=%6d N=%5d K=%5d: %8.1f us %7.1f TF/s\n", cfg, M, N, K, best, 2.0 * M * N * K / best / 1e6);
        hipFree(x), hipFree(w), hipFree(y);
    }
    return 0;
}
