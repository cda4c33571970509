// Tile shape / ring depth of k_conv3x3_spg (csrc/az_conv_spg.h) on a shape without a weight-stationary kernel: 19x19 x 256, 1024 boards,
// and on the latency case (17x17 x 64, 1 and 16 boards).  Prints us per launch and a checksum (all variants compute the same bits).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I ../../alpha_zero_amd/csrc -I ../../include -o spg_tile_probe spg_tile_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "az_conv_spg.h"

template <int KSUB, int NT, int NJ, int R> static void run(const char* tag, int boards, int S, int C, const unsigned char* x, const _Float16* w, const float* b, unsigned char* y, size_t ybytes) {
    const long long nct = ((long long)S * S + 15) / 16, items = boards * ((nct + NJ - 1) / NJ) * (C / (16 * NT)), grid = (items + 3) / 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = boards > 64 ? 10 : 200;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_conv3x3_spg<true, KSUB, NT, NJ, 1, R>), dim3((unsigned)grid), dim3(256), 0, 0, x, w, b, x, y, boards, S, C, 1, nullptr);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_conv3x3_spg<true, KSUB, NT, NJ, 1, R>), dim3((unsigned)grid), dim3(256), 0, 0, x, w, b, x, y, boards, S, C, 1, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(ybytes / 4);
    hipMemcpy(h.data(), y, ybytes, hipMemcpyDeviceToHost);
    unsigned long long cs = 0;
    for (size_t i = 0; i < h.size(); ++i) cs = cs * 1000003ull + h[i];
    const double flop = 2.0 * boards * S * S * (double)C * C * 9;
    printf("%-28s %dx%dx%d boards=%d  NT=%d NJ=%d R=%d  %.1f us  fp32-equiv %.1f TF/s  frac(x3 / 2500) %.3f  checksum %016llx\n", tag, S, S, C, boards, NT, NJ, R, ms / reps * 1e3,
           flop / (ms / reps * 1e-3) / 1e12, 3 * flop / (ms / reps * 1e-3) / 1e12 / 2500.0, cs);
}

template <int KSUB, int NT> static void run_w(const char* tag, int boards, int S, int C, const unsigned char* x, const _Float16* w, const float* b, unsigned char* y, size_t ybytes) {
    const long long nct = ((long long)S * S + 15) / 16, grid = (boards * ((nct + 2) / 3) * (C / (64 * NT)) + 7) / 8 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 10;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_conv3x3_spgw<true, KSUB, NT, 1>), dim3((unsigned)grid), dim3(256), 0, 0, x, w, b, x, y, boards, S, C, 1, nullptr);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_conv3x3_spgw<true, KSUB, NT, 1>), dim3((unsigned)grid), dim3(256), 0, 0, x, w, b, x, y, boards, S, C, 1, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(ybytes / 4);
    hipMemcpy(h.data(), y, ybytes, hipMemcpyDeviceToHost);
    unsigned long long cs = 0;
    for (size_t i = 0; i < h.size(); ++i) cs = cs * 1000003ull + h[i];
    const double flop = 2.0 * boards * S * S * (double)C * C * 9;
    printf("%-28s %dx%dx%d boards=%d  NT=%d NJ=3 B through LDS  %.1f us  fp32-equiv %.1f TF/s  frac(x3 / 2500) %.3f  checksum %016llx\n", tag, S, S, C, boards, NT, ms / reps * 1e3,
           flop / (ms / reps * 1e-3) / 1e12, 3 * flop / (ms / reps * 1e-3) / 1e12 / 2500.0, cs);
}

int main() {
    const int S = 19, C = 256, boards = 1024;
    const size_t xb = (size_t)boards * 2 * (C / 8) * S * S * 16, wb = (size_t)2 * 9 * C * C * 2;
    std::vector<_Float16> hx(xb / 2), hw(wb / 2);
    srand(1);
    for (auto& v : hx) v = (_Float16)((rand() % 2) ? (rand() % 1000) / 500.0f : 0.0f);
    for (auto& v : hw) v = (_Float16)(((rand() % 2001) - 1000) / 50000.0f);
    std::vector<float> hb(C, 0.05f);
    unsigned char *x, *y;
    _Float16* w;
    float* b;
    hipMalloc(&x, xb), hipMalloc(&y, xb), hipMalloc(&w, wb), hipMalloc(&b, C * 4);
    hipMemcpy(x, hx.data(), xb, hipMemcpyHostToDevice), hipMemcpy(w, hw.data(), wb, hipMemcpyHostToDevice), hipMemcpy(b, hb.data(), C * 4, hipMemcpyHostToDevice);
    run<8, 2, 3, 3>("throughput (round-6 first)", boards, S, C, x, w, b, y, xb);
    run<8, 2, 2, 4>("", boards, S, C, x, w, b, y, xb);
    run<8, 2, 2, 5>("", boards, S, C, x, w, b, y, xb);
    run<8, 2, 2, 6>("", boards, S, C, x, w, b, y, xb);
    run<8, 1, 3, 5>("", boards, S, C, x, w, b, y, xb);
    run<8, 1, 2, 6>("", boards, S, C, x, w, b, y, xb);
    run<8, 2, 3, 4>("", boards, S, C, x, w, b, y, xb);
    run<8, 4, 2, 3>("", boards, S, C, x, w, b, y, xb);
    run_w<8, 2>("workgroup-shared B", boards, S, C, x, w, b, y, xb);
    run_w<8, 4>("workgroup-shared B", boards, S, C, x, w, b, y, xb);
    run_w<8, 1>("workgroup-shared B", boards, S, C, x, w, b, y, xb);
    {   // 19x19 x 64, 4096 boards (a 64-filter shape without a tailored kernel)
        const size_t yb3 = (size_t)4096 * 2 * 8 * 361 * 16;
        run<2, 2, 3, 3>("per-wave", 4096, 19, 64, x, w, b, y, yb3);
        run_w<2, 1>("workgroup-shared B", 4096, 19, 64, x, w, b, y, yb3);
    }
    // latency: 17x17 x 64 (KSUB 2), 1 and 16 boards
    for (int bl : {1, 16}) {
        const size_t yb2 = (size_t)bl * 2 * 8 * 289 * 16;
        run<2, 1, 2, 3>("latency (round-6 first)", bl, 17, 64, x, w, b, y, yb2);
        run<2, 1, 2, 5>("", bl, 17, 64, x, w, b, y, yb2);
        run<2, 1, 2, 8>("", bl, 17, 64, x, w, b, y, yb2);
        run<2, 1, 1, 8>("", bl, 17, 64, x, w, b, y, yb2);
    }
    return 0;
}
