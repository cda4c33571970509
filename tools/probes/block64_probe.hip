// Timing harness for the fused ResNetBlock kernel of the 64-filter towers (alpha_zero_amd/csrc/az_conv64.h: k_resblock64<G, R>):
// the product instantiation (R = 4) beside other ring depths and the two-launch path, on N(0,1)-like / post-ReLU-like / zero data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o block64_probe block64_probe.hip
//   ./block64_probe [boards = 32768] [data mode: 0 zeros, 1 uniform(-1,1), 2 post-ReLU-like] [S = 17 | 9]
// (template parameters only: the product header carries no ablation switches)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../alpha_zero_amd/csrc/az_conv64.h"

__global__ void k_fill(unsigned short* p, size_t n, unsigned seed, int mode, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        float v = (((h & 0xffffff) / 8388608.0f) - 1.0f) * scale;
        if (mode == 0) v = 0.0f;
        if (mode == 2) v = v < 0.0f ? 0.0f : v;
        unsigned u = __float_as_uint(v);
        p[i] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
}

template <class G> int run(int boards, int mode) {
    const long long ntiles = (boards + G::TB - 1) / G::TB;
    const size_t n = (size_t)ntiles * G::P2 * 64;
    unsigned short *x, *m, *y, *w1, *w2;
    float* bias;
    hipMalloc(&x, n * 2); hipMalloc(&m, n * 2); hipMalloc(&y, n * 2); hipMalloc(&w1, 9 * 64 * 64 * 2); hipMalloc(&w2, 9 * 64 * 64 * 2); hipMalloc(&bias, 64 * 4);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, x, n, 1u, mode, 1.0f);
    hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, 0, w1, (size_t)9 * 64 * 64, 3u, 1, 0.08f);
    hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, 0, w2, (size_t)9 * 64 * 64, 4u, 1, 0.08f);
    hipMemset(bias, 0, 64 * 4);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256)), block(CW_THREADS);
    const unsigned char *xb = (const unsigned char*)x, *mb = (const unsigned char*)m;
    const double flops = 2.0 * 2.0 * (double)boards * G::S * G::S * 64 * 64 * 9;
    auto time_it = [&](const char* name, auto launch) {
        for (int i = 0; i < 3; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 20;
        printf("%-34s S=%d boards=%d data=%d: %.4f ms per block  %.1f TFLOP/s  (%s)\n", name, G::S, boards, mode, ms, flops / ms / 1e9, hipGetErrorString(hipGetLastError()));
    };
    time_it("k_resblock64<R=4> (product)", [&] { hipLaunchKernelGGL((k_resblock64<G, 4>), grid, block, 0, 0, xb, w1, bias, w2, bias, (unsigned char*)y, (int)ntiles); });
    time_it("k_resblock64<R=6>", [&] { hipLaunchKernelGGL((k_resblock64<G, 6>), grid, block, 0, 0, xb, w1, bias, w2, bias, (unsigned char*)y, (int)ntiles); });
    time_it("k_resblock64<R=3>", [&] { hipLaunchKernelGGL((k_resblock64<G, 3>), grid, block, 0, 0, xb, w1, bias, w2, bias, (unsigned char*)y, (int)ntiles); });
    time_it("two k_conv3x3_t64 launches", [&] {
        hipLaunchKernelGGL((k_conv3x3_t64<G, false, 8>), grid, block, 0, 0, xb, w1, bias, (const unsigned char*)nullptr, (unsigned char*)m, (int)ntiles, 1);
        hipLaunchKernelGGL((k_conv3x3_t64<G, true, 8>), grid, block, 0, 0, mb, w2, bias, xb, (unsigned char*)y, (int)ntiles, 1);
    });
    return 0;
}

int main(int argc, char** argv) {
    const int boards = argc > 1 ? atoi(argv[1]) : 32768, mode = argc > 2 ? atoi(argv[2]) : 2, S = argc > 3 ? atoi(argv[3]) : 17;
    return S == 9 ? run<C6Geo<9>>(boards, mode) : run<C6Geo<17>>(boards, mode);
}
