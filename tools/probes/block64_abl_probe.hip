// Ablation harness for k_resblock64<C6Geo<17>, 4>: one part removed per build (tools/probes/make_block64_abl.py generates the header).
//   python make_block64_abl.py && hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DABL_NO_...] -o block64_abl_X block64_abl_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "az_conv64_abl.h"

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
int main(int argc, char** argv) {
    typedef C6Geo<17> G;
    const int boards = argc > 1 ? atoi(argv[1]) : 32768, mode = argc > 2 ? atoi(argv[2]) : 2;
    const size_t n = (size_t)boards * G::P2 * 64;
    unsigned short *x, *y, *w1, *w2;
    float* bias;
    hipMalloc(&x, n * 2); hipMalloc(&y, n * 2); hipMalloc(&w1, 9 * 64 * 64 * 2); hipMalloc(&w2, 9 * 64 * 64 * 2); hipMalloc(&bias, 64 * 4);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, x, n, 1u, mode, 1.0f);
    hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, 0, w1, (size_t)9 * 64 * 64, 3u, 1, 0.08f);
    hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, 0, w2, (size_t)9 * 64 * 64, 4u, 1, 0.08f);
    hipMemset(bias, 0, 64 * 4);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&] {
        hipLaunchKernelGGL((k_resblock64<G, 4>), dim3(256), dim3(CW_THREADS), 0, 0, (const unsigned char*)x, w1, bias, w2, bias, (unsigned char*)y, boards);
    };
    for (int i = 0; i < 40; ++i) launch();  // long warm-up: the clock settles (the first ~30 ms of a process run at a lower clock)
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 40; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 40;
    printf("%-22s %.4f ms per block  %.1f TFLOP/s (%s)\n", argc > 3 ? argv[3] : "variant", ms, 2.0 * 2.0 * boards * 289.0 * 64 * 64 * 9 / ms / 1e9, hipGetErrorString(hipGetLastError()));
    return 0;
}
