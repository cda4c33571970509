// Ablation harness for k_conv3x3_hb19 (19x19 x 256): one launch = one 128-channel half of a convolution (launch B shape: addend in place).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DCP_ABL_NO_FRAG | -DCP_ABL_NO_DMA] -o conv19_probe_X conv19_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../../alpha_zero_amd/csrc/az_conv19.h"  // the PRODUCT kernel (round 3: pipelined epilogue); conv19_probe.hip times the frozen round-2 copy

__global__ void k_fill(unsigned short* p, size_t n, unsigned seed, int mode) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        float v = ((h & 0xffffff) / 8388608.0f) - 1.0f;
        if (mode == 0) v = 0.0f;
        if (mode == 2) v = v < 0.0f ? 0.0f : v;
        unsigned u = __float_as_uint(v);
        p[i] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
}
int main(int argc, char** argv) {
    const int boards = argc > 1 ? atoi(argv[1]) : 4096, mode = argc > 2 ? atoi(argv[2]) : 2;
    const size_t n = (size_t)boards * 361 * 256;
    unsigned short *x, *y, *w;
    float* bias;
    hipMalloc(&x, n * 2); hipMalloc(&y, n * 2); hipMalloc(&w, (size_t)9 * 256 * 256 * 2); hipMalloc(&bias, 256 * 4);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, x, n, 1u, mode);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, y, n, 2u, mode);
    hipLaunchKernelGGL(k_fill, dim3(256), dim3(256), 0, 0, w, (size_t)9 * 256 * 256, 3u, 1);
    hipMemset(bias, 0, 256 * 4);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int variant = argc > 3 ? atoi(argv[3]) : 0;  // 0: launch B (addend in place), 1: launch A plain (no addend), 2: launch A with a residual tensor
    unsigned short* r2 = nullptr;
    if (variant == 2) { hipMalloc(&r2, n * 2); hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, r2, n, 5u, mode); }
    auto launch = [&] {
        if (variant == 0)
            hipLaunchKernelGGL((k_conv3x3_hb19<true, 16>), dim3(256), dim3(256), 0, 0, (const unsigned char*)x, w, bias, (const unsigned char*)y, (unsigned char*)y,
                               boards, 1, 0, 256, 128, 32, 16);
        else if (variant == 1)
            hipLaunchKernelGGL((k_conv3x3_hb19<false, 16>), dim3(256), dim3(256), 0, 0, (const unsigned char*)x, w, bias, (const unsigned char*)nullptr, (unsigned char*)y,
                               boards, 0, 1, 256, 0, 32, 0);
        else
            hipLaunchKernelGGL((k_conv3x3_hb19<true, 16>), dim3(256), dim3(256), 0, 0, (const unsigned char*)x, w, bias, (const unsigned char*)r2, (unsigned char*)y,
                               boards, 0, 1, 256, 0, 32, 0);
    };
    for (int i = 0; i < 3; ++i) launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 10;
    const double flops = 2.0 * boards * 361 * 256 * 128 * 9;
    printf("half-convolution launch (variant %d), %d boards, data mode %d: %.4f ms  %.1f TFLOP/s\n", variant, boards, mode, ms, flops / ms / 1e9);
    return 0;
}
