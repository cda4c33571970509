// Ablation harness for k_conv3x3_tiled (no torch: runs in seconds).  Build one binary per switch:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off [-DCP_ABL_NO_DMA | -DCP_ABL_NO_EPI | ...] -o probe_X conv_pipe_probe.hip
// Prints ms per launch, and the chip clock implied by s_memtime of one wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "az_conv_abl.h"  // frozen round-2 copy of alpha_zero_amd/csrc/az_conv.h with the ablation switches (the product header has none)

__global__ void k_fill(unsigned short* p, size_t n, unsigned seed, int mode) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u + seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        // mode 0: zeros; 1: uniform(-1,1) bf16; 2: post-ReLU like (half zeros, rest uniform(0,1))
        float v = ((h & 0xffffff) / 8388608.0f) - 1.0f;
        if (mode == 0) v = 0.0f;
        if (mode == 2) v = v < 0.0f ? 0.0f : v;
        unsigned u = __float_as_uint(v);
        p[i] = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
}
__global__ void k_clock(long long* out) {  // s_memtime ticks per microsecond reference: spin ~ fixed MFMA count not needed; host measures
    out[0] = clock64();
}

int main(int argc, char** argv) {
    const int boards = argc > 1 ? atoi(argv[1]) : 32768, mode = argc > 2 ? atoi(argv[2]) : 2;
    const int ntiles = (boards + 2) / 3;
    const size_t tile_elems = CT_TILE / 2, n = (size_t)ntiles * tile_elems;
    unsigned short *x, *r, *y, *w;
    float* bias;
    hipMalloc(&x, n * 2); hipMalloc(&r, n * 2); hipMalloc(&y, n * 2); hipMalloc(&w, 9 * 128 * 128 * 2); hipMalloc(&bias, 128 * 4);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, x, n, 1u, mode);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, r, n, 2u, mode);
    hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, 0, w, (size_t)9 * 128 * 128, 3u, 1);
    hipMemset(bias, 0, 128 * 4);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int resid = 0; resid < 2; ++resid) {
        auto launch = [&] {
            if (resid) hipLaunchKernelGGL((k_conv3x3_tiled<true, 16>), dim3(256), dim3(256), 0, 0, (const unsigned char*)x, w, bias, (const unsigned char*)r, (unsigned char*)y, ntiles, 1);
            else hipLaunchKernelGGL((k_conv3x3_tiled<false, 16>), dim3(256), dim3(256), 0, 0, (const unsigned char*)x, w, bias, (const unsigned char*)nullptr, (unsigned char*)y, ntiles, 1);
        };
        for (int i = 0; i < 5; ++i) launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 20;
        const double flops = 2.0 * boards * 81 * 128 * 128 * 9;
        printf("residual=%d data_mode=%d: %.4f ms  %.1f TFLOP/s  (MFMA-only floor at 2.4 GHz: %.4f ms)\n", resid, mode, ms, flops / ms / 1e9,
               43.0 * 576 * 32 / 2.4e6);
    }
    return 0;
}
