// Sustained (power-limited) MFMA throughput by instruction shape and operand data: 256 CUs x 4 waves, register-resident operands,
// dense random bf16 vs zeros, ~0.5 s per case.  Informs which MFMA shape burns fewer joules per flop (design probe, not product).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ unsigned hashu(unsigned h) { h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16; return h; }

// SHAPE 0: 32x32x16 (16 KMAC), 1: 16x16x32 (8 KMAC).  MODE 0 zeros, 1 dense random in (-1,1), 2 half zeros (post-ReLU like B operand)
template <int SHAPE> __global__ void __launch_bounds__(256, 1) k_mfma(float* out, int iters, int mode) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float va = ((hashu(tid * 131u + i * 17u + e) & 0xffff) / 32768.0f) - 1.0f;
            float vb = ((hashu(tid * 257u + i * 29u + e + 7777u) & 0xffff) / 32768.0f) - 1.0f;
            if (mode == 0) va = vb = 0.0f;
            if (mode == 2 && vb < 0.0f) vb = 0.0f;
            a[i][e] = (__bf16)va;
            b[i][e] = (__bf16)vb;
        }
    if (SHAPE == 0) {
        f32x16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[(t + j) & 7], acc[j], 0, 0, 0);
        float s = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += acc[j][e];
        out[tid] = s;
    } else {
        f32x4 acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j][e] = 0.0f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[t], b[(t + j) & 7], acc[j], 0, 0, 0);
        float s = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += acc[j][e];
        out[tid] = s;
    }
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* mn[3] = {"zeros", "dense random", "A dense, B half zeros"};
    for (int shape = 0; shape < 2; ++shape)
        for (int mode = 0; mode < 3; ++mode) {
            const int iters = 400000 / (shape == 0 ? 1 : 1);
            const double macs_per_iter = shape == 0 ? 32.0 * 16384 : 64.0 * 8192;  // per wave
            auto launch = [&] {
                if (shape == 0) hipLaunchKernelGGL(k_mfma<0>, dim3(256), dim3(256), 0, 0, out, iters, mode);
                else hipLaunchKernelGGL(k_mfma<1>, dim3(256), dim3(256), 0, 0, out, iters, mode);
            };
            launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            launch();
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double tf = 2.0 * macs_per_iter * iters * 1024 / (ms * 1e-3) / 1e12;
            printf("%s  %-22s %.1f ms  %.0f TFLOP/s\n", shape == 0 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_bf16", mn[mode], ms, tf);
        }
    return 0;
}
