// Does operand reuse between consecutive MFMAs change the power-limited throughput?  32x32x16 bf16, dense random operands, 1024 waves;
// RA = consecutive MFMAs sharing the same A registers, RB = sharing the same B registers (design probe, not product).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ unsigned hashu(unsigned h) { h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16; return h; }

template <int RA, int RB> __global__ void __launch_bounds__(256, 1) k_mfma(float* out, int iters) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    bf16x8 a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a[i][e] = (__bf16)(((hashu(tid * 131u + i * 17u + e) & 0xffff) / 32768.0f) - 1.0f);
            float vb = ((hashu(tid * 257u + i * 29u + e + 7777u) & 0xffff) / 32768.0f) - 1.0f;
            b[i][e] = (__bf16)(vb < 0.0f ? 0.0f : vb);
        }
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int t = 0; t < 32; ++t) acc[t & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(t / RA) & 7], b[(t / RB) & 7], acc[t & 3], 0, 0, 0);
    float s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[tid] = s;
}

template <int RA, int RB> void run(float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 400000;
    hipLaunchKernelGGL((k_mfma<RA, RB>), dim3(256), dim3(256), 0, 0, out, iters / 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mfma<RA, RB>), dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("A held for %d MFMAs, B held for %d: %.1f ms  %.0f TFLOP/s\n", RA, RB, ms, 2.0 * 32 * 16384.0 * iters * 1024 / (ms * 1e-3) / 1e12);
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 256 * 4);
    run<1, 1>(out); run<2, 1>(out); run<4, 1>(out); run<8, 1>(out); run<1, 2>(out); run<1, 4>(out); run<32, 1>(out); run<1, 32>(out); run<1, 1>(out);
    return 0;
}
