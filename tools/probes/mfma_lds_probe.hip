// Power-limited MFMA throughput against the number of LDS fragment reads per MFMA (0, 1/2, 1): what would halving the convolution's
// B-fragment reads buy?  32x32x16 bf16, A in registers (dense random), B from LDS (post-ReLU-like), 1024 waves (design probe).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ unsigned hashu(unsigned h) { h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16; return h; }

// READS: LDS reads per 4 MFMAs (0, 2 or 4)
template <int READS> __global__ void __launch_bounds__(256, 1) k_mfma(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[65536];  // 128 KiB of bf16
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < 65536; i += 256) {
        float v = ((hashu(i * 2654435761u + blockIdx.x) & 0xffff) / 32768.0f) - 1.0f;
        v = v < 0.0f ? 0.0f : v;
        unsigned u = __float_as_uint(v);
        lds[i] = (unsigned short)(u >> 16);
    }
    __syncthreads();
    bf16x8 a[8], b[4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)(((hashu(tid * 131u + i * 17u + e) & 0xffff) / 32768.0f) - 1.0f);
    const int lane = threadIdx.x & 63;
    const unsigned char* base = (const unsigned char*)lds + lane * 16;
#pragma unroll
    for (int j = 0; j < 4; ++j) b[j] = *(const bf16x8*)(base + j * 1024);
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int off = ((it * 8 + t) & 31) * 4096;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (j < READS) b[j] = *(const bf16x8*)(base + off + j * 1024);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[t], b[j], acc[j], 0, 0, 0);
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[tid] = s;
}
template <int READS> void run(float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 400000;
    hipLaunchKernelGGL((k_mfma<READS>), dim3(256), dim3(256), 0, 0, out, iters / 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_mfma<READS>), dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%d LDS fragment reads per 4 MFMAs: %.1f ms  %.0f TFLOP/s\n", READS, ms, 2.0 * 32 * 16384.0 * iters * 1024 / (ms * 1e-3) / 1e12);
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 256 * 4);
    run<0>(out); run<2>(out); run<4>(out); run<0>(out); run<2>(out); run<4>(out);
    return 0;
}
