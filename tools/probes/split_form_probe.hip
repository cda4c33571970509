// What would the fp32-class tower kernel sustain with half the LDS fragment reads per MFMA -- on its own instruction form (16x16x32, a wave
// holding 32 couts: the 2 x 2 cin split) or on the other one (32x32x16)?  Split operand mix (w hi / lo x post-ReLU x hi / lo, 3 products per
// multiply), A in registers, B fragments from LDS at the kernel's ratio, 256 CUs x 4 waves, ~0.4 s per case (package power limit).
//   A: 16x16x32, 16 couts per wave  -- today's k_conv3x3_sp: per (k-step, column tile) 2 fragment reads, 3 MFMAs of 8 KMAC
//   B: 16x16x32, 32 couts per wave  -- 2 fragment reads, 6 MFMAs of 8 KMAC
//   C: 32x32x16, 32 couts per wave  -- 2 fragment reads (32 positions x 16 k), 3 MFMAs of 16 KMAC
// Design probe, not product.  Output: issued TFLOP/s of f16 products.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
__device__ unsigned hashu(unsigned h) { h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16; return h; }
__device__ void split(float v, _Float16& h, _Float16& l) { h = (_Float16)v; l = (_Float16)((v - (float)h) * 2048.0f); }

template <int MODE> __global__ void __launch_bounds__(256, 1) k_probe(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[2][32768];  // hi / lo planes, 64 KiB each
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    for (int i = threadIdx.x; i < 32768; i += 256) {
        float x = (((hashu(i * 2654435761u + blockIdx.x) & 0xffff) / 32768.0f) - 1.0f) * 3.0f;
        if (x < 0.0f) x = 0.0f;
        split(x, lds[0][i], lds[1][i]);
    }
    __syncthreads();
    constexpr int NW = MODE == 0 ? 1 : 2;  // cout tiles of 16 per wave (MODE 2: one 32-cout tile)
    f16x8 wh[2][4], wl[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float w = (((hashu(tid * 131u + c * 977u + i * 17u + e) & 0xffff) / 32768.0f) - 1.0f) * 0.08f;
                _Float16 h, l;
                split(w, h, l);
                wh[c][i][e] = h, wl[c][i][e] = l;
            }
    const int lane = threadIdx.x & 63;
    const unsigned char* bh = (const unsigned char*)lds[0] + lane * 16;
    const unsigned char* bl = (const unsigned char*)lds[1] + lane * 16;
    float s = 0;
    if constexpr (MODE < 2) {
        f32x4 am[NW][4], ac[NW][4];
#pragma unroll
        for (int c = 0; c < NW; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) am[c][j][e] = ac[c][j][e] = 0.0f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int off = ((it * 4 + t) & 15) * 4096;
#pragma unroll
                for (int j = 0; j < 4; ++j) {  // 4 column tiles per k-step
                    const f16x8 xh = *(const f16x8*)(bh + off + j * 1024), xl = *(const f16x8*)(bl + off + j * 1024);
#pragma unroll
                    for (int c = 0; c < NW; ++c) {
                        am[c][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c][t], xh, am[c][j], 0, 0, 0);
                        ac[c][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[c][t], xl, ac[c][j], 0, 0, 0);
                        ac[c][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[c][t], xh, ac[c][j], 0, 0, 0);
                    }
                }
            }
#pragma unroll
        for (int c = 0; c < NW; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) s += am[c][j][e] + ac[c][j][e];
    } else {
        f32x16 am[2], ac[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) am[j][e] = ac[j][e] = 0.0f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int off = ((it * 4 + t) & 15) * 4096;
#pragma unroll
                for (int j = 0; j < 2; ++j) {  // 2 column tiles of 32 positions per k-step of 16
                    const f16x8 xh = *(const f16x8*)(bh + off + j * 1024), xl = *(const f16x8*)(bl + off + j * 1024);
                    am[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0][t], xh, am[j], 0, 0, 0);
                    ac[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[0][t], xl, ac[j], 0, 0, 0);
                    ac[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[0][t], xh, ac[j], 0, 0, 0);
                }
            }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) s += am[j][e] + ac[j][e];
    }
    out[tid] = s;
}
template <int MODE> void run(float* out, const char* name) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    // MACs per wave per iteration: A: 4 k-steps x 4 tiles x 3 x 8K; B: x 2; C: 4 x 2 x 3 x 16K
    const double macs = MODE == 0 ? 4.0 * 4 * 3 * 8192 : MODE == 1 ? 4.0 * 4 * 6 * 8192 : 4.0 * 2 * 3 * 16384;
    const int iters = MODE == 0 ? 400000 : 200000;
    hipLaunchKernelGGL((k_probe<MODE>), dim3(256), dim3(256), 0, 0, out, iters / 8);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < 2; ++r) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k_probe<MODE>), dim3(256), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%-58s %.1f ms  %.0f TFLOP/s of f16 products\n", name, ms, 2.0 * macs * iters * 1024 / (ms * 1e-3) / 1e12);
    }
}
int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 256 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<0>(out, "A 16x16x32, 16 couts/wave: 2 reads per 3 MFMAs (today)");
        run<1>(out, "B 16x16x32, 32 couts/wave: 2 reads per 6 MFMAs (2x2 split)");
        run<2>(out, "C 32x32x16, 32 couts/wave: 2 reads per 3 MFMAs of 16 KMAC");
    }
    return 0;
}
