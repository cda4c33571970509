// Issue rate of v_mfma_f32_16x16x16_f16 against v_mfma_f32_16x16x32_f16 on gfx950 (one wave per SIMD, 8 independent accumulators, register operands):
// does the K = 16 form cost half the cycles of the K = 32 form?  (a 48-channel tower would contract K = 32 + 16 per tap.)
// Prints cycles per instruction from s_memtime (shader clock) for both forms.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int K32> __global__ void __launch_bounds__(256, 1) k_rate(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f16x8 a8, b8;
    f16x4 a4, b4;
#pragma unroll
    for (int e = 0; e < 8; ++e) a8[e] = (_Float16)(0.01f * (threadIdx.x % 7 + e)), b8[e] = (_Float16)(0.02f * (threadIdx.x % 5 + e));
#pragma unroll
    for (int e = 0; e < 4; ++e) a4[e] = a8[e], b4[e] = b8[e];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (K32) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, acc[j], 0, 0, 0);
                else acc[j] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[j], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[K32] = t1 - t0;
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 256 * 4);
    hipMallocManaged(&cyc, 16);
    const int iters = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(k_rate<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        hipLaunchKernelGGL(k_rate<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
        hipDeviceSynchronize();
    }
    const double n = 64.0 * iters;
    printf("v_mfma_f32_16x16x16_f16: %.2f s_memtime ticks per instruction\n", cyc[0] / n);
    printf("v_mfma_f32_16x16x32_f16: %.2f s_memtime ticks per instruction\n", cyc[1] / n);
    printf("(s_memtime counts at a fixed 100 MHz on gfx9: compare the two lines as a RATIO; the x32 form issues every 16 shader cycles)\n");
    return 0;
}
