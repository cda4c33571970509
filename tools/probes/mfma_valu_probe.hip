// Does a wave's VALU work hide in the shadow of its own MFMAs (one wave per SIMD), and does it depend on which register
// bank (ArchVGPR / AccVGPR) the accumulators live in?  Does a second wave on the SIMD hide it?  (design probe, not product)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// ACC: 0 = accumulators in ArchVGPRs ("v"), 1 = in AccVGPRs ("a");  WA: 0 = A operand in VGPR, 1 = in AGPR
// NV = VALU instructions per MFMA, KIND: 0 v_add_u32, 1 v_cvt_pk_bf16_f32, 2 v_add_f32, 3 v_accvgpr_read (only ACC=1), 4 v_pk_add_f32
template <int ACC, int WA, int NV, int KIND, int THREADS>
__global__ void __launch_bounds__(THREADS, THREADS / 256) k_probe(float* out, int iters, long long* cyc) {
    const int lane = threadIdx.x & 63;
    bf16x8 wa[8], bb[4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) wa[i][e] = (__bf16)(float)((i + e + lane) & 7);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) bb[i][e] = (__bf16)(float)((i * 3 + e + lane) & 3);
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    unsigned d[8] = {1u, 2u, 3u, 4u, 5u, 6u, 7u, 8u};
    float f[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
    typedef __attribute__((ext_vector_type(2))) float f2;
    f2 p[4] = {{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}, {7.f, 8.f}};
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (ACC == 0 && WA == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "a"(wa[t]), "v"(bb[j]));
                if (ACC == 0 && WA == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(wa[t]), "v"(bb[j]));
                if (ACC == 1 && WA == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "a"(wa[t]), "v"(bb[j]));
                if (ACC == 1 && WA == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[j]) : "v"(wa[t]), "v"(bb[j]));
#pragma unroll
                for (int q = 0; q < NV; ++q) {
                    const int r = (q + j * NV) & 7;
                    if (KIND == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(d[r]) : "v"(lane));
                    if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d[r]) : "v"(f[r]), "v"(f[(r + 1) & 7]));
                    if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[r]) : "v"(f[(r + 3) & 7]));
                    if (KIND == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[r & 3]) : "v"(p[(r + 1) & 3]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const long long t1 = clock64();
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    for (int i = 0; i < 8; ++i) s += (float)d[i] + f[i];
    for (int i = 0; i < 4; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <class F> static float timed(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* fout;
    long long* cyc;
    hipMalloc(&fout, 256 * 512 * 4);
    hipMalloc(&cyc, 8);
    const int iters = 1000;
#define RUN(ACC, WA, NV, KIND, TH)                                                                                                   \
    {                                                                                                                               \
        float ms = timed([&] { hipLaunchKernelGGL((k_probe<ACC, WA, NV, KIND, TH>), dim3(256), dim3(TH), 0, 0, fout, iters, cyc); }); \
        long long c;                                                                                                                \
        hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);                                                                              \
        printf("acc=%s A=%s valu/mfma=%d kind=%d waves/simd=%d: %.3f ms, ticks per MFMA (per wave) %.1f, per SIMD %.1f\n",          \
               ACC ? "agpr" : "vgpr", WA ? "agpr" : "vgpr", NV, KIND, TH / 256, ms, (double)c / (iters * 32.0),                    \
               (double)c / (iters * 32.0) / (TH / 256));                                                                          \
    }
    RUN(0, 1, 0, 0, 256) RUN(0, 1, 2, 0, 256) RUN(0, 1, 4, 0, 256)
    RUN(1, 0, 0, 0, 256) RUN(1, 0, 2, 0, 256) RUN(1, 0, 4, 0, 256)
    RUN(1, 1, 0, 0, 256) RUN(1, 1, 2, 0, 256) RUN(1, 1, 4, 0, 256)
    RUN(0, 0, 0, 0, 256) RUN(0, 0, 2, 0, 256)
    RUN(1, 0, 2, 1, 256) RUN(1, 0, 2, 2, 256) RUN(1, 0, 2, 4, 256) RUN(0, 1, 2, 1, 256) RUN(0, 1, 2, 2, 256)
    RUN(0, 1, 0, 0, 512) RUN(0, 1, 2, 0, 512) RUN(0, 1, 4, 0, 512) RUN(1, 0, 4, 0, 512) RUN(0, 1, 4, 2, 512)
    return 0;
}
