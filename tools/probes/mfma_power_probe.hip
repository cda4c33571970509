// Sustained (power-limited) MFMA throughput by instruction shape and operand data: 256 CUs x 4 waves, register-resident operands,
// dense random bf16 vs zeros, ~0.5 s per case.  Informs which MFMA shape burns fewer joules per flop (design probe, not product).
// Round 5: + the split-precision kernel's own instruction and operand mix (v_mfma_f32_16x16x32_f16; A = hi / lo halves of weights,
// B = hi / lo halves of post-ReLU activations, the three products w_hi x_hi, w_hi x_lo, w_lo x_hi in the kernel's ratio).  With an
// argument the results are also written as JSON to that path (profiles/mfma_power_probe.json: bench.py reads its ceilings from there).
// Round 6: the all-zero-operand cases are ISSUE ceilings (power cannot bind with zero operands): v_mfma_f32_16x16x32 issues at ~0.80 of
// the rate of v_mfma_f32_32x32x16 per flop.  + the split-precision mix on v_mfma_f32_32x32x16_f16 (the other instruction form: what a
// 2 x 2 cin split of k_conv3x3_sp would issue), with real and with all-zero operands, so that form x power can be separated.
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

typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
// the split kernel's mix: per k-step 5 column tiles x (w_hi x_hi, w_hi x_lo, w_lo x_hi); operands built like sp_split builds them
__global__ void __launch_bounds__(256, 1) k_mfma_split(float* out, int iters) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    f16x8 wh[4], wl[4], xh[5], xl[5];
    auto split = [](float v, _Float16& h, _Float16& l) { h = (_Float16)v; l = (_Float16)((v - (float)h) * 2048.0f); };
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float w = (((hashu(tid * 131u + i * 17u + e) & 0xffff) / 32768.0f) - 1.0f) * 0.08f;  // Kaiming-like weights
            _Float16 h, l;
            split(w, h, l);
            wh[i][e] = h, wl[i][e] = l;
        }
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = (((hashu(tid * 257u + i * 29u + e + 7777u) & 0xffff) / 32768.0f) - 1.0f) * 3.0f;  // post-ReLU: half zeros
            if (x < 0.0f) x = 0.0f;
            _Float16 h, l;
            split(x, h, l);
            xh[i][e] = h, xl[i][e] = l;
        }
    f32x4 am[5], ac[5];
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) am[j][e] = ac[j][e] = 0.0f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int j = 0; j < 5; ++j) am[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], xh[j], am[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 5; ++j) ac[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[t], xl[j], ac[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 5; ++j) ac[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[t], xh[j], ac[j], 0, 0, 0);
        }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) s += am[j][e] + ac[j][e];
    out[tid] = s;
}

// the same split mix on the 32x32x16 form: per k-step 3 column tiles of 32 positions x (w_hi x_hi, w_hi x_lo, w_lo x_hi); zero != 0: all-zero operands
__global__ void __launch_bounds__(256, 1) k_mfma_split32(float* out, int iters, int zero) {
    const unsigned tid = blockIdx.x * 256 + threadIdx.x;
    f16x8 wh[4], wl[4], xh[3], xl[3];
    auto split = [](float v, _Float16& h, _Float16& l) { h = (_Float16)v; l = (_Float16)((v - (float)h) * 2048.0f); };
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float w = zero ? 0.0f : (((hashu(tid * 131u + i * 17u + e) & 0xffff) / 32768.0f) - 1.0f) * 0.08f;
            _Float16 h, l;
            split(w, h, l);
            wh[i][e] = h, wl[i][e] = l;
        }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = zero ? 0.0f : (((hashu(tid * 257u + i * 29u + e + 7777u) & 0xffff) / 32768.0f) - 1.0f) * 3.0f;
            if (x < 0.0f) x = 0.0f;
            _Float16 h, l;
            split(x, h, l);
            xh[i][e] = h, xl[i][e] = l;
        }
    f32x16 am[3], ac[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) am[j][e] = ac[j][e] = 0.0f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int j = 0; j < 3; ++j) am[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t], xh[j], am[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 3; ++j) ac[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[t], xl[j], ac[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 3; ++j) ac[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[t], xh[j], ac[j], 0, 0, 0);
        }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += am[j][e] + ac[j][e];
    out[tid] = s;
}

int main(int argc, char** argv) {
    FILE* js = argc > 1 ? fopen(argv[1], "w") : nullptr;
    if (js) fprintf(js, "{\"tool\": \"tools/probes/mfma_power_probe.hip\", \"cases\": [");
    bool first = true;
    float* out;
    hipMalloc(&out, 256 * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const char* mn[3] = {"zeros", "dense random", "A dense, B half zeros"};
    double zeros_tf[2] = {0, 0};  // issue ceilings of the two instruction forms (all-zero operands: power cannot bind)
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
            if (js) fprintf(js, "%s{\"mfma\": \"%s\", \"operands\": \"%s\", \"ms\": %.2f, \"tflops\": %.1f}", first ? "" : ", ",
                            shape == 0 ? "v_mfma_f32_32x32x16_bf16" : "v_mfma_f32_16x16x32_bf16", mn[mode], ms, tf);
            first = false;
            if (mode == 0) zeros_tf[shape] = tf;
        }
    float best32 = 0, zero32 = 0;
    for (int zero = 1; zero >= 0; --zero) {  // the split mix on the 32x32x16 form (36 MFMAs of 16 KMAC per k-loop body per wave)
        const int iters = 25000;
        const double macs_per_iter = 36.0 * 16384;
        hipLaunchKernelGGL(k_mfma_split32, dim3(256), dim3(256), 0, 0, out, iters, zero);
        hipDeviceSynchronize();
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mfma_split32, dim3(256), dim3(256), 0, 0, out, 8 * iters, zero);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double tf = 2.0 * macs_per_iter * 8 * iters * 1024 / (ms * 1e-3) / 1e12;
            printf("v_mfma_f32_32x32x16_f16   split-precision mix%s %.1f ms  %.0f TFLOP/s (issued products)\n", zero ? ", ALL-ZERO operands" : "                   ", ms, tf);
            if (zero && tf > zero32) zero32 = (float)tf;
            if (!zero && tf > best32) best32 = (float)tf;
            if (js) fprintf(js, ", {\"mfma\": \"v_mfma_f32_32x32x16_f16\", \"operands\": \"split-precision mix%s\", \"ms\": %.2f, \"tflops\": %.1f}", zero ? ", all-zero operands" : " (w hi/lo x post-ReLU x hi/lo, 3 products)", ms, tf);
        }
    }
    {   // the split-precision kernel's instruction and operand mix
        const int iters = 50000;
        const double macs_per_iter = 60.0 * 8192;  // per wave: 4 k-steps x 15 MFMAs of 16 x 16 x 32
        hipLaunchKernelGGL(k_mfma_split, dim3(256), dim3(256), 0, 0, out, iters);
        hipDeviceSynchronize();
        float best = 0;
        for (int r = 0; r < 3; ++r) {  // ~0.4 s each: long enough for the package power limit to bite
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_mfma_split, dim3(256), dim3(256), 0, 0, out, 8 * iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double tf = 2.0 * macs_per_iter * 8 * iters * 1024 / (ms * 1e-3) / 1e12;
            printf("v_mfma_f32_16x16x32_f16   split-precision mix     %.1f ms  %.0f TFLOP/s (issued products)\n", ms, tf);
            if (tf > best) best = (float)tf;
            if (js) fprintf(js, ", {\"mfma\": \"v_mfma_f32_16x16x32_f16\", \"operands\": \"split-precision mix (w hi/lo x post-ReLU x hi/lo, 3 products)\", \"ms\": %.2f, \"tflops\": %.1f}", ms, tf);
        }
        if (js) fprintf(js, "], \"split_mix_mfma_only_tflops\": %.1f, \"split_mix_32x32x16_tflops\": %.1f, \"split_mix_32x32x16_zero_operands_tflops\": %.1f, "
                            "\"issue_ceiling_16x16x32_tflops\": %.1f, \"issue_ceiling_32x32x16_tflops\": %.1f}\n", best, best32, zero32, zeros_tf[1], zeros_tf[0]);
    }
    if (js) fclose(js);
    return 0;
}
