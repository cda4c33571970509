// LDS bank-conflict / MFMA issue probes for the conv kernel design (run under rocprofv3 --pmc ...; not part of the product).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int PAT> __device__ unsigned pat_addr(int lane) {
    const int l31 = lane & 31, hi = lane >> 5;
    if (PAT == 0) return lane * 16;                                    // linear
    if (PAT == 1) return l31 * 16 + hi * 4144;                         // tiled layout, two chunk blocks
    if (PAT == 2) return (l31 + 7) * 16 + hi * 4144;                   // shifted rows
    if (PAT == 3) return l31 * 256 + ((hi ^ (l31 & 15)) << 4);         // row-major 256 B rows, XOR swizzle
    if (PAT == 4) return l31 * 256 + hi * 16;                          // row-major no swizzle (16-way)
    if (PAT == 5) return l31 * 16 + hi * 4096;                         // chunk stride multiple of 256 B
    if (PAT == 6) return l31 * 16 + hi * 512;                          // both halves inside one 1 KiB
    if (PAT == 7) return ((l31 * 5) & 31) * 16 + hi * 4144;            // permuted rows
    return 0;
}

template <int PAT> __global__ void __launch_bounds__(256) k_lds(unsigned* out, int iters) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[65536];
    for (int i = threadIdx.x; i < 65536 / 4; i += 256) ((unsigned*)lds)[i] = i;
    __syncthreads();
    const unsigned a = pat_addr<PAT>(threadIdx.x & 63);
    u32x4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        u32x4 v0, v1, v2, v3;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:8288\n\tds_read_b128 %2, %4 offset:16576\n\tds_read_b128 %3, %4 offset:24864\n\ts_waitcnt lgkmcnt(0)"
                     : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a));
        acc += v0 + v1 + v2 + v3;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

// MFMA issue probes: MODE 0 = MFMAs only (A in AGPR), 1 = + one ds_read_b128 per MFMA whose result feeds a later MFMA
// (ring depth 3 steps like the conv kernel), 2 = like 1 with 4 accumulators (4 column tiles per step)
template <int MODE> __global__ void __launch_bounds__(256, 1) k_mfma(float* out, int iters, long long* cyc, unsigned char* scratch) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[132608];
    for (int i = threadIdx.x; i < 132608 / 4; i += 256) ((unsigned*)lds)[i] = 0x3c003c00u;
    __syncthreads();
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    bf16x8 wa[16];
#pragma unroll
    for (int i = 0; i < 16; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) wa[i][e] = (__bf16)(float)(i + e + lane);
    const unsigned char* bp0 = lds + hi * 4144 + l31 * 16;
    const unsigned char* bp1 = bp0 + 32 * 16;
    const unsigned char* bp2 = bp0 + 64 * 16;
    const unsigned char* bp3 = bp0 + 96 * 16;
    f32x16 acc[4];
    unsigned dummy[4] = {1u, 2u, 3u, 4u};
    unsigned sdummy[4] = {1u, 2u, 3u, 4u};
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const u32x2 stdata = {threadIdx.x, blockIdx.x};
    const unsigned long long stbase = (unsigned long long)(scratch + (size_t)blockIdx.x * 65536);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    bf16x8 bb[4][4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) bb[s][j] = *(const bf16x8*)(bp0 + j * 512 + s * 8288);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (MODE >= 1) {
                const int s = t + 3;
                bb[s & 3][0] = *(const bf16x8*)(bp0 + (s & 7) * 8288);
                bb[s & 3][1] = *(const bf16x8*)(bp1 + (s & 7) * 8288);
                if (MODE == 2) {
                    bb[s & 3][2] = *(const bf16x8*)(bp2 + (s & 7) * 8288);
                    bb[s & 3][3] = *(const bf16x8*)(bp3 + (s & 7) * 8288);
                }
            }
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[0]) : "a"(wa[t]), "v"(bb[t & 3][0]));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[1]) : "a"(wa[t]), "v"(bb[t & 3][1]));
            if (MODE == 2) {
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[2]) : "a"(wa[t]), "v"(bb[t & 3][2]));
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[3]) : "a"(wa[t]), "v"(bb[t & 3][3]));
            }
            if (MODE >= 3 && MODE <= 6) {  // independent VALU work in the MFMA shadow: (MODE - 2) * 2 ops per MFMA pair... see main()
#pragma unroll
                for (int q = 0; q < (MODE - 2) * 4; ++q) asm volatile("v_add_u32 %0, %0, %1" : "+v"(dummy[q & 3]) : "v"(lane));
            }
            if (MODE == 7 || MODE == 8) {  // scalar work between MFMAs: 4 / 8 dependent-free SALU ops per step (2 MFMAs)
#pragma unroll
                for (int q = 0; q < (MODE - 6) * 4; ++q) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sdummy[q & 3]));
            }
            if (MODE == 9 && (t & 3) == 1) {  // one 8-byte-per-lane global store per 4 steps (8 MFMAs)
                asm volatile("global_store_dwordx2 %0, %1, %2" : : "v"((unsigned)(threadIdx.x * 8 + t * 2048)), "v"(stdata), "s"(stbase) : "memory");
            }
            if (MODE == 10 && (t & 3) == 1) {  // one LDS-DMA piece per 4 steps, bare (m0 write + instruction)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" : : "v"((unsigned)(lane * 16 + t * 1024)), "s"((unsigned)(65536 + (t & 15) * 1024)), "s"(stbase) : "memory");
            }
            if (MODE == 11 && (t & 3) == 1) {  // the conv kernel's full DMA statement (exec + m0 save / restore)
                unsigned long long save; unsigned keep;
                asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %2\n\ts_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %5\n\ts_mov_b32 m0, %1\n\ts_mov_b64 exec, %0"
                             : "=&s"(save), "=&s"(keep) : "s"(0x0fffffffffffffffull), "s"((unsigned)(65536 + (t & 15) * 1024)), "v"((unsigned)(lane * 16 + t * 1024)), "s"(stbase) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const long long t1 = clock64();
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s + (float)(dummy[0] + dummy[1] + dummy[2] + dummy[3] + sdummy[0] + sdummy[1] + sdummy[2] + sdummy[3]);
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
    unsigned* out;
    float* fout;
    long long* cyc;
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&fout, 256 * 256 * 4);
    hipMalloc(&cyc, 8);
    unsigned char* scratch;
    hipMalloc(&scratch, 256 * 65536);
    const int iters = 2000;
#define RUN_LDS(P) { float ms = timed([&] { hipLaunchKernelGGL(k_lds<P>, dim3(256), dim3(256), 0, 0, out, iters); }); \
        printf("lds pattern %d: %.3f ms, %.2f ns per ds_read_b128 per wave\n", P, ms, ms * 1e6 / (iters * 4.0)); }
    RUN_LDS(0) RUN_LDS(1) RUN_LDS(2) RUN_LDS(3) RUN_LDS(4) RUN_LDS(5) RUN_LDS(6) RUN_LDS(7)
#define RUN_MFMA(M, NM) { float ms = timed([&] { hipLaunchKernelGGL(k_mfma<M>, dim3(256), dim3(256), 0, 0, fout, iters, cyc, scratch); }); \
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); \
        printf("mfma mode %d: %.3f ms, %.1f ns per MFMA, clock64 ticks per MFMA %.1f\n", M, ms, ms * 1e6 / (iters * 16.0 * NM), (double)c / (iters * 16.0 * NM)); }
    RUN_MFMA(0, 2) RUN_MFMA(1, 2) RUN_MFMA(2, 4) RUN_MFMA(3, 2) RUN_MFMA(4, 2) RUN_MFMA(5, 2) RUN_MFMA(6, 2) RUN_MFMA(7, 2) RUN_MFMA(8, 2) RUN_MFMA(9, 2) RUN_MFMA(10, 2) RUN_MFMA(11, 2)
    return 0;
}
