// Does an LDS-DMA (global_load_lds_dwordx4, destination = M0) of one workgroup land in ANOTHER workgroup's LDS when the two share a CU?
// Round 3 found k_head_tiled's results changing when its workgroups were co-resident with workgroups of a convolution-family kernel of a
// second stream (tools/concurrency_probe2.py; gone when the head kernel is padded to 96 KB of LDS so that nothing can share its CU).
//   victim kernel V: small LDS allocation filled with a pattern, re-checked for a few milliseconds; mismatches are counted and sampled
//   DMA kernel D   : 40 KB of LDS, copies 0xdeadbeef words from global memory into ALL of it with LDS-DMA pieces, again and again
// V and D run on two streams at the same time.  hipcc --offload-arch=gfx950 -O3 -o lds_dma_coresidency_probe lds_dma_coresidency_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

__global__ void __launch_bounds__(256) k_victim(unsigned* err, unsigned* sample, int iters) {
    extern __shared__ unsigned v[];  // 2048 B
    const int n = 512;
    for (int i = threadIdx.x; i < n; i += 256) v[i] = 0xabcd0000u | (unsigned)i;
    __syncthreads();
    for (int it = 0; it < iters; ++it) {
        for (int i = threadIdx.x; i < n; i += 256) {
            const unsigned x = v[i];
            if (x != (0xabcd0000u | (unsigned)i)) {
                const unsigned k = atomicAdd(err, 1u);
                if (k < 16) {
                    sample[2 * k] = (unsigned)i;
                    sample[2 * k + 1] = x;
                }
                v[i] = 0xabcd0000u | (unsigned)i;
            }
        }
        __builtin_amdgcn_s_sleep(64);
    }
}

template <bool DMA> __global__ void __launch_bounds__(256) k_dma(const unsigned char* src, unsigned* sink, int iters) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[40 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
        for (int p = wave; p < 40; p += 4) {  // 40 pieces of 1 KiB
            if (DMA) {
                const unsigned dst = lds0 + (unsigned)p * 1024u, off = (unsigned)lane * 16u;
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(dst), "v"(off), "s"(src + (size_t)p * 1024) : "memory");
            } else {  // the same bytes through registers and ds_write (control)
                const uint4 d = *(const uint4*)(src + (size_t)p * 1024 + lane * 16);
                *(uint4*)(lds + p * 1024 + lane * 16) = d;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        for (int i = threadIdx.x; i < 40 * 256; i += 256)  // the DMA kernel checks its own LDS image, too (a misplaced piece would show here)
            if (*(const unsigned*)(lds + i * 4) != 0xdeadbeefu) atomicAdd(sink + 1, 1u);
        __syncthreads();
        for (int i = threadIdx.x; i < 40 * 64; i += 256) *(uint4*)(lds + i * 16) = (uint4){0u, 0u, 0u, 0u};
        __syncthreads();
    }
    if (acc == 0x12345u) sink[0] = acc;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    unsigned *err, *sample, *sink;
    unsigned char* src;
    hipMalloc(&err, 4); hipMalloc(&sample, 32 * 4); hipMalloc(&sink, 8); hipMalloc(&src, 64 * 1024);
    hipMemset(src, 0, 64 * 1024);
    unsigned* h = (unsigned*)malloc(64 * 1024);
    for (int i = 0; i < 16 * 1024; ++i) h[i] = 0xdeadbeefu;
    hipMemcpy(src, h, 64 * 1024, hipMemcpyHostToDevice);
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    for (int mode = 0; mode < 5; ++mode) {  // 0: victim alone, 1: beside the register-copy control, 2: beside the LDS-DMA kernel (DMA first),
        hipMemset(err, 0, 4); hipMemset(sample, 0, 32 * 4); hipMemset(sink, 0, 8);  // 3 / 4: victims first, so that DMA workgroups start at a non-zero LDS base
        hipDeviceSynchronize();
        if (mode == 1) hipLaunchKernelGGL(k_dma<false>, dim3(1024), dim3(256), 0, s2, (const unsigned char*)src, sink, iters * 4);
        if (mode == 2) hipLaunchKernelGGL(k_dma<true>, dim3(1024), dim3(256), 0, s2, (const unsigned char*)src, sink, iters * 4);
        hipLaunchKernelGGL(k_victim, dim3(mode >= 3 ? 16384 : 4096), dim3(256), mode == 4 ? 1536 : 2048, s1, err, sample, iters);
        if (mode >= 3) hipLaunchKernelGGL(k_dma<true>, dim3(2048), dim3(256), 0, s2, (const unsigned char*)src, sink, iters);
        hipDeviceSynchronize();
        unsigned e = 0, smp[32], dm[2];
        hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost); hipMemcpy(smp, sample, 32 * 4, hipMemcpyDeviceToHost); hipMemcpy(dm, sink, 8, hipMemcpyDeviceToHost);
        printf("mode %d (%s): DMA kernel's own mismatches = %u, corrupted victim words = %u", mode, mode == 0 ? "victim alone" : mode == 1 ? "beside global_load + ds_write copies" : mode == 2 ? "beside LDS-DMA copies" : mode == 3 ? "victims (2048 B) first, then LDS-DMA" : "victims (1536 B) first, then LDS-DMA", dm[1], e);
        for (unsigned k = 0; k < (e < 4 ? e : 4); ++k) printf("  [word %u = 0x%08x]", smp[2 * k], smp[2 * k + 1]);
        printf("  (%s)\n", hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
