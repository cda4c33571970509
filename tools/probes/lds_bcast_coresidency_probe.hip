// Round 4 follow-up of lds_dma_coresidency_probe.hip.  tools/probes/make_head_probe.py + tools/concurrency_probe2.py showed: k_head_tiled's
// results differ beside a second forward ONLY when it reads its 1x1 weights from its LDS copy (reading them from global memory: exact);
// the LDS copy itself is intact at the end of the kernel.  So a ds_read RETURNED other data than the LDS holds.  The head kernel's reads
// are BROADCAST reads (every lane the same address); its neighbours on the CU are convolution workgroups that stream LDS-DMA pieces
// (global_load_lds_dwordx4 under a lane mask written to EXEC, destination in M0).  This probe isolates that pair:
//   victim V : a small LDS table, read again and again with wave-uniform (broadcast) ds_read_b32 / ds_read_b128 or with per-lane
//              addresses; every value is compared with the expected pattern (mismatches counted and sampled, the table is never rewritten)
//   neighbour D: 40 KB of LDS filled again and again -- by LDS-DMA with all lanes, by LDS-DMA under a partial EXEC mask restored to -1
//              afterwards (what the convolution kernels do), or through registers and ds_write (control)
// hipcc --offload-arch=gfx950 -O3 -o lds_bcast_coresidency_probe lds_bcast_coresidency_probe.hip ; run: ./lds_bcast_coresidency_probe [iters]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

// MODE 0: per-lane addresses, 1: broadcast b32, 2: broadcast b128, 3: broadcast ds_read2_b32
template <int MODE> __global__ void __launch_bounds__(256) k_victim(unsigned* err, unsigned* sample, int iters) {
    extern __shared__ unsigned v[];  // 1536 B = 384 words (the head kernel's weight table: 3 x 128 floats)
    const int n = 384;
    for (int i = threadIdx.x; i < n; i += 256) v[i] = 0xabcd0000u | (unsigned)i;
    __syncthreads();
    unsigned bad = 0, first_i = 0, first_x = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            for (int i = threadIdx.x; i < n; i += 256) {
                const unsigned x = *(volatile unsigned*)&v[i];
                if (x != (0xabcd0000u | (unsigned)i)) {
                    if (!bad) first_i = i, first_x = x;
                    ++bad;
                }
            }
        } else if (MODE == 1) {
            for (int i = 0; i < n; ++i) {  // wave-uniform address: one broadcast read per word
                const unsigned x = *(volatile unsigned*)&v[i];
                if (x != (0xabcd0000u | (unsigned)i)) {
                    if (!bad) first_i = i, first_x = x;
                    ++bad;
                }
            }
        } else if (MODE == 3) {  // the head kernel's own instruction: ds_read2_b32 with a wave-uniform address
            for (int i = 0; i < n; i += 2) {
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                u32x2 x;
                const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)&v[i];
                asm volatile("ds_read2_b32 %0, %1 offset1:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(a) : "memory");
                if (x.x != (0xabcd0000u | (unsigned)i) || x.y != (0xabcd0000u | (unsigned)(i + 1))) {
                    if (!bad) first_i = i, first_x = x.x != (0xabcd0000u | (unsigned)i) ? x.x : x.y;
                    ++bad;
                }
            }
        } else {
            for (int i = 0; i < n; i += 4) {
                const u32x4 x = *(volatile u32x4*)&v[i];
                const unsigned xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (xs[e] != (0xabcd0000u | (unsigned)(i + e))) {
                        if (!bad) first_i = i + e, first_x = xs[e];
                        ++bad;
                    }
            }
        }
    }
    if (bad) {
        const unsigned k = atomicAdd(err, 1u);  // threads that saw at least one wrong value
        atomicAdd(err + 1, bad);
        if (k < 8) {
            sample[4 * k] = first_i, sample[4 * k + 1] = first_x, sample[4 * k + 2] = threadIdx.x, sample[4 * k + 3] = blockIdx.x;
        }
    }
    // the table itself at the end
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256)
        if (v[i] != (0xabcd0000u | (unsigned)i)) atomicAdd(err + 2, 1u);
}

// KIND 0: LDS-DMA, all lanes; 1: LDS-DMA under a partial EXEC mask, EXEC restored to -1 (the convolution kernels' dma_piece); 2: registers + ds_write
template <int KIND> __global__ void __launch_bounds__(256) k_neighbour(const unsigned char* src, unsigned* sink, int iters) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[40 * 1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const unsigned long long mask = 0x0ffffffffffffff0ull;  // lanes 4 .. 59
    for (int it = 0; it < iters; ++it) {
        for (int p = wave; p < 40; p += 4) {  // 40 pieces of 1 KiB
            const unsigned dst = lds0 + (unsigned)p * 1024u, off = (unsigned)lane * 16u;
            if (KIND == 0) {
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(dst), "v"(off), "s"(src + (size_t)p * 1024) : "memory");
            } else if (KIND == 1) {
                asm volatile("s_mov_b64 exec, %0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1"
                             :
                             : "s"(mask), "s"(dst), "v"(off), "s"(src + (size_t)p * 1024)
                             : "memory");
            } else {
                const uint4 d = *(const uint4*)(src + (size_t)p * 1024 + lane * 16);
                *(uint4*)(lds + p * 1024 + lane * 16) = d;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        unsigned s = 0;
        for (int i = threadIdx.x; i < 40 * 64; i += 256) s += ((const uint4*)lds)[i].x;  // the neighbour reads its image back (ds_read_b128, like B fragments)
        if (s == 0x12345u) sink[0] = s;
        __syncthreads();
    }
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 300;
    unsigned *err, *sample, *sink;
    unsigned char* src;
    hipMalloc(&err, 16); hipMalloc(&sample, 32 * 4); hipMalloc(&sink, 8); hipMalloc(&src, 64 * 1024);
    unsigned* h = (unsigned*)malloc(64 * 1024);
    for (int i = 0; i < 16 * 1024; ++i) h[i] = 0xdeadbeefu;
    hipMemcpy(src, h, 64 * 1024, hipMemcpyHostToDevice);
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    const char* vname[4] = {"per-lane ds_read_b32", "broadcast ds_read_b32", "broadcast ds_read_b128", "broadcast ds_read2_b32"};
    const char* nname[4] = {"alone", "LDS-DMA (all lanes)", "LDS-DMA (EXEC mask, restored)", "global_load + ds_write"};
    for (int vm = 0; vm < 4; ++vm)
        for (int nb = 0; nb < 4; ++nb) {
            hipMemset(err, 0, 16); hipMemset(sample, 0, 32 * 4);
            hipDeviceSynchronize();
            // neighbours first (persistent-ish: one per CU x 4), victims stream in beside them
            if (nb == 1) hipLaunchKernelGGL(k_neighbour<0>, dim3(1024), dim3(256), 0, s2, (const unsigned char*)src, sink, iters * 2);
            if (nb == 2) hipLaunchKernelGGL(k_neighbour<1>, dim3(1024), dim3(256), 0, s2, (const unsigned char*)src, sink, iters * 2);
            if (nb == 3) hipLaunchKernelGGL(k_neighbour<2>, dim3(1024), dim3(256), 0, s2, (const unsigned char*)src, sink, iters * 2);
            if (vm == 0) hipLaunchKernelGGL(k_victim<0>, dim3(8192), dim3(256), 1536, s1, err, sample, iters * 8);
            if (vm == 1) hipLaunchKernelGGL(k_victim<1>, dim3(8192), dim3(256), 1536, s1, err, sample, iters / 4 + 1);
            if (vm == 2) hipLaunchKernelGGL(k_victim<2>, dim3(8192), dim3(256), 1536, s1, err, sample, iters);
            if (vm == 3) hipLaunchKernelGGL(k_victim<3>, dim3(8192), dim3(256), 1536, s1, err, sample, iters / 2 + 1);
            hipDeviceSynchronize();
            unsigned e[4] = {0, 0, 0, 0}, smp[32];
            hipMemcpy(e, err, 16, hipMemcpyDeviceToHost); hipMemcpy(smp, sample, 32 * 4, hipMemcpyDeviceToHost);
            printf("victim %-22s beside %-30s: threads with a wrong read %u, wrong reads %u, table words wrong at the end %u", vname[vm], nname[nb], e[0], e[1], e[2]);
            for (unsigned k = 0; k < (e[0] < 3 ? e[0] : 3); ++k) printf("  [word %u read 0x%08x, thread %u block %u]", smp[4 * k], smp[4 * k + 1], smp[4 * k + 2], smp[4 * k + 3]);
            printf("  (%s)\n", hipGetErrorString(hipGetLastError()));
        }
    return 0;
}
