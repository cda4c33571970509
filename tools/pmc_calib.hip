// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the engine's access shapes (MI355X_MICROARCH.md
// says only wide 16-B/lane streams are calibrated: FETCH_SIZE reads 1/2 there).  Known byte counts:
//   k_read4 : every wave reads whole 384-B rows with 4-B-per-lane loads (the select kernel's N/W/P row reads), rows far apart
//   k_write2: every wave writes 2-B-per-lane contiguous runs (the bf16 feature planes)
//   k_write4: 4-B-per-lane contiguous writes (row initialisation in expand)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void k_read4(const float* src, float* sink, int rows_per_wave, long long stride_rows) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float acc = 0.f;
    for (int r = 0; r < rows_per_wave; ++r) {
        const float* row = src + ((wave * rows_per_wave + r) * stride_rows % (1 << 22)) * 96;
        acc += row[lane];
        if (lane < 32) acc += row[64 + lane];
    }
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ void k_write2(unsigned short* dst, int n_per_wave) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int i = lane; i < n_per_wave; i += 64) dst[wave * n_per_wave + i] = (unsigned short)i;
}
__global__ void k_write4(float* dst, int n_per_wave) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int i = lane; i < n_per_wave; i += 64) dst[wave * n_per_wave + i] = (float)i;
}
int main() {
    float *src, *sink; unsigned short* d2; float* d4;
    const size_t nsrc = (size_t)(1 << 22) * 96;  // 1.6 GB of rows
    hipMalloc(&src, nsrc * 4); hipMalloc(&sink, 64); hipMalloc(&d2, (size_t)4096 * 11016 * 2); hipMalloc(&d4, (size_t)4096 * 8 * 96 * 3 * 4);
    hipMemset(src, 0, nsrc * 4);
    for (int it = 0; it < 3; ++it) {
        hipLaunchKernelGGL(k_read4, dim3(1024), dim3(256), 0, 0, src, sink, 30, 977LL);    // 4096 waves x 30 rows x 384 B = 47.2 MB
        hipLaunchKernelGGL(k_write2, dim3(1024), dim3(256), 0, 0, d2, 11016);              // 4096 waves x 11016 x 2 B = 90.2 MB
        hipLaunchKernelGGL(k_write4, dim3(1024), dim3(256), 0, 0, d4, 8 * 96 * 3);         // 4096 waves x 2304 x 4 B = 37.7 MB
    }
    hipDeviceSynchronize();
    printf("known bytes: k_read4 %.1f MB, k_write2 %.1f MB, k_write4 %.1f MB\n", 4096 * 30 * 384 / 1e6, 4096 * 11016 * 2 / 1e6, 4096.0 * 2304 * 4 / 1e6);
    return 0;
}
