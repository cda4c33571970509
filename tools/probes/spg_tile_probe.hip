// Tile shape / ring depth / operand sharing of k_conv3x3_spg (csrc/az_conv_spg.h) on a shape without a weight-stationary kernel: 19x19 x 256, 1024 boards,
// and on the latency case (17x17 x 64, 1 and 16 boards).  Prints us per launch and a checksum (all variants compute the same bits).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I ../../alpha_zero_amd/csrc -I ../../include -o spg_tile_probe spg_tile_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "az_conv_spg.h"

// ---- experiment (not in the product: 3 % at 256 filters, 11 % at 64 -- profiles/r06_spg_tile_probe.txt) ----
// The same again with EIGHT waves per workgroup (512 threads) and BOTH operands through LDS: the workgroup computes 4 neighbouring cout groups x
// 2 neighbouring position groups (64 NT couts x 96 positions); a k-step's 8 NT A fragments and 12 B fragments are loaded ONCE per workgroup
// (3 - 4 wave-loads per wave instead of 5 - 6) and every wave reads its 2 NT + 6 fragments from LDS: L1 bytes per MFMA 0.31 -> 0.19 KB at NT = 2
// (k_conv3x3_spgw is L1-bound: one access per CU and cycle, profiles/r06_pmc_spg19.txt).  Ring of three (8 NT + 12) KB LDS slots, one barrier per
// k-step, as in k_conv3x3_spgw.  Same chains, same epilogue, same bits.
template <bool RES, int KSUB, int NT, int HALVES> __global__ void __launch_bounds__(512)
k_conv3x3_spgw8(const unsigned char* __restrict__ x, const _Float16* __restrict__ w, const float* __restrict__ bias, const unsigned char* __restrict__ res,
                unsigned char* __restrict__ y, int nboards, int S, int C, int relu, unsigned* range) {
    constexpr int NJ = 3, NFA = 8 * NT, NFB = 12, NF = NFA + NFB, NL = (NF + 7) / 8, CIN = 32 * KSUB, NCHI = 4 * KSUB, NST = 9 * KSUB, KH = KSUB / HALVES;
    static_assert(KSUB % HALVES == 0, "k-steps");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[3][NF][1024];
    static_assert(sizeof(lds) <= 160 * 1024, "LDS budget");
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), cgi = wave & 3, jgi = wave >> 2;
    const int P2 = S * S, NCT = (P2 + 15) >> 4, NJG = (NCT + NJ - 1) / NJ, NJP = (NJG + 1) / 2, NCQ = C / (64 * NT);
    const long long per_xcd = gridDim.x >> 3, wg = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);  // XCD-aware order (see k_conv3x3_spgw)
    if (wg >= (long long)nboards * NJP * NCQ) return;
    const int cq = (int)(wg % NCQ), jp = (int)((wg / NCQ) % NJP);
    const long long board = wg / ((long long)NCQ * NJP);
    const int cg = cq * 4 + cgi, jg = jp * 2 + jgi;
    const size_t xplane = (size_t)NCHI * P2 * 16, yplane = (size_t)(C / 8) * P2 * 16;
    const unsigned char* xb = x + (size_t)board * 2 * xplane + (size_t)kg * P2 * 16;
    auto geometry = [&](int g, int j, int& pp, unsigned& m) __attribute__((always_inline)) {  // position + tap mask of column tile j of position group g
        const int p = (g * NJ + j) * 16 + l15;
        pp = p < P2 ? p : P2 - 1;
        const int r = pp / S, c = pp - r * S;
        m = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = r + tap / 3 - 1, xx = c + tap % 3 - 1;
            m |= (yy >= 0 && xx >= 0 && yy < S && xx < S) ? (1u << tap) : 0u;
        }
    };
    int pos[NJ];
    bool live[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        unsigned m;
        geometry(jg, j, pos[j], m);
        live[j] = (jg * NJ + j) * 16 + l15 < P2;
    }
    // this wave's share of a k-step's fragments: f = wave + 8 k.  f < NFA: A fragment (cout group f / (2 NT), tile (f / 2) % NT, plane f % 2);
    // otherwise B fragment f - NFA = (position group, plane, column tile) = (fb / 6, (fb % 6) / 3, fb % 3)
    const _Float16* la[NL];
    const unsigned char* lb[NL];
    int lpos[NL];
    unsigned lin[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int f = wave + 8 * k;
        la[k] = w + ((size_t)(f & 1) * 9 * C + (size_t)((cq * 4 + f / (2 * NT)) * NT + (f / 2) % NT) * 16 + l15) * CIN + kg * 8;
        const int fb = f >= NFA && f < NF ? f - NFA : 0;
        geometry(jp * 2 + fb / 6, fb % 3, lpos[k], lin[k]);
        lb[k] = xb + (size_t)((fb % 6) / 3) * xplane;
    }
    sp_f16x8 rg[2][NL];
    auto load_f = [&](int s, int par) __attribute__((always_inline)) {
        const int tap = s / KSUB, ks = s % KSUB, d = (tap / 3 - 1) * S + (tap % 3 - 1);
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int f = wave + 8 * k;  // (wave-uniform branches)
            if (f < NFA) rg[par][k] = *(const sp_f16x8*)(la[k] + (size_t)tap * C * CIN + ks * 32);
            else if (f < NF) rg[par][k] = *(const sp_f16x8*)(lb[k] + ((size_t)ks * 4 * P2 + lpos[k] + (((lin[k] >> tap) & 1u) ? d : 0)) * 16);
        }
    };
    auto store_f = [&](int s, int par) __attribute__((always_inline)) {
        const int tap = s / KSUB;
        const sp_f16x8 zero = (sp_f16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int f = wave + 8 * k;
            if (f < NFA) *(sp_f16x8*)(&lds[s % 3][f][lane * 16]) = rg[par][k];
            else if (f < NF) *(sp_f16x8*)(&lds[s % 3][f][lane * 16]) = ((lin[k] >> tap) & 1u) ? rg[par][k] : zero;
        }
    };
    c6_f32x4 am[HALVES][NT][NJ], ac[HALVES][NT][NJ];
    const float lo_clamp = relu ? 0.0f : -SP_F16_MAX;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int own = HALVES == 2 ? ((cg * NT + t) & 1) : 0;
        c6_f32x4 bv;
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[e] = bias[(cg * NT + t) * 16 + 4 * kg + e];
#pragma unroll
        for (int h = 0; h < HALVES; ++h)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                am[h][t][j] = h == own ? bv : (c6_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                ac[h][t][j] = (c6_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            }
    }
    load_f(0, 0);
    load_f(1, 1);
    store_f(0, 0);
    CV_BARRIER();
    cp_for_each([&](auto SC) __attribute__((always_inline)) {
        constexpr int s = decltype(SC)::value, h = (s % KSUB) / KH;
        sp_f16x8 a[2][NT], bh[NJ], bl[NJ];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            a[0][t] = *(const sp_f16x8*)(&lds[s % 3][(cgi * NT + t) * 2][lane * 16]);
            a[1][t] = *(const sp_f16x8*)(&lds[s % 3][(cgi * NT + t) * 2 + 1][lane * 16]);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bh[j] = *(const sp_f16x8*)(&lds[s % 3][NFA + jgi * 6 + j][lane * 16]);
            bl[j] = *(const sp_f16x8*)(&lds[s % 3][NFA + jgi * 6 + NJ + j][lane * 16]);
        }
        if constexpr (s + 1 < NST) store_f(s + 1, (s + 1) & 1);
        if constexpr (s + 2 < NST) load_f(s + 2, s & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                am[h][t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][t], bh[j], am[h][t][j], 0, 0, 0);
                ac[h][t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[0][t], bl[j], ac[h][t][j], 0, 0, 0);
                ac[h][t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[1][t], bh[j], ac[h][t][j], 0, 0, 0);
            }
        if constexpr (s + 1 < NST) CV_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
    }, typename CpMakeSeq<NST>::type{});
    spg_epilogue<RES, NT, NJ, HALVES>(am, ac, pos, live, res, y, board, yplane, P2, cg, kg, lo_clamp, range);
}


template <int KSUB, int NT, int NJ, int R> static void run(const char* tag, int boards, int S, int C, const unsigned char* x, const _Float16* w, const float* b, unsigned char* y, size_t ybytes) {
    const long long nct = ((long long)S * S + 15) / 16, items = boards * ((nct + NJ - 1) / NJ) * (C / (16 * NT)), grid = (items + 3) / 4;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = boards > 64 ? 10 : 200;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_conv3x3_spg<true, KSUB, NT, NJ, 1, R>), dim3((unsigned)grid), dim3(256), 0, 0, x, w, b, x, y, boards, S, C, 1, nullptr);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_conv3x3_spg<true, KSUB, NT, NJ, 1, R>), dim3((unsigned)grid), dim3(256), 0, 0, x, w, b, x, y, boards, S, C, 1, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(ybytes / 4);
    hipMemcpy(h.data(), y, ybytes, hipMemcpyDeviceToHost);
    unsigned long long cs = 0;
    for (size_t i = 0; i < h.size(); ++i) cs = cs * 1000003ull + h[i];
    const double flop = 2.0 * boards * S * S * (double)C * C * 9;
    printf("%-28s %dx%dx%d boards=%d  NT=%d NJ=%d R=%d  %.1f us  fp32-equiv %.1f TF/s  frac(x3 / 2500) %.3f  checksum %016llx\n", tag, S, S, C, boards, NT, NJ, R, ms / reps * 1e3,
           flop / (ms / reps * 1e-3) / 1e12, 3 * flop / (ms / reps * 1e-3) / 1e12 / 2500.0, cs);
}

template <int KSUB, int NT> static void run_w(const char* tag, int boards, int S, int C, const unsigned char* x, const _Float16* w, const float* b, unsigned char* y, size_t ybytes) {
    const long long nct = ((long long)S * S + 15) / 16, grid = (boards * ((nct + 2) / 3) * (C / (64 * NT)) + 7) / 8 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 10;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_conv3x3_spgw<true, KSUB, NT, 1>), dim3((unsigned)grid), dim3(256), 0, 0, x, w, b, x, y, boards, S, C, 1, nullptr);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_conv3x3_spgw<true, KSUB, NT, 1>), dim3((unsigned)grid), dim3(256), 0, 0, x, w, b, x, y, boards, S, C, 1, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(ybytes / 4);
    hipMemcpy(h.data(), y, ybytes, hipMemcpyDeviceToHost);
    unsigned long long cs = 0;
    for (size_t i = 0; i < h.size(); ++i) cs = cs * 1000003ull + h[i];
    const double flop = 2.0 * boards * S * S * (double)C * C * 9;
    printf("%-28s %dx%dx%d boards=%d  NT=%d NJ=3 B through LDS  %.1f us  fp32-equiv %.1f TF/s  frac(x3 / 2500) %.3f  checksum %016llx\n", tag, S, S, C, boards, NT, ms / reps * 1e3,
           flop / (ms / reps * 1e-3) / 1e12, 3 * flop / (ms / reps * 1e-3) / 1e12 / 2500.0, cs);
}

template <int KSUB, int NT> static void run_w8(const char* tag, int boards, int S, int C, const unsigned char* x, const _Float16* w, const float* b, unsigned char* y, size_t ybytes) {
    const long long nct = ((long long)S * S + 15) / 16, njg = (nct + 2) / 3, grid = (boards * ((njg + 1) / 2) * (C / (64 * NT)) + 7) / 8 * 8;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = 10;
    hipMemset(y, 0, ybytes);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k_conv3x3_spgw8<true, KSUB, NT, 1>), dim3((unsigned)grid), dim3(512), 0, 0, x, w, b, x, y, boards, S, C, 1, nullptr);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((k_conv3x3_spgw8<true, KSUB, NT, 1>), dim3((unsigned)grid), dim3(512), 0, 0, x, w, b, x, y, boards, S, C, 1, nullptr);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned> h(ybytes / 4);
    hipMemcpy(h.data(), y, ybytes, hipMemcpyDeviceToHost);
    unsigned long long cs = 0;
    for (size_t i = 0; i < h.size(); ++i) cs = cs * 1000003ull + h[i];
    const double flop = 2.0 * boards * S * S * (double)C * C * 9;
    printf("%-28s %dx%dx%d boards=%d  NT=%d 8 waves, A and B through LDS  %.1f us  fp32-equiv %.1f TF/s  frac(x3 / 2500) %.3f  checksum %016llx\n", tag, S, S, C, boards, NT, ms / reps * 1e3,
           flop / (ms / reps * 1e-3) / 1e12, 3 * flop / (ms / reps * 1e-3) / 1e12 / 2500.0, cs);
}

int main() {
    const int S = 19, C = 256, boards = 1024;
    const size_t xb = (size_t)boards * 2 * (C / 8) * S * S * 16, wb = (size_t)2 * 9 * C * C * 2;
    std::vector<_Float16> hx(xb / 2), hw(wb / 2);
    srand(1);
    for (auto& v : hx) v = (_Float16)((rand() % 2) ? (rand() % 1000) / 500.0f : 0.0f);
    for (auto& v : hw) v = (_Float16)(((rand() % 2001) - 1000) / 50000.0f);
    std::vector<float> hb(C, 0.05f);
    unsigned char *x, *y;
    _Float16* w;
    float* b;
    hipMalloc(&x, xb), hipMalloc(&y, xb), hipMalloc(&w, wb), hipMalloc(&b, C * 4);
    hipMemcpy(x, hx.data(), xb, hipMemcpyHostToDevice), hipMemcpy(w, hw.data(), wb, hipMemcpyHostToDevice), hipMemcpy(b, hb.data(), C * 4, hipMemcpyHostToDevice);
    run<8, 2, 3, 3>("throughput (round-6 first)", boards, S, C, x, w, b, y, xb);
    run<8, 2, 2, 4>("", boards, S, C, x, w, b, y, xb);
    run<8, 2, 2, 5>("", boards, S, C, x, w, b, y, xb);
    run<8, 2, 2, 6>("", boards, S, C, x, w, b, y, xb);
    run<8, 1, 3, 5>("", boards, S, C, x, w, b, y, xb);
    run<8, 1, 2, 6>("", boards, S, C, x, w, b, y, xb);
    run<8, 2, 3, 4>("", boards, S, C, x, w, b, y, xb);
    run<8, 4, 2, 3>("", boards, S, C, x, w, b, y, xb);
    run_w<8, 2>("workgroup-shared B", boards, S, C, x, w, b, y, xb);
    run_w<8, 4>("workgroup-shared B", boards, S, C, x, w, b, y, xb);
    run_w<8, 1>("workgroup-shared B", boards, S, C, x, w, b, y, xb);
    run_w8<8, 2>("workgroup-shared A and B", boards, S, C, x, w, b, y, xb);
    run_w8<8, 1>("workgroup-shared A and B", boards, S, C, x, w, b, y, xb);
    {   // 19x19 x 64, 4096 boards (a 64-filter shape without a tailored kernel)
        const size_t yb3 = (size_t)4096 * 2 * 8 * 361 * 16;
        run<2, 2, 3, 3>("per-wave", 4096, 19, 64, x, w, b, y, yb3);
        run_w<2, 1>("workgroup-shared B", 4096, 19, 64, x, w, b, y, yb3);
        run_w8<2, 1>("workgroup-shared A and B", 4096, 19, 64, x, w, b, y, yb3);
    }
    // latency: 17x17 x 64 (KSUB 2), 1 and 16 boards
    for (int bl : {1, 16}) {
        const size_t yb2 = (size_t)bl * 2 * 8 * 289 * 16;
        run<2, 1, 2, 3>("latency (round-6 first)", bl, 17, 64, x, w, b, y, yb2);
        run<2, 1, 2, 5>("", bl, 17, 64, x, w, b, y, yb2);
        run<2, 1, 2, 8>("", bl, 17, 64, x, w, b, y, yb2);
        run<2, 1, 1, 8>("", bl, 17, 64, x, w, b, y, yb2);
    }
    return 0;
}
