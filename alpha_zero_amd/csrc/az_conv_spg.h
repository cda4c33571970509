// az_conv_spg.h -- the fp32-class 3x3 convolution of az_conv_sp.h (hi + lo f16 planes, three v_mfma_f32_16x16x32_f16 products per multiply,
// two fp32 accumulators) for ANY plane size and 64 / 128 / 256 filters, one WAVE per output tile, no LDS, no barrier:
//     y = relu(conv3x3(x, w) + bias [+ residual])        x, residual, y: [board][plane: hi, lo][C/8 chunks][S*S positions][8 channels] f16
// Two jobs (round 6):
//   * LATENCY.  The weight-stationary kernels (k_conv3x3_sp / _sp2 / _sp17, k_resblock_sp) give one CU a whole board: built for 32 768
//     boards per launch, they take 18 - 32 us on ONE board (a batch-1 forward of the drop-in uct_search: 14 launches, 370 us of GPU time,
//     profiles/r06_kernel_stats_dropin_c1.txt).  Here a board is ceil(S*S / 16 / NJ) x C / 16 / NT independent waves that fill the chip.
//   * SHAPES WITHOUT A TAILORED KERNEL (19x19, 13x13 Go planes, 256 filters ...): the fp32-class tower no longer needs the library.
// A wave computes NT cout tiles (16 couts) x NJ column tiles (16 positions) over all input channels.  Per k-step (one tap x 32 input
// channels) it loads its A fragments from the weights [2][9][C][Cin] and its B fragments from the activations straight from global
// memory (both are MFMA-fragment shaped 16-byte cells: lane (n = lane & 15, kg = lane >> 4) reads channels 32 ks + 8 kg .. + 8 of
// cout / position n; neighbouring waves of a workgroup share (board, column tiles) and differ in couts: their B loads hit L1), through a
// ring of R k-steps in registers.  A tap outside the plane is a zero fragment (the load is redirected to the lane's own position and
// the result replaced by zeros).
// BIT-IDENTICAL to the weight-stationary kernels of the same shape (tests/test_split_tower.py), so that the evaluator's result for a
// position does not depend on the batch it arrives in: the same MFMA chains -- main: bias, then w_hi x_hi over the k-steps in tap-major
// order; corr: w_hi x_lo, w_lo x_hi per k-step -- the same join v = fma(corr, 2^-11, main), residual join + add, range record in front
// of the ReLU, median clamp, packed convert, exact remainder.  HALVES = 2 reproduces k_conv3x3_sp2's two chains (az_conv_sp2.h: the cin
// half with the parity of the cout tile carries the bias; v = own + other).  Zero fragments are multiplied like any other (the 9x9
// kernels skip the taps of their corner position instead: adding +0 products changes nothing but the sign of an exact zero sum).
#pragma once
#include "az_conv_sp.h"

#if defined(__HIPCC__)
// Epilogue of a wave's NT x NJ output tiles (shared by the two kernels below): the micro-op sequence of the weight-stationary kernels'
// riding epilogues (az_conv_sp.h sp_epi_*), as plain code.
template <bool RES, int NT, int NJ, int HALVES>
__device__ __forceinline__ void spg_epilogue(const c6_f32x4 (&am)[HALVES][NT][NJ], const c6_f32x4 (&ac)[HALVES][NT][NJ], const int (&pos)[NJ], const bool (&live)[NJ],
                                             const unsigned char* __restrict__ res, unsigned char* __restrict__ y, long long board, size_t yplane, int P2, int cg,
                                             int kg, float lo_clamp, unsigned* range) {
    float mx = 0.0f;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int own = HALVES == 2 ? ((cg * NT + t) & 1) : 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            // this lane's four couts 16 (cg NT + t) + 4 kg .. + 4 of position pos[j]: half (kg & 1) of the cell in chunk 2 (cg NT + t) + kg / 2
            const size_t off = (size_t)board * 2 * yplane + ((size_t)((cg * NT + t) * 2 + (kg >> 1)) * P2 + pos[j]) * 16 + (kg & 1) * 8;
            cv_u32x2 rh = (cv_u32x2){0u, 0u}, rl = (cv_u32x2){0u, 0u};
            if (RES) {
                rh = *(const cv_u32x2*)(res + off);
                rl = *(const cv_u32x2*)(res + off + yplane);
            }
            unsigned hpk[2], lpk[2];
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                float ev[2];
#pragma unroll
                for (int ei = 0; ei < 2; ++ei) {
                    const int e = 2 * pr + ei;
                    if constexpr (HALVES == 2) {
                        const float a = own ? fmaf(ac[1][t][j][e], SP_INV_SCALE, am[1][t][j][e]) : fmaf(ac[0][t][j][e], SP_INV_SCALE, am[0][t][j][e]);
                        const float b = own ? fmaf(ac[0][t][j][e], SP_INV_SCALE, am[0][t][j][e]) : fmaf(ac[1][t][j][e], SP_INV_SCALE, am[1][t][j][e]);
                        ev[ei] = cw_add_f32(a, b);
                    } else {
                        ev[ei] = fmaf(ac[0][t][j][e], SP_INV_SCALE, am[0][t][j][e]);
                    }
                    if (RES) {
                        const unsigned h2 = pr == 0 ? rh.x : rh.y, l2 = pr == 0 ? rl.x : rl.y;
                        const float t0 = ei == 0 ? sp_mix_join<0>(h2, l2) : sp_mix_join<1>(h2, l2);
                        ev[ei] = cw_add_f32(ev[ei], t0);
                    }
                }
                mx = sp_max3_abs(mx, ev[0], ev[1]);
                ev[0] = __builtin_amdgcn_fmed3f(ev[0], lo_clamp, SP_F16_MAX);
                ev[1] = __builtin_amdgcn_fmed3f(ev[1], lo_clamp, SP_F16_MAX);
                hpk[pr] = sp_cvt_pk(ev[0], ev[1]);
                const float s0 = sp_mix_diff<0>(hpk[pr], ev[0]), s1 = sp_mix_diff<1>(hpk[pr], ev[1]);
                lpk[pr] = sp_scale_cvt_hi(sp_scale_cvt_lo(s0), s1);
            }
            if (live[j]) {
                *(cv_u32x2*)(y + off) = (cv_u32x2){hpk[0], hpk[1]};
                *(cv_u32x2*)(y + off + yplane) = (cv_u32x2){lpk[0], lpk[1]};
            }
        }
    }
    sp_range_report(mx, range);
}

template <bool RES, int KSUB, int NT, int NJ, int HALVES, int R = 3> __global__ void __launch_bounds__(256)
k_conv3x3_spg(const unsigned char* __restrict__ x, const _Float16* __restrict__ w, const float* __restrict__ bias, const unsigned char* __restrict__ res,
              unsigned char* __restrict__ y, int nboards, int S, int C, int relu, unsigned* range) {
    constexpr int CIN = 32 * KSUB, NCHI = 4 * KSUB, NST = 9 * KSUB, KH = KSUB / HALVES;  // R: k-steps in the fragment ring
    static_assert(KSUB % HALVES == 0 && NST >= R, "k-steps");
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P2 = S * S, NCT = (P2 + 15) >> 4, NJG = (NCT + NJ - 1) / NJ, NCG = C / (16 * NT);
    const long long item = (long long)blockIdx.x * 4 + wave;
    if (item >= (long long)nboards * NJG * NCG) return;  // (uniform per wave; the kernel has no barrier)
    const int cg = (int)(item % NCG), jg = (int)((item / NCG) % NJG);
    const long long board = item / ((long long)NCG * NJG);
    const size_t xplane = (size_t)NCHI * P2 * 16, yplane = (size_t)(C / 8) * P2 * 16;
    const unsigned char* xb = x + (size_t)board * 2 * xplane + (size_t)kg * P2 * 16;
    const _Float16* wb = w + (size_t)(cg * NT * 16 + l15) * CIN + kg * 8;

    int pos[NJ];
    unsigned inside[NJ];  // bit tap: the tap's source cell is on the plane
    bool live[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int p = (jg * NJ + j) * 16 + l15;
        live[j] = p < P2;
        pos[j] = live[j] ? p : P2 - 1;
        const int r = pos[j] / S, c = pos[j] - r * S;
        unsigned m = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = r + tap / 3 - 1, xx = c + tap % 3 - 1;
            m |= (yy >= 0 && xx >= 0 && yy < S && xx < S) ? (1u << tap) : 0u;
        }
        inside[j] = m;
    }
    sp_f16x8 ra[R][2][NT], rb[R][2][NJ];  // the ring: [k-step slot][plane][tile]
    auto load_step = [&](int s, int slot) __attribute__((always_inline)) {
        const int tap = s / KSUB, ks = s % KSUB;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            ra[slot][0][t] = *(const sp_f16x8*)(wb + ((size_t)tap * C + t * 16) * CIN + ks * 32);
            ra[slot][1][t] = *(const sp_f16x8*)(wb + ((size_t)(9 + tap) * C + t * 16) * CIN + ks * 32);
        }
        const int d = (tap / 3 - 1) * S + (tap % 3 - 1);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int q = pos[j] + (((inside[j] >> tap) & 1u) ? d : 0);
            const unsigned char* src = xb + ((size_t)ks * 4 * P2 + q) * 16;
            rb[slot][0][j] = *(const sp_f16x8*)src;
            rb[slot][1][j] = *(const sp_f16x8*)(src + xplane);
        }
    };
    c6_f32x4 am[HALVES][NT][NJ], ac[HALVES][NT][NJ];
    const float lo_clamp = relu ? 0.0f : -SP_F16_MAX;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int own = HALVES == 2 ? ((cg * NT + t) & 1) : 0;  // the chain that carries the bias (az_conv_sp2.h: the cout tile's own cin half)
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
#pragma unroll
    for (int s = 0; s < R - 1; ++s) load_step(s, s);
    cp_for_each([&](auto SC) __attribute__((always_inline)) {
        constexpr int s = decltype(SC)::value, slot = s % R, tap = s / KSUB, ks = s % KSUB, h = ks / KH;
        if constexpr (s + R - 1 < NST) load_step(s + R - 1, (s + R - 1) % R);
        __builtin_amdgcn_sched_barrier(0);
        const sp_f16x8 zero = (sp_f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        sp_f16x8 bh[NJ], bl[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const bool in = (inside[j] >> tap) & 1u;
            bh[j] = in ? rb[slot][0][j] : zero;
            bl[j] = in ? rb[slot][1][j] : zero;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                am[h][t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[slot][0][t], bh[j], am[h][t][j], 0, 0, 0);
                ac[h][t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[slot][0][t], bl[j], ac[h][t][j], 0, 0, 0);
                ac[h][t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[slot][1][t], bh[j], ac[h][t][j], 0, 0, 0);
            }
        __builtin_amdgcn_sched_barrier(0);
    }, typename CpMakeSeq<NST>::type{});

    spg_epilogue<RES, NT, NJ, HALVES>(am, ac, pos, live, res, y, board, yplane, P2, cg, kg, lo_clamp, range);
}

// The same tiles with the B fragments SHARED BY A WORKGROUP (large calls on shapes without a weight-stationary kernel): the four waves of a
// workgroup compute four neighbouring cout groups of the same (board, 3 column tiles); k_conv3x3_spg loads their common activations four
// times through L1 -- it is L1-bound (tools/probes/spg_tile_probe.hip: time follows the bytes per MFMA, not the ring depth).  Here the
// six B fragments of a k-step (3 column tiles x hi, lo) are loaded ONCE per workgroup (waves 0, 1 two each, waves 2, 3 one), masked, and
// handed over through a ring of three 6 KB LDS slots: L1 traffic per k-step 4 x (NT + 3) x 2 KB -> (4 NT + 3) x 2 KB.  Per k-step s a wave
// reads slot s % 3, writes its share of k-step s + 1 into slot (s + 1) % 3 (last read two barriers ago), issues the global loads of
// k-step s + 2, multiplies, and meets the others at ONE barrier.  The A fragments stay a per-wave register ring.  Same chains, same
// epilogue: the same bits as k_conv3x3_spg.
template <bool RES, int KSUB, int NT, int HALVES> __global__ void __launch_bounds__(256)
k_conv3x3_spgw(const unsigned char* __restrict__ x, const _Float16* __restrict__ w, const float* __restrict__ bias, const unsigned char* __restrict__ res,
               unsigned char* __restrict__ y, int nboards, int S, int C, int relu, unsigned* range) {
    constexpr int NJ = 3, NB = 2 * NJ, CIN = 32 * KSUB, NCHI = 4 * KSUB, NST = 9 * KSUB, KH = KSUB / HALVES, RA = 3;
    static_assert(KSUB % HALVES == 0, "k-steps");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[3][NB][1024];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int P2 = S * S, NCT = (P2 + 15) >> 4, NJG = (NCT + NJ - 1) / NJ, NCQ = C / (64 * NT);  // groups of four cout groups
    // XCD-aware order: workgroup ids go round-robin over the 8 XCDs (each with its own L2); all workgroups of a board (NJG x NCQ of them, each
    // reading the board's activations with a halo) are given to ONE XCD, so that the board is fetched from HBM once instead of once per XCD
    // (PMC, 19x19 x 256: FETCH_SIZE 4.5 GB per launch = 12 x the input with the plain order).  The launcher pads the grid to a multiple of 8.
    const long long per_xcd = gridDim.x >> 3, wg = (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (wg >= (long long)nboards * NJG * NCQ) return;  // (uniform per workgroup, in front of every barrier)
    const int cg = (int)(wg % NCQ) * 4 + wave, jg = (int)((wg / NCQ) % NJG);
    const long long board = wg / ((long long)NCQ * NJG);
    const size_t xplane = (size_t)NCHI * P2 * 16, yplane = (size_t)(C / 8) * P2 * 16;
    const unsigned char* xb = x + (size_t)board * 2 * xplane + (size_t)kg * P2 * 16;
    const _Float16* wb = w + (size_t)(cg * NT * 16 + l15) * CIN + kg * 8;
    int pos[NJ];
    unsigned inside[NJ];
    bool live[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int p = (jg * NJ + j) * 16 + l15;
        live[j] = p < P2;
        pos[j] = live[j] ? p : P2 - 1;
        const int r = pos[j] / S, c = pos[j] - r * S;
        unsigned m = 0;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = r + tap / 3 - 1, xx = c + tap % 3 - 1;
            m |= (yy >= 0 && xx >= 0 && yy < S && xx < S) ? (1u << tap) : 0u;
        }
        inside[j] = m;
    }
    // this wave's share of a k-step's B fragments: fragment f = plane * 3 + column tile; wave q loads f = q and (q < 2) f = q + 4
    const int f0 = wave, f1 = wave + 4;
    const int j0 = f0 % NJ, j1 = f1 % NJ;                    // (f1 = 4, 5 -> plane 1, column tiles 1, 2)
    const int p0 = pos[0] * (j0 == 0) + pos[1] * (j0 == 1) + pos[2] * (j0 == 2), p1 = j1 == 1 ? pos[1] : pos[2];
    const unsigned in0 = j0 == 0 ? inside[0] : (j0 == 1 ? inside[1] : inside[2]), in1 = j1 == 1 ? inside[1] : inside[2];
    const size_t pl0 = (size_t)(f0 / NJ) * xplane;           // f0 = 0 .. 3: planes 0, 0, 0, 1;   f1: plane 1
    sp_f16x8 ra[RA][2][NT], rg[2][2];                        // A ring [slot][plane][tile]; B loads in flight [k-step parity][this wave's 2]
    auto load_a = [&](int s, int slot) __attribute__((always_inline)) {
        const int tap = s / KSUB, ks = s % KSUB;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            ra[slot][0][t] = *(const sp_f16x8*)(wb + ((size_t)tap * C + t * 16) * CIN + ks * 32);
            ra[slot][1][t] = *(const sp_f16x8*)(wb + ((size_t)(9 + tap) * C + t * 16) * CIN + ks * 32);
        }
    };
    auto load_b = [&](int s, int par) __attribute__((always_inline)) {
        const int tap = s / KSUB, ks = s % KSUB, d = (tap / 3 - 1) * S + (tap % 3 - 1);
        rg[par][0] = *(const sp_f16x8*)(xb + pl0 + ((size_t)ks * 4 * P2 + p0 + (((in0 >> tap) & 1u) ? d : 0)) * 16);
        if (wave < 2) rg[par][1] = *(const sp_f16x8*)(xb + xplane + ((size_t)ks * 4 * P2 + p1 + (((in1 >> tap) & 1u) ? d : 0)) * 16);
    };
    auto store_b = [&](int s, int par) __attribute__((always_inline)) {  // masked hand-over of k-step s into slot s % 3
        const int tap = s / KSUB;
        const sp_f16x8 zero = (sp_f16x8){0, 0, 0, 0, 0, 0, 0, 0};
        *(sp_f16x8*)(&lds[s % 3][f0][lane * 16]) = ((in0 >> tap) & 1u) ? rg[par][0] : zero;
        if (wave < 2) *(sp_f16x8*)(&lds[s % 3][f1][lane * 16]) = ((in1 >> tap) & 1u) ? rg[par][1] : zero;
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
    load_b(0, 0);
    load_b(1, 1);
#pragma unroll
    for (int s = 0; s < RA - 1; ++s) load_a(s, s);
    store_b(0, 0);
    CV_BARRIER();
    cp_for_each([&](auto SC) __attribute__((always_inline)) {
        constexpr int s = decltype(SC)::value, h = (s % KSUB) / KH;
        sp_f16x8 bh[NJ], bl[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bh[j] = *(const sp_f16x8*)(&lds[s % 3][j][lane * 16]);
            bl[j] = *(const sp_f16x8*)(&lds[s % 3][NJ + j][lane * 16]);
        }
        if constexpr (s + 1 < NST) store_b(s + 1, (s + 1) & 1);
        if constexpr (s + 2 < NST) load_b(s + 2, s & 1);
        if constexpr (s + RA - 1 < NST) load_a(s + RA - 1, (s + RA - 1) % RA);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                am[h][t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[s % RA][0][t], bh[j], am[h][t][j], 0, 0, 0);
                ac[h][t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[s % RA][0][t], bl[j], ac[h][t][j], 0, 0, 0);
                ac[h][t][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ra[s % RA][1][t], bh[j], ac[h][t][j], 0, 0, 0);
            }
        if constexpr (s + 1 < NST) CV_BARRIER();
        __builtin_amdgcn_sched_barrier(0);
    }, typename CpMakeSeq<NST>::type{});
    spg_epilogue<RES, NT, NJ, HALVES>(am, ac, pos, live, res, y, board, yplane, P2, cg, kg, lo_clamp, range);
}
#endif
