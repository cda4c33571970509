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
template <bool RES, int KSUB, int NT, int NJ, int HALVES> __global__ void __launch_bounds__(256)
k_conv3x3_spg(const unsigned char* __restrict__ x, const _Float16* __restrict__ w, const float* __restrict__ bias, const unsigned char* __restrict__ res,
              unsigned char* __restrict__ y, int nboards, int S, int C, int relu, unsigned* range) {
    constexpr int CIN = 32 * KSUB, NCHI = 4 * KSUB, NST = 9 * KSUB, R = 3, KH = KSUB / HALVES;
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
#endif
