// az_conv64.h -- weight-stationary 3x3 convolution of a 64-filter residual tower on 17x17 planes (the reference's 13x13 Gomoku
// network: stem padding 3 turns 13x13 boards into 17x17 planes, alpha_zero/core/network.py:101-105; BASELINE config C2: 6 x 64).
//     y = relu(conv3x3(x, w) + bias [+ residual])      x, y, residual in the tiled layout [board][8 chunks][289 positions][8 ch] bf16
// Same ideas as k_conv3x3_tiled (az_conv.h), re-balanced for 64 channels:
//   * one persistent 256-thread workgroup per CU, one wave per SIMD; EVERY wave holds the whole filter bank (64 couts x 576 (tap, cin)
//     = 288 registers: 64 A fragments in AGPRs + 8 in VGPRs) and the waves split the board's positions, so a B fragment is read from
//     LDS by exactly one wave and feeds four MFMAs (the four 16-cout tiles).
//   * v_mfma_f32_16x16x32_bf16: a column tile is 16 positions; 289 positions = 20 column tiles (5 per wave; 31 slots of 320 repeat a
//     position, never masked).  k-step = one tap x 32 input channels (18 per tile).
//   * LDS image per 8-channel chunk: 352 cells of 16 B, cell(y, x) = 19 + 18 y + x, zero cells between rows: a tap (dy, dx) is the
//     constant cell offset 18 dy + dx, every B address is a per-lane base + an immediate; no VALU in the k-loop.  The strip size is a
//     multiple of 256 B, so the four 8-channel groups of a B fragment (lanes l, l+16, l+32, l+48 read neighbouring strips) fall on the
//     same banks and a ds_read_b128 lane group = the 16 positions of the column tile, chosen (cw64_map) distinct mod 16: conflict-free.
//   * LDS-DMA double buffering of the next board under the MFMAs, one barrier per board, epilogue (residual, one bf16 rounding, ReLU)
//     straight from the accumulators: the D layout hands a lane 4 consecutive couts of its position = an 8-byte slot.
#pragma once
#include "az_conv.h"

#if defined(__HIPCC__)
#define C6_S 17
#define C6_P2 (C6_S * C6_S)                 // 289 positions per board = per tile
#define C6_C 64
#define C6_NCH 8                            // channel chunks
#define C6_GBLK (C6_P2 * 16)                // 4,624 B: one chunk block of a board in global memory
#define C6_TILE (C6_NCH * C6_GBLK)          // 36,992 B per board
#define C6_PITCH 18
#define C6_CELL0 19                         // cell of (0, 0); (-1, -1) is cell 0
#define C6_CELLS 352                        // 19 + 17 * 18 + 19 = 344, rounded up to a multiple of 16 (strip = multiple of 256 B)
#define C6_LBLK (C6_CELLS * 16)             // 5,632 B per chunk strip
#define C6_NCT 20                           // column tiles of 16 positions

typedef __attribute__((ext_vector_type(4))) float c6_f32x4;

// (column tile, lane & 15) -> position / cell: column tile k takes the k-th position cell of every residue class mod 16 (no class
// has more than 20 of the 289 cells); the 31 unfilled slots repeat the first position of a residue the tile still lacks.
struct Cw64Map {
    unsigned short cell[C6_NCT * 16], pos[C6_NCT * 16];
};
constexpr Cw64Map cw64_make_map() {
    Cw64Map m{};
    int cnt[16] = {}, fill[C6_NCT] = {};
    bool used[C6_NCT][16] = {};
    for (int p = 0; p < C6_P2; ++p) {
        const int cell = C6_CELL0 + C6_PITCH * (p / C6_S) + p % C6_S, r = cell & 15, k = cnt[r]++;
        m.cell[k * 16 + fill[k]] = (unsigned short)cell;
        m.pos[k * 16 + fill[k]] = (unsigned short)p;
        fill[k]++;
        used[k][r] = true;
    }
    for (int k = 0; k < C6_NCT; ++k)
        for (int r = 0; r < 16 && fill[k] < 16; ++r) {
            if (used[k][r]) continue;
            for (int p = 0; p < C6_P2; ++p) {
                const int cell = C6_CELL0 + C6_PITCH * (p / C6_S) + p % C6_S;
                if ((cell & 15) == r) {
                    m.cell[k * 16 + fill[k]] = (unsigned short)cell;
                    m.pos[k * 16 + fill[k]] = (unsigned short)p;
                    fill[k]++;
                    break;
                }
            }
        }
    return m;
}
static __device__ const Cw64Map cw64_map = cw64_make_map();

__device__ __forceinline__ void c6_mfma_a(c6_f32x4& acc, const cv_bf16x8& wa, const cv_bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(wa), "v"(b));
}
__device__ __forceinline__ void c6_mfma_v(c6_f32x4& acc, const cv_bf16x8& wa, const cv_bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(b));
}
__device__ __forceinline__ void c6_mfma_ac(c6_f32x4& acc, const cv_bf16x8& wa, const cv_bf16x8& b, const c6_f32x4& c) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "a"(wa), "v"(b), "v"(c));
}

// NCH = input-channel chunks: 8 = tower layer (64 -> 64, input = a tiled 17x17 board); 4 = stem (17 planes padded to 32 -> 64): the
// input is the engine's tiled 13x13 feature board, embedded at offset (2, 2) of the zero 17x17 plane by the DMA masks -- a pad-3
// convolution of the 13x13 board IS the pad-1 convolution of that embedded plane (network.py:101-105).
template <bool RES, int NCH> __global__ void __launch_bounds__(CW_THREADS, 1)
k_conv3x3_t64(const unsigned char* __restrict__ x, const unsigned short* __restrict__ w, const float* __restrict__ bias,
              const unsigned char* __restrict__ res, unsigned char* __restrict__ y, int ntiles, int relu) {
    constexpr int HALVES = NCH / 4, NSTEP = 9 * HALVES;   // k-steps (32 input channels each)
    constexpr int LBUF = NCH * C6_LBLK;                     // LDS buffer: NCH strips
    constexpr int IN_S = NCH == 8 ? C6_S : 13, IN_OFF = NCH == 8 ? 0 : 2;  // input board size and its offset in the 17x17 plane
    constexpr int IN_GBLK = IN_S * IN_S * 16, XTILE = NCH * IN_GBLK;
    constexpr int SPW = NCH / 4, NPIECE = 6 * SPW;          // strips and DMA pieces per wave
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * LBUF];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    for (int i = tid; i < 2 * LBUF / 16; i += CW_THREADS) *(cv_u32x4*)(lds + i * 16) = (cv_u32x4){0u, 0u, 0u, 0u};

    // A fragments: step s = (tap, input half), cout tile q: lane (cout = 16 q + l15, cin = 32 half + 8 kg .. + 8)
    cv_bf16x8 wf[NSTEP * 4];
#pragma unroll
    for (int t = 0; t < NSTEP * 4; ++t) {
        const int s = t >> 2, q = t & 3;
        wf[t] = *(const cv_bf16x8*)(w + ((size_t)((s / HALVES) * C6_C + q * 16 + l15)) * (8 * NCH) + (s % HALVES) * 32 + kg * 8);
    }
    c6_f32x4 bv[4];  // bias in the D layout (rows = couts 16 q + 4 kg + e): the C operand of the first k-step
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[q][e] = bias[q * 16 + 4 * kg + e];
    const unsigned lo16 = relu ? 0u : 0x80008000u;

    // LDS-DMA plan: a strip is 6 pieces of 64 cells (the last one holds 4 position cells); wave q moves strips SPW q .. SPW q + SPW - 1
    unsigned dsrc[6];
    unsigned long long dmask[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int cell = 64 * i + lane, k = cell - C6_CELL0, yy = k / C6_PITCH - IN_OFF, xx = k % C6_PITCH - IN_OFF;
        const bool ok = k >= 0 && yy >= 0 && xx >= 0 && yy < IN_S && xx < IN_S;
        dsrc[i] = (unsigned)((yy * IN_S + xx) * 16);
        dmask[i] = __builtin_amdgcn_ballot_w64(ok);
    }
    auto dma_piece = [&](const unsigned char* src, unsigned dstbuf, bool live, int i) {  // i in [0, NPIECE): strip SPW wave + i / 6, piece i % 6
        const int c = SPW * wave + i / 6, pc = i % 6;
        const unsigned long long base = (unsigned long long)(src + (size_t)c * IN_GBLK);
        const unsigned long long mask = live ? dmask[pc] : 0ull;
        const unsigned dst = dstbuf + (unsigned)(c * C6_LBLK + pc * 1024);
        asm volatile("s_mov_b64 exec, %0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1"
                     :
                     : "s"(mask), "s"(dst), "v"(dsrc[pc]), "s"(base)
                     : "memory");
    };

    // this lane's 5 column tiles (wave's tiles: 4 j + wave): LDS byte offset of the (-1, -1) neighbour in its own 8-channel group
    // (low 16 bits) and the position (high 16 bits)
    unsigned lmap[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const int idx = (4 * j + wave) * 16 + l15;
        lmap[j] = (unsigned)((cw64_map.cell[idx] - C6_CELL0) * 16 + kg * C6_LBLK) | ((unsigned)cw64_map.pos[idx] << 16);
    }
    cv_bf16x8 bb[2][5];  // B fragments: k-step s lives in slot s & 1
    auto load_step = [&](const unsigned char* const (&bp)[5], int s) {  // tap s / HALVES = constant cell offset, input half = 4 strips on
        const int tap = s / HALVES;
        const int off = ((tap / 3) * C6_PITCH + (tap % 3)) * 16 + (s % HALVES) * (4 * C6_LBLK);
#pragma unroll
        for (int j = 0; j < 5; ++j) bb[s & 1][j] = *(const cv_bf16x8*)(bp[j] + off);
    };

    {   // first board: all pieces at once
        const unsigned char* src = x + (size_t)blockIdx.x * XTILE;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) dma_piece(src, lds0, true, i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CV_BARRIER();
    }
    {
        const unsigned char* bp0[5] = {lds + (lmap[0] & 0xffffu), lds + (lmap[1] & 0xffffu), lds + (lmap[2] & 0xffffu), lds + (lmap[3] & 0xffffu),
                                       lds + (lmap[4] & 0xffffu)};
        load_step(bp0, 0);
    }
#pragma unroll
    for (int t = 0; t < NSTEP * 4; ++t) {  // the compiler's wait for the weight loads belongs in front of the loop (see az_conv.h)
        if (t < 64) asm volatile("" : : "a"(wf[t]));
        else asm volatile("" : : "v"(wf[t]));
    }
    asm volatile("" : : "v"(bv[0]), "v"(bv[3]), "v"(lmap[0]), "v"(lmap[4]), "v"(dsrc[0]), "v"(dsrc[5]));

    int it = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        const unsigned char* Xs = lds + buf * LBUF;
        const unsigned char* Xn = lds + (buf ^ 1) * LBUF;
        const bool has_next = tile + (int)gridDim.x < ntiles;
        const unsigned char* nsrc = x + (size_t)(has_next ? tile + (int)gridDim.x : tile) * XTILE;
        const unsigned ndst = lds0 + (unsigned)((buf ^ 1) * LBUF);
        const unsigned char* rbase = RES ? res + (size_t)tile * C6_TILE : nullptr;
        unsigned char* ybase = y + (size_t)tile * C6_TILE;
        const unsigned char* bp[5];
        cv_u32x2 rr[5][4];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            bp[j] = Xs + (lmap[j] & 0xffffu);
            if (RES) {
                const unsigned gq = (lmap[j] >> 16) * 16u + (unsigned)((kg >> 1) * C6_GBLK + (kg & 1) * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) rr[j][q] = *(const cv_u32x2*)(rbase + (2 * q) * C6_GBLK + gq);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        c6_f32x4 acc[5][4];
#pragma unroll
        for (int t = 0; t < NSTEP; ++t) {  // the fragments of step 0 are already in flight (issued before the previous epilogue)
            if (t + 1 < NSTEP) load_step(bp, t + 1);
#pragma unroll
            for (int j = 0; j < 5; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (t == 0) c6_mfma_ac(acc[j][q], wf[q], bb[0][j], bv[q]);
                    else if (t * 4 + q < 64) c6_mfma_a(acc[j][q], wf[t * 4 + q], bb[t & 1][j]);
                    else c6_mfma_v(acc[j][q], wf[t * 4 + q], bb[t & 1][j]);
                }
            if (2 * t + 1 < NPIECE) {  // the next board's pieces ride in the shadow of the first k-steps' MFMAs
                dma_piece(nsrc, ndst, has_next, 2 * t);
                dma_piece(nsrc, ndst, has_next, 2 * t + 1);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // everything this wave has in flight is old (pieces issued >= 12 k-steps ago, the previous board's stores): after the
        // barrier every wave's pieces of the next board have landed and this buffer may be overwritten
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CV_BARRIER();
        {
            const unsigned char* bpn[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) bpn[j] = Xn + (lmap[j] & 0xffffu);
            load_step(bpn, 0);  // the next board's first fragments fly while the epilogue runs
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 15\n\ts_nop 15"
                     : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]),
                       "+v"(acc[1][3]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]), "+v"(acc[3][0]), "+v"(acc[3][1]),
                       "+v"(acc[3][2]), "+v"(acc[3][3]), "+v"(acc[4][0]), "+v"(acc[4][1]), "+v"(acc[4][2]), "+v"(acc[4][3]));
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            // cout tile q, lane group kg: couts 16 q + 4 kg .. + 4 = chunk 2 q + kg / 2, half kg % 2
            const unsigned gq = (lmap[j] >> 16) * 16u + (unsigned)((kg >> 1) * C6_GBLK + (kg & 1) * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v0 = acc[j][q][0], v1 = acc[j][q][1], v2 = acc[j][q][2], v3 = acc[j][q][3];
                if (RES) {
                    const cv_u32x2 r2 = rr[j][q];
                    v0 += cv_bf16_lo(r2.x);
                    v1 += cv_bf16_hi(r2.x);
                    v2 += cv_bf16_lo(r2.y);
                    v3 += cv_bf16_hi(r2.y);
                }
                *(cv_u32x2*)(ybase + (2 * q) * C6_GBLK + gq) =
                    (cv_u32x2){cw_pk_max_i16(cw_pk_bf16(v0, v1), lo16), cw_pk_max_i16(cw_pk_bf16(v2, v3), lo16)};
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
#endif  // __HIPCC__
