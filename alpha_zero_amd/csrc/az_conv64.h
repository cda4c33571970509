// az_conv64.h -- weight-stationary 3x3 convolutions of a 64-filter residual tower, two tile geometries:
//   C6Geo<17>: 17x17 planes, one board per tile (the reference's 13x13 Gomoku network: stem padding 3 turns 13x13 boards into 17x17
//              planes, alpha_zero/core/network.py:101-105; BASELINE config C2: 6 x 64)
//   C6Geo<9> : 9x9 planes, three boards per tile (the reference's second shipped Go run, 9x9 with 64 filters, logs/go/9x9_12b64/run.log:1)
//     y = relu(conv3x3(x, w) + bias [+ residual])      x, y, residual in the tiled layout [tile][8 chunks][P2 positions][8 ch] bf16
// Same ideas as k_conv3x3_tiled (az_conv.h), re-balanced for 64 channels:
//   * one persistent 256-thread workgroup per CU, one wave per SIMD; EVERY wave holds the whole filter bank (64 couts x 576 (tap, cin)
//     = 288 registers: 64 A fragments in AGPRs + 8 in VGPRs) and the waves split the tile's positions, so a B fragment is read from
//     LDS by exactly one wave and feeds four MFMAs (the four 16-cout tiles).
//   * v_mfma_f32_16x16x32_bf16: a column tile is 16 positions; 289 positions = 20 column tiles (5 per wave; 31 slots of 320 repeat a
//     position, never masked), 243 positions = 16 column tiles (4 per wave, 13 repeats).  k-step = one tap x 32 input channels.
//   * LDS image per 8-channel chunk: a strip of 16-byte cells with zero cells between board rows (17x17: cell(y, x) = 20 + 19 y + x;
//     9x9: cell(b, y, x) = 11 + 101 b + 10 y + x as in az_conv.h): a tap (dy, dx) is the constant cell offset PITCH dy + dx, every B
//     address is a per-lane base + an immediate; no VALU in the k-loop.  The strip size is a multiple of 256 B, so the four 8-channel
//     groups of a B fragment (lanes l, l+16, l+32, l+48 read neighbouring strips) fall on the same banks and a ds_read_b128 lane group
//     = the 16 positions of the column tile, chosen (Cw64Map) distinct mod 16: conflict-free.
//   * LDS-DMA double buffering of the next tile under the MFMAs, one barrier per tile, epilogue (residual, one bf16 rounding, ReLU)
//     straight from the accumulators: the D layout hands a lane 4 consecutive couts of its position = an 8-byte slot.
// k_resblock64 (below) fuses the two convolutions of a ResNetBlock on the same geometry and keeps the intermediate activation in LDS.
#pragma once
#include "az_conv.h"

#if defined(__HIPCC__)
#define C6_C 64
#define C6_NCH 8                            // channel chunks

template <int S_> struct C6Geo;
template <> struct C6Geo<17> {
    // Row pitch 19 (two zero cells between board rows), not 18: cell mod 16 = (4 + 3 y + x) mod 16, and since 3 is coprime to 16 the
    // residue a row holds twice (x = 0 and x = 16) moves through all 16 classes -- every class has 18 cells, one has 19.  The
    // residue-class map then fills 18 column tiles completely and leaves ONE position for tile 18 (289 = 16 * 18 + 1).  (With pitch
    // 18 the doubled residues are all odd: classes of 17 / 19 / 20 cells, 17 positions spread over three part-filled tiles.)
    static constexpr int S = 17, TB = 1, P2 = 289, PITCH = 19, CELL0 = 20;
    static constexpr int CELLS = 368;  // 20 + 17 * 19 - 2 + 20 = 361 with the tap reach, rounded up to a multiple of 16 (strip = multiple of 256 B)
    static constexpr int NCT = 20;     // column tiles of 16 positions (k_conv3x3_t64: 5 per wave; tile 19 only repeats positions)
    static constexpr int NCT_REAL = 19;  // tiles that hold a position of their own: 18 full ones + tile 18 with one
    static constexpr int STEM_S = 13, STEM_OFF = 2;  // the stem's input board and its offset in the plane (pad-3 stem, network.py:101-105)
    static constexpr int cell_of(int p) { return CELL0 + PITCH * (p / S) + p % S; }
    // position index inside an input tile of in_s x in_s boards embedded at (off, off) of the plane that `cell` belongs to, or -1
    static constexpr int pos_of_cell(int cell, int in_s, int off) {
        const int k = cell - CELL0;
        if (k < 0) return -1;
        const int yy = k / PITCH - off, xx = k % PITCH - off;
        return (yy >= 0 && xx >= 0 && yy < in_s && xx < in_s) ? yy * in_s + xx : -1;
    }
};
template <> struct C6Geo<9> {
    static constexpr int S = 9, TB = 3, P2 = 243, PITCH = 10, BPITCH = 101, CELL0 = 11;
    static constexpr int CELLS = 320;  // 11 + 2 * 101 + 100 = 313, rounded up to a multiple of 16
    static constexpr int NCT = 16;
    static constexpr int NCT_REAL = 16;  // 243 = 16 * 15 + 3: tile 15 holds three positions of its own
    static constexpr int STEM_S = 9, STEM_OFF = 0;
    static constexpr int cell_of(int p) { return CELL0 + BPITCH * (p / 81) + PITCH * ((p % 81) / 9) + p % 9; }
    static constexpr int pos_of_cell(int cell, int in_s, int off) {
        const int k = cell - CELL0;
        if (k < 0 || in_s != S || off != 0) return -1;
        const int b = k / BPITCH, r = k % BPITCH, yy = r / PITCH, xx = r % PITCH;
        return (b < TB && yy < S && xx < S) ? b * 81 + yy * 9 + xx : -1;
    }
};

typedef __attribute__((ext_vector_type(4))) float c6_f32x4;

// (column tile, lane & 15) -> position / cell: column tile k takes the k-th position cell of every residue class mod 16 (no class
// has more than NCT of the tile's cells); the unfilled slots repeat the first position of a residue the tile still lacks.
template <class G> struct Cw64Map {
    unsigned short cell[G::NCT * 16], pos[G::NCT * 16];
    int real_tiles;  // column tiles that hold at least one position of their own (the later ones only repeat positions)
    bool ok;
};
template <class G> constexpr Cw64Map<G> cw64_make_map() {
    Cw64Map<G> m{};
    int cnt[16] = {}, fill[G::NCT] = {};
    bool used[G::NCT][16] = {};
    bool ok = true;
    for (int p = 0; p < G::P2; ++p) {
        const int cell = G::cell_of(p), r = cell & 15, k = cnt[r]++;
        if (k >= G::NCT || G::pos_of_cell(cell, G::S, 0) != p) {
            ok = false;
            continue;
        }
        m.cell[k * 16 + fill[k]] = (unsigned short)cell;
        m.pos[k * 16 + fill[k]] = (unsigned short)p;
        fill[k]++;
        used[k][r] = true;
        if (k + 1 > m.real_tiles) m.real_tiles = k + 1;
    }
    for (int k = 0; k < G::NCT; ++k)
        for (int r = 0; r < 16 && fill[k] < 16; ++r) {
            if (used[k][r]) continue;
            for (int p = 0; p < G::P2; ++p) {
                const int cell = G::cell_of(p);
                if ((cell & 15) == r) {
                    m.cell[k * 16 + fill[k]] = (unsigned short)cell;
                    m.pos[k * 16 + fill[k]] = (unsigned short)p;
                    fill[k]++;
                    used[k][r] = true;
                    break;
                }
            }
        }
    for (int k = 0; k < G::NCT; ++k) {  // every column tile: 16 slots, 16 distinct residues (= conflict-free ds_read_b128 lane groups)
        bool seen[16] = {};
        if (fill[k] != 16) ok = false;
        for (int s = 0; s < 16; ++s) {
            const int r = m.cell[k * 16 + s] & 15;
            if (seen[r]) ok = false;
            seen[r] = true;
        }
    }
    m.ok = ok;
    return m;
}
static __device__ const Cw64Map<C6Geo<17>> cw64_map17 = cw64_make_map<C6Geo<17>>();
static __device__ const Cw64Map<C6Geo<9>> cw64_map9 = cw64_make_map<C6Geo<9>>();
static_assert(cw64_make_map<C6Geo<17>>().ok && cw64_make_map<C6Geo<9>>().ok, "column-tile maps must cover every position conflict-free");
static_assert(cw64_make_map<C6Geo<17>>().real_tiles == C6Geo<17>::NCT_REAL && cw64_make_map<C6Geo<9>>().real_tiles == C6Geo<9>::NCT_REAL,
              "k_resblock64 computes the column tiles [0, NCT_REAL): every position must live in one of them");
template <class G> __device__ __forceinline__ const Cw64Map<G>& cw64_map() {
    if constexpr (G::S == 17) return cw64_map17;
    else return cw64_map9;
}

__device__ __forceinline__ void c6_mfma_a(c6_f32x4& acc, const cv_bf16x8& wa, const cv_bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(wa), "v"(b));
}
__device__ __forceinline__ void c6_mfma_v(c6_f32x4& acc, const cv_bf16x8& wa, const cv_bf16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(b));
}
__device__ __forceinline__ void c6_mfma_ac(c6_f32x4& acc, const cv_bf16x8& wa, const cv_bf16x8& b, const c6_f32x4& c) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %3" : "=&v"(acc) : "a"(wa), "v"(b), "v"(c));
}
// the inline-asm MFMAs are opaque to the compiler's hazard recognizer: wait states before VALU reads of their results
template <int NJ> __device__ __forceinline__ void c6_settle(c6_f32x4 (&acc)[NJ][4]) {
    static_assert(NJ == 4 || NJ == 5, "column tiles per wave");
    if constexpr (NJ == 5)
        asm volatile("s_nop 15\n\ts_nop 15"
                     : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]),
                       "+v"(acc[1][3]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]), "+v"(acc[3][0]), "+v"(acc[3][1]),
                       "+v"(acc[3][2]), "+v"(acc[3][3]), "+v"(acc[4][0]), "+v"(acc[4][1]), "+v"(acc[4][2]), "+v"(acc[4][3]));
    else
        asm volatile("s_nop 15\n\ts_nop 15"
                     : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]),
                       "+v"(acc[1][3]), "+v"(acc[2][0]), "+v"(acc[2][1]), "+v"(acc[2][2]), "+v"(acc[2][3]), "+v"(acc[3][0]), "+v"(acc[3][1]),
                       "+v"(acc[3][2]), "+v"(acc[3][3]));
}

// NCH = input-channel chunks: 8 = tower layer (64 -> 64, input = a tile of the tower); 4 = stem (17 planes padded to 32 -> 64): the
// input is the engine's tiled feature tile; for C6Geo<17> that is a 13x13 board, embedded at offset (2, 2) of the zero 17x17 plane by
// the DMA masks -- a pad-3 convolution of the 13x13 board IS the pad-1 convolution of that embedded plane (network.py:101-105).
template <class G, bool RES, int NCH> __global__ void __launch_bounds__(CW_THREADS, 1)
k_conv3x3_t64(const unsigned char* __restrict__ x, const unsigned short* __restrict__ w, const float* __restrict__ bias,
              const unsigned char* __restrict__ res, unsigned char* __restrict__ y, int ntiles, int relu) {
    constexpr int HALVES = NCH / 4, NSTEP = 9 * HALVES;    // k-steps (32 input channels each)
    constexpr int NJ = G::NCT / 4;                          // column tiles per wave
    constexpr int LBLK = G::CELLS * 16, LBUF = NCH * LBLK;  // chunk strip and buffer in LDS
    constexpr int GBLK = G::P2 * 16, TILE = C6_NCH * GBLK;  // chunk block and tile of the tower in global memory
    constexpr int IN_S = NCH == 8 ? G::S : G::STEM_S, IN_OFF = NCH == 8 ? 0 : G::STEM_OFF;  // input board size and its offset in the plane
    constexpr int IN_GBLK = G::TB * IN_S * IN_S * 16, XTILE = NCH * IN_GBLK;
    constexpr int NP = (G::CELLS + 63) / 64;                // DMA pieces of 64 cells per strip
    constexpr int SPW = NCH / 4, NPIECE = NP * SPW;         // strips and DMA pieces per wave
    static_assert(2 * NSTEP >= NPIECE, "the next tile's pieces ride in the k-steps");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * LBUF];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    for (int i = tid; i < 2 * LBUF / 16; i += CW_THREADS) *(cv_u32x4*)(lds + i * 16) = (cv_u32x4){0u, 0u, 0u, 0u};
    CV_BARRIER();  // the zero cells are final before any LDS-DMA piece can land

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

    // LDS-DMA plan: a strip is NP pieces of 64 cells; wave q moves strips SPW q .. SPW q + SPW - 1
    unsigned dsrc[NP];
    unsigned long long dmask[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = G::pos_of_cell(64 * i + lane, IN_S, IN_OFF);
        dsrc[i] = (unsigned)((p < 0 ? 0 : p) * 16);
        dmask[i] = __builtin_amdgcn_ballot_w64(p >= 0);
    }
    auto dma_piece = [&](const unsigned char* src, unsigned dstbuf, bool live, int i) {  // i in [0, NPIECE): strip SPW wave + i / NP, piece i % NP
        const int c = SPW * wave + i / NP, pc = i % NP;
        const unsigned long long base = (unsigned long long)(src + (size_t)c * IN_GBLK);
        const unsigned long long mask = live ? dmask[pc] : 0ull;
        const unsigned dst = dstbuf + (unsigned)(c * LBLK + pc * 1024);
        asm volatile("s_mov_b64 exec, %0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1"
                     :
                     : "s"(mask), "s"(dst), "v"(dsrc[pc]), "s"(base)
                     : "memory");
    };

    // this lane's NJ column tiles (wave's tiles: 4 j + wave): LDS byte offset of the (-1, -1) neighbour in its own 8-channel group
    // (low 16 bits) and the position (high 16 bits)
    unsigned lmap[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int idx = (4 * j + wave) * 16 + l15;
        lmap[j] = (unsigned)((cw64_map<G>().cell[idx] - G::CELL0) * 16 + kg * LBLK) | ((unsigned)cw64_map<G>().pos[idx] << 16);
    }
    cv_bf16x8 bb[2][NJ];  // B fragments: k-step s lives in slot s & 1
    auto load_step = [&](const unsigned char* const (&bp)[NJ], int s) {  // tap s / HALVES = constant cell offset, input half = 4 strips on
        const int tap = s / HALVES;
        const int off = ((tap / 3) * G::PITCH + (tap % 3)) * 16 + (s % HALVES) * (4 * LBLK);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bb[s & 1][j] = *(const cv_bf16x8*)(bp[j] + off);
    };

    {   // first tile: all pieces at once
        const unsigned char* src = x + (size_t)blockIdx.x * XTILE;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) dma_piece(src, lds0, true, i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CV_BARRIER();
    }
    {
        const unsigned char* bp0[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bp0[j] = lds + (lmap[j] & 0xffffu);
        load_step(bp0, 0);
    }
#pragma unroll
    for (int t = 0; t < NSTEP * 4; ++t) {  // the compiler's wait for the weight loads belongs in front of the loop (see az_conv.h)
        if (t < 64) asm volatile("" : : "a"(wf[t]));
        else asm volatile("" : : "v"(wf[t]));
    }
    asm volatile("" : : "v"(bv[0]), "v"(bv[3]), "v"(lmap[0]), "v"(lmap[NJ - 1]), "v"(dsrc[0]), "v"(dsrc[NP - 1]));

    int it = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        const unsigned char* Xs = lds + buf * LBUF;
        const unsigned char* Xn = lds + (buf ^ 1) * LBUF;
        const bool has_next = tile + (int)gridDim.x < ntiles;
        const unsigned char* nsrc = x + (size_t)(has_next ? tile + (int)gridDim.x : tile) * XTILE;
        const unsigned ndst = lds0 + (unsigned)((buf ^ 1) * LBUF);
        const unsigned char* rbase = RES ? res + (size_t)tile * TILE : nullptr;
        unsigned char* ybase = y + (size_t)tile * TILE;
        const unsigned char* bp[NJ];
        cv_u32x2 rr[NJ][4];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bp[j] = Xs + (lmap[j] & 0xffffu);
            if (RES) {
                const unsigned gq = (lmap[j] >> 16) * 16u + (unsigned)((kg >> 1) * GBLK + (kg & 1) * 8);
#pragma unroll
                for (int q = 0; q < 4; ++q) rr[j][q] = *(const cv_u32x2*)(rbase + (2 * q) * GBLK + gq);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        c6_f32x4 acc[NJ][4];
#pragma unroll
        for (int t = 0; t < NSTEP; ++t) {  // the fragments of step 0 are already in flight (issued before the previous epilogue)
            if (t + 1 < NSTEP) load_step(bp, t + 1);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (t == 0) c6_mfma_ac(acc[j][q], wf[q], bb[0][j], bv[q]);
                    else if (t * 4 + q < 64) c6_mfma_a(acc[j][q], wf[t * 4 + q], bb[t & 1][j]);
                    else c6_mfma_v(acc[j][q], wf[t * 4 + q], bb[t & 1][j]);
                }
            // the next tile's pieces ride in the shadow of the first k-steps' MFMAs
            if (2 * t < NPIECE) dma_piece(nsrc, ndst, has_next, 2 * t);
            if (2 * t + 1 < NPIECE) dma_piece(nsrc, ndst, has_next, 2 * t + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        // everything this wave has in flight is old (pieces issued >= 12 k-steps ago, the previous tile's stores): after the
        // barrier every wave's pieces of the next tile have landed and this buffer may be overwritten
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CV_BARRIER();
        {
            const unsigned char* bpn[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bpn[j] = Xn + (lmap[j] & 0xffffu);
            load_step(bpn, 0);  // the next tile's first fragments fly while the epilogue runs
        }
        __builtin_amdgcn_sched_barrier(0);
        c6_settle<NJ>(acc);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            // cout tile q, lane group kg: couts 16 q + 4 kg .. + 4 = chunk 2 q + kg / 2, half kg % 2
            const unsigned gq = (lmap[j] >> 16) * 16u + (unsigned)((kg >> 1) * GBLK + (kg & 1) * 8);
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
                *(cv_u32x2*)(ybase + (2 * q) * GBLK + gq) =
                    (cv_u32x2){cw_pk_max_i16(cw_pk_bf16(v0, v1), lo16), cw_pk_max_i16(cw_pk_bf16(v2, v3), lo16)};
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_resblock64<G, R>: one whole ResNetBlock (alpha_zero/core/network.py:42-82, eval mode, BatchNorm folded) of a 64-filter tower in ONE
// launch:      y = relu(conv3x3(relu(conv3x3(x, w1) + b1), w2) + b2 + x)
// The layer-at-a-time kernel above is HBM-bound by construction (230 flop/B: profiles/r01_pmc_conv64.txt, 4.8 TB/s on the skip
// layers); here the intermediate activation never leaves the CU and the skip comes from the input tile that is already in LDS,
// so a block moves 2 tensor passes (x in, y out) instead of 5.
//   * both filter banks are register-resident: wave (h = wave & 1, ph = wave >> 1) holds couts [32 h, 32 h + 32) of BOTH
//     convolutions (2 x 36 A fragments = 288 registers) and computes them for the column tiles 2 m + ph of position half ph.
//     Its two 16-cout tiles sit in "slots": slot s = cout tile 2 h + (s ^ ph).
//   * 17x17: the residue-class map fills 18 column tiles and leaves ONE position for tile 18 (289 = 16 * 18 + 1).  That tile is
//     SHARED: every wave computes it for its slot-0 cout tile only (the four waves' slot 0 are the four cout tiles), so a wave does
//     9 tiles x 2 slots + 1 = 19 MFMA columns per k-step and convolution instead of the 20 of an even split (-5 % matrix work,
//     perfectly balanced).  9x9: 16 tiles, 8 per wave, no shared tile.
//   * LDS: x double buffer (LDS-DMA of the next tile under this tile's MFMAs) + ONE intermediate image with the same zero-cell
//     layout, written by the first convolution's epilogue (ds_write_b64 of the bf16-rounded ReLU) and read by the second one.
//   * a tile is 2 NU units of 2 column tiles x 2 slots (4 independent accumulators, 18 k-steps of 4 MFMAs; the last unit of a 17x17
//     convolution is the wave's ninth tile + the shared tile's slot 0: 3 MFMAs per k-step); two accumulator sets: the epilogue of
//     unit i - 1 (mid write / skip add, rounding, ReLU, 8-byte global stores) rides in the MFMA stream of unit i, the B-fragment
//     ring (R slots) runs on across units and tiles.  Only the last unit of the first convolution has an exposed epilogue (the
//     intermediate image must be complete before the second convolution starts).
//   * three barriers per tile: B1 (start of unit 1: every wave is done with the previous intermediate image and the previous
//     tile's skip reads -> the image may be overwritten, the next tile's DMA may start), B2 (intermediate image complete), M (start
//     of the last unit, behind a counted vmcnt: the next tile has landed; the ring's prefetch then crosses into it).
//   * arithmetic (MFMA shape, k order, bias as the C operand of the first k-step, one bf16 rounding of the intermediate and of the
//     output) is exactly that of two k_conv3x3_t64 launches: results are bit-identical to the layer-at-a-time path.
template <class G, int R = 4> __global__ void __launch_bounds__(CW_THREADS, 1)
k_resblock64(const unsigned char* __restrict__ x, const unsigned short* __restrict__ w1, const float* __restrict__ b1,
             const unsigned short* __restrict__ w2, const float* __restrict__ b2, unsigned char* __restrict__ y, int ntiles) {
    constexpr bool SHARED = G::NCT_REAL % 2 == 1;           // an odd tile count: the last tile is shared by the four waves
    constexpr int NOWN = G::NCT_REAL / 2;                    // column tiles a wave owns (both slots)
    constexpr int NU = (NOWN + (SHARED ? 1 : 0) + 1) / 2, NUNIT = 2 * NU;  // units per convolution / per tile (2 column tiles each)
    constexpr int KS = 18;                                   // k-steps per unit: 9 taps x 2 halves of 32 input channels
    constexpr int LBLK = G::CELLS * 16, LBUF = C6_NCH * LBLK;
    constexpr int GBLK = G::P2 * 16, TILE = C6_NCH * GBLK;
    constexpr int NP = (G::CELLS + 63) / 64, NPIECE = 2 * NP;  // wave q moves strips 2 q, 2 q + 1
    constexpr int VM_AT_M = 4 * (NU - 2);                    // vector-memory operations younger than the last DMA piece at barrier M: the
                                                             // stores of the second convolution's units 0 .. NU - 3 (riding in units NU + 1 .. 2 NU - 2)
    static_assert(NOWN + (SHARED ? 1 : 0) == NUNIT, "units are pairs of column tiles");
    static_assert((NUNIT * KS) % R == 0, "a tile's k-steps keep the ring phase");
    static_assert(NU >= 4 && NPIECE <= 12, "the next tile's pieces ride in units 1-3");
    static_assert(3 * LBUF <= 160 * 1024, "x double buffer + intermediate image");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[3 * LBUF];
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = wave & 1, ph = wave >> 1;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    for (int i = tid; i < 3 * LBUF / 16; i += CW_THREADS) *(cv_u32x4*)(lds + i * 16) = (cv_u32x4){0u, 0u, 0u, 0u};
    CV_BARRIER();  // the zero cells are final before any LDS-DMA piece can land
    unsigned char* const MID = lds + 2 * LBUF;

    // A fragments: f = 36 conv + 2 s + slot, k-step s = (tap, input half): lane (cout = 16 (2 h + (slot ^ ph)) + l15, cin = 32 half + 8 kg .. + 8)
    cv_bf16x8 wf[72];
#pragma unroll
    for (int f = 0; f < 72; ++f) {
        const int s = (f % 36) >> 1, ct = 2 * h + ((f & 1) ^ ph);
        const unsigned short* w = f < 36 ? w1 : w2;
        wf[f] = *(const cv_bf16x8*)(w + ((size_t)((s >> 1) * C6_C + 16 * ct + l15)) * C6_C + (s & 1) * 32 + kg * 8);
    }
    c6_f32x4 bv[2][2];  // bias in the D layout (rows = couts 16 ct + 4 kg + e): the C operand of a unit's first k-step
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl)
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[c][sl][e] = (c ? b2 : b1)[16 * (2 * h + (sl ^ ph)) + 4 * kg + e];

    unsigned dsrc[NP];
    unsigned long long dmask[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = G::pos_of_cell(64 * i + lane, G::S, 0);
        dsrc[i] = (unsigned)((p < 0 ? 0 : p) * 16);
        dmask[i] = __builtin_amdgcn_ballot_w64(p >= 0);
    }
    auto dma_piece = [&](const unsigned char* src, unsigned dstbuf, bool live, int i) {  // strip 2 wave + i / NP, piece i % NP
        const int c = 2 * wave + i / NP, pc = i % NP;
        const unsigned long long base = (unsigned long long)(src + (size_t)c * GBLK);
        const unsigned long long mask = live ? dmask[pc] : 0ull;
        const unsigned dst = dstbuf + (unsigned)(c * LBLK + pc * 1024);
        asm volatile("s_mov_b64 exec, %0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1"
                     :
                     : "s"(mask), "s"(dst), "v"(dsrc[pc]), "s"(base)
                     : "memory");
    };

    // this lane's column tiles, local tile m = 2 u + j: own tiles m < NOWN = column tile 2 m + ph, then (17x17) the shared tile:
    // B-fragment base offset ((-1, -1) neighbour, own 8-channel group; low 16 bits) and the position (high 16 bits)
    unsigned lmap[NUNIT];
#pragma unroll
    for (int m = 0; m < NUNIT; ++m) {
        const int idx = (m < NOWN ? 2 * m + ph : G::NCT_REAL - 1) * 16 + l15;
        lmap[m] = (unsigned)((cw64_map<G>().cell[idx] - G::CELL0) * 16 + kg * LBLK) | ((unsigned)cw64_map<G>().pos[idx] << 16);
    }
    // the 8-byte slot of (position, couts 16 ct + 4 kg .. + 4), ct = 2 h + (slot ^ ph): chunk 2 ct + kg / 2, half kg % 2
    int wconst[2];       // + (lmap & 0xffff) = offset inside an LDS image
    unsigned gconst[2];  // + 16 position = offset inside a global tile
#pragma unroll
    for (int sl = 0; sl < 2; ++sl) {
        const int chunk = 2 * (2 * h + (sl ^ ph)) + (kg >> 1);
        wconst[sl] = G::CELL0 * 16 + (chunk - kg) * LBLK + (kg & 1) * 8;
        gconst[sl] = (unsigned)(chunk * GBLK + (kg & 1) * 8);
    }

    cv_bf16x8 bb[R][2];
    auto load_step = [&](const unsigned char* p0, const unsigned char* p1, int s, int slot) {  // s = 2 tap + half
        const int tap = s >> 1;
        const int off = ((tap / 3) * G::PITCH + (tap % 3)) * 16 + (s & 1) * (4 * LBLK);
        bb[slot][0] = *(const cv_bf16x8*)(p0 + off);
        bb[slot][1] = *(const cv_bf16x8*)(p1 + off);
    };

    {   // first tile: all pieces at once, then the first fragments
        const unsigned char* src = x + (size_t)blockIdx.x * TILE;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) dma_piece(src, lds0, true, i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CV_BARRIER();
#pragma unroll
        for (int s = 0; s < R - 1; ++s) load_step(lds + (lmap[0] & 0xffffu), lds + (lmap[1] & 0xffffu), s, s);
    }
#pragma unroll
    for (int f = 0; f < 72; ++f) {  // the compiler's wait for the weight loads belongs in front of the loop (see az_conv.h)
        if (f < 64) asm volatile("" : : "a"(wf[f]));
        else asm volatile("" : : "v"(wf[f]));
    }
    asm volatile("" : : "v"(bv[0][0]), "v"(bv[1][1]), "v"(lmap[0]), "v"(lmap[NUNIT - 1]), "v"(dsrc[0]), "v"(dsrc[NP - 1]));

    c6_f32x4 acc[2][2][2];  // [unit parity][column tile j][slot]
    cv_u32x2 rr[2][4];      // skip values of a second-convolution unit: [unit parity][quad q = 2 j + slot]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[a][q >> 1][q & 1] = (c6_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            rr[a][q] = (cv_u32x2){0u, 0u};
        }
    float ev[4];  // one epilogue quad between its phases

    // Epilogues as MICRO-OPS: one or two instructions each, issued one per MFMA gap.  Measured with the ablation harness
    // (tools/probes/block64_abl_probe.hip, profiles/r03_block64_ablation.txt): with the epilogue of a quad issued as one block of 8-16
    // VALU instructions behind a k-step's four 16-cycle MFMAs the riders cost 18.6 % of the launch's cycles (matrix pipe busy 66 % ->
    // 81 % without them) -- a 16x16x32 MFMA leaves room for about one VALU instruction in its shadow, not for a burst.
    unsigned epa = 0, epb = 0;  // packed halves of the quad in flight
    // first-convolution unit, quad q = 2 j + slot: ReLU, one bf16 rounding, 8 bytes into the intermediate image.  6 micro-ops.
    auto epi_mid_op = [&](int set, int m0, int q, int op) {
        const int j = q >> 1, sl = q & 1;
        if (op == 0) epa = cw_pk_bf16(acc[set][j][sl][0], acc[set][j][sl][1]);
        else if (op == 1) epb = cw_pk_bf16(acc[set][j][sl][2], acc[set][j][sl][3]);
        else if (op == 2) epa = cw_pk_max_i16(epa, 0u);
        else if (op == 3) epb = cw_pk_max_i16(epb, 0u);
        else if (op == 4) *(cv_u32x2*)(MID + ((int)(lmap[m0 + j] & 0xffffu) + wconst[sl])) = (cv_u32x2){epa, epb};
    };
    constexpr int MID_OPS = 5;
    // second-convolution unit, quad q: skip add (plain v_add_f32, see az_conv.h), rounding, ReLU, 8-byte store.  9 micro-ops.
    auto epi_out_op = [&](int set, int m0, unsigned char* out, int q, int op, bool store_ok) {
        const int j = q >> 1, sl = q & 1;
        if (op == 0) ev[0] = cw_add_f32(acc[set][j][sl][0], cv_bf16_lo(rr[set][q].x));
        else if (op == 1) ev[1] = cw_add_f32(acc[set][j][sl][1], cv_bf16_hi(rr[set][q].x));
        else if (op == 2) ev[2] = cw_add_f32(acc[set][j][sl][2], cv_bf16_lo(rr[set][q].y));
        else if (op == 3) ev[3] = cw_add_f32(acc[set][j][sl][3], cv_bf16_hi(rr[set][q].y));
        else if (op == 4) epa = cw_pk_bf16(ev[0], ev[1]);
        else if (op == 5) epb = cw_pk_bf16(ev[2], ev[3]);
        else if (op == 6) epa = cw_pk_max_i16(epa, 0u);
        else if (op == 7) epb = cw_pk_max_i16(epb, 0u);
        else if (op == 8) {
            if (store_ok) *(cv_u32x2*)(out + ((lmap[m0 + j] >> 16) * 16u + gconst[sl])) = (cv_u32x2){epa, epb};
        }
    };
    constexpr int OUT_OPS = 9;
    auto epi_mid = [&](int set, int m0, int q) {  // whole quad at once (the exposed epilogue before B2)
#pragma unroll
        for (int op = 0; op < MID_OPS; ++op) epi_mid_op(set, m0, q, op);
    };
    auto epi_out = [&](int set, int m0, unsigned char* out, int q, bool store_ok) {  // whole quad at once (after the last tile)
#pragma unroll
        for (int op = 0; op < OUT_OPS; ++op) epi_out_op(set, m0, out, q, op, store_ok);
    };

    int it = 0;
    unsigned char* yprev = y;  // output base of the previous tile (the epilogue of its last unit runs inside this tile's unit 0)
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        const unsigned char* Xs = lds + buf * LBUF;
        const unsigned char* Xn = lds + (buf ^ 1) * LBUF;
        const bool has_next = tile + (int)gridDim.x < ntiles;
        const unsigned char* nsrc = x + (size_t)(has_next ? tile + (int)gridDim.x : tile) * TILE;
        const unsigned ndst = lds0 + (unsigned)((buf ^ 1) * LBUF);
        unsigned char* ybase = y + (size_t)tile * TILE;
        const bool have_prev = it > 0;

        auto unit = [&](auto IC) __attribute__((always_inline)) {
            constexpr int i = decltype(IC)::value;
            constexpr int conv = i / NU, u = i % NU, set = i & 1, pset = set ^ 1;
            constexpr int pi = (i + NUNIT - 1) % NUNIT, pconv = pi / NU, pu = pi % NU;  // the unit whose epilogue rides here
            constexpr int ni = (i + 1) % NUNIT, nconv = ni / NU, nu = ni % NU;          // the unit the ring runs on into
            constexpr bool cross = (i + 1 == NU);  // first -> second convolution: the intermediate image is not complete yet
            // quads of a unit: 4, or 3 in the unit that ends a 17x17 convolution (column tile j = 1 is the shared tile: slot 0 only)
            constexpr int nq = (SHARED && u == NU - 1) ? 3 : 4, pnq = (SHARED && pu == NU - 1) ? 3 : 4;
            const unsigned char* img = conv ? MID : Xs;
            const unsigned char* nimg = nconv ? MID : (i + 1 == NUNIT ? Xn : Xs);
            const unsigned char* b0 = img + (lmap[2 * u] & 0xffffu);
            const unsigned char* b1p = img + (lmap[2 * u + 1] & 0xffffu);
            const unsigned char* nb0 = nimg + (lmap[2 * nu] & 0xffffu);
            const unsigned char* nb1 = nimg + (lmap[2 * nu + 1] & 0xffffu);
            unsigned char* pout = i == 0 ? yprev : ybase;
            const bool pstore = i > 0 || have_prev;
            cp_for_each([&](auto TC) __attribute__((always_inline)) {
                constexpr int t = decltype(TC)::value;
                // B1: every read of the previous intermediate image / of the previous tile's x buffer has been consumed by this wave
                if constexpr (i == 1 && t == 0) asm volatile("s_barrier" ::: "memory");
                // M: the next tile's pieces are older than the VM_AT_M youngest vector-memory operations of this wave
                if constexpr (i == NUNIT - 1 && t == 0) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(VM_AT_M) : "memory");
                if constexpr (t + R - 1 < KS) load_step(b0, b1p, t + R - 1, (i * KS + t + R - 1) % R);
                else if constexpr (!cross) load_step(nb0, nb1, t + R - 1 - KS, (i * KS + t + R - 1) % R);
                // the unit's MFMAs, each followed by ONE micro-op of the previous unit's epilogue (slot = running MFMA index of the unit)
                cp_for_each([&](auto QC) __attribute__((always_inline)) {
                    constexpr int q = decltype(QC)::value, j = q >> 1, sl = q & 1;
                    constexpr int fb = conv * 36 + 2 * t;
                    if constexpr (t == 0) c6_mfma_ac(acc[set][j][sl], wf[fb + sl], bb[(i * KS + t) % R][j], bv[conv][sl]);
                    else if constexpr (fb < 64) c6_mfma_a(acc[set][j][sl], wf[fb + sl], bb[(i * KS + t) % R][j]);
                    else c6_mfma_v(acc[set][j][sl], wf[fb + sl], bb[(i * KS + t) % R][j]);
                    constexpr int slot = t * nq + q;            // 0 .. 18 nq - 1
                    constexpr int S0 = 6;                       // first slot that may touch the previous unit's accumulators
                    if constexpr (pconv == 0 && i != NU) {      // previous unit: first convolution -> intermediate image
                        constexpr int per = (pnq * MID_OPS + (KS * nq - S0) - 1) / (KS * nq - S0);  // micro-ops per slot (1, or 2 in a short unit)
#pragma unroll
                        for (int k = 0; k < per; ++k) {
                            const int o = (slot - S0) * per + k;
                            if (slot >= S0 && o < pnq * MID_OPS) epi_mid_op(pset, 2 * pu, o / MID_OPS, o % MID_OPS);
                        }
                    } else if constexpr (pconv == 1) {
                        constexpr int per = (pnq * OUT_OPS + (KS * nq - S0) - 1) / (KS * nq - S0);
#pragma unroll
                        for (int k = 0; k < per; ++k) {
                            const int o = (slot - S0) * per + k;
                            if (slot >= S0 && o < pnq * OUT_OPS) epi_out_op(pset, 2 * pu, pout, o / OUT_OPS, o % OUT_OPS, pstore);
                        }
                    }
                    if constexpr (conv == 1 && slot < nq) {  // this unit's skip values from the resident x tile (used by its epilogue inside the next unit)
                        rr[set][slot] = *(const cv_u32x2*)(Xs + ((int)(lmap[2 * u + (slot >> 1)] & 0xffffu) + wconst[slot & 1]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }, typename CpMakeSeq<nq>::type{});
                if constexpr (i >= 1 && i <= 3 && (t == 2 || t == 6 || t == 11 || t == 15)) {  // the next tile's DMA pieces
                    constexpr int p = (i - 1) * 4 + (t == 2 ? 0 : t == 6 ? 1 : t == 11 ? 2 : 3);
                    if constexpr (p < NPIECE) dma_piece(nsrc, ndst, has_next, p);
                }
                __builtin_amdgcn_sched_barrier(0);
            }, typename CpMakeSeq<KS>::type{});
            if constexpr (cross) {
                // the last unit of the first convolution: exposed epilogue, then B2 (intermediate image complete) and the ring restarts
                asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[set][0][0]), "+v"(acc[set][0][1]), "+v"(acc[set][1][0]), "+v"(acc[set][1][1]));
#pragma unroll
                for (int q = 0; q < nq; ++q) epi_mid(set, 2 * u, q);
                CV_BARRIER();
#pragma unroll
                for (int s = 0; s < R - 1; ++s) load_step(MID + (lmap[0] & 0xffffu), MID + (lmap[1] & 0xffffu), s, ((i + 1) * KS + s) % R);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        cp_for_each(unit, typename CpMakeSeq<NUNIT>::type{});
        yprev = ybase;
    }
    // epilogue of the very last unit (second convolution, accumulator set 1)
    if (it > 0) {
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[1][0][0]), "+v"(acc[1][0][1]), "+v"(acc[1][1][0]), "+v"(acc[1][1][1]));
#pragma unroll
        for (int q = 0; q < (SHARED ? 3 : 4); ++q) epi_out(1, 2 * (NU - 1), yprev, q, true);
    }
}
#endif  // __HIPCC__
