// az_conv_sp17.h -- the 3x3 convolution of a 64-filter tower on 17x17 planes at the REFERENCE'S precision class (fp32-class split
// arithmetic of az_conv_sp.h: every value travels as hi = f16(v), lo = f16((v - hi) 2^11); a product is three f16 MFMAs into two fp32
// accumulators):
//     y = relu(conv3x3(x, w) + bias [+ residual])
// This is the tower of the reference's 13x13 Gomoku network (alpha_zero/core/network.py:101-105: the pad-3 stem turns 13x13 boards into
// 17x17 planes; BASELINE config C2, 6 x 64; the shipped checkpoints are 10 x 40, zero-widened to 64: core/network.py widen_network),
// which the reference evaluates in fp32 (core/pipeline.py:91-123).  x, residual, y in the split layout of az_conv_sp.h:
//     [board][plane: hi, lo][C/8 chunks][289 positions][8 channels] f16
// Kernel = the weight-stationary scheme of k_conv3x3_sp with HALF-BOARD tiles (two f16 planes of a whole 17x17 x 64 board are a 94 KB
// LDS image: no double buffer; the bf16 kernels of az_conv19.h solve the same problem the same way):
//   * one persistent 256-thread workgroup per CU, one wave per SIMD; wave q keeps the hi AND lo filter banks of couts [16 q, 16 q + 16)
//     in registers (2 planes x 9 taps x 2 k-steps x 4 = 144 registers, all AGPRs) for the whole launch.
//   * a board is TWO tiles: output rows 0-8 (153 positions = 10 column tiles of 16) and rows 9-16 (136 positions = 9 column tiles);
//     the LDS image of a tile holds its 11 input rows (one halo row above and below; rows outside the board are never written and stay
//     zero) per (plane, chunk) as a strip of 224 16-byte cells, cell(rs, x) = 1 + 19 rs + x: two zero cells between the rows, so a tap
//     (dy, dx) is the constant cell offset 19 dy + dx and every B-fragment address is a per-lane base + an immediate.  Strips are a
//     multiple of 256 B, so the four 8-channel lane groups of a fragment share banks and a ds_read_b128 service group = the 16 positions
//     of a column tile, chosen distinct mod 16 (Sp17Map, checked at compile time): conflict-free.  15 of the 304 column slots of a board
//     repeat a position (the repeats compute and store the same value: no lane is ever masked).
//   * the two tiles of a board alternate between the two LDS buffers (buffer 0 only ever holds upper halves, buffer 1 lower halves: their
//     zero rows are static); the next tile's strips arrive by LDS-DMA (global_load_lds_dwordx4, masked to the position cells) in the
//     shadow of the first unit's MFMAs; ONE barrier per tile.
//   * a tile is 4 units of <= 3 column tiles with two accumulator sets; the epilogue of unit i - 1 (join of the two accumulators,
//     residual, range check, clamp, split into hi / lo, two 8-byte stores per column tile) rides in the MFMA stream of unit i as
//     one-instruction micro-ops spread evenly over the unit's MFMA gaps (v_fma_mix_f32 joins / remainders straight on the packed f16
//     halves, packed converts: 38 micro-ops per column tile against 54 MFMAs); the B-fragment ring (3 k-steps) runs on across units, tiles and boards.
//   * NCH = 4 is the STEM (17 planes padded to 32 channels -> 64): the input is the 13x13 feature board of azsp_split_features, embedded
//     at (2, 2) of the zero 17x17 plane by the DMA masks -- a pad-3 convolution of the board IS the pad-1 convolution of that plane.
// Per board and wave: 19 column tiles x 18 k-steps x 3 products = 1026 MFMAs (v_mfma_f32_16x16x32_f16); HBM: each tensor once
// (the halo rows 8 / 9 are read by both tiles of a board: the second read is an L2 hit).
#pragma once
#include "az_conv_sp.h"

#if defined(__HIPCC__)
struct Sp17Geo {
    static constexpr int S = 17, P2 = 289, PITCH = 19, CELL0 = 1, IMG_ROWS = 11;
    static constexpr int CELLS = 224;                      // 1 + 11 * 19 = 210 with the tap reach, rounded up to a multiple of 16
    static constexpr int NCT0 = 10, NCT1 = 9, NCT = 19;    // column tiles of the upper / lower half, of a board
    static constexpr int r0(int h) { return h ? 9 : 0; }   // first output row of half h
    static constexpr int nr(int h) { return h ? 8 : 9; }   // output rows of half h
    static constexpr int nct(int h) { return h ? NCT1 : NCT0; }
    static constexpr int ct0(int h) { return h ? NCT0 : 0; }  // first column tile of half h in the board's list of 19
    // LDS cell of the (-1, -1) neighbour of output position (r, x) in the image of its half = the B-fragment base of that position
    static constexpr int base_cell(int h, int r, int x) { return PITCH * (r - r0(h)) + x; }
    // source position (in an in_s x in_s input board embedded at (off, off) of the plane) of image cell `cell` of half h, or -1
    static constexpr int pos_of_cell(int h, int cell, int in_s, int off) {
        const int k = cell - CELL0;
        if (k < 0) return -1;
        const int rs = k / PITCH, xx = k % PITCH;
        if (rs >= IMG_ROWS || xx >= S) return -1;
        const int yy = r0(h) - 1 + rs - off, xi = xx - off;
        return (yy >= 0 && xi >= 0 && yy < in_s && xi < in_s) ? yy * in_s + xi : -1;
    }
};
// (column tile of the board, lane & 15) -> position: column tile k of a half takes the k-th position of every residue class (base cell
// mod 16) of that half; unfilled slots repeat the first position of a residue the tile still lacks.
struct Sp17Map {
    unsigned short pos[Sp17Geo::NCT * 16];
    bool ok;
};
constexpr Sp17Map sp17_make_map() {
    typedef Sp17Geo G;
    Sp17Map m{};
    bool ok = true;
    bool seen_pos[G::P2] = {};
    for (int h = 0; h < 2; ++h) {
        int cnt[16] = {}, fill[G::NCT0] = {};
        bool used[G::NCT0][16] = {};
        const int nct = G::nct(h), c0 = G::ct0(h);
        for (int r = G::r0(h); r < G::r0(h) + G::nr(h); ++r)
            for (int x = 0; x < G::S; ++x) {
                const int res = G::base_cell(h, r, x) & 15, k = cnt[res]++;
                if (k >= nct) {
                    ok = false;
                    continue;
                }
                m.pos[(c0 + k) * 16 + fill[k]++] = (unsigned short)(r * G::S + x);
                used[k][res] = true;
                seen_pos[r * G::S + x] = true;
            }
        for (int k = 0; k < nct; ++k)
            for (int res = 0; res < 16 && fill[k] < 16; ++res) {
                if (used[k][res]) continue;
                bool found = false;
                for (int r = G::r0(h); r < G::r0(h) + G::nr(h) && !found; ++r)
                    for (int x = 0; x < G::S && !found; ++x)
                        if ((G::base_cell(h, r, x) & 15) == res) {
                            m.pos[(c0 + k) * 16 + fill[k]++] = (unsigned short)(r * G::S + x);
                            used[k][res] = true;
                            found = true;
                        }
                if (!found) ok = false;
            }
        for (int k = 0; k < nct; ++k) {  // every column tile: 16 slots of this half, 16 distinct residues
            bool seen[16] = {};
            if (fill[k] != 16) ok = false;
            for (int s = 0; s < 16; ++s) {
                const int p = m.pos[(c0 + k) * 16 + s], r = p / G::S, x = p % G::S;
                if (r < G::r0(h) || r >= G::r0(h) + G::nr(h)) ok = false;
                const int res = G::base_cell(h, r, x) & 15;
                if (seen[res]) ok = false;
                seen[res] = true;
            }
        }
    }
    for (int p = 0; p < G::P2; ++p)
        if (!seen_pos[p]) ok = false;
    m.ok = ok;
    return m;
}
static_assert(sp17_make_map().ok, "column-tile map of the 17x17 split kernel: every position covered, conflict-free lane groups");
static_assert((Sp17Geo::CELLS * 16) % 256 == 0, "strips are a multiple of 256 B: the four 8-channel groups of a fragment share banks");
static __device__ const Sp17Map sp17_map = sp17_make_map();

// units of a tile: column tiles per unit and the first column tile of each unit (index into the board's list of 19)
__host__ __device__ constexpr int sp17_unj(int h, int u) { return h == 0 ? (u < 2 ? 3 : 2) : (u < 1 ? 3 : 2); }
__host__ __device__ constexpr int sp17_uj0(int h, int u) {
    int j = Sp17Geo::ct0(h);
    for (int i = 0; i < u; ++i) j += sp17_unj(h, i);
    return j;
}
static_assert(sp17_uj0(0, 3) + sp17_unj(0, 3) == Sp17Geo::NCT0 && sp17_uj0(1, 3) + sp17_unj(1, 3) == Sp17Geo::NCT, "units cover the column tiles");

// Vector-memory instructions a wave issues between the last LDS-DMA piece of the next tile (unit 0, k-step NPIECE) and the barrier of
// the tile (unit 3, k-step KS - 2) of half h: the stores of the riding epilogues (2 per column tile) and the residual loads (2 per
// column tile, first slots of a unit).  Mirrors the schedule of k_conv3x3_sp17 below (same constants, same SpSpread).
template <bool RES, int NCH, bool XLO0 = false> __host__ __device__ constexpr int sp17_vm_younger(int h) {
    constexpr int KSUB = NCH / 4, KS = 9 * KSUB, R = 3, S0 = 6, NP = (Sp17Geo::CELLS + 63) / 64, NPIECE = NP * ((XLO0 ? 1 : 2) * NCH / 4);
    constexpr int CT_OPS = sp_epi_ct_ops(RES);
    int n = 0;
    for (int u = 0; u < 4; ++u) {
        const int nj = sp17_unj(h, u), NQ = (XLO0 ? 2 : 3) * nj;
        const int ph = u == 0 ? h ^ 1 : h, pu = (u + 3) & 3, pnj = sp17_unj(ph, pu), P_OPS = pnj * CT_OPS;
        const int AVAIL = (u == 3 ? (KS - (R - 1)) * NQ : NQ * KS - 4) - S0;
        // slot -> cumulative micro-ops (SpSpread::cum with run-time arguments)
        auto cum = [&](int sl) {
            if (sl < S0) return 0;
            const long long c = ((long long)(sl - S0 + 1) * P_OPS + AVAIL - 1) / AVAIL;
            return c > P_OPS ? P_OPS : (int)c;
        };
        for (int c = 0; c < pnj; ++c)
            for (int st = 0; st < 2; ++st) {  // the two stores of column tile c are its last two micro-ops
                const int o = c * CT_OPS + CT_OPS - 2 + st;
                int sl = S0;
                while (cum(sl) <= o) ++sl;
                if (u > 0 || sl / NQ > NPIECE) ++n;  // a DMA piece is issued behind all MFMA slots of its k-step
            }
        if (RES && u > 0) n += 2 * nj;  // (unit 0 loads its residual in its first slots, before the first piece)
    }
    return n;
}

// NCH = input-channel chunks of 8: 8 = tower layer (64 -> 64), 4 = stem (17 planes padded to 32 -> 64, 13x13 input board at (2, 2)).
// w: [plane: hi, lo][9 taps][64 couts][8 NCH cin] f16 with lo = (w - hi) * 2^11; bias fp32 [64].
// XLO0: the input's lo plane is all zero (exact f16 values: AZSP_FEAT_F16_SPLIT features) -- its strips, fragments and the w_hi x_lo
// product are skipped (see k_conv3x3_sp).
template <bool RES, int NCH, bool XLO0 = false> __global__ void __launch_bounds__(CW_THREADS, 1)
k_conv3x3_sp17(const unsigned char* __restrict__ x, const _Float16* __restrict__ w, const float* __restrict__ bias,
               const unsigned char* __restrict__ res, unsigned char* __restrict__ y, int nboards, int relu, unsigned* range) {
    typedef Sp17Geo G;
    constexpr int C = 64, CIN = 8 * NCH, KSUB = NCH / 4;
    constexpr int KS = 9 * KSUB;                             // k-steps per unit (one tap x 32 input channels)
    constexpr int NJM = 3, R = 3;                            // most column tiles per unit; ring slots (k-steps)
    constexpr int IN_S = NCH == 8 ? G::S : 13, IN_OFF = NCH == 8 ? 0 : 2;
    constexpr int LBLK = G::CELLS * 16, LPLANE = NCH * LBLK, LBUF = 2 * LPLANE;
    constexpr int GBLK_IN = IN_S * IN_S * 16, XPLANE = NCH * GBLK_IN, XTILE = 2 * XPLANE;
    constexpr int GBLK = G::P2 * 16, YPLANE = (C / 8) * GBLK, YTILE = 2 * YPLANE;
    constexpr int NP = (G::CELLS + 63) / 64;                 // DMA pieces of 64 cells per strip
    constexpr int SPW = (XLO0 ? 1 : 2) * NCH / 4, NPIECE = NP * SPW;  // strips and DMA pieces per wave and tile (XLO0: hi strips only)
    constexpr int NPROD = XLO0 ? 2 : 3;
    static_assert(!XLO0 || (!RES && NCH == 4), "exact-f16 inputs: the stem");
    constexpr int NF = 2 * KS;
    constexpr int E1 = sp_epi_e1(RES), PAIR = sp_epi_pair(RES), CT_OPS = sp_epi_ct_ops(RES);  // epilogue micro-ops (az_conv_sp.h sp_epi_e1)
    constexpr int S0 = 6;                                    // first MFMA slot of a unit that may touch the previous unit's accumulators
    static_assert(NF <= 64, "all A fragments live in AGPRs");
    static_assert((4 * KS) % R == 0, "a tile's k-steps keep the ring phase");
    static_assert(KS - 1 >= NPIECE, "the next tile's pieces ride in unit 0");
    static_assert((G::PITCH * 8 + 16) * 16 + 3 * LBLK + (2 * G::PITCH + 2) * 16 + (KSUB - 1) * 4 * LBLK + LPLANE < 65536,
                  "fragment addresses are a base + a 16-bit immediate");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * LBUF];
    static_assert(2 * LBUF <= 160 * 1024, "LDS budget");
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int slot = (int)blockIdx.x, nslot = (int)gridDim.x;
    for (int i = tid; i < 2 * LBUF / 16; i += CW_THREADS) *(cv_u32x4*)(lds + i * 16) = (cv_u32x4){0u, 0u, 0u, 0u};
    CV_BARRIER();  // the zero cells are final before any LDS-DMA piece can land
    if (slot >= nboards) return;  // (uniform per workgroup)

    // A fragments: fragment f = plane * KS + (tap * KSUB + ks): lane (cout = 16 wave + l15, cin = 32 ks + 8 kg .. + 8)
    sp_f16x8 wf[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int pl = f / KS, st = f % KS;
        wf[f] = *(const sp_f16x8*)(w + ((size_t)((pl * 9 + st / KSUB) * C + wave * 16 + l15)) * CIN + (st % KSUB) * 32 + kg * 8);
    }
    c6_f32x4 bv;  // bias in the D layout (rows = couts 4 kg + e of the wave's 16): the C operand of the first k-step
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = bias[wave * 16 + 4 * kg + e];
    const float lo_clamp = relu ? 0.0f : -SP_F16_MAX;  // lower bound of the epilogue's median: ReLU and the range clamp in one instruction

    // LDS-DMA plan per half: a strip is NP pieces of 64 cells; wave q moves strips SPW q .. SPW q + SPW - 1 (strip = plane * NCH + chunk)
    unsigned dsrc[2][NP];
    unsigned long long dmask[2][NP];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int p = G::pos_of_cell(h, 64 * i + lane, IN_S, IN_OFF);
            dsrc[h][i] = (unsigned)((p < 0 ? 0 : p) * 16);
            dmask[h][i] = __builtin_amdgcn_ballot_w64(p >= 0);
        }
    auto dma_piece = [&](const unsigned char* src, unsigned dstbuf, bool live, int h, int i) {
        const int c = SPW * wave + i / NP, pc = i % NP;
        const unsigned long long base = (unsigned long long)(src + (size_t)c * GBLK_IN);
        const unsigned long long mask = live ? dmask[h][pc] : 0ull;
        const unsigned dst = dstbuf + (unsigned)(c * LBLK + pc * 1024);
        asm volatile("s_mov_b64 exec, %0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1"
                     :
                     : "s"(mask), "s"(dst), "v"(dsrc[h][pc]), "s"(base)
                     : "memory");
    };
    unsigned lmap[G::NCT];
#pragma unroll
    for (int j = 0; j < G::NCT; ++j) {
        const int h = j >= G::NCT0 ? 1 : 0, pos = sp17_map.pos[j * 16 + l15];
        // low 16 bits: LDS byte offset of the (-1, -1) neighbour in this lane's 8-channel group; high 16 bits: byte offset of this lane's
        // 8-byte output slot inside the wave's two chunk strips of a plane (couts 4 kg .. + 4 = chunk kg / 2, half kg % 2)
        lmap[j] = (unsigned)((G::PITCH * (pos / G::S - G::r0(h)) + pos % G::S) * 16 + kg * LBLK) |
                  ((unsigned)(pos * 16 + (kg >> 1) * GBLK + (kg & 1) * 8) << 16);
    }
    static_assert(G::P2 * 16 + GBLK + 8 < 65536, "output slot offsets fit 16 bits");
    sp_f16x8 bb[R][2][NJM];  // ring of B fragments [k-step slot][plane][column tile of the unit]
    auto load_step = [&](const unsigned char* img, int j0, int nj, int s, int rs) {
        const int tap = s / KSUB;
        const int off = ((tap / 3) * G::PITCH + (tap % 3)) * 16 + (s % KSUB) * (4 * LBLK);
#pragma unroll
        for (int j = 0; j < NJM; ++j)
            if (j < nj) bb[rs][0][j] = *(const sp_f16x8*)(img + (lmap[j0 + j] & 0xffffu) + off);
        if constexpr (!XLO0) {
#pragma unroll
            for (int j = 0; j < NJM; ++j)
                if (j < nj) bb[rs][1][j] = *(const sp_f16x8*)(img + (lmap[j0 + j] & 0xffffu) + off + LPLANE);
        }
    };

    // fragment f of a k-step (plane f / nj, column tile f % nj): inside the units the requests are spread over the first MFMA gaps of the
    // k-step R - 1 before their use (one ds_read_b128 per gap instead of a burst in front of a k-step; see k_conv3x3_sp)
    auto load_frag = [&](const unsigned char* img, int j0, int nj, int s, int rs, int f) __attribute__((always_inline)) {
        const int tap = s / KSUB;
        const int off = ((tap / 3) * G::PITCH + (tap % 3)) * 16 + (s % KSUB) * (4 * LBLK);
        const int pl = f / nj, j = f % nj;
        bb[rs][pl][j] = *(const sp_f16x8*)(img + (lmap[j0 + j] & 0xffffu) + off + pl * LPLANE);
    };

    {   // first tile (upper half of the first board): all pieces at once, then the first fragments
        const unsigned char* src = x + (size_t)slot * XTILE;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) dma_piece(src, lds0, true, 0, i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CV_BARRIER();
#pragma unroll
        for (int s = 0; s < R - 1; ++s) load_step(lds, sp17_uj0(0, 0), sp17_unj(0, 0), s, s);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) asm volatile("" : : "a"(wf[f]));  // the compiler's wait for the weight loads belongs in front of the loop
    asm volatile("" : : "v"(bv), "v"(lmap[0]), "v"(lmap[G::NCT - 1]), "v"(dsrc[0][0]), "v"(dsrc[1][NP - 1]));

    c6_f32x4 accm[2][NJM], accc[2][NJM];  // [unit parity][column tile of the unit]
    cv_u32x2 rr[2][NJM][2];               // residual of a unit: [unit parity][column tile][plane]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < NJM; ++j) {
            accm[a][j] = (c6_f32x4){0.0f, 0.0f, 0.0f, 0.0f}, accc[a][j] = (c6_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            rr[a][j][0] = (cv_u32x2){0u, 0u}, rr[a][j][1] = (cv_u32x2){0u, 0u};
        }
    float evv[2] = {0.0f, 0.0f}, sc[2] = {0.0f, 0.0f}, t0 = 0.0f, mx = 0.0f;  // mx: largest |value| this lane produced (range record)
    unsigned hpk[2] = {0u, 0u}, lpk[2] = {0u, 0u};
    // micro-op `o` of the epilogue of column tile j (lmap index mj) of the unit with accumulator set `set`: ONE VALU / memory instruction
    auto epi_op = [&](int set, int j, int mj, unsigned char* out, sp_gptr out_lo, int o, bool store_ok) {
        if (o < 2 * PAIR) {
            const int pr = o / PAIR, k = o % PAIR;  // pair pr = elements 2 pr, 2 pr + 1 (one packed dword of each plane)
            if (k < 2 * E1) {
                const int ei = k / E1, kk = k % E1, e = 2 * pr + ei;
                const unsigned rh = pr == 0 ? rr[set][j][0].x : rr[set][j][0].y, rl = pr == 0 ? rr[set][j][1].x : rr[set][j][1].y;
                if (kk == 0) evv[ei] = fmaf(accc[set][j][e], SP_INV_SCALE, accm[set][j][e]);
                else if (RES && kk == 1) t0 = ei == 0 ? sp_mix_join<0>(rh, rl) : sp_mix_join<1>(rh, rl);
                else if (RES && kk == 2) evv[ei] = cw_add_f32(evv[ei], t0);
            } else {
                const int kk = k - 2 * E1;
                if (kk == 0) mx = sp_max3_abs(mx, evv[0], evv[1]);                                            // what the reference would carry on ...
                else if (kk == 1) evv[0] = __builtin_amdgcn_fmed3f(evv[0], lo_clamp, SP_F16_MAX);                  // ... is clamped here (ReLU in the same median)
                else if (kk == 2) evv[1] = __builtin_amdgcn_fmed3f(evv[1], lo_clamp, SP_F16_MAX);
                else if (kk == 3) hpk[pr] = sp_cvt_pk(evv[0], evv[1]);
                else if (kk == 4) sc[0] = sp_mix_diff<0>(hpk[pr], evv[0]);
                else if (kk == 5) sc[1] = sp_mix_diff<1>(hpk[pr], evv[1]);
                else if (kk == 6) lpk[pr] = sp_scale_cvt_lo(sc[0]);
                else lpk[pr] = sp_scale_cvt_hi(lpk[pr], sc[1]);
            }
        } else {
            const unsigned gq = lmap[mj] >> 16;
            if (o == 2 * PAIR) {
                if (store_ok) *(cv_u32x2*)(out + gq) = (cv_u32x2){hpk[0], hpk[1]};
            } else if (store_ok) *(__attribute__((address_space(1))) cv_u32x2*)(out_lo + gq) = (cv_u32x2){lpk[0], lpk[1]};
        }
    };

    int it = 0;
    unsigned char* yprev = y;
    sp_gptr yprev_lo = (sp_gptr)(unsigned long long)y;
    for (int board = slot; board < nboards; board += nslot, ++it) {
        const bool has_next = board + nslot < nboards;
        const unsigned char* xb = x + (size_t)board * XTILE;
        const unsigned char* xnb = x + (size_t)(has_next ? board + nslot : board) * XTILE;
        const size_t yo = (size_t)board * YTILE + (size_t)(wave * 2) * GBLK;  // uniform: the lane part is in lmap
        const unsigned char* rbase = RES ? res + yo : nullptr;
        unsigned char* ybase = y + yo;
        // one opaque uniform base per plane: every residual load / y store is scalar base + 32-bit lane offset (see k_conv3x3_sp)
        unsigned long long rlo = (unsigned long long)(RES ? res + yo : y + yo) + YPLANE, ylo = (unsigned long long)(y + yo) + YPLANE;
        asm volatile("" : "+s"(rlo), "+s"(ylo));
        const sp_gcptr rbase_lo = (sp_gcptr)rlo;
        const sp_gptr ybase_lo = (sp_gptr)ylo;
        const bool have_prev = it > 0;
        // unit U of half H (tile H of the board lives in LDS buffer H)
        auto unit = [&](auto HC, auto UC) __attribute__((always_inline)) {
            constexpr int H = decltype(HC)::value, U = decltype(UC)::value;
            constexpr int set = U & 1, pset = set ^ 1;
            constexpr int nj = sp17_unj(H, U), j0 = sp17_uj0(H, U);
            constexpr int PH = U == 0 ? H ^ 1 : H, PU = (U + 3) & 3;                 // the previous unit (its epilogue rides here)
            constexpr int pnj = sp17_unj(PH, PU), pj0 = sp17_uj0(PH, PU);
            constexpr int NH = U == 3 ? H ^ 1 : H, NU = (U + 1) & 3;                 // the next unit (the ring runs on into it)
            constexpr int nnj = sp17_unj(NH, NU), nj0 = sp17_uj0(NH, NU);
            constexpr int NQ = NPROD * nj, P_OPS = pnj * CT_OPS;                     // MFMAs per k-step; micro-ops of the riding epilogue
            // the riders end 4 slots before the unit does; in the tile's last unit they end before the barrier (whose counted wait
            // knows exactly which vector-memory instructions are younger than the next tile's DMA pieces)
            constexpr int AVAIL = (U == 3 ? (KS - (R - 1)) * NQ : NQ * KS - 4) - S0;
            typedef SpSpread<P_OPS, S0, AVAIL> SP;
            static_assert(SP::MAXPER <= (NCH >= 8 ? 2 : XLO0 ? 6 : 4), "the previous unit's epilogue fits this unit's MFMA gaps");
            const unsigned char* Xs = lds + H * LBUF;
            const unsigned char* Xn = lds + (H ^ 1) * LBUF;
            // previous unit's output: the first unit of a board finishes the previous board's lower half
            unsigned char* pout = (H == 0 && U == 0) ? yprev : ybase;
            const sp_gptr pout_lo = (H == 0 && U == 0) ? yprev_lo : ybase_lo;
            const bool pstore = (H == 0 && U == 0) ? have_prev : true;
            // the tile after this one: the lower half of this board, or the upper half of this workgroup's next board
            const unsigned char* nsrc = H == 0 ? xb : xnb;
            const bool nlive = H == 0 ? true : has_next;
            const unsigned ndst = lds0 + (unsigned)((H ^ 1) * LBUF);
            cp_for_each([&](auto TC) __attribute__((always_inline)) {
                constexpr int t = decltype(TC)::value;
                constexpr int g = U * KS + t;  // running k-step of the tile
                if constexpr (U == 3 && t == KS - (R - 1)) {
                    // every read of this buffer has been issued.  This wave's pieces of the next tile (unit 0) are older than the
                    // VM_YOUNGER youngest vector-memory instructions it has issued (residual loads and stores of units 0-3, counted at
                    // compile time: each is issued unconditionally); those may stay in flight
                    constexpr int VM_YOUNGER = sp17_vm_younger<RES, NCH, XLO0>(H);
                    static_assert(VM_YOUNGER < 63, "vmcnt field");
                    if (H == 0 && !have_prev) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // first tile: the stores riding in its unit 0 were skipped
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_YOUNGER) : "memory");
                    CV_BARRIER();
                }
                // the fragments of k-step t + R - 1 (of this unit, of the next unit of this tile, or of the first unit of the next tile's image)
                constexpr bool lsame = t + R - 1 < KS;
                constexpr int ls = lsame ? t + R - 1 : t + R - 1 - KS, lnj = lsame ? nj : nnj, lj0 = lsame ? j0 : nj0, lrs = (g + R - 1) % R;
                const unsigned char* limg = (lsame || U < 3) ? Xs : Xn;
                cp_for_each([&](auto QC) __attribute__((always_inline)) {
                    constexpr int q = decltype(QC)::value, j = q % nj;
                    cp_for_each([&](auto FC) __attribute__((always_inline)) {  // fragment q in gap q (the rest in the last gap of a short k-step)
                        constexpr int f = decltype(FC)::value;
                        if constexpr ((f < NQ ? f : NQ - 1) == q) load_frag(limg, lj0, lnj, ls, lrs, f);
                    }, typename CpMakeSeq<(XLO0 ? 1 : 2) * lnj>::type{});
                    constexpr int prod = XLO0 ? (q / nj == 0 ? 0 : 2) : q / nj;       // product 0: main, 1: w_hi x_lo (skipped when x_lo = 0), 2: w_lo x_hi
                    constexpr int fa = prod == 2 ? KS + t : t, pl = prod == 1 ? 1 : 0;
                    if constexpr (prod == 0) {
                        if constexpr (t == 0) sp_mfma_ac(accm[set][j], wf[fa], bb[g % R][pl][j], bv);
                        else sp_mfma_a(accm[set][j], wf[fa], bb[g % R][pl][j]);
                    } else if constexpr (prod == 1) {
                        if constexpr (t == 0) sp_mfma_a0(accc[set][j], wf[fa], bb[g % R][pl][j]);
                        else sp_mfma_a(accc[set][j], wf[fa], bb[g % R][pl][j]);
                    } else if constexpr (XLO0 && t == 0) {
                        sp_mfma_a0(accc[set][j], wf[fa], bb[g % R][pl][j]);  // (the correction accumulator starts here: no w_hi x_lo product)
                    } else {
                        sp_mfma_a(accc[set][j], wf[fa], bb[g % R][pl][j]);
                    }
                    constexpr int sl = t * NQ + q;  // MFMA slot of the unit
                    cp_for_each([&](auto KC) __attribute__((always_inline)) {
                        constexpr int o = SP::cum(sl - 1) + decltype(KC)::value;
                        if constexpr (o < SP::cum(sl)) epi_op(pset, o / CT_OPS, pj0 + o / CT_OPS, pout, pout_lo, o % CT_OPS, pstore);
                    }, typename CpMakeSeq<SP::MAXPER>::type{});
                    if constexpr (RES && sl < 2 * nj) {  // this unit's residual (used by its epilogue inside the next unit)
                        constexpr int rj = sl >> 1, rp = sl & 1;
                        if constexpr (rp == 0) rr[set][rj][rp] = *(const cv_u32x2*)(rbase + (lmap[j0 + rj] >> 16));
                        else rr[set][rj][rp] = *(const __attribute__((address_space(1))) cv_u32x2*)(rbase_lo + (lmap[j0 + rj] >> 16));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }, typename CpMakeSeq<NQ>::type{});
                if constexpr (U == 0 && t >= 1 && t - 1 < NPIECE) dma_piece(nsrc, ndst, nlive, H ^ 1, t - 1);
                __builtin_amdgcn_sched_barrier(0);
            }, typename CpMakeSeq<KS>::type{});
        };
        unit(CpInt<0>{}, CpInt<0>{});
        unit(CpInt<0>{}, CpInt<1>{});
        unit(CpInt<0>{}, CpInt<2>{});
        unit(CpInt<0>{}, CpInt<3>{});
        unit(CpInt<1>{}, CpInt<0>{});
        unit(CpInt<1>{}, CpInt<1>{});
        unit(CpInt<1>{}, CpInt<2>{});
        unit(CpInt<1>{}, CpInt<3>{});
        yprev = ybase, yprev_lo = ybase_lo;
    }
    // epilogue of the very last unit (lower half, unit 3: accumulator set 1)
    {
        constexpr int nj = sp17_unj(1, 3), j0 = sp17_uj0(1, 3);
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(accm[1][0]), "+v"(accm[1][1]), "+v"(accc[1][0]), "+v"(accc[1][1]));
#pragma unroll
        for (int j = 0; j < nj; ++j)
#pragma unroll
            for (int o = 0; o < CT_OPS; ++o) epi_op(1, j, j0 + j, yprev, yprev_lo, o, true);
    }
    sp_range_report(mx, range);
}
#endif  // __HIPCC__
