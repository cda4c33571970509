// az_resblock_sp17.h -- one whole ResNetBlock of a 64-filter tower on 17x17 planes in ONE launch, at the REFERENCE'S precision class
// (the fp32-class split arithmetic of az_conv_sp.h: every value a hi + lo f16 pair, three f16 MFMA products per multiply, fp32 accumulation):
//     y = relu(conv3x3(relu(conv3x3(x, w1) + b1), w2) + b2 + x)             (alpha_zero/core/network.py:42-82, eval mode, BatchNorm folded;
//                                                                             evaluated in fp32 by core/pipeline.py:91-123)
// This is the block of the reference's 13x13 Gomoku tower (network.py:101-105: the pad-3 stem makes 17x17 planes; BASELINE C2: 6 x 64).
// Two k_conv3x3_sp17 launches per block move five tensor passes through HBM (x in, m out; m in, x in again as the skip, y out) and at 64
// filters that kernel leans on both roofs (round 4: matrix pipe busy 0.64 - 0.71 at a clock the HBM traffic's share of the package
// power pulls down, 3.3 TB/s; 0.427 of the f16 MFMA peak).  Here the intermediate activation m never leaves the CU:
//   * tile = HALF a board (output rows 0-8 / 9-16), two phases per tile on the same persistent 256-thread workgroup (one wave per SIMD,
//     wave q = couts [16 q, 16 q + 16) of BOTH convolutions: 2 x 144 = 288 weight registers, 256 of them AGPRs):
//       phase A  conv1 on the x image in LDS (13 image rows) -> the m rows the tile's conv2 needs (rows 0-9 / 8-16: one halo row is
//                recomputed per half: 21 + 19 = 40 column tiles of 16 positions per board against 38 unfused, + 5 % MFMAs), bias + ReLU +
//                range record + split into hi / lo, written straight into the m image in LDS (ds_write_b64);
//       barrier; phase B  conv2 on the m image, the skip from the x IMAGE IN LDS (round 6; round 5 re-read x from global memory, and a
//                third of those re-reads missed the L2: HBM reads x 1.59 of x), epilogue as in k_conv3x3_sp17<RES>, y to global memory;
//                the NEXT tile's x image arrives by LDS-DMA during phase B, each 64-cell piece as soon as the last skip read of its
//                cells is done (below); barrier.
//     HBM per block: x once (rows 7-10 twice, the second time from L2) + y once = 2 passes instead of 5.
//   * skip from LDS without a second x buffer: wave q's skip values (its 16 couts = input channels 16 q .. 16 q + 15, both planes) live in
//     exactly four strips of the x image -- (plane, chunk) = (0, 2q), (0, 2q + 1), (1, 2q), (1, 2q + 1) -- and wave q is the wave that
//     DMAs exactly these four strips for the next tile: whether a cell may be overwritten is a matter of ONE wave's program order, no
//     barrier.  The units of phase B walk the board top-down, so piece pc (cells 64 pc ..) of the wave's strips is issued right after
//     the last unit whose skip reads touch it (compile-time checked against the lane maps: SbSkip); the skip of the LAST unit is
//     fetched one unit early (8 registers), so that every piece is on its way >= 19 k-steps before the tile-end barrier needs it.
//   * LDS: x image 16 strips (plane, chunk) x 256 cells x 16 B = 64 KB, m image 16 x 240 cells = 60 KB, one buffer each (the phases
//     alternate between them, nothing is double-buffered), + a 20 KB lane table = 144 KB.  Image rows: cell(ri, x) = 1 + 19 ri + x with two
//     zero cells between rows; the two halves use shifted row windows so that the zero rows they need (above the board / below it)
//     are cells nobody ever writes: x image H0: ri = r + 1 (ri 0 = row -1), H1: ri = r - 5 (ri 12 = row 17); m image H0: mi = r + 1, H1: mi = r - 6.
//   * units of <= 2 column tiles with two accumulator sets; the epilogue of unit u rides in the MFMA stream of unit u + 1 as
//     one-instruction micro-ops (SpSpread); only the last unit of phase A finishes exposed (its m values must be in LDS before the barrier).
//     Per-(column tile, lane) offsets (B-fragment base, output slot) live in an LDS table and are fetched one unit ahead into a rotating
//     set of 3 x 2 register pairs: the kernel carries 6 such registers instead of 40 (the 288 weight registers leave ~220 for everything else).
// Results are BIT-IDENTICAL to two azsp_conv3x3_split launches (same MFMA order per output, same roundings): tests/test_split_tower.py.
// y must not alias x (the lower half re-reads x rows 7-8 after the upper half has written y rows 0-8).
#pragma once
#include "az_conv_sp17.h"

#if defined(__HIPCC__)
// Geometry of a fused block kernel (k_resblock_sp<G, R>).  A BLOCK is what a persistent workgroup takes per iteration: G::BPB boards, worked off as
// G::NH tiles of two phases each (segments).  Positions are "virtual": pv = vr * S + x with virtual rows vr = 0 .. VROWS - 1.
//   Sb17: one 17x17 board = two half-board tiles (the 13x13 Gomoku tower; round 5)
//   Sb9 : TWO 9x9 boards stacked with one separator row (vr = 9 has no positions) = ONE tile: the zero cells between the boards are exactly
//         the padding each board's convolutions need, conv1 produces all 162 intermediate positions before conv2 starts, 11 column tiles
//         per convolution (162 of 176 slots) = 5.5 per board against 5 + 1 / 16 of the unfused kernel's corner trick: + 8.6 % MFMAs
struct Sb17 {
    static constexpr int S = 17, P2 = 289, PITCH = 19, VROWS = 17, BPB = 1, NH = 2;
    static constexpr int XCELLS = 256, MCELLS = 240;       // 1 + 13 * 19 = 248 / 1 + 12 * 19 = 229, rounded up to multiples of 16 cells
    static constexpr int NSEG = 4, NSEQ = 21, NTILE = 40;  // segments (A,H0) (B,H0) (A,H1) (B,H1); units and column tiles per board
    static constexpr bool row_has_pos(int) { return true; }
    static constexpr int gpos_off(int pv) { return pv * 16; }  // byte offset of a virtual position in a (plane, chunk) block of the BLOCK's first board
    static constexpr int seg_b(int h) { return 2 * h + 1; }    // the phase-B segment of tile kind h
    static constexpr int xcell_of_base(int h) { return h ? 1 + 2 * PITCH : 1 + PITCH; }  // x-image cell of a phase-B position = its B-fragment base cell (m image) + this
    static constexpr int xcell(int h, int vr, int x) { return 1 + PITCH * (h ? vr - 5 : vr + 1) + x; }
    static constexpr int seg_ph(int g) { return g & 1; }   // 0 = phase A (conv1: x image -> m image), 1 = phase B (conv2: m image -> y)
    static constexpr int seg_h(int g) { return g >> 1; }
    static constexpr int seg_r0(int g) { return g == 0 ? 0 : g == 1 ? 0 : g == 2 ? 8 : 9; }     // first output row of the segment
    static constexpr int seg_nr(int g) { return g == 0 ? 10 : g == 1 ? 9 : g == 2 ? 9 : 8; }    // output rows
    static constexpr int seg_nct(int g) { return g == 0 ? 11 : g == 1 ? 10 : g == 2 ? 10 : 9; }  // column tiles
    static constexpr int seg_rbase(int g) { return g == 0 ? 0 : g == 1 ? 0 : g == 2 ? 6 : 7; }   // B-fragment base cell of (r, x) = 19 (r - rbase) + x
    static constexpr int seg_nu(int g) { return (seg_nct(g) + 1) / 2; }
    static constexpr int seg_tile0(int g) { return g == 0 ? 0 : g == 1 ? 11 : g == 2 ? 21 : 31; }
    static constexpr int seg_seq0(int g) { return g == 0 ? 0 : g == 1 ? 6 : g == 2 ? 11 : 16; }
    static constexpr int unit_nj(int g, int u) { return seg_nct(g) - 2 * u >= 2 ? 2 : 1; }
    // m-image cell of an m position = its phase-A base cell + this (H0: mi = r + 1 -> 1 + 19 (r + 1) + x = base + 20; H1: mi = r - 6 -> base + 1)
    static constexpr int m_cell_of_base(int h) { return h ? 1 : 20; }
    // source row of x-image row ri of half h (-1: a row the DMA never writes)
    static constexpr int x_src_row(int h, int ri) {
        if (h == 0) return (ri >= 1 && ri <= 11) ? ri - 1 : -1;
        return (ri >= 2 && ri <= 11) ? ri + 5 : -1;
    }
    static constexpr int x_pos_of_cell(int h, int cell) {
        const int k = cell - 1;
        if (k < 0) return -1;
        const int ri = k / PITCH, xx = k % PITCH;
        if (xx >= S) return -1;
        const int r = x_src_row(h, ri);
        return r < 0 ? -1 : r * S + xx;
    }
};
struct Sb9 {
    static constexpr int S = 9, P2 = 81, PITCH = 11, VROWS = 19, BPB = 2, NH = 1;
    static constexpr int XCELLS = 240, MCELLS = 240;       // 1 + 21 * 11 = 232 (image rows -1 .. 19), rounded up to a multiple of 16 cells
    static constexpr int NSEG = 2, NSEQ = 12, NTILE = 22;  // segments (A) (B); 6 units and 11 column tiles each
    static constexpr int GTILEB = 2 * 8 * 81 * 16;         // bytes of one board in the split layout (64 channels)
    static constexpr bool row_has_pos(int vr) { return vr != 9; }
    static constexpr int gpos_off(int pv) { return pv / S >= 10 ? GTILEB + ((pv / S - 10) * S + pv % S) * 16 : pv * 16; }
    static constexpr int seg_b(int) { return 1; }
    static constexpr int xcell_of_base(int) { return 1 + PITCH; }
    static constexpr int xcell(int, int vr, int x) { return 1 + PITCH * (vr + 1) + x; }
    static constexpr int seg_ph(int g) { return g; }
    static constexpr int seg_h(int) { return 0; }
    static constexpr int seg_r0(int) { return 0; }
    static constexpr int seg_nr(int) { return 19; }
    static constexpr int seg_nct(int) { return 11; }
    static constexpr int seg_rbase(int) { return 0; }
    static constexpr int seg_nu(int g) { return (seg_nct(g) + 1) / 2; }
    static constexpr int seg_tile0(int g) { return 11 * g; }
    static constexpr int seg_seq0(int g) { return 6 * g; }
    static constexpr int unit_nj(int g, int u) { return seg_nct(g) - 2 * u >= 2 ? 2 : 1; }
    static constexpr int m_cell_of_base(int) { return 1 + PITCH; }  // mi = vr + 1 -> 1 + 11 (vr + 1) + x = base + 12
    static constexpr int x_src_row(int, int ri) { return (ri >= 1 && ri <= 19 && ri != 10) ? ri - 1 : -1; }
    static constexpr int x_pos_of_cell(int h, int cell) {
        const int k = cell - 1;
        if (k < 0) return -1;
        const int ri = k / PITCH, xx = k % PITCH;
        if (xx >= S) return -1;
        const int r = x_src_row(h, ri);
        return r < 0 ? -1 : r * S + xx;
    }
};
template <class G> constexpr bool sb_numbering_ok() {
    return G::seg_seq0(G::NSEG - 1) + G::seg_nu(G::NSEG - 1) == G::NSEQ && G::seg_tile0(G::NSEG - 1) + G::seg_nct(G::NSEG - 1) == G::NTILE && G::NSEQ % 3 == 0;
}
static_assert(sb_numbering_ok<Sb17>() && sb_numbering_ok<Sb9>(), "unit / tile numbering; the rotating lane-table registers keep their phase from block to block");

// (segment, column tile, lane & 15) -> position: column tile k of a segment takes the k-th position of every residue class (base cell
// mod 16); unfilled slots repeat the last position of a residue class the tile still lacks (the repeats compute and store the same value).
template <class G> struct SbMap {
    unsigned short pos[G::NTILE * 16];
    bool ok;
};
template <class G> constexpr SbMap<G> sb_make_map() {
    SbMap<G> m{};
    bool ok = true;
    for (int g = 0; g < G::NSEG; ++g) {
        int cnt[16] = {}, fill[11] = {};
        bool used[11][16] = {};
        bool seen_pos[G::VROWS * G::S] = {};
        const int nct = G::seg_nct(g), t0 = G::seg_tile0(g), r0 = G::seg_r0(g), r1 = r0 + G::seg_nr(g), rb = G::seg_rbase(g);
        for (int r = r0; r < r1; ++r)
            for (int x = 0; x < G::S; ++x) {
                if (!G::row_has_pos(r)) continue;
                const int res = (G::PITCH * (r - rb) + x) & 15, k = cnt[res]++;
                if (k >= nct) {
                    ok = false;
                    continue;
                }
                m.pos[(t0 + k) * 16 + fill[k]++] = (unsigned short)(r * G::S + x);
                used[k][res] = true;
                seen_pos[r * G::S + x] = true;
            }
        for (int k = 0; k < nct; ++k)
            for (int res = 0; res < 16 && fill[k] < 16; ++res) {
                if (used[k][res]) continue;
                bool found = false;
                // (round 6: the LAST position of the class, i.e. a cell near the ones the late column tiles read anyway -- the skip of a
                // padding slot is read from the x image like everyone's, and the early rows' cells are overwritten by then)
                for (int r = r1 - 1; r >= r0 && !found; --r)
                    for (int x = G::S - 1; x >= 0 && !found; --x)
                        if (G::row_has_pos(r) && ((G::PITCH * (r - rb) + x) & 15) == res) {
                            m.pos[(t0 + k) * 16 + fill[k]++] = (unsigned short)(r * G::S + x);
                            used[k][res] = true;
                            found = true;
                        }
                if (!found) ok = false;
            }
        for (int k = 0; k < nct; ++k) {  // every column tile: 16 slots inside the segment's rows, 16 distinct residues
            bool seen[16] = {};
            if (fill[k] != 16) ok = false;
            for (int s = 0; s < 16; ++s) {
                const int p = m.pos[(t0 + k) * 16 + s], r = p / G::S, x = p % G::S;
                if (r < r0 || r >= r1 || !G::row_has_pos(r)) ok = false;
                const int res = (G::PITCH * (r - rb) + x) & 15;
                if (seen[res]) ok = false;
                seen[res] = true;
            }
        }
        for (int r = r0; r < r1; ++r)
            for (int x = 0; x < G::S; ++x)
                if (G::row_has_pos(r) && !seen_pos[r * G::S + x]) ok = false;
    }
    m.ok = ok;
    return m;
}
static_assert(sb_make_map<Sb17>().ok && sb_make_map<Sb9>().ok, "column-tile maps of the fused blocks: every position of every segment covered, conflict-free lane groups");
template <class G> struct SbMapDev {
    static __device__ const SbMap<G> map;
};
template <class G> __device__ const SbMap<G> SbMapDev<G>::map = sb_make_map<G>();
// The skip of phase B comes from the x image (G::xcell).  SbSkip<G> derives, from the lane maps, WHEN each 64-cell DMA piece of the next tile's
// x image may be issued inside phase B of tile kind h: in the last unit in which a skip read touches its cells (unit 0 if none does).
//   * the skip values of unit u are read in k-step 0 of unit u -- except the LAST unit's, which are fetched in k-step 4 of the unit before
//     (8 registers), so that no piece has to wait for the last unit;
//   * a skip read issued in k-step t has returned when the MFMAs of k-step t + R + 1 issue (the compiler's wait for the fragments requested in
//     k-step t + 1 covers every older LDS read): reads of k-step 0 -> the piece from k-step 8 on, the early fetch (k-step 4) -> from k-step 13;
//   * a piece = this wave's four strips = 4 DMA instructions, two per k-step.
template <class G> struct SbSkip {
    static constexpr int NP = (G::XCELLS + 63) / 64;
    struct Tab {
        int unit[2][4], t0[2][4];
        bool ok;
    };
    static constexpr int read_unit(int g, int u) { return u == G::seg_nu(g) - 1 ? u - 1 : u; }
    static constexpr Tab make() {
        Tab tb{};
        const SbMap<G> m = sb_make_map<G>();
        bool ok = NP <= 4;
        for (int h = 0; h < G::NH; ++h) {
            const int g = G::seg_b(h), nu = G::seg_nu(g);
            if (nu < 3) ok = false;
            for (int pc = 0; pc < 4; ++pc) {
                int last = -1;
                bool early = false;  // the early fetch (of the last unit's skip, in unit nu - 2) touches the piece
                for (int u = 0; u < nu; ++u)
                    for (int j = 0; j < G::unit_nj(g, u); ++j)
                        for (int s = 0; s < 16; ++s) {
                            const int p = m.pos[(G::seg_tile0(g) + 2 * u + j) * 16 + s];
                            if (G::xcell(h, p / G::S, p % G::S) / 64 != pc) continue;
                            if (read_unit(g, u) > last) last = read_unit(g, u);
                            if (u == nu - 1) early = true;
                        }
                tb.unit[h][pc] = last < 0 ? 0 : last;
                tb.t0[h][pc] = last < 0 ? 1 : (early && last == nu - 2 ? 13 : 8);
            }
        }
        tb.ok = ok;
        return tb;
    }
    static constexpr Tab tab = make();
    static constexpr int dma_unit(int h, int pc) { return tab.unit[h][pc]; }
    static constexpr int dma_t0(int h, int pc) { return tab.t0[h][pc]; }
};
static_assert(SbSkip<Sb17>::tab.ok && SbSkip<Sb9>::tab.ok, "fused blocks: DMA piece schedule");
// (round 6's first version hard-coded Sb17's schedule; the derivation reproduces it)
static_assert(SbSkip<Sb17>::dma_unit(0, 3) == 0 && SbSkip<Sb17>::dma_t0(0, 3) == 1 && SbSkip<Sb17>::dma_unit(0, 0) == 1 && SbSkip<Sb17>::dma_t0(0, 0) == 8 &&
              SbSkip<Sb17>::dma_unit(0, 1) == 3 && SbSkip<Sb17>::dma_t0(0, 1) == 8 && SbSkip<Sb17>::dma_unit(0, 2) == 3 && SbSkip<Sb17>::dma_t0(0, 2) == 13 &&
              SbSkip<Sb17>::dma_unit(1, 0) == 0 && SbSkip<Sb17>::dma_unit(1, 1) == 1 && SbSkip<Sb17>::dma_unit(1, 2) == 3 && SbSkip<Sb17>::dma_t0(1, 2) == 8 &&
              SbSkip<Sb17>::dma_unit(1, 3) == 3 && SbSkip<Sb17>::dma_t0(1, 3) == 13, "Sb17: the derived DMA schedule");
typedef __attribute__((address_space(1))) unsigned char* sb17_gptr;         // pointers into global memory whose value the compiler must take
typedef const __attribute__((address_space(1))) unsigned char* sb17_gcptr;  // as given (an opaque scalar base per plane, see the board loop)

// f32 v minus the f16 half H of hpk: the remainder v - hi of the split, exact in fp32, one v_fma_mix_f32
template <int H> __device__ __forceinline__ float sb17_mix_diff(unsigned hpk, float v) {
    float r;
    if constexpr (H == 0) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(v));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(v));
    return r;
}
// f16(d * 2^11) (the product is exact in fp32: one rounding, to nearest even) into the low / high half of a packed pair
__device__ __forceinline__ unsigned sb17_scale_cvt_lo(float d) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(r) : "v"(d), "s"(SP_SCALE));
    return r;
}
__device__ __forceinline__ unsigned sb17_scale_cvt_hi(unsigned lo, float d) {
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(lo) : "v"(d), "s"(SP_SCALE));
    return lo;
}

// w1, w2: [plane: hi, lo][9 taps][64 couts][64 cin] f16 with lo = (w - hi) * 2^11 (the packing of azsp_conv3x3_split); b1, b2 fp32 [64].
// R = slots of the B-fragment ring (k-steps): a fragment is requested R - 1 k-steps before its MFMAs.
// nblocks = blocks of G::BPB boards (Sb17: boards; Sb9: PAIRS of boards -- an odd last board is the launcher's business).
template <class G, int R> __global__ void __launch_bounds__(CW_THREADS, 1)
k_resblock_sp(const unsigned char* __restrict__ x, const _Float16* __restrict__ w1, const float* __restrict__ b1, const _Float16* __restrict__ w2,
              const float* __restrict__ b2, unsigned char* __restrict__ y, int nblocks, unsigned* range) {
    typedef SbSkip<G> SK;
    constexpr int C = 64, NCH = 8, CIN = 64, KSUB = 2;
    constexpr int KS = 9 * KSUB;                             // k-steps per unit (one tap x 32 input channels)
    constexpr int NJM = 2;                                   // most column tiles per unit
    constexpr int XBLK = G::XCELLS * 16, XPLANE = NCH * XBLK, XBUF = 2 * XPLANE;    // 4096, 32768, 65536
    constexpr int MBLK = G::MCELLS * 16, MPLANE = NCH * MBLK, MBUF = 2 * MPLANE;    // 3840, 30720, 61440
    constexpr int TBL0 = XBUF + MBUF, TBL = G::NTILE * 64 * 8;                      // lane table: [tile][lane] {B base offset, output offset}
    constexpr int GBLK = G::P2 * 16, GPLANE = NCH * GBLK, GTILE = 2 * GPLANE;       // split layout of a board in global memory
    constexpr int GBLOCK = G::BPB * GTILE;                                          // ... of a block
    constexpr int NP = (G::XCELLS + 63) / 64;                // DMA pieces of 64 cells per strip (the last one may reach past the strip: those lanes are masked)
    constexpr int SPW = 2 * NCH / 4, NPIECE = NP * SPW;      // strips and DMA pieces per wave and tile: 4, 16
    constexpr int NFC = 2 * KS, NF = 2 * NFC;                // A fragments per convolution (2 planes x 18 k-steps), in all: 72
    // epilogue micro-ops per column tile (sp_epi_* in az_conv_sp.h).  TO THE M IMAGE (phase A): per element 1 (join), per pair of elements 8
    // (range record of both in one v_max3_f32, 2 x ReLU + clamp in one v_med3_f32, packed hi convert, 2 exact remainders v - hi, 2
    // scale-and-convert v_fma_mixlo/hi_f16), 2 LDS stores: 22.  TO GLOBAL MEMORY (phase B): per element 3 (join, skip join, add), per pair 8, 2
    // stores: 30.  (k_conv3x3_sp17 of round 4 spent 30 / 38 on the same arithmetic; the results are bit-identical.)
    constexpr int E1M = sp_epi_e1(false), PAIRM = sp_epi_pair(false), CTM = sp_epi_ct_ops(false);
    constexpr int E1G = sp_epi_e1(true), PAIRG = sp_epi_pair(true), CTG = sp_epi_ct_ops(true);
    constexpr int S0 = 6;                                    // first MFMA slot of a unit that may touch the previous unit's accumulators
    static_assert(KS % R == 0, "every unit starts at ring phase 0");
    static_assert(SPW == 4 && NP == 4 && KS >= 15, "four strips per wave, four pieces per strip: a piece = 4 DMA instructions = 2 k-steps");
    static_assert((2 * G::PITCH + 2) * 16 + (KSUB - 1) * 4 * XBLK + XPLANE < 65536, "fragment addresses are a per-lane base + a 16-bit immediate");
    static_assert(XBLK % 256 == 0 && MBLK % 256 == 0, "strips are multiples of 256 B: the four 8-channel groups of a fragment share banks");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[TBL0 + TBL];
    static_assert(TBL0 + TBL <= 160 * 1024, "LDS budget");
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    const int slot = (int)blockIdx.x, nslot = (int)gridDim.x;
    for (int i = tid; i < TBL0 / 16; i += CW_THREADS) *(cv_u32x4*)(lds + i * 16) = (cv_u32x4){0u, 0u, 0u, 0u};
    // lane table (identical for the four waves: the wave's own channel offset is wave-uniform and added where it is used)
    for (int i = tid; i < G::NTILE * 64; i += CW_THREADS) {
        const int tile = i >> 6, ln = i & 63, tl15 = ln & 15, tkg = ln >> 4;
        int g = 0;
        while (g < G::NSEG - 1 && tile >= G::seg_tile0(g + 1)) ++g;
        const int pos = SbMapDev<G>::map.pos[tile * 16 + tl15], r = pos / G::S, xx = pos % G::S;
        const int base = G::PITCH * (r - G::seg_rbase(g)) + xx;
        unsigned boff, ooff;
        // B bases are offsets from the start of the LDS block (phase B's include the m image's own offset XBUF): every fragment
        // address is lane register + an immediate below 64 K, the k-loop carries no address arithmetic
        if (G::seg_ph(g) == 0) {
            boff = (unsigned)(base * 16 + tkg * XBLK);
            ooff = (unsigned)(XBUF + (base + G::m_cell_of_base(G::seg_h(g))) * 16 + (tkg & 1) * 8 + (tkg >> 1) * MBLK);
        } else {
            boff = (unsigned)(XBUF + base * 16 + tkg * MBLK);
            ooff = (unsigned)(G::gpos_off(pos) + (tkg >> 1) * GBLK + (tkg & 1) * 8);
        }
        *(cv_u32x2*)(lds + TBL0 + i * 8) = (cv_u32x2){boff, ooff};
    }
    CV_BARRIER();  // the zero cells and the table are final before any LDS-DMA piece can land
    if (slot >= nblocks) return;  // (uniform per workgroup)

    // A fragments: fragment f = conv * NFC + plane * KS + (tap * KSUB + ks): lane (cout = 16 wave + l15, cin = 32 ks + 8 kg .. + 8)
    sp_f16x8 wf[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int cv = f / NFC, pl = (f % NFC) / KS, st = f % KS;
        const _Float16* w = cv ? w2 : w1;
        wf[f] = *(const sp_f16x8*)(w + ((size_t)((pl * 9 + st / KSUB) * C + wave * 16 + l15)) * CIN + (st % KSUB) * 32 + kg * 8);
    }
    c6_f32x4 bv[2];  // biases in the D layout (rows = couts 4 kg + e of the wave's 16): the C operand of a unit's first k-step
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[0][e] = b1[wave * 16 + 4 * kg + e], bv[1][e] = b2[wave * 16 + 4 * kg + e];

    // LDS-DMA plan per half: a strip is NP pieces of 64 cells; wave q moves the four strips its OWN skip values live in: (plane, chunk) =
    // (0, 2q), (0, 2q + 1), (1, 2q), (1, 2q + 1) (strip = plane * NCH + chunk): nobody else reads the skip from them (header)
    unsigned dsrc[G::NH][NP];
    unsigned long long dmask[G::NH][NP];
#pragma unroll
    for (int h = 0; h < G::NH; ++h)
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int p = G::x_pos_of_cell(h, 64 * i + lane);
            dsrc[h][i] = (unsigned)(p < 0 ? 0 : G::gpos_off(p));
            dmask[h][i] = __builtin_amdgcn_ballot_w64(p >= 0);
        }
    auto dma_piece = [&](const unsigned char* src, bool live, int h, int i) __attribute__((always_inline)) {
        const int sx = i / NP, c = (sx >> 1) * NCH + 2 * wave + (sx & 1), pc = i % NP;
        const unsigned long long base = (unsigned long long)(src + (size_t)c * GBLK);
        const unsigned long long mask = live ? dmask[h][pc] : 0ull;
        const unsigned dst = lds0 + (unsigned)(c * XBLK + pc * 1024);
        asm volatile("s_mov_b64 exec, %0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1"
                     :
                     : "s"(mask), "s"(dst), "v"(dsrc[h][pc]), "s"(base)
                     : "memory");
    };
    const unsigned char* Xs = lds;
    const unsigned char* Ms = lds;                           // (phase B's lane-table offsets carry the m image's offset XBUF)
    unsigned char* Mw = lds + (wave * 2) * MBLK;             // this wave's two chunk strips of the m image (table offsets carry XBUF; lo plane at + MPLANE)
    const unsigned char* tbl = lds + TBL0 + lane * 8;
    // skip address in the x image = a phase-B B-fragment base (lane table: XBUF + 16 base + kg MBLK) + this + 16 G::xcell_of_base(H) (+ XPLANE):
    // chunk 2 wave + (kg >> 1), the lane's half (kg & 1) of the 16-byte cell -- the four channels its accumulators hold
    const unsigned skc = (unsigned)(-XBUF - kg * MBLK + (kg >> 1) * XBLK + (kg & 1) * 8 + 2 * wave * XBLK);
    cv_u32x2 lm[3][NJM];  // rotating lane-table registers [unit sequence number % 3][column tile of the unit] = {B base offset, output offset}
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int j = 0; j < NJM; ++j) lm[a][j] = (cv_u32x2){0u, 0u};
    auto load_lm = [&](int rot, int tile0, int nj) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NJM; ++j)
            if (j < nj) lm[rot][j] = *(const cv_u32x2*)(tbl + (tile0 + j) * 512);
    };
    sp_f16x8 bb[R][2][NJM];  // ring of B fragments [k-step slot][plane][column tile of the unit]
    // fragments of k-step s of a unit whose lane-table registers are lm[rot] from image `img` (PH = 0: x image, 1: m image)
    auto load_step = [&](auto PHC, const unsigned char* img, int rot, int nj, int s, int rs) __attribute__((always_inline)) {
        constexpr int PH = decltype(PHC)::value, BLK = PH ? MBLK : XBLK, PLANE = PH ? MPLANE : XPLANE;
        const int tap = s / KSUB;
        const int off = ((tap / 3) * G::PITCH + (tap % 3)) * 16 + (s % KSUB) * (4 * BLK);
#pragma unroll
        for (int j = 0; j < NJM; ++j)
            if (j < nj) bb[rs][0][j] = *(const sp_f16x8*)(img + lm[rot][j].x + off);
#pragma unroll
        for (int j = 0; j < NJM; ++j)
            if (j < nj) bb[rs][1][j] = *(const sp_f16x8*)(img + lm[rot][j].x + off + PLANE);
    };

    // fragment f of a k-step's 2 nj (plane f / nj, column tile f % nj): inside the units the requests are spread over the first MFMA gaps of
    // the k-step R - 1 before their use (one ds_read_b128 per gap instead of a burst of 2 nj in front of a k-step; az_conv_sp2.h)
    auto load_frag = [&](auto PHC, const unsigned char* img, int rot, int nj, int s, int rs, int f) __attribute__((always_inline)) {
        constexpr int PH = decltype(PHC)::value, BLK = PH ? MBLK : XBLK, PLANE = PH ? MPLANE : XPLANE;
        const int tap = s / KSUB;
        const int off = ((tap / 3) * G::PITCH + (tap % 3)) * 16 + (s % KSUB) * (4 * BLK);
        const int pl = f / nj, j = f % nj;
        bb[rs][pl][j] = *(const sp_f16x8*)(img + lm[rot][j].x + off + pl * PLANE);
    };

    {   // first tile (upper half of the first board): all pieces at once, then the first lane-table registers and fragments
        const unsigned char* src = x + (size_t)slot * GBLOCK;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) dma_piece(src, true, 0, i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CV_BARRIER();
        load_lm(0, G::seg_tile0(0), G::unit_nj(0, 0));
#pragma unroll
        for (int s = 0; s < R - 1; ++s) load_step(CpInt<0>{}, Xs, 0, G::unit_nj(0, 0), s, s);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {  // the compiler's wait for the weight loads belongs in front of the loop (see az_conv.h)
        if (f < 64) asm volatile("" : : "a"(wf[f]));
        else asm volatile("" : : "v"(wf[f]));
    }
    asm volatile("" : : "v"(bv[0]), "v"(bv[1]), "v"(dsrc[0][0]), "v"(dsrc[G::NH - 1][NP - 1]));

    c6_f32x4 accm[2][NJM], accc[2][NJM];  // [accumulator set][column tile of the unit]
    cv_u32x2 rr[2][NJM][2];               // skip values of a phase-B unit: [accumulator set][column tile][plane]
    cv_u32x2 rrx[NJM][2];                 // skip values of the LAST unit of a phase B, fetched one unit early (see SbSkip)
#pragma unroll
    for (int j = 0; j < NJM; ++j) rrx[j][0] = (cv_u32x2){0u, 0u}, rrx[j][1] = (cv_u32x2){0u, 0u};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < NJM; ++j) {
            accm[a][j] = (c6_f32x4){0.0f, 0.0f, 0.0f, 0.0f}, accc[a][j] = (c6_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            rr[a][j][0] = (cv_u32x2){0u, 0u}, rr[a][j][1] = (cv_u32x2){0u, 0u};
        }
    float evv[2] = {0.0f, 0.0f}, sc[2] = {0.0f, 0.0f}, t0 = 0.0f, mx = 0.0f;  // mx: largest |value| this lane produced (range record)
    unsigned hpk[2] = {0u, 0u}, lpk[2] = {0u, 0u};
    // micro-op `o` of the epilogue of column tile j of the unit with accumulator set `set`: ONE VALU / memory instruction.
    // GLOBAL = false: phase A (bias + ReLU -> m image at Mw + ooff); true: phase B (bias + skip + ReLU -> global memory at out + ooff)
    auto epi_op = [&](auto GC, int set, int j, unsigned ooff, unsigned char* out, sb17_gptr out_lo, int o, bool store_ok) __attribute__((always_inline)) {
        constexpr bool GLOBAL = decltype(GC)::value != 0;
        constexpr int E1 = GLOBAL ? E1G : E1M, PAIR = GLOBAL ? PAIRG : PAIRM;
        if (o < 2 * PAIR) {
            const int pr = o / PAIR, k = o % PAIR;  // pair pr = elements 2 pr, 2 pr + 1 (one packed dword of each plane)
            if (k < 2 * E1) {
                const int ei = k / E1, kk = k % E1, e = 2 * pr + ei;
                const unsigned rh = pr == 0 ? rr[set][j][0].x : rr[set][j][0].y, rl = pr == 0 ? rr[set][j][1].x : rr[set][j][1].y;
                if (kk == 0) evv[ei] = fmaf(accc[set][j][e], SP_INV_SCALE, accm[set][j][e]);
                else if (GLOBAL && kk == 1) t0 = ei == 0 ? sp_mix_join<0>(rh, rl) : sp_mix_join<1>(rh, rl);
                else if (GLOBAL && kk == 2) evv[ei] = cw_add_f32(evv[ei], t0);
            } else {
                const int kk = k - 2 * E1;
                if (kk == 0) mx = sp_max3_abs(mx, evv[0], evv[1]);                                            // what the reference would carry on ...
                else if (kk == 1) evv[0] = __builtin_amdgcn_fmed3f(evv[0], 0.0f, SP_F16_MAX);                  // ... is clamped here (ReLU in the same median)
                else if (kk == 2) evv[1] = __builtin_amdgcn_fmed3f(evv[1], 0.0f, SP_F16_MAX);
                else if (kk == 3) hpk[pr] = sp_cvt_pk(evv[0], evv[1]);
                else if (kk == 4) sc[0] = sb17_mix_diff<0>(hpk[pr], evv[0]);
                else if (kk == 5) sc[1] = sb17_mix_diff<1>(hpk[pr], evv[1]);
                else if (kk == 6) lpk[pr] = sb17_scale_cvt_lo(sc[0]);
                else lpk[pr] = sb17_scale_cvt_hi(lpk[pr], sc[1]);
            }
        } else if constexpr (GLOBAL) {
            if (o == 2 * PAIR) {
                if (store_ok) *(cv_u32x2*)(out + ooff) = (cv_u32x2){hpk[0], hpk[1]};
            } else if (store_ok) *(__attribute__((address_space(1))) cv_u32x2*)(out_lo + ooff) = (cv_u32x2){lpk[0], lpk[1]};
        } else {
            if (o == 2 * PAIR) *(cv_u32x2*)(Mw + ooff) = (cv_u32x2){hpk[0], hpk[1]};
            else *(cv_u32x2*)(Mw + MPLANE + ooff) = (cv_u32x2){lpk[0], lpk[1]};
        }
    };

    int it = 0;
    unsigned char* yprev = y;
    sb17_gptr yprev_lo = (sb17_gptr)(unsigned long long)y;
    for (int board = slot; board < nblocks; board += nslot, ++it) {
        const bool has_next = board + nslot < nblocks;
        const unsigned char* xb = x + (size_t)board * GBLOCK;
        const unsigned char* xnb = x + (size_t)(has_next ? board + nslot : board) * GBLOCK;
        const size_t yo = (size_t)board * GBLOCK + (size_t)(wave * 2) * GBLK;  // uniform: the lane part comes from the lane table
        // one uniform base per plane, opaque to the compiler (it would otherwise fold base + GPLANE + lane offset into 64-bit VALU adds
        // per access: GPLANE exceeds the 13-bit immediate): every y store is scalar base + lane offset
        unsigned long long ylo = (unsigned long long)(y + yo) + GPLANE;
        asm volatile("" : "+s"(ylo));  // (an opaque scalar, cast back to a GLOBAL-address-space pointer: no flat accesses)
        unsigned char* ybase = y + yo;
        const sb17_gptr ybase_lo = (sb17_gptr)ylo;
        const bool have_prev = it > 0;
        // unit U of segment SG
        auto unit = [&](auto GC, auto UC) __attribute__((always_inline)) {
            constexpr int SG = decltype(GC)::value, U = decltype(UC)::value;
            constexpr int PH = G::seg_ph(SG), H = G::seg_h(SG), NU = G::seg_nu(SG);
            constexpr int SEQ = G::seg_seq0(SG) + U, ROT = SEQ % 3, PROT = (SEQ + 2) % 3, NROT = (SEQ + 1) % 3;
            constexpr int nj = G::unit_nj(SG, U);
            constexpr int set = PH == 0 ? (U & 1) : ((U + NU) & 1);                  // phase B: its last unit uses set 1 (phase A starts on set 0)
            constexpr bool LAST = U == NU - 1, FIRST = U == 0;
            // the next unit (the ring and the lane-table registers run one unit ahead)
            constexpr int NSG = LAST ? (SG + 1) % G::NSEG : SG, NUU = LAST ? 0 : U + 1;
            constexpr int nnj = G::unit_nj(NSG, NUU), ntile0 = G::seg_tile0(NSG) + 2 * NUU;
            // the previous unit, whose epilogue rides here: phase A, first unit: the last unit of the previous phase B (global); phase A,
            // later units: the previous unit of this phase (m image); phase B, first unit: nothing (phase A finished exposed); later: global
            constexpr int PSG = FIRST ? (SG + G::NSEG - 1) % G::NSEG : SG, PU = FIRST ? G::seg_nu(PSG) - 1 : U - 1;
            constexpr bool RIDE = !(PH == 1 && FIRST);
            constexpr bool RGLOBAL = G::seg_ph(PSG) == 1;
            constexpr int pnj = G::unit_nj(PSG, PU);
            constexpr int pset = G::seg_ph(PSG) == 0 ? (PU & 1) : ((PU + G::seg_nu(PSG)) & 1);
            static_assert(!RIDE || pset != set, "a unit and the epilogue riding in it use different accumulator sets");
            constexpr int CT_OPS = RGLOBAL ? CTG : CTM;
            constexpr int NQ = 3 * nj, P_OPS = RIDE ? pnj * CT_OPS : 1;              // MFMAs per k-step; micro-ops of the riding epilogue
            constexpr int AVAIL = NQ * KS - 4 - S0;                                  // the riders end 4 slots before the unit does
            typedef SpSpread<P_OPS, S0, AVAIL> SP;
            static_assert(SP::MAXPER <= 2, "the previous unit's epilogue fits this unit's MFMA gaps");
            const unsigned char* img = PH ? Ms : Xs;
            // where the riding epilogue stores: phase B units of the previous board (first unit of a board) or of this board
            unsigned char* pout = SG == 0 ? yprev : ybase;
            const sb17_gptr pout_lo = SG == 0 ? yprev_lo : ybase_lo;
            const bool pstore = SG == 0 && FIRST ? have_prev : true;
            cp_for_each([&](auto TC) __attribute__((always_inline)) {
                constexpr int t = decltype(TC)::value;
                if constexpr (PH == 1 && LAST && t == KS - (R - 1)) {
                    // end of the tile: every read of the m image has been issued, this wave's pieces of the next x image have landed
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    CV_BARRIER();
                }
                if constexpr (t == 3) load_lm(NROT, ntile0, nnj);  // the next unit's lane-table registers
                // the fragments of k-step t + R - 1: of this unit, of the next unit, or (last unit of phase B) of the next tile's phase A (behind the
                // barrier).  (Last unit of phase A: the m image is complete only behind the barrier -- phase B's first fragments are read there.)
                constexpr bool lsame = t + R - 1 < KS, lhave = lsame || !LAST || PH == 1;
                constexpr int lph = (!lsame && LAST) ? 0 : PH, lrot = lsame ? ROT : NROT, lnj = lsame ? nj : nnj;
                constexpr int ls = lsame ? t + R - 1 : t + R - 1 - KS, lrs = (t + R - 1) % R;
                const unsigned char* limg = (!lsame && LAST) ? Xs : img;
                cp_for_each([&](auto QC) __attribute__((always_inline)) {
                    constexpr int q = decltype(QC)::value, j = q % nj, prod = q / nj;  // product 0: main, 1: w_hi x_lo, 2: w_lo x_hi
                    constexpr int fa = PH * NFC + (prod == 2 ? KS + t : t), pl = prod == 1 ? 1 : 0;
                    if constexpr (prod == 0) {
                        if constexpr (t == 0) sp_mfma_ac(accm[set][j], wf[fa], bb[t % R][pl][j], bv[PH]);
                        else sp_mfma_a(accm[set][j], wf[fa], bb[t % R][pl][j]);
                    } else if constexpr (prod == 1) {
                        if constexpr (t == 0) sp_mfma_a0(accc[set][j], wf[fa], bb[t % R][pl][j]);
                        else sp_mfma_a(accc[set][j], wf[fa], bb[t % R][pl][j]);
                    } else {
                        if constexpr (fa < 64) sp_mfma_a(accc[set][j], wf[fa], bb[t % R][pl][j]);
                        else sp_mfma_v(accc[set][j], wf[fa], bb[t % R][pl][j]);
                    }
                    constexpr int sl = t * NQ + q;  // MFMA slot of the unit
                    if constexpr (lhave) {  // fragment q in gap q; a unit with fewer MFMAs per k-step than the next unit has fragments: the rest in its last gap
                        cp_for_each([&](auto FC) __attribute__((always_inline)) {
                            constexpr int f = decltype(FC)::value;
                            if constexpr ((f < NQ ? f : NQ - 1) == q) load_frag(CpInt<lph>{}, limg, lrot, lnj, ls, lrs, f);
                        }, typename CpMakeSeq<2 * lnj>::type{});
                    }
                    if constexpr (RIDE) {
                        cp_for_each([&](auto KC) __attribute__((always_inline)) {
                            constexpr int o = SP::cum(sl - 1) + decltype(KC)::value;
                            if constexpr (o < SP::cum(sl)) epi_op(CpInt<RGLOBAL ? 1 : 0>{}, pset, o / CT_OPS, lm[PROT][o / CT_OPS].y, pout, pout_lo, o % CT_OPS, pstore);
                        }, typename CpMakeSeq<SP::MAXPER>::type{});
                    }
                    if constexpr (PH == 1 && sl < 2 * nj) {  // this unit's skip values (used by its epilogue inside the next unit): from the x image
                        constexpr int rj = sl >> 1, rp = sl & 1;
                        if constexpr (LAST) rr[set][rj][rp] = rrx[rj][rp];  // (fetched in the unit before: the cells are being overwritten by now)
                        else rr[set][rj][rp] = *(const cv_u32x2*)(Xs + (lm[ROT][rj].x + skc) + G::xcell_of_base(H) * 16 + rp * XPLANE);
                    }
                    if constexpr (PH == 1 && U == NU - 2 && t == 4 && q < 2 * nnj) {  // the LAST unit's skip values, one unit early (its lane-table registers arrived in k-step 3)
                        constexpr int rj = q >> 1, rp = q & 1;
                        rrx[rj][rp] = *(const cv_u32x2*)(Xs + (lm[NROT][rj].x + skc) + G::xcell_of_base(H) * 16 + rp * XPLANE);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }, typename CpMakeSeq<NQ>::type{});
                if constexpr (PH == 1) {
                    // the tile after this one (the lower half of this board, or the upper half of this workgroup's next board) arrives piece by
                    // piece behind the skip reads (SbSkip): piece pc of this wave's four strips in unit dma_unit(H, pc), k-steps dma_t0 and + 1
                    cp_for_each([&](auto PC) __attribute__((always_inline)) {
                        constexpr int pc = decltype(PC)::value;
                        if constexpr (SK::dma_unit(H, pc) == U && (t == SK::dma_t0(H, pc) || t == SK::dma_t0(H, pc) + 1)) {
                            constexpr int sx = 2 * (t - SK::dma_t0(H, pc));  // strips sx, sx + 1 of the wave's four
                            // the tile after this one: the next tile kind of this block, or tile kind 0 of this workgroup's next block
                            if constexpr (H + 1 < G::NH) dma_piece(xb, true, H + 1, sx * NP + pc), dma_piece(xb, true, H + 1, (sx + 1) * NP + pc);
                            else dma_piece(xnb, has_next, 0, sx * NP + pc), dma_piece(xnb, has_next, 0, (sx + 1) * NP + pc);
                        }
                    }, typename CpMakeSeq<NP>::type{});
                }
                __builtin_amdgcn_sched_barrier(0);
            }, typename CpMakeSeq<KS>::type{});
            if constexpr (PH == 0 && LAST) {
                // ---- end of phase A: this unit's m values go to LDS exposed, then the barrier, then phase B's first fragments
                if constexpr (nj == 2) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(accm[set][0]), "+v"(accm[set][1]), "+v"(accc[set][0]), "+v"(accc[set][1]));
                else asm volatile("s_nop 15\n\ts_nop 15" : "+v"(accm[set][0]), "+v"(accc[set][0]));
#pragma unroll
                for (int j = 0; j < nj; ++j)
#pragma unroll
                    for (int o = 0; o < CTM; ++o) epi_op(CpInt<0>{}, set, j, lm[ROT][j].y, nullptr, (sb17_gptr)0, o, true);
                CV_BARRIER();
#pragma unroll
                for (int s = 0; s < R - 1; ++s) load_step(CpInt<1>{}, Ms, NROT, nnj, s, s);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto segment = [&](auto GC) __attribute__((always_inline)) {
            constexpr int SG = decltype(GC)::value;
            cp_for_each([&](auto UC) __attribute__((always_inline)) { unit(CpInt<SG>{}, UC); }, typename CpMakeSeq<G::seg_nu(SG)>::type{});
        };
        cp_for_each([&](auto GC) __attribute__((always_inline)) { segment(GC); }, typename CpMakeSeq<G::NSEG>::type{});
        yprev = ybase, yprev_lo = ybase_lo;
    }
    // epilogue of the very last unit (the last phase B's last unit: accumulator set 1, lane-table registers of the last sequence number)
    {
        constexpr int LSG = G::NSEG - 1, nj = G::unit_nj(LSG, G::seg_nu(LSG) - 1), ROT = (G::NSEQ - 1) % 3;
        static_assert(nj == 1 && ((G::seg_nu(LSG) - 1 + G::seg_nu(LSG)) & 1) == 1, "the block's last unit holds one column tile and accumulates into set 1");
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(accm[1][0]), "+v"(accc[1][0]));
#pragma unroll
        for (int j = 0; j < nj; ++j)
#pragma unroll
            for (int o = 0; o < CTG; ++o) epi_op(CpInt<1>{}, 1, j, lm[ROT][j].y, yprev, yprev_lo, o, true);
    }
    sp_range_report(mx, range);
}
#endif  // __HIPCC__
