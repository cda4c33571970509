// az_conv19.h -- weight-stationary 3x3 convolution of the 256-filter residual tower on 19x19 boards (the reference's jumbo Go
// configuration, alpha_zero/training_go_jumbo.py:46: 20 blocks x 256 filters; BASELINE config C5).
//     y = act(conv3x3(x, w) + bias [+ addend])     x, y, addend in the tiled layout [board][32 chunks][361 positions][8 ch] bf16
// (reference: alpha_zero/core/network.py:42-82 ResNetBlock in eval mode, BatchNorm folded into w / bias).
//
// A CU's register files hold 295 KB of weights = 128 couts x 128 cin x 9 taps, a quarter of a 256 -> 256 filter bank.  Splitting
// the couts four ways would make every CU read ALL input channels of its tiles (4x the L2 -> LDS traffic, and the convolution
// launches are energy-bound: profiles/r02_conv_ablation.txt), splitting cin needs fp32 partial sums to cross CUs.  So one
// convolution runs as TWO launches of the same kernel, each contracting one 128-channel half of the input:
//     launch A:  p = conv(x[:, 0:128])  + bias + residual         (no activation, bf16)
//     launch B:  y = act(conv(x[:, 128:256]) + p)                  (p read and y written in place, tile by tile)
// so a CU holds 128 couts (wave q: 32 couts x 128 cin x 9 taps = 288 registers, 256 of them AGPRs read in place by the MFMAs --
// exactly the per-wave shape of k_conv3x3_tiled), reads half the input channels of its tiles, and the two cout halves of a tile
// stream run on CUs of the same XCD (one HBM read, one L2 hit).  The price is one extra bf16 rounding of the partial sum and
// its HBM round trip (5 tensor passes per convolution instead of 3).
//   * tile = HALF a board: output rows r0 .. r0 + 9 with r0 = 0 or 9 (row 9 is computed by both halves and stored by the first
//     only), 190 positions = 6 column tiles of 32 = 6 units; v_mfma_f32_32x32x16_bf16, 72 k-steps per column tile.
//   * LDS image per 8-channel chunk: 242 cells of 16 B, cell(rs, x) = 1 + 20 rs + x for the 12 input rows rs (r0 - 1 .. r0 + 10)
//     with one zero cell between rows; rows outside the board are never written (zero).  A tap (dy, dx) is the constant cell offset
//     20 dy + dx.  Double buffered (2 x 61,952 B), filled by LDS-DMA: wave q moves cells [64 q, 64 q + 64) of every chunk strip.
//     A CU only ever sees tiles of one half (its zero rows are static): tile streams are per (cout half, board half).
//   * (column tile, lane) -> position from a residue-class table per board half as in az_conv.h (C9Map): six residue classes of a
//     10-row tile would have 13 members for 12 lane groups; the upper half leaves six positions of the shared row 9 to the lower half
//     (which holds that row anyway), so every class of both halves has <= 12 members: conflict-free (round 5; rounds 2-4 had a 2-way
//     conflict in 6 of 12 lane groups); unused slots repeat a cell and are never stored.
//   * epilogue straight from the accumulators, 8-byte slots; SOFTWARE-PIPELINED (round 3): a unit is ONE column tile (72 MFMAs that
//     accumulate into the same 16 registers back to back), two accumulator sets, the epilogue of unit u - 1 (addend add, bf16
//     rounding, ReLU, stores) is issued one instruction per MFMA gap inside unit u, the B-fragment ring runs on across units and
//     tiles, one barrier per tile behind a counted vmcnt, bias from LDS.  Round 2 ran the epilogue as one serial block after every
//     three column tiles (the two-column-tile units of az_conv.h would need three sets here or spill: 6 tiles = 3 units is odd).
#pragma once
#include "az_conv.h"

#if defined(__HIPCC__)
#define C9_S 19
#define C9_P2 361
#define C9_GBLK (C9_P2 * 16)          // 5,776 B: one 8-channel chunk block of a board in global memory
#define C9_ROWS 10                    // output rows per tile
#define C9_PITCH 20
#define C9_CELL0 21                   // cell of output (row 0 of the tile, column 0); its (-1, -1) neighbour is cell 0
#define C9_CELLS 242                  // 1 + 12 * 20 + 1
#define C9_LBLK (C9_CELLS * 16)       // 3,872 B per chunk strip
#define C9_NCT 6                      // column tiles of 32 positions
#define C9_NPOS (C9_ROWS * C9_S)      // 190

struct C9Map {
    unsigned short cell[C9_NCT * 32], pos[C9_NCT * 32];  // pos: tile-relative (row * 19 + col), 0xffff = no store
    bool ok;
};
// Board row 9 lies in both halves' images.  Six of the sixteen residue classes (cell mod 16) of a 10-row tile have 13 members for the
// 12 lane groups of its 6 column tiles (rounds 2-4: a 2-way bank conflict in 6 of 12 groups, 13.6 % of the LDS cycles in the PMC pass);
// each of them has a member in row 9, so the UPPER half (hf = 0) leaves those six positions of row 9 -- columns 0, 1, 2, 12, 13, 14 -- to
// the LOWER half (hf = 1), which computes rows 10-18 + exactly these six: every class of both maps has <= 12 members, every lane
// group reads 16 distinct residues: conflict-free.
constexpr bool c9_handed_over(int col) { return col <= 2 || (col >= 12 && col <= 14); }
constexpr bool c9_in_map(int hf, int p) {  // tile-relative position p = row * 19 + col of half hf: computed AND stored by that half?
    const int row = p / C9_S, col = p % C9_S;
    if (hf == 0) return !(row == C9_ROWS - 1 && c9_handed_over(col));
    return row >= 1 || c9_handed_over(col);
}
constexpr C9Map c9_make_map(int hf) {
    C9Map m{};
    const int lanes[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    const int NG = 2 * C9_NCT;
    int cnt[16] = {}, fill[NG] = {};
    bool used[NG][16] = {};
    bool ok = true;
    for (int i = 0; i < C9_NCT * 32; ++i) {
        m.cell[i] = 0;
        m.pos[i] = 0xffff;
    }
    for (int p = 0; p < C9_NPOS; ++p) {
        if (!c9_in_map(hf, p)) continue;
        const int cell = C9_CELL0 + C9_PITCH * (p / C9_S) + p % C9_S, r = cell & 15, k = cnt[r]++;
        if (k >= NG) {  // a class with more members than lane groups would need a 2-way conflict
            ok = false;
            continue;
        }
        const int idx = (k >> 1) * 32 + lanes[k & 1][fill[k]++];
        m.cell[idx] = (unsigned short)cell;
        m.pos[idx] = (unsigned short)p;
        used[k][r] = true;
    }
    for (int k = 0; k < NG; ++k)  // unused slots: a cell of a residue the group lacks (conflict-free), never stored
        for (int r = 0; r < 16 && fill[k] < 16; ++r) {
            if (used[k][r]) continue;
            for (int p = 0; p < C9_NPOS; ++p) {
                const int cell = C9_CELL0 + C9_PITCH * (p / C9_S) + p % C9_S;
                if ((cell & 15) == r) {
                    const int idx = (k >> 1) * 32 + lanes[k & 1][fill[k]++];
                    m.cell[idx] = (unsigned short)cell;
                    used[k][r] = true;
                    break;
                }
            }
        }
    for (int k = 0; k < NG; ++k) {  // every lane group: 16 slots, 16 distinct residues
        bool seen[16] = {};
        if (fill[k] != 16) ok = false;
        for (int i = 0; i < 16; ++i) {
            const int r = m.cell[(k >> 1) * 32 + lanes[k & 1][i]] & 15;
            if (seen[r]) ok = false;
            seen[r] = true;
        }
    }
    m.ok = ok;
    return m;
}
constexpr bool c9_maps_cover_the_board() {  // every board position is stored by exactly one half
    const C9Map m0 = c9_make_map(0), m1 = c9_make_map(1);
    int seen[C9_P2] = {};
    for (int i = 0; i < C9_NCT * 32; ++i) {
        if (m0.pos[i] != 0xffff) seen[m0.pos[i]]++;
        if (m1.pos[i] != 0xffff) seen[m1.pos[i] + 9 * C9_S]++;
    }
    for (int p = 0; p < C9_P2; ++p)
        if (seen[p] != 1) return false;
    return true;
}
static_assert(c9_make_map(0).ok && c9_make_map(1).ok, "19x19 column-tile maps: conflict-free lane groups in both board halves");
static_assert(c9_maps_cover_the_board(), "19x19 column-tile maps: every position stored exactly once");
static __device__ const C9Map c9_maps[2] = {c9_make_map(0), c9_make_map(1)};

// ADD: an addend tensor (the residual in launch A, the partial sum in launch B; may alias y).  NCH = input chunks contracted by this
// launch: 16 (one half of the tower's 256 channels) or 4 (the stem: 17 planes padded to 32).  cin_total = row length of w_packed
// [9 taps][256 couts][cin_total], cin_off = first input channel of this launch, x_chunks = chunks per board of x, x_chunk0 = first
// chunk read.  The output always has 32 chunks (256 couts).
template <bool ADD, int NCH> struct C9Sched {  // static schedule of one unit = one column tile: everything compile-time
    static constexpr int KS = NCH / 2, NSTEP = 9 * KS, NS = NSTEP;      // k-steps = MFMA slots per unit
    static constexpr int NU = C9_NCT;                                    // units per tile
    static constexpr int R = 4;                                          // B-fragment ring slots
    static constexpr int OPS = ADD ? 9 : 5, NQ = 4;                      // micro-ops per epilogue quad, quads (register quads) per unit
    static constexpr int S0 = 5;                                         // first slot that may touch the previous unit's accumulators
    static constexpr int PER = (NQ * OPS + (NS - S0 - 3) - 1) / (NS - S0 - 3);  // epilogue micro-ops per slot
    static constexpr int NPIECE = NCH;                                   // DMA pieces per wave per tile: unit 0, k-step 1 + 4 p
    static constexpr int T_BAR = NSTEP - (R - 1);                        // last unit: barrier before the ring crosses into the next tile
    // this unit's addend of quad q is loaded right behind the store micro-op of the previous unit's quad q (which consumed the same
    // registers a few micro-ops earlier): one addend set of 8 registers
    static constexpr int store_slot(int q) { return S0 + (q * OPS + OPS - 1) / PER; }
    static constexpr int dma_slot(int p) { return 1 + 4 * p; }
    // vector-memory operations issued after the last DMA piece (unit 0) and before the barrier (last unit, slot T_BAR), in program order
    static constexpr int vm_after_dma() {
        int n = 0;
        const int last = dma_slot(NPIECE - 1);
        for (int u = 0; u < NU; ++u)
            for (int q = 0; q < NQ; ++q) {
                const int ss = store_slot(q);
                if ((u > 0 || ss > last) && (u < NU - 1 || ss < T_BAR)) n += ADD ? 2 : 1;  // store of the previous unit's quad (+ this unit's addend load)
            }
        return n;
    }
    static_assert(NQ * OPS <= PER * (NS - S0 - 3), "the previous unit's epilogue must fit into the unit, before its set is re-initialised");
    static_assert(dma_slot(NPIECE - 1) < NSTEP, "DMA pieces ride in unit 0");
    static_assert((NU * NSTEP) % R == 0, "a tile's k-steps keep the ring phase");
    static_assert(store_slot(NQ - 1) < T_BAR, "no vector-memory rider between the barrier and the end of the last unit (counted wait)");
};

template <bool ADD, int NCH> __global__ void __launch_bounds__(CW_THREADS, 1)
k_conv3x3_hb19(const unsigned char* __restrict__ x, const unsigned short* __restrict__ w, const float* __restrict__ bias, const unsigned char* add,
               unsigned char* y, int nboards, int relu, int add_bias, int cin_total, int cin_off, int x_chunks, int x_chunk0) {
    typedef C9Sched<ADD, NCH> SC;
    constexpr int KS = SC::KS, NSTEP = SC::NSTEP, R = SC::R, NU = SC::NU;
    constexpr int LBUF = NCH * C9_LBLK;
    constexpr int NPIECE = SC::NPIECE;  // DMA pieces per wave per tile: its 64-cell quarter of every chunk strip
    constexpr int OTILE = 32 * C9_GBLK;  // output / addend board: 256 channels
    constexpr int VM_AFTER_DMA = SC::vm_after_dma();
    static_assert(VM_AFTER_DMA < 63, "vmcnt field");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * LBUF];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    for (int i = tid; i < 2 * LBUF / 16; i += CW_THREADS) *(cv_u32x4*)(lds + i * 16) = (cv_u32x4){0u, 0u, 0u, 0u};
    __syncthreads();  // the zero cells are in place before any wave's DMA lands

    // this CU's role: tile stream s, cout half ch, board half hf.  With a full grid the four roles of a stream share an XCD
    // (block b runs on XCD b % 8), so the second cout half finds the input tile in that XCD's L2.
    const int b = (int)blockIdx.x, nst = (int)gridDim.x >> 2;
    int s, type;
    if ((gridDim.x & 31u) == 0u) {
        s = (b & 7) + 8 * (b >> 5);
        type = (b >> 3) & 3;
    } else {
        s = b >> 2;
        type = b & 3;
    }
    const int ch = type >> 1, hf = type & 1, r0 = hf * 9;
    const size_t xboard = (size_t)x_chunks * C9_GBLK;

    cv_bf16x8 wf[NSTEP];  // this wave's 32 couts x (9 taps x 8 NCH cin): the A operand of every MFMA below
    const int cout0 = ch * 128 + wave * 32;
#pragma unroll
    for (int t = 0; t < NSTEP; ++t)
        wf[t] = *(const cv_bf16x8*)(w + ((size_t)((t / KS) * 256 + cout0 + l31)) * cin_total + cin_off + ((t % KS) * 2 + hi) * 8);
    // bias in the accumulator layout, kept in LDS ([wave][lane half][16 floats], broadcast reads): a unit's accumulators are initialised
    // from it a few k-steps before the unit starts, which frees the 16 registers a C operand would pin (as in az_conv.h)
    __shared__ __attribute__((aligned(64))) float bias_lds[4 * 2 * 16];
    if (lane < 32) bias_lds[(wave * 2 + (lane >> 4)) * 16 + (lane & 15)] = add_bias ? bias[cout0 + 8 * ((lane & 15) >> 2) + 4 * (lane >> 4) + (lane & 3)] : 0.0f;
    const cv_f32x16* bias_ptr = (const cv_f32x16*)(bias_lds + (wave * 2 + hi) * 16);
    const unsigned lo16 = relu ? 0u : 0x80008000u;

    // LDS-DMA: this wave's lane -> cell 64 wave + lane of a chunk strip; source = that cell's board position (rows outside the board
    // and the zero cells between rows are masked off)
    const int dcell = wave * 64 + lane, dk = dcell - 1, drs = dk / C9_PITCH, dxx = dk - drs * C9_PITCH, drow = r0 - 1 + drs;
    const bool dok = dcell >= 1 && dcell < C9_CELLS - 1 && dxx < C9_S && drow >= 0 && drow < C9_S;
    const unsigned dsrc = dok ? (unsigned)((drow * C9_S + dxx) * 16) : 0u;
    const unsigned long long dmask = __builtin_amdgcn_ballot_w64(dok);
    auto dma_piece = [&](const unsigned char* src, unsigned dstbuf, bool live, int c) {
        const unsigned long long base = (unsigned long long)(src + (size_t)c * C9_GBLK);
        const unsigned long long mask = live ? dmask : 0ull;
        const unsigned dst = dstbuf + (unsigned)(c * C9_LBLK + wave * 1024);
        asm volatile("s_mov_b64 exec, %0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1"
                     :
                     : "s"(mask), "s"(dst), "v"(dsrc), "s"(base)
                     : "memory");
    };

    // this lane's 6 output positions: LDS byte offset of the (-1, -1) neighbour of its cell in its own chunk half (low 16 bits), its
    // board position (high 16 bits; 0xffff = computed but not stored: the padding slots of this half's map)
    unsigned lmap[C9_NCT];
#pragma unroll
    for (int ct = 0; ct < C9_NCT; ++ct) {
        const unsigned tp = c9_maps[hf].pos[ct * 32 + l31];
        const unsigned gp = tp != 0xffffu ? tp + (unsigned)(r0 * C9_S) : 0xffffu;
        lmap[ct] = (unsigned)((c9_maps[hf].cell[ct * 32 + l31] - C9_CELL0) * 16 + hi * C9_LBLK) | (gp << 16);
    }
    unsigned long long smask[C9_NCT];  // lanes of a column tile that store their result
#pragma unroll
    for (int ct = 0; ct < C9_NCT; ++ct) smask[ct] = __builtin_amdgcn_ballot_w64((lmap[ct] >> 16) != 0xffffu);
    // 8-byte store under a lane mask, no branch: the instruction is always issued (the counted vmcnt below relies on it)
    auto store8 = [&](unsigned char* base, unsigned voff, unsigned a, unsigned b2, unsigned long long mask) {
        const cv_u32x2 d = (cv_u32x2){a, b2};
        asm volatile("s_mov_b64 exec, %0\n\tglobal_store_dwordx2 %1, %2, %3\n\ts_mov_b64 exec, -1" : : "s"(mask), "v"(voff), "v"(d), "s"(base) : "memory");
    };

    cv_bf16x8 bb[R];  // ring of B fragments: k-step t of unit u lives in slot (u NSTEP + t) % R
    auto load_step = [&](const unsigned char* p0, int st, int slot) {
        const int tap = st / KS, ks = st % KS;
        bb[slot] = *(const cv_bf16x8*)(p0 + ((tap / 3) * C9_PITCH + (tap % 3)) * 16 + ks * (2 * C9_LBLK));
    };

    if (s < nboards) {  // first tile: all pieces at once
        const unsigned char* src = x + (size_t)s * xboard + (size_t)x_chunk0 * C9_GBLK;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) dma_piece(src, lds0, true, i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CV_BARRIER();
#pragma unroll
    for (int st = 0; st < R - 1; ++st) load_step(lds + (lmap[0] & 0xffffu), st, st);
#pragma unroll
    for (int t = 0; t < NSTEP; ++t) {  // the compiler's wait for the weight loads belongs in front of the loop (see az_conv.h)
        if (t < 64) asm volatile("" : : "a"(wf[t]));
        else asm volatile("" : : "v"(wf[t]));
    }

    cv_f32x16 acc[2];   // [accumulator set = unit parity]
    cv_u32x2 rr[4];     // addend of the unit whose epilogue comes next: [register quad]
    acc[0] = *bias_ptr;  // (bias_lds was published by the barrier above)
    acc[1] = *bias_ptr;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq) rr[rq] = (cv_u32x2){0u, 0u};
    float ev[4];
    unsigned epa = 0, epb = 0;
    // one epilogue micro-op of register quad rq of the unit that computed column tile ct
    auto epi_op = [&](int set, int ct, unsigned char* out, int rq, int op, bool store_ok) {
        if (ADD) {
            if (op == 0) ev[0] = cw_add_f32(acc[set][rq * 4 + 0], cv_bf16_lo(rr[rq].x));
            else if (op == 1) ev[1] = cw_add_f32(acc[set][rq * 4 + 1], cv_bf16_hi(rr[rq].x));
            else if (op == 2) ev[2] = cw_add_f32(acc[set][rq * 4 + 2], cv_bf16_lo(rr[rq].y));
            else if (op == 3) ev[3] = cw_add_f32(acc[set][rq * 4 + 3], cv_bf16_hi(rr[rq].y));
            else if (op == 4) epa = cw_pk_bf16(ev[0], ev[1]);
            else if (op == 5) epb = cw_pk_bf16(ev[2], ev[3]);
        } else {
            if (op == 0) epa = cw_pk_bf16(acc[set][rq * 4 + 0], acc[set][rq * 4 + 1]);
            else if (op == 1) epb = cw_pk_bf16(acc[set][rq * 4 + 2], acc[set][rq * 4 + 3]);
        }
        if (op == SC::OPS - 3) epa = cw_pk_max_i16(epa, lo16);
        else if (op == SC::OPS - 2) epb = cw_pk_max_i16(epb, lo16);
        else if (op == SC::OPS - 1) {
            unsigned lm = lmap[ct];
            asm volatile("" : "+v"(lm));  // recompute the offset here (two VALU in an MFMA gap) instead of hoisting 24 of them out of the loop
            store8(out, (unsigned)(rq * C9_GBLK) + (lm >> 16) * 16u + (unsigned)(hi * 8), epa, epb, store_ok ? smask[ct] : 0ull);
        }
    };

    int it = 0;
    unsigned char* yprev = y;  // output base of the previous tile (the epilogue of its last unit runs inside this tile's unit 0)
    for (int board = s; board < nboards; board += nst, ++it) {
        const int buf = it & 1;
        const unsigned char* Xs = lds + buf * LBUF;
        const unsigned char* Xn = lds + (buf ^ 1) * LBUF;
        const bool has_next = board + nst < nboards;
        const unsigned char* nsrc = x + (size_t)(has_next ? board + nst : board) * xboard + (size_t)x_chunk0 * C9_GBLK;
        const unsigned ndst = lds0 + (unsigned)((buf ^ 1) * LBUF);
        const size_t obase = (size_t)board * OTILE + (size_t)(ch * 16 + wave * 4) * C9_GBLK;
        const unsigned char* abase = ADD ? add + obase : nullptr;
        unsigned char* ybase = y + obase;
        const bool have_prev = it > 0;
        cp_for_each([&](auto UC) __attribute__((always_inline)) {
            constexpr int u = decltype(UC)::value, set = u & 1, pset = set ^ 1;
            constexpr int pct = (u + NU - 1) % NU;  // the previous unit's column tile
            unsigned char* pout = u == 0 ? yprev : ybase;
            const bool pstore = u > 0 || have_prev;
            const unsigned char* b0 = Xs + (lmap[u] & 0xffffu);
            const unsigned char* nb0 = (u < NU - 1 ? Xs : Xn) + (lmap[(u + 1) % NU] & 0xffffu);
            cp_for_each([&](auto TC) __attribute__((always_inline)) {
                constexpr int t = decltype(TC)::value;
                if constexpr (u == NU - 1 && t == SC::T_BAR) {
                    // every read of this buffer has been issued (the ring runs R - 1 k-steps ahead); all DMA pieces of the next tile are
                    // older than the VM_AFTER_DMA youngest vector-memory operations of this wave
                    if (have_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_AFTER_DMA) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // first tile: its unit-0 epilogue stores were skipped
                    CV_BARRIER();
                }
                if constexpr (t + R - 1 < NSTEP) load_step(b0, t + R - 1, (u * NSTEP + t + R - 1) % R);
                else load_step(nb0, t + R - 1 - NSTEP, (u * NSTEP + t + R - 1) % R);
                if constexpr (t < 64) cw_mfma_a(acc[set], wf[t], bb[(u * NSTEP + t) % R]);
                else cw_mfma_v(acc[set], wf[t], bb[(u * NSTEP + t) % R]);
                // ---- riders of this MFMA gap -----------------------------------------------------------------------------------
#pragma unroll
                for (int k = 0; k < SC::PER; ++k) {  // the previous unit's epilogue, PER micro-ops per gap
                    const int o = (t - SC::S0) * SC::PER + k;
                    if (t >= SC::S0 && o < SC::NQ * SC::OPS) epi_op(pset, pct, pout, o / SC::OPS, o % SC::OPS, pstore);
                }
                if constexpr (ADD) {  // this unit's addend of quad q, right behind the store of the previous unit's quad q
#pragma unroll
                    for (int q = 0; q < SC::NQ; ++q)
                        if (t == SC::store_slot(q)) {
                            unsigned lm = lmap[u];
                            asm volatile("" : "+v"(lm));  // (not hoisted, see epi_op)
                            const unsigned gp = lm >> 16;
                            rr[q] = *(const cv_u32x2*)(abase + q * C9_GBLK + ((gp == 0xffffu ? 0u : gp) * 16u + (unsigned)(hi * 8)));
                        }
                }
                if constexpr (u == 0 && t % 4 == 1 && t / 4 < NPIECE) dma_piece(nsrc, ndst, has_next, t / 4);
                if constexpr (t == NSTEP - 2) acc[pset] = *bias_ptr;  // the next unit's accumulators start from the bias (its epilogue riders are long done)
                __builtin_amdgcn_sched_barrier(0);
            }, typename CpMakeSeq<NSTEP>::type{});
        }, typename CpMakeSeq<NU>::type{});
        yprev = ybase;
    }
    // epilogue of the very last unit: column tile NU - 1, accumulator set (NU - 1) & 1
    if (it > 0) {
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]));
#pragma unroll
        for (int o = 0; o < SC::NQ * SC::OPS; ++o) epi_op((NU - 1) & 1, NU - 1, yprev, o / SC::OPS, o % SC::OPS, true);
    }
}
// =====================================================================================================================================
// k_conv3x3_op19<ADD> (round 6) -- the SAME convolution in ONE launch: y = act(conv3x3(x, w) + bias [+ residual]), all 256 input
// channels contracted inside one CU, fp32 accumulation end to end (no bf16 partial sum through HBM, one rounding per output).
//   * a CU owns 64 couts x 256 cin of one board half: wave (a, b) holds couts [32 a, 32 a + 32) x cin half b -- the per-wave shape of
//     k_conv3x3_hb19 (288 weight registers, 72 v_mfma_f32_32x32x16 per column tile).  The two cin halves of a cout group meet through
//     LDS: after a unit (= column tile) each wave hands the accumulator quads its PARTNER finalises (8 registers per lane) to the
//     partner and runs the epilogue (bias is in wave b = 0's accumulators; residual, rounding, ReLU, store) of its own two quads.
//     The weight rows of wave b = 1 are rotated by 16 couts so that "own quads" are registers 0-7 in both waves (compile-time indices).
//   * the tile image holds all 32 input chunks of the half board: 123,904 B -- ONE buffer.  It is refilled on the fly: the column
//     tiles take the half's positions in row-major order (conflict-free: C1Map), so unit u only reads image cells [~34 u, ~34 u + 76); the strip of a chunk
//     is four DMA BANDS of 64 cells, band 0 is dead after unit 1, band 1 after unit 3, bands 2-3 after unit 5 (compile-time checked
//     against the lane maps), and the next tile's band is DMA'd into the same cells as soon as a workgroup barrier has certified that
//     every wave is past its last reader.  One barrier per unit does triple duty: partial-sum exchange, "band free", "band landed"
//     (behind counted s_waitcnt vmcnt):
//         unit 0: B0 -> DMA band 2 of THIS tile      unit 1: B1 [band 2 landed] -> DMA band 3 of this tile
//         unit 2: B2 -> DMA band 0 of the NEXT tile   unit 3: B3 [band 3 landed]
//         unit 4: B4 -> DMA band 1 of the next tile   unit 5: B5 [bands 0, 1 of the next tile landed]
//     Every band has >= 56 MFMA slots (~1.8 k matrix-pipe cycles at full rate, more at the real clock) between its last piece and
//     the barrier that needs it.
//   * HBM traffic: x once (the four cout groups of a tile stream run on one XCD: one HBM read, three L2 hits), residual once, y once
//     = 2 / 3 tensor passes per convolution instead of 4 / 5; bytes arriving in CUs: 4 cout groups x 1.2 (halo rows) = 4.8 passes
//     (two launches: 2.4).
#define C1_NCH 32
#define C1_IMG (C1_NCH * C9_LBLK)          // 123,904 B
#define C1_XB_WAVE 2048                    // exchange: 2 quads x 64 lanes x 16 B per wave
#define C1_XB (2 * 4 * C1_XB_WAVE)         // [unit parity][wave]
#define C1_NBAND 4
#define C1_VARIANT 8  // 8 = k_conv3x3_op19q (the four-way cin split, below); 0 = k_conv3x3_op19; 6 = its counted-lgkmcnt / later-barrier build

struct C1Map {
    unsigned short cell[C9_NCT * 32], pos[C9_NCT * 32];  // pos: tile-relative (row * 19 + col), 0xffff = computed, never stored
    short rmin[C9_NCT], rmax[C9_NCT];                    // first / last image cell the real positions of a column tile read (taps included)
    bool ok;
};
constexpr C1Map c1_make_map(int hf) {
    C1Map m{};
    const int lanes[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    int own[C9_NCT * 32] = {}, n = 0;
    for (int p = 0; p < C9_NPOS; ++p)
        if (c9_in_map(hf, p)) own[n++] = p;  // the half's own positions, row-major
    bool ok = n > (C9_NCT - 1) * 32 && n <= C9_NCT * 32;
    for (int i = 0; i < C9_NCT * 32; ++i) {
        m.cell[i] = 0;
        m.pos[i] = 0xffff;
    }
    // Row-major order, conflict-free: a position goes into the first of the tile's two service groups (16 lanes each) that does not hold its
    // bank residue (cell mod 16) yet; if both do, it waits for the next column tile (at most a handful do: consecutive cells carry
    // consecutive residues, only the zero cell at a row end shifts them).  A tile closes when it is full or 6 positions are waiting.
    int carry[16] = {}, ncarry = 0, next = 0;
    for (int ct = 0; ct < C9_NCT; ++ct) {
        bool used[2][16] = {};
        int fill[2] = {0, 0}, cmin = 1 << 20, cmax = -1;
        int wait[16] = {}, nwait = 0;
        auto place = [&](int p) {
            const int cell = C9_CELL0 + C9_PITCH * (p / C9_S) + p % C9_S, r = cell & 15;
            for (int g = 0; g < 2; ++g)
                if (!used[g][r] && fill[g] < 16) {
                    const int idx = ct * 32 + lanes[g][fill[g]++];
                    m.cell[idx] = (unsigned short)cell;
                    m.pos[idx] = (unsigned short)p;
                    used[g][r] = true;
                    cmin = cell < cmin ? cell : cmin;
                    cmax = cell > cmax ? cell : cmax;
                    return true;
                }
            return false;
        };
        for (int i = 0; i < ncarry; ++i)
            if (!place(carry[i])) wait[nwait++] = carry[i];
        while (next < n && fill[0] + fill[1] < 32 && nwait <= 6) {
            const int p = own[next++];
            if (!place(p)) wait[nwait++] = p;
        }
        ncarry = nwait;
        for (int i = 0; i < nwait; ++i) carry[i] = wait[i];
        if (ct == C9_NCT - 1 && (ncarry > 0 || next < n)) ok = false;  // every position found a slot
        for (int g = 0; g < 2; ++g)  // padding slots: a cell of this column tile's own span (same bands), of a residue the group lacks if there is one
            while (fill[g] < 16) {
                int pick = cmin;
                for (int c = cmin; c <= cmax; ++c)
                    if (!used[g][c & 15] && (c - 1) % C9_PITCH != C9_S) {  // (not the zero cell between two rows)
                        pick = c;
                        break;
                    }
                m.cell[ct * 32 + lanes[g][fill[g]++]] = (unsigned short)pick;
                used[g][pick & 15] = true;
            }
        m.rmin[ct] = (short)(cmin - C9_CELL0);
        m.rmax[ct] = (short)(cmax + C9_CELL0);
        for (int g = 0; g < 2; ++g) {  // every service group: 16 slots, real positions on 16 distinct bank residues
            bool seen[16] = {};
            if (fill[g] != 16) ok = false;
            for (int i = 0; i < 16; ++i) {
                const int idx = ct * 32 + lanes[g][i];
                if (m.pos[idx] == 0xffff) continue;
                if (seen[m.cell[idx] & 15]) ok = false;
                seen[m.cell[idx] & 15] = true;
            }
        }
    }
    m.ok = ok;
    return m;
}
constexpr bool c1_maps_cover_the_board() {
    const C1Map m0 = c1_make_map(0), m1 = c1_make_map(1);
    int seen[C9_P2] = {};
    for (int i = 0; i < C9_NCT * 32; ++i) {
        if (m0.pos[i] != 0xffff) seen[m0.pos[i]]++;
        if (m1.pos[i] != 0xffff) seen[m1.pos[i] + 9 * C9_S]++;
    }
    for (int p = 0; p < C9_P2; ++p)
        if (seen[p] != 1) return false;
    return true;
}
// the refill schedule of the single-buffered image: band 0 (cells 0-63) is read by units 0-1 only, band 1 (64-127) by units 0-3, band 2
// (128-191) by units 2-5, band 3 (192-241) by units 4-5
constexpr bool c1_band_schedule_ok(int hf) {
    const C1Map m = c1_make_map(hf);
    return m.ok && m.rmin[0] >= 0 && m.rmax[C9_NCT - 1] < C9_CELLS && m.rmin[2] >= 64 && m.rmin[4] >= 128 && m.rmax[1] <= 127 && m.rmax[3] <= 191;
}
static_assert(c1_band_schedule_ok(0) && c1_band_schedule_ok(1), "19x19 one-pass kernel: the DMA band schedule does not match the column-tile maps");
static_assert(c1_maps_cover_the_board(), "19x19 one-pass maps: every position stored exactly once");
static __device__ const C1Map c1_maps[2] = {c1_make_map(0), c1_make_map(1)};

// V: build variant (bit 0: 6 ring slots instead of 4; bit 1: the barrier waits for the hand-over writes only -- a counted lgkmcnt -- instead
// of draining the fragment ring; bit 2: barrier in front of slot 12 instead of 8).  The product launches C1_VARIANT; the others exist
// for same-box A/B runs (AZSP_OP19_VARIANT, read once per process).
template <bool ADD, int V> struct C1Sched {  // static schedule of one unit = one column tile = 72 k-steps per wave (its cin half)
    static constexpr int KS = 8, NSTEP = 72, NU = C9_NCT, R = (V & 1) ? 6 : 4;
    static constexpr bool LGKM_COUNTED = (V & 2) != 0;
    static constexpr int S0 = 5;            // first slot that may touch the previous unit's accumulators: the hand-over writes ride in S0, S0 + 1
    static constexpr int SB = (V & 4) ? 12 : 8;  // the unit's barrier sits in front of slot SB
    // LDS operations this wave issues between its second hand-over write (rider of slot S0 + 1) and the barrier: one fragment read per slot
    static constexpr int LGKM_AFTER_XWRITE = SB - (S0 + 1) - 1;
    static constexpr int OPS = ADD ? 9 : 5;  // epilogue micro-ops per register quad
    // rider stream behind the barrier, one micro-op per slot from SB + 1: 2 reads of the partner's quads, then per own quad 4 adds + OPS epilogue ops
    static constexpr int NRID = 2 + 2 * (4 + OPS);
    static constexpr int store_slot(int q) { return SB + 1 + 2 + (q + 1) * (4 + OPS) - 1; }
    static constexpr int NPIECE = 8;        // DMA pieces per wave per band: its 8 chunks
    static constexpr int dma_slot(int i) { return SB + 2 + 2 * i; }
    static constexpr bool dma_unit(int u) { return u == 0 || u == 1 || u == 2 || u == 4; }
    // Vector-memory operations of a wave, in program order, are the same in every unit of every tile (stores, addend loads and DMA pieces
    // are issued under masks, never skipped), and ALL of them are inline assembly: the compiler inserts no vmcnt wait of its own inside
    // the tile loop (a compiler-issued addend load would make it wait on "everything but its own younger loads", i.e. on the DMA pieces
    // issued a few slots earlier: a full memory latency per unit).  In slot t of unit u: first the riders (the store of the previous
    // unit's quad q and, behind it, this unit's addend load of quad q, both in slot store_slot(q)), then the DMA piece.
    static constexpr int rider_vm(int t) {
        int n = 0;
        for (int q = 0; q < 2; ++q)
            if (t == store_slot(q)) n += ADD ? 2 : 1;
        return n;
    }
    static constexpr bool dma_at(int u, int t) {
        for (int i = 0; i < NPIECE; ++i)
            if (dma_unit(u) && t == dma_slot(i)) return true;
        return false;
    }
    static constexpr int vm_ops(int u, int t) { return rider_vm(t) + (dma_at(u, t) ? 1 : 0); }
    // ... issued after the last DMA piece of unit ui and before the barrier of unit uw (ui < uw, same tile)
    static constexpr int vm_between(int ui, int uw) {
        int n = 0;
        for (int t = dma_slot(NPIECE - 1) + 1; t < NSTEP; ++t) n += vm_ops(ui, t);
        for (int u = ui + 1; u < uw; ++u)
            for (int t = 0; t < NSTEP; ++t) n += vm_ops(u, t);
        for (int t = 0; t < SB; ++t) n += vm_ops(uw, t);
        return n;
    }
    // first rider slot that reads the addend of own quad q (loaded in the previous unit's slot store_slot(q)), and the number of
    // vector-memory operations issued between that load and the start of this slot in unit u
    static constexpr int addend_use_slot(int q) { return SB + 1 + 2 + q * (4 + OPS) + 4; }
    static constexpr int vm_after_addend(int u, int q) {
        const int pu = (u + NU - 1) % NU;
        int n = dma_at(pu, store_slot(q)) ? 1 : 0;  // (the DMA piece of the load's own slot comes after it)
        for (int t = store_slot(q) + 1; t < NSTEP; ++t) n += vm_ops(pu, t);
        for (int t = 0; t < addend_use_slot(q); ++t) n += vm_ops(u, t);
        return n;
    }
    static_assert(SB + 1 + NRID < NSTEP - 3, "the previous unit's epilogue must be done before its accumulator set is re-initialised");
    static_assert(dma_slot(NPIECE - 1) < NSTEP - 4 && S0 + 1 < SB, "slot layout");
    static_assert((NU * NSTEP) % R == 0, "a tile's k-steps keep the ring phase");
    static_assert(vm_between(0, 1) < 63 && vm_between(1, 3) < 63 && vm_between(4, 5) < 63, "vmcnt field");
    static_assert(vm_after_addend(0, 0) < 63 && vm_after_addend(1, 1) < 63 && vm_after_addend(2, 1) < 63 && vm_after_addend(3, 1) < 63, "vmcnt field");
    static_assert(addend_use_slot(0) < store_slot(0) && addend_use_slot(1) > store_slot(0) && addend_use_slot(1) < store_slot(1), "an addend register is re-loaded after its last use");
};

template <bool ADD, int V> __global__ void __launch_bounds__(CW_THREADS, 1)
k_conv3x3_op19(const unsigned char* __restrict__ x, const unsigned short* __restrict__ w, const float* __restrict__ bias, const unsigned char* add,
               unsigned char* y, int nboards, int relu) {
    typedef C1Sched<ADD, V> SC;
    constexpr int KS = SC::KS, NSTEP = SC::NSTEP, R = SC::R, NU = SC::NU, SB = SC::SB;
    constexpr int OTILE = 32 * C9_GBLK;   // one board of x / y / addend: 32 chunks
    __shared__ __attribute__((aligned(1024))) unsigned char lds[C1_IMG + C1_XB];
    __shared__ __attribute__((aligned(64))) float bias_lds[4 * 2 * 16];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave >> 1, wb = wave & 1;  // cout half of the CU's 64 couts, cin half
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    for (int i = tid; i < (C1_IMG + C1_XB) / 16; i += CW_THREADS) *(cv_u32x4*)(lds + i * 16) = (cv_u32x4){0u, 0u, 0u, 0u};
    __syncthreads();  // the zero cells are in place before any wave's DMA lands

    // this CU's role: tile stream s, cout group cg (64 couts), board half hf.  With a full grid the eight roles of a stream share an XCD
    // (block b runs on XCD b % 8): the second to fourth cout group find the input tile in that XCD's L2.
    const int bi = (int)blockIdx.x, nst = (int)gridDim.x >> 3;
    int s, role;
    if ((gridDim.x & 63u) == 0u) {
        s = (bi & 7) + 8 * (bi >> 6);
        role = (bi >> 3) & 7;
    } else {
        s = bi >> 3;
        role = bi & 7;
    }
    const int cg = role >> 1, hf = role & 1, r0 = hf * 9;

    cv_bf16x8 wf[NSTEP];  // this wave's 32 couts x (9 taps x 128 cin of its half); row m of the A operand = cout 32 a + ((m + 16 b) & 31)
    const int crow = cg * 64 + wa * 32 + ((l31 + 16 * wb) & 31);
#pragma unroll
    for (int t = 0; t < NSTEP; ++t)
        wf[t] = *(const cv_bf16x8*)(w + ((size_t)((t / KS) * 256 + crow)) * 256 + wb * 128 + ((t % KS) * 2 + hi) * 8);
    // bias in the accumulator layout ([wave][lane half][16 floats]); it enters through wave b = 0's accumulators only
    if (lane < 32) bias_lds[(wave * 2 + (lane >> 4)) * 16 + (lane & 15)] = wb == 0 ? bias[cg * 64 + wa * 32 + 8 * ((lane & 15) >> 2) + 4 * (lane >> 4) + (lane & 3)] : 0.0f;
    const cv_f32x16* bias_ptr = (const cv_f32x16*)(bias_lds + (wave * 2 + hi) * 16);
    const unsigned lo16 = relu ? 0u : 0x80008000u;

    // LDS-DMA: band j = cells [64 j, 64 j + 64) of a chunk strip, lane -> cell 64 j + lane; wave w moves chunks c = w + 4 i
    unsigned dsrc[C1_NBAND];
    unsigned long long dmask[C1_NBAND];
#pragma unroll
    for (int j = 0; j < C1_NBAND; ++j) {
        const int dcell = j * 64 + lane, dk = dcell - 1, drs = dk / C9_PITCH, dxx = dk - drs * C9_PITCH, drow = r0 - 1 + drs;
        const bool dok = dcell >= 1 && dcell < C9_CELLS - 1 && dxx < C9_S && drow >= 0 && drow < C9_S;
        dsrc[j] = dok ? (unsigned)((drow * C9_S + dxx) * 16) : 0u;
        dmask[j] = __builtin_amdgcn_ballot_w64(dok);
    }
    auto dma_piece = [&](const unsigned char* src, bool live, int j, int i) {
        const int c = wave + 4 * i;
        const unsigned long long base = (unsigned long long)(src + (size_t)c * C9_GBLK);
        const unsigned long long mask = live ? dmask[j] : 0ull;
        const unsigned dst = lds0 + (unsigned)(c * C9_LBLK + j * 1024);
        asm volatile("s_mov_b64 exec, %0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1"
                     :
                     : "s"(mask), "s"(dst), "v"(dsrc[j]), "s"(base)
                     : "memory");
    };

    // this lane's 6 output positions: LDS byte offset of the (-1, -1) neighbour of its cell in the first chunk of its k-half (low 16 bits),
    // board position (high 16 bits; 0xffff = a padding slot: computed, never stored)
    unsigned lmap[C9_NCT];
#pragma unroll
    for (int ct = 0; ct < C9_NCT; ++ct) {
        const unsigned tp = c1_maps[hf].pos[ct * 32 + l31];
        const unsigned gp = tp != 0xffffu ? tp + (unsigned)(r0 * C9_S) : 0xffffu;
        lmap[ct] = (unsigned)((c1_maps[hf].cell[ct * 32 + l31] - C9_CELL0) * 16 + hi * C9_LBLK) | (gp << 16);
    }
    unsigned long long smask[C9_NCT];
#pragma unroll
    for (int ct = 0; ct < C9_NCT; ++ct) smask[ct] = __builtin_amdgcn_ballot_w64((lmap[ct] >> 16) != 0xffffu);
    auto store8 = [&](unsigned char* base, unsigned voff, unsigned a, unsigned b2, unsigned long long mask) {
        const cv_u32x2 d = (cv_u32x2){a, b2};
        asm volatile("s_mov_b64 exec, %0\n\tglobal_store_dwordx2 %1, %2, %3\n\ts_mov_b64 exec, -1" : : "s"(mask), "v"(voff), "v"(d), "s"(base) : "memory");
    };

    const unsigned char* Xw = lds + wb * (16 * C9_LBLK);  // this wave's cin half of the image
    cv_bf16x8 bb[R];
    auto load_step = [&](const unsigned char* p0, int st, int slot) {
        const int tap = st / KS, ks = st % KS;
        bb[slot] = *(const cv_bf16x8*)(p0 + ((tap / 3) * C9_PITCH + (tap % 3)) * 16 + ks * (2 * C9_LBLK));
    };
    // hand-over buffers: [unit parity][wave][quad j][lane] 16 B
    unsigned char* const xb_mine = lds + C1_IMG + wave * C1_XB_WAVE + lane * 16;
    const unsigned char* const xb_partner = lds + C1_IMG + (wave ^ 1) * C1_XB_WAVE + lane * 16;

    if (s < nboards) {  // first tile: the whole image at once
        const unsigned char* src = x + (size_t)s * OTILE;
#pragma unroll
        for (int j = 0; j < C1_NBAND; ++j)
#pragma unroll
            for (int i = 0; i < SC::NPIECE; ++i) dma_piece(src, true, j, i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CV_BARRIER();
#pragma unroll
    for (int st = 0; st < R - 1; ++st) load_step(Xw + (lmap[0] & 0xffffu), st, st);
#pragma unroll
    for (int t = 0; t < NSTEP; ++t) {  // the compiler's wait for the weight loads belongs in front of the loop (see az_conv.h)
        if (t < 64) asm volatile("" : : "a"(wf[t]));
        else asm volatile("" : : "v"(wf[t]));
    }

    cv_f32x16 acc[2];     // [accumulator set = unit parity]
    cv_u32x2 rr[2];       // addend of the unit whose epilogue comes next: [own quad]
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    f32x4 xr[2];          // the partner's partial sums of my two quads
    acc[0] = *bias_ptr;   // (bias_lds was published by the barrier above)
    acc[1] = *bias_ptr;
    rr[0] = rr[1] = (cv_u32x2){0u, 0u};
    xr[0] = xr[1] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    float ev[4];
    unsigned epa = 0, epb = 0;
    // hand the partner's quads (registers 8-15 of set `set`) over: one ds_write_b128 per quad
    auto xwrite = [&](int set, int j) {
        *(f32x4*)(xb_mine + (set * 4 * C1_XB_WAVE) + j * 1024) = (f32x4){acc[set][8 + 4 * j], acc[set][9 + 4 * j], acc[set][10 + 4 * j], acc[set][11 + 4 * j]};
    };
    auto xread = [&](int set, int j) { xr[j] = *(const f32x4*)(xb_partner + (set * 4 * C1_XB_WAVE) + j * 1024); };
    // micro-op `o` of the rider stream that finishes the unit which accumulated into `set` (column tile ct)
    auto rider = [&](int set, int ct, unsigned char* out, int o, bool store_ok) {
        if (o < 2) {
            xread(set, o);
            return;
        }
        const int q = (o - 2) / (4 + SC::OPS), k = (o - 2) % (4 + SC::OPS);
        if (k < 4) {
            acc[set][q * 4 + k] = cw_add_f32(acc[set][q * 4 + k], xr[q][k]);
            return;
        }
        const int op = k - 4;
        if (ADD) {
            if (op == 0) ev[0] = cw_add_f32(acc[set][q * 4 + 0], cv_bf16_lo(rr[q].x));
            else if (op == 1) ev[1] = cw_add_f32(acc[set][q * 4 + 1], cv_bf16_hi(rr[q].x));
            else if (op == 2) ev[2] = cw_add_f32(acc[set][q * 4 + 2], cv_bf16_lo(rr[q].y));
            else if (op == 3) ev[3] = cw_add_f32(acc[set][q * 4 + 3], cv_bf16_hi(rr[q].y));
            else if (op == 4) epa = cw_pk_bf16(ev[0], ev[1]);
            else if (op == 5) epb = cw_pk_bf16(ev[2], ev[3]);
        } else {
            if (op == 0) epa = cw_pk_bf16(acc[set][q * 4 + 0], acc[set][q * 4 + 1]);
            else if (op == 1) epb = cw_pk_bf16(acc[set][q * 4 + 2], acc[set][q * 4 + 3]);
        }
        if (op == SC::OPS - 3) epa = cw_pk_max_i16(epa, lo16);
        else if (op == SC::OPS - 2) epb = cw_pk_max_i16(epb, lo16);
        else if (op == SC::OPS - 1) {
            unsigned lm = lmap[ct];
            asm volatile("" : "+v"(lm));  // recompute the offset here instead of hoisting 12 of them out of the loop
            store8(out, (unsigned)(q * C9_GBLK) + (lm >> 16) * 16u + (unsigned)(hi * 8), epa, epb, store_ok ? smask[ct] : 0ull);
        }
    };

    int it = 0;
    unsigned char* yprev = y;
    for (int board = s; board < nboards; board += nst, ++it) {
        const bool has_next = board + nst < nboards, have_prev = it > 0;  // (the first tile issues the same operations, masked off)
        const unsigned char* csrc = x + (size_t)board * OTILE;                                   // bands 2, 3 of THIS tile (units 0, 1)
        const unsigned char* nsrc = x + (size_t)(has_next ? board + nst : board) * OTILE;        // bands 0, 1 of the NEXT tile (units 2, 4)
        const size_t obase = (size_t)board * OTILE + (size_t)(cg * 8 + wa * 4 + wb * 2) * C9_GBLK;  // my two quads = chunks 2 b, 2 b + 1 of the wave pair's four
        const unsigned char* abase = ADD ? add + obase : nullptr;
        unsigned char* ybase = y + obase;
        cp_for_each([&](auto UC) __attribute__((always_inline)) {
            constexpr int u = decltype(UC)::value, set = u & 1, pset = set ^ 1;
            constexpr int pct = (u + NU - 1) % NU;  // the previous unit's column tile
            unsigned char* pout = u == 0 ? yprev : ybase;
            const bool pstore = u > 0 || have_prev;
            const unsigned char* b0 = Xw + (lmap[u] & 0xffffu);
            const unsigned char* nb0 = Xw + (lmap[(u + 1) % NU] & 0xffffu);
            cp_for_each([&](auto TC) __attribute__((always_inline)) {
                constexpr int t = decltype(TC)::value;
                if constexpr (t == SB) {
                    // B_u: every wave has handed over the previous unit's partial sums, has finished the reads of the unit before, and
                    // (units 1, 3, 5) its DMA pieces of the band the next reads need are older than the N youngest vector-memory operations
                    if constexpr (u == 1 || u == 3 || u == 5) {
                        constexpr int N = u == 1 ? SC::vm_between(0, 1) : u == 3 ? SC::vm_between(1, 3) : SC::vm_between(4, 5);
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
                    }
                    // LDS completes a wave's operations in order and the tile loop issues no scalar loads: "all but the N youngest" covers the
                    // hand-over writes and leaves the fragment ring's look-ahead reads in flight
                    if constexpr (SC::LGKM_COUNTED) asm volatile("s_waitcnt lgkmcnt(%0)\n\ts_barrier" ::"n"(SC::LGKM_AFTER_XWRITE) : "memory");
                    else CV_BARRIER();
                }
                if constexpr (t + R - 1 < NSTEP) load_step(b0, t + R - 1, (u * NSTEP + t + R - 1) % R);
                else load_step(nb0, t + R - 1 - NSTEP, (u * NSTEP + t + R - 1) % R);
                if constexpr (t < 64) cw_mfma_a(acc[set], wf[t], bb[(u * NSTEP + t) % R]);
                else cw_mfma_v(acc[set], wf[t], bb[(u * NSTEP + t) % R]);
                // ---- riders of this MFMA gap -----------------------------------------------------------------------------------
                if constexpr (t == SC::S0 || t == SC::S0 + 1) xwrite(pset, t - SC::S0);
                if constexpr (ADD) {  // the previous unit's addend of quad q has landed: all but the N youngest vector-memory operations are done
                    if constexpr (t == SC::addend_use_slot(0)) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rr[0]) : "n"(SC::vm_after_addend(u, 0)));
                    if constexpr (t == SC::addend_use_slot(1)) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rr[1]) : "n"(SC::vm_after_addend(u, 1)));
                }
                if constexpr (t > SB && t - SB - 1 < SC::NRID) rider(pset, pct, pout, t - SB - 1, pstore);
                if constexpr (ADD) {  // this unit's addend of own quad q, right behind the store of the previous unit's quad q
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        if (t == SC::store_slot(q)) {
                            unsigned lm = lmap[u];
                            asm volatile("" : "+v"(lm));
                            const unsigned gp = lm >> 16;
                            const unsigned voff = (unsigned)(q * C9_GBLK) + (gp == 0xffffu ? 0u : gp) * 16u + (unsigned)(hi * 8);
                            asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(rr[q]) : "v"(voff), "s"(abase) : "memory");
                        }
                }
                if constexpr (SC::dma_unit(u) && t >= SC::dma_slot(0) && t <= SC::dma_slot(SC::NPIECE - 1) && (t - SC::dma_slot(0)) % 2 == 0) {
                    constexpr int i = (t - SC::dma_slot(0)) / 2;
                    if constexpr (u == 0) dma_piece(csrc, have_prev, 2, i);
                    else if constexpr (u == 1) dma_piece(csrc, have_prev, 3, i);
                    else if constexpr (u == 2) dma_piece(nsrc, has_next, 0, i);
                    else dma_piece(nsrc, has_next, 1, i);
                }
                if constexpr (t == NSTEP - 2) acc[pset] = *bias_ptr;  // the next unit's accumulators start from the bias (wave b = 1: zero)
                __builtin_amdgcn_sched_barrier(0);
            }, typename CpMakeSeq<NSTEP>::type{});
        }, typename CpMakeSeq<NU>::type{});
        yprev = ybase;
    }
    // the very last unit (column tile NU - 1, accumulator set (NU - 1) & 1): hand over, meet, finish
    if (it > 0) {
        constexpr int lset = (NU - 1) & 1;
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]));
        xwrite(lset, 0);
        xwrite(lset, 1);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rr[0]), "+v"(rr[1]));  // the last unit's addends
        CV_BARRIER();
#pragma unroll
        for (int o = 0; o < SC::NRID; ++o) rider(lset, NU - 1, yprev, o, true);
    }
}
// =====================================================================================================================================
// k_conv3x3_op19q<ADD> (round 6, second step) -- the one-pass kernel with the CU's 64 couts x 256 cin split FOUR ways along cin: wave w holds
// all 64 couts (two 32-cout tiles) x the cin QUARTER w.  In k_conv3x3_op19 the waves (0, b) and (1, b) read the same B fragments (one
// fragment per 32-cycle MFMA); here a fragment is read by one wave and feeds TWO MFMAs: half the LDS fragment reads per flop -- the change
// that bought the 9x9 x 128 kernel 4 % (k_conv3x3_sp2: at the package power limit the reads that are no longer made come back as clock).
//   * every output is the sum of four partials: a wave finalises 2 of the 8 register quads (16 couts) and hands the other 6 to their owners
//     through a single 24 KB buffer [destination wave][source][quad][lane]: 6 ds_write_b128 + 6 ds_read_b128 + 24 v_add_f32 per unit (op19: 2 +
//     2 + 8).  Weight rows are rotated per wave so that the own quads are registers 0-7 of tile 0 in every wave; the bias enters through the
//     owner's accumulators only.
//   * two barriers per unit: F in front of the hand-over writes (every wave is past the previous unit, i.e. past its reads of the previous
//     hand-over: the buffer is single) and B in front of the reads (= op19's barrier: band free / band landed / partials written).
//   * a slot is one MFMA (72 per unit), a k-step (one fragment) is two slots; everything else -- image, bands, DMA schedule, lane maps,
//     counted vector-memory waits -- is k_conv3x3_op19's.
template <bool ADD> struct C1QSched {
    static constexpr int NSTEP = 72, NKS = 36, NU = C9_NCT, R = 4;  // MFMA slots / k-steps per unit, ring slots (k-steps)
    static constexpr int S0 = 5;             // barrier F in front of slot S0, the 6 hand-over writes in the slots S0 .. S0 + 5
    static constexpr int SB = 12;            // barrier B in front of slot SB
    static constexpr int OPS = ADD ? 9 : 5;
    static constexpr int NRID = 3 * 10 + 2 * OPS;  // per source: 2 reads + 8 adds; then per own quad OPS epilogue micro-ops
    static constexpr int store_slot(int q) { return SB + 1 + 30 + (q + 1) * OPS - 1; }
    static constexpr int addend_use_slot(int q) { return SB + 1 + 30 + q * OPS; }
    static constexpr int NPIECE = 8;
    static constexpr int dma_slot(int i) { return SB + 2 + 2 * i; }
    static constexpr bool dma_unit(int u) { return u == 0 || u == 1 || u == 2 || u == 4; }
    static constexpr int rider_vm(int t) {
        int n = 0;
        for (int q = 0; q < 2; ++q)
            if (t == store_slot(q)) n += ADD ? 2 : 1;
        return n;
    }
    static constexpr bool dma_at(int u, int t) {
        for (int i = 0; i < NPIECE; ++i)
            if (dma_unit(u) && t == dma_slot(i)) return true;
        return false;
    }
    static constexpr int vm_ops(int u, int t) { return rider_vm(t) + (dma_at(u, t) ? 1 : 0); }
    static constexpr int vm_between(int ui, int uw) {
        int n = 0;
        for (int t = dma_slot(NPIECE - 1) + 1; t < NSTEP; ++t) n += vm_ops(ui, t);
        for (int u = ui + 1; u < uw; ++u)
            for (int t = 0; t < NSTEP; ++t) n += vm_ops(u, t);
        for (int t = 0; t < SB; ++t) n += vm_ops(uw, t);
        return n;
    }
    static constexpr int vm_after_addend(int u, int q) {
        const int pu = (u + NU - 1) % NU;
        int n = dma_at(pu, store_slot(q)) ? 1 : 0;
        for (int t = store_slot(q) + 1; t < NSTEP; ++t) n += vm_ops(pu, t);
        for (int t = 0; t < addend_use_slot(q); ++t) n += vm_ops(u, t);
        return n;
    }
    static_assert(SB + 1 + NRID < NSTEP - 3 && S0 + 6 < SB && dma_slot(NPIECE - 1) < NSTEP - 4, "slot layout");
    static_assert((NU * NKS) % R == 0, "a tile's k-steps keep the ring phase");
    static_assert(vm_between(0, 1) < 63 && vm_between(1, 3) < 63 && vm_between(4, 5) < 63 && vm_after_addend(0, 0) < 63 && vm_after_addend(3, 1) < 63, "vmcnt field");
    static_assert(addend_use_slot(0) + 3 < store_slot(0) && addend_use_slot(1) + 3 < store_slot(1) && addend_use_slot(1) > store_slot(0), "an addend register is re-loaded after its last use");
};
// first MFMA of an accumulator that starts from zero: the C operand is the inline constant 0
__device__ __forceinline__ void c1q_mfma_a0(cv_f32x16& acc, const cv_bf16x8& wa, const cv_bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "a"(wa), "v"(b));
}
#define C1Q_XB (4 * 3 * 2048)  // hand-over buffer: [destination wave][source 0..2][quad j][lane] 16 B

template <bool ADD> __global__ void __launch_bounds__(CW_THREADS, 1)
k_conv3x3_op19q(const unsigned char* __restrict__ x, const unsigned short* __restrict__ w, const float* __restrict__ bias, const unsigned char* add,
                unsigned char* y, int nboards, int relu) {
    typedef C1QSched<ADD> SC;
    constexpr int NSTEP = SC::NSTEP, NKS = SC::NKS, R = SC::R, NU = SC::NU, SB = SC::SB;
    constexpr int OTILE = 32 * C9_GBLK;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[C1_IMG + C1Q_XB];
    __shared__ __attribute__((aligned(64))) float bias_lds[4 * 2 * 32];  // [wave][lane half][16 floats: own bias in 0-7, zeros in 8-15 | 16 zeros]
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    for (int i = tid; i < (C1_IMG + C1Q_XB) / 16; i += CW_THREADS) *(cv_u32x4*)(lds + i * 16) = (cv_u32x4){0u, 0u, 0u, 0u};
    __syncthreads();
    const int bi = (int)blockIdx.x, nst = (int)gridDim.x >> 3;
    int s, role;
    if ((gridDim.x & 63u) == 0u) {
        s = (bi & 7) + 8 * (bi >> 6);
        role = (bi >> 3) & 7;
    } else {
        s = bi >> 3;
        role = bi & 7;
    }
    const int cg = role >> 1, hf = role & 1, r0 = hf * 9;

    // A fragments: tile index tt = cout tile (tt + (wave >> 1)) & 1 of the CU's two, rows rotated by 16 (wave & 1): the wave's OWN 16 couts
    // (64 cg + 16 wave ..) are rows 0-15 of tile index 0; cin quarter `wave`; fragment (tt, k-step ks = tap * 4 + kk) at wf[tt * 36 + ks]
    cv_bf16x8 wf[NSTEP];
#pragma unroll
    for (int f = 0; f < NSTEP; ++f) {
        const int tt = f / NKS, ks = f % NKS;
        const int crow = cg * 64 + ((tt + (wave >> 1)) & 1) * 32 + ((l31 + 16 * (wave & 1)) & 31);
        wf[f] = *(const cv_bf16x8*)(w + ((size_t)((ks / 4) * 256 + crow)) * 256 + wave * 64 + ((ks % 4) * 2 + hi) * 8);
    }
    if (lane < 32) {
        const int i = lane & 15, h = lane >> 4;
        bias_lds[(wave * 2 + h) * 32 + i] = i < 8 ? bias[cg * 64 + wave * 16 + 8 * (i >> 2) + 4 * h + (i & 3)] : 0.0f;
        bias_lds[(wave * 2 + h) * 32 + 16 + i] = 0.0f;
    }
    const cv_f32x16* bias_ptr = (const cv_f32x16*)(bias_lds + (wave * 2 + hi) * 32);
    const unsigned lo16 = relu ? 0u : 0x80008000u;

    unsigned dsrc[C1_NBAND];
    unsigned long long dmask[C1_NBAND];
#pragma unroll
    for (int j = 0; j < C1_NBAND; ++j) {
        const int dcell = j * 64 + lane, dk = dcell - 1, drs = dk / C9_PITCH, dxx = dk - drs * C9_PITCH, drow = r0 - 1 + drs;
        const bool dok = dcell >= 1 && dcell < C9_CELLS - 1 && dxx < C9_S && drow >= 0 && drow < C9_S;
        dsrc[j] = dok ? (unsigned)((drow * C9_S + dxx) * 16) : 0u;
        dmask[j] = __builtin_amdgcn_ballot_w64(dok);
    }
    auto dma_piece = [&](const unsigned char* src, bool live, int j, int i) {
        const int c = wave + 4 * i;
        const unsigned long long base = (unsigned long long)(src + (size_t)c * C9_GBLK);
        const unsigned long long mask = live ? dmask[j] : 0ull;
        const unsigned dst = lds0 + (unsigned)(c * C9_LBLK + j * 1024);
        asm volatile("s_mov_b64 exec, %0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1"
                     :
                     : "s"(mask), "s"(dst), "v"(dsrc[j]), "s"(base)
                     : "memory");
    };
    unsigned lmap[C9_NCT];
#pragma unroll
    for (int ct = 0; ct < C9_NCT; ++ct) {
        const unsigned tp = c1_maps[hf].pos[ct * 32 + l31];
        const unsigned gp = tp != 0xffffu ? tp + (unsigned)(r0 * C9_S) : 0xffffu;
        lmap[ct] = (unsigned)((c1_maps[hf].cell[ct * 32 + l31] - C9_CELL0) * 16 + hi * C9_LBLK) | (gp << 16);
    }
    unsigned long long smask[C9_NCT];
#pragma unroll
    for (int ct = 0; ct < C9_NCT; ++ct) smask[ct] = __builtin_amdgcn_ballot_w64((lmap[ct] >> 16) != 0xffffu);
    auto store8 = [&](unsigned char* base, unsigned voff, unsigned a, unsigned b2, unsigned long long mask) {
        const cv_u32x2 d = (cv_u32x2){a, b2};
        asm volatile("s_mov_b64 exec, %0\n\tglobal_store_dwordx2 %1, %2, %3\n\ts_mov_b64 exec, -1" : : "s"(mask), "v"(voff), "v"(d), "s"(base) : "memory");
    };
    const unsigned char* Xw = lds + wave * (8 * C9_LBLK);  // this wave's cin quarter of the image: chunks 8 wave .. 8 wave + 7
    cv_bf16x8 bb[R];
    auto load_step = [&](const unsigned char* p0, int ks, int slot) {
        const int tap = ks / 4, kk = ks % 4;
        bb[slot] = *(const cv_bf16x8*)(p0 + ((tap / 3) * C9_PITCH + (tap % 3)) * 16 + kk * (2 * C9_LBLK));
    };
    unsigned char* const xb = lds + C1_IMG + lane * 16;

    if (s < nboards) {
        const unsigned char* src = x + (size_t)s * OTILE;
#pragma unroll
        for (int j = 0; j < C1_NBAND; ++j)
#pragma unroll
            for (int i = 0; i < SC::NPIECE; ++i) dma_piece(src, true, j, i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CV_BARRIER();
#pragma unroll
    for (int st = 0; st < R - 1; ++st) load_step(Xw + (lmap[0] & 0xffffu), st, st);
#pragma unroll
    for (int t = 0; t < NSTEP; ++t) {
        if (t < 64) asm volatile("" : : "a"(wf[t]));
        else asm volatile("" : : "v"(wf[t]));
    }

    typedef __attribute__((ext_vector_type(4))) float f32x4;
    cv_f32x16 acc[2][2];  // [accumulator set = unit parity][tile index]
    cv_u32x2 rr[2];
    f32x4 xr[2];
    // A unit's accumulators start from the bias THROUGH THE C OPERAND of its first two MFMAs (tile 0: the bias registers below; tile 1 holds
    // partner quads only: the inline constant 0) -- rounds 2-6 re-read 2 x 16 registers of bias from LDS per unit and wave (8 ds_read_b128 = 14 %
    // of the kernel's LDS read traffic, issued as one burst)
    const cv_f32x16 biasv = bias_ptr[0];
    acc[0][0] = acc[1][0] = biasv;
    acc[0][1] = acc[1][1] = bias_ptr[1];
    rr[0] = rr[1] = (cv_u32x2){0u, 0u};
    xr[0] = xr[1] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    float ev[4];
    unsigned epa = 0, epb = 0;
    // hand-over write o (0..5) of set `set`: destination d = wave ^ (o / 2 + 1), its source slot (d ^ wave) - 1 = o / 2; quad j = o % 2.
    // to wave ^ 1: tile 0 registers 8-15; to wave ^ 2: tile 1 registers 0-7; to wave ^ 3: tile 1 registers 8-15
    auto xwrite = [&](int set, int o) {
        const int sidx = o >> 1, j = o & 1, d = wave ^ (sidx + 1);
        const int tt = sidx == 0 ? 0 : 1, r8 = (sidx == 1 ? 0 : 8) + 4 * j;
        *(f32x4*)(xb + d * 6144 + sidx * 2048 + j * 1024) = (f32x4){acc[set][tt][r8 + 0], acc[set][tt][r8 + 1], acc[set][tt][r8 + 2], acc[set][tt][r8 + 3]};
    };
    auto rider = [&](int set, int ct, unsigned char* out, int o, bool store_ok) {
        if (o < 30) {
            const int sidx = o / 10, k = o % 10;
            if (k < 2) {
                xr[k] = *(const f32x4*)(xb + wave * 6144 + sidx * 2048 + k * 1024);
            } else {
                const int e = k - 2;
                acc[set][0][e] = cw_add_f32(acc[set][0][e], xr[e >> 2][e & 3]);
            }
            return;
        }
        const int q = (o - 30) / SC::OPS, op = (o - 30) % SC::OPS;
        if (ADD) {
            if (op == 0) ev[0] = cw_add_f32(acc[set][0][q * 4 + 0], cv_bf16_lo(rr[q].x));
            else if (op == 1) ev[1] = cw_add_f32(acc[set][0][q * 4 + 1], cv_bf16_hi(rr[q].x));
            else if (op == 2) ev[2] = cw_add_f32(acc[set][0][q * 4 + 2], cv_bf16_lo(rr[q].y));
            else if (op == 3) ev[3] = cw_add_f32(acc[set][0][q * 4 + 3], cv_bf16_hi(rr[q].y));
            else if (op == 4) epa = cw_pk_bf16(ev[0], ev[1]);
            else if (op == 5) epb = cw_pk_bf16(ev[2], ev[3]);
        } else {
            if (op == 0) epa = cw_pk_bf16(acc[set][0][q * 4 + 0], acc[set][0][q * 4 + 1]);
            else if (op == 1) epb = cw_pk_bf16(acc[set][0][q * 4 + 2], acc[set][0][q * 4 + 3]);
        }
        if (op == SC::OPS - 3) epa = cw_pk_max_i16(epa, lo16);
        else if (op == SC::OPS - 2) epb = cw_pk_max_i16(epb, lo16);
        else if (op == SC::OPS - 1) {
            unsigned lm = lmap[ct];
            asm volatile("" : "+v"(lm));
            store8(out, (unsigned)(q * C9_GBLK) + (lm >> 16) * 16u + (unsigned)(hi * 8), epa, epb, store_ok ? smask[ct] : 0ull);
        }
    };

    int it = 0;
    unsigned char* yprev = y;
    for (int board = s; board < nboards; board += nst, ++it) {
        const bool has_next = board + nst < nboards, have_prev = it > 0;
        const unsigned char* csrc = x + (size_t)board * OTILE;
        const unsigned char* nsrc = x + (size_t)(has_next ? board + nst : board) * OTILE;
        const size_t obase = (size_t)board * OTILE + (size_t)(cg * 8 + wave * 2) * C9_GBLK;  // the own 16 couts = chunks 2 wave, 2 wave + 1 of the CU's eight
        const unsigned char* abase = ADD ? add + obase : nullptr;
        unsigned char* ybase = y + obase;
        cp_for_each([&](auto UC) __attribute__((always_inline)) {
            constexpr int u = decltype(UC)::value, set = u & 1, pset = set ^ 1;
            constexpr int pct = (u + NU - 1) % NU;
            unsigned char* pout = u == 0 ? yprev : ybase;
            const bool pstore = u > 0 || have_prev;
            const unsigned char* b0 = Xw + (lmap[u] & 0xffffu);
            const unsigned char* nb0 = Xw + (lmap[(u + 1) % NU] & 0xffffu);
            cp_for_each([&](auto TC) __attribute__((always_inline)) {
                constexpr int t = decltype(TC)::value, ks = t >> 1, tt = t & 1;
#ifndef C1Q_ABLATE_NO_F
                if constexpr (t == SC::S0) CV_BARRIER();  // F: every wave is past the previous unit and its reads of the previous hand-over
#endif
                if constexpr (t == SB) {
                    if constexpr (u == 1 || u == 3 || u == 5) {
                        constexpr int N = u == 1 ? SC::vm_between(0, 1) : u == 3 ? SC::vm_between(1, 3) : SC::vm_between(4, 5);
                        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
                    }
                    CV_BARRIER();  // B: partials written, band free / landed (k_conv3x3_op19's barrier)
                }
                if constexpr (tt == 0) {  // one fragment per k-step, R - 1 k-steps ahead
                    if constexpr (ks + R - 1 < NKS) load_step(b0, ks + R - 1, (u * NKS + ks + R - 1) % R);
                    else load_step(nb0, ks + R - 1 - NKS, (u * NKS + ks + R - 1) % R);
                }
                static_assert(NKS < 64, "the first fragments of both tiles are AGPR operands");
                if constexpr (t == 0) cw_mfma_ac(acc[set][0], wf[0], bb[(u * NKS) % R], biasv);
                else if constexpr (t == 1) c1q_mfma_a0(acc[set][1], wf[NKS], bb[(u * NKS) % R]);
                else if constexpr (tt * NKS + ks < 64) cw_mfma_a(acc[set][tt], wf[tt * NKS + ks], bb[(u * NKS + ks) % R]);
                else cw_mfma_v(acc[set][tt], wf[tt * NKS + ks], bb[(u * NKS + ks) % R]);
                if constexpr (t >= SC::S0 && t < SC::S0 + 6) xwrite(pset, t - SC::S0);
                if constexpr (ADD) {
                    if constexpr (t == SC::addend_use_slot(0)) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rr[0]) : "n"(SC::vm_after_addend(u, 0)));
                    if constexpr (t == SC::addend_use_slot(1)) asm volatile("s_waitcnt vmcnt(%1)" : "+v"(rr[1]) : "n"(SC::vm_after_addend(u, 1)));
                }
                if constexpr (t > SB && t - SB - 1 < SC::NRID) rider(pset, pct, pout, t - SB - 1, pstore);
                if constexpr (ADD) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        if (t == SC::store_slot(q)) {
                            unsigned lm = lmap[u];
                            asm volatile("" : "+v"(lm));
                            const unsigned gp = lm >> 16;
                            const unsigned voff = (unsigned)(q * C9_GBLK) + (gp == 0xffffu ? 0u : gp) * 16u + (unsigned)(hi * 8);
                            asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(rr[q]) : "v"(voff), "s"(abase) : "memory");
                        }
                }
                if constexpr (SC::dma_unit(u) && t >= SC::dma_slot(0) && t <= SC::dma_slot(SC::NPIECE - 1) && (t - SC::dma_slot(0)) % 2 == 0) {
                    constexpr int i = (t - SC::dma_slot(0)) / 2;
                    if constexpr (u == 0) dma_piece(csrc, have_prev, 2, i);
                    else if constexpr (u == 1) dma_piece(csrc, have_prev, 3, i);
                    else if constexpr (u == 2) dma_piece(nsrc, has_next, 0, i);
                    else dma_piece(nsrc, has_next, 1, i);
                }
                __builtin_amdgcn_sched_barrier(0);
            }, typename CpMakeSeq<NSTEP>::type{});
        }, typename CpMakeSeq<NU>::type{});
        yprev = ybase;
    }
    if (it > 0) {
        constexpr int lset = (NU - 1) & 1;
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[lset][0]), "+v"(acc[lset][1]));
        CV_BARRIER();
#pragma unroll
        for (int o = 0; o < 6; ++o) xwrite(lset, o);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rr[0]), "+v"(rr[1]));
        CV_BARRIER();
#pragma unroll
        for (int o = 0; o < SC::NRID; ++o) rider(lset, NU - 1, yprev, o, true);
    }
}
#endif  // __HIPCC__
