// PROBE COPY (tools/probes): the round-2 kernel source with its timing-ablation switches (CP_ABL_*), used by conv_pipe_probe.hip /
// conv19_probe.hip only.  The product headers under alpha_zero_amd/csrc carry no ablation code.
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
//     only), 190 positions = 6 column tiles of 32 = 2 units of 3; v_mfma_f32_32x32x16_bf16, 72 k-steps per column tile.
//   * LDS image per 8-channel chunk: 242 cells of 16 B, cell(rs, x) = 1 + 20 rs + x for the 12 input rows rs (r0 - 1 .. r0 + 10)
//     with one zero cell between rows; rows outside the board are never written (zero).  A tap (dy, dx) is the constant cell offset
//     20 dy + dx.  Double buffered (2 x 61,952 B), filled by LDS-DMA: wave q moves cells [64 q, 64 q + 64) of every chunk strip.
//     A CU only ever sees tiles of one half (its zero rows are static): tile streams are per (cout half, board half).
//   * (column tile, lane) -> position from a residue-class table as in az_conv.h; six of the sixteen classes have 13 members for
//     12 lane groups, so 6 of the 190 positions sit in a group that already holds their residue (a 2-way conflict on 6 of 192
//     lanes); unused slots repeat a cell and are never stored.
//   * epilogue straight from the accumulators, 8-byte slots, bias as the C operand of the first MFMA.
#pragma once
#include "az_conv_abl.h"

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
};
constexpr C9Map c9_make_map() {
    C9Map m{};
    const int lanes[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    const int NG = 2 * C9_NCT;
    int cnt[16] = {}, fill[NG] = {};
    bool used[NG][16] = {};
    int deferred[32] = {}, ndef = 0;
    for (int i = 0; i < C9_NCT * 32; ++i) {
        m.cell[i] = 0;
        m.pos[i] = 0xffff;
    }
    for (int p = 0; p < C9_NPOS; ++p) {
        const int cell = C9_CELL0 + C9_PITCH * (p / C9_S) + p % C9_S, r = cell & 15, k = cnt[r]++;
        if (k >= NG) {
            deferred[ndef++] = p;
            continue;
        }
        const int idx = (k >> 1) * 32 + lanes[k & 1][fill[k]++];
        m.cell[idx] = (unsigned short)cell;
        m.pos[idx] = (unsigned short)p;
        used[k][r] = true;
    }
    for (int d = 0; d < ndef; ++d) {  // classes with more members than lane groups: into the emptiest group (one 2-way conflict each)
        const int p = deferred[d], cell = C9_CELL0 + C9_PITCH * (p / C9_S) + p % C9_S;
        int best = 0;
        for (int k = 1; k < NG; ++k)
            if (fill[k] < fill[best]) best = k;
        const int idx = (best >> 1) * 32 + lanes[best & 1][fill[best]++];
        m.cell[idx] = (unsigned short)cell;
        m.pos[idx] = (unsigned short)p;
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
    return m;
}
static __device__ const C9Map c9_map = c9_make_map();

// ADD: an addend tensor (the residual in launch A, the partial sum in launch B; may alias y).  NCH = input chunks contracted by this
// launch: 16 (one half of the tower's 256 channels) or 4 (the stem: 17 planes padded to 32).  cin_total = row length of w_packed
// [9 taps][256 couts][cin_total], cin_off = first input channel of this launch, x_chunks = chunks per board of x, x_chunk0 = first
// chunk read.  The output always has 32 chunks (256 couts).
template <bool ADD, int NCH> __global__ void __launch_bounds__(CW_THREADS, 1)
k_conv3x3_hb19(const unsigned char* __restrict__ x, const unsigned short* __restrict__ w, const float* __restrict__ bias, const unsigned char* add,
               unsigned char* y, int nboards, int relu, int add_bias, int cin_total, int cin_off, int x_chunks, int x_chunk0) {
    constexpr int KS = NCH / 2, NSTEP = 9 * KS;
    constexpr int LBUF = NCH * C9_LBLK;
    constexpr int NPIECE = NCH;  // DMA pieces per wave per tile: its 64-cell quarter of every chunk strip
    constexpr int OTILE = 32 * C9_GBLK;  // output / addend board: 256 channels
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
    cv_f32x16 bv;
#pragma unroll
    for (int rq = 0; rq < 4; ++rq)
#pragma unroll
        for (int e = 0; e < 4; ++e) bv[rq * 4 + e] = add_bias ? bias[cout0 + 8 * rq + 4 * hi + e] : 0.0f;
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
    // board position (high 16 bits; 0xffff = computed but not stored: repeated slots, and row 9 in the second board half)
    unsigned lmap[C9_NCT];
#pragma unroll
    for (int ct = 0; ct < C9_NCT; ++ct) {
        const unsigned tp = c9_map.pos[ct * 32 + l31];
        unsigned gp = 0xffffu;
        if (tp != 0xffffu && !(hf == 1 && tp < (unsigned)C9_S)) gp = tp + (unsigned)(r0 * C9_S);
        lmap[ct] = (unsigned)((c9_map.cell[ct * 32 + l31] - C9_CELL0) * 16 + hi * C9_LBLK) | (gp << 16);
    }

    cv_bf16x8 bb[4][3];  // ring of B fragments: k-step s lives in slot s & 3
    auto load_step = [&](const unsigned char* const (&bp)[3], int st) {
        const int tap = st / KS, ks = st % KS;
        const int off = ((tap / 3) * C9_PITCH + (tap % 3)) * 16 + ks * (2 * C9_LBLK);
#pragma unroll
        for (int j = 0; j < 3; ++j) bb[st & 3][j] = *(const cv_bf16x8*)(bp[j] + off);
    };

    if (s < nboards) {  // first tile: all pieces at once
        const unsigned char* src = x + (size_t)s * xboard + (size_t)x_chunk0 * C9_GBLK;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) dma_piece(src, lds0, true, i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CV_BARRIER();
    {
        const unsigned char* bp0[3] = {lds + (lmap[0] & 0xffffu), lds + (lmap[1] & 0xffffu), lds + (lmap[2] & 0xffffu)};
        load_step(bp0, 0);
        load_step(bp0, 1);
        load_step(bp0, 2);
    }
    int it = 0;
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
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const unsigned char* bp[3];
            cv_u32x2 rr[3][4];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const unsigned mj = lmap[u * 3 + j];
                bp[j] = Xs + (mj & 0xffffu);
                if (ADD) {
                    const unsigned gp = mj >> 16, gq = (gp == 0xffffu ? 0u : gp) * 16u + (unsigned)(hi * 8);
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) rr[j][rq] = *(const cv_u32x2*)(abase + rq * C9_GBLK + gq);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            cv_f32x16 acc[3];
#pragma unroll
            for (int t = 0; t < NSTEP; ++t) {  // fragments of steps 0..2 are already in flight
#ifndef CP_ABL_NO_FRAG  // (ablation switch of tools/probes/conv19_probe.hip, never defined in the product build)
                if (t + 3 < NSTEP) load_step(bp, t + 3);
#endif
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    if (t == 0) cw_mfma_ac(acc[j], wf[0], bb[0][j], bv);
                    else if (t < 64) cw_mfma_a(acc[j], wf[t], bb[t & 3][j]);
                    else cw_mfma_v(acc[j], wf[t], bb[t & 3][j]);
                }
                // the next tile's DMA pieces ride in the shadow of unit 0's MFMAs (its buffer was released by the previous barrier)
#ifndef CP_ABL_NO_DMA
                if (u == 0 && t % 3 == 1 && t / 3 < NPIECE) dma_piece(nsrc, ndst, has_next, t / 3);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            if (u == 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                CV_BARRIER();
            }
            {   // the next unit's first fragments fly while the epilogue below runs
                const unsigned char* bpn[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) bpn[j] = (u == 0 ? Xs : Xn) + (lmap[(1 - u) * 3 + j] & 0xffffu);
                load_step(bpn, 0);
                load_step(bpn, 1);
                load_step(bpn, 2);
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const unsigned gp = lmap[u * 3 + j] >> 16;
                const unsigned gq = (gp == 0xffffu ? 0u : gp) * 16u + (unsigned)(hi * 8);
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    float v0 = acc[j][rq * 4 + 0], v1 = acc[j][rq * 4 + 1], v2 = acc[j][rq * 4 + 2], v3 = acc[j][rq * 4 + 3];
                    if (ADD) {
                        const cv_u32x2 r2 = rr[j][rq];
                        v0 += cv_bf16_lo(r2.x);
                        v1 += cv_bf16_hi(r2.x);
                        v2 += cv_bf16_lo(r2.y);
                        v3 += cv_bf16_hi(r2.y);
                    }
                    const cv_u32x2 o = (cv_u32x2){cw_pk_max_i16(cw_pk_bf16(v0, v1), lo16), cw_pk_max_i16(cw_pk_bf16(v2, v3), lo16)};
                    if (gp != 0xffffu) *(cv_u32x2*)(ybase + rq * C9_GBLK + gq) = o;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
#endif  // __HIPCC__
