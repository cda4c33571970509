// az_conv.h -- fused 3x3 convolution of the leaf evaluator's residual tower for gfx950:
//     y = relu(conv3x3(x, w) + bias [+ residual])      x, y, residual: [boards][9][9][128] bf16 (channels-last)
// (reference: alpha_zero/core/network.py:42-82 ResNetBlock in eval mode, BatchNorm folded into w / bias).
//
// Implicit GEMM on the matrix cores, D[cout][position] = sum over 9 taps x 128 cin:
//   * v_mfma_f32_32x32x16_bf16; A operand = weights W[tap][cout][cin], B operand = activations X[position][cin]
//     (both fragments are 8 consecutive cin = one 16-B LDS read per lane).
//   * one persistent 512-thread workgroup per CU, tile = 3 boards = 243 positions (padded to 256 columns);
//     8 waves = 2 cout halves x 4 column quarters, each 2x2 MFMA tiles (64 accumulator registers), 2 waves per SIMD.
//   * the 3 boards sit in LDS once (row = position, plus one all-zero row that off-board taps read), so the 9 taps
//     are 9 shifted reads of the same tile: HBM sees every activation exactly once per convolution; weights stream
//     per tap through a 2 x 32 KB LDS ring.
//   * all LDS rows are 256 B; the 16-B chunk index is XOR-ed with (row & 15).  A 16-lane ds_read_b128 group always
//     holds 16 rows that are distinct mod 16 (positions shifted by a per-tap constant), i.e. 16 distinct chunks =
//     all 64 banks exactly once: conflict-free for every tap (measured before this layout: 40 % of LDS cycles were
//     bank-conflict cycles with 11x11 halo-padded rows).
//   * epilogue fused: bias + residual + ReLU + one bf16 rounding straight on the accumulators (the D layout gives each
//     lane 4 consecutive couts of its position = 8-byte slots of the channels-last row); no LDS round trip.
#pragma once
#include <stdint.h>

#define CV_C 128
#define CV_S 9
#define CV_TB 3
// boards per tile of the tiled activation layout: as many whole boards as fit 256 MFMA columns (3 at 9x9, 1 from 13x13 up)
static inline int cv_tile_boards(int S) { return 256 / (S * S) > 0 ? 256 / (S * S) : 1; }
#define CV_P2 (CV_S * CV_S)                   // 81
#define CV_ROWB (CV_C * 2)                    // 256 bytes per position / per cout row
#define CV_NPOS (CV_TB * CV_P2)               // 243
#define CV_ZROW CV_NPOS                        // LDS row 243 is all zeros: what an off-board tap reads
#define CV_XS_BYTES ((CV_NPOS + 1) * CV_ROWB)  // 62,464: rows = the tile's positions (no halo padding) + the zero row
#define CV_WBUF (CV_C * CV_ROWB)              // 32,768 per tap
#define CV_WS_BYTES (2 * CV_WBUF)

#if defined(__HIPCC__)
typedef __attribute__((ext_vector_type(8))) __bf16 cv_bf16x8;
typedef __attribute__((ext_vector_type(16))) float cv_f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int cv_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int cv_u32x2;

__device__ __forceinline__ unsigned cv_swz(unsigned row, unsigned chunk) { return row * CV_ROWB + ((chunk ^ (row & 15u)) << 4); }
__device__ __forceinline__ unsigned cv_pack_bf16(float a, float b) {  // round to nearest even
    unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    ua = (ua + 0x7fffu + ((ua >> 16) & 1u)) >> 16;
    ub = (ub + 0x7fffu + ((ub >> 16) & 1u)) >> 16;
    return ua | (ub << 16);
}
__device__ __forceinline__ float cv_bf16_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float cv_bf16_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }

// Workgroup barrier that waits for this wave's LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it would wait
// for the global prefetches (next tap's weights, next tile's activations, the residual) that were issued precisely so
// that they stay in flight under the MFMAs; the compiler still inserts the counted vmcnt wait at their first use.
#define CV_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
// ---------------------------------------------------------------------------------------------------------------------
// Weight-stationary kernel on the TILED activation layout (the layout the residual tower keeps between its layers):
//     [tile = 3 boards][16 cin-chunks][243 positions][8 channels] bf16          (62,208 B per tile, CT_* below)
// Measured on MI355X (tools/probes/lds_probe.hip): with ONE wave per SIMD an MFMA only overlaps with that wave's LDS
// reads, not with its VALU instructions (every VALU op costs ~4.5 cycles of matrix-core time), so the k-loop below
// contains no VALU instruction at all and the epilogue is kept to ~0.5 VALU per MFMA.
//   * 4 waves per CU (one per SIMD); wave q owns couts [32q, 32q+32) and keeps its A fragments for all 9 taps x 8 k-steps
//     in registers for the lifetime of the persistent workgroup: 64 fragments in AGPRs (read by the MFMA directly), 8 in
//     VGPRs.  No weight traffic, no weight staging, no per-tap barriers.
//   * LDS image of a tile: per cin-chunk a strip of 313 16-byte cells: 11 zero cells, then per board row 9 positions + 1
//     zero cell, + 11 zero cells per board, i.e. cell(b, y, x) = 11 + 101 b + 10 y + x.  A tap (dy, dx) is the constant cell
//     offset 10 dy + dx (off-board neighbours ARE zero cells), so every B-fragment address is one per-lane base + an
//     immediate.  Zero cells are written once; LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, masked to the
//     position cells) copies the compact global tile into the double buffer while the previous tile is multiplied.
//   * which position a (column tile, lane) computes is a table (cw_map): the two 16-lane groups a ds_read_b128 is
//     serviced in each hold cells that are distinct mod 16 = all 64 banks once, for every tap (a tap shifts all cells of
//     a group alike).  The board pitch 101 makes that possible: no residue class has more than 16 of the 243 cells.
//   * a tile's 8 column tiles run as 2 units of 4 (128 positions, 288 MFMAs); bias enters as the C operand of the first
//     MFMA; epilogue = residual add, one bf16 rounding, ReLU on the packed result, 512 contiguous bytes per store.
#define CT_ROWS CV_NPOS                     // 243 positions per tile
#define CT_GBLK (CT_ROWS * 16)              // 3,888 B: one cin-chunk block of a tile in global memory
#define CT_TILE (16 * CT_GBLK)              // 62,208 B per tile
#define CT_CELL0 11                         // first position cell of board 0
#define CT_BPITCH 101                       // cells from one board to the next
#define CT_CELLS (CT_CELL0 + 2 * CT_BPITCH + 100)  // 313 cells per chunk strip
#define CT_LBLK (CT_CELLS * 16)             // 5,008 B: chunk strip in LDS
#define CT_LBUF (16 * CT_LBLK)              // 80,128 B per buffer

// (column tile, lane & 31) -> position / LDS cell.  Group k (k = 2 * column tile + {0: lanes 0-3,12-15,20-27; 1: the others})
// takes the k-th position cell of every residue class mod 16, so a group spans ~17 consecutive cells.
struct CwMap {
    unsigned short cell[256], pos[256];
};
constexpr CwMap cw_make_map() {
    CwMap m{};
    const int lanes[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    int cnt[16] = {}, fill[16] = {};
    bool used[16][16] = {};
    for (int i = 0; i < 256; ++i) {
        m.cell[i] = 0;
        m.pos[i] = 0xffff;
    }
    for (int p = 0; p < CV_NPOS; ++p) {
        const int b = p / CV_P2, q = p % CV_P2, cell = CT_CELL0 + CT_BPITCH * b + 10 * (q / CV_S) + q % CV_S;
        const int r = cell & 15, k = cnt[r]++, slot = fill[k]++;
        const int idx = (k >> 1) * 32 + lanes[k & 1][slot];
        m.cell[idx] = (unsigned short)cell;
        m.pos[idx] = (unsigned short)p;
        used[k][r] = true;
    }
    // unused slots (13 of 256, all in the last groups): a second copy of a real position whose residue the group lacks; it
    // computes and stores the same value as the primary copy, so no lane of the kernel is ever masked
    for (int k = 0; k < 16; ++k)
        for (int r = 0; r < 16 && fill[k] < 16; ++r) {
            if (used[k][r]) continue;
            for (int p = 0; p < CV_NPOS; ++p) {
                const int b = p / CV_P2, q = p % CV_P2, cell = CT_CELL0 + CT_BPITCH * b + 10 * (q / CV_S) + q % CV_S;
                if ((cell & 15) == r) {
                    const int idx = (k >> 1) * 32 + lanes[k & 1][fill[k]++];
                    m.cell[idx] = (unsigned short)cell;
                    m.pos[idx] = (unsigned short)p;
                    break;
                }
            }
        }
    return m;
}
static __device__ const CwMap cw_map = cw_make_map();
#define CW_THREADS 256

__device__ __forceinline__ void cw_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst))
                 : "memory");
}
// MFMA with the A operand (weights) in an AGPR / a VGPR; inline asm so that the 256 weight AGPRs are read in place
// instead of being copied to VGPRs before every use.
__device__ __forceinline__ void cw_mfma_a(cv_f32x16& acc, const cv_bf16x8& wa, const cv_bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(wa), "v"(b));
}
__device__ __forceinline__ void cw_mfma_v(cv_f32x16& acc, const cv_bf16x8& wa, const cv_bf16x8& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(b));
}
__device__ __forceinline__ void cw_mfma_ac(cv_f32x16& acc, const cv_bf16x8& wa, const cv_bf16x8& b, const cv_f32x16& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "a"(wa), "v"(b), "v"(c));
}
__device__ __forceinline__ unsigned cw_pk_bf16(float a, float b) {  // one v_cvt_pk_bf16_f32 (round to nearest even)
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    const b2 r = __builtin_convertvector((f2){a, b}, b2);
    return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ float cw_add_f32(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ unsigned cw_pk_max_i16(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// The evaluator kernels exist in two activation / weight formats with the same MFMA rate: bf16 (the default) and f16 (three more
// significand bits, activations of the BatchNorm-folded networks are far inside its range).  CvFmt<F16> is everything that differs:
// the MFMA mnemonic, the rounding of two accumulators into one dword, the widening of a packed pair.  The packed-integer ReLU
// (v_pk_max_i16 against 0: negative floats are negative integers) and every layout are format-independent.
typedef __attribute__((ext_vector_type(8))) _Float16 cv_f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 cv_f16x2;
template <bool F16> struct CvFmt;
template <> struct CvFmt<false> {
    static __device__ __forceinline__ void mfma_a(cv_f32x16& acc, const cv_bf16x8& wa, const cv_bf16x8& b) { cw_mfma_a(acc, wa, b); }
    static __device__ __forceinline__ void mfma_v(cv_f32x16& acc, const cv_bf16x8& wa, const cv_bf16x8& b) { cw_mfma_v(acc, wa, b); }
    static __device__ __forceinline__ unsigned pk(float a, float b) { return cw_pk_bf16(a, b); }
    static __device__ __forceinline__ float lo(unsigned v) { return cv_bf16_lo(v); }
    static __device__ __forceinline__ float hi(unsigned v) { return cv_bf16_hi(v); }
    static __device__ __forceinline__ unsigned short one(float v) { return (unsigned short)(cv_pack_bf16(v, 0.0f) & 0xffffu); }
    static __device__ __forceinline__ cv_f32x16 mfma32(const cv_bf16x8& a, const cv_bf16x8& b, const cv_f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct CvFmt<true> {
    static __device__ __forceinline__ void mfma_a(cv_f32x16& acc, const cv_bf16x8& wa, const cv_bf16x8& b) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(wa), "v"(b));
    }
    static __device__ __forceinline__ void mfma_v(cv_f32x16& acc, const cv_bf16x8& wa, const cv_bf16x8& b) {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(b));
    }
    static __device__ __forceinline__ unsigned pk(float a, float b) {
        // one v_cvt_pk_f16_f32 (round to nearest even); + infinity (0x7C00) is clamped to the largest finite value by a packed integer min
        unsigned r;
        asm("v_pk_min_i16 %0, %1, %2" : "=v"(r) : "v"(__builtin_bit_cast(unsigned, (cv_f16x2){(_Float16)a, (_Float16)b})), "v"(0x7BFF7BFFu));
        return r;
    }
    static __device__ __forceinline__ float lo(unsigned v) { return (float)__builtin_bit_cast(cv_f16x2, v)[0]; }
    static __device__ __forceinline__ float hi(unsigned v) { return (float)__builtin_bit_cast(cv_f16x2, v)[1]; }
    static __device__ __forceinline__ unsigned short one(float v) { return (unsigned short)(pk(v, 0.0f) & 0xffffu); }
    static __device__ __forceinline__ cv_f32x16 mfma32(const cv_bf16x8& a, const cv_bf16x8& b, const cv_f32x16& c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(cv_f16x8, a), __builtin_bit_cast(cv_f16x8, b), c, 0, 0, 0);
    }
};

// ---------------------------------------------------------------------------------------------------------------------
// k_conv3x3_tiled<RES, NCH> (NCH = input-channel chunks of 8: 16 for the tower 128 -> 128, 4 for the stem 17 planes padded to 32 -> 128):
// the weight-stationary convolution described above with the EPILOGUE SOFTWARE-PIPELINED into the MFMA stream.
// Measured (tools/probes/mfma_valu_probe.hip, profiles/r02_mfma_valu_probe.txt): with one wave per SIMD up to 4 plain VALU
// instructions placed in EVERY gap between two MFMAs are free (32.6-32.8 cycles per MFMA with 0, 2 or 4 of them; only packed-f32
// VALU is not: v_pk_add_f32 costs +11 cycles each), whereas the previous kernel ran the epilogue of a unit as one serial block
// after its MFMAs (~10 % of the launch).  Here a tile is 4 units of 2 column tiles with two accumulator sets: while unit u
// multiplies into set u & 1, the residual add / bf16 rounding / ReLU / stores of unit u - 1 are issued from the other set a few
// instructions per k-step, the B-fragment ring runs on across unit and tile boundaries, the next tile's LDS-DMA pieces ride in units
// 0-2, and the only synchronisation per tile is one barrier (3 k-steps before the end of unit 3, when every read of the
// current buffer has been issued) behind an exactly counted s_waitcnt vmcnt(N) that leaves the younger stores in flight.
#define CP_RING 4  // slots of the B-fragment ring: a k-step's fragments are requested CP_RING - 1 k-steps ahead (6 measured the same:
                  // the stream is not waiting on fragment latency, profiles/r02_conv_ablation.txt)
template <int V> struct CpInt {
    static constexpr int value = V;
};
template <int... I> struct CpSeq {};
template <int N, int... I> struct CpMakeSeq : CpMakeSeq<N - 1, N - 1, I...> {};
template <int... I> struct CpMakeSeq<0, I...> {
    typedef CpSeq<I...> type;
};
// f(CpInt<0>{}), f(CpInt<1>{}), ...: compile-time k-step index (no reliance on the loop unroller's size heuristics)
template <class F, int... I> __device__ __forceinline__ void cp_for_each(F&& f, CpSeq<I...>) { (f(CpInt<I>{}), ...); }
template <int NSTEP> struct CpSched {  // static schedule of one unit's k-steps (everything below is compile-time)
    static constexpr int E0 = 4;  // first k-step that may touch the previous unit's accumulators (>= 4 MFMAs behind their last write)
    // residual layers: slot s (column tile s / 4, register quad s % 4) takes 3 phases (2 x unpack + add, then round / ReLU / store)
    static constexpr int RSTRIDE = (NSTEP - E0 - 4) / 24 > 0 ? (NSTEP - E0 - 4) / 24 : 1;
    static constexpr int PSTRIDE = (NSTEP - E0 - 4) / 8 > 0 ? (NSTEP - E0 - 4) / 8 : 1;
    static constexpr int res_step(int s, int ph) { return E0 + (3 * s + ph) * RSTRIDE; }
    static constexpr int plain_step(int s) { return E0 + s * PSTRIDE; }
    static constexpr int store_step(bool res, int s) { return res ? res_step(s, 2) : plain_step(s); }
    // DMA piece i of the next tile: unit i % 3, k-step 3 + 4 (i / 3)  (odd steps: never a store step's neighbour in program order)
    static constexpr int dma_unit(int i) { return i % 3; }
    static constexpr int dma_step(int i) { return 3 + 4 * (i / 3); }
    // inverse maps (one candidate per k-step: keeps the loop body small enough for the unroller)
    static constexpr int dma_at(int u, int t, int npiece) {  // piece issued at (unit u, k-step t) or -1
        if (t < 3 || (t - 3) % 4 != 0 || u > 2) return -1;
        const int i = 3 * ((t - 3) / 4) + u;
        return i < npiece ? i : -1;
    }
    static constexpr int res_at(int t) {  // 3 * slot + phase handled at k-step t of a residual layer, or -1
        if (t < E0 || (t - E0) % RSTRIDE != 0) return -1;
        const int k = (t - E0) / RSTRIDE;
        return k < 24 ? k : -1;
    }
    static constexpr int plain_at(int t) {  // slot handled at k-step t of a plain layer, or -1
        if (t < E0 || (t - E0) % PSTRIDE != 0) return -1;
        const int k = (t - E0) / PSTRIDE;
        return k < 8 ? k : -1;
    }
    static constexpr int BAR_STEP = NSTEP - (CP_RING - 1);  // unit 3: barrier before the first fragment loads of the next tile
    static constexpr int RL0 = NSTEP >= 72 ? 40 : 0;  // first k-step of the unit's residual loads (late: the two sets' live ranges barely overlap)
    // VMEM operations issued after the last DMA piece and before the barrier: stores of unit 1's epilogue still to come in unit 2,
    // unit 3's residual loads (8, k-steps 0-3) and the stores of unit 2's epilogue (all before BAR_STEP)
    static constexpr int after_last_dma(bool res, int npiece) {
        int last_u = 0, last_t = -1;
        for (int i = 0; i < npiece; ++i)
            if (dma_unit(i) > last_u || (dma_unit(i) == last_u && dma_step(i) > last_t)) {
                last_u = dma_unit(i);
                last_t = dma_step(i);
            }
        int n = 0;
        for (int u = last_u; u < 4; ++u)
            for (int s = 0; s < 8; ++s) {
                const int st = store_step(res, s);
                if ((u > last_u || st >= last_t) && (u < 3 || st < BAR_STEP)) ++n;  // a store of the DMA's own k-step follows it
            }
        for (int u = last_u; u < 4; ++u)  // residual loads: 8 per unit in k-steps RL0 .. RL0 + 3
            for (int k = 0; k < 4; ++k)
                if (res && (u > last_u || RL0 + k > last_t) && (u < 3 || RL0 + k < BAR_STEP)) n += 2;
        return n;
    }
};

template <bool RES, int NCH, bool F16 = false> __global__ void __launch_bounds__(CW_THREADS, 1)
k_conv3x3_tiled(const unsigned char* __restrict__ x, const unsigned short* __restrict__ w, const float* __restrict__ bias,
                  const unsigned char* __restrict__ res, unsigned char* __restrict__ y, int ntiles, int relu) {
    constexpr int KS = NCH / 2, NSTEP = 9 * KS;
    constexpr int LBUF = NCH * CT_LBLK, XTILE = NCH * CT_GBLK, NPIECE = NCH + NCH / 4;
    typedef CpSched<NSTEP> SC;
    typedef CvFmt<F16> FM;
    static_assert((RES ? SC::res_step(7, 2) : SC::plain_step(7)) < SC::BAR_STEP, "epilogue must end before the barrier step");
    static_assert(SC::dma_step(NPIECE - 1) < NSTEP - 4, "DMA pieces must be issued early in their unit");
    constexpr int VM_AFTER_DMA = SC::after_last_dma(RES, NPIECE);
    static_assert(VM_AFTER_DMA < 63, "vmcnt field");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * LBUF];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    for (int i = tid; i < 2 * LBUF / 16; i += CW_THREADS) *(cv_u32x4*)(lds + i * 16) = (cv_u32x4){0u, 0u, 0u, 0u};
    // the zero cells must be final before ANY wave's LDS-DMA piece can land: the waves of a workgroup do not start in the same cycle
    // (seen under two concurrent streams: a late wave's zeroing wiped cells another wave's first pieces had already filled)
    CV_BARRIER();

    cv_bf16x8 wf[NSTEP];
#pragma unroll
    for (int t = 0; t < NSTEP; ++t)
        wf[t] = *(const cv_bf16x8*)(w + ((size_t)((t / KS) * CV_C + wave * 32 + l31)) * (8 * NCH) + ((t % KS) * 2 + hi) * 8);
    // bias in the accumulator layout, kept in LDS ([wave][lane half][16 floats], broadcast reads): a unit's accumulators are
    // initialised with it a few k-steps before the unit starts, which frees the 16 registers a C operand would pin
    __shared__ __attribute__((aligned(64))) float bias_lds[4 * 2 * 16];
    if (lane < 32) bias_lds[(wave * 2 + (lane >> 4)) * 16 + (lane & 15)] = bias[wave * 32 + 8 * ((lane & 15) >> 2) + 4 * (lane >> 4) + (lane & 3)];
    const cv_f32x16* bias_ptr = (const cv_f32x16*)(bias_lds + (wave * 2 + hi) * 16);
    const unsigned lo16 = relu ? 0u : 0x80008000u;

    auto cell_src = [&](int cell, bool& ok) -> unsigned {
        const int k = cell - CT_CELL0, b = k / CT_BPITCH, r = k - b * CT_BPITCH, yy = r / 10, xx = r - yy * 10;
        ok = k >= 0 && cell < CT_CELLS && b < CV_TB && yy < CV_S && xx < CV_S;
        return (unsigned)((b * CV_P2 + yy * CV_S + xx) * 16);
    };
    bool dok_a, dok_b;
    const unsigned dsrc_a = cell_src(wave * 64 + lane, dok_a), dsrc_b = cell_src(256 + lane, dok_b);
    const unsigned long long mask_a = __builtin_amdgcn_ballot_w64(dok_a), mask_b = __builtin_amdgcn_ballot_w64(dok_b);
    auto dma_piece = [&](const unsigned char* src, unsigned dstbuf, bool live, int i) {
        const int c = i < NCH ? i : (i - NCH) * 4 + wave;
        const unsigned long long base = (unsigned long long)(src + (size_t)c * CT_GBLK);
        const unsigned long long mask = live ? (i < NCH ? mask_a : mask_b) : 0ull;
        const unsigned dst = dstbuf + (unsigned)(c * CT_LBLK + (i < NCH ? wave : 4) * 1024);
        asm volatile("s_mov_b64 exec, %0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1"
                     :
                     : "s"(mask), "s"(dst), "v"(i < NCH ? dsrc_a : dsrc_b), "s"(base)
                     : "memory");
    };
    unsigned lmap[8];
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
        lmap[ct] = (unsigned)((cw_map.cell[ct * 32 + l31] - CT_CELL0) * 16 + hi * CT_LBLK) | ((unsigned)cw_map.pos[ct * 32 + l31] << 16);

    cv_bf16x8 bb[CP_RING][2];  // ring of B fragments: running k-step s lives in slot s % CP_RING
    // k-step s of a unit: tap s / KS = constant cell offset, cin chunks 2 (s % KS) + hi; `slot` = (running k-step count) & 3
    auto load_step = [&](const unsigned char* b0, const unsigned char* b1, int s, int slot) {
        const int tap = s / KS, ks = s % KS;
        const int off = ((tap / 3) * 10 + (tap % 3)) * 16 + ks * (2 * CT_LBLK);
        bb[slot][0] = *(const cv_bf16x8*)(b0 + off);
        bb[slot][1] = *(const cv_bf16x8*)(b1 + off);
    };
    static_assert((4 * NSTEP) % CP_RING == 0, "a tile's k-steps keep the ring phase");

    {   // first tile: all pieces at once, then the first fragments
        const unsigned char* src = x + (size_t)blockIdx.x * XTILE;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) dma_piece(src, lds0, true, i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CV_BARRIER();
#pragma unroll
        for (int i = 0; i < CP_RING - 1; ++i) load_step(lds + (lmap[0] & 0xffffu), lds + (lmap[1] & 0xffffu), i, i);
    }
    cv_f32x16 acc[2][2];   // [unit parity][column tile]
    cv_u32x2 rr[2][2][4];  // residual of the unit: [unit parity][column tile][register quad]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][j][e] = 0.0f;
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) rr[a][j][rq] = (cv_u32x2){0u, 0u};
        }
    acc[0][0] = *bias_ptr;  // (bias_lds was published by the barrier above)
    acc[0][1] = *bias_ptr;
    float ev[4];  // one epilogue slot between its phases

    // epilogue phase `ph` of slot s of the unit with parity q, column tiles mapped by lmap[mb + j]; out = that unit's output base
    auto epi = [&](int q, int mb, unsigned char* out, int s, int ph, bool store_ok) {
        const int j = s >> 2, rq = s & 3;
        if (RES) {
            const cv_u32x2 r2 = rr[q][j][rq];
            // plain v_add_f32: the compiler would SLP-pack two adds into v_pk_add_f32, which costs ~11 cycles of matrix-core time
            // beside MFMAs (profiles/r02_mfma_valu_probe.txt) where a scalar add is free
            if (ph == 0) {
                ev[0] = cw_add_f32(acc[q][j][rq * 4 + 0], FM::lo(r2.x));
                ev[1] = cw_add_f32(acc[q][j][rq * 4 + 1], FM::hi(r2.x));
            } else if (ph == 1) {
                ev[2] = cw_add_f32(acc[q][j][rq * 4 + 2], FM::lo(r2.y));
                ev[3] = cw_add_f32(acc[q][j][rq * 4 + 3], FM::hi(r2.y));
            }
        } else if (ph == 2) {
            ev[0] = acc[q][j][rq * 4 + 0], ev[1] = acc[q][j][rq * 4 + 1], ev[2] = acc[q][j][rq * 4 + 2], ev[3] = acc[q][j][rq * 4 + 3];
        }
        if (ph == 2) {
            const unsigned gq = (lmap[mb + j] >> 16) * 16u + (unsigned)(hi * 8);
            const cv_u32x2 o = (cv_u32x2){cw_pk_max_i16(FM::pk(ev[0], ev[1]), lo16), cw_pk_max_i16(FM::pk(ev[2], ev[3]), lo16)};
            if (store_ok) *(cv_u32x2*)(out + rq * CT_GBLK + gq) = o;
        }
    };

    int it = 0;
    unsigned char* yprev = y;  // output base of the previous tile (epilogue of its unit 3 runs inside this tile's unit 0)
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, ++it) {
        const int buf = it & 1;
        const unsigned char* Xs = lds + buf * LBUF;
        const unsigned char* Xn = lds + (buf ^ 1) * LBUF;
        const bool has_next = tile + (int)gridDim.x < ntiles;
        const unsigned char* nsrc = x + (size_t)(has_next ? tile + (int)gridDim.x : tile) * XTILE;
        const unsigned ndst = lds0 + (unsigned)((buf ^ 1) * LBUF);
        const unsigned char* rbase = RES ? res + (size_t)tile * CT_TILE + (size_t)(wave * 4) * CT_GBLK : nullptr;
        unsigned char* ybase = y + (size_t)tile * CT_TILE + (size_t)(wave * 4) * CT_GBLK;
        const bool have_prev = it > 0;
        // one unit = 2 column tiles x NSTEP k-steps; instantiated four times (a single 4 x NSTEP nest exceeds the unroller's budget)
        auto unit = [&](auto UC) {
            constexpr int u = decltype(UC)::value;
            constexpr int q = u & 1, pq = q ^ 1;                  // accumulator set of this unit / of the previous one
            constexpr int pmb = ((u + 3) & 3) * 2;                // lmap base of the previous unit's column tiles
            unsigned char* pout = u == 0 ? yprev : ybase;          // where the previous unit's outputs go
            const bool pstore = u > 0 || have_prev;
            const unsigned char* b0 = Xs + (lmap[u * 2] & 0xffffu);
            const unsigned char* b1 = Xs + (lmap[u * 2 + 1] & 0xffffu);
            const unsigned char* nb0 = (u < 3 ? Xs : Xn) + (lmap[((u + 1) & 3) * 2] & 0xffffu);
            const unsigned char* nb1 = (u < 3 ? Xs : Xn) + (lmap[((u + 1) & 3) * 2 + 1] & 0xffffu);
            cp_for_each([&](auto TC) __attribute__((always_inline)) {
                constexpr int t = decltype(TC)::value;
                if constexpr (u == 3 && t == SC::BAR_STEP) {
                    // every read of this buffer has been issued (the ring runs 3 k-steps ahead).  All DMA pieces of the next tile are
                    // older than the VM_AFTER_DMA youngest vector-memory operations of this wave, which may stay in flight.
                    if (have_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_AFTER_DMA) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // first tile: its unit-0 epilogue stores were skipped
                    CV_BARRIER();
                }
                if constexpr (t + CP_RING - 1 < NSTEP) load_step(b0, b1, t + CP_RING - 1, (u * NSTEP + t + CP_RING - 1) % CP_RING);
                else load_step(nb0, nb1, t + CP_RING - 1 - NSTEP, (u * NSTEP + t + CP_RING - 1) % CP_RING);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if constexpr (t < 64) FM::mfma_a(acc[q][j], wf[t], bb[(u * NSTEP + t) % CP_RING][j]);
                    else FM::mfma_v(acc[q][j], wf[t], bb[(u * NSTEP + t) % CP_RING][j]);
                }
                // ---- riders of this k-step -------------------------------------------------------------------------------------
                if constexpr (t == NSTEP - 6) acc[pq][0] = *bias_ptr;  // next unit's accumulators start from the bias
                if constexpr (t == NSTEP - 5) acc[pq][1] = *bias_ptr;
                if constexpr (RES && t >= SC::RL0 && t < SC::RL0 + 4) {  // this unit's residual, two 8-byte loads per k-step (used by the epilogue inside the next unit)
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int s = (t - SC::RL0) * 2 + k, j = s >> 2, rq = s & 3;
                        const unsigned gq = (lmap[u * 2 + j] >> 16) * 16u + (unsigned)(hi * 8);
                        rr[q][j][rq] = *(const cv_u32x2*)(rbase + rq * CT_GBLK + gq);
                    }
                }
                if constexpr (SC::dma_at(u, t, NPIECE) >= 0) dma_piece(nsrc, ndst, has_next, SC::dma_at(u, t, NPIECE));
                if constexpr (RES) {
                    if constexpr (SC::res_at(t) >= 0) epi(pq, pmb, pout, SC::res_at(t) / 3, SC::res_at(t) % 3, pstore);
                } else if constexpr (SC::plain_at(t) >= 0) {
                    epi(pq, pmb, pout, SC::plain_at(t), 2, pstore);
                }
                __builtin_amdgcn_sched_barrier(0);
            }, typename CpMakeSeq<NSTEP>::type{});
        };
        unit(CpInt<0>{});
        unit(CpInt<1>{});
        unit(CpInt<2>{});
        unit(CpInt<3>{});
        yprev = ybase;
    }
    // epilogue of the very last unit (accumulator set 1, column tiles 6 and 7)
    if (it > 0) {
        asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[1][0]), "+v"(acc[1][1]));
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (RES) {
                epi(1, 6, yprev, s, 0, true);
                epi(1, 6, yprev, s, 1, true);
            }
            epi(1, 6, yprev, s, 2, true);
        }
    }
}

// NHWC rows <-> tiled layout (tower entry / exit): one 16-B chunk per thread.
__global__ void k_tile_layout(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, long long nchunks, int to_tiled, int nch,
                              int tile_rows) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // index over (row, chunk) of the NHWC tensor
    if (i >= nchunks) return;
    const long long row = i / nch;
    const int c = (int)(i - row * nch);
    const long long tile = row / tile_rows;
    const int p = (int)(row - tile * tile_rows);
    const size_t t_off = ((size_t)(tile * nch + c) * tile_rows + (size_t)p) * 16, n_off = (size_t)i * 16;
    if (to_tiled) *(cv_u32x4*)(dst + t_off) = *(const cv_u32x4*)(src + n_off);
    else *(cv_u32x4*)(dst + n_off) = *(const cv_u32x4*)(src + t_off);
}

// 1x1 head convolution on the tiled layout (policy / value heads, core/network.py:131-156 conv + BatchNorm folded + ReLU):
// out[b][pl][q] = relu(sum_c w[pl][c] x[b][q][c] + bias[pl]); planes [0, npol) go to pol_out [boards][npol][81], the rest to
// val_out [boards][NPL - npol][S*S] (bf16; plane-major per board = nn.Flatten order; rows pol_stride / val_stride elements
// apart), any (S, C) of the tiled layout.  HBM-bound: one pass over the tower output.
// The NPL x C weights are read through wave-uniform (scalar) loads straight from global memory -- the kernel uses NO LDS.  Rounds 1-3
// staged them in an LDS table first; with two evaluator forwards in flight on two streams a wave of that version occasionally
// computed with a value other than the table held before and after (round 4, tools/probes/make_head_probe.py + tools/concurrency_probe2.py,
// profiles/r04_concurrency_probe2_*.txt: LDS table -> 36 / 40 rounds differ from a serial run; weights from global memory -> 0 / 40;
// the table verified intact at the end of every workgroup; tower input identical).  Without the table the question does not arise.
template <int NPL, bool F16 = false> __global__ void __launch_bounds__(256)
k_head_tiled(const unsigned char* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, unsigned short* __restrict__ pol_out,
             unsigned short* __restrict__ val_out, long long npos, int npol, int C, int P2, int tile_rows, int pol_stride, int val_stride) {
    const float* __restrict__ ws = w;  // [NPL][C], wave-uniform indices below -> s_load
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npos) return;
    const int nch = C / 8;
    const long long tile = i / tile_rows, board = i / P2;
    const int p = (int)(i - tile * tile_rows), q = (int)(i - board * P2);
    float acc[NPL];
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) acc[pl] = bias[pl];
    const unsigned char* src = x + ((size_t)tile * nch * tile_rows + (size_t)p) * 16;
#pragma unroll 4
    for (int c = 0; c < nch; ++c) {
        const cv_u32x4 v = *(const cv_u32x4*)(src + (size_t)c * tile_rows * 16);
        typedef CvFmt<F16> FM;
        const float f[8] = {FM::lo(v.x), FM::hi(v.x), FM::lo(v.y), FM::hi(v.y), FM::lo(v.z), FM::hi(v.z), FM::lo(v.w), FM::hi(v.w)};
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[pl] += f[e] * ws[pl * C + c * 8 + e];
    }
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
        const float v = fmaxf(acc[pl], 0.0f);
        const unsigned short h = CvFmt<F16>::one(v);
        if (pl < npol) pol_out[(size_t)board * pol_stride + pl * P2 + q] = h;
        else val_out[(size_t)board * val_stride + (pl - npol) * P2 + q] = h;
    }
}

// Fully connected layers of both heads + softmax / tanh (core/network.py:136-156) on the head planes k_head_tiled wrote:
//     priors[b] = softmax(Wp pol[b] + bp),   value[b] = tanh(W2 relu(W1 val[b] + b1) + b2)
// as MFMA GEMMs D[neuron][board] with one wave per 32 boards: A = zero-padded bf16 weights [NT * 32][KS * 16] streamed from L2,
// B = the boards' head planes straight from global memory (row stride = KS * 16 elements, so every fragment is one aligned
// 16-byte load), softmax / tanh on the accumulators (a board's 32 t neurons live in lanes l and l + 32).
template <int NT1, int NT2, bool F16 = false> __global__ void __launch_bounds__(256)
k_fc_heads(const unsigned short* __restrict__ pol, const unsigned short* __restrict__ val, const unsigned short* __restrict__ wp,
           const float* __restrict__ bp, int ks1, const unsigned short* __restrict__ w1, const float* __restrict__ b1, int ks2,
           const float* __restrict__ w2, float b2, float* __restrict__ priors, float* __restrict__ values, long long boards, int A) {
    const int lane = threadIdx.x & 63, bl = lane & 31, hi = lane >> 5;
    const long long task = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), b0 = task * 32;
    if (b0 >= boards) return;
    const long long b = b0 + bl < boards ? b0 + bl : boards - 1;  // tail lanes recompute the last board, never store
    const bool live = b0 + bl < boards;
    {   // ---- policy head: logits = Wp . pol + bp, softmax over the A actions
        cv_f32x16 acc[NT1];
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
        const unsigned short* xrow = pol + (size_t)b * ks1 * 16 + hi * 8;
        const unsigned short* wrow = wp + (size_t)bl * ks1 * 16 + hi * 8;
        for (int s = 0; s < ks1; ++s) {
            const cv_bf16x8 bf = *(const cv_bf16x8*)(xrow + s * 16);
#pragma unroll
            for (int t = 0; t < NT1; ++t) {
                const cv_bf16x8 af = *(const cv_bf16x8*)(wrow + (size_t)t * 32 * ks1 * 16 + s * 16);
                acc[t] = CvFmt<F16>::mfma32(af, bf, acc[t]);
            }
        }
        float mx = -__builtin_inff();
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = 32 * t + 8 * (e >> 2) + 4 * hi + (e & 3);
                const float v = n < A ? acc[t][e] + bp[n] : -__builtin_inff();
                acc[t][e] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.0f;
#pragma unroll
        for (int t = 0; t < NT1; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float ex = __expf(acc[t][e] - mx);  // exp(-inf) = 0 for the padding neurons
                acc[t][e] = ex;
                sum += ex;
            }
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        if (live) {
            float* prow = priors + (size_t)b * A;
#pragma unroll
            for (int t = 0; t < NT1; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int n = 32 * t + 8 * (e >> 2) + 4 * hi + (e & 3);
                    if (n < A) prow[n] = acc[t][e] * inv;
                }
        }
    }
    {   // ---- value head: tanh(W2 . relu(W1 . val + b1) + b2)
        cv_f32x16 acc[NT2];
#pragma unroll
        for (int t = 0; t < NT2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
        const unsigned short* xrow = val + (size_t)b * ks2 * 16 + hi * 8;
        const unsigned short* wrow = w1 + (size_t)bl * ks2 * 16 + hi * 8;
        for (int s = 0; s < ks2; ++s) {
            const cv_bf16x8 bf = *(const cv_bf16x8*)(xrow + s * 16);
#pragma unroll
            for (int t = 0; t < NT2; ++t) {
                const cv_bf16x8 af = *(const cv_bf16x8*)(wrow + (size_t)t * 32 * ks2 * 16 + s * 16);
                acc[t] = CvFmt<F16>::mfma32(af, bf, acc[t]);
            }
        }
        float part = 0.0f;
#pragma unroll
        for (int t = 0; t < NT2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = 32 * t + 8 * (e >> 2) + 4 * hi + (e & 3);  // padding neurons carry zero weights w1 / b1 / w2
                part += fmaxf(acc[t][e] + b1[n], 0.0f) * w2[n];
            }
        part += __shfl_xor(part, 32);
        if (live && hi == 0) values[b] = tanhf(part + b2);
    }
}
#endif  // __HIPCC__

