// az_conv_sp.h -- the residual tower's 3x3 convolution at the REFERENCE'S precision class on the bf16/f16-rate matrix cores:
//     y = relu(conv3x3(x, w) + bias [+ residual])              (alpha_zero/core/network.py:42-82, eval mode, BatchNorm folded;
//                                                               the reference evaluates in fp32: core/pipeline.py:91-123)
// gfx950 has no TF32-like MFMA and its f32-input MFMA runs at 1/16 of the f16 rate (157 TFLOP/s), so an fp32 tower on the library
// is 13x slower than the bf16 one.  Here every fp32 value v travels as TWO f16 numbers
//     hi = f16(v),   lo = f16((v - hi) * 2^11)            (v - hi is exact in fp32; |v - hi - lo 2^-11| <= 2^-22 |v|)
// and a product is three f16 MFMAs into two fp32 accumulators
//     main += w_hi x_hi          corr += w_hi x_lo + w_lo x_hi          result = main + 2^-11 corr
// (the dropped w_lo x_lo term is <= 2^-22 |w x|): 22-bit significands, exact f16 x f16 products, fp32 accumulation -- the error
// of one output is of the size of fp32 summation error itself (measured against fp64 next to the library's fp32 convolution:
// tests/test_split_tower.py::test_gpu_split_conv_error_vs_fp64), at 1/3 of the f16 MFMA rate = 5.3x the f32 MFMA peak.
// The lo halves are scaled by 2^11 so that they sit in f16's normal range whenever the value itself does (no reliance on
// subnormal inputs); activations are clamped to +-65504 (f16's largest finite value) when they are split.
//
// Activation layout between the layers ("split layout", azsp_split_layout converts fp32 channels-last rows):
//     [board][plane: hi, lo][C/8 chunks][S*S positions][8 channels] f16                     (2 * C * S*S * 2 bytes per board)
// Kernel = the weight-stationary scheme of az_conv.h / az_conv64.h re-balanced for two planes:
//   * one persistent 256-thread workgroup per CU, one wave per SIMD; a workgroup owns 64 output channels (a 128-filter layer is two
//     cout groups, launched so that the two groups of a board run on the same XCD = one L2 fetch of the input), wave q of it keeps
//     the hi AND lo filter banks of its 16 couts in registers: 2 x 9 taps x C/32 k-steps x 4 registers = 288 at C = 128 (256 in
//     AGPRs, read by the MFMA in place).
//   * v_mfma_f32_16x16x32_f16, tile = ONE board: 80 of its 81 positions = exactly 5 column tiles of 16, per k-step (one tap x 32 input
//     channels) 10 B fragments from LDS (5 hi + 5 lo) feed 15 MFMAs.  The 81st position, the corner (8, 0), has only 4 taps on the
//     board: its 4 input cells are copied to a small LDS side buffer while the board is resident, and every 16 boards each wave
//     multiplies one extra [16 couts] x [16 boards] x [4 taps x cin] tile on its resident filter banks (48 MFMAs per 16 boards).
//     (With all 81 positions in 6 column tiles a sixth of the matrix work multiplied repeated positions; a separate launch for the
//     corner cost 0.14 - 0.24 ms per layer: its scattered 16-byte cells are HBM sector over-fetch.)
//   * LDS image per (plane, chunk): a strip of 120 16-byte cells, cell(y, x) = 12 + 11 y + x with zero cells around the rows, so a
//     tap is a constant cell offset and every fragment address is a per-lane base + an immediate; strips are 7.5 x 256 B and the
//     lane -> position table (tools/gen_sp_map.py) makes every ds_read_b128 lane group hit all 64 banks once: conflict-free.
//   * the next board's 2 x C/8 strips arrive by LDS-DMA (global_load_lds_dwordx4, masked to the position cells) in the shadow of
//     the first k-steps; one barrier per board; epilogue on the accumulators (bias enters as the C operand of the first MFMA).
// Measured on MI355X (profiles/r03_pmc_split.txt, r03_split_bench.txt, r03_bench_driver_cmd_e.json; 32768 boards, 128 filters): 1.71 - 1.90 ms
// per layer depending on the box = 412 - 458 TFLOP/s fp32-equivalent (the library's fp32 convolution + epilogue: 7.9 - 9.1 ms), matrix
// pipe busy 72 % at a power-limited ~1.8 GHz, HBM traffic = the tensors once each (the second cout group's input read hits the XCD's L2).
#pragma once
#include "az_conv64.h"

#if defined(__HIPCC__)
#define SP_SCALE 2048.0f
#define SP_INV_SCALE (1.0f / 2048.0f)
#define SP_F16_MAX 65504.0f

struct SpGeo9 {
    // cell(y, x) = 12 + 11 y + x: two zero cells between board rows.  Position (8, 0) is NOT a column of the per-board tiles (it is
    // computed for 16 boards at a time: a corner has only 4 taps on the board): the other 80 positions are exactly 5 column tiles of 16.
    static constexpr int S = 9, P2 = 81, PITCH = 11, CELL0 = 12, CORNER = 72;
    static constexpr int CELLS = 120;  // 12 + 8 * 11 + 8 + 12; a strip is 1920 B = 7.5 x 256 B: neighbouring strips are shifted by 8 cells in bank space
    static constexpr int NCT = 5;
    static constexpr int cell_of(int p) { return CELL0 + PITCH * (p / S) + p % S; }
    static constexpr int pos_of_cell(int cell) {
        const int k = cell - CELL0;
        if (k < 0) return -1;
        const int yy = k / PITCH, xx = k % PITCH;
        return (yy < S && xx < S) ? yy * S + xx : -1;
    }
};
// Position of (column tile, lane & 15), generated by tools/gen_sp_map.py.  A ds_read_b128 is serviced in groups of 16 lanes that hold
// lanes n in A = {0-3, 12-15} of one 8-channel group kg and n in B = {4-11} of the next (whose strip is 8 cells further in bank space):
// a column tile is conflict-free iff {cell(n) mod 16 : n in A} and {cell(n) + 8 mod 16 : n in B} partition Z16 (checked below).
struct SpMap {
    unsigned char pos[SpGeo9::NCT * 16];
};
constexpr SpMap sp_map_table = {{
    46, 61, 68, 69, 34, 47, 54, 55, 59, 63, 64, 79, 70, 76, 77, 78,
    45, 56, 58, 60, 32, 33, 44, 49, 57, 62, 67, 73, 71, 74, 75, 80,
    41, 42, 43, 50, 27, 28, 29, 36, 37, 38, 40, 53, 51, 52, 65, 66,
    18, 19, 20, 23,  4,  5,  6,  7,  8,  9, 21, 22, 30, 31, 35, 48,
    14, 15, 16, 17,  0,  1,  2,  3, 10, 11, 12, 13, 24, 25, 26, 39,
}};
constexpr bool sp_map_ok(const SpMap& m) {
    bool used[SpGeo9::P2] = {};
    for (int t = 0; t < SpGeo9::NCT; ++t) {
        bool bank[16] = {};
        for (int n = 0; n < 16; ++n) {
            const int p = m.pos[t * 16 + n];
            if (p >= SpGeo9::P2 || p == SpGeo9::CORNER || used[p]) return false;
            used[p] = true;
            const int r = (SpGeo9::cell_of(p) + ((n >= 4 && n < 12) ? 8 : 0)) & 15;
            if (bank[r]) return false;
            bank[r] = true;
        }
    }
    return true;
}
static_assert(sp_map_ok(sp_map_table), "column-tile map of the split kernel: every position but the corner once, conflict-free lane groups");
static_assert((SpGeo9::CELLS * 16) % 256 == 128, "neighbouring strips must be 8 cells apart in bank space");
static __device__ const SpMap sp_map9 = sp_map_table;

typedef __attribute__((ext_vector_type(8))) _Float16 sp_f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 sp_f16x2;

__device__ __forceinline__ void sp_mfma_a(c6_f32x4& acc, const sp_f16x8& wa, const sp_f16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(wa), "v"(b));
}
__device__ __forceinline__ void sp_mfma_v(c6_f32x4& acc, const sp_f16x8& wa, const sp_f16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(b));
}
__device__ __forceinline__ void sp_mfma_ac(c6_f32x4& acc, const sp_f16x8& wa, const sp_f16x8& b, const c6_f32x4& c) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(acc) : "a"(wa), "v"(b), "v"(c));
}
// fp32 -> (hi, lo) and back
__device__ __forceinline__ void sp_split(float v, _Float16& h, _Float16& l) {
    v = fminf(fmaxf(v, -SP_F16_MAX), SP_F16_MAX);
    h = (_Float16)v;                               // v_cvt_f16_f32, round to nearest even
    l = (_Float16)((v - (float)h) * SP_SCALE);     // the difference is exact in fp32
}
__device__ __forceinline__ float sp_join(_Float16 h, _Float16 l) { return fmaf((float)l, SP_INV_SCALE, (float)h); }
// Sticky range record of the split-precision evaluator: two device words, [0] = number of lanes that met a value beyond f16's
// finite range while splitting it (the value was clamped to +-65504 where the reference's fp32 network would carry it on), [1] = the
// bits of the largest such |v| (positive floats order like unsigned integers).  Every kernel that splits values keeps the largest |v|
// it produced in a register (one v_max_f32 per element) and reports once, at its end, if that exceeded the range.  The record is the
// CALLER'S (the `range_rec` argument of the azsp_*_split entries: one per network, so that two evaluators in one process never see
// each other's events; read / reset with azsp_split_range_read); a null pointer selects the library's per-device default record
// g_sp_range (azsp_split_range_status).
__device__ unsigned g_sp_range[2];
__device__ __forceinline__ void sp_range_report(float mx, unsigned* rec) {
    if (mx > SP_F16_MAX) {  // (rare: one atomic pair per lane that saw an overflow; +inf counts -- the callers map NaN inputs to +inf)
        unsigned* r = rec ? rec : g_sp_range;
        atomicAdd(&r[0], 1u);
        atomicMax(&r[1], __float_as_uint(mx));
    }
}
// |v| for the range record of values that arrive from OUTSIDE the evaluator (azsp_split_layout / azsp_split_features): a NaN is an
// event too (v_max_f32 would drop it; the clamp of sp_split turns it into -65504)
__device__ __forceinline__ float sp_range_abs(float v) { return v != v ? __builtin_inff() : fabsf(v); }
__device__ __forceinline__ unsigned sp_pack(_Float16 a, _Float16 b) { return __builtin_bit_cast(unsigned, (sp_f16x2){a, b}); }
__device__ __forceinline__ _Float16 sp_lo16(unsigned v) { return __builtin_bit_cast(sp_f16x2, v)[0]; }
__device__ __forceinline__ _Float16 sp_hi16(unsigned v) { return __builtin_bit_cast(sp_f16x2, v)[1]; }

// ---- epilogue building blocks (one VALU instruction each; the epilogues below issue them as micro-ops between MFMAs) ----
// (f16) half `H` of rl * 2^-11 + (f16) half `H` of rh in ONE v_fma_mix_f32 = sp_join of a packed residual element (f16 inputs are exact
// in fp32 and the fma rounds once: bit-identical to the convert / convert / fma sequence)
template <int H> __device__ __forceinline__ float sp_mix_join(unsigned rh, unsigned rl) {
    float r;
    if constexpr (H == 0) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(rl), "s"(SP_INV_SCALE), "v"(rh));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(rl), "s"(SP_INV_SCALE), "v"(rh));
    return r;
}
// s - 2^11 * (f16) half `H` of hpk, with s = 2^11 v: the scaled remainder (v - hi) 2^11 of sp_split in one v_fma_mix_f32 (both products
// are exact in fp32 and so is their difference: bit-identical to the subtract-then-scale of sp_split)
template <int H> __device__ __forceinline__ float sp_mix_rem(unsigned hpk, float s) {
    float r;
    if constexpr (H == 0) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "s"(-SP_SCALE), "v"(s));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "s"(-SP_SCALE), "v"(s));
    return r;
}
__device__ __forceinline__ unsigned sp_cvt_pk(float a, float b) {  // one v_cvt_pk_f16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, (sp_f16x2){(_Float16)a, (_Float16)b});
}
// f32 v minus the f16 half H of hpk: the remainder v - hi of the split, exact in fp32, one v_fma_mix_f32
template <int H> __device__ __forceinline__ float sp_mix_diff(unsigned hpk, float v) {
    float r;
    if constexpr (H == 0) asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(v));
    else asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(hpk), "v"(v));
    return r;
}
// f16(d * 2^11) (the product is exact in fp32: one rounding, to nearest even, as v_cvt_pk_f16_f32 rounds) into the low / high half of a pair
__device__ __forceinline__ unsigned sp_scale_cvt_lo(float d) {
    unsigned r;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(r) : "v"(d), "s"(SP_SCALE));
    return r;
}
__device__ __forceinline__ unsigned sp_scale_cvt_hi(unsigned lo, float d) {
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(lo) : "v"(d), "s"(SP_SCALE));
    return lo;
}
// Micro-ops of a riding epilogue (shared by the kernels and by the compile-time mirrors of their schedules, sp9_vm_younger /
// sp17_vm_younger): per element E1 = join [, residual join, add]; per pair of elements 8 more: range record of both (one v_max3_f32), 2 x
// ReLU + clamp (one v_med3_f32 each: the lower bound is 0 with ReLU, -65504 without), packed hi convert, 2 exact remainders v - hi,
// 2 scale-and-convert v_fma_mixlo/hi_f16; per column tile 2 stores.  (Round 4: 38 / 30 micro-ops per column tile; now 30 / 22.)
__host__ __device__ constexpr int sp_epi_e1(bool res) { return res ? 3 : 1; }
__host__ __device__ constexpr int sp_epi_pair(bool res) { return 2 * sp_epi_e1(res) + 8; }
__host__ __device__ constexpr int sp_epi_ct_ops(bool res) { return 2 * sp_epi_pair(res) + 2; }
// mx = max(mx, |a|, |b|) in one v_max3_f32 (the range record of a pair of elements)
__device__ __forceinline__ float sp_max3_abs(float mx, float a, float b) {
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(mx) : "v"(a), "v"(b));
    return mx;
}
typedef __attribute__((address_space(1))) unsigned char* sp_gptr;         // pointers into global memory whose value the compiler takes as
typedef const __attribute__((address_space(1))) unsigned char* sp_gcptr;  // given: one opaque scalar base per plane (saddr + lane offset)

// Static schedule of a riding epilogue: P_OPS micro-ops spread evenly over the MFMA slots [S0, S0 + AVAIL) of the carrying unit
// (a 16-cycle MFMA hides about one VALU instruction: a flat ceil(P_OPS / AVAIL) per slot would issue bursts of two early and none late)
template <int P_OPS, int S0, int AVAIL> struct SpSpread {
    static constexpr int MAXPER = (P_OPS + AVAIL - 1) / AVAIL;
    static constexpr int cum(int sl) {  // micro-ops issued in the slots <= sl
        if (sl < S0) return 0;
        const long long c = ((long long)(sl - S0 + 1) * P_OPS + AVAIL - 1) / AVAIL;
        return c > P_OPS ? P_OPS : (int)c;
    }
    static constexpr int slot_of(int o) {  // the slot that issues micro-op o
        int sl = S0;
        while (cum(sl) <= o) ++sl;
        return sl;
    }
    static_assert(AVAIL > 0 && cum(S0 + AVAIL - 1) == P_OPS, "every micro-op has a slot");
};

// fp32 channels-last rows [boards * P2][C] <-> split layout; one thread per (board, chunk, position), positions fastest
// (the 16-byte accesses of the split side are contiguous per chunk strip)
__global__ void __launch_bounds__(256)
k_split_layout(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, long long nchunks, int to_split, int nch, int p2,
               unsigned* range) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nchunks) return;
    const long long b = i / ((long long)nch * p2);
    const int r = (int)(i - b * (long long)nch * p2), c = r / p2, p = r - c * p2;
    const size_t plane = (size_t)nch * p2 * 16, so = (size_t)b * 2 * plane + ((size_t)c * p2 + p) * 16;
    const size_t fo = (((size_t)b * p2 + p) * nch + c) * 32;
    if (to_split) {
        const cv_u32x4 a0 = *(const cv_u32x4*)(src + fo), a1 = *(const cv_u32x4*)(src + fo + 16);
        const float f[8] = {__uint_as_float(a0.x), __uint_as_float(a0.y), __uint_as_float(a0.z), __uint_as_float(a0.w),
                            __uint_as_float(a1.x), __uint_as_float(a1.y), __uint_as_float(a1.z), __uint_as_float(a1.w)};
        _Float16 h[8], l[8];
        float mx = 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            mx = fmaxf(mx, sp_range_abs(f[e]));
            sp_split(f[e], h[e], l[e]);
        }
        sp_range_report(mx, range);
        *(cv_u32x4*)(dst + so) = (cv_u32x4){sp_pack(h[0], h[1]), sp_pack(h[2], h[3]), sp_pack(h[4], h[5]), sp_pack(h[6], h[7])};
        *(cv_u32x4*)(dst + so + plane) = (cv_u32x4){sp_pack(l[0], l[1]), sp_pack(l[2], l[3]), sp_pack(l[4], l[5]), sp_pack(l[6], l[7])};
    } else {
        const cv_u32x4 h = *(const cv_u32x4*)(src + so), l = *(const cv_u32x4*)(src + so + plane);
        const unsigned hv[4] = {h.x, h.y, h.z, h.w}, lv[4] = {l.x, l.y, l.z, l.w};
        float f[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f[2 * e] = sp_join(sp_lo16(hv[e]), sp_lo16(lv[e]));
            f[2 * e + 1] = sp_join(sp_hi16(hv[e]), sp_hi16(lv[e]));
        }
        *(cv_u32x4*)(dst + fo) = (cv_u32x4){__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
        *(cv_u32x4*)(dst + fo + 16) = (cv_u32x4){__float_as_uint(f[4]), __float_as_uint(f[5]), __float_as_uint(f[6]), __float_as_uint(f[7])};
    }
}

// Observation planes [boards][Cin][P2] fp32 (NCHW, what the engine's AZSP_FEAT_F32 features are) -> split layout with 32 channels
// (4 chunks; channels >= Cin zero) = the stem's input.  One thread per (board, chunk, position), positions fastest.
__global__ void __launch_bounds__(256)
k_split_features(const float* __restrict__ src, unsigned char* __restrict__ dst, long long nitems, int cin, int p2, unsigned* range) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nitems) return;
    const long long b = i / (4 * p2);
    const int r = (int)(i - b * 4 * p2), c = r / p2, p = r - c * p2;
    _Float16 h[8], l[8];
    float mx = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = 8 * c + e;
        const float v = ch < cin ? src[((size_t)b * cin + ch) * p2 + p] : 0.0f;
        mx = fmaxf(mx, sp_range_abs(v));
        sp_split(v, h[e], l[e]);
    }
    sp_range_report(mx, range);
    const size_t plane = (size_t)4 * p2 * 16, so = (size_t)b * 2 * plane + ((size_t)c * p2 + p) * 16;
    *(cv_u32x4*)(dst + so) = (cv_u32x4){sp_pack(h[0], h[1]), sp_pack(h[2], h[3]), sp_pack(h[4], h[5]), sp_pack(h[6], h[7])};
    *(cv_u32x4*)(dst + so + plane) = (cv_u32x4){sp_pack(l[0], l[1]), sp_pack(l[2], l[3]), sp_pack(l[4], l[5]), sp_pack(l[6], l[7])};
}

// Both heads of the network (core/network.py:118-156) in fp32 on the split layout, one pass over the tower output:
//   head planes  hp[pl][pos] = relu(sum_c hw[pl][c] x[c][pos] + hb[pl])      (the two 1x1 convolutions, BatchNorm folded; pl < 3)
//   priors = softmax(Wp flatten(hp[0 .. npol)) + bp)       value = tanh(w2 . relu(W1 flatten(hp[npol .. 3)) + b1) + b2)
// BPB boards per 256-thread workgroup (the fully connected weights, stored TRANSPOSED [inputs][outputs] so that neighbouring threads
// read neighbouring outputs, are streamed from L2 once per BPB boards; BPB = 8 was measured SLOWER than 4 at 17x17 (1.77 vs 1.15 ms:
// fewer, longer workgroups -- the fully connected loop is latency-bound, not traffic-bound); plain fp32 FMAs -- HBM-bound (2 x C x P2 x 2 bytes per board).
// The 3 x C weights of the 1x1 convolutions are read through wave-uniform (scalar) loads straight from global memory, as k_head_tiled
// does since round 4: no product kernel keeps a wave-uniformly read LDS weight table.  LDS holds only data the workgroup
// itself produced (head planes, logits).  dynamic LDS: BPB (3 ceil4(P2) + A + F) floats.
template <int BPB> __global__ void __launch_bounds__(256)
k_head_split(const unsigned char* __restrict__ x, const float* __restrict__ hw, const float* __restrict__ hb, const float* __restrict__ wp_t,
             const float* __restrict__ bp, const float* __restrict__ w1_t, const float* __restrict__ b1, const float* __restrict__ w2, float b2,
             float* __restrict__ priors, float* __restrict__ values, long long boards, int C, int P2, int A, int F, int npol) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int P2P = (P2 + 3) & ~3;                   // a head plane's row in LDS, padded to 16 bytes: the fully connected loop reads float4s
    const float* __restrict__ ws = hw;               // [3][C], wave-uniform indices below -> s_load
    float* hp = sm;                                  // [BPB][3][P2P]
    float* out = hp + BPB * 3 * P2P;                 // [BPB][A + F]
    const int tid = threadIdx.x, nch = C / 8;
    const long long b0 = (long long)blockIdx.x * BPB;
    const size_t plane = (size_t)nch * P2 * 16;
    // head planes: one thread per (board, position) computes all three planes from ONE pass over the position's C channels (the first
    // version walked (board, plane, position) items and read every activation three times: 1.13 ms per forward at 17x17, rocprofv3 round 4)
    for (int it = tid; it < BPB * P2; it += 256) {
        const int b = it / P2, pos = it - b * P2;
        float acc[3] = {0.0f, 0.0f, 0.0f};
        if (b0 + b < boards) {
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) acc[pl] = hb[pl];
            const unsigned char* src = x + (size_t)(b0 + b) * 2 * plane + (size_t)pos * 16;
#pragma unroll 4
            for (int c = 0; c < nch; ++c) {
                const cv_u32x4 h = *(const cv_u32x4*)(src + (size_t)c * P2 * 16), l = *(const cv_u32x4*)(src + (size_t)c * P2 * 16 + plane);
                const unsigned hv[4] = {h.x, h.y, h.z, h.w}, lv[4] = {l.x, l.y, l.z, l.w};
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] = sp_join(sp_lo16(hv[e]), sp_lo16(lv[e]));
                    v[2 * e + 1] = sp_join(sp_hi16(hv[e]), sp_hi16(lv[e]));
                }
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[pl] = fmaf(v[e], ws[pl * C + c * 8 + e], acc[pl]);
            }
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) acc[pl] = fmaxf(acc[pl], 0.0f);
        }
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) hp[(b * 3 + pl) * P2P + pos] = acc[pl];
    }
    __syncthreads();
    // Fully connected layers: thread o = one output neuron for the BPB boards.  Inputs in nn.Flatten order k = plane * P2 + position; the
    // accumulation order of every output is k ascending (as a plain loop over k).  The head planes are read as float4 (one broadcast
    // ds_read_b128 per board and 4 inputs): with one ds_read_b32 per input the loop was bound by LDS instruction issue (rocprofv3, round 4:
    // 1.15 ms per forward at 17x17, 0.39 ms at 9x9 -- 2312 / 648 LDS instructions per thread and board group).
    for (int o = tid; o < A + F; o += 256) {
        float acc[BPB];
        const bool pol = o < A;
        const float* wcol = pol ? wp_t + o : w1_t + (o - A);
        const int ld = pol ? A : F, pl0 = pol ? 0 : npol, pl1 = pol ? npol : 3;
        const float bias0 = pol ? bp[o] : b1[o - A];
#pragma unroll
        for (int b = 0; b < BPB; ++b) acc[b] = bias0;
        for (int pl = pl0; pl < pl1; ++pl) {
            const float* wpl = wcol + (size_t)(pl - pl0) * P2 * ld;
            const float* hpl = hp + pl * P2P;
            int pos = 0;
            for (; pos + 8 <= P2; pos += 8) {
                float wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) wv[u] = wpl[(size_t)(pos + u) * ld];
#pragma unroll
                for (int b = 0; b < BPB; ++b) {
                    const c6_f32x4 h0 = *(const c6_f32x4*)(hpl + b * 3 * P2P + pos), h1 = *(const c6_f32x4*)(hpl + b * 3 * P2P + pos + 4);
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[b] = fmaf(h0[u], wv[u], acc[b]);
#pragma unroll
                    for (int u = 0; u < 4; ++u) acc[b] = fmaf(h1[u], wv[4 + u], acc[b]);
                }
            }
            for (; pos < P2; ++pos) {
                const float wv = wpl[(size_t)pos * ld];
#pragma unroll
                for (int b = 0; b < BPB; ++b) acc[b] = fmaf(hpl[b * 3 * P2P + pos], wv, acc[b]);
            }
        }
#pragma unroll
        for (int b = 0; b < BPB; ++b) out[b * (A + F) + o] = pol ? acc[b] : fmaxf(acc[b], 0.0f);
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;  // wave w finishes boards b0 + w, b0 + w + 4, ...
    for (int bb = wave; bb < BPB; bb += 4) {
        if (b0 + bb >= boards) break;
        const float* o = out + bb * (A + F);
        float mx = -__builtin_inff();
        for (int a = lane; a < A; a += 64) mx = fmaxf(mx, o[a]);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
        float sum = 0.0f;
        for (int a = lane; a < A; a += 64) sum += expf(o[a] - mx);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor(sum, d);
        const float inv = 1.0f / sum;
        float* prow = priors + (size_t)(b0 + bb) * A;
        for (int a = lane; a < A; a += 64) prow[a] = expf(o[a] - mx) * inv;
        float v = 0.0f;
        for (int f = lane; f < F; f += 64) v = fmaf(o[A + f], w2[f], v);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
        if (lane == 0) values[b0 + bb] = tanhf(v + b2);
    }
}

// first MFMA of an accumulator that starts from zero: the C operand is the inline constant 0 (no zeroed registers to keep)
__device__ __forceinline__ void sp_mfma_a0(c6_f32x4& acc, const sp_f16x8& wa, const sp_f16x8& b) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "a"(wa), "v"(b));
}
// Vector-memory instructions a wave of k_conv3x3_sp issues between the last LDS-DMA piece of the next board (unit 0, k-step NPIECE) and
// the board's barrier (unit 1, k-step KS - 2): the stores of the riding epilogues (2 per column tile) and unit 1's residual loads (2 per
// column tile, first slots of the unit).  Mirrors the kernel's schedule (same constants, same SpSpread arithmetic).
// Slots of the B-fragment ring (k-steps): a fragment is requested R - 1 k-steps before its MFMAs.  (Round 5 tried 4 slots on the plain
// 128-filter layer after the PMC wave-cycle breakdown showed 14 - 15 % of the wave time parked at s_waitcnt: no gain on the same box,
// 1.792 vs 1.794 ms, profiles/r05_split_ablation.txt -- the waits are not fragment latency here; the skip layer has no registers for it.)
__host__ __device__ constexpr int sp9_ring(int, bool) { return 3; }
template <bool RES, int NCH, bool XLO0 = false> __host__ __device__ constexpr int sp9_vm_younger() {
    constexpr int KSUB = NCH / 4, KS = 9 * KSUB, R = sp9_ring(NCH, RES), S0 = 6, NP = (SpGeo9::CELLS + 63) / 64, NPIECE = NP * ((XLO0 ? 1 : 2) * NCH / 4);
    constexpr int CT_OPS = sp_epi_ct_ops(RES);
    int n = 0;
    for (int i = 0; i < 2; ++i) {
        const int nj = i == 0 ? 3 : 2, pnj = i == 0 ? 2 : 3, NQ = (XLO0 ? 2 : 3) * nj, P_OPS = pnj * CT_OPS;
        const int AVAIL = (i == 1 ? (KS - (R - 1)) * NQ : NQ * KS - 4) - S0;
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
                if (i > 0 || sl / NQ > NPIECE) ++n;  // a DMA piece is issued behind all MFMA slots of its k-step
            }
        if (RES && i > 0) n += 2 * nj;  // (unit 0 loads its residual in its first slots, before the first piece)
    }
    return n;
}

// ---------------------------------------------------------------------------------------------------------------------
// k_conv3x3_sp<RES, NCH, NCG>: NCH = input-channel chunks of 8 (16: 128 -> C layer, 8: 64 -> C layer, 4: the stem, 17 planes padded to 32
// -> C); NCG = cout groups of 64 (C = 64 NCG).
// w: [plane: hi, lo][9 taps][C couts][8 NCH cin] f16 with lo = (w - hi) * 2^11; bias fp32 [C].
// The EPILOGUE is SOFTWARE-PIPELINED into the MFMA stream (the scheme of k_conv3x3_tiled / k_resblock64): a board is 2 units (column
// tiles 0-2 and 3-4) with two accumulator sets; while unit i multiplies into set i, the epilogue of the previous unit (join of the two
// accumulators, residual join + add, clamp, split into hi / lo, two 8-byte stores per column tile) is issued one micro-op per MFMA
// gap from the other set (measured against an exposed per-board epilogue: -4 %); the B-fragment ring (3 k-steps) runs on across units
// and boards; the next board's LDS-DMA pieces ride in unit 0; ONE barrier per board (unit 1, two k-steps before its end: every
// read of the current buffer has been issued and each wave's pieces of the next board have landed).
// XLO0: the caller guarantees that the input's lo plane is all zero (exact f16 values: the engine's 0 / 1 observation planes,
// AZSP_FEAT_F16_SPLIT) -- the lo strips are not loaded, their fragments not read and the w_hi x_lo product is skipped (it is exactly zero).
template <bool RES, int NCH, int NCG, bool XLO0 = false> __global__ void __launch_bounds__(CW_THREADS, 1)
k_conv3x3_sp(const unsigned char* __restrict__ x, const _Float16* __restrict__ w, const float* __restrict__ bias,
             const unsigned char* __restrict__ res, unsigned char* __restrict__ y, int ntiles, int relu, unsigned* range) {
    typedef SpGeo9 G;
    constexpr int C = 64 * NCG, CIN = 8 * NCH, KSUB = NCH / 4;
    constexpr int KS = 9 * KSUB;                             // k-steps per unit (one tap x 32 input channels)
    constexpr int NJ0 = 3, NJ1 = 2, R = sp9_ring(NCH, RES);       // column tiles of unit 0 / unit 1, ring slots (k-steps)
    constexpr int LBLK = G::CELLS * 16, LPLANE = NCH * LBLK, LBUF = 2 * LPLANE;
    constexpr int GBLK = G::P2 * 16, XPLANE = NCH * GBLK, XTILE = 2 * XPLANE;
    constexpr int YPLANE = (C / 8) * GBLK, YTILE = 2 * YPLANE;
    constexpr int NP = (G::CELLS + 63) / 64;
    constexpr int SPW = (XLO0 ? 1 : 2) * NCH / 4, NPIECE = NP * SPW;  // (XLO0: the hi plane's strips only = strips 0 .. NCH - 1)
    constexpr int NPROD = XLO0 ? 2 : 3;                               // MFMA products per multiply
    static_assert(!XLO0 || (!RES && NCH == 4), "exact-f16 inputs: the stem");
    constexpr int NF = 2 * KS, NF_A = NF < 64 ? NF : 64;
    constexpr int E1 = sp_epi_e1(RES), PAIR = sp_epi_pair(RES), CT_OPS = sp_epi_ct_ops(RES);  // epilogue micro-ops (see sp_epi_e1)
    constexpr int S0 = 6;                                    // first MFMA slot of a unit that may touch the previous unit's accumulators
    static_assert((2 * KS) % R == 0, "a board's k-steps keep the ring phase");
    static_assert(KS - 1 >= NPIECE, "the next board's pieces ride in unit 0");
    static_assert(LPLANE + 3 * 4 * LBLK + (2 * G::PITCH + 2) * 16 < 65536, "fragment addresses are a base + a 16-bit immediate");
    // x double buffer, + one zero cell (the (+1, +1) tap of (8, 8) in the last strip), + the corner side buffer: for each of the last
    // <= 16 boards of this workgroup the 4 cells the corner (8, 0) multiplies ((7, 0), (7, 1), (8, 0), (8, 1); both planes, all chunks)
    constexpr int SIDE_BOARD = 8 * NCH * 16 + 16;  // bytes per board; + 16: neighbouring boards (= MFMA columns) fall on different banks
    constexpr int SIDE0 = 2 * LBUF + 16;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[SIDE0 + 16 * SIDE_BOARD];
    static_assert(SIDE0 + 16 * SIDE_BOARD <= 160 * 1024, "LDS budget");
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
    // workgroup -> (cout group, board slot).  Workgroups are dealt to the 8 XCDs round-robin; when the grid allows it the NCG groups of a
    // board are workgroups b and b + 8, ... (same XCD, launched together): the board's second read hits that XCD's L2.
    int cg, slot, nslot;
    {
        const int b = (int)blockIdx.x, nb = (int)gridDim.x;
        nslot = nb / NCG;
        if (nb % (8 * NCG) == 0) {
            const int xcd = b & 7, k = b >> 3;
            cg = k % NCG;
            slot = xcd + 8 * (k / NCG);
        } else {
            cg = b % NCG;
            slot = b / NCG;
        }
    }
    for (int i = tid; i < (SIDE0 + 16 * SIDE_BOARD) / 16; i += CW_THREADS) *(cv_u32x4*)(lds + i * 16) = (cv_u32x4){0u, 0u, 0u, 0u};
    CV_BARRIER();  // the zero cells are final before any LDS-DMA piece can land
    if (slot >= ntiles) return;  // (uniform per workgroup)

    // A fragments: fragment f = plane * KS + (tap * KSUB + ks): lane (cout = 64 cg + 16 wave + l15, cin = 32 ks + 8 kg .. + 8)
    sp_f16x8 wf[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int pl = f / KS, st = f % KS;
        wf[f] = *(const sp_f16x8*)(w + ((size_t)((pl * 9 + st / KSUB) * C + cg * 64 + wave * 16 + l15)) * CIN + (st % KSUB) * 32 + kg * 8);
    }
    c6_f32x4 bv;  // bias in the D layout (rows = couts 4 kg + e of the wave's 16): the C operand of the first k-step
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = bias[cg * 64 + wave * 16 + 4 * kg + e];
    const float lo_relu = relu ? 0.0f : -__builtin_inff();
    const float lo_clamp = relu ? 0.0f : -SP_F16_MAX;  // lower bound of the epilogue's median: ReLU and the range clamp in one instruction

    // LDS-DMA plan: a strip is NP pieces of 64 cells; wave q moves strips SPW q .. SPW q + SPW - 1 (strip = plane * NCH + chunk)
    unsigned dsrc[NP];
    unsigned long long dmask[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = 64 * i + lane < G::CELLS ? G::pos_of_cell(64 * i + lane) : -1;
        dsrc[i] = (unsigned)((p < 0 ? 0 : p) * 16);
        dmask[i] = __builtin_amdgcn_ballot_w64(p >= 0);
    }
    auto dma_piece = [&](const unsigned char* src, unsigned dstbuf, bool live, int i) {
        const int c = SPW * wave + i / NP, pc = i % NP;
        const unsigned long long base = (unsigned long long)(src + (size_t)c * GBLK);
        const unsigned long long mask = live ? dmask[pc] : 0ull;
        const unsigned dst = dstbuf + (unsigned)(c * LBLK + pc * 1024);
        asm volatile("s_mov_b64 exec, %0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1"
                     :
                     : "s"(mask), "s"(dst), "v"(dsrc[pc]), "s"(base)
                     : "memory");
    };
    unsigned lmap[G::NCT];
#pragma unroll
    for (int j = 0; j < G::NCT; ++j) {
        const int pos = sp_map9.pos[j * 16 + l15];
        // low 16 bits: LDS byte offset of the (-1, -1) neighbour in this lane's 8-channel group; high 16 bits: byte offset of this lane's
        // 8-byte output slot inside the wave's two chunk strips of a plane (couts 4 kg .. + 4 = chunk kg / 2, half kg % 2)
        lmap[j] = (unsigned)((G::CELL0 + G::PITCH * (pos / G::S) + pos % G::S - G::CELL0) * 16 + kg * LBLK) |
                  ((unsigned)(pos * 16 + (kg >> 1) * GBLK + (kg & 1) * 8) << 16);
    }
    static_assert(G::P2 * 16 + GBLK + 8 < 65536, "output slot offsets fit 16 bits");
    // the output slot offsets once more as plain 32-bit lane offsets: a global access is then scalar base + this register (the
    // `out_off` barrier below keeps the zero-extension next to its use, where instruction selection folds it into the saddr form)
    unsigned omap[G::NCT];
#pragma unroll
    for (int j = 0; j < G::NCT; ++j) omap[j] = lmap[j] >> 16;
    auto out_off = [&](int j) __attribute__((always_inline)) {
        unsigned v = omap[j];
        asm volatile("" : "+v"(v));
        return v;
    };
    sp_f16x8 bb[R][2][NJ0];  // ring of B fragments [k-step slot][plane][column tile of the unit]
    auto load_step = [&](const unsigned char* img, int j0, int nj, int s, int rs) {
        const int tap = s / KSUB;
        const int off = ((tap / 3) * G::PITCH + (tap % 3)) * 16 + (s % KSUB) * (4 * LBLK);
#pragma unroll
        for (int j = 0; j < NJ0; ++j)
            if (j < nj) bb[rs][0][j] = *(const sp_f16x8*)(img + (lmap[j0 + j] & 0xffffu) + off);
        if constexpr (!XLO0) {
#pragma unroll
            for (int j = 0; j < NJ0; ++j)
                if (j < nj) bb[rs][1][j] = *(const sp_f16x8*)(img + (lmap[j0 + j] & 0xffffu) + off + LPLANE);
        }
    };

    // fragment f of a k-step (plane f / nj, column tile f % nj): inside the units the requests are SPREAD over the first MFMA gaps of the
    // k-step R - 1 before their use -- one ds_read_b128 per gap instead of a burst of 2 nj reads in front of a k-step, during which no
    // MFMA is issued (measured on k_conv3x3_sp2, same box: -4.0 % plain / -3.5 % residual per launch, bit-identical)
    auto load_frag = [&](const unsigned char* img, int j0, int nj, int s, int rs, int f) __attribute__((always_inline)) {
        const int tap = s / KSUB;
        const int off = ((tap / 3) * G::PITCH + (tap % 3)) * 16 + (s % KSUB) * (4 * LBLK);
        const int pl = f / nj, j = f % nj;
        bb[rs][pl][j] = *(const sp_f16x8*)(img + (lmap[j0 + j] & 0xffffu) + off + pl * LPLANE);
    };

    {   // first board: all pieces at once, then the first fragments
        const unsigned char* src = x + (size_t)slot * XTILE;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) dma_piece(src, lds0, true, i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CV_BARRIER();
#pragma unroll
        for (int s = 0; s < R - 1; ++s) load_step(lds, 0, NJ0, s, s);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {  // the compiler's wait for the weight loads belongs in front of the loop (see az_conv.h)
        if (f < NF_A) asm volatile("" : : "a"(wf[f]));
        else asm volatile("" : : "v"(wf[f]));
    }
    asm volatile("" : : "v"(bv), "v"(lmap[0]), "v"(lmap[G::NCT - 1]), "v"(dsrc[0]), "v"(dsrc[NP - 1]));

    c6_f32x4 accm[2][NJ0], accc[2][NJ0];  // [unit][column tile of the unit]
    cv_u32x2 rr[2][NJ0][2];             // residual of a unit: [unit][column tile][plane]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < NJ0; ++j) {
            accm[a][j] = (c6_f32x4){0.0f, 0.0f, 0.0f, 0.0f}, accc[a][j] = (c6_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            rr[a][j][0] = (cv_u32x2){0u, 0u}, rr[a][j][1] = (cv_u32x2){0u, 0u};
        }
    float evv[2] = {0.0f, 0.0f}, sc[2] = {0.0f, 0.0f}, t0 = 0.0f, mx = 0.0f;  // mx: the largest |value| this lane produced (range record, sp_range_report)
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
            const unsigned gq = out_off(mj);
            if (o == 2 * PAIR) {
                if (store_ok) *(cv_u32x2*)(out + gq) = (cv_u32x2){hpk[0], hpk[1]};
            } else if (store_ok) *(__attribute__((address_space(1))) cv_u32x2*)(out_lo + gq) = (cv_u32x2){lpk[0], lpk[1]};
        }
    };

    // Corner side buffer.  Thread tid < 8 NCH copies one 16-byte cell per board: cell index c = tid = ((kg * 4 + tap) * 2 + plane) * KSUB + ks
    // (tap = 2 dy' + dx': board cells (7 + dy', dx')), i.e. a column's four 8-channel groups are 512 B apart = on the same banks, and the
    // corner phase reads B fragment (tap, plane, ks) of lane (board n = l15, group kg) at  n SIDE_BOARD + 16 c : conflict-free.
    const bool copier = tid < 8 * NCH;
    int csrc, cdst;
    {
        const int c = copier ? tid : 0, ks = c % KSUB, pl = (c / KSUB) & 1, tp = (c / (2 * KSUB)) & 3, g4 = c / (8 * KSUB);
        csrc = (pl * NCH + 4 * ks + g4) * LBLK + (G::CELL0 + G::PITCH * (G::S - 2 + (tp >> 1)) + (tp & 1)) * 16;
        cdst = SIDE0 + c * 16;
    }
    cv_u32x4 ctmp = (cv_u32x4){0u, 0u, 0u, 0u};
    static_assert(KS >= 6, "the side-buffer copy rides in unit 0");

    int it = 0;
    unsigned char* yprev = y;
    sp_gptr yprev_lo = (sp_gptr)(unsigned long long)y;
    for (int tile = slot; tile < ntiles; tile += nslot, ++it) {
        const int buf = it & 1;
        const unsigned char* Xs = lds + buf * LBUF;
        const unsigned char* Xn = lds + (buf ^ 1) * LBUF;
        const bool has_next = tile + nslot < ntiles;
        const unsigned char* nsrc = x + (size_t)(has_next ? tile + nslot : tile) * XTILE;
        const unsigned ndst = lds0 + (unsigned)((buf ^ 1) * LBUF);
        const size_t yo = (size_t)tile * YTILE + (size_t)(cg * 8 + wave * 2) * GBLK;  // uniform: the lane part is in lmap
        const unsigned char* rbase = RES ? res + yo : nullptr;
        unsigned char* ybase = y + yo;
        // one uniform base per plane, opaque to the compiler (it would otherwise fold base + YPLANE + lane offset into 64-bit VALU adds per
        // access: YPLANE exceeds the 13-bit immediate): every residual load / y store is scalar base + 32-bit lane offset
        unsigned long long rlo = (unsigned long long)(RES ? res + yo : y + yo) + YPLANE, ylo = (unsigned long long)(y + yo) + YPLANE;
        asm volatile("" : "+s"(rlo), "+s"(ylo));
        const sp_gcptr rbase_lo = (sp_gcptr)rlo;
        const sp_gptr ybase_lo = (sp_gptr)ylo;
        const bool have_prev = it > 0;
        auto unit = [&](auto IC) __attribute__((always_inline)) {
            constexpr int i = decltype(IC)::value, set = i, pset = i ^ 1;
            constexpr int nj = i == 0 ? NJ0 : NJ1, j0 = i == 0 ? 0 : NJ0;          // this unit's column tiles
            constexpr int pnj = i == 0 ? NJ1 : NJ0, pj0 = i == 0 ? NJ0 : 0;        // the previous unit's
            constexpr int nnj = pnj, nj0 = pj0;                                     // the next unit's (= the other one)
            constexpr int NQ = NPROD * nj, P_OPS = pnj * CT_OPS;                    // MFMAs per k-step; micro-ops of the riding epilogue
            // the riders are spread evenly over the unit's MFMA gaps and end 4 slots before the unit does; in unit 1 they end before the
            // barrier (whose counted wait knows exactly which vector-memory instructions are younger than the next board's DMA pieces)
            constexpr int AVAIL = (i == 1 ? (KS - (R - 1)) * NQ : NQ * KS - 4) - S0;
            typedef SpSpread<P_OPS, S0, AVAIL> SP;
            static_assert(SP::MAXPER <= (NCH >= 16 ? 1 : NCH >= 8 ? 2 : XLO0 ? 6 : 4), "the previous unit's epilogue fits this unit's MFMA gaps");
            unsigned char* pout = i == 0 ? yprev : ybase;
            const sp_gptr pout_lo = i == 0 ? yprev_lo : ybase_lo;
            const bool pstore = i > 0 || have_prev;
            cp_for_each([&](auto TC) __attribute__((always_inline)) {
                constexpr int t = decltype(TC)::value;
                constexpr int g = i * KS + t;  // running k-step of the board
                if constexpr (i == 1 && t == KS - (R - 1)) {
                    // every read of this buffer has been issued.  This wave's pieces of the next board (unit 0) are older than the
                    // VM_YOUNGER youngest vector-memory instructions it has issued (stores of the riding epilogues, unit 1's residual
                    // loads: counted at compile time, each is issued unconditionally); those may stay in flight
                    constexpr int VM_YOUNGER = sp9_vm_younger<RES, NCH, XLO0>();
                    static_assert(VM_YOUNGER < 63, "vmcnt field");
                    if (have_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_YOUNGER) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // first board: the stores riding in its unit 0 were skipped
                    CV_BARRIER();
                }
                // the fragments of k-step t + R - 1 (of this unit, or of the next unit: unit 1 of this board / unit 0 of the next board's image)
                constexpr bool lsame = t + R - 1 < KS;
                constexpr int ls = lsame ? t + R - 1 : t + R - 1 - KS, lnj = lsame ? nj : nnj, lj0 = lsame ? j0 : nj0, lrs = (g + R - 1) % R;
                const unsigned char* limg = (lsame || i == 0) ? Xs : Xn;
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
                        if constexpr (fa < NF_A) sp_mfma_a(accc[set][j], wf[fa], bb[g % R][pl][j]);
                        else sp_mfma_v(accc[set][j], wf[fa], bb[g % R][pl][j]);
                    }
                    constexpr int sl = t * NQ + q;  // MFMA slot of the unit
                    cp_for_each([&](auto KC) __attribute__((always_inline)) {
                        constexpr int o = SP::cum(sl - 1) + decltype(KC)::value;
                        if constexpr (o < SP::cum(sl)) epi_op(pset, o / CT_OPS, pj0 + o / CT_OPS, pout, pout_lo, o % CT_OPS, pstore);
                    }, typename CpMakeSeq<SP::MAXPER>::type{});
                    if constexpr (RES && sl < 2 * nj) {  // this unit's residual (used by its epilogue inside the next unit)
                        constexpr int rj = sl >> 1, rp = sl & 1;
                        if constexpr (rp == 0) rr[set][rj][rp] = *(const cv_u32x2*)(rbase + out_off(j0 + rj));
                        else rr[set][rj][rp] = *(const __attribute__((address_space(1))) cv_u32x2*)(rbase_lo + out_off(j0 + rj));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }, typename CpMakeSeq<NQ>::type{});
                if constexpr (i == 0 && t >= 1 && t - 1 < NPIECE) dma_piece(nsrc, ndst, has_next, t - 1);
                if constexpr (i == 0 && t == 2) {  // this board's corner cells -> side buffer column it & 15 (read now, written three k-steps on)
                    if (copier) ctmp = *(const cv_u32x4*)(Xs + csrc);
                }
                if constexpr (i == 0 && t == 5) {
                    if (copier) *(cv_u32x4*)(lds + cdst + (it & 15) * SIDE_BOARD) = ctmp;
                }
                __builtin_amdgcn_sched_barrier(0);
            }, typename CpMakeSeq<KS>::type{});
        };
        unit(CpInt<0>{});
        unit(CpInt<1>{});
        yprev = ybase, yprev_lo = ybase_lo;
        if ((it & 15) == 15 || !has_next) {
            // ---- the corner (8, 0) of the last (it & 15) + 1 boards: a [16 couts] x [<= 16 boards] x [4 taps x cin] tile per wave on the resident
            // filter banks.  The side buffer is complete: every wave passed this board's barrier after the copy of its column.
            const int n_ok = (it & 15) + 1, it0 = it - (it & 15);
            c6_f32x4 cm, cc;
            const unsigned char* sb = lds + SIDE0 + l15 * SIDE_BOARD + kg * (8 * KSUB * 16);
            // D layout: column = board l15 of the group, rows = couts 4 kg + e of the wave's 16
            const int nb = l15 < n_ok ? l15 : 0;
            const size_t co = (size_t)(slot + (size_t)(it0 + nb) * nslot) * YTILE + (size_t)(cg * 8 + wave * 2 + (kg >> 1)) * GBLK + G::CORNER * 16 + (kg & 1) * 8;
            // The phase is exposed (nothing rides behind it): the residual of the 16 corners is requested first and arrives under the MFMAs (in
            // unit 0's residual registers, free here: its epilogue has run, the next board reloads them), and the side-buffer fragments of step
            // u + 1 are read before the MFMAs of step u.  (Ablation, round 4, profiles/r04_split_ablation.txt: the whole phase costs 1.4 % of a plain
            // and 2.4 % of a residual launch, three times its 48 MFMAs; overlapping these latencies recovered only 0.2 % -- the rest is the break in
            // the software pipeline around the phase and its barrier.)
            if (RES) {
                rr[0][0][0] = *(const cv_u32x2*)(res + co);
                rr[0][0][1] = *(const cv_u32x2*)(res + co + YPLANE);
            }
            sp_f16x8 fb[2][2];  // [step parity][plane]
            fb[0][0] = *(const sp_f16x8*)(sb);
            if constexpr (!XLO0) fb[0][1] = *(const sp_f16x8*)(sb + KSUB * 16);
            cp_for_each([&](auto TC) __attribute__((always_inline)) {
                constexpr int u = decltype(TC)::value, tp = u / KSUB, ks = u % KSUB;         // tap = (7 + tp / 2, tp % 2) relative to (8, 0)
                constexpr int fa = ((tp >> 1) * 3 + 1 + (tp & 1)) * KSUB + ks;              // weight k-step of tap (dy, dx) = (tp / 2 - 1, tp % 2)
                if constexpr (u + 1 < 4 * KSUB) {
                    constexpr int tn = (u + 1) / KSUB, kn = (u + 1) % KSUB;
                    fb[(u + 1) & 1][0] = *(const sp_f16x8*)(sb + ((tn * 2 + 0) * KSUB + kn) * 16);
                    if constexpr (!XLO0) fb[(u + 1) & 1][1] = *(const sp_f16x8*)(sb + ((tn * 2 + 1) * KSUB + kn) * 16);
                }
                const sp_f16x8& bh = fb[u & 1][0];
                if constexpr (u == 0) sp_mfma_ac(cm, wf[fa], bh, bv);
                else sp_mfma_a(cm, wf[fa], bh);
                if constexpr (!XLO0) {
                    const sp_f16x8& bl = fb[u & 1][1];
                    if constexpr (u == 0) sp_mfma_a0(cc, wf[fa], bl);
                    else sp_mfma_a(cc, wf[fa], bl);
                }
                if constexpr (XLO0 && u == 0) sp_mfma_a0(cc, wf[KS + fa], bh);
                else if constexpr (KS + fa < NF_A) sp_mfma_a(cc, wf[KS + fa], bh);
                else sp_mfma_v(cc, wf[KS + fa], bh);
                __builtin_amdgcn_sched_barrier(0);
            }, typename CpMakeSeq<4 * KSUB>::type{});
            asm volatile("s_nop 15\n\ts_nop 15" : "+v"(cm), "+v"(cc));
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(cc[e], SP_INV_SCALE, cm[e]);
            if (RES) {
                const cv_u32x2 rh = rr[0][0][0], rl = rr[0][0][1];
                v[0] += sp_join(sp_lo16(rh.x), sp_lo16(rl.x));
                v[1] += sp_join(sp_hi16(rh.x), sp_hi16(rl.x));
                v[2] += sp_join(sp_lo16(rh.y), sp_lo16(rl.y));
                v[3] += sp_join(sp_hi16(rh.y), sp_hi16(rl.y));
            }
            _Float16 h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = fmaxf(v[e], lo_relu);
                if (l15 < n_ok) mx = fmaxf(mx, fabsf(v[e]));
                sp_split(v[e], h[e], l[e]);
            }
            if (l15 < n_ok) {
                *(cv_u32x2*)(y + co) = (cv_u32x2){sp_pack(h[0], h[1]), sp_pack(h[2], h[3])};
                *(cv_u32x2*)(y + co + YPLANE) = (cv_u32x2){sp_pack(l[0], l[1]), sp_pack(l[2], l[3])};
            }
            CV_BARRIER();  // the next group's copies may overwrite the side buffer
        }
    }
    // epilogue of the very last unit (accumulator set 1, column tiles 3 and 4)
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(accm[1][0]), "+v"(accm[1][1]), "+v"(accc[1][0]), "+v"(accc[1][1]));
#pragma unroll
    for (int j = 0; j < NJ1; ++j)
#pragma unroll
        for (int o = 0; o < CT_OPS; ++o) epi_op(1, j, NJ0 + j, yprev, yprev_lo, o, true);
    sp_range_report(mx, range);
}

#endif  // __HIPCC__
