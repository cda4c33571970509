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
// tests/test_network.py::test_gpu_split_conv_error_vs_fp64), at 1/3 of the f16 MFMA rate = 5.3x the f32 MFMA peak.
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
//   * v_mfma_f32_16x16x32_f16, tile = ONE board: 81 positions = 6 column tiles of 16 (15 of the 96 slots repeat a position, never
//     masked); per k-step (one tap x 32 input channels) 12 B fragments from LDS (6 hi + 6 lo) feed 18 MFMAs.
//   * LDS image per (plane, chunk): a strip of 112 16-byte cells, cell(y, x) = 11 + 10 y + x with zero cells around the rows, so a
//     tap is a constant cell offset and every fragment address is a per-lane base + an immediate; strips are 7 x 256 B, the 16
//     positions of a column tile are distinct mod 16 (residue-class map as in az_conv64.h): conflict-free ds_read_b128.
//   * the next board's 2 x C/8 strips arrive by LDS-DMA (global_load_lds_dwordx4, masked to the position cells) in the shadow of
//     the first k-steps; one barrier per board; epilogue on the accumulators (bias enters as the C operand of the first MFMA).
#pragma once
#include "az_conv64.h"

#if defined(__HIPCC__)
#define SP_SCALE 2048.0f
#define SP_INV_SCALE (1.0f / 2048.0f)
#define SP_F16_MAX 65504.0f

struct SpGeo9 {
    static constexpr int S = 9, TB = 1, P2 = 81, PITCH = 10, CELL0 = 11;
    static constexpr int CELLS = 112;  // 11 + 8 * 10 + 8 + 11 = 110 with the tap reach, rounded up to a multiple of 16
    static constexpr int NCT = 6, NCT_REAL = 6;  // no residue class mod 16 holds more than 6 of the 81 cells
    static constexpr int cell_of(int p) { return CELL0 + PITCH * (p / S) + p % S; }
    static constexpr int pos_of_cell(int cell, int in_s, int off) {
        const int k = cell - CELL0;
        if (k < 0 || in_s != S || off != 0) return -1;
        const int yy = k / PITCH, xx = k % PITCH;
        return (yy < S && xx < S) ? yy * S + xx : -1;
    }
};
static __device__ const Cw64Map<SpGeo9> sp_map9 = cw64_make_map<SpGeo9>();
static_assert(cw64_make_map<SpGeo9>().ok && cw64_make_map<SpGeo9>().real_tiles == SpGeo9::NCT_REAL, "column-tile map of the split kernel");

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
__device__ __forceinline__ unsigned sp_pack(_Float16 a, _Float16 b) { return __builtin_bit_cast(unsigned, (sp_f16x2){a, b}); }
__device__ __forceinline__ _Float16 sp_lo16(unsigned v) { return __builtin_bit_cast(sp_f16x2, v)[0]; }
__device__ __forceinline__ _Float16 sp_hi16(unsigned v) { return __builtin_bit_cast(sp_f16x2, v)[1]; }

// fp32 channels-last rows [boards * P2][C] <-> split layout; one thread per (board, chunk, position), positions fastest
// (the 16-byte accesses of the split side are contiguous per chunk strip)
__global__ void __launch_bounds__(256)
k_split_layout(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, long long nchunks, int to_split, int nch, int p2) {
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
#pragma unroll
        for (int e = 0; e < 8; ++e) sp_split(f[e], h[e], l[e]);
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

// the inline-asm MFMAs are opaque to the compiler's hazard recognizer: wait states before VALU reads of their results
__device__ __forceinline__ void sp_settle(c6_f32x4 (&a)[6], c6_f32x4 (&b)[6]) {
    asm volatile("s_nop 15\n\ts_nop 15"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]),
                   "+v"(b[5]));
}

// NCH = input-channel chunks of 8 (16: 128 -> C layer, 8: 64 -> C layer); NCG = cout groups of 64 (C = 64 NCG).
// w: [plane: hi, lo][9 taps][C couts][8 NCH cin] f16 with lo = (w - hi) * 2^11; bias fp32 [C].
template <bool RES, int NCH, int NCG> __global__ void __launch_bounds__(CW_THREADS, 1)
k_conv3x3_sp(const unsigned char* __restrict__ x, const _Float16* __restrict__ w, const float* __restrict__ bias,
             const unsigned char* __restrict__ res, unsigned char* __restrict__ y, int ntiles, int relu) {
    typedef SpGeo9 G;
    constexpr int C = 64 * NCG, CIN = 8 * NCH;
    constexpr int KS = NCH / 4, NSTEP = 9 * KS;             // k-steps (one tap x 32 input channels)
    constexpr int NJ = G::NCT;                               // column tiles: every wave multiplies all of them for its 16 couts
    constexpr int LBLK = G::CELLS * 16, LPLANE = NCH * LBLK, LBUF = 2 * LPLANE;
    constexpr int GBLK = G::P2 * 16, XPLANE = NCH * GBLK, XTILE = 2 * XPLANE;  // input board
    constexpr int YPLANE = (C / 8) * GBLK, YTILE = 2 * YPLANE;                  // output / residual board
    constexpr int NP = (G::CELLS + 63) / 64;                 // DMA pieces of 64 cells per strip (2)
    constexpr int SPW = 2 * NCH / 4, NPIECE = NP * SPW;      // strips (both planes) and pieces per wave
    constexpr int NF = 2 * NSTEP, NF_A = NF < 64 ? NF : 64;  // A fragments: hi bank then lo bank; the first 64 live in AGPRs
    static_assert(NSTEP >= NPIECE, "the next board's pieces ride in the k-steps");
    static_assert(LPLANE + 3 * 4 * LBLK + 22 * 16 < 65536, "fragment addresses are a base + a 16-bit immediate");
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * LBUF];
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
    for (int i = tid; i < 2 * LBUF / 16; i += CW_THREADS) *(cv_u32x4*)(lds + i * 16) = (cv_u32x4){0u, 0u, 0u, 0u};
    CV_BARRIER();  // the zero cells are final before any LDS-DMA piece can land

    // A fragments: fragment f = plane * NSTEP + (tap * KS + ks): lane (cout = 64 cg + 16 wave + l15, cin = 32 ks + 8 kg .. + 8)
    sp_f16x8 wf[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int pl = f / NSTEP, s = f % NSTEP;
        wf[f] = *(const sp_f16x8*)(w + ((size_t)((pl * 9 + s / KS) * C + cg * 64 + wave * 16 + l15)) * CIN + (s % KS) * 32 + kg * 8);
    }
    c6_f32x4 bv;  // bias in the D layout (rows = couts 4 kg + e of the wave's 16): the C operand of the first k-step
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = bias[cg * 64 + wave * 16 + 4 * kg + e];
    const float lo_bound = relu ? 0.0f : -SP_F16_MAX;

    // LDS-DMA plan: a strip is NP pieces of 64 cells; wave q moves strips SPW q .. SPW q + SPW - 1 (strip = plane * NCH + chunk)
    unsigned dsrc[NP];
    unsigned long long dmask[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int p = G::pos_of_cell(64 * i + lane, G::S, 0);
        dsrc[i] = (unsigned)((p < 0 ? 0 : p) * 16);
        dmask[i] = __builtin_amdgcn_ballot_w64(p >= 0 && 64 * i + lane < G::CELLS);
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

    // this lane's column tiles: LDS byte offset of the (-1, -1) neighbour in its own 8-channel group (low 16 bits), position (high 16 bits)
    unsigned lmap[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int idx = j * 16 + l15;
        lmap[j] = (unsigned)((sp_map9.cell[idx] - G::CELL0) * 16 + kg * LBLK) | ((unsigned)sp_map9.pos[idx] << 16);
    }
    sp_f16x8 bb[2][2][NJ];  // B fragments [k-step parity][plane][column tile]
    auto load_step = [&](const unsigned char* const (&bp)[NJ], int s) {
        const int tap = s / KS;
        const int off = ((tap / 3) * G::PITCH + (tap % 3)) * 16 + (s % KS) * (4 * LBLK);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bb[s & 1][0][j] = *(const sp_f16x8*)(bp[j] + off);
#pragma unroll
        for (int j = 0; j < NJ; ++j) bb[s & 1][1][j] = *(const sp_f16x8*)(bp[j] + off + LPLANE);
    };

    if (slot < ntiles) {  // first board: all pieces at once
        const unsigned char* src = x + (size_t)slot * XTILE;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) dma_piece(src, lds0, true, i);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    CV_BARRIER();
    {
        const unsigned char* bp0[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) bp0[j] = lds + (lmap[j] & 0xffffu);
        load_step(bp0, 0);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {  // the compiler's wait for the weight loads belongs in front of the loop (see az_conv.h)
        if (f < NF_A) asm volatile("" : : "a"(wf[f]));
        else asm volatile("" : : "v"(wf[f]));
    }
    asm volatile("" : : "v"(bv), "v"(lmap[0]), "v"(lmap[NJ - 1]), "v"(dsrc[0]), "v"(dsrc[NP - 1]));

    int it = 0;
    for (int tile = slot; tile < ntiles; tile += nslot, ++it) {
        const int buf = it & 1;
        const unsigned char* Xs = lds + buf * LBUF;
        const unsigned char* Xn = lds + (buf ^ 1) * LBUF;
        const bool has_next = tile + nslot < ntiles;
        const unsigned char* nsrc = x + (size_t)(has_next ? tile + nslot : tile) * XTILE;
        const unsigned ndst = lds0 + (unsigned)((buf ^ 1) * LBUF);
        // this wave's 16 couts = chunks 8 cg + 2 wave + {0, 1}; lane group kg holds couts 4 kg .. + 4 = chunk kg / 2, half kg % 2
        const size_t yo = (size_t)tile * YTILE + (size_t)(cg * 8 + wave * 2 + (kg >> 1)) * GBLK + (size_t)((kg & 1) * 8);
        const unsigned char* rbase = RES ? res + yo : nullptr;
        unsigned char* ybase = y + yo;
        const unsigned char* bp[NJ];
        cv_u32x2 rr[NJ][2];
        c6_f32x4 accm[NJ], accc[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bp[j] = Xs + (lmap[j] & 0xffffu);
            accc[j] = (c6_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            if (RES) {
                const unsigned gq = (lmap[j] >> 16) * 16u;
                rr[j][0] = *(const cv_u32x2*)(rbase + gq);
                rr[j][1] = *(const cv_u32x2*)(rbase + YPLANE + gq);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NSTEP; ++t) {  // the fragments of step 0 are already in flight (issued before the previous epilogue)
            if (t + 1 < NSTEP) load_step(bp, t + 1);
            // three products, each over the 6 column tiles (6 independent accumulators between two uses of one)
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (t == 0) sp_mfma_ac(accm[j], wf[0], bb[0][0][j], bv);
                else sp_mfma_a(accm[j], wf[t], bb[t & 1][0][j]);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) sp_mfma_a(accc[j], wf[t], bb[t & 1][1][j]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (NSTEP + t < NF_A) sp_mfma_a(accc[j], wf[NSTEP + t], bb[t & 1][0][j]);
                else sp_mfma_v(accc[j], wf[NSTEP + t], bb[t & 1][0][j]);
            }
            if (t < NPIECE) dma_piece(nsrc, ndst, has_next, t);  // the next board's pieces ride in the shadow of the first k-steps
            __builtin_amdgcn_sched_barrier(0);
        }
        // everything this wave has in flight is old (pieces issued >= NSTEP - NPIECE k-steps ago, the previous board's stores): after
        // the barrier every wave's pieces of the next board have landed and this buffer may be overwritten
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CV_BARRIER();
        {
            const unsigned char* bpn[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bpn[j] = Xn + (lmap[j] & 0xffffu);
            load_step(bpn, 0);  // the next board's first fragments fly while the epilogue runs
        }
        __builtin_amdgcn_sched_barrier(0);
        sp_settle(accm, accc);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const unsigned gq = (lmap[j] >> 16) * 16u;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(accc[j][e], SP_INV_SCALE, accm[j][e]);
            if (RES) {
                const cv_u32x2 rh = rr[j][0], rl = rr[j][1];
                v[0] += sp_join(sp_lo16(rh.x), sp_lo16(rl.x));
                v[1] += sp_join(sp_hi16(rh.x), sp_hi16(rl.x));
                v[2] += sp_join(sp_lo16(rh.y), sp_lo16(rl.y));
                v[3] += sp_join(sp_hi16(rh.y), sp_hi16(rl.y));
            }
            _Float16 h[4], l[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) sp_split(fmaxf(v[e], lo_bound), h[e], l[e]);
            *(cv_u32x2*)(ybase + gq) = (cv_u32x2){sp_pack(h[0], h[1]), sp_pack(h[2], h[3])};
            *(cv_u32x2*)(ybase + YPLANE + gq) = (cv_u32x2){sp_pack(l[0], l[1]), sp_pack(l[2], l[3])};
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
#endif  // __HIPCC__
