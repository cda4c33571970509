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
#define CV_P2 (CV_S * CV_S)                   // 81
#define CV_PS (CV_S + 2)                      // 11
#define CV_PP (CV_PS * CV_PS)                 // 121
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
#define CV_THREADS 512
#define CV_XCH ((CV_NPOS * 16 + CV_THREADS - 1) / CV_THREADS)  // 16-B activation chunks staged per thread (8)
#define CV_WCH (CV_C * 16 / CV_THREADS)                        // 16-B weight chunks per thread per tap (4)

// 8 waves = 2 (cout halves) x 4 (column quarters); two waves share each SIMD, so one wave's LDS fragment reads hide
// behind the other's MFMAs.  Next tile's activations and this tile's residual are prefetched into registers while
// the matrix cores run, so HBM latency never sits on the critical path.
__global__ void __launch_bounds__(CV_THREADS, 2)
k_conv3x3_c128_s9(const unsigned short* __restrict__ x, const unsigned short* __restrict__ w, const float* __restrict__ bias,
                  const unsigned short* __restrict__ res, unsigned short* __restrict__ y, int nboards, int relu) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[CV_XS_BYTES + CV_WS_BYTES];
    unsigned char* Xs = lds;
    unsigned char* Ws = lds + CV_XS_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int mh = wave & 1, nq = wave >> 1;  // cout half, column quarter

    // the zero row (never overwritten)
    if (tid < 16) *(cv_u32x4*)(Xs + CV_ZROW * CV_ROWB + tid * 16) = (cv_u32x4){0u, 0u, 0u, 0u};

    // this lane's two output columns (positions inside the tile) and their board coordinates
    int pos[2], py[2], px[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int p = nq * 64 + nt * 32 + l31;
        pos[nt] = p;
        const int b = p / CV_P2, q = p - b * CV_P2;
        py[nt] = p < CV_NPOS ? q / CV_S : -100;  // padding columns: every tap is "off board" -> zero row
        px[nt] = q - (q / CV_S) * CV_S;
    }
    const int ntiles = (nboards + CV_TB - 1) / CV_TB;
    cv_u32x4 xreg[CV_XCH];
    {   // first tile's activations
        const int tile = blockIdx.x;
        const int rows = tile < ntiles ? min(CV_TB, nboards - tile * CV_TB) * CV_P2 : 0;
        const size_t gbase = (size_t)tile * CV_NPOS * CV_C;
#pragma unroll
        for (int i = 0; i < CV_XCH; ++i) {
            xreg[i] = (cv_u32x4){0u, 0u, 0u, 0u};
            if (((tid + CV_THREADS * i) >> 4) < rows) xreg[i] = *(const cv_u32x4*)(x + gbase + (size_t)(tid + CV_THREADS * i) * 8);
        }
    }
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int rows = min(CV_TB, nboards - tile * CV_TB) * CV_P2;  // valid positions in this tile
        const size_t gbase = (size_t)tile * CV_NPOS * CV_C;          // element offset of the tile in x / y / res
        CV_BARRIER();  // every wave is done reading the previous tile's Xs / Ws

        // ---- activations: prefetched registers -> LDS (zero-haloed 11x11 boards, swizzled chunks) ----
#pragma unroll
        for (int i = 0; i < CV_XCH; ++i) {
            const int idx = tid + CV_THREADS * i, r = idx >> 4, c = idx & 15;
            if (r < CV_NPOS) *(cv_u32x4*)(Xs + cv_swz((unsigned)r, (unsigned)c)) = xreg[i];
        }

        cv_f32x16 acc[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[mt][nt][e] = 0.0f;

        // weights of tap 0 -> registers (4 x 16 B per thread: 128 couts x 16 chunks)
        cv_u32x4 wreg[CV_WCH];
#pragma unroll
        for (int i = 0; i < CV_WCH; ++i) wreg[i] = *(const cv_u32x4*)(w + (size_t)(tid + CV_THREADS * i) * 8);
        cv_u32x2 rres[2][2][4];

#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            unsigned char* Wb = Ws + (tap & 1) * CV_WBUF;
            // registers -> LDS ring slot (last read two taps ago, before the previous barrier)
#pragma unroll
            for (int i = 0; i < CV_WCH; ++i) {
                const int idx = tid + CV_THREADS * i, r = idx >> 4, c = idx & 15;
                *(cv_u32x4*)(Wb + cv_swz(r, c)) = wreg[i];
            }
            if (tap < 8) {
#pragma unroll
                for (int i = 0; i < CV_WCH; ++i)
                    wreg[i] = *(const cv_u32x4*)(w + (size_t)(tap + 1) * CV_C * CV_C + (size_t)(tid + CV_THREADS * i) * 8);
            }
            if (tap == 2) {  // next tile's activations: in flight under the remaining taps
                const int nxt = tile + gridDim.x;
                const int nrows = nxt < ntiles ? min(CV_TB, nboards - nxt * CV_TB) * CV_P2 : 0;
                const size_t nbase = (size_t)nxt * CV_NPOS * CV_C;
#pragma unroll
                for (int i = 0; i < CV_XCH; ++i) {
                    xreg[i] = (cv_u32x4){0u, 0u, 0u, 0u};
                    if (((tid + CV_THREADS * i) >> 4) < nrows) xreg[i] = *(const cv_u32x4*)(x + nbase + (size_t)(tid + CV_THREADS * i) * 8);
                }
            }
            if (tap == 5 && res) {  // this tile's residual, already in the accumulator layout: 4 consecutive couts = 8 B per slot
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int rq = 0; rq < 4; ++rq) {
                            const unsigned cout0 = (unsigned)((2 * mh + mt) * 32 + 8 * rq + 4 * hi);
                            rres[nt][mt][rq] = (cv_u32x2){0u, 0u};
                            if (pos[nt] < rows) rres[nt][mt][rq] = *(const cv_u32x2*)(res + gbase + (size_t)pos[nt] * CV_C + cout0);
                        }
            }
            CV_BARRIER();
            const int dy = tap / 3 - 1, dx = tap % 3 - 1, shift = dy * CV_S + dx;
            const bool in0 = (unsigned)(py[0] + dy) < (unsigned)CV_S && (unsigned)(px[0] + dx) < (unsigned)CV_S;
            const bool in1 = (unsigned)(py[1] + dy) < (unsigned)CV_S && (unsigned)(px[1] + dx) < (unsigned)CV_S;
            const unsigned pr0 = in0 ? (unsigned)(pos[0] + shift) : (unsigned)CV_ZROW;
            const unsigned pr1 = in1 ? (unsigned)(pos[1] + shift) : (unsigned)CV_ZROW;
            // fragment double buffer: the LDS reads of k-step ks+1 are in flight while the MFMAs of k-step ks issue
            // (a ring of depth 3 was measured 5 % slower: the loop is bound by LDS bandwidth shared by 8 waves, not by latency)
            const unsigned arow0 = (unsigned)((2 * mh) * 32 + l31), arow1 = arow0 + 32u;
            cv_bf16x8 a0[2], b0[2], a1[2], b1[2];
            a0[0] = *(const cv_bf16x8*)(Wb + cv_swz(arow0, (unsigned)hi));
            a0[1] = *(const cv_bf16x8*)(Wb + cv_swz(arow1, (unsigned)hi));
            b0[0] = *(const cv_bf16x8*)(Xs + cv_swz(pr0, (unsigned)hi));
            b0[1] = *(const cv_bf16x8*)(Xs + cv_swz(pr1, (unsigned)hi));
#pragma unroll
            for (int kp = 0; kp < 4; ++kp) {
                const unsigned ch1 = (unsigned)((2 * kp + 1) * 2 + hi), ch2 = (unsigned)((2 * kp + 2) * 2 + hi);
                a1[0] = *(const cv_bf16x8*)(Wb + cv_swz(arow0, ch1));
                a1[1] = *(const cv_bf16x8*)(Wb + cv_swz(arow1, ch1));
                b1[0] = *(const cv_bf16x8*)(Xs + cv_swz(pr0, ch1));
                b1[1] = *(const cv_bf16x8*)(Xs + cv_swz(pr1, ch1));
                __builtin_amdgcn_sched_barrier(0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b0[0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[0], b0[1], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], b0[0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[1], b0[1], acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kp < 3) {
                    a0[0] = *(const cv_bf16x8*)(Wb + cv_swz(arow0, ch2));
                    a0[1] = *(const cv_bf16x8*)(Wb + cv_swz(arow1, ch2));
                    b0[0] = *(const cv_bf16x8*)(Xs + cv_swz(pr0, ch2));
                    b0[1] = *(const cv_bf16x8*)(Xs + cv_swz(pr1, ch2));
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b1[0], acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[0], b1[1], acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1], b1[0], acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1[1], b1[1], acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- epilogue straight from the accumulators: D layout of mfma_f32_32x32x16 gives each lane, for its column
        //      (position), 4 consecutive couts per register quad = one 8-byte slot of the channels-last row.  No LDS round
        //      trip and no extra barrier; the 16-B pieces of a row are merged into full lines by the L2 before they reach HBM.
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int p = pos[nt];
            if (p < rows) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        const unsigned cout0 = (unsigned)((2 * mh + mt) * 32 + 8 * rq + 4 * hi);
                        const float4 bv = *(const float4*)(bias + cout0);  // L1-resident, 512 B in total
                        float v0 = acc[mt][nt][rq * 4 + 0] + bv.x, v1 = acc[mt][nt][rq * 4 + 1] + bv.y;
                        float v2 = acc[mt][nt][rq * 4 + 2] + bv.z, v3 = acc[mt][nt][rq * 4 + 3] + bv.w;
                        if (res) {
                            const cv_u32x2 rr = rres[nt][mt][rq];
                            v0 += cv_bf16_lo(rr.x); v1 += cv_bf16_hi(rr.x); v2 += cv_bf16_lo(rr.y); v3 += cv_bf16_hi(rr.y);
                        }
                        if (relu) { v0 = fmaxf(v0, 0.0f); v1 = fmaxf(v1, 0.0f); v2 = fmaxf(v2, 0.0f); v3 = fmaxf(v3, 0.0f); }
                        *(cv_u32x2*)(y + gbase + (size_t)p * CV_C + cout0) = (cv_u32x2){cv_pack_bf16(v0, v1), cv_pack_bf16(v2, v3)};
                    }
            }
        }
    }
}
#endif  // __HIPCC__

// Plain reference loop (host twin build only: lets the CPU tier exercise the ABI entry on tiny inputs).
static inline float cv_h_bf16(unsigned short h) {
    union { unsigned u; float f; } v;
    v.u = (unsigned)h << 16;
    return v.f;
}
static inline unsigned short cv_h_to_bf16(float f) {
    union { unsigned u; float f; } v;
    v.f = f;
    return (unsigned short)((v.u + 0x7fffu + ((v.u >> 16) & 1u)) >> 16);
}
static inline void cv_host_conv3x3(const unsigned short* x, const unsigned short* w, const float* bias, const unsigned short* res,
                                   unsigned short* y, int nboards, int S, int C, int relu) {
    for (int b = 0; b < nboards; ++b)
        for (int yy = 0; yy < S; ++yy)
            for (int xx = 0; xx < S; ++xx)
                for (int co = 0; co < C; ++co) {
                    float acc = 0.0f;
                    for (int tap = 0; tap < 9; ++tap) {
                        const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
                        if (sy < 0 || sx < 0 || sy >= S || sx >= S) continue;
                        const unsigned short* xi = x + ((size_t)(b * S + sy) * S + sx) * C;
                        const unsigned short* wi = w + ((size_t)tap * C + co) * C;
                        for (int ci = 0; ci < C; ++ci) acc += cv_h_bf16(xi[ci]) * cv_h_bf16(wi[ci]);
                    }
                    const size_t o = ((size_t)(b * S + yy) * S + xx) * C + co;
                    float v = acc + bias[co] + (res ? cv_h_bf16(res[o]) : 0.0f);
                    if (relu && v < 0.0f) v = 0.0f;
                    y[o] = cv_h_to_bf16(v);
                }
}
