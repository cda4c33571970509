// az_wave.h -- the 64-lane wavefront as a programming model.
//
// Every engine routine is written once, against a `Wave` policy:
//   * code outside a `Wv::lanes(...)` section is WAVE-UNIFORM: on gfx950 all 64 lanes run it
//     redundantly on identical values (the compiler keeps such values in SGPRs / scalar ALU),
//   * `Wv::lanes(f)` runs f(lane) on every lane (SPMD section; lanes only write their own data),
//   * `Wv::ballot / any / argmax_first / sum_*` are the cross-lane primitives.
// `WaveDev` maps these to CDNA4 wave64 hardware (v_cmp -> 64-bit ballot masks, DPP/bpermute
// reductions).  (tests/hosttwin/az_host_policies.h holds a `WaveHost` that replays the same sections with a
// 64-iteration loop so that the *identical* engine source can be unit-tested without a GPU; it is not part of the product.)
#pragma once
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;
typedef long long i64;

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define AZ_HD __host__ __device__ __forceinline__
#define AZ_D __device__ __forceinline__
#else
#define AZ_HD inline
#define AZ_D inline
#endif

#define AZ_WAVE 64

#if defined(__HIPCC__)
struct WaveDev {
    static AZ_D int lane() { return (int)(threadIdx.x & 63u); }
    static AZ_D bool first() { return lane() == 0; }
    template <class F> static AZ_D void lanes(F&& f) { f(lane()); }
    template <class F> static AZ_D u64 ballot(F&& f) { return __ballot(f(lane()) ? 1 : 0); }
    // Orders this wave's earlier LDS/global accesses before later ones (cross-lane hand-over inside ONE
    // wave).  A wave's memory instructions are issued and serviced in order, so a compiler-level
    // barrier is all that is needed; no cross-wave traffic exists in this engine (one wave == one game).
    static AZ_D void sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    // Max over lanes of a double; every lane gets the result.
    static AZ_D double max_f64(double v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            double w = __shfl_xor(v, o, 64);
            v = w > v ? w : v;
        }
        return v;
    }
    static AZ_D int min_i32(int v) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            int w = __shfl_xor(v, o, 64);
            v = w < v ? w : v;
        }
        return v;
    }
    // ---- wave-wide reductions on the DPP network (VALU latency) instead of ds_bpermute butterflies (an LDS round trip per
    // step): quad swaps, half-row / row mirrors, then row_bcast15 / row_bcast31 carry the row results to lane 63.  The select
    // kernel's arg-max spent 12 dependent bpermute steps per tree level in the first version (tools/sel_abl.sh: 30 % of the launch).
    template <int CTRL, int ROW_MASK> static AZ_D u32 dpp(u32 own, u32 v) {
        return (u32)__builtin_amdgcn_update_dpp((int)own, (int)v, CTRL, ROW_MASK, 0xF, false);
    }
    template <int CTRL, int ROW_MASK> static AZ_D u64 max_step_u64(u64 v) {
        const u32 lo = (u32)v, hi = (u32)(v >> 32);
        const u64 o = ((u64)dpp<CTRL, ROW_MASK>(hi, hi) << 32) | dpp<CTRL, ROW_MASK>(lo, lo);
        return o > v ? o : v;
    }
    static AZ_D u64 max_u64(u64 v) {  // every lane contributes; the result is wave-uniform
        v = max_step_u64<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
        v = max_step_u64<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
        v = max_step_u64<0x141, 0xF>(v);  // row_half_mirror
        v = max_step_u64<0x140, 0xF>(v);  // row_mirror: every lane of a 16-lane row holds the row maximum
        v = max_step_u64<0x142, 0xA>(v);  // row_bcast15 into rows 1 and 3
        v = max_step_u64<0x143, 0xC>(v);  // row_bcast31 into rows 2 and 3: lane 63 holds the wave maximum
        const u32 lo = (u32)__builtin_amdgcn_readlane((int)(u32)v, 63), hi = (u32)__builtin_amdgcn_readlane((int)(u32)(v >> 32), 63);
        return ((u64)hi << 32) | lo;
    }
    template <int CTRL, int ROW_MASK> static AZ_D u32 min_step_u32(u32 v) {
        const u32 o = dpp<CTRL, ROW_MASK>(v, v);
        return o < v ? o : v;
    }
    static AZ_D u32 min_u32(u32 v) {
        v = min_step_u32<0xB1, 0xF>(v);
        v = min_step_u32<0x4E, 0xF>(v);
        v = min_step_u32<0x141, 0xF>(v);
        v = min_step_u32<0x140, 0xF>(v);
        v = min_step_u32<0x142, 0xA>(v);
        v = min_step_u32<0x143, 0xC>(v);
        return (u32)__builtin_amdgcn_readlane((int)v, 63);
    }
    // IEEE double -> unsigned key with the same order (no NaNs reach the search: scores are finite)
    static AZ_D u64 order_key(double x) {
        u64 b;
        __builtin_memcpy(&b, &x, sizeof b);
        return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    }
    // f(lane, score&, idx&): the lane's best candidate (idx < 0: none).  Returns the index with the
    // largest score, lowest index on ties (np.argmax semantics, mcts_v2.py:178).
    template <class F> static AZ_D int argmax_first(F&& f) {
        double s = -1.0e300;
        int idx = -1;
        f(lane(), s, idx);
        const u64 key = idx >= 0 ? order_key(s) : 0ull;  // 0 sorts below every finite score
        const u64 m = max_u64(key);
        const u32 cand = (idx >= 0 && key == m) ? (u32)idx : 0x7fffffffu;
        return (int)min_u32(cand);
    }
    static AZ_D int bcast0(int v) { return __builtin_amdgcn_readfirstlane(v); }
    // Tell the compiler that `v` is wave-uniform (it is, by construction: loaded through a uniform address or
    // produced by a reduction): the value moves to SGPRs and everything computed from it runs on the scalar unit
    // instead of being replicated in 64 VGPR lanes.  Memory loads are "divergent" to LLVM unless proven otherwise.
    template <class T> static AZ_D T uni(T v) {
        static_assert(sizeof(T) % 4 == 0 && sizeof(T) <= 16, "uni(): 4/8/16-byte values");
        int w[sizeof(T) / 4];
        __builtin_memcpy(w, &v, sizeof(T));
#pragma unroll
        for (unsigned i = 0; i < sizeof(T) / 4; ++i) w[i] = __builtin_amdgcn_readfirstlane(w[i]);
        __builtin_memcpy(&v, w, sizeof(T));
        return v;
    }
    // Sum over lanes of a double (butterfly: every lane ends with the same value).
    template <class F> static AZ_D double sum_f64(F&& f) {
        double v = f(lane());
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    }
    template <class F> static AZ_D int sum_i32(F&& f) {
        int v = f(lane());
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        return v;
    }
};
#endif

// The plain-C++ replay of these primitives (`WaveHost`, a 64-iteration loop per section) that lets the CPU test tier run the
// identical engine source lives in tests/hosttwin/az_host_policies.h; that build defines AZ_HOST_TWIN_POLICIES to pull it in here.
#if defined(AZ_HOST_TWIN_POLICIES)
#include AZ_HOST_TWIN_POLICIES
#endif
