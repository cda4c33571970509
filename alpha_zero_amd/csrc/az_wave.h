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
    // f(lane, score&, idx&): the lane's best candidate (idx < 0: none).  Returns the index with the
    // largest score, lowest index on ties (np.argmax semantics, mcts_v2.py:178).
    template <class F> static AZ_D int argmax_first(F&& f) {
        double s = -1.0e300;
        int idx = -1;
        f(lane(), s, idx);
        if (idx < 0) s = -1.0e300;
        double m = max_f64(s);
        int cand = (idx >= 0 && s == m) ? idx : 0x7fffffff;
        return __builtin_amdgcn_readfirstlane(min_i32(cand));
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
