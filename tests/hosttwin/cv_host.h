// Plain reference loops of the convolution ABI entries -- TEST INFRASTRUCTURE (host twin build only: lets the CPU tier exercise
// azsp_conv3x3_tiled / azsp_stem_tiled through the C ABI on tiny inputs).
#pragma once
#include <math.h>
#include <stddef.h>
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
static inline unsigned short sp_h_from_f32(float f) {  // fp32 -> f16, round to nearest even, subnormals kept
    union { unsigned u; float f; } v;
    v.f = f;
    const unsigned sign = (v.u >> 16) & 0x8000u, x = v.u & 0x7fffffffu;
    if (x >= 0x7f800000u) return (unsigned short)(sign | 0x7c00u | (x > 0x7f800000u ? 0x200u : 0u));
    if (x >= 0x477ff000u) return (unsigned short)(sign | 0x7c00u);  // >= 65520 rounds to infinity
    if (x < 0x38800000u) {                                           // below 2^-14: a multiple of 2^-24
        v.u = x;
        return (unsigned short)(sign | (unsigned)nearbyintf(v.f * 16777216.0f));
    }
    unsigned h = (((x >> 23) - 112u) << 10) | ((x & 0x7fffffu) >> 13);
    const unsigned rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
    return (unsigned short)(sign | h);
}
static inline float sp_h_to_f32(unsigned short h) {
    const int e = (h >> 10) & 31, m = h & 0x3ff;
    float v;
    if (e == 0) v = ldexpf((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf((float)(m | 0x400), e - 25);
    return (h & 0x8000) ? -v : v;
}
// element <-> float in the evaluator's activation format: bf16 (f16 = 0) or f16 (f16 = 1, clamped to its finite range like the kernels)
static inline float cv_h_in(unsigned short h, int f16) { return f16 ? sp_h_to_f32(h) : cv_h_bf16(h); }
static inline unsigned short cv_h_out(float v, int f16) {
    if (!f16) return cv_h_to_bf16(v);
    return sp_h_from_f32(v > 65504.0f ? 65504.0f : v);
}
static inline void cv_host_conv3x3_io(const unsigned short* x, const unsigned short* w, const float* bias, const unsigned short* res,
                                      unsigned short* y, int nboards, int S, int Cin, int C, int relu, int f16 = 0);
static inline void cv_host_conv3x3(const unsigned short* x, const unsigned short* w, const float* bias, const unsigned short* res,
                                   unsigned short* y, int nboards, int S, int C, int relu) {
    cv_host_conv3x3_io(x, w, bias, res, y, nboards, S, C, C, relu);
}
// x: [boards][S][S][Cin], w: [9][C][Cin], y / res: [boards][S][S][C]
static inline void cv_host_conv3x3_io(const unsigned short* x, const unsigned short* w, const float* bias, const unsigned short* res,
                                      unsigned short* y, int nboards, int S, int Cin, int C, int relu, int f16) {
    for (int b = 0; b < nboards; ++b)
        for (int yy = 0; yy < S; ++yy)
            for (int xx = 0; xx < S; ++xx)
                for (int co = 0; co < C; ++co) {
                    float acc = 0.0f;
                    for (int tap = 0; tap < 9; ++tap) {
                        const int sy = yy + tap / 3 - 1, sx = xx + tap % 3 - 1;
                        if (sy < 0 || sx < 0 || sy >= S || sx >= S) continue;
                        const unsigned short* xi = x + ((size_t)(b * S + sy) * S + sx) * Cin;
                        const unsigned short* wi = w + ((size_t)tap * C + co) * Cin;
                        for (int ci = 0; ci < Cin; ++ci) acc += cv_h_in(xi[ci], f16) * cv_h_in(wi[ci], f16);
                    }
                    const size_t o = ((size_t)(b * S + yy) * S + xx) * C + co;
                    float v = acc + bias[co] + (res ? cv_h_in(res[o], f16) : 0.0f);
                    if (relu && v < 0.0f) v = 0.0f;
                    y[o] = cv_h_out(v, f16);
                }
}
