// Plain reference loops of the convolution ABI entries -- TEST INFRASTRUCTURE (host twin build only: lets the CPU tier exercise
// azsp_conv3x3_tiled / azsp_stem_tiled through the C ABI on tiny inputs).
#pragma once
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
static inline void cv_host_conv3x3_io(const unsigned short* x, const unsigned short* w, const float* bias, const unsigned short* res,
                                      unsigned short* y, int nboards, int S, int Cin, int C, int relu);
static inline void cv_host_conv3x3(const unsigned short* x, const unsigned short* w, const float* bias, const unsigned short* res,
                                   unsigned short* y, int nboards, int S, int C, int relu) {
    cv_host_conv3x3_io(x, w, bias, res, y, nboards, S, C, C, relu);
}
// x: [boards][S][S][Cin], w: [9][C][Cin], y / res: [boards][S][S][C]
static inline void cv_host_conv3x3_io(const unsigned short* x, const unsigned short* w, const float* bias, const unsigned short* res,
                                      unsigned short* y, int nboards, int S, int Cin, int C, int relu) {
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
                        for (int ci = 0; ci < Cin; ++ci) acc += cv_h_bf16(xi[ci]) * cv_h_bf16(wi[ci]);
                    }
                    const size_t o = ((size_t)(b * S + yy) * S + xx) * C + co;
                    float v = acc + bias[co] + (res ? cv_h_bf16(res[o]) : 0.0f);
                    if (relu && v < 0.0f) v = 0.0f;
                    y[o] = cv_h_to_bf16(v);
                }
}
