// Host twin of libazsp.so -- TEST INFRASTRUCTURE ONLY.
// Compiles the *same* engine source (alpha_zero_amd/csrc/az_*.h, azsp_impl.h) with the WaveHost
// policy: every "kernel" is a loop over games, every wave section a loop over 64 lanes.  It lets the
// CPU-only test tier exercise tree search, rules and the C ABI logic without a GPU.  The product
// package never loads this library (alpha_zero_amd/_lib.py only accepts libazsp.so + a HIP device).
#include <math.h>
#include <vector>
#include <stdlib.h>

#define AZ_HOST_TWIN_POLICIES "../../tests/hosttwin/az_host_policies.h"
#define AZ_GEOM_WAVE WaveHost
#include "../../alpha_zero_amd/csrc/azsp_impl.h"
#include "cv_host.h"

namespace azb {
void* alloc(size_t n) { return calloc(1, n); }
void release(void* p) { free(p); }
int h2d(void* d, const void* s, size_t n, void*) { memcpy(d, s, n); return 0; }
int d2h(void* d, const void* s, size_t n, void*) { memcpy(d, s, n); return 0; }
int zero(void* d, size_t n, void*) { memset(d, 0, n); return 0; }
int sync(void*) { return 0; }
void* host_alloc(size_t n) { return calloc(1, n); }
void host_release(void* p) { free(p); }
int set_device(int) { return 0; }
const char* backend_error() { return "host twin"; }
template <int N, int GAME, class Op> int launch(const AzCfg& c, const AzMem& m, const Op& op, void*, int g0, int g1) {
    for (int g = g0; g < g1; ++g) {
        typename Engine<WaveHost, N, GAME>::SC sc;
        Engine<WaveHost, N, GAME> e(c, m, g, sc);
        op(e);
    }
    return 0;
}
int launch_dihedral(const DihedralArgs& a, long long total, void*) {
    for (long long t = 0; t < total; ++t) az_dihedral_elem(a, t);
    return 0;
}
int launch_replay_gather(const ReplayGatherArgs& a, long long total, void*) {
    for (long long t = 0; t < total; ++t) az_replay_gather_elem(a, t);
    return 0;
}
int launch_harvest_scan(const int* len, int* ofs, int* counts, int n2, int rot, int cap, int max_games, void*) {
    int best[2] = {0, 0};
    az_harvest_scan_range(len, ofs, n2, rot, cap, max_games, 0, n2, 0, 0, best);
    counts[0] = best[0];
    counts[1] = best[1];
    return 0;
}
int launch_bias_act(const BiasActArgs& a, void*) {
    for (long long i = 0; i < a.nvec; ++i) az_bias_act_vec(a, i);
    return 0;
}
// tiled layout on the host: [tile][C/8][3*S*S][8] <-> channels-last rows, then the plain loop
static void host_tile_layout(const unsigned short* src, unsigned short* dst, long long boards, int S, int C, int to_tiled) {
    const long long rows = boards * S * S, trows = (long long)cv_tile_boards(S) * S * S, nch = C / 8;
    for (long long r = 0; r < rows; ++r)
        for (long long c = 0; c < nch; ++c) {
            const long long tile = r / trows, p = r % trows;
            const size_t t = ((size_t)(tile * nch + c) * trows + p) * 8, n = ((size_t)r * nch + c) * 8;
            for (int e = 0; e < 8; ++e) {
                if (to_tiled) dst[t + e] = src[n + e];
                else dst[n + e] = src[t + e];
            }
        }
}
int launch_tile_layout(const void* src, void* dst, long long boards, int S, int C, int to_tiled, void*) {
    host_tile_layout((const unsigned short*)src, (unsigned short*)dst, boards, S, C, to_tiled);
    return 0;
}
int launch_conv3x3_tiled(const void* x, const void* w, const float* bias, const void* res, void* y, long long boards, int S, int C,
                         int relu, void*, int f16) {
    if (f16 && (S != 9 || C != 128)) return 1;
    const size_t n = (size_t)boards * S * S * C;
    std::vector<unsigned short> xn(n), rn(res ? n : 0), yn(n);
    host_tile_layout((const unsigned short*)x, xn.data(), boards, S, C, 0);
    if (res) host_tile_layout((const unsigned short*)res, rn.data(), boards, S, C, 0);
    cv_host_conv3x3_io(xn.data(), (const unsigned short*)w, bias, res ? rn.data() : nullptr, yn.data(), (int)boards, S, C, C, relu, f16);
    host_tile_layout(yn.data(), (unsigned short*)y, boards, S, C, 1);
    return 0;
}
int launch_resblock_tiled(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, long long boards, int S, int C,
                          void*) {
    // plain loops: y = relu(conv(relu(conv(x, w1) + b1), w2) + b2 + x), the intermediate rounded to bf16 like the kernel's LDS image
    const size_t n = (size_t)boards * S * S * C;
    std::vector<unsigned short> xn(n), mn(n), yn(n);
    host_tile_layout((const unsigned short*)x, xn.data(), boards, S, C, 0);
    cv_host_conv3x3(xn.data(), (const unsigned short*)w1, b1, nullptr, mn.data(), (int)boards, S, C, 1);
    cv_host_conv3x3(mn.data(), (const unsigned short*)w2, b2, xn.data(), yn.data(), (int)boards, S, C, 1);
    host_tile_layout(yn.data(), (unsigned short*)y, boards, S, C, 1);
    return 0;
}
// ---- split-precision tower (azsp_split_layout / azsp_conv3x3_split): plain loops on the same hi / lo f16 arithmetic ----
static unsigned g_sp_range_host[2] = {0u, 0u};  // the device library's default range record (azsp_split_range_status), per split value here
static unsigned* g_sp_range_cur = g_sp_range_host;  // the record of the running call (the caller's `range_rec`, or the default)
struct SpRangeScope {
    explicit SpRangeScope(unsigned* r) { g_sp_range_cur = r ? r : g_sp_range_host; }
    ~SpRangeScope() { g_sp_range_cur = g_sp_range_host; }
};
// range record of one value (a NaN is an event too, recorded as +inf: sp_range_abs in az_conv_sp.h)
static inline void sp_h_note(float v) {
    if (!(fabsf(v) <= 65504.0f)) {
        unsigned bits;
        const float a = v != v ? INFINITY : fabsf(v);
        memcpy(&bits, &a, 4);
        g_sp_range_cur[0] += 1u;
        if (bits > g_sp_range_cur[1]) g_sp_range_cur[1] = bits;
    }
}
static inline void sp_h_split_quiet(float v, unsigned short& h, unsigned short& l) {  // clamp + split, no record
    v = fminf(fmaxf(v, -65504.0f), 65504.0f);
    h = sp_h_from_f32(v);
    l = sp_h_from_f32((v - sp_h_to_f32(h)) * 2048.0f);
}
static inline void sp_h_split(float v, unsigned short& h, unsigned short& l) {
    sp_h_note(v);
    sp_h_split_quiet(v, h, l);
}
static inline float sp_h_join(unsigned short h, unsigned short l) { return fmaf(sp_h_to_f32(l), 1.0f / 2048.0f, sp_h_to_f32(h)); }
// split layout [board][plane][C/8][P2][8] f16 <-> channels-last fp32 rows
int launch_split_layout(const void* src, void* dst, long long boards, int S, int C, int to_split, void*, unsigned* range) {
    SpRangeScope scope(range);
    const int P2 = S * S, nch = C / 8;
    const size_t plane = (size_t)nch * P2 * 8;
    for (long long b = 0; b < boards; ++b)
        for (int c = 0; c < nch; ++c)
            for (int p = 0; p < P2; ++p)
                for (int e = 0; e < 8; ++e) {
                    const size_t so = (size_t)b * 2 * plane + ((size_t)c * P2 + p) * 8 + e, fo = ((size_t)b * P2 + p) * C + c * 8 + e;
                    if (to_split) sp_h_split(((const float*)src)[fo], ((unsigned short*)dst)[so], ((unsigned short*)dst)[so + plane]);
                    else ((float*)dst)[fo] = sp_h_join(((const unsigned short*)src)[so], ((const unsigned short*)src)[so + plane]);
                }
    return 0;
}
// x: split layout with Cin channels, w: [2][9][C][Cin] f16, y / res: split layout with C channels
static int host_conv_split(const void* x, const void* w, const float* bias, const void* res, void* y, long long boards, int S, int Cin, int C,
                           int relu) {
    const int P2 = S * S;
    const size_t xplane = (size_t)(Cin / 8) * P2 * 8, yplane = (size_t)(C / 8) * P2 * 8, wplane = (size_t)9 * C * Cin;
    const unsigned short *xs = (const unsigned short*)x, *ws = (const unsigned short*)w, *rs = (const unsigned short*)res;
    unsigned short* ys = (unsigned short*)y;
    auto at = [&](int ch, int p) { return ((size_t)(ch / 8) * P2 + p) * 8 + ch % 8; };
    std::vector<float> wh(wplane), wl(wplane);
    for (size_t i = 0; i < wplane; ++i) wh[i] = sp_h_to_f32(ws[i]), wl[i] = sp_h_to_f32(ws[wplane + i]);
    std::vector<float> xh((size_t)P2 * Cin), xl((size_t)P2 * Cin);
    for (long long b = 0; b < boards; ++b) {
        const size_t xo = (size_t)b * 2 * xplane, yo = (size_t)b * 2 * yplane;
        for (int p = 0; p < P2; ++p)
            for (int ci = 0; ci < Cin; ++ci)
                xh[(size_t)p * Cin + ci] = sp_h_to_f32(xs[xo + at(ci, p)]), xl[(size_t)p * Cin + ci] = sp_h_to_f32(xs[xo + xplane + at(ci, p)]);
        for (int p = 0; p < P2; ++p)
            for (int co = 0; co < C; ++co) {
                float main = bias[co], corr = 0.0f;
                for (int tap = 0; tap < 9; ++tap) {
                    const int sy = p / S + tap / 3 - 1, sx = p % S + tap % 3 - 1;
                    if (sy < 0 || sx < 0 || sy >= S || sx >= S) continue;
                    const float *h = &xh[(size_t)(sy * S + sx) * Cin], *l = &xl[(size_t)(sy * S + sx) * Cin];
                    const float *a = &wh[((size_t)tap * C + co) * Cin], *d = &wl[((size_t)tap * C + co) * Cin];
                    for (int ci = 0; ci < Cin; ++ci) {
                        main += a[ci] * h[ci];
                        corr += a[ci] * l[ci];
                        corr += d[ci] * h[ci];
                    }
                }
                float v = fmaf(corr, 1.0f / 2048.0f, main);
                if (rs) v += sp_h_join(rs[yo + at(co, p)], rs[yo + yplane + at(co, p)]);
                // the device epilogues record |v| IN FRONT of the ReLU (one v_max3_f32 |a|, |b| per pair, then ReLU + clamp in one v_med3_f32):
                // a pre-activation below -65504 counts as a range event although the ReLU zeroes it -- deliberately conservative (the
                // post-ReLU record would cost one more VALU per pair in the MFMA shadow; a tower whose negative pre-activations leave
                // the range has positive ones about to).  The twin records at the same place.
                sp_h_note(v);
                if (relu && v < 0.0f) v = 0.0f;
                sp_h_split_quiet(v, ys[yo + at(co, p)], ys[yo + yplane + at(co, p)]);
            }
    }
    return 0;
}
int launch_conv3x3_split(const void* x, const void* w, const float* bias, const void* res, void* y, long long boards, int S, int C, int relu,
                         void*, unsigned* range) {
    if (S < 3 || S > 64 || (C != 64 && C != 128 && C != 256)) return 1;  // (round 6: any plane size, k_conv3x3_spg)
    SpRangeScope scope(range);
    return host_conv_split(x, w, bias, res, y, boards, S, C, C, relu);
}
int launch_resblock_split(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, long long boards, int S, int C,
                          void*, unsigned* range) {
    if ((S != 17 && S != 9) || C != 64) return 1;
    SpRangeScope scope(range);
    std::vector<unsigned short> mid((size_t)boards * 2 * S * S * C);
    if (host_conv_split(x, w1, b1, nullptr, mid.data(), boards, S, C, C, 1)) return 1;
    return host_conv_split(mid.data(), w2, b2, x, y, boards, S, C, C, 1);
}
long long small_batch_waves(long long n) {  // (the twin's plain loops do not depend on the batch size: the value is only kept)
    static long long cur = 1024;
    const long long old = cur;
    if (n >= 0) cur = n;
    return old;
}
int split_range_read(const unsigned* rec, unsigned out[2], int reset, void*) {
    unsigned* r = rec ? (unsigned*)rec : g_sp_range_host;
    out[0] = r[0], out[1] = r[1];
    if (reset) r[0] = r[1] = 0u;
    return 0;
}
int launch_split_features(const float* src, void* dst, long long boards, int S, int cin, void*, unsigned* range) {
    if (cin < 1 || cin > 32) return 1;
    SpRangeScope scope(range);
    const int P2 = S * S;
    const size_t plane = (size_t)4 * P2 * 8;
    unsigned short* d = (unsigned short*)dst;
    for (long long b = 0; b < boards; ++b)
        for (int ch = 0; ch < 32; ++ch)
            for (int p = 0; p < P2; ++p) {
                const size_t so = (size_t)b * 2 * plane + ((size_t)(ch / 8) * P2 + p) * 8 + ch % 8;
                sp_h_split(ch < cin ? src[((size_t)b * cin + ch) * P2 + p] : 0.0f, d[so], d[so + plane]);
            }
    return 0;
}
int launch_stem_split(const void* x, const void* w, const float* bias, void* y, long long boards, int S, int C, int pad, int relu, void*, int,
                      unsigned* range) {
    if (!((S == 9 && (C == 128 || C == 64) && pad == 1) || (S == 13 && C == 64 && pad == 3))) return 1;
    SpRangeScope scope(range);
    if (pad == 1) return host_conv_split(x, w, bias, nullptr, y, boards, S, 32, C, relu);
    // embed the board at (pad - 1, pad - 1) of a zero plane of S + 2 (pad - 1): the pad-3 convolution of the board is the pad-1 convolution of that plane
    const int So = S + 2 * (pad - 1), off = pad - 1, P2 = S * S, Po = So * So;
    std::vector<unsigned short> xe((size_t)boards * 2 * 4 * Po * 8, 0);
    const unsigned short* xs = (const unsigned short*)x;
    for (long long b = 0; b < boards; ++b)
        for (int pc = 0; pc < 8; ++pc)  // (plane, chunk)
            for (int yy = 0; yy < S; ++yy)
                for (int xx = 0; xx < S; ++xx)
                    memcpy(&xe[(((size_t)b * 8 + pc) * Po + (size_t)(yy + off) * So + xx + off) * 8], &xs[(((size_t)b * 8 + pc) * P2 + (size_t)yy * S + xx) * 8], 16);
    return host_conv_split(xe.data(), w, bias, nullptr, y, boards, So, 32, C, relu);
}
int launch_head_split(const HeadSplitArgs& a, void*) {
    const int P2 = a.S * a.S, nch = a.C / 8, kp = a.npol * P2, kv = (3 - a.npol) * P2;
    if (a.C % 8 || a.npol < 1 || a.npol > 2) return 1;
    const size_t plane = (size_t)nch * P2 * 8;
    const unsigned short* xs = (const unsigned short*)a.x;
    std::vector<float> hp((size_t)3 * P2), out((size_t)a.A + a.F);
    for (long long b = 0; b < a.boards; ++b) {
        for (int pl = 0; pl < 3; ++pl)
            for (int p = 0; p < P2; ++p) {
                float acc = a.hb[pl];
                for (int ch = 0; ch < a.C; ++ch) {
                    const size_t so = (size_t)b * 2 * plane + ((size_t)(ch / 8) * P2 + p) * 8 + ch % 8;
                    acc = fmaf(sp_h_join(xs[so], xs[so + plane]), a.hw[pl * a.C + ch], acc);
                }
                hp[(size_t)pl * P2 + p] = acc > 0.0f ? acc : 0.0f;
            }
        for (int o = 0; o < a.A; ++o) {
            float acc = a.bp[o];
            for (int k = 0; k < kp; ++k) acc = fmaf(hp[k], a.wp_t[(size_t)k * a.A + o], acc);
            out[o] = acc;
        }
        for (int f = 0; f < a.F; ++f) {
            float acc = a.b1[f];
            for (int k = 0; k < kv; ++k) acc = fmaf(hp[kp + k], a.w1_t[(size_t)k * a.F + f], acc);
            out[a.A + f] = acc > 0.0f ? acc : 0.0f;
        }
        float mx = -INFINITY, sum = 0.0f, v = 0.0f;
        for (int o = 0; o < a.A; ++o) mx = fmaxf(mx, out[o]);
        for (int o = 0; o < a.A; ++o) sum += expf(out[o] - mx);
        for (int o = 0; o < a.A; ++o) a.priors[(size_t)b * a.A + o] = expf(out[o] - mx) / sum;
        for (int f = 0; f < a.F; ++f) v = fmaf(out[a.A + f], a.w2[f], v);
        a.values[b] = tanhf(v + a.b2);
    }
    return 0;
}
int launch_stem_tiled(const void* x, const void* w, const float* bias, void* y, long long boards, int S, int C, int pad, int relu, void*, int f16) {
    if (f16 && (S != 9 || C != 128 || pad != 1)) return 1;
    // un-tile the 32-channel features, embed the board at (pad - 1, pad - 1) of a zero plane of S + 2 (pad - 1), pad-1 convolution
    const int So = S + 2 * (pad - 1), off = pad - 1;
    std::vector<unsigned short> xn((size_t)boards * S * S * 32), xe((size_t)boards * So * So * 32, 0), yn((size_t)boards * So * So * C);
    host_tile_layout((const unsigned short*)x, xn.data(), boards, S, 32, 0);
    for (long long b = 0; b < boards; ++b)
        for (int yy = 0; yy < S; ++yy)
            for (int xx = 0; xx < S; ++xx)
                for (int c = 0; c < 32; ++c)
                    xe[(((size_t)b * So + yy + off) * So + xx + off) * 32 + c] = xn[(((size_t)b * S + yy) * S + xx) * 32 + c];
    cv_host_conv3x3_io(xe.data(), (const unsigned short*)w, bias, nullptr, yn.data(), (int)boards, So, 32, C, relu, f16);
    host_tile_layout(yn.data(), (unsigned short*)y, boards, So, C, 1);
    return 0;
}
int launch_head_tiled(const void* x, const float* w, const float* bias, void* pol, void* val, long long boards, int S, int C, int npol, int nval,
                      int pol_stride, int val_stride, void*, int f16) {
    const size_t n = (size_t)boards * S * S * C;
    std::vector<unsigned short> xn(n);
    host_tile_layout((const unsigned short*)x, xn.data(), boards, S, C, 0);
    const int P2 = S * S;
    for (long long b = 0; b < boards; ++b)
        for (int q = 0; q < P2; ++q)
            for (int pl = 0; pl < npol + nval; ++pl) {
                float acc = bias[pl];
                for (int ci = 0; ci < C; ++ci) acc += cv_h_in(xn[((size_t)b * P2 + q) * C + ci], f16) * w[pl * C + ci];
                const unsigned short h = cv_h_out(acc > 0.0f ? acc : 0.0f, f16);
                if (pl < npol) ((unsigned short*)pol)[(size_t)b * pol_stride + pl * P2 + q] = h;
                else ((unsigned short*)val)[(size_t)b * val_stride + (pl - npol) * P2 + q] = h;
            }
    return 0;
}
int launch_fc_heads(const FcHeadsArgs& a, void*) {
    const unsigned short *pol = (const unsigned short*)a.pol, *val = (const unsigned short*)a.val, *wp = (const unsigned short*)a.wp,
                         *w1 = (const unsigned short*)a.w1;
    const int k1 = a.ks1 * 16, k2 = a.ks2 * 16;
    std::vector<float> lg(a.A);
    for (long long b = 0; b < a.boards; ++b) {
        float mx = -1e30f;
        for (int n = 0; n < a.A; ++n) {
            float acc = 0.0f;
            for (int k = 0; k < k1; ++k) acc += cv_h_in(wp[(size_t)n * k1 + k], a.f16) * cv_h_in(pol[(size_t)b * k1 + k], a.f16);
            lg[n] = acc + a.bp[n];
            if (lg[n] > mx) mx = lg[n];
        }
        float sum = 0.0f;
        for (int n = 0; n < a.A; ++n) {
            lg[n] = expf(lg[n] - mx);
            sum += lg[n];
        }
        for (int n = 0; n < a.A; ++n) a.priors[(size_t)b * a.A + n] = lg[n] / sum;
        float v = 0.0f;
        for (int n = 0; n < a.F; ++n) {
            float acc = 0.0f;
            for (int k = 0; k < k2; ++k) acc += cv_h_in(w1[(size_t)n * k2 + k], a.f16) * cv_h_in(val[(size_t)b * k2 + k], a.f16);
            acc += a.b1[n];
            v += (acc > 0.0f ? acc : 0.0f) * a.w2[n];
        }
        a.values[b] = tanhf(v + a.b2);
    }
    return 0;
}
}  // namespace azb
