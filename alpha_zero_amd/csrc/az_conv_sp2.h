// az_conv_sp2.h -- k_conv3x3_sp2<RES>: the fp32-class 3x3 convolution of the 9x9 x 128 tower (az_conv_sp.h: hi + lo f16 pairs, three
// v_mfma_f32_16x16x32_f16 products per multiply, fp32 accumulation) with a 2 x 2 SPLIT OF THE CU'S WORK BETWEEN ITS FOUR WAVES (round 6):
//     y = relu(conv3x3(x, w) + bias [+ residual])              (alpha_zero/core/network.py:42-82, eval mode, BatchNorm folded;
//                                                               the reference evaluates in fp32: core/pipeline.py:91-123)
// k_conv3x3_sp gives every wave 16 couts x all 128 cin: all four waves read EVERY B fragment of the board from LDS (2 reads per 3 MFMAs).
// Measured (tools/probes/split_form_probe.hip, profiles/r06_split_form_probe.txt): with those reads in the loop the matrix cores sustain
// 1.50 PF/s; with half of them 1.70 PF/s.  Here wave (a, b) holds 32 couts (two 16-cout tiles) x the cin HALF b: the same 288 weight
// registers, every fragment feeds 6 MFMAs instead of 3, a wave reads half the image.  The two cin halves of a cout tile meet through LDS:
// after a unit each wave joins (main + 2^-11 corr) the partial sums of the tile its PARTNER finalises, hands them over (one ds_write_b128
// per column tile), and runs the epilogue of its own tile with the partner's partial added in.  Wave b = 1 loads its weight tiles swapped,
// so "own tile" is index 0 in both waves (compile-time register indices); the bias enters through the own tile's main accumulator only.
//   * LDS: the two x images (120 KB) + the corner side buffer + 12 KB of hand-over buffer (3 column tiles x 4 waves x 1 KB) -- the side
//     buffer batches 13 boards instead of 16 to make the room (the corner phase runs every 13 boards: + 0.3 % of a launch).
//   * barriers per board: one in front of each unit's hand-over reads, one in front of unit 1's hand-over writes (it orders them behind the
//     partner's reads of the previous hand-over: the buffer is single), the board barrier.
//   * arithmetic: result = (main_a + 2^-11 corr_a) + (main_b + 2^-11 corr_b) [+ residual]: two fp32 accumulation chains of 64 cin x 9 taps
//     instead of one of 128 x 9, joined once -- fp32 round-off class as before (tests/test_split_tower.py bounds), not bit-identical to
//     k_conv3x3_sp.
#pragma once
#include "az_conv_sp.h"

#if defined(__HIPCC__)
#define SP2_NB 13  // boards per corner batch
// 1 (the product): fragment f of a k-step is requested in the MFMA gap behind MFMA f of the k-step before; 0 (-DSP2_SPREAD_FRAGS=0, for
// same-box A/B builds): rounds 3-6's burst of all 2 nj ds_read_b128 in front of a k-step.  Same arithmetic either way (bit-identical);
// the burst costs 3.8 % / 3.2 % of a launch (profiles/r06_spread_frags_ab.txt): with one wave per SIMD nothing issues MFMAs meanwhile.
#ifndef SP2_SPREAD_FRAGS
#define SP2_SPREAD_FRAGS 1
#endif

// Vector-memory instructions a wave issues between the last LDS-DMA piece of the next board (unit 0, behind the MFMAs of k-step NPIECE) and the
// board barrier (unit 1, k-step KS - (R - 1)): the stores of the finishing riders (2 per column tile) and unit 1's residual loads.  Mirrors the
// kernel's schedule (same constants, same SpSpread arithmetic); every one of them is issued unconditionally.
// one v_fma_f32, opaque to the compiler: left to itself it pairs the four joins of a hand-over into two v_pk_fma_f32, and a packed fp32
// instruction costs about four scalar ones beside the MFMA stream (measured: profiles/r06_packed_epilogue_ab.txt)
__device__ __forceinline__ float sp2_fma_f32(float a, float s, float c) {
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(s), "v"(c));
    return r;
}
template <bool RES> __host__ __device__ constexpr int sp2_vm_younger() {
    constexpr int KS = 18, R = 2, S0 = 6, NPIECE = 16, NT = 2;
    constexpr int E1 = RES ? 4 : 2, PAIR = 2 * E1 + 8, CT_OPS = 2 * PAIR + 2, P2CT = CT_OPS + 1;
    int n = 0;
    for (int i = 0; i < 2; ++i) {
        const int nj = i == 0 ? 3 : 2, pnj = i == 0 ? 2 : 3, NQ = NT * 3 * nj, SBX = S0 + 5 * pnj + 1, P2 = pnj * P2CT;
        const int AVAIL = (i == 1 ? (KS - (R - 1)) * NQ : NQ * KS - 4) - (SBX + 1);
        auto cum = [&](int sl) {
            if (sl < SBX + 1) return 0;
            const long long c = ((long long)(sl - (SBX + 1) + 1) * P2 + AVAIL - 1) / AVAIL;
            return c > P2 ? P2 : (int)c;
        };
        for (int c = 0; c < pnj; ++c)
            for (int st = 0; st < 2; ++st) {  // the two stores of column tile c are its last two micro-ops
                const int o = c * P2CT + P2CT - 2 + st;
                int sl = SBX + 1;
                while (cum(sl) <= o) ++sl;
                if (i > 0 || sl / NQ > NPIECE) ++n;  // a DMA piece is issued behind all MFMA slots of its k-step
            }
        if (RES && i > 0) n += 2 * nj;  // (unit 0 loads its residual in its first slots, before the first piece)
    }
    return n;
}

template <bool RES> __global__ void __launch_bounds__(CW_THREADS, 1)
k_conv3x3_sp2(const unsigned char* __restrict__ x, const _Float16* __restrict__ w, const float* __restrict__ bias,
              const unsigned char* __restrict__ res, unsigned char* __restrict__ y, int ntiles, int relu, unsigned* range) {
    typedef SpGeo9 G;
    constexpr int C = 128, CIN = 128, NCH = 16, NCG = 2;
    constexpr int KSUB = 2, KS = 9 * KSUB;                    // k-steps of a unit: one tap x 32 input channels of the wave's 64-channel half
    constexpr int NT = 2;                                     // cout tiles of 16 per wave: index 0 = own (finalised here), 1 = the partner's
    constexpr int NJ0 = 3, NJ1 = 2, R = 2;
    constexpr int LBLK = G::CELLS * 16, LPLANE = NCH * LBLK, LBUF = 2 * LPLANE;
    constexpr int GBLK = G::P2 * 16, XPLANE = NCH * GBLK, XTILE = 2 * XPLANE;
    constexpr int YPLANE = (C / 8) * GBLK, YTILE = 2 * YPLANE;
    constexpr int NP = (G::CELLS + 63) / 64, SPW = 2 * NCH / 4, NPIECE = NP * SPW;
    constexpr int NF = NT * 2 * KS;                           // A fragments: f = (tile * 2 + plane) * KS + k-step
    constexpr int NF_A = 64;
    // epilogue micro-ops per column tile: per element join, + the partner's partial, (residual join, add); per pair 8; 2 stores; + 1 hand-over read
    constexpr int E1 = RES ? 4 : 2, PAIR = 2 * E1 + 8, CT_OPS = 2 * PAIR + 2, P2CT = CT_OPS + 1;
    constexpr int S0 = 6;
    static_assert((2 * KS) % R == 0 && KS - 1 >= NPIECE, "ring phase; the next board's pieces ride in unit 0");
    constexpr int SIDE_BOARD = 8 * NCH * 16 + 16, SIDE0 = 2 * LBUF + 16, XB0 = SIDE0 + SP2_NB * SIDE_BOARD, XBCT = 4 * 1024;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[XB0 + NJ0 * XBCT];
    static_assert(XB0 + NJ0 * XBCT <= 160 * 1024 && XB0 % 16 == 0, "LDS budget");
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave >> 1, wb = wave & 1;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
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
    for (int i = tid; i < (XB0 + NJ0 * XBCT) / 16; i += CW_THREADS) *(cv_u32x4*)(lds + i * 16) = (cv_u32x4){0u, 0u, 0u, 0u};
    CV_BARRIER();
    if (slot >= ntiles) return;

    // A fragments of tile index tt = cout tile (tt + b) & 1 of the wave pair's 32 couts, cin half b
    sp_f16x8 wf[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int tt = f / (2 * KS), pl = (f / KS) & 1, st = f % KS;
        wf[f] = *(const sp_f16x8*)(w + ((size_t)((pl * 9 + st / KSUB) * C + cg * 64 + wa * 32 + ((tt + wb) & 1) * 16 + l15)) * CIN + wb * 64 + (st % KSUB) * 32 + kg * 8);
    }
    c6_f32x4 bv;  // bias of the OWN tile in the D layout
#pragma unroll
    for (int e = 0; e < 4; ++e) bv[e] = bias[cg * 64 + wa * 32 + wb * 16 + 4 * kg + e];
    const float lo_relu = relu ? 0.0f : -__builtin_inff();
    const float lo_clamp = relu ? 0.0f : -SP_F16_MAX;

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
    unsigned lmap[G::NCT], omap[G::NCT];
#pragma unroll
    for (int j = 0; j < G::NCT; ++j) {
        const int pos = sp_map9.pos[j * 16 + l15];
        lmap[j] = (unsigned)((G::PITCH * (pos / G::S) + pos % G::S) * 16 + kg * LBLK + wb * (8 * LBLK));  // (-1, -1) neighbour, this lane's group of the wave's cin half
        omap[j] = (unsigned)(pos * 16 + (kg >> 1) * GBLK + (kg & 1) * 8);
    }
    auto out_off = [&](int j) __attribute__((always_inline)) {
        unsigned v = omap[j];
        asm volatile("" : "+v"(v));
        return v;
    };
    sp_f16x8 bb[R][2][NJ0];
    auto load_step = [&](const unsigned char* img, int j0, int nj, int s, int rs) {
        const int tap = s / KSUB;
        const int off = ((tap / 3) * G::PITCH + (tap % 3)) * 16 + (s % KSUB) * (4 * LBLK);
#pragma unroll
        for (int j = 0; j < NJ0; ++j)
            if (j < nj) bb[rs][0][j] = *(const sp_f16x8*)(img + lmap[j0 + j] + off);
#pragma unroll
        for (int j = 0; j < NJ0; ++j)
            if (j < nj) bb[rs][1][j] = *(const sp_f16x8*)(img + lmap[j0 + j] + off + LPLANE);
    };
    // fragment f of a k-step's 2 nj (plane f / nj, column tile f % nj) -- inside the units the requests are SPREAD over the first MFMA gaps of
    // the k-step before (one ds_read_b128 per gap) instead of being issued as a burst of 2 nj reads in front of it
    auto load_frag = [&](const unsigned char* img, int j0, int nj, int s, int rs, int f) __attribute__((always_inline)) {
        const int tap = s / KSUB;
        const int off = ((tap / 3) * G::PITCH + (tap % 3)) * 16 + (s % KSUB) * (4 * LBLK);
        const int pl = f / nj, j = f % nj;
        bb[rs][pl][j] = *(const sp_f16x8*)(img + lmap[j0 + j] + off + pl * LPLANE);
    };
    {
        const unsigned char* src = x + (size_t)slot * XTILE;
#pragma unroll
        for (int i = 0; i < NPIECE; ++i) dma_piece(src, lds0, true, i);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CV_BARRIER();
#pragma unroll
        for (int s = 0; s < R - 1; ++s) load_step(lds, 0, NJ0, s, s);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        if (f < NF_A) asm volatile("" : : "a"(wf[f]));
        else asm volatile("" : : "v"(wf[f]));
    }
    asm volatile("" : : "v"(bv), "v"(lmap[0]), "v"(lmap[G::NCT - 1]), "v"(dsrc[0]), "v"(dsrc[NP - 1]));

    c6_f32x4 accm[2][NJ0][NT], accc[2][NJ0][NT];  // [unit][column tile][cout tile]
    cv_u32x2 rr[2][NJ0][2];                       // residual of the own tile: [unit][column tile][plane]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int j = 0; j < NJ0; ++j) {
#pragma unroll
            for (int t = 0; t < NT; ++t) accm[a][j][t] = (c6_f32x4){0.0f, 0.0f, 0.0f, 0.0f}, accc[a][j][t] = (c6_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            rr[a][j][0] = (cv_u32x2){0u, 0u}, rr[a][j][1] = (cv_u32x2){0u, 0u};
        }
    float evv[2] = {0.0f, 0.0f}, sc[2] = {0.0f, 0.0f}, t0 = 0.0f, mx = 0.0f;
    unsigned hpk[2] = {0u, 0u}, lpk[2] = {0u, 0u};
    c6_f32x4 xs = (c6_f32x4){0.0f, 0.0f, 0.0f, 0.0f}, xr = (c6_f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    unsigned char* const xb_mine = lds + XB0 + wave * 1024 + lane * 16;
    const unsigned char* const xb_partner = lds + XB0 + (wave ^ 1) * 1024 + lane * 16;
    // hand-over micro-op o of column tile c of the unit with accumulator set `set`: join the partner's tile (4), write it (1)
    auto send_op = [&](int set, int c, int o) {
        if (o < 4) xs[o] = sp2_fma_f32(accc[set][c][1][o], SP_INV_SCALE, accm[set][c][1][o]);
        else *(c6_f32x4*)(xb_mine + c * XBCT) = xs;
    };
    // micro-op o of the finishing of column tile c (lmap index mj): the partner's partial (1 read), then the epilogue of the own tile
    auto fin_op = [&](int set, int c, int mj, unsigned char* out, sp_gptr out_lo, int o, bool store_ok) {
        if (o == 0) {
            xr = *(const c6_f32x4*)(xb_partner + c * XBCT);
            return;
        }
        o -= 1;
        if (o < 2 * PAIR) {
            const int pr = o / PAIR, k = o % PAIR;
            if (k < 2 * E1) {
                const int ei = k / E1, kk = k % E1, e = 2 * pr + ei;
                const unsigned rh = pr == 0 ? rr[set][c][0].x : rr[set][c][0].y, rl = pr == 0 ? rr[set][c][1].x : rr[set][c][1].y;
                if (kk == 0) evv[ei] = fmaf(accc[set][c][0][e], SP_INV_SCALE, accm[set][c][0][e]);
                else if (kk == 1) evv[ei] = cw_add_f32(evv[ei], xr[e]);
                else if (RES && kk == 2) t0 = ei == 0 ? sp_mix_join<0>(rh, rl) : sp_mix_join<1>(rh, rl);
                else if (RES && kk == 3) evv[ei] = cw_add_f32(evv[ei], t0);
            } else {
                const int kk = k - 2 * E1;
                if (kk == 0) mx = sp_max3_abs(mx, evv[0], evv[1]);
                else if (kk == 1) evv[0] = __builtin_amdgcn_fmed3f(evv[0], lo_clamp, SP_F16_MAX);
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

    const bool copier = tid < 8 * NCH;
    int csrc, cdst;
    {
        constexpr int KF = NCH / 4;  // k-steps of 32 over all 128 input channels (the side buffer holds both halves)
        const int c = copier ? tid : 0, ks = c % KF, pl = (c / KF) & 1, tp = (c / (2 * KF)) & 3, g4 = c / (8 * KF);
        csrc = (pl * NCH + 4 * ks + g4) * LBLK + (G::CELL0 + G::PITCH * (G::S - 2 + (tp >> 1)) + (tp & 1)) * 16;
        cdst = SIDE0 + c * 16;
    }
    cv_u32x4 ctmp = (cv_u32x4){0u, 0u, 0u, 0u};

    int it = 0, cb = 0;  // cb: boards in the corner side buffer
    unsigned char* yprev = y;
    sp_gptr yprev_lo = (sp_gptr)(unsigned long long)y;
    for (int tile = slot; tile < ntiles; tile += nslot, ++it) {
        const int buf = it & 1;
        const unsigned char* Xs = lds + buf * LBUF;
        const unsigned char* Xn = lds + (buf ^ 1) * LBUF;
        const bool has_next = tile + nslot < ntiles;
        const unsigned char* nsrc = x + (size_t)(has_next ? tile + nslot : tile) * XTILE;
        const unsigned ndst = lds0 + (unsigned)((buf ^ 1) * LBUF);
        const size_t yo = (size_t)tile * YTILE + (size_t)(cg * 8 + wa * 4 + wb * 2) * GBLK;  // the own tile's two chunk strips
        const unsigned char* rbase = RES ? res + yo : nullptr;
        unsigned char* ybase = y + yo;
        unsigned long long rlo = (unsigned long long)(RES ? res + yo : y + yo) + YPLANE, ylo = (unsigned long long)(y + yo) + YPLANE;
        asm volatile("" : "+s"(rlo), "+s"(ylo));
        const sp_gcptr rbase_lo = (sp_gcptr)rlo;
        const sp_gptr ybase_lo = (sp_gptr)ylo;
        const bool have_prev = it > 0;
        auto unit = [&](auto IC) __attribute__((always_inline)) {
            constexpr int i = decltype(IC)::value, set = i, pset = i ^ 1;
            constexpr int nj = i == 0 ? NJ0 : NJ1, j0 = i == 0 ? 0 : NJ0;
            constexpr int pnj = i == 0 ? NJ1 : NJ0, pj0 = i == 0 ? NJ0 : 0;
            constexpr int nnj = pnj, nj0 = pj0;
            constexpr int NQ = NT * 3 * nj;                     // MFMAs per k-step
            constexpr int P1 = 5 * pnj, SBX = S0 + P1 + 1;      // hand-over micro-ops in the slots S0 .. S0 + P1 - 1, their barrier in front of slot SBX
            constexpr int P2 = pnj * P2CT;
            constexpr int AVAIL = (i == 1 ? (KS - (R - 1)) * NQ : NQ * KS - 4) - (SBX + 1);
            typedef SpSpread<P2, SBX + 1, AVAIL> SP;
            static_assert(SP::MAXPER <= 1 && SBX + 1 < NQ * 2, "the previous unit's finishing fits this unit's MFMA gaps");
            unsigned char* pout = i == 0 ? yprev : ybase;
            const sp_gptr pout_lo = i == 0 ? yprev_lo : ybase_lo;
            const bool pstore = i > 0 || have_prev;
            cp_for_each([&](auto TC) __attribute__((always_inline)) {
                constexpr int t = decltype(TC)::value;
                constexpr int g = i * KS + t;
                if constexpr (i == 1 && t == KS - (R - 1)) {
                    // the board barrier: every read of this image has been issued; this wave's pieces of the next board (unit 0) are older than
                    // the VM_YOUNGER youngest vector-memory instructions it has issued (finishing stores, unit 1's residual loads), which may stay in flight
                    constexpr int VM_YOUNGER = sp2_vm_younger<RES>();
                    static_assert(VM_YOUNGER < 63, "vmcnt field");
                    if (have_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_YOUNGER) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // first board: conservative
                    CV_BARRIER();
                }
                // the fragments of k-step t + R - 1 (of this unit, or of the next unit: unit 1 of this board / unit 0 of the next board's image)
                constexpr bool lsame = t + R - 1 < KS;
                constexpr int ls = lsame ? t + R - 1 : t + R - 1 - KS, lnj = lsame ? nj : nnj, lj0 = lsame ? j0 : nj0, lrs = (g + R - 1) % R;
                const unsigned char* limg = (lsame || i == 0) ? Xs : Xn;
                if constexpr (!SP2_SPREAD_FRAGS) load_step(limg, lj0, lnj, ls, lrs);
                cp_for_each([&](auto QC) __attribute__((always_inline)) {
                    constexpr int q = decltype(QC)::value, j = q % nj, r6 = q / nj, tt = r6 / 3, prod = r6 % 3;
                    constexpr int sl = t * NQ + q;  // MFMA slot of the unit
                    if constexpr (i == 1 && sl == S0 - 1) CV_BARRIER();  // every wave has finished unit 0 and with it its reads of the previous hand-over: the buffer is free
                    if constexpr (sl == SBX) CV_BARRIER();                // the hand-over writes of all waves are in LDS
                    constexpr int fa = (tt * 2 + (prod == 2 ? 1 : 0)) * KS + t, pl = prod == 1 ? 1 : 0;
                    if constexpr (prod == 0) {
                        if constexpr (t == 0 && tt == 0) sp_mfma_ac(accm[set][j][tt], wf[fa], bb[g % R][pl][j], bv);
                        else if constexpr (t == 0) sp_mfma_a0(accm[set][j][tt], wf[fa], bb[g % R][pl][j]);
                        else if constexpr (fa < NF_A) sp_mfma_a(accm[set][j][tt], wf[fa], bb[g % R][pl][j]);
                        else sp_mfma_v(accm[set][j][tt], wf[fa], bb[g % R][pl][j]);
                    } else if constexpr (prod == 1) {
                        if constexpr (t == 0) sp_mfma_a0(accc[set][j][tt], wf[fa], bb[g % R][pl][j]);
                        else if constexpr (fa < NF_A) sp_mfma_a(accc[set][j][tt], wf[fa], bb[g % R][pl][j]);
                        else sp_mfma_v(accc[set][j][tt], wf[fa], bb[g % R][pl][j]);
                    } else {
                        if constexpr (fa < NF_A) sp_mfma_a(accc[set][j][tt], wf[fa], bb[g % R][pl][j]);
                        else sp_mfma_v(accc[set][j][tt], wf[fa], bb[g % R][pl][j]);
                    }
                    static_assert(2 * lnj <= NQ, "a fragment per MFMA gap");
                    if constexpr (SP2_SPREAD_FRAGS && q < 2 * lnj) load_frag(limg, lj0, lnj, ls, lrs, q);
                    if constexpr (sl >= S0 && sl < S0 + P1) send_op(pset, (sl - S0) / 5, (sl - S0) % 5);
                    if constexpr (sl > SBX) {
                        constexpr int o = SP::cum(sl - 1);
                        if constexpr (o < SP::cum(sl)) fin_op(pset, o / P2CT, pj0 + o / P2CT, pout, pout_lo, o % P2CT, pstore);
                    }
                    if constexpr (RES && sl < 2 * nj) {  // this unit's residual of the own tile (used by its finishing inside the next unit)
                        constexpr int rj = sl >> 1, rp = sl & 1;
                        if constexpr (rp == 0) rr[set][rj][rp] = *(const cv_u32x2*)(rbase + out_off(j0 + rj));
                        else rr[set][rj][rp] = *(const __attribute__((address_space(1))) cv_u32x2*)(rbase_lo + out_off(j0 + rj));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }, typename CpMakeSeq<NQ>::type{});
                if constexpr (i == 0 && t >= 1 && t - 1 < NPIECE) dma_piece(nsrc, ndst, has_next, t - 1);
                if constexpr (i == 0 && t == 2) {
                    if (copier) ctmp = *(const cv_u32x4*)(Xs + csrc);
                }
                if constexpr (i == 0 && t == 5) {
                    if (copier) *(cv_u32x4*)(lds + cdst + cb * SIDE_BOARD) = ctmp;
                }
                __builtin_amdgcn_sched_barrier(0);
            }, typename CpMakeSeq<KS>::type{});
        };
        unit(CpInt<0>{});
        unit(CpInt<1>{});
        yprev = ybase, yprev_lo = ybase_lo;
        cb += 1;
        if (cb == SP2_NB || !has_next) {
            // ---- the corner (8, 0) of the last cb boards: [2 x 16 couts] x [<= 13 boards] x [4 taps x this wave's 64 cin] per wave, then the hand-over
            const int n_ok = cb, it0 = it - (cb - 1);
            c6_f32x4 cm[NT], cc[NT];
            constexpr int KF = NCH / 4;
            const unsigned char* sb = lds + SIDE0 + l15 * SIDE_BOARD + kg * (8 * KF * 16);
            const int nb = l15 < n_ok ? l15 : 0;
            const size_t co = (size_t)(slot + (size_t)(it0 + nb) * nslot) * YTILE + (size_t)(cg * 8 + wa * 4 + wb * 2 + (kg >> 1)) * GBLK + G::CORNER * 16 + (kg & 1) * 8;
            if (RES) {
                rr[0][0][0] = *(const cv_u32x2*)(res + co);
                rr[0][0][1] = *(const cv_u32x2*)(res + co + YPLANE);
            }
            cp_for_each([&](auto TC) __attribute__((always_inline)) {
                constexpr int u = decltype(TC)::value, tp = u / KSUB, ks = u % KSUB;
                constexpr int fw = ((tp >> 1) * 3 + 1 + (tp & 1)) * KSUB + ks;  // weight k-step of tap (dy, dx) = (tp / 2 - 1, tp % 2)
                const sp_f16x8 bh = *(const sp_f16x8*)(sb + ((tp * 2 + 0) * KF + wb * KSUB + ks) * 16);
                const sp_f16x8 bl = *(const sp_f16x8*)(sb + ((tp * 2 + 1) * KF + wb * KSUB + ks) * 16);
#pragma unroll
                for (int tt = 0; tt < NT; ++tt) {
                    const int fh = (tt * 2 + 0) * KS + fw, fl = (tt * 2 + 1) * KS + fw;
                    if (u == 0 && tt == 0) sp_mfma_ac(cm[tt], wf[fh], bh, bv);
                    else if (u == 0) sp_mfma_a0(cm[tt], wf[fh], bh);
                    else if (fh < NF_A) sp_mfma_a(cm[tt], wf[fh], bh);
                    else sp_mfma_v(cm[tt], wf[fh], bh);
                    if (u == 0) sp_mfma_a0(cc[tt], wf[fh], bl);
                    else if (fh < NF_A) sp_mfma_a(cc[tt], wf[fh], bl);
                    else sp_mfma_v(cc[tt], wf[fh], bl);
                    if (fl < NF_A) sp_mfma_a(cc[tt], wf[fl], bh);
                    else sp_mfma_v(cc[tt], wf[fl], bh);
                }
                __builtin_amdgcn_sched_barrier(0);
            }, typename CpMakeSeq<4 * KSUB>::type{});
            asm volatile("s_nop 15\n\ts_nop 15" : "+v"(cm[0]), "+v"(cc[0]), "+v"(cm[1]), "+v"(cc[1]));
            c6_f32x4 hs;
#pragma unroll
            for (int e = 0; e < 4; ++e) hs[e] = fmaf(cc[1][e], SP_INV_SCALE, cm[1][e]);
            *(c6_f32x4*)(xb_mine) = hs;
            CV_BARRIER();
            const c6_f32x4 hr = *(const c6_f32x4*)(xb_partner);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fmaf(cc[0][e], SP_INV_SCALE, cm[0][e]) + hr[e];
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
                if (l15 < n_ok) mx = fmaxf(mx, fabsf(v[e]));
                v[e] = fmaxf(v[e], lo_relu);
                sp_split(v[e], h[e], l[e]);
            }
            if (l15 < n_ok) {
                *(cv_u32x2*)(y + co) = (cv_u32x2){sp_pack(h[0], h[1]), sp_pack(h[2], h[3])};
                *(cv_u32x2*)(y + co + YPLANE) = (cv_u32x2){sp_pack(l[0], l[1]), sp_pack(l[2], l[3])};
            }
            cb = 0;
            CV_BARRIER();  // the next batch's copies may overwrite the side buffer, the next hand-over the buffer
        }
    }
    // the very last unit (accumulator set 1, column tiles 3 and 4): hand over, meet, finish
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(accm[1][0][0]), "+v"(accm[1][1][0]), "+v"(accc[1][0][0]), "+v"(accc[1][1][0]), "+v"(accm[1][0][1]), "+v"(accm[1][1][1]),
                 "+v"(accc[1][0][1]), "+v"(accc[1][1][1]));
#pragma unroll
    for (int c = 0; c < NJ1; ++c)
#pragma unroll
        for (int o = 0; o < 5; ++o) send_op(1, c, o);
    CV_BARRIER();
#pragma unroll
    for (int c = 0; c < NJ1; ++c)
#pragma unroll
        for (int o = 0; o < P2CT; ++o) fin_op(1, c, NJ0 + c, yprev, yprev_lo, o, true);
    sp_range_report(mx, range);
}
#endif  // __HIPCC__
