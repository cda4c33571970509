// az_bitboard.h -- N x N boards as W = ceil(N*N/64) 64-bit words, bit i == point i (row-major,
// i = row*N + col), which is also the action index, so a legal-move bitboard IS the action mask and
// a 64-lane ballot over "point = lane + 64*k" IS word k of a bitboard.
#pragma once
#include "az_wave.h"

template <int N> struct Geo {
    static constexpr int NP = N * N;
    static constexpr int W = (NP + 63) / 64;
};

template <int W> struct BB {
    u64 w[W];
};

template <int N> struct BBOps {
    static constexpr int NP = Geo<N>::NP;
    static constexpr int W = Geo<N>::W;
    typedef BB<W> B;
    struct Masks {
        u64 nl[W], nf[W];
        constexpr Masks() : nl(), nf() {
            for (int i = 0; i < W; ++i) {
                u64 a = 0, b = 0;
                for (int k = 0; k < 64; ++k) {
                    int p = 64 * i + k;
                    if (p < NP && (p % N) != N - 1) a |= 1ull << k;
                    if (p < NP && (p % N) != 0) b |= 1ull << k;
                }
                nl[i] = a;
                nf[i] = b;
            }
        }
    };

    static AZ_HD B zero() {
        B r;
        for (int i = 0; i < W; ++i) r.w[i] = 0;
        return r;
    }
    static AZ_HD u64 valid_word(int i) {
        int rem = NP - 64 * i;
        return rem >= 64 ? ~0ull : ((1ull << rem) - 1ull);
    }
    static AZ_HD B band(B a, B b) { for (int i = 0; i < W; ++i) a.w[i] &= b.w[i]; return a; }
    static AZ_HD B bor(B a, B b) { for (int i = 0; i < W; ++i) a.w[i] |= b.w[i]; return a; }
    static AZ_HD B andnot(B a, B b) { for (int i = 0; i < W; ++i) a.w[i] &= ~b.w[i]; return a; }
    static AZ_HD B inv(B a) { for (int i = 0; i < W; ++i) a.w[i] = ~a.w[i] & valid_word(i); return a; }
    static AZ_HD bool any(B a) { u64 m = 0; for (int i = 0; i < W; ++i) m |= a.w[i]; return m != 0; }
    static AZ_HD bool eq(B a, B b) { u64 m = 0; for (int i = 0; i < W; ++i) m |= a.w[i] ^ b.w[i]; return m == 0; }
    static AZ_HD int count(B a) { int c = 0; for (int i = 0; i < W; ++i) c += __builtin_popcountll(a.w[i]); return c; }
    // No dynamic indexing of the word array: a runtime index would push the (register-resident) bitboard to scratch.
    static AZ_HD bool test(const B& a, int p) {
        const int wi = p >> 6;
        u64 word = 0;
        for (int i = 0; i < W; ++i) word = (i == wi) ? a.w[i] : word;
        return (word >> (p & 63)) & 1ull;
    }
    static AZ_HD B bit(int p) {
        const int wi = p >> 6;
        const u64 m = 1ull << (p & 63);
        B r;
        for (int i = 0; i < W; ++i) r.w[i] = (i == wi) ? m : 0ull;
        return r;
    }
    static AZ_HD int first(B a) {  // lowest set bit, -1 if none
        for (int i = 0; i < W; ++i)
            if (a.w[i]) return 64 * i + __builtin_ctzll(a.w[i]);
        return -1;
    }
    template <int S> static AZ_HD B shl(B a) {  // towards higher indices, S < 64
        B r;
        for (int i = W - 1; i >= 0; --i) {
            u64 lo = i > 0 ? (a.w[i - 1] >> (64 - S)) : 0ull;
            r.w[i] = ((a.w[i] << S) | lo) & valid_word(i);
        }
        return r;
    }
    template <int S> static AZ_HD B shr(B a) {
        B r;
        for (int i = 0; i < W; ++i) {
            u64 hi = i + 1 < W ? (a.w[i + 1] << (64 - S)) : 0ull;
            r.w[i] = (a.w[i] >> S) | hi;
        }
        return r;
    }
    // The 4-neighbourhood of a set (go_engine.py:50 NEIGHBORS), excluding nothing: result may overlap a.
    static AZ_HD B nbr(B a) {
        constexpr Masks M = Masks();
        B e, wst;
        for (int i = 0; i < W; ++i) {
            e.w[i] = a.w[i] & M.nl[i];
            wst.w[i] = a.w[i] & M.nf[i];
        }
        B r = shl<1>(e);
        r = bor(r, shr<1>(wst));
        r = bor(r, shl<N>(a));
        r = bor(r, shr<N>(a));
        return r;
    }
    // Grow `seed` through `mask` to a fixed point (connected components of `mask` touching `seed`).
    static AZ_HD B flood(B seed, B mask) {
        B cur = band(seed, mask);
        for (;;) {
            B nxt = band(bor(cur, nbr(cur)), mask);
            if (eq(nxt, cur)) return cur;
            cur = nxt;
        }
    }
};
