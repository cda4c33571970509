// azsp_impl.h -- the C ABI of include/azsp.h, written once against a tiny backend interface
// (namespace azb: alloc / copy / launch).  alpha_zero_amd/csrc/azsp_hip.hip provides the HIP
// backend (the product, libazsp.so); tests/hosttwin/azsp_host.cpp provides a host backend that
// runs the identical engine source wave-by-wave on the CPU for unit tests without a GPU.
#pragma once
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/azsp.h"
#include "az_conv.h"
#include "az_engine.h"

#define AZ_FOR_EACH_VARIANT(X) \
    X(5, AZ_GO) X(9, AZ_GO) X(13, AZ_GO) X(19, AZ_GO) X(7, AZ_GOMOKU) X(9, AZ_GOMOKU) X(13, AZ_GOMOKU) X(15, AZ_GOMOKU)

// ------------------------------------------------------------------------------------------------
// per-game operations (functors: trivially copyable, passed by value to the kernel)
// ------------------------------------------------------------------------------------------------
struct OpReset {
    template <class E> AZ_HD void operator()(E& e) const {
        if (E::Wave::first()) {
            e.gr.games_done = 0;
            e.gr.cur_buf = 0;
            e.gr.noise_ready = 0;
            e.gr.warm_override = -1;
            int* sh = e.m.stg_hdr + (size_t)e.g * 2 * SH_COUNT;
            sh[SH_STATE] = AZB_FREE;
            sh[SH_COUNT + SH_STATE] = AZB_FREE;
        }
        E::Wave::sync();
        e.new_game();
    }
};
// One engine round = three launches over all games (one wave per game each):
//   OpBackup   expand + backup of the previous leaf batch                       (hot, light)
//   OpEndMove  policy / move / sample / env step / re-root / noise for the games whose budget is met
//              (about 1 game in 18 per round; heavy double-precision code lives only here)
//   OpSelect   the next <= P descents per game + observation planes             (hot, the dominant kernel)
// Keeping the rare, register-hungry end-of-move code out of the two hot kernels keeps them at 4 waves per SIMD.
struct OpBackup {
    const float* priors;
    const float* values;
    template <class E> AZ_HD void operator()(E& e) const {
        e.backup_phase(priors, values);
        e.flush_counters();
    }
};
struct OpEndMove {
    template <class E> AZ_HD void operator()(E& e) const {
        e.endmove_phase();
        e.flush_counters();
    }
};
struct OpSelect {
    void* feat;
    unsigned char* valid;
    template <class E> AZ_HD void operator()(E& e) const {
        e.select(feat, valid);
        e.cnt[AZC_ROUNDS]++;
        e.flush_counters();
    }
};
struct OpBeginMove {
    int warm;
    template <class E> AZ_HD void operator()(E& e) const {
        if (E::Wave::first()) {
            e.gr.noise_ready = 1;
            e.gr.warm_override = warm;
        }
        E::Wave::sync();
    }
};
struct OpCommit {
    const int* moves;
    template <class E> AZ_HD void operator()(E& e) const {
        e.commit_host(moves[e.g]);
        e.flush_counters();
    }
};
struct OpStatus {
    int* status;
    double* q;
    template <class E> AZ_HD void operator()(E& e) const {
        if (E::Wave::first()) {
            int* s = status + (size_t)e.g * 8;
            s[0] = e.gr.status;
            s[1] = e.gr.ply;
            s[2] = e.gr.root_N;
            s[3] = e.gr.n_leaves;
            s[4] = e.gr.out_move;
            s[5] = e.gr.games_done;
            s[6] = e.gr.root_eval_pending;
            s[7] = e.gr.noise_pending;
            q[(size_t)e.g * 2] = e.gr.out_root_q;
            q[(size_t)e.g * 2 + 1] = e.gr.out_child_q;
        }
    }
};
// Drop-in mode (azsp_dropin_step): one iteration of uct_search's simulation loop in ONE launch.  A game is one wave and games never
// interact, so expand / backup -> end of search -> selection of the next leaves -> read-back are simply run back to back by the game's wave
// (the three-launch round of the batched actor exists to keep the rare end-of-move code out of the hot kernels' register budget, which
// does not matter for the handful of games of a drop-in caller).  priors / values may live in page-locked HOST memory (read over PCIe:
// a few hundred bytes); status / q / valid / fault flags / the leaves' observation planes are written straight to page-locked host
// memory: the host needs one launch and one stream synchronisation per call, no copy commands.
struct OpDropinStep {
    const float* priors;  // null: first call of a search (nothing to back up)
    const float* values;
    void* feat;
    unsigned char* valid;
    int* status;               // host-visible: [G][8]
    double* q;                 // host-visible: [G][2]
    int* fault;                // host-visible: [G], the engine fault flags as this game's wave sees them when it is done
    unsigned char* valid_out;  // host-visible: [G * P]
    unsigned char* feat_out;   // host-visible: the first feat_bytes bytes of `feat`
    long long feat_bytes, game_feat_bytes;  // game_feat_bytes: bytes of one game's P feature rows
    template <class E> AZ_HD void operator()(E& e) const {
        if (priors) {
            e.backup_phase(priors, values);
            E::Wave::sync();
            e.endmove_phase();
            E::Wave::sync();
        }
        e.select(feat, valid);
        e.cnt[AZC_ROUNDS]++;
        e.flush_counters();
        E::Wave::sync();
        OpStatus st = {status, q};
        st(e);
        if (E::Wave::first()) fault[e.g] = *e.m.err;
        const int P = e.c.P;
        E::Wave::lanes([&](int lane) {
            if (lane < P) valid_out[(size_t)e.g * P + lane] = valid[(size_t)e.g * P + lane];
        });
        const long long lo = (long long)e.g * game_feat_bytes;
        long long n = feat_bytes - lo;
        n = n < 0 ? 0 : (n > game_feat_bytes ? game_feat_bytes : n);
        const unsigned char* src = (const unsigned char*)feat + lo;
        for (long long b0 = 0; b0 < n; b0 += AZ_WAVE)
            E::Wave::lanes([&](int lane) {
                if (b0 + lane < n) feat_out[lo + b0 + lane] = src[b0 + lane];
            });
    }
};
struct OpRngProbe {
    double* noise;
    double* unif;
    int plies, tries;
    template <class E> AZ_HD void operator()(E& e) const { e.probe_rng(noise, unif, plies, tries); }
};
// packed position uploaded by azsp_set_state: u64 stones[2][W], u64 hist[8][2][W], int scalars[8]
struct OpSetState {
    int slot;
    const u64* packed;
    template <class E> AZ_HD void operator()(E& e) const {
        if (e.g != slot) return;
        typedef typename E::R R;
        typedef typename E::O O;
        const int W = E::W;
        const int* sc = (const int*)(packed + 2 * W + 16 * W);
        typename E::S s;
        for (int q = 0; q < 2; ++q)
            for (int w = 0; w < W; ++w) s.stones[q][w] = packed[q * W + w];
        s.to_play = (uint8_t)sc[0];
        s.steps = (int16_t)sc[1];
        s.ko = (int16_t)sc[2];
        s.flags = sc[3] ? AZF_LASTPASS : 0;
        s.caps[0] = (uint16_t)sc[4];
        s.caps[1] = (uint16_t)sc[5];
        s.winner = -1;
        s.reward = 0;
        s.area[0] = s.area[1] = 0;
        const typename E::B sb = R::ld(s.stones[0]), sw = R::ld(s.stones[1]);
        const typename E::B own = s.to_play == 0 ? sb : sw, opp = s.to_play == 0 ? sw : sb;
        const typename E::B legal = E::GAME_ID == AZ_GO ? R::go_legal(own, opp, s.ko) : O::inv(O::bor(own, opp));
        R::st(s.legal, legal);
        if (E::Wave::first()) {
            e.gr.env = s;
            for (int k = 0; k < 8; ++k)
                for (int q = 0; q < 2; ++q)
                    for (int w = 0; w < W; ++w) e.gr.hist[k][q][w] = packed[2 * W + (k * 2 + q) * W + w];
            e.gr.ply = 0;
            e.gr.num_passes = 0;
            e.gr.marked_player = -1;
            e.gr.n_leaves = 0;
            e.gr.root_eval_pending = 0;
            e.gr.noise_pending = 0;
            e.gr.noise_ready = 0;
            e.gr.status = AZS_NEED_ROOT;
        }
        E::Wave::sync();
        e.free_all_nodes();
    }
};
struct OpEnvStep {
    const int* actions;
    int8_t* board;
    int8_t* legal;
    int* scalars;
    int8_t* obs;
    template <class E> AZ_HD void operator()(E& e) const {
        typedef typename E::R R;
        typedef typename E::S S;
        const int NP = E::NP, A = E::A;
        const bool go = E::GAME_ID == AZ_GO;
        const int a = actions ? actions[e.g] : -2;
        int illegal = 0;
        if (a != -2) {
            const S cur = e.gr.env;
            S ns;
            if (cur.flags & AZF_TERMINAL) illegal = 1;                    // RuntimeError('Game is over') go.py:90-91
            else if (a == -1 && go) R::go_resign(cur, ns);
            else if (a < 0 || a >= A) illegal = 2;                        // ValueError('Invalid action') go.py:92-93
            else if (!e.action_legal(cur, a)) illegal = 3;                // ValueError('Illegal action') go.py:94-95
            else R::template step<E::GAME_ID>(cur, a, e.c.rc, ns);
            if (!illegal) {
                if (E::Wave::first()) {
                    e.gr.env = ns;
                    e.gr.ply += 1;
                }
                E::Wave::sync();
                e.push_history(ns);
            }
        }
        const S& s = e.gr.env;
        const int b_id = 1, w_id = go ? -1 : 2;
        E::Wave::lanes([&](int lane) {
            for (int p = lane; p < NP; p += AZ_WAVE) {
                const bool bk = (s.stones[0][p >> 6] >> (p & 63)) & 1ull, wh = (s.stones[1][p >> 6] >> (p & 63)) & 1ull;
                if (board) board[(size_t)e.g * NP + p] = (int8_t)(bk ? b_id : (wh ? w_id : 0));
            }
            if (legal)
                for (int x = lane; x < A; x += AZ_WAVE) legal[(size_t)e.g * A + x] = e.action_legal(s, x) ? 1 : 0;
        });
        if (scalars) {
            int ab = 0, aw = 0;
            if (go) R::go_area(R::ld(s.stones[0]), R::ld(s.stones[1]), ab, aw);
            if (E::Wave::first()) {
                int* o = scalars + (size_t)e.g * 12;
                o[0] = s.ko;
                o[1] = s.caps[0];
                o[2] = s.caps[1];
                o[3] = s.steps;
                o[4] = s.to_play == 0 ? b_id : w_id;
                o[5] = (s.flags & AZF_TERMINAL) ? 1 : 0;
                o[6] = s.reward;
                o[7] = s.winner < 0 ? 0 : (s.winner == 0 ? b_id : w_id);
                o[8] = ab;
                o[9] = aw;
                o[10] = illegal;
                o[11] = (s.flags & AZF_LASTPASS) ? 1 : 0;
            }
        }
        if (obs) {
            e.gather_planes(-1, -1, s.to_play);
            int8_t* out = obs + (size_t)e.g * 17 * NP;
            e.template emit_planes<int8_t>(out, (int8_t)1, s.to_play);
        }
        E::Wave::sync();
    }
};
// Harvest = three launches, deterministic: (1) OpHarvestCount publishes the length of every COMPLETE staging buffer, (2) a scan
// (az_harvest_scan_range) hands out output rows in buffer order starting from a slot that rotates from harvest to harvest -- the
// buffers that fit the caller's capacity form a prefix of that order, the rest stay COMPLETE for the next call -- (3) OpHarvest copies.
// Same seed and same call sequence => the same games in the same rows (round 2 reserved rows with a CAS race: correct, but which
// games a full harvest kept back, and the row order, depended on the order the waves arrived).
struct OpHarvestCount {
    int* len;  // [G][2]
    template <class E> AZ_HD void operator()(E& e) const {
        if (E::Wave::first())
            for (int b = 0; b < 2; ++b) {
                const int* sh = e.m.stg_hdr + ((size_t)e.g * 2 + b) * SH_COUNT;
                len[(size_t)e.g * 2 + b] = sh[SH_STATE] == AZB_COMPLETE ? sh[SH_LEN] : 0;
            }
    }
};
// element i of the rotated order (i in [lo, hi)) with the exclusive prefix (ps samples, pg games) of everything before lo
AZ_HD void az_harvest_scan_range(const int* len, int* ofs, int n2, int rot, int cap, int max_games, int lo, int hi, int ps, int pg, int* best) {
    for (int i = lo; i < hi; ++i) {
        const int idx = (i + rot) % n2, l = len[idx];
        int st = -1, gi = -1;
        if (l > 0) {
            if (ps + l <= cap && pg < max_games) {  // (a buffer that does not fit keeps later ones out too: the prefix sums run on)
                st = ps;
                gi = pg;
                best[0] = ps + l;
                best[1] = pg + 1;
            }
            ps += l;
            pg += 1;
        }
        ofs[2 * idx] = st;
        ofs[2 * idx + 1] = gi;
    }
}
struct OpHarvest {
    int8_t* states;
    float* pi;
    float* z;
    int cap;
    int* games;
    int max_games;
    const int* ofs;   // [G][2][2] = (first output row, game index) of a staging buffer, -1 = not this time
    int16_t* moves;   // optional: the move played from every sample's position (azsp_harvest_moves)
    int* extra;       // [max_games][4] = {training_steps at game end, resign threshold double bits lo, hi, straddled a weight swap}
    template <class E> AZ_HD void operator()(E& e) const {
        const int NP = E::NP, A = E::A, W = E::W;
        const bool go = E::GAME_ID == AZ_GO;
        const int b_id = 1, w_id = go ? -1 : 2;
        for (int b = 0; b < 2; ++b) {
            int* sh = e.m.stg_hdr + ((size_t)e.g * 2 + b) * SH_COUNT;
            if (sh[SH_STATE] != AZB_COMPLETE) continue;
            const int len = sh[SH_LEN];
            int start = -1, gi = 0;
            if (E::Wave::first()) {
                start = ofs[((size_t)e.g * 2 + b) * 2];
                gi = ofs[((size_t)e.g * 2 + b) * 2 + 1];
            }
            start = E::Wave::bcast0(start);
            gi = E::Wave::bcast0(gi);
            if (start < 0) continue;
            const int reward = sh[SH_REWARD], last_player = sh[SH_LAST_PLAYER];
            for (int k = 0; k < len; ++k) {
                const size_t idx = ((size_t)e.g * 2 + b) * e.c.stage_cap + k;
                const u64* pl = e.m.stg_planes + idx * 16 * W;
                const float* spi = e.m.stg_pi + idx * A;
                const int black = e.m.stg_meta[idx];
                int8_t* so = states + (size_t)(start + k) * 17 * NP;
                float* po = pi + (size_t)(start + k) * A;
                E::Wave::lanes([&](int lane) {
                    for (int x = lane; x < 17 * NP; x += AZ_WAVE) {
                        const int plane = x / NP, p = x - plane * NP;
                        so[x] = plane < 16 ? (int8_t)((pl[plane * W + (p >> 6)] >> (p & 63)) & 1ull) : (int8_t)black;
                    }
                    for (int x = lane; x < A; x += AZ_WAVE) po[x] = spi[x];
                });
                if (E::Wave::first()) {
                    // pipeline.py:349-354: z = reward for the samples of the last player, -reward for the others
                    const int mover = black ? 0 : 1;
                    z[start + k] = reward == 0 ? 0.0f : (mover == last_player ? (float)reward : (float)-reward);
                    if (moves) moves[start + k] = e.m.stg_move[idx];
                }
            }
            if (E::Wave::first()) {
                int* o = games + (size_t)gi * 16;
                o[0] = start;
                o[1] = len;
                o[2] = sh[SH_WINNER] < 0 ? 0 : (sh[SH_WINNER] == 0 ? b_id : w_id);
                o[3] = sh[SH_AREA_B];
                o[4] = sh[SH_AREA_W];
                o[5] = sh[SH_PASSES];
                o[6] = sh[SH_RESIGNED];
                o[7] = sh[SH_RESIGN_DISABLED];
                o[8] = sh[SH_MARKED];
                o[9] = sh[SH_COULD_WON];
                o[10] = sh[SH_MARKED_PLAYER] < 0 ? 0 : (sh[SH_MARKED_PLAYER] == 0 ? b_id : w_id);
                o[11] = sh[SH_UID];
                o[12] = sh[SH_TRAINING_STEPS];
                int* x = extra + (size_t)gi * 4;
                x[0] = sh[SH_TS_END];
                x[1] = sh[SH_THR_LO];
                x[2] = sh[SH_THR_HI];
                x[3] = sh[SH_TS_END] != sh[SH_TRAINING_STEPS] ? 1 : 0;
                o[13] = reward;
                o[14] = last_player == 0 ? b_id : w_id;
                o[15] = e.g;
                sh[SH_STATE] = AZB_FREE;
            }
            E::Wave::sync();
        }
    }
};

// Dihedral-8 gather (utils/transformation.py:34-110): flat element kernel, pure permutation of bytes.
struct DihedralArgs {
    const unsigned char* sin;
    unsigned char* sout;
    const unsigned char* pin;
    unsigned char* pout;
    int ses, pes, batch, ch, n, A, op;
};
AZ_HD int az_dihedral_src(int op, int n, int i, int j) {
    int r, c;
    switch (op) {
        case 1: r = i; c = n - 1 - j; break;          // hflip: reverse the last dim (:46)
        case 2: r = n - 1 - i; c = j; break;          // vflip (:71)
        case 3: r = j; c = n - 1 - i; break;          // rot90 counter-clockwise (transformation_test.py:231-274)
        case 4: r = n - 1 - i; c = n - 1 - j; break;  // rot180
        case 5: r = n - 1 - j; c = i; break;          // rot270
        case 6: r = j; c = i; break;                  // transpose (extra)
        case 7: r = n - 1 - j; c = n - 1 - i; break;  // anti-transpose (extra)
        default: r = i; c = j; break;
    }
    return r * n + c;
}
AZ_HD void az_copy_elem(unsigned char* d, const unsigned char* s, int es) {
    for (int b = 0; b < es; ++b) d[b] = s[b];
}
AZ_HD void az_dihedral_elem(const DihedralArgs& a, long long t) {
    const int np = a.n * a.n;
    const long long ns = (long long)a.batch * a.ch * np;
    if (t < ns) {
        if (!a.sin) return;
        const int p = (int)(t % np);
        const long long base = t - p;
        const int src = az_dihedral_src(a.op, a.n, p / a.n, p % a.n);
        az_copy_elem(a.sout + t * a.ses, a.sin + (base + src) * a.ses, a.ses);
    } else {
        t -= ns;
        if (!a.pin || t >= (long long)a.batch * a.A) return;
        const int x = (int)(t % a.A);
        const long long base = t - x;
        const int src = x < np ? az_dihedral_src(a.op, a.n, x / a.n, x % a.n) : x;  // pass column untouched
        az_copy_elem(a.pout + t * a.pes, a.pin + (base + src) * a.pes, a.pes);
    }
}

// Replay sampling (core/replay.py:72-83 sample, core/pipeline.py:634-643 tensors + apply_random_transformation): one flat
// element kernel gathers the sampled rows of the HBM ring, applies one dihedral op to the whole batch and casts the planes.
struct ReplayGatherArgs {
    const int8_t* rs;   // ring states [capacity][ch][n][n]
    const float* rp;    // ring pi [capacity][A]
    const float* rz;    // ring z [capacity]
    const long long* idx;
    void* os;
    float* op_;
    float* oz;
    int batch, ch, n, A, op, dtype;
};
AZ_HD uint16_t az_f32_to_bf16(float f);
AZ_HD uint16_t az_f32_to_f16(float f);
AZ_HD void az_replay_gather_elem(const ReplayGatherArgs& a, long long t) {
    const int np = a.n * a.n;
    const long long per = (long long)a.ch * np, ns = (long long)a.batch * per;
    if (t < ns) {
        const long long b = t / per;
        const int e = (int)(t - b * per), p = e % np;
        const int src = az_dihedral_src(a.op, a.n, p / a.n, p % a.n);
        const int8_t v = a.rs[a.idx[b] * per + (e - p) + src];
        switch (a.dtype) {
            case 0: ((int8_t*)a.os)[t] = v; break;
            case 1: ((float*)a.os)[t] = (float)v; break;
            case 2: ((uint16_t*)a.os)[t] = az_f32_to_bf16((float)v); break;
            default: ((uint16_t*)a.os)[t] = az_f32_to_f16((float)v); break;
        }
        return;
    }
    t -= ns;
    if (t < (long long)a.batch * a.A) {
        const long long b = t / a.A;
        const int x = (int)(t - b * a.A);
        const int src = x < np ? az_dihedral_src(a.op, a.n, x / a.n, x % a.n) : x;  // pass column untouched
        a.op_[t] = a.rp[a.idx[b] * a.A + src];
        return;
    }
    t -= (long long)a.batch * a.A;
    if (t < a.batch) a.oz[t] = a.rz[a.idx[t]];
}

// Fused conv epilogue: 8 consecutive channels (16 B of bf16/fp16, 32 B of fp32) per work item, fp32 math, one rounding.
struct BiasActArgs {
    void* y;
    const void* bias;
    const void* res;
    long long nvec;  // rows * channels / 8
    int channels, dtype, relu;
};
AZ_HD float az_bf16_to_f32(uint16_t h) {
    union { u32 u; float f; } v;
    v.u = (u32)h << 16;
    return v.f;
}
AZ_HD uint16_t az_f32_to_bf16(float f) {  // round to nearest even
    union { u32 u; float f; } v;
    v.f = f;
    const u32 r = v.u + 0x7fffu + ((v.u >> 16) & 1u);
    return (uint16_t)(r >> 16);
}
AZ_HD float az_f16_to_f32(uint16_t h) {
    union { u32 u; float f; } v;
    const u32 sign = (u32)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    if (exp == 0) {  // zero / subnormal: man * 2^-24
        v.f = (float)man * (1.0f / 16777216.0f);
        v.u |= sign;
        return v.f;
    }
    v.u = sign | (exp == 31 ? 0x7f800000u : ((exp + 112u) << 23)) | (man << 13);
    return v.f;
}
AZ_HD uint16_t az_f32_to_f16(float f) {  // round to nearest even, overflow -> inf
    union { u32 u; float f; } v;
    v.f = f;
    const u32 sign = (v.u >> 16) & 0x8000u, x = v.u & 0x7fffffffu;
    if (x >= 0x47800000u) return (uint16_t)(sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (x < 0x38800000u) {  // subnormal half (or zero): units of 2^-24, round to nearest even via the fp32 adder
        union { u32 u; float f; } t;
        t.u = x;
        t.f = t.f + 0.5f;   // aligns the 2^-24 grid with the fp32 mantissa lsb of values in [0.5, 1)
        return (uint16_t)(sign | (t.u - 0x3f000000u));
    }
    const u32 r = x + 0xfffu + ((x >> 13) & 1u);
    return (uint16_t)(sign | ((r - 0x38000000u) >> 13));
}
AZ_HD void az_bias_act_vec(const BiasActArgs& a, long long i) {
    const long long e0 = i * 8;
    const int c0 = (int)(e0 % a.channels);
    float v[8];
    if (a.dtype == AZSP_FEAT_BF16) {
        const uint16_t* y = (const uint16_t*)a.y + e0;
        const uint16_t* b = (const uint16_t*)a.bias + c0;
        const uint16_t* r = a.res ? (const uint16_t*)a.res + e0 : nullptr;
        struct alignas(16) V8 { uint16_t h[8]; };
        const V8 yv = *(const V8*)y, bv = *(const V8*)b;
        V8 rv = yv, ov;
        if (r) rv = *(const V8*)r;
        for (int k = 0; k < 8; ++k) {
            v[k] = az_bf16_to_f32(yv.h[k]) + az_bf16_to_f32(bv.h[k]) + (r ? az_bf16_to_f32(rv.h[k]) : 0.0f);
            if (a.relu && v[k] < 0.0f) v[k] = 0.0f;
            ov.h[k] = az_f32_to_bf16(v[k]);
        }
        *(V8*)((uint16_t*)a.y + e0) = ov;
    } else if (a.dtype == AZSP_FEAT_F16) {
        struct alignas(16) V8 { uint16_t h[8]; };
        const V8 yv = *(const V8*)((const uint16_t*)a.y + e0), bv = *(const V8*)((const uint16_t*)a.bias + c0);
        V8 rv = yv, ov;
        if (a.res) rv = *(const V8*)((const uint16_t*)a.res + e0);
        for (int k = 0; k < 8; ++k) {
            v[k] = az_f16_to_f32(yv.h[k]) + az_f16_to_f32(bv.h[k]) + (a.res ? az_f16_to_f32(rv.h[k]) : 0.0f);
            if (a.relu && v[k] < 0.0f) v[k] = 0.0f;
            ov.h[k] = az_f32_to_f16(v[k]);
        }
        *(V8*)((uint16_t*)a.y + e0) = ov;
    } else {
        float* y = (float*)a.y + e0;
        const float* b = (const float*)a.bias + c0;
        const float* r = a.res ? (const float*)a.res + e0 : nullptr;
        for (int k = 0; k < 8; ++k) {
            float t = y[k] + b[k] + (r ? r[k] : 0.0f);
            if (a.relu && t < 0.0f) t = 0.0f;
            y[k] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// engine handle
// ------------------------------------------------------------------------------------------------
struct AzHandle {
    AzspConfig pub;
    AzCfg cfg;
    AzMem mem;
    std::string err;
    std::vector<void*> allocs;
    int A, AP, W, NP, REC, GREC;
    long long bytes;
    int* d_status;
    double* d_q;
    int* d_moves;
    u64* d_packed;
    int* d_hcounts;
    int* d_hlen = nullptr;   // [G][2] lengths of the COMPLETE staging buffers (azsp_harvest)
    int* d_hofs = nullptr;   // [G][2][2] output rows / game indices handed out by the scan
    unsigned harvest_seq = 0;  // rotates the slot a harvest starts from
    int* d_games;
    int d_games_cap;
    int16_t* harvest_moves = nullptr;  // optional per-sample move output of azsp_harvest (azsp_harvest_moves)
    int* d_gextra = nullptr;           // [d_games_cap][4] per-game extras (azsp_harvest_extra)
    int32_t* harvest_extra = nullptr;  // host destination of the extras, optional
    unsigned char* pin = nullptr;      // page-locked staging of azsp_dropin_step (one packed upload + one packed read-back per call)
    size_t pin_bytes = 0;
};

namespace azb {  // implemented by the backend translation unit
void* alloc(size_t n);
void release(void* p);
int h2d(void* dst, const void* src, size_t n, void* stream);
int d2h(void* dst, const void* src, size_t n, void* stream);
int zero(void* dst, size_t n, void* stream);
int sync(void* stream);
void* host_alloc(size_t n);  // page-locked host memory (staging of azsp_dropin_step)
void host_release(void* p);
int set_device(int dev);
const char* backend_error();
template <int N, int GAME, class Op> int launch(const AzCfg& c, const AzMem& m, const Op& op, void* stream, int g0, int g1);  // games [g0, g1)
int launch_dihedral(const DihedralArgs& a, long long total, void* stream);
int launch_bias_act(const BiasActArgs& a, void* stream);
int launch_replay_gather(const ReplayGatherArgs& a, long long total, void* stream);
// exclusive scan of the staging-buffer lengths in rotated buffer order -> ofs[n2][2], counts[0..1] = samples / games handed out
int launch_harvest_scan(const int* len, int* ofs, int* counts, int n2, int rot, int cap, int max_games, void* stream);
// returns 0 ok, 1 unsupported shape, -1 launch error
// f16 = 1: activations and weights are f16 instead of bf16 (same layouts, same kernels, the f16 MFMA / conversions: CvFmt in az_conv.h)
int launch_conv3x3_tiled(const void* x, const void* w, const float* bias, const void* res, void* y, long long boards, int S, int C,
                         int relu, void* stream, int f16 = 0);
int launch_resblock_tiled(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, long long boards, int S, int C,
                          void* stream);
int launch_tile_layout(const void* src, void* dst, long long boards, int S, int C, int to_tiled, void* stream);
// range: the caller's range record (two device words: events, bits of the largest |v|), null = the per-device default record
int launch_split_layout(const void* src, void* dst, long long boards, int S, int C, int to_split, void* stream, unsigned* range);
int launch_conv3x3_split(const void* x, const void* w, const float* bias, const void* res, void* y, long long boards, int S, int C, int relu,
                         void* stream, unsigned* range);
// one whole ResNetBlock on the split layout in one launch (17x17 x 64: az_resblock_sp17.h); 1 = unsupported shape
int launch_resblock_split(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, long long boards, int S, int C,
                          void* stream, unsigned* range);
struct HeadSplitArgs {
    const void* x;
    const float *hw, *hb, *wp_t, *bp, *w1_t, *b1, *w2;
    float b2;
    float *priors, *values;
    long long boards;
    int S, C, A, F, npol;
};
int launch_split_features(const float* src, void* dst, long long boards, int S, int cin, void* stream, unsigned* range);
int launch_stem_split(const void* x, const void* w, const float* bias, void* y, long long boards, int S, int C, int pad, int relu, void* stream,
                      int x_lo_zero, unsigned* range);
int split_range_read(const unsigned* rec, unsigned out[2], int reset, void* stream);  // rec null = the default record; synchronises the stream
long long small_batch_waves(long long n);  // n >= 0 sets the tile-wave count up to which the wave-per-tile kernel runs; returns the previous value
int launch_head_split(const HeadSplitArgs& a, void* stream);
int launch_stem_tiled(const void* x, const void* w, const float* bias, void* y, long long boards, int S, int C, int pad, int relu, void* stream,
                      int f16 = 0);
int launch_head_tiled(const void* x, const float* w, const float* bias, void* pol, void* val, long long boards, int S, int C, int npol, int nval,
                      int pol_stride, int val_stride, void* stream, int f16 = 0);
struct FcHeadsArgs {
    const void *pol, *val, *wp, *w1;  // bf16: head planes (rows ks * 16 elements apart) and zero-padded weights [NT * 32][ks * 16]
    const float *bp, *b1, *w2;        // fp32, padded to NT * 32
    float b2;
    float *priors, *values;
    long long boards;
    int ks1, ks2, A, F;
    int f16;  // head planes and weights are f16 instead of bf16
};
int launch_fc_heads(const FcHeadsArgs& a, void* stream);
}  // namespace azb

template <class T> static T* az_new(AzHandle* h, size_t count) {
    size_t n = count * sizeof(T);
    if (n == 0) n = sizeof(T);
    void* p = azb::alloc(n);
    if (!p) return nullptr;
    h->allocs.push_back(p);
    h->bytes += (long long)n;
    return (T*)p;
}

template <class Op> static int az_run(AzHandle* h, const Op& op, void* stream, int g0 = 0, int g1 = -1) {
    int rc = AZSP_EINVAL;
    if (g1 < 0) g1 = h->cfg.G;
    switch (h->cfg.game * 100 + h->cfg.n) {
#define AZ_CASE(NN, GG) \
    case (GG) * 100 + (NN): rc = azb::launch<NN, GG, Op>(h->cfg, h->mem, op, stream, g0, g1); break;
        AZ_FOR_EACH_VARIANT(AZ_CASE)
#undef AZ_CASE
        default: break;
    }
    if (rc != 0) {
        h->err = std::string("kernel launch failed: ") + azb::backend_error();
        return AZSP_EDEVICE;
    }
    return AZSP_OK;
}

#ifndef AZ_GEOM_WAVE
#define AZ_GEOM_WAVE WaveDev  // only the policy-independent constants of Engine<> are used here
#endif
static int az_geometry_of(int game, int n, int* A, int* AP, int* W, int* REC, int* GREC) {
    switch (game * 100 + n) {
#define AZ_CASE(NN, GG)                              \
    case (GG) * 100 + (NN): {                        \
        typedef Engine<AZ_GEOM_WAVE, NN, GG> E;       \
        *A = E::A; *AP = E::AP; *W = E::W; *REC = E::REC; *GREC = E::GREC; \
        return 0;                                    \
    }
        AZ_FOR_EACH_VARIANT(AZ_CASE)
#undef AZ_CASE
        default: return -1;
    }
}

static int az_check_engine_fault(AzHandle* h, void* stream) {
    int e = 0;
    if (azb::d2h(&e, h->mem.err, sizeof(int), stream) != 0) {
        h->err = std::string("copy failed: ") + azb::backend_error();
        return AZSP_EDEVICE;
    }
    if (e) {
        char buf[160];
        snprintf(buf, sizeof buf, "engine fault flags 0x%x (1=node pool exhausted, 2=tree deeper than %d, 4=move sampling, 8=staging full)", e,
                 AZ_PATH_CAP);
        h->err = buf;
        return AZSP_EENGINE;
    }
    return AZSP_OK;
}

extern "C" {

int azsp_create(const AzspConfig* p, void** out) {
    if (!p || !out) return AZSP_EINVAL;
    *out = nullptr;
    AzHandle* h = new AzHandle();
    h->pub = *p;
    h->bytes = 0;
    if (az_geometry_of(p->game, p->board_size, &h->A, &h->AP, &h->W, &h->REC, &h->GREC) != 0 || p->num_games < 1 ||
        p->num_parallel < 1 || p->num_parallel > AZ_MAXP || p->num_simulations < 1) {
        delete h;
        return AZSP_EINVAL;
    }
    if (azb::set_device(p->device) != 0) {
        delete h;
        return AZSP_EDEVICE;
    }
    h->NP = p->board_size * p->board_size;
    AzCfg& c = h->cfg;
    memset(&c, 0, sizeof c);
    c.game = p->game;
    c.n = p->board_size;
    c.G = p->num_games;
    c.P = p->num_parallel;
    c.sims = p->num_simulations;
    c.parallel_mode = p->num_parallel > 1 ? 1 : 0;                    // pipeline.py:132
    c.budget = p->num_simulations + (c.parallel_mode ? c.P : 0);      // mcts_v2.py:378 / :568
    c.max_nodes = p->max_nodes > 0 ? p->max_nodes : c.budget + 2 * c.P + 8;
    if (c.max_nodes > 32000) {
        delete h;
        return AZSP_EINVAL;
    }
    c.root_noise = p->root_noise;
    c.deterministic = p->deterministic;
    c.reuse_tree = p->reuse_tree;
    c.warm_up_steps = p->warm_up_steps;
    c.has_resign = (p->game == AZSP_GAME_GO && p->has_resign) ? 1 : 0;  // env.has_resign_move; the threshold itself is per game (azsp_set_actor_state)
    c.check_resign_after = p->check_resign_after_steps;
    c.force_resign_disabled = p->force_resign_disabled;
    c.inject = p->inject_random;
    c.inj_moves = p->inject_moves > 0 ? p->inject_moves : 1;
    c.stop_after_move = p->stop_after_move;
    c.max_plies = p->max_plies;
    c.stop_at_game_end = p->stop_at_game_end;
    c.feat_dtype = p->feature_dtype;
    c.log_moves = p->log_moves;
    c.log_cap = (p->log_moves && p->log_capacity > 0) ? p->log_capacity : 1;
    c.tab_len = c.budget + 3 * c.P + 16;
    c.training_steps = p->training_steps;
    c.one_minus_eps_f32 = (float)(1.0 - p->dirichlet_eps);            // child_P * (1 - eps): float32 row times Python float
    c.disable_resign_ratio = p->disable_resign_ratio;
    c.eps = p->dirichlet_eps;
    c.alpha = p->dirichlet_alpha;
    c.resign_threshold = p->resign_threshold;
    c.rc.max_steps = p->max_steps > 0 ? p->max_steps : 2 * h->NP;     // go.py:48
    c.rc.num_to_win = p->num_to_win > 0 ? p->num_to_win : 5;
    c.rc.komi = p->komi;
    c.seed = p->seed;
    c.rank = p->rank;
    c.stage_cap = p->stop_after_move ? 1 : (p->game == AZSP_GAME_GO ? c.rc.max_steps : h->NP);

    const size_t G = (size_t)c.G;
    AzMem& m = h->mem;
    memset(&m, 0, sizeof m);
    m.nodes = az_new<unsigned char>(h, G * c.max_nodes * h->REC);
    m.games = az_new<unsigned char>(h, G * h->GREC);
    m.rootP = az_new<double>(h, G * h->AP);
    m.free_stack = az_new<int16_t>(h, G * c.max_nodes);
    m.leaf_path = az_new<int>(h, G * c.P * AZ_PATH_CAP);
    m.pbc_np = az_new<double>(h, c.tab_len);
    m.pbc_py = az_new<double>(h, c.tab_len);
    m.sqrt32 = az_new<float>(h, c.tab_len);
    m.inj_noise = az_new<double>(h, c.inject ? G * c.inj_moves * h->A : (c.stop_after_move ? G * h->A : 1));
    m.inj_unif = az_new<double>(h, c.inject ? G * c.inj_moves * AZ_INJ_K : 1);
    m.stg_planes = az_new<u64>(h, G * 2 * c.stage_cap * 16 * h->W);
    m.stg_pi = az_new<float>(h, G * 2 * c.stage_cap * h->A);
    m.stg_meta = az_new<unsigned char>(h, G * 2 * c.stage_cap);
    m.stg_move = az_new<int16_t>(h, G * 2 * c.stage_cap);
    m.stg_hdr = az_new<int>(h, G * 2 * SH_COUNT);
    m.log_pi = az_new<double>(h, G * c.log_cap * h->A);
    m.log_childN = az_new<float>(h, G * c.log_cap * h->A);
    m.log_q = az_new<double>(h, G * c.log_cap * 4);
    m.counters = az_new<u64>(h, G * AZC_COUNT);
    m.err = az_new<int>(h, 4);
    h->d_status = az_new<int>(h, G * 8);
    h->d_q = az_new<double>(h, G * 2);
    h->d_moves = az_new<int>(h, G);
    h->d_packed = az_new<u64>(h, 18 * h->W + 8);
    h->d_hcounts = az_new<int>(h, 4);
    h->d_hlen = az_new<int>(h, 2 * G);
    h->d_hofs = az_new<int>(h, 4 * G);
    h->d_games_cap = (int)(2 * G);
    h->d_games = az_new<int>(h, (size_t)h->d_games_cap * 16);
    h->d_gextra = az_new<int>(h, (size_t)h->d_games_cap * 4);
    if (!m.nodes || !m.games || !m.rootP || !m.free_stack || !m.leaf_path || !m.stg_planes || !m.stg_pi || !m.log_pi ||
        !h->d_games || !h->d_gextra || !m.err || !h->d_hlen || !h->d_hofs) {
        for (void* q : h->allocs) azb::release(q);
        delete h;
        return AZSP_ENOMEM;
    }
    // the injected-noise path is also how drop-in mode receives its per-call Dirichlet draw
    if (c.stop_after_move && !c.inject) {
        c.inject = 1;
        c.inj_moves = 1;
    }
    *out = h;
    int rc = az_run(h, OpReset(), nullptr);
    if (rc == 0) rc = azb::sync(nullptr) == 0 ? 0 : AZSP_EDEVICE;
    return rc;
}

int azsp_destroy(void* e) {
    if (!e) return AZSP_EINVAL;
    AzHandle* h = (AzHandle*)e;
    azb::sync(nullptr);
    for (void* q : h->allocs) azb::release(q);
    if (h->pin) azb::host_release(h->pin);
    delete h;
    return AZSP_OK;
}

const char* azsp_last_error(void* e) { return e ? ((AzHandle*)e)->err.c_str() : "null engine"; }

int azsp_geometry(void* e, AzspGeometry* g) {
    if (!e || !g) return AZSP_EINVAL;
    AzHandle* h = (AzHandle*)e;
    g->num_actions = h->A;
    g->num_points = h->NP;
    g->planes = 17;
    g->batch_rows = h->cfg.G * h->cfg.P;
    g->max_nodes = h->cfg.max_nodes;
    g->budget = h->cfg.budget;
    g->table_len = h->cfg.tab_len;
    g->stage_capacity = h->cfg.stage_cap;
    g->device_bytes = h->bytes;
    g->node_record_bytes = h->REC;
    g->reserved = 0;
    return AZSP_OK;
}

int azsp_set_tables(void* e, const double* a, const double* b, const float* s, int32_t len) {
    AzHandle* h = (AzHandle*)e;
    if (!h || !a || !b || !s || len != h->cfg.tab_len) return AZSP_EINVAL;
    if (azb::h2d((void*)h->mem.pbc_np, a, sizeof(double) * len, nullptr) || azb::h2d((void*)h->mem.pbc_py, b, sizeof(double) * len, nullptr) ||
        azb::h2d((void*)h->mem.sqrt32, s, sizeof(float) * len, nullptr))
        return AZSP_EDEVICE;
    return AZSP_OK;
}

int azsp_set_injection(void* e, const double* noise, const double* unif, int32_t moves) {
    AzHandle* h = (AzHandle*)e;
    if (!h || !h->pub.inject_random || moves != h->cfg.inj_moves) return AZSP_EINVAL;
    const size_t G = h->cfg.G;
    if (noise && azb::h2d((void*)h->mem.inj_noise, noise, sizeof(double) * G * moves * h->A, nullptr)) return AZSP_EDEVICE;
    if (unif && azb::h2d((void*)h->mem.inj_unif, unif, sizeof(double) * G * moves * AZ_INJ_K, nullptr)) return AZSP_EDEVICE;
    return AZSP_OK;
}

int azsp_reset_games(void* e, void* stream) {
    AzHandle* h = (AzHandle*)e;
    if (!h) return AZSP_EINVAL;
    return az_run(h, OpReset(), stream);
}

int azsp_env_step(void* e, const int32_t* actions, int8_t* board, int8_t* legal, int32_t* scalars, int8_t* obs, void* stream) {
    AzHandle* h = (AzHandle*)e;
    if (!h) return AZSP_EINVAL;
    OpEnvStep op = {actions, board, legal, scalars, obs};
    return az_run(h, op, stream);
}

int azsp_set_state(void* e, int32_t slot, const int8_t* board, const int8_t* hist, int32_t to_play, int32_t steps, int32_t ko,
                   int32_t last_was_pass, int32_t caps_b, int32_t caps_w, void* stream) {
    AzHandle* h = (AzHandle*)e;
    if (!h || !board || !hist || slot < 0 || slot >= h->cfg.G) return AZSP_EINVAL;
    const int W = h->W, NP = h->NP;
    const int w_id = h->cfg.game == AZ_GO ? -1 : 2;
    std::vector<u64> pk(18 * W + 8, 0);
    for (int p = 0; p < NP; ++p) {
        if (board[p] == 1) pk[0 * W + (p >> 6)] |= 1ull << (p & 63);
        else if (board[p] == w_id) pk[1 * W + (p >> 6)] |= 1ull << (p & 63);
    }
    for (int k = 0; k < 8; ++k)
        for (int p = 0; p < NP; ++p) {
            const int8_t v = hist[k * NP + p];
            if (v == 1) pk[2 * W + (k * 2 + 0) * W + (p >> 6)] |= 1ull << (p & 63);
            else if (v == w_id) pk[2 * W + (k * 2 + 1) * W + (p >> 6)] |= 1ull << (p & 63);
        }
    int* sc = (int*)(pk.data() + 18 * W);
    sc[0] = to_play == 1 ? 0 : 1;
    sc[1] = steps;
    sc[2] = ko;
    sc[3] = last_was_pass;
    sc[4] = caps_b;
    sc[5] = caps_w;
    if (azb::h2d(h->d_packed, pk.data(), pk.size() * sizeof(u64), stream)) return AZSP_EDEVICE;
    OpSetState op = {slot, h->d_packed};
    return az_run(h, op, stream);
}

int azsp_begin_move(void* e, const double* noise, int32_t warm_up, void* stream) {
    AzHandle* h = (AzHandle*)e;
    if (!h) return AZSP_EINVAL;
    if (noise && azb::h2d((void*)h->mem.inj_noise, noise, sizeof(double) * (size_t)h->cfg.G * h->A, stream)) return AZSP_EDEVICE;
    OpBeginMove op = {warm_up};
    return az_run(h, op, stream);
}

int azsp_select(void* e, void* feat, uint8_t* valid, void* stream) {
    AzHandle* h = (AzHandle*)e;
    if (!h || !feat || !valid) return AZSP_EINVAL;
    OpSelect op = {feat, valid};
    return az_run(h, op, stream);
}

int azsp_expand_backup(void* e, const float* priors, const float* values, void* stream) {
    AzHandle* h = (AzHandle*)e;
    if (!h || !priors || !values) return AZSP_EINVAL;
    OpBackup op = {priors, values};
    int rc = az_run(h, op, stream);
    if (rc) return rc;
    return az_run(h, OpEndMove(), stream);
}

static int az_range_ok(AzHandle* h, int32_t g0, int32_t g1) { return h && g0 >= 0 && g0 < g1 && g1 <= h->cfg.G && g0 % 32 == 0; }

int azsp_select_range(void* e, void* feat, uint8_t* valid, int32_t g0, int32_t g1, void* stream) {
    AzHandle* h = (AzHandle*)e;
    if (!h || !feat || !valid || !az_range_ok(h, g0, g1)) return AZSP_EINVAL;
    OpSelect op = {feat, valid};
    return az_run(h, op, stream, g0, g1);
}

int azsp_expand_backup_range(void* e, const float* priors, const float* values, int32_t g0, int32_t g1, void* stream) {
    AzHandle* h = (AzHandle*)e;
    if (!h || !priors || !values || !az_range_ok(h, g0, g1)) return AZSP_EINVAL;
    OpBackup op = {priors, values};
    int rc = az_run(h, op, stream, g0, g1);
    if (rc) return rc;
    return az_run(h, OpEndMove(), stream, g0, g1);
}

int azsp_round(void* e, const float* priors, const float* values, void* feat, uint8_t* valid, void* stream) {
    int rc = azsp_expand_backup(e, priors, values, stream);
    if (rc) return rc;
    return azsp_select(e, feat, valid, stream);
}

int azsp_get_status(void* e, int32_t* status, double* q, void* stream) {
    AzHandle* h = (AzHandle*)e;
    if (!h) return AZSP_EINVAL;
    OpStatus op = {h->d_status, h->d_q};
    int rc = az_run(h, op, stream);
    if (rc) return rc;
    if (status && azb::d2h(status, h->d_status, sizeof(int) * 8 * (size_t)h->cfg.G, stream)) return AZSP_EDEVICE;
    if (q && azb::d2h(q, h->d_q, sizeof(double) * 2 * (size_t)h->cfg.G, stream)) return AZSP_EDEVICE;
    return az_check_engine_fault(h, stream);
}

int azsp_dropin_step(void* e, const float* priors_host, const float* values_host, float* priors_dev, float* values_dev, void* feat_dev,
                     uint8_t* valid_dev, int32_t* status_host, double* q_host, uint8_t* valid_host, void* feat_host, int64_t feat_bytes, void* stream) {
    AzHandle* h = (AzHandle*)e;
    if (!h || !priors_dev || !values_dev || !feat_dev || !valid_dev || !status_host || !valid_host || feat_bytes < 0 || (feat_bytes > 0 && !feat_host) ||
        (priors_host == nullptr) != (values_host == nullptr))
        return AZSP_EINVAL;
    const size_t G = (size_t)h->cfg.G, rows = G * (size_t)h->cfg.P, A = (size_t)h->A;
    static const int elem_of[7] = {1, 4, 2, 2, 0, 0, 0};  // AZSP_FEAT_I8 / F32 / BF16 / F16: plain [rows][17][N][N] tensors only
    const int fd = h->pub.feature_dtype;
    if (fd < 0 || fd > 6 || elem_of[fd] == 0) return AZSP_EINVAL;
    const size_t game_feat = (size_t)h->cfg.P * 17 * h->NP * elem_of[fd];
    if ((size_t)feat_bytes > G * game_feat) return AZSP_EINVAL;
    // page-locked staging, device-visible: [priors rows*A f32][values rows f32] | [status G*8 i32][q G*2 f64][fault G i32][valid rows u8][features]
    const size_t o_val = rows * A * 4, up = o_val + rows * 4, o_st = (up + 15) & ~(size_t)15, o_q = o_st + G * 32, o_err = o_q + G * 16,
                 o_valid = (o_err + G * 4 + 15) & ~(size_t)15, o_feat = o_valid + ((rows + 15) & ~(size_t)15), total = o_feat + (size_t)feat_bytes;
    if (h->pin_bytes < total) {
        if (h->pin) {
            if (azb::sync(stream)) return AZSP_EDEVICE;  // no launch may still be writing the old staging
            azb::host_release(h->pin);
        }
        h->pin = (unsigned char*)azb::host_alloc(total);
        h->pin_bytes = h->pin ? total : 0;
        if (!h->pin) return AZSP_ENOMEM;
    }
    unsigned char* pin = h->pin;
    if (priors_host) {  // eval_func's outputs for the leaves of the previous call (core/mcts_v2.py:614-625 consume them)
        memcpy(pin, priors_host, rows * A * 4);
        memcpy(pin + o_val, values_host, rows * 4);
    }
    // (priors_dev / values_dev stay the caller's tensors for the separate entries; this entry reads the staging directly)
    OpDropinStep op = {priors_host ? (const float*)pin : nullptr, priors_host ? (const float*)(pin + o_val) : nullptr, feat_dev, valid_dev,
                       (int*)(pin + o_st), (double*)(pin + o_q), (int*)(pin + o_err), pin + o_valid, pin + o_feat, (long long)feat_bytes, (long long)game_feat};
    int rc = az_run(h, op, stream);
    if (rc) return rc;
    if (azb::sync(stream)) {
        h->err = std::string("synchronisation failed: ") + azb::backend_error();
        return AZSP_EDEVICE;
    }
    memcpy(status_host, pin + o_st, G * 32);
    if (q_host) memcpy(q_host, pin + o_q, G * 16);
    memcpy(valid_host, pin + o_valid, rows);
    if (feat_bytes > 0) memcpy(feat_host, pin + o_feat, (size_t)feat_bytes);
    int ef = 0;
    for (size_t g = 0; g < G; ++g) ef |= ((const int*)(pin + o_err))[g];
    if (ef) {
        char buf[160];
        snprintf(buf, sizeof buf, "engine fault flags 0x%x (1=node pool exhausted, 2=tree deeper than %d, 4=move sampling, 8=staging full)", ef, AZ_PATH_CAP);
        h->err = buf;
        return AZSP_EENGINE;
    }
    return AZSP_OK;
}

int azsp_rng_probe(void* e, int32_t plies, int32_t tries, double* noise_host, double* unif_host, void* stream) {
    AzHandle* h = (AzHandle*)e;
    if (!h || plies < 1 || plies > 1024 || tries < 0 || tries > 64 || (!noise_host && !unif_host)) return AZSP_EINVAL;
    const size_t nn = (size_t)h->cfg.G * plies * h->A, nu = (size_t)h->cfg.G * plies * (tries > 0 ? tries : 1);
    double* dn = noise_host ? (double*)azb::alloc(nn * sizeof(double)) : nullptr;
    double* du = (unif_host && tries > 0) ? (double*)azb::alloc(nu * sizeof(double)) : nullptr;
    int rc = AZSP_OK;
    if ((noise_host && !dn) || (unif_host && tries > 0 && !du)) rc = AZSP_ENOMEM;
    if (rc == AZSP_OK) {
        OpRngProbe op = {dn, du, plies, tries};
        rc = az_run(h, op, stream);
    }
    if (rc == AZSP_OK && dn && azb::d2h(noise_host, dn, nn * sizeof(double), stream)) rc = AZSP_EDEVICE;
    if (rc == AZSP_OK && du && azb::d2h(unif_host, du, nu * sizeof(double), stream)) rc = AZSP_EDEVICE;
    if (rc == AZSP_OK && azb::sync(stream)) rc = AZSP_EDEVICE;
    if (dn) azb::release(dn);
    if (du) azb::release(du);
    return rc;
}

int azsp_get_search(void* e, int32_t slot, int32_t ply, double* pi, float* cn, double* q, void* stream) {
    AzHandle* h = (AzHandle*)e;
    if (!h || slot < 0 || slot >= h->cfg.G || ply < 0 || ply >= h->cfg.log_cap) return AZSP_EINVAL;
    const size_t o = (size_t)slot * h->cfg.log_cap + ply;
    if (pi && azb::d2h(pi, h->mem.log_pi + o * h->A, sizeof(double) * h->A, stream)) return AZSP_EDEVICE;
    if (cn && azb::d2h(cn, h->mem.log_childN + o * h->A, sizeof(float) * h->A, stream)) return AZSP_EDEVICE;
    if (q && azb::d2h(q, h->mem.log_q + o * 4, sizeof(double) * 4, stream)) return AZSP_EDEVICE;
    return AZSP_OK;
}

int azsp_commit_move(void* e, const int32_t* moves, void* stream) {
    AzHandle* h = (AzHandle*)e;
    if (!h || !moves) return AZSP_EINVAL;
    if (azb::h2d(h->d_moves, moves, sizeof(int) * (size_t)h->cfg.G, stream)) return AZSP_EDEVICE;
    OpCommit op = {h->d_moves};
    return az_run(h, op, stream);
}

int azsp_harvest(void* e, int8_t* states, float* pi, float* z, int32_t cap, int32_t* games, int32_t max_games, int32_t* n_samples,
                 int32_t* n_games, void* stream) {
    AzHandle* h = (AzHandle*)e;
    if (!h || !states || !pi || !z || !games || !n_samples || !n_games || cap < 1) return AZSP_EINVAL;
    if (max_games > h->d_games_cap) max_games = h->d_games_cap;
    if (azb::zero(h->d_hcounts, sizeof(int) * 4, stream)) return AZSP_EDEVICE;
    OpHarvestCount cnt_op = {h->d_hlen};
    int rc = az_run(h, cnt_op, stream);
    if (rc) return rc;
    const int n2 = 2 * h->cfg.G;
    const int rot = (int)(((unsigned long long)h->harvest_seq++ * 2ull * 9973ull) % (unsigned long long)n2);  // a different first slot every harvest
    if (azb::launch_harvest_scan(h->d_hlen, h->d_hofs, h->d_hcounts, n2, rot, cap, max_games, stream)) {
        h->err = std::string("kernel launch failed: ") + azb::backend_error();
        return AZSP_EDEVICE;
    }
    OpHarvest op = {states, pi, z, cap, h->d_games, max_games, h->d_hofs, h->harvest_moves, h->d_gextra};
    rc = az_run(h, op, stream);
    if (rc) return rc;
    int cnt[4];
    if (azb::d2h(cnt, h->d_hcounts, sizeof cnt, stream)) return AZSP_EDEVICE;
    *n_samples = cnt[0];
    *n_games = cnt[1];
    if (cnt[1] > 0 && azb::d2h(games, h->d_games, sizeof(int) * 16 * (size_t)cnt[1], stream)) return AZSP_EDEVICE;
    if (cnt[1] > 0 && h->harvest_extra && azb::d2h(h->harvest_extra, h->d_gextra, sizeof(int) * 4 * (size_t)cnt[1], stream)) return AZSP_EDEVICE;
    return az_check_engine_fault(h, stream);
}

int azsp_harvest_extra(void* e, int32_t* extra_host) {
    AzHandle* h = (AzHandle*)e;
    if (!h) return AZSP_EINVAL;
    h->harvest_extra = extra_host;
    return AZSP_OK;
}

int azsp_set_actor_state(void* e, double resign_threshold, int32_t training_steps) {
    AzHandle* h = (AzHandle*)e;
    if (!h) return AZSP_EINVAL;
    h->cfg.resign_threshold = resign_threshold;  // AzCfg travels by value with every launch: the next new_game() reads it
    h->cfg.training_steps = training_steps;
    return AZSP_OK;
}

int azsp_harvest_moves(void* e, int16_t* moves_dev) {
    AzHandle* h = (AzHandle*)e;
    if (!h) return AZSP_EINVAL;
    h->harvest_moves = moves_dev;
    return AZSP_OK;
}

int azsp_counters(void* e, uint64_t* out, int32_t reset, void* stream) {
    AzHandle* h = (AzHandle*)e;
    if (!h || !out) return AZSP_EINVAL;
    const size_t G = (size_t)h->cfg.G;
    std::vector<u64> rows(G * AZC_COUNT);
    if (azb::d2h(rows.data(), h->mem.counters, sizeof(u64) * G * AZC_COUNT, stream)) return AZSP_EDEVICE;
    for (int i = 0; i < AZC_COUNT; ++i) out[i] = 0;
    for (size_t g = 0; g < G; ++g)  // telemetry only: per-game rows are summed here instead of contended device atomics
        for (int i = 0; i < AZC_COUNT; ++i) out[i] += rows[g * AZC_COUNT + i];
    if (reset && azb::zero(h->mem.counters, sizeof(u64) * G * AZC_COUNT, stream)) return AZSP_EDEVICE;
    return AZSP_OK;
}

int azsp_dihedral(const void* sin, void* sout, int32_t ses, const void* pin, void* pout, int32_t pes, int32_t batch, int32_t ch,
                  int32_t n, int32_t A, int32_t op, void* stream) {
    if (batch < 0 || n < 1 || op < 0 || op > 7) return AZSP_EINVAL;
    if (pin && A != n * n && A != n * n + 1) return AZSP_EINVAL;  // ValueError('Expect ...') transformation.py:36-39
    if ((sin && (ses < 1 || ses > 8 || !sout)) || (pin && (pes < 1 || pes > 8 || !pout))) return AZSP_EINVAL;
    DihedralArgs a = {(const unsigned char*)sin, (unsigned char*)sout, (const unsigned char*)pin, (unsigned char*)pout, ses, pes, batch, ch, n, A, op};
    const long long total = (long long)batch * ch * n * n + (long long)batch * A;
    if (total == 0) return AZSP_OK;
    return azb::launch_dihedral(a, total, stream) == 0 ? AZSP_OK : AZSP_EDEVICE;
}

int azsp_bias_act(void* y, const void* bias, const void* res, int64_t rows, int32_t channels, int32_t dtype, int32_t relu, void* stream) {
    if (!y || !bias || rows < 0 || channels < 8 || channels % 8 != 0 || dtype < AZSP_FEAT_F32 || dtype > AZSP_FEAT_F16) return AZSP_EINVAL;
    if (rows == 0) return AZSP_OK;
    BiasActArgs a = {y, bias, res, (long long)rows * channels / 8, channels, dtype, relu};
    return azb::launch_bias_act(a, stream) == 0 ? AZSP_OK : AZSP_EDEVICE;
}

int64_t azsp_tiled_bytes(int64_t boards, int32_t S, int32_t C) {
    if (boards < 0 || S <= 0 || C <= 0 || C % 8) return -1;
    const int tb = cv_tile_boards(S);
    return (boards + tb - 1) / tb * (int64_t)tb * S * S * C * 2;
}

int azsp_tile_layout(const void* src, void* dst, int64_t boards, int32_t S, int32_t C, int32_t to_tiled, void* stream) {
    if (!src || !dst || boards < 0 || boards > 0x7fffffff || S <= 0 || C <= 0 || C % 8) return AZSP_EINVAL;
    if (boards == 0) return AZSP_OK;
    const int rc = azb::launch_tile_layout(src, dst, (long long)boards, S, C, to_tiled, stream);
    return rc == 0 ? AZSP_OK : (rc > 0 ? AZSP_EINVAL : AZSP_EDEVICE);
}

static int az_conv3x3_tiled(const void* x, const void* w, const float* bias, const void* res, void* y, int64_t boards, int32_t S, int32_t C,
                            int32_t relu, void* stream, int f16) {
    if (!x || !w || !bias || !y || boards < 0 || boards > 0x7fffffff) return AZSP_EINVAL;
    if (boards == 0) return AZSP_OK;
    const int rc = azb::launch_conv3x3_tiled(x, w, bias, res, y, (long long)boards, S, C, relu, stream, f16);
    return rc == 0 ? AZSP_OK : (rc > 0 ? AZSP_EINVAL : AZSP_EDEVICE);
}
int azsp_conv3x3_tiled(const void* x, const void* w, const float* bias, const void* res, void* y, int64_t boards, int32_t S, int32_t C,
                       int32_t relu, void* stream) {
    return az_conv3x3_tiled(x, w, bias, res, y, boards, S, C, relu, stream, 0);
}
int azsp_conv3x3_tiled_f16(const void* x, const void* w, const float* bias, const void* res, void* y, int64_t boards, int32_t S, int32_t C,
                           int32_t relu, void* stream) {
    return az_conv3x3_tiled(x, w, bias, res, y, boards, S, C, relu, stream, 1);
}

int64_t azsp_split_bytes(int64_t boards, int32_t S, int32_t C) {
    if (boards < 0 || S <= 0 || C <= 0 || C % 8) return -1;
    return boards * 2 * (int64_t)S * S * C * 2;
}

int azsp_split_layout(const void* src, void* dst, int64_t boards, int32_t S, int32_t C, int32_t to_split, uint32_t* range_rec, void* stream) {
    if (!src || !dst || boards < 0 || boards > 0x7fffffff || S <= 0 || C <= 0 || C % 8) return AZSP_EINVAL;
    if (boards == 0) return AZSP_OK;
    const int rc = azb::launch_split_layout(src, dst, (long long)boards, S, C, to_split, stream, range_rec);
    return rc == 0 ? AZSP_OK : (rc > 0 ? AZSP_EINVAL : AZSP_EDEVICE);
}

int azsp_conv3x3_split(const void* x, const void* w, const float* bias, const void* res, void* y, int64_t boards, int32_t S, int32_t C,
                       int32_t relu, uint32_t* range_rec, void* stream) {
    if (!x || !w || !bias || !y || x == y || boards < 0 || boards > 0x7fffffff) return AZSP_EINVAL;
    // 17x17: 15 of a board's 304 column slots repeat a position of an EARLIER column tile, whose store has happened by the time the
    // repeat loads its residual -- in place (residual == y) the repeat would add the skip twice.  Refused, not silently wrong.
    if (S == 17 && res == y) return AZSP_EINVAL;
    if (boards == 0) return AZSP_OK;
    const int rc = azb::launch_conv3x3_split(x, w, bias, res, y, (long long)boards, S, C, relu, stream, range_rec);
    return rc == 0 ? AZSP_OK : (rc > 0 ? AZSP_EINVAL : AZSP_EDEVICE);
}

int azsp_resblock_split(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, int64_t boards, int32_t S, int32_t C,
                        uint32_t* range_rec, void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !y || x == y || boards < 0 || boards > 0x7fffffff) return AZSP_EINVAL;
    if (boards == 0) return AZSP_OK;
    const int rc = azb::launch_resblock_split(x, w1, b1, w2, b2, y, (long long)boards, S, C, stream, range_rec);
    return rc == 0 ? AZSP_OK : (rc > 0 ? AZSP_EINVAL : AZSP_EDEVICE);
}

int azsp_split_features(const float* planes, void* dst, int64_t boards, int32_t S, int32_t cin, uint32_t* range_rec, void* stream) {
    if (!planes || !dst || boards < 0 || boards > 0x7fffffff || S <= 0 || cin < 1 || cin > 32) return AZSP_EINVAL;
    if (boards == 0) return AZSP_OK;
    const int rc = azb::launch_split_features(planes, dst, (long long)boards, S, cin, stream, range_rec);
    return rc == 0 ? AZSP_OK : (rc > 0 ? AZSP_EINVAL : AZSP_EDEVICE);
}

int azsp_stem_split(const void* x, const void* w, const float* bias, void* y, int64_t boards, int32_t S, int32_t C, int32_t pad, int32_t relu,
                    uint32_t* range_rec, void* stream) {
    if (!x || !w || !bias || !y || x == y || boards < 0 || boards > 0x7fffffff) return AZSP_EINVAL;
    if (boards == 0) return AZSP_OK;
    const int rc = azb::launch_stem_split(x, w, bias, y, (long long)boards, S, C, pad, relu, stream, 0, range_rec);
    return rc == 0 ? AZSP_OK : (rc > 0 ? AZSP_EINVAL : AZSP_EDEVICE);
}

int azsp_stem_split_exact(const void* x, const void* w, const float* bias, void* y, int64_t boards, int32_t S, int32_t C, int32_t pad, int32_t relu,
                          uint32_t* range_rec, void* stream) {
    if (!x || !w || !bias || !y || x == y || boards < 0 || boards > 0x7fffffff) return AZSP_EINVAL;
    if (boards == 0) return AZSP_OK;
    const int rc = azb::launch_stem_split(x, w, bias, y, (long long)boards, S, C, pad, relu, stream, 1, range_rec);
    return rc == 0 ? AZSP_OK : (rc > 0 ? AZSP_EINVAL : AZSP_EDEVICE);
}

int azsp_split_range_read(const uint32_t* range_rec, uint32_t* events, float* max_abs, int32_t reset, void* stream) {
    unsigned r[2] = {0u, 0u};
    if (azb::split_range_read(range_rec, r, reset, stream) != 0) return AZSP_EDEVICE;
    if (events) *events = r[0];
    if (max_abs) memcpy(max_abs, &r[1], sizeof(float));
    return AZSP_OK;
}

int azsp_split_range_status(uint32_t* events, float* max_abs, int32_t reset, void* stream) {
    return azsp_split_range_read(nullptr, events, max_abs, reset, stream);
}

int64_t azsp_small_batch_waves(int64_t waves) { return (int64_t)azb::small_batch_waves((long long)waves); }

int azsp_head_split(const void* x, const float* head_w, const float* head_b, const float* pol_fc_wt, const float* pol_fc_b, const float* val_fc1_wt,
                    const float* val_fc1_b, const float* val_fc2_w, float val_fc2_b, float* priors, float* values, int64_t boards, int32_t S,
                    int32_t C, int32_t A, int32_t F, int32_t npol, void* stream) {
    if (!x || !head_w || !head_b || !pol_fc_wt || !pol_fc_b || !val_fc1_wt || !val_fc1_b || !val_fc2_w || !priors || !values || boards < 0 ||
        boards > 0x7fffffff || S <= 0 || C <= 0 || A <= 0 || F <= 0)
        return AZSP_EINVAL;
    if (boards == 0) return AZSP_OK;
    const azb::HeadSplitArgs a = {x, head_w, head_b, pol_fc_wt, pol_fc_b, val_fc1_wt, val_fc1_b, val_fc2_w, val_fc2_b, priors, values, (long long)boards,
                             S, C, A, F, npol};
    const int rc = azb::launch_head_split(a, stream);
    return rc == 0 ? AZSP_OK : (rc > 0 ? AZSP_EINVAL : AZSP_EDEVICE);
}

int azsp_resblock_tiled(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, int64_t boards, int32_t S, int32_t C,
                        void* stream) {
    if (!x || !w1 || !b1 || !w2 || !b2 || !y || boards < 0 || boards > 0x7fffffff) return AZSP_EINVAL;
    if (boards == 0) return AZSP_OK;
    const int rc = azb::launch_resblock_tiled(x, w1, b1, w2, b2, y, (long long)boards, S, C, stream);
    return rc == 0 ? AZSP_OK : (rc > 0 ? AZSP_EINVAL : AZSP_EDEVICE);
}

int azsp_replay_gather(const int8_t* ring_states, const float* ring_pi, const float* ring_z, const int64_t* idx, int32_t batch, int32_t channels,
                       int32_t n, int32_t A, int32_t op, int32_t state_dtype, void* out_states, float* out_pi, float* out_z, void* stream) {
    if (!ring_states || !ring_pi || !ring_z || !idx || !out_states || !out_pi || !out_z || batch < 0 || channels < 1 || n < 1 ||
        (A != n * n && A != n * n + 1) || op < 0 || op > 7 || state_dtype < AZSP_FEAT_I8 || state_dtype > AZSP_FEAT_F16)
        return AZSP_EINVAL;
    if (batch == 0) return AZSP_OK;
    ReplayGatherArgs a = {ring_states, ring_pi, ring_z, (const long long*)idx, out_states, out_pi, out_z, batch, channels, n, A, op, state_dtype};
    const long long total = (long long)batch * ((long long)channels * n * n + A + 1);
    return azb::launch_replay_gather(a, total, stream) == 0 ? AZSP_OK : AZSP_EDEVICE;
}

static int az_stem_tiled(const void* x, const void* w, const float* bias, void* y, int64_t boards, int32_t S, int32_t C, int32_t pad, int32_t relu,
                         void* stream, int f16) {
    if (!x || !w || !bias || !y || boards < 0 || boards > 0x7fffffff || pad < 1) return AZSP_EINVAL;
    if (boards == 0) return AZSP_OK;
    const int rc = azb::launch_stem_tiled(x, w, bias, y, (long long)boards, S, C, pad, relu, stream, f16);
    return rc == 0 ? AZSP_OK : (rc > 0 ? AZSP_EINVAL : AZSP_EDEVICE);
}
int azsp_stem_tiled(const void* x, const void* w, const float* bias, void* y, int64_t boards, int32_t S, int32_t C, int32_t pad, int32_t relu,
                    void* stream) {
    return az_stem_tiled(x, w, bias, y, boards, S, C, pad, relu, stream, 0);
}
int azsp_stem_tiled_f16(const void* x, const void* w, const float* bias, void* y, int64_t boards, int32_t S, int32_t C, int32_t pad, int32_t relu,
                        void* stream) {
    return az_stem_tiled(x, w, bias, y, boards, S, C, pad, relu, stream, 1);
}

static int az_head_tiled(const void* x, const float* w, const float* bias, void* pol, void* val, int64_t boards, int32_t S, int32_t C, int32_t npol,
                         int32_t nval, int32_t pol_stride, int32_t val_stride, void* stream, int f16) {
    if (!x || !w || !bias || !pol || !val || boards < 0 || boards > 0x7fffffff || npol < 1 || nval < 1) return AZSP_EINVAL;
    if (pol_stride == 0) pol_stride = npol * S * S;
    if (val_stride == 0) val_stride = nval * S * S;
    if (pol_stride < npol * S * S || val_stride < nval * S * S) return AZSP_EINVAL;
    if (boards == 0) return AZSP_OK;
    const int rc = azb::launch_head_tiled(x, w, bias, pol, val, (long long)boards, S, C, npol, nval, pol_stride, val_stride, stream, f16);
    return rc == 0 ? AZSP_OK : (rc > 0 ? AZSP_EINVAL : AZSP_EDEVICE);
}
int azsp_head_tiled(const void* x, const float* w, const float* bias, void* pol, void* val, int64_t boards, int32_t S, int32_t C, int32_t npol,
                    int32_t nval, int32_t pol_stride, int32_t val_stride, void* stream) {
    return az_head_tiled(x, w, bias, pol, val, boards, S, C, npol, nval, pol_stride, val_stride, stream, 0);
}
int azsp_head_tiled_f16(const void* x, const float* w, const float* bias, void* pol, void* val, int64_t boards, int32_t S, int32_t C, int32_t npol,
                        int32_t nval, int32_t pol_stride, int32_t val_stride, void* stream) {
    return az_head_tiled(x, w, bias, pol, val, boards, S, C, npol, nval, pol_stride, val_stride, stream, 1);
}

static int az_fc_heads(const void* pol, const void* val, const void* wp, const float* bp, int32_t ks1, const void* w1, const float* b1, int32_t ks2,
                       const float* w2, float b2, float* priors, float* values, int64_t boards, int32_t A, int32_t F, void* stream, int f16) {
    if (!pol || !val || !wp || !bp || !w1 || !b1 || !w2 || !priors || !values || boards < 0 || ks1 < 1 || ks2 < 1 || A < 1 || F < 1) return AZSP_EINVAL;
    if (boards == 0) return AZSP_OK;
    azb::FcHeadsArgs a = {pol, val, wp, w1, bp, b1, w2, b2, priors, values, (long long)boards, ks1, ks2, A, F, f16};
    const int rc = azb::launch_fc_heads(a, stream);
    return rc == 0 ? AZSP_OK : (rc > 0 ? AZSP_EINVAL : AZSP_EDEVICE);
}
int azsp_fc_heads(const void* pol, const void* val, const void* wp, const float* bp, int32_t ks1, const void* w1, const float* b1, int32_t ks2,
                  const float* w2, float b2, float* priors, float* values, int64_t boards, int32_t A, int32_t F, void* stream) {
    return az_fc_heads(pol, val, wp, bp, ks1, w1, b1, ks2, w2, b2, priors, values, boards, A, F, stream, 0);
}
int azsp_fc_heads_f16(const void* pol, const void* val, const void* wp, const float* bp, int32_t ks1, const void* w1, const float* b1, int32_t ks2,
                      const float* w2, float b2, float* priors, float* values, int64_t boards, int32_t A, int32_t F, void* stream) {
    return az_fc_heads(pol, val, wp, bp, ks1, w1, b1, ks2, w2, b2, priors, values, boards, A, F, stream, 1);
}

}  // extern "C"
