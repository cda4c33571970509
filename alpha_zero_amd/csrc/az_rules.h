// az_rules.h -- Go (liberty / capture / suicide / ko / Tromp-Taylor area) and free-style Gomoku as
// bitboard transforms executed by one 64-lane wave per position.
//
// Reference behaviour restated here (bit-exact, see tests/ and oracle/rules.c):
//   Go     alpha_zero/envs/go_engine.py:386-507 (suicide, legal mask, pass, play_move, ko rule),
//          :123-152 (area_score), alpha_zero/envs/go.py:88-192 (step, rewards, termination)
//   Gomoku alpha_zero/envs/gomoku.py:45-136 (step, win scan through the last move, draw)
// The reference keeps groups and liberties incrementally in Python sets (LibertyTracker); on the
// GPU they are recomputed on demand: whole-board quantities ("which opponent stones still have a
// liberty") are single flood fills on wave-uniform bitboards, per-group liberty COUNTS (needed only
// for candidate-suicide points) are one flood per lane, seeded at "point = lane + 64*k", and the
// verdicts come back as ballot masks that are themselves bitboard words.
#pragma once
#include "az_bitboard.h"

enum { AZ_GO = 0, AZ_GOMOKU = 1 };
enum { AZF_TERMINAL = 1, AZF_LASTPASS = 2, AZF_RESIGNED = 4 };
enum { AZ_MOVE_NONE = -2, AZ_MOVE_RESIGN = -1 };

struct RuleCfg {
    int max_steps;   // go.py:48 (2*N*N by default)
    int num_to_win;  // gomoku.py:26
    double komi;     // go.py:46
};

// One position: 3*W words + 16 bytes = 64 B for 9x9, 88 B for 13x13, 160 B for 19x19.
template <int W> struct EnvState {
    u64 stones[2][W];  // [0] black, [1] white
    u64 legal[W];      // legal POINTS for the player to move (pass is legal iff Go and not terminal)
    int16_t ko;        // point or -1 (Position.ko)
    int16_t steps;
    uint8_t to_play;    // 0 black, 1 white
    uint8_t flags;      // AZF_*
    int8_t winner;      // -1 none, 0 black, 1 white
    int8_t reward;      // reward of the LAST MOVER when terminal (go.py:150-156, gomoku.py:74-76)
    uint16_t caps[2];   // Position.caps
    int16_t area[2];    // Tromp-Taylor areas when a Go game ended by score
};

template <class Wv, int N> struct Rules {
    typedef BBOps<N> O;
    typedef typename O::B B;
    static constexpr int NP = O::NP;
    static constexpr int W = O::W;
    typedef EnvState<W> S;

    static AZ_HD B ld(const u64* p) {  // wave-uniform load of a bitboard (SGPRs on the device)
        B r;
        for (int i = 0; i < W; ++i) r.w[i] = Wv::uni(p[i]);
        return r;
    }
    // a scalarised copy of a position: every field pinned to SGPRs
    static AZ_HD S uni_state(const S& s) {
        S o;
        for (int q = 0; q < 2; ++q)
            for (int i = 0; i < W; ++i) o.stones[q][i] = Wv::uni(s.stones[q][i]);
        for (int i = 0; i < W; ++i) o.legal[i] = Wv::uni(s.legal[i]);
        o.ko = (int16_t)Wv::uni((int)s.ko);
        o.steps = (int16_t)Wv::uni((int)s.steps);
        o.to_play = (uint8_t)Wv::uni((int)s.to_play);
        o.flags = (uint8_t)Wv::uni((int)s.flags);
        o.winner = (int8_t)Wv::uni((int)s.winner);
        o.reward = (int8_t)Wv::uni((int)s.reward);
        o.caps[0] = (uint16_t)Wv::uni((int)s.caps[0]);
        o.caps[1] = (uint16_t)Wv::uni((int)s.caps[1]);
        o.area[0] = (int16_t)Wv::uni((int)s.area[0]);
        o.area[1] = (int16_t)Wv::uni((int)s.area[1]);
        return o;
    }
    static AZ_HD void st(u64* p, const B& b) {
        for (int i = 0; i < W; ++i) p[i] = b.w[i];
    }

    static AZ_HD void reset(S& s, int game) {
        B z = O::zero();
        st(s.stones[0], z);
        st(s.stones[1], z);
        st(s.legal, O::inv(z));
        s.ko = -1;
        s.steps = 0;
        s.to_play = 0;
        s.flags = 0;
        s.winner = -1;
        s.reward = 0;
        s.caps[0] = s.caps[1] = 0;
        s.area[0] = s.area[1] = 0;
        (void)game;
    }

    // ---- Go ---------------------------------------------------------------------------------
    // go_engine.py:417-441 all_legal_moves for the player owning `own`.
    static AZ_HD B go_legal(const B& own, const B& opp, int ko) {
        const B occ = O::bor(own, opp);
        const B empty = O::inv(occ);
        // :423-429 "surrounded spots": empty points without an empty neighbour (board edge counts as stone)
        const B cand = O::andnot(empty, O::nbr(empty));
        B legal = empty;
        if (O::any(cand)) {
            // Only stones next to a candidate matter.  One flood per such stone, one stone per lane:
            // a friendly group with >= 2 liberties keeps the point alive (:394-402), an opponent group
            // in atari is captured by playing there (:396-398).
            const B need = O::band(O::nbr(cand), occ);
            B okstones;
            for (int k = 0; k < W; ++k) {
                okstones.w[k] = Wv::ballot([&](int lane) -> bool {
                    const int p = lane + 64 * k;
                    if (p >= NP || !O::test(need, p)) return false;
                    const bool mine = O::test(own, p);
                    const B grp = O::flood(O::bit(p), mine ? own : opp);
                    const int libs = O::count(O::band(O::nbr(grp), empty));
                    return mine ? (libs >= 2) : (libs == 1);
                });
            }
            legal = O::bor(O::andnot(empty, cand), O::band(cand, O::nbr(okstones)));
        }
        if (ko >= 0) legal = O::andnot(legal, O::bit(ko));  // :437-438
        return legal;
    }

    // go_engine.py:123-152 area_score (no dead-stone removal): an empty region counts for a colour
    // iff it touches that colour only <=> its points are reachable through empties from that colour
    // and not from the other.
    static AZ_HD void go_area(const B& black, const B& white, int& ab, int& aw) {
        const B empty = O::inv(O::bor(black, white));
        const B rb = O::flood(O::band(O::nbr(black), empty), empty);
        const B rw = O::flood(O::band(O::nbr(white), empty), empty);
        ab = O::count(black) + O::count(O::andnot(rb, rw));
        aw = O::count(white) + O::count(O::andnot(rw, rb));
    }

    static AZ_HD void go_finish(S& o, const RuleCfg& rc, int mover) {
        // go.py:139-156: legal mask cleared, winner from black - (white + komi), reward for the last mover
        o.flags |= AZF_TERMINAL;
        st(o.legal, O::zero());
        int ab, aw;
        go_area(ld(o.stones[0]), ld(o.stones[1]), ab, aw);
        o.area[0] = (int16_t)ab;
        o.area[1] = (int16_t)aw;
        const double score = (double)ab - ((double)aw + rc.komi);  // go_engine.py:509-516
        o.winner = score > 0 ? 0 : (score < 0 ? 1 : -1);
        o.reward = o.winner < 0 ? 0 : (o.winner == mover ? 1 : -1);
    }

    // GoEnv.step for a board point or pass (go.py:121-161).  `a` must be legal.
    static AZ_HD void go_step(const S& s_in, int a, const RuleCfg& rc, S& o) {
        const S s = uni_state(s_in);
        const int c = s.to_play;
        // colour-indexed fields are picked with selects (a runtime array index would spill the position to scratch)
        const B sb = ld(s.stones[0]), sw = ld(s.stones[1]);
        B own = c == 0 ? sb : sw, opp = c == 0 ? sw : sb;
        int ko = -1, ncap = 0;
        if (a != NP) {
            const B mb = O::bit(a);
            const B nb = O::nbr(mb);
            // is_koish on the board BEFORE the stone is placed (go_engine.py:479, :91-99)
            const bool koish = !O::any(O::andnot(nb, opp));
            own = O::bor(own, mb);
            if (O::any(O::band(nb, opp))) {
                // LibertyTracker.add_stone :239-245: opponent groups left without liberty are captured
                const B empty2 = O::inv(O::bor(own, opp));
                const B alive = O::flood(O::band(opp, O::nbr(empty2)), opp);
                const B captured = O::andnot(opp, alive);
                ncap = O::count(captured);
                opp = O::andnot(opp, captured);
                if (ncap == 1 && koish) ko = O::first(captured);  // :491-494
            }
        }
        st(o.stones[0], c == 0 ? own : opp);
        st(o.stones[1], c == 0 ? opp : own);
        o.caps[0] = (uint16_t)(s.caps[0] + (c == 0 ? ncap : 0));  // :496-499
        o.caps[1] = (uint16_t)(s.caps[1] + (c == 0 ? 0 : ncap));
        o.ko = (int16_t)ko;  // pass clears ko (:448)
        o.steps = (int16_t)(s.steps + 1);
        o.to_play = (uint8_t)(1 - c);
        o.winner = -1;
        o.reward = 0;
        o.area[0] = o.area[1] = 0;
        o.flags = (a == NP) ? AZF_LASTPASS : 0;
        // go.py:176-192: max_steps, or the last two recorded moves are passes
        const bool over = (o.steps >= rc.max_steps) || (a == NP && (s.flags & AZF_LASTPASS));
        if (over) {
            go_finish(o, rc, c);
        } else {
            st(o.legal, go_legal(opp, own, ko));  // next player owns `opp`
        }
    }

    // Resignation (go.py:103-119): turn flips, ko cleared, board unchanged, reward -1 for the resigner.
    static AZ_HD void go_resign(const S& s, S& o) {
        o = s;
        const int c = s.to_play;
        o.ko = -1;
        o.steps = (int16_t)(s.steps + 1);
        o.to_play = (uint8_t)(1 - c);
        o.flags = AZF_TERMINAL | AZF_RESIGNED;
        st(o.legal, O::zero());
        o.winner = (int8_t)(1 - c);
        o.reward = -1;
    }

    // ---- Gomoku -----------------------------------------------------------------------------
    static AZ_HD int run_len(const B& b, int r, int c, int dr, int dc) {
        // gomoku.py:236-303 count_same_color_stones (start included)
        int n = 1;
        for (;;) {
            r += dr;
            c += dc;
            if (r < 0 || c < 0 || r >= N || c >= N || !O::test(b, r * N + c)) return n;
            ++n;
        }
    }
    static AZ_HD void gomoku_step(const S& s_in, int a, const RuleCfg& rc, S& o) {
        const S s = uni_state(s_in);
        const int c = s.to_play;
        const B sb = ld(s.stones[0]), sw = ld(s.stones[1]);
        B own = O::bor(c == 0 ? sb : sw, O::bit(a));
        const B opp = c == 0 ? sw : sb;
        st(o.stones[0], c == 0 ? own : opp);
        st(o.stones[1], c == 0 ? opp : own);
        st(o.legal, O::inv(O::bor(own, opp)));  // gomoku.py:63: only the played point changes; never cleared at the end
        o.ko = -1;
        o.steps = (int16_t)(s.steps + 1);
        o.caps[0] = o.caps[1] = 0;
        o.area[0] = o.area[1] = 0;
        o.winner = -1;
        o.reward = 0;
        o.flags = 0;
        bool won = false;
        if (o.steps >= (rc.num_to_win - 1) * 2) {  // gomoku.py:88-90
            const int r = a / N, q = a % N;
            // gomoku.py:98-127: left/right, up/down, two diagonals through the last move
            won = run_len(own, r, q, 0, -1) + run_len(own, r, q, 0, 1) - 1 >= rc.num_to_win ||
                  run_len(own, r, q, -1, 0) + run_len(own, r, q, 1, 0) - 1 >= rc.num_to_win ||
                  run_len(own, r, q, -1, -1) + run_len(own, r, q, 1, 1) - 1 >= rc.num_to_win ||
                  run_len(own, r, q, -1, 1) + run_len(own, r, q, 1, -1) - 1 >= rc.num_to_win;
        }
        if (won) {
            o.winner = (int8_t)c;
            o.reward = 1;
        }
        if (won || O::count(O::bor(own, opp)) == NP) o.flags |= AZF_TERMINAL;  // gomoku.py:131-136
        o.to_play = (uint8_t)(1 - c);
    }

    template <int GAME> static AZ_HD void step(const S& s, int a, const RuleCfg& rc, S& o) {
        if (GAME == AZ_GO) go_step(s, a, rc, o);
        else gomoku_step(s, a, rc, o);
    }
};
