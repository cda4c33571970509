// az_engine.h -- batched PUCT self-play: one 64-lane wave owns one game (tree + real position).
//
// Restates, for thousands of concurrent games resident in HBM, the reference search and actor:
//   alpha_zero/core/mcts_v2.py   best_child :142-185, expand :188-210, backup :213-232,
//                                add_dirichlet_noise :235-262, generate_search_policy :265-298,
//                                uct_search :301-450, virtual loss :453-482, parallel_uct_search :485-657
//   alpha_zero/core/pipeline.py  play_and_record_one_game :289-382 (samples, resignation, z back-fill)
//   alpha_zero/envs/base.py      observation :228-259
// Data layout (per game, all in HBM):
//   node record  = [Hdr | N f32[AP] | W f32[AP] | P f32[AP] | child i16[AP] | hint i16[AP]]   (REC bytes, 128-B aligned rows)
//                  hint[a] = the node the search last continued to BELOW child a (a prefetch hint only, see descend())
//                  Hdr carries the position reached by the node, so a descent is an index walk and
//                  a child position is computed once, when the child is created (the reference
//                  re-steps a deep-copied env from the root on every simulation: mcts_v2.py:382-402)
//   GameRec      = real position, 8-deep board history, root statistics, per-round leaf table
// Numerics: every floating-point expression keeps the operand precision and evaluation order of the
// NumPy original (float32 rows, float64 noisy root priors, Python-double root statistics on a fresh
// root), compiled with -ffp-contract=off, so visit counts match the reference exactly.
#pragma once
#include "az_rules.h"
#include <math.h>

#define AZ_MAXP 32        // max leaves per game per round (num_parallel)
#ifndef AZ_PATH_CAP
#define AZ_PATH_CAP 64    // tree levels handled lane-parallel (one lane per edge); deeper paths walk the parent links
#endif
#define AZ_INJ_K 16       // injected sampling uniforms per move

enum { AZS_NEED_ROOT = 0, AZS_SEARCH = 1, AZS_MOVE_DONE = 2, AZS_IDLE = 3, AZS_WAIT_BUF = 4 };
enum { AZ_ERR_NODES = 1, AZ_ERR_SAMPLE = 4, AZ_ERR_STAGE = 8 };  // bit 2 (tree depth) retired: deep paths walk parent links
enum { AZ_FEAT_I8 = 0, AZ_FEAT_F32 = 1, AZ_FEAT_BF16 = 2, AZ_FEAT_F16 = 3, AZ_FEAT_BF16_TILED = 4, AZ_FEAT_F16_TILED = 5, AZ_FEAT_F16_SPLIT = 6 };
enum { AZB_FREE = 0, AZB_FILLING = 1, AZB_COMPLETE = 2 };
// statistics counters (u64 each)
enum { AZC_SIMS = 0, AZC_NODE_VISITS, AZC_BACKUP_EDGES, AZC_LEAVES, AZC_DUP_LEAVES, AZC_TERMINAL_HITS, AZC_MOVES,
       AZC_GAMES, AZC_ROOT_EVALS, AZC_NODES_CREATED, AZC_ROUNDS, AZC_STALLS, AZC_HINT_PREFETCH, AZC_HINT_HITS, AZC_COUNT = 16 };

AZ_HD u64 az_bits_f64(double v) {
    u64 b;
    __builtin_memcpy(&b, &v, sizeof b);
    return b;
}

struct AzCfg {
    int game, n, G, P, sims, budget, parallel_mode, max_nodes;
    int root_noise, deterministic, reuse_tree, warm_up_steps;
    int has_resign, check_resign_after, force_resign_disabled;  // force: -1 draw per game, 0/1 fixed
    int inject, inj_moves, stop_after_move, max_plies, stop_at_game_end;
    int feat_dtype, tab_len, log_moves, log_cap, stage_cap, training_steps;
    float one_minus_eps_f32, disable_resign_ratio;
    double eps, alpha, resign_threshold;
    RuleCfg rc;
    u64 seed;
    int rank;
};

struct AzMem {
    unsigned char* nodes;    // [G][max_nodes][REC]
    unsigned char* games;    // [G] GameRec
    double* rootP;           // [G][AP]   float64 root priors (after Dirichlet noise)
    int16_t* free_stack;     // [G][max_nodes]
    int* leaf_path;          // [G][P][AZ_PATH_CAP]  (node << 16) | move per edge, root first
    const double* pbc_np;    // [tab_len] pb_c for a node whose visit count is an np.float32
    const double* pbc_py;    // [tab_len] pb_c for a fresh root whose visit count is a Python float
    const float* sqrt32;     // [tab_len] float32(sqrt(n))
    const double* inj_noise; // [G][inj_moves][A]
    const double* inj_unif;  // [G][inj_moves][AZ_INJ_K]
    u64* stg_planes;         // [G][2][stage_cap][16][W]
    float* stg_pi;           // [G][2][stage_cap][A]
    unsigned char* stg_meta; // [G][2][stage_cap]   1 = black to move
    int16_t* stg_move;       // [G][2][stage_cap]   the move played from the sample's position (-1 = resigned)
    int* stg_hdr;            // [G][2][16]  see SH_*
    // move log (tests / shim): per game per logged move
    double* log_pi;          // [G][log_cap][A]
    float* log_childN;       // [G][log_cap][A]
    double* log_q;           // [G][log_cap][4]  root_q, child_q, root_N before search (unused), move
    u64* counters;           // [G][AZC_COUNT]  per-game rows (no atomics: 4096 waves hammering 12 shared words cost ~0.6 ms per launch)
    int* err;                // [1]
};

enum { SH_STATE = 0, SH_LEN, SH_WINNER, SH_AREA_B, SH_AREA_W, SH_PASSES, SH_RESIGNED, SH_RESIGN_DISABLED, SH_MARKED,
       SH_COULD_WON, SH_MARKED_PLAYER, SH_UID, SH_TRAINING_STEPS, SH_REWARD, SH_LAST_PLAYER,
       SH_THR_LO, SH_THR_HI, SH_TS_END, SH_COUNT = 20 };  // THR_*: the game's own resign threshold (double bits); TS_END: weights tag at its end

template <int W> struct GameRec {
    EnvState<W> env;     // the real position (== the root node's position)
    u64 hist[8][2][W];   // boards, newest first; hist[0] is the current board (base.py:261-266)
    double root_W;       // DummyNode slots of the root (mcts_v2.py:56-62)
    double out_root_q, out_child_q;
    int root_N;
    int root, n_free, status, root_fresh, root_noisy, ply, n_leaves, root_eval_pending;
    int uid, num_passes, marked_player, resign_disabled, cur_buf, out_move, noise_pending, noise_ready, games_done, warm_override;
    // per-game actor state, read when the game starts like the reference actor does before every game (pipeline.py:232-246):
    double resign_thr;   // var_resign_threshold.value at game start (<= -1: no resignation in this game)
    int train_steps;     // training_steps of the weights in use at game start (pipeline.py:237, :271)
    int pad0;
    int16_t leaf_node[AZ_MAXP];
    uint8_t leaf_depth[AZ_MAXP];
};


// Philox4x32-10 counter-based generator (production randomness: one independent stream per
// (seed, rank, game slot, game uid, ply, purpose, lane); replaces the actor's global np.random state).
struct Philox {
    static AZ_HD void round(u32 (&c)[4], u32 k0, u32 k1) {
        const u64 p0 = (u64)0xD2511F53u * c[0], p1 = (u64)0xCD9E8D57u * c[2];
        const u32 n0 = (u32)(p1 >> 32) ^ c[1] ^ k0, n1 = (u32)p1, n2 = (u32)(p0 >> 32) ^ c[3] ^ k1, n3 = (u32)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    }
    static AZ_HD void gen(u64 key, u32 c0, u32 c1, u32 c2, u32 c3, u32 (&out)[4]) {
        u32 c[4] = {c0, c1, c2, c3};
        u32 k0 = (u32)key, k1 = (u32)(key >> 32);
        for (int i = 0; i < 10; ++i) {
            round(c, k0, k1);
            k0 += 0x9E3779B9u;
            k1 += 0xBB67AE85u;
        }
        for (int i = 0; i < 4; ++i) out[i] = c[i];
    }
    static AZ_HD double u01(u32 hi, u32 lo) {  // 53-bit uniform in [0,1)
        return (double)((((u64)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
    }
};

#define AZ_LDS_NODES 1024  // parent table of the re-root pass lives in LDS up to this pool size

#define AZ_FREE_PREFETCH 64
typedef u64 __attribute__((may_alias)) u64a;  // word copies of typed records (no strict-aliasing assumptions)

// One node record staged from HBM into LDS in one burst (header + N / W / P / child rows).
template <int AP, int HW> struct StagedRec {
    float rN[AP], rW[AP], rP[AP];
    int16_t rC[AP];
    int16_t rH[AP];  // prefetch hints (never read by the search arithmetic)
    u64a hdrw[HW];  // header (position, parent, move, expanded); may_alias words, read back as Hdr
};

template <int W, int AP, int HW> struct Scratch {
    u64 planes[16][W];          // observation planes being assembled / history shift buffer
    int path[AZ_PATH_CAP];      // (node << 16) | move per edge of the current descent (first AZ_PATH_CAP levels)
    int anc[8];                 // node id at level (depth & 7): the last 8 nodes of the descent, for the history planes
    union {  // phase-private scratch: the end-of-move work and the select phase never overlap within a wave's launch
        struct {
            double pi[AP];              // search policy of the move being finished
            double cdf[AP];             // its running sum (np.cumsum order)
            float tmpf[AP];             // float32 policy terms (Gomoku)
            int16_t parent[AZ_LDS_NODES];
        };
        StagedRec<AP, HW> nxt;  // select: speculatively staged record of the node the descent will probably reach one level further down
    };
    StagedRec<AP, HW> root;     // the root's record: staged ONCE per round, kept current in LDS while the P descents update it
    StagedRec<AP, HW> cur;      // the record of the node the current descent is looking at (levels >= 1)
    double rP64[AP];            // float64 root priors (noisy root only)
    float wpath[AZ_PATH_CAP];   // W and N of the chosen child at every level of the current descent, captured at selection
    float npath[AZ_PATH_CAP];   //   time: virtual loss / terminal backup then need no read-modify-write round trip
    u64 ancst[8][2][W];         // stones of the last 8 nodes of the descent (history planes without re-reading headers)
    u64 hist[8][2][W];          // the real game's board history, staged once per round
    u64 leafst[2][W];           // the leaf's stones while its observation planes are assembled
    int16_t freetop[AZ_FREE_PREFETCH];
    int free_base;              // freetop[i] == free_stack[free_base + i]
};

template <class Wv, int N, int GAME> struct Engine {
    typedef Wv Wave;
    typedef Rules<Wv, N> R;
    typedef BBOps<N> O;
    typedef typename O::B B;
    static constexpr int GAME_ID = GAME;
    static constexpr int NP = N * N;
    static constexpr int A = NP + (GAME == AZ_GO ? 1 : 0);
    static constexpr int W = O::W;
    static constexpr int AP = (A + 31) / 32 * 32;
    static constexpr int EPL = (AP + 63) / 64;
    typedef EnvState<W> S;
    typedef GameRec<W> GR;
    struct Hdr {
        S st;
        int16_t parent, move;
        uint8_t expanded, pad_[3];
    };
    static constexpr int HDR = ((int)sizeof(Hdr) + 127) / 128 * 128;
    static constexpr int HW = ((int)sizeof(Hdr) + 7) / 8;
    typedef Scratch<W, AP, HW> SC;
    static constexpr int REC = (HDR + 3 * AP * 4 + 2 * AP * 2 + 127) / 128 * 128;
    static constexpr int GREC = ((int)sizeof(GR) + 127) / 128 * 128;

    const AzCfg& c;
    const AzMem& m;
    const int g;
    SC& sc;
    GR& gr;
    u64 cnt[AZC_COUNT];
    // Hot wave-uniform scalars of the GameRec, cached in registers for the duration of a select / backup phase: every `gr.x` is a
    // global-memory access (a load + wait + readfirstlane, i.e. an L2 round trip in the middle of a dependent chain).  The first
    // version read root_fresh / root_noisy / the pb_c table inside EVERY arg-max and n_free / root_W in every leaf; doing the arg-max
    // twice showed it cost 36 % of the select kernel (tools/sel_abl.sh).  hot_load() / hot_store() bracket the phase.
    int hs_root, hs_rootN, hs_nfree, hs_fresh, hs_noisy;
    double hs_rootW;
    int hs_tabN;        // the visit count the cached root table entries below belong to (-1: none)
    double hs_tabPbc;
    float hs_tabSq;

    AZ_HD void hot_load() {
        hs_root = Wv::uni(gr.root);
        hs_rootN = Wv::uni(gr.root_N);
        hs_nfree = Wv::uni(gr.n_free);
        hs_fresh = Wv::uni(gr.root_fresh);
        hs_noisy = Wv::uni(gr.root_noisy);
        hs_rootW = Wv::uni(gr.root_W);
        hs_tabN = -1;
    }
    AZ_HD void hot_store() {
        if (Wv::first()) {
            gr.root = hs_root;
            gr.root_N = hs_rootN;
            gr.n_free = hs_nfree;
            gr.root_W = hs_rootW;
        }
        Wv::sync();
    }
    // pb_c(n) and sqrt(n) of a node with n visits (host tables, azsp_set_tables): a global load each -- issued by the caller as early
    // as n is known so that it overlaps the staging of the node's record
    AZ_HD void table_entry(int n, bool fresh, double& pbc, float& sq) const {
        const int ti = n < c.tab_len ? n : c.tab_len - 1;
        pbc = Wv::uni(fresh ? m.pbc_py[ti] : m.pbc_np[ti]);
        sq = Wv::uni(m.sqrt32[ti]);
    }

    AZ_HD Engine(const AzCfg& c_, const AzMem& m_, int g_, SC& sc_)
        : c(c_), m(m_), g(g_), sc(sc_), gr(*(GR*)(m_.games + (size_t)g_ * GREC)) {
        for (int i = 0; i < AZC_COUNT; ++i) cnt[i] = 0;
    }

    // ---- addressing -------------------------------------------------------------------------
    AZ_HD unsigned char* rec(int node) const { return m.nodes + ((size_t)g * c.max_nodes + node) * REC; }
    AZ_HD Hdr& hdr(int node) const { return *(Hdr*)rec(node); }
    AZ_HD float* rowN(int node) const { return (float*)(rec(node) + HDR); }
    AZ_HD float* rowW(int node) const { return (float*)(rec(node) + HDR + AP * 4); }
    AZ_HD float* rowP(int node) const { return (float*)(rec(node) + HDR + 2 * AP * 4); }
    AZ_HD int16_t* rowC(int node) const { return (int16_t*)(rec(node) + HDR + 3 * AP * 4); }
    AZ_HD int16_t* rowH(int node) const { return (int16_t*)(rec(node) + HDR + 3 * AP * 4 + AP * 2); }
    AZ_HD double* rootP() const { return m.rootP + (size_t)g * AP; }
    AZ_HD int* leaf_path(int slot) const { return m.leaf_path + ((size_t)g * c.P + slot) * AZ_PATH_CAP; }
    AZ_HD void fail(int code) const {
        if (Wv::first()) *m.err |= code;
    }

    AZ_HD bool action_legal(const S& s, int a) const {
        if (a < NP) {
            const int wi = a >> 6;
            u64 word = 0;
            for (int i = 0; i < W; ++i) word = (i == wi) ? s.legal[i] : word;
            return (word >> (a & 63)) & 1ull;
        }
        return GAME == AZ_GO && !(s.flags & AZF_TERMINAL);  // go_engine.py:441 pass is always legal
    }

    // ---- node pool --------------------------------------------------------------------------
    AZ_HD int alloc_node() {  // (select phase: between hot_load() and hot_store())
        if (hs_nfree <= 0) {
            fail(AZ_ERR_NODES);
            return 0;
        }
        const int pos = hs_nfree - 1, rel = pos - Wv::uni(sc.free_base);
        const int idx = Wv::uni((rel >= 0 && rel < AZ_FREE_PREFETCH) ? (int)sc.freetop[rel] : (int)m.free_stack[(size_t)g * c.max_nodes + pos]);
        hs_nfree -= 1;
        cnt[AZC_NODES_CREATED]++;
        return idx;
    }
    AZ_HD void free_all_nodes() {
        int16_t* fs = m.free_stack + (size_t)g * c.max_nodes;
        const int mn = c.max_nodes;
        for (int base = 0; base < mn; base += AZ_WAVE)
            Wv::lanes([&](int lane) {
                const int i = base + lane;
                if (i < mn) {
                    fs[i] = (int16_t)(mn - 1 - i);  // pops yield 0,1,2,...
                    hdr(i).parent = -2;             // tag: not in use (see reroot)
                }
            });
        if (Wv::first()) {
            gr.n_free = mn;
            gr.root = -1;
        }
        Wv::sync();
    }

    // ---- game lifecycle ---------------------------------------------------------------------
    AZ_HD void stage_begin() {  // claim a staging buffer for the game that starts now
        int* sh = m.stg_hdr + ((size_t)g * 2 + gr.cur_buf) * SH_COUNT;
        if (Wv::first()) {
            sh[SH_STATE] = AZB_FILLING;
            sh[SH_LEN] = 0;
        }
    }
    AZ_HD void new_game() {
        if (Wv::first()) {
            R::reset(gr.env, GAME);
            if (GAME == AZ_GOMOKU) gr.env.flags = 0;
            gr.ply = 0;
            gr.num_passes = 0;
            gr.marked_player = -1;
            gr.uid = gr.games_done * c.G + g;
            gr.n_leaves = 0;
            gr.root_eval_pending = 0;
            gr.noise_pending = 0;
            gr.resign_thr = c.has_resign ? c.resign_threshold : -1.0;  // pipeline.py:241-242: re-read before every game
            gr.train_steps = c.training_steps;
            int rd = 1;  // pipeline.py:244-246
            if (c.force_resign_disabled >= 0) rd = c.force_resign_disabled;
            else if (c.has_resign && gr.resign_thr > -1.0) {
                u32 r[4];
                Philox::gen(c.seed + (u64)c.rank, (u32)g, (u32)gr.uid, 0u, 0x5E51u, r);
                rd = Philox::u01(r[0], r[1]) > (double)c.disable_resign_ratio ? 0 : 1;
            }
            gr.resign_disabled = rd;
            gr.status = AZS_NEED_ROOT;
        }
        Wv::lanes([&](int lane) {
            u64* flat = &gr.hist[0][0][0];
            for (int t = lane; t < 16 * W; t += AZ_WAVE) flat[t] = 0;
        });
        Wv::sync();
        free_all_nodes();
        stage_begin();
        Wv::sync();
    }

    // ---- row arithmetic (operand precision of the NumPy original) ----------------------------
    // root W: Python float while the root is fresh, np.float32 after a re-root (mcts_v2.py:439-443)
    AZ_HD void root_add_W(double v) {  // (between hot_load() and hot_store())
        if (hs_fresh) hs_rootW = hs_rootW + v;
        else hs_rootW = (double)((float)hs_rootW + (float)v);
    }
    // Apply `delta` to W (and +1 to N when count) on every edge of a path and on the root slot.
    // Edge d (0 = root's child) receives delta * (-1)^(depth-1-d); the root receives delta*(-1)^depth.
    // One lane per edge: edges of one path are distinct (parent,move) slots, and the same slot is
    // always handled by the same lane (its depth), so program order == reference order per slot.
    AZ_HD void path_update(const int* path, int depth, int leaf, float delta, bool flip, bool count) {
        if (depth <= AZ_PATH_CAP) {
            Wv::lanes([&](int lane) {
                if (lane < depth) {
                    const int e = path[lane];
                    const int node = e >> 16, mv = e & 0xffff;
                    const float v = (flip && ((depth - 1 - lane) & 1)) ? -delta : delta;
                    float* w = rowW(node) + mv;
                    *w = *w + v;  // float32 += (mcts_v2.py:230 / :466 / :481)
                    if (count) {
                        float* n = rowN(node) + mv;
                        *n = *n + 1.0f;
                    }
                }
            });
            root_add_W((double)((flip && (depth & 1)) ? -delta : delta));
        } else {
            // deeper than the lane-parallel path store (rare): walk the parent links from the leaf, exactly like the
            // reference's `while isinstance(node, Node)` loop
            float v = delta;
            int node = leaf;
            for (int hops = 0; hops < c.max_nodes; ++hops) {
                const int parent = Wv::uni((int)hdr(node).parent), mv = Wv::uni((int)hdr(node).move);
                if (parent < 0) break;
                if (Wv::first()) {
                    float* w = rowW(parent) + mv;
                    *w = *w + v;
                    if (count) {
                        float* n = rowN(parent) + mv;
                        *n = *n + 1.0f;
                    }
                }
                if (flip) v = -v;
                node = parent;
            }
            root_add_W((double)v);
        }
        if (count) hs_rootN += 1;
        if (count) cnt[AZC_BACKUP_EDGES] += (u64)depth + 1;
        Wv::sync();
    }

    // ---- PUCT selection (mcts_v2.py:99-109, :142-185) -----------------------------------------
    // Stage one whole node record (header + N/W/P/child rows, + the float64 root priors) into LDS: all loads of the
    // burst are in flight together, so a tree level costs ONE HBM/L2 round trip instead of one per dependent field.
    typedef StagedRec<AP, HW> SR;
    AZ_HD void stage_node(int node, SR& dst, bool with_root_p) {
        const float* rn = rowN(node);
        const float* rw = rowW(node);
        const float* rp = rowP(node);
        const int16_t* rc = rowC(node);
        const int16_t* rh = rowH(node);
        const u64a* hw = (const u64a*)rec(node);
        const double* rp64 = rootP();
        Wv::lanes([&](int lane) {
            if (lane < HW) dst.hdrw[lane] = hw[lane];
            for (int a = lane; a < AP; a += AZ_WAVE) {
                dst.rN[a] = rn[a];
                dst.rW[a] = rw[a];
                dst.rP[a] = rp[a];
                dst.rC[a] = rc[a];
                dst.rH[a] = rh[a];
                if (with_root_p) sc.rP64[a] = rp64[a];
            }
        });
        Wv::sync();
    }
    // Two records in ONE burst: the node the descent moves to and the node it will probably reach next (its hint).  All loads of
    // both records are issued before the first LDS store, so the pair costs one HBM / L2 round trip.
    AZ_HD void stage_two(int na, SR& da, int nb, SR& db) {
        const unsigned char* ra = rec(na);
        const unsigned char* rb = rec(nb);
        Wv::lanes([&](int lane) {
            u64a ha = 0, hb = 0;
            if (lane < HW) {
                ha = ((const u64a*)ra)[lane];
                hb = ((const u64a*)rb)[lane];
            }
            float an[EPL], aw[EPL], ap[EPL], bn[EPL], bw[EPL], bp[EPL];
            int16_t ac[EPL], ah[EPL], bc[EPL], bh[EPL];
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int a = lane + AZ_WAVE * j;
                if (a < AP) {
                    an[j] = ((const float*)(ra + HDR))[a];
                    aw[j] = ((const float*)(ra + HDR + AP * 4))[a];
                    ap[j] = ((const float*)(ra + HDR + 2 * AP * 4))[a];
                    ac[j] = ((const int16_t*)(ra + HDR + 3 * AP * 4))[a];
                    ah[j] = ((const int16_t*)(ra + HDR + 3 * AP * 4 + AP * 2))[a];
                    bn[j] = ((const float*)(rb + HDR))[a];
                    bw[j] = ((const float*)(rb + HDR + AP * 4))[a];
                    bp[j] = ((const float*)(rb + HDR + 2 * AP * 4))[a];
                    bc[j] = ((const int16_t*)(rb + HDR + 3 * AP * 4))[a];
                    bh[j] = ((const int16_t*)(rb + HDR + 3 * AP * 4 + AP * 2))[a];
                }
            }
            if (lane < HW) {
                da.hdrw[lane] = ha;
                db.hdrw[lane] = hb;
            }
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const int a = lane + AZ_WAVE * j;
                if (a < AP) {
                    da.rN[a] = an[j], da.rW[a] = aw[j], da.rP[a] = ap[j], da.rC[a] = ac[j], da.rH[a] = ah[j];
                    db.rN[a] = bn[j], db.rW[a] = bw[j], db.rP[a] = bp[j], db.rC[a] = bc[j], db.rH[a] = bh[j];
                }
            }
        });
        Wv::sync();
    }
    static AZ_HD const Hdr& hdr_of(const SR& r) { return *(const Hdr*)r.hdrw; }

    AZ_HD int puct_argmax(const SR& r, bool at_root, double pbc64, float sq32) {
        const S& s = hdr_of(r).st;
        const float pbc32 = (float)pbc64;
        const bool noisy = at_root && hs_noisy;
        u64 lg[W];
        for (int i = 0; i < W; ++i) lg[i] = Wv::uni(s.legal[i]);
        const bool pass_ok = GAME == AZ_GO && !(Wv::uni((int)s.flags) & AZF_TERMINAL);
        cnt[AZC_NODE_VISITS]++;
        return Wv::argmax_first([&](int lane, double& best, int& bi) {
            for (int j = 0; j < EPL; ++j) {
                const int a = lane + 64 * j;  // word index of point a is j, its bit is the lane
                if (a >= A) continue;
                if (!(a < NP ? (j < W && ((lg[j < W ? j : 0] >> lane) & 1ull) != 0) : pass_ok)) continue;
                const float n = r.rN[a], w = r.rW[a];
                const float q = w / (n > 0.0f ? n : 1.0f);
                const float rr = sq32 / (1.0f + n);
                double sco;
                if (noisy) {
                    const double u = (pbc64 * sc.rP64[a]) * (double)rr;  // float64 priors at the noisy root
                    sco = (double)(-q) + u;
                } else {
                    const float u = (pbc32 * r.rP[a]) * rr;
                    sco = (double)(-q + u);
                }
                if (bi < 0 || sco > best) {
                    best = sco;
                    bi = a;
                }
            }
        });
    }

    // One descent from the root.  Returns 0 = leaf reached (unexpanded, non-terminal), 1 = terminal.
    // On return `leaf_state` holds the leaf's position (used for the observation planes).
    AZ_HD int descend(int& node_out, int& depth_out, S& leaf_state) {
        const int root = hs_root;
        int node = root, depth = 0, n_self = hs_rootN;
        if (hs_tabN != hs_rootN) {  // the root's table entries change only when its visit count does (terminal backups)
            table_entry(hs_rootN, hs_fresh != 0, hs_tabPbc, hs_tabSq);
            hs_tabN = hs_rootN;
        }
        double pbc64 = hs_tabPbc;
        float sq32 = hs_tabSq;
        // Speculation (performance only, never changes what is selected): every edge remembers the node the search last continued
        // to below its child (`hint`).  Moving to a child, its record AND the record of its hinted successor are fetched in one
        // burst; if the next arg-max then picks that successor, its record is already in LDS and a tree level costs no memory
        // round trip.  A prefetched record is only used when its node index equals the authoritative child index, and nothing
        // writes a node record between the prefetch and its use (one wave owns the game), so a stale hint merely wastes a load.
        int cur_is_nxt = 0;   // which LDS record holds `node` at levels >= 1: 0 = sc.cur, 1 = sc.nxt
        int spec = -1;        // node whose record sits in the OTHER record (valid when >= 0)
        for (;;) {
            const SR& r = depth == 0 ? sc.root : (cur_is_nxt ? sc.nxt : sc.cur);  // the root record lives in LDS for the whole round
            int mv = puct_argmax(r, depth == 0, pbc64, sq32);
            int child = Wv::uni((int)r.rC[mv]);
            const int hint = Wv::uni((int)r.rH[mv]);
            n_self = Wv::uni((int)r.rN[mv]);
            table_entry(n_self, false, pbc64, sq32);  // the child's pb_c / sqrt: in flight while its record is staged
            if (Wv::first()) {
                if (depth < AZ_PATH_CAP) {
                    sc.path[depth] = (node << 16) | mv;
                    sc.wpath[depth] = r.rW[mv];
                    sc.npath[depth] = r.rN[mv];
                }
                sc.anc[depth & 7] = node;
                const S& hs = hdr_of(r).st;
                for (int q = 0; q < 2; ++q)
                    for (int w = 0; w < W; ++w) sc.ancst[depth & 7][q][w] = hs.stones[q][w];
            }
            depth++;
            if (child < 0) {  // lazy child creation (mcts_v2.py:182-183); its position is computed once, here
                child = alloc_node();
                S ns;
                R::template step<GAME>(hdr_of(r).st, mv, c.rc, ns);
                if (Wv::first()) {
                    Hdr& h = hdr(child);
                    h.st = ns;
                    h.parent = (int16_t)node;
                    h.move = (int16_t)mv;
                    h.expanded = 0;
                    rowC(node)[mv] = (int16_t)child;
                    if (node == root) sc.root.rC[mv] = (int16_t)child;
                }
                note_hint(depth - 1, child);
                leaf_state = ns;
                node_out = child;
                depth_out = depth;
                Wv::sync();
                return (ns.flags & AZF_TERMINAL) ? 1 : 0;
            }
            note_hint(depth - 1, child);
            node = child;
            if (depth >= 2 && spec == child) {
                cur_is_nxt ^= 1;  // predicted: the record is already staged in the other buffer
                spec = -1;
                cnt[AZC_HINT_HITS]++;
            } else {
                SR& dst = depth == 1 ? sc.cur : (cur_is_nxt ? sc.nxt : sc.cur);
                if (depth == 1) cur_is_nxt = 0;
                SR& oth = cur_is_nxt ? sc.cur : sc.nxt;
                if (hint >= 0 && hint != node && hint < c.max_nodes) {
                    stage_two(node, dst, hint, oth);
                    spec = hint;
                    cnt[AZC_HINT_PREFETCH]++;
                } else {
                    stage_node(node, dst, false);
                    spec = -1;
                }
            }
            const SR& cr = cur_is_nxt ? sc.nxt : sc.cur;
            const Hdr& h = hdr_of(cr);
            const int hflags = Wv::uni((int)h.st.flags), hexp = Wv::uni((int)h.expanded);
            if ((hflags & AZF_TERMINAL) || !hexp) {
                leaf_state = R::uni_state(h.st);
                node_out = node;
                depth_out = depth;
                Wv::sync();
                return (hflags & AZF_TERMINAL) ? 1 : 0;
            }
        }
    }
    // The descent just moved from the node of path level `lvl` to `succ`: remember it on the edge ABOVE that node (the hint of the
    // grandparent's edge), in memory and in the LDS copy of the root record.
    AZ_HD void note_hint(int lvl, int succ) {
        if (lvl < 1 || lvl > AZ_PATH_CAP) return;
        if (Wv::first()) {
            const int e = sc.path[lvl - 1];
            const int gp = e >> 16, gmv = e & 0xffff;
            rowH(gp)[gmv] = (int16_t)succ;
            if (gp == hs_root) sc.root.rH[gmv] = (int16_t)succ;
        }
    }

    // Select-phase path update (virtual loss / terminal backup) from the values captured during the descent: plain stores,
    // no read-modify-write round trip.  The captured W / N ARE the memory contents (only this wave touches its game, and
    // nothing else ran since the capture), so the float32 results are identical to the in-memory += of the reference.
    AZ_HD void path_apply(int depth, int leaf, float delta, bool flip, bool count) {
        if (depth > AZ_PATH_CAP) {  // rare deep path: parent-link walk in memory, then refresh the LDS copy of the root
            path_update(sc.path, depth, leaf, delta, flip, count);
            stage_node(hs_root, sc.root, false);
            return;
        }
        const int root = hs_root;
        Wv::lanes([&](int lane) {
            if (lane < depth) {
                const int e = sc.path[lane];
                const int node = e >> 16, mv = e & 0xffff;
                const float v = (flip && ((depth - 1 - lane) & 1)) ? -delta : delta;
                const float w = sc.wpath[lane] + v;  // float32 += (mcts_v2.py:230 / :466)
                rowW(node)[mv] = w;
                if (node == root) sc.root.rW[mv] = w;
                if (count) {
                    const float n = sc.npath[lane] + 1.0f;
                    rowN(node)[mv] = n;
                    if (node == root) sc.root.rN[mv] = n;
                }
            }
        });
        root_add_W((double)((flip && (depth & 1)) ? -delta : delta));
        if (count) hs_rootN += 1;
        if (count) cnt[AZC_BACKUP_EDGES] += (u64)depth + 1;
        Wv::sync();
    }

    // ---- observation planes (base.py:228-259) -------------------------------------------------
    // board k plies back from `leaf`: the leaf, its ancestors up to the root, then the real-game history
    AZ_HD void gather_planes(int leaf, int depth, int me, const S* leaf_state = nullptr) {
        if (leaf_state) {  // register-resident position -> LDS, so that the per-lane plane pick is an LDS index, not scratch
            if (Wv::first())
                for (int q = 0; q < 2; ++q)
                    for (int w = 0; w < W; ++w) sc.leafst[q][w] = leaf_state->stones[q][w];
            Wv::sync();
        }
        Wv::lanes([&](int lane) {
            for (int t = lane; t < 16 * W; t += AZ_WAVE) {
                const int w = t % W, pc = t / W, k = pc >> 1, col = (pc & 1) ? 1 - me : me;
                u64 v;
                if (depth < 0) v = gr.hist[k][col][w];  // the real position: the history ring itself
                else if (k == 0) v = leaf_state ? sc.leafst[col][w] : hdr(leaf).st.stones[col][w];
                else if (k <= depth) v = sc.ancst[(depth - k) & 7][col][w];
                else v = sc.hist[k - depth][col][w];
                sc.planes[pc][w] = v;
            }
        });
        Wv::sync();
    }
    template <class T> AZ_HD void emit_planes(T* out, T one, int me) {
        const int total = 17 * NP;
        const bool black = me == 0;
        Wv::lanes([&](int lane) {
            for (int e = lane; e < total; e += AZ_WAVE) {
                const int pl = e / NP, p = e - pl * NP;
                bool v = pl < 16 ? ((sc.planes[pl][p >> 6] >> (p & 63)) & 1ull) : black;
                out[e] = v ? one : (T)0;
            }
        });
    }
    // The evaluator's tiled input layout (include/azsp.h azsp_stem_tiled): [tile = T rows][4 chunks][T NP positions][8] bf16 (or f16) with
    // T = max(1, 256 / NP) boards per tile (3 at 9x9), the 17 planes zero-padded to 32 channels.  Chunks 0..1 = the 16 stone planes, chunk 2 = colour plane + 7 zeros; chunk 3 and
    // the padding are never written (the tensor is zero-initialised by its owner).
    // AZ_FEAT_F16_SPLIT = the input of the fp32-class evaluator's stem (include/azsp.h azsp_stem_split): the split layout
    // [row][plane: hi, lo][4 chunks][NP positions][8] f16 -- the same chunk strips with ONE board per tile and a second (lo) plane that is
    // never written: observation planes are 0 / 1, exactly representable in f16, so their lo halves are zero (the owner zero-initialises).
    template <int TBF, int PLANES> AZ_HD void emit_tiled_t(void* feat, int slot, int me, u32 one) {
        const size_t r = (size_t)g * c.P + slot, tile = r / TBF;
        const int sub = (int)(r - tile * TBF);
        uint16_t* base = (uint16_t*)feat + tile * (size_t)(PLANES * 4 * TBF * NP * 8) + (size_t)sub * NP * 8;
        const u32 black = me == 0 ? one : 0u;
        // One lane per POSITION (81 positions: lanes 0-63, then 0-16): the 16 plane words of its 64-position group are
        // wave-uniform LDS reads (broadcast), a stone is one bit-field extract, two planes make one dword (bf16 1.0 = 0x3F80) with two
        // multiply-adds, a chunk is one 16-byte store.  (The first version walked (chunk, position) pairs with 8 per-lane 64-bit
        // shifts each: 54 % of the select kernel, tools/sel_abl.sh.)
        Wv::lanes([&](int lane) {
            for (int p = lane; p < NP; p += AZ_WAVE) {
                const int w = p >> 6, b = p & 63;
                const bool hi_half = b >= 32;
                const u32 sh = (u32)(b & 31);
#pragma unroll
                for (int cc = 0; cc < 2; ++cc) {
                    u32 d[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const u64 w0 = sc.planes[cc * 8 + 2 * k][w], w1 = sc.planes[cc * 8 + 2 * k + 1][w];
                        const u32 x0 = hi_half ? (u32)(w0 >> 32) : (u32)w0, x1 = hi_half ? (u32)(w1 >> 32) : (u32)w1;
                        d[k] = ((x0 >> sh) & 1u) * one + ((x1 >> sh) & 1u) * (one << 16);
                    }
                    u32* o = (u32*)(base + ((size_t)cc * TBF * NP + p) * 8);
                    o[0] = d[0];
                    o[1] = d[1];
                    o[2] = d[2];
                    o[3] = d[3];
                }
                u32* o = (u32*)(base + ((size_t)2 * TBF * NP + p) * 8);
                o[0] = black;
                o[1] = 0u;
                o[2] = 0u;
                o[3] = 0u;
            }
        });
    }
    AZ_HD void write_features(void* feat, int slot, int me) {
        if (c.feat_dtype == AZ_FEAT_BF16_TILED || c.feat_dtype == AZ_FEAT_F16_TILED) {
            constexpr int TBF = (256 / NP) > 0 ? 256 / NP : 1;
            emit_tiled_t<TBF, 1>(feat, slot, me, c.feat_dtype == AZ_FEAT_F16_TILED ? 0x3C00u : 0x3F80u);  // 1.0 in f16 / bf16
            return;
        }
        if (c.feat_dtype == AZ_FEAT_F16_SPLIT) {
            emit_tiled_t<1, 2>(feat, slot, me, 0x3C00u);
            return;
        }
        const size_t row = ((size_t)g * c.P + slot) * (size_t)(17 * NP);
        switch (c.feat_dtype) {
            case AZ_FEAT_I8: emit_planes<int8_t>((int8_t*)feat + row, (int8_t)1, me); break;
            case AZ_FEAT_F32: emit_planes<u32>((u32*)feat + row, 0x3F800000u, me); break;
            case AZ_FEAT_BF16: emit_planes<uint16_t>((uint16_t*)feat + row, (uint16_t)0x3F80u, me); break;
            default: emit_planes<uint16_t>((uint16_t*)feat + row, (uint16_t)0x3C00u, me); break;
        }
    }

    // ---- select phase -------------------------------------------------------------------------
    AZ_HD void select(void* feat, unsigned char* valid) {
        unsigned char* vrow = valid + (size_t)g * c.P;
        if (Wv::first()) sc.free_base = -(1 << 30);  // no staged free-stack window yet
        Wv::sync();
        const int status = Wv::uni(gr.status);
        hot_load();
        if (status == AZS_NEED_ROOT) {
            // mcts_v2.py:364-368: the root position itself is evaluated first
            if (hs_root < 0) {
                const int r = alloc_node();
                if (Wv::first()) {
                    Hdr& h = hdr(r);
                    h.st = gr.env;
                    h.parent = -1;
                    h.move = -1;
                    h.expanded = 0;
                }
                hs_root = r;
                Wv::sync();
            }
            gather_planes(hs_root, -1, gr.env.to_play);  // the root IS the real position: hist[0..7]
            write_features(feat, 0, gr.env.to_play);
            Wv::lanes([&](int lane) {
                if (lane < c.P) vrow[lane] = lane == 0 ? 1 : 0;
            });
            if (Wv::first()) {
                gr.root_eval_pending = 1;
                gr.n_leaves = 0;
            }
            cnt[AZC_ROOT_EVALS]++;
            hot_store();
            return;
        }
        int nleaf = 0;
        if (status == AZS_SEARCH && !Wv::uni(gr.noise_pending)) {
            int attempts = 0;
            {  // the next pops of the free stack, staged once per round
                const int base = hs_nfree - AZ_FREE_PREFETCH;
                const int16_t* fs = m.free_stack + (size_t)g * c.max_nodes;
                Wv::lanes([&](int lane) {
                    const int i = base + lane;
                    if (i >= 0) sc.freetop[lane] = fs[i];
                });
                if (Wv::first()) sc.free_base = base;
                // the root record and the real game's history: staged once, then kept current in LDS
                const u64* gh = &gr.hist[0][0][0];
                u64* lh = &sc.hist[0][0][0];
                Wv::lanes([&](int lane) {
                    for (int t = lane; t < 16 * W; t += AZ_WAVE) lh[t] = gh[t];
                });
                Wv::sync();
                stage_node(hs_root, sc.root, hs_noisy != 0);
            }
            // mcts_v2.py:572: up to P leaves in at most 2P attempts, one after another (each descent
            // sees the virtual losses of the previous ones); uct_search (:378-418) is the P == 1 case
            const int max_att = c.parallel_mode ? 2 * c.P : 1;
            while (nleaf < c.P && attempts < max_att) {
                attempts++;
                int node, depth;
                S leaf;
                const int term = descend(node, depth, leaf);
                cnt[AZC_SIMS]++;
                if (term) {
                    // mcts_v2.py:407-411 / :604-608: back up -reward, node stays unexpanded
                    cnt[AZC_TERMINAL_HITS]++;
                    path_apply(depth, node, (float)(-(int)leaf.reward), true, true);
                    if (!c.parallel_mode && hs_rootN < c.budget) attempts = 0;  // uct_search keeps looping (:378)
                    if (!c.parallel_mode && hs_rootN >= c.budget) break;
                    continue;
                }
                if (c.parallel_mode) path_apply(depth, node, 1.0f, false, false);  // add_virtual_loss :453-467
                int* lp = leaf_path(nleaf);
                Wv::lanes([&](int lane) {
                    if (lane < depth && lane < AZ_PATH_CAP) lp[lane] = sc.path[lane];
                });
                if (Wv::first()) {
                    gr.leaf_node[nleaf] = (int16_t)node;
                    gr.leaf_depth[nleaf] = (uint8_t)(depth < 255 ? depth : 255);
                }
                const int me = leaf.to_play;
                gather_planes(node, depth, me, &leaf);
                write_features(feat, nleaf, me);
                nleaf++;
                cnt[AZC_LEAVES]++;
            }
        }
        Wv::lanes([&](int lane) {
            if (lane < c.P) vrow[lane] = lane < nleaf ? 1 : 0;
        });
        if (Wv::first()) gr.n_leaves = nleaf;
        hot_store();
    }

    // ---- expand + backup phase (mcts_v2.py:188-232, :616-625) -----------------------------------
    AZ_HD void expand_node(int node, const float* prior) {
        float* rn = rowN(node);
        float* rw = rowW(node);
        float* rp = rowP(node);
        int16_t* rc = rowC(node);
        int16_t* rh = rowH(node);
        Wv::lanes([&](int lane) {
            for (int a = lane; a < AP; a += AZ_WAVE) {
                rp[a] = a < A ? prior[a] : 0.0f;  // stored unmasked, not renormalised (:209)
                rn[a] = 0.0f;
                rw[a] = 0.0f;
                rc[a] = -1;
                rh[a] = -1;
            }
        });
        if (Wv::first()) hdr(node).expanded = 1;
        Wv::sync();
    }
    AZ_HD void apply_outputs(const float* priors, const float* values) {
        const size_t row0 = (size_t)g * c.P;
        if (Wv::uni(gr.root_eval_pending)) {
            expand_node(Wv::uni(gr.root), priors + row0 * A);
            if (Wv::first()) {
                gr.root_N = 1;  // backup(root, value) on DummyNode slots: 0.0 + 1, 0.0 + value
                gr.root_W = (double)values[row0];
                gr.root_fresh = 1;
                gr.root_noisy = 0;
                gr.root_eval_pending = 0;
                gr.status = AZS_SEARCH;
                gr.noise_pending = c.root_noise;
            }
            Wv::sync();
            return;
        }
        const int nl = Wv::uni(gr.n_leaves);
        if (nl == 0) return;
        hot_load();
        for (int s = 0; s < nl; ++s) {
            const int node = Wv::uni((int)gr.leaf_node[s]), depth = Wv::uni((int)gr.leaf_depth[s]);
            const int* lp = leaf_path(s);
            if (c.parallel_mode) path_update(lp, depth, node, -1.0f, false, false);  // revert_virtual_loss :470-482
            if (Wv::uni((int)hdr(node).expanded)) {  // picked twice in one round: evaluation wasted (:621-622)
                cnt[AZC_DUP_LEAVES]++;
                continue;
            }
            expand_node(node, priors + (row0 + s) * A);
            path_update(lp, depth, node, values[row0 + s], true, true);
        }
        if (Wv::first()) gr.n_leaves = 0;
        hot_store();
    }

    // ---- Dirichlet noise at the root (mcts_v2.py:235-262) ---------------------------------------
    AZ_HD double gamma_sample(double alpha, u64 key, u32 c0, u32 c1, u32 c2) {
        // Marsaglia-Tsang for alpha+1, boosted by U^(1/alpha) (alpha < 1)
        const double d = alpha + 1.0 - 1.0 / 3.0, cc = 1.0 / sqrt(9.0 * d);
        for (u32 it = 0;; ++it) {
            u32 r[4], q[4];
            Philox::gen(key, c0, c1, c2, 2u * it, r);
            Philox::gen(key, c0, c1, c2, 2u * it + 1u, q);
            const double u1 = Philox::u01(r[0], r[1]) + 1.0 / 18014398509481984.0, u2 = Philox::u01(r[2], r[3]);
            const double x = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
            const double v0 = 1.0 + cc * x;
            if (v0 <= 0.0) continue;
            const double v = v0 * v0 * v0, u = Philox::u01(q[0], q[1]) + 1.0 / 18014398509481984.0;
            if (log(u) < 0.5 * x * x + d - d * v + d * log(v) || it > 64) {
                const double ub = Philox::u01(q[2], q[3]) + 1.0 / 18014398509481984.0;
                return d * v * exp(log(ub) / alpha);
            }
        }
    }
    AZ_HD void apply_noise() {
        const int root = gr.root;
        const S& s = hdr(root).st;
        const float* rp = rowP(root);
        double* rp64 = rootP();
        const double* inj = c.inject ? m.inj_noise + ((size_t)g * c.inj_moves + (gr.ply < c.inj_moves ? gr.ply : c.inj_moves - 1)) * A : nullptr;
        double total = 1.0;
        if (!c.inject) {
            // Dirichlet(alpha) over ALL actions = normalised Gamma(alpha) draws (:259-260)
            const u64 key = c.seed + (u64)c.rank;
            const double part = Wv::sum_f64([&](int lane) -> double {
                double acc = 0.0;
                for (int a = lane; a < A; a += AZ_WAVE) {
                    const double gsample = gamma_sample(c.alpha, key, (u32)g, (u32)gr.uid, ((u32)gr.ply << 12) | (u32)a);
                    rp64[a] = gsample;
                    acc += gsample;
                }
                return acc;
            });
            Wv::sync();
            total = part > 0.0 ? part : 1.0;
        }
        Wv::lanes([&](int lane) {
            for (int a = lane; a < AP; a += AZ_WAVE) {
                double v = 0.0;
                if (a < A) {
                    const double nz = c.inject ? inj[a] : rp64[a] / total;
                    // child_P * (1 - eps) stays float32, noise * eps is float64, the sum is float64 (:262)
                    v = (double)(rp[a] * c.one_minus_eps_f32) + (action_legal(s, a) ? nz : 0.0) * c.eps;
                }
                rp64[a] = v;
            }
        });
        if (Wv::first()) {
            gr.root_noisy = 1;
            gr.noise_pending = 0;
        }
        Wv::sync();
    }

    // Diagnostics (azsp_rng_probe): the production random streams of this slot for plies 0..plies-1 of its current game, drawn
    // exactly as apply_noise / sample_move draw them, without touching any state.
    AZ_HD void probe_rng(double* noise, double* unif, int plies, int tries) {
        const u64 key = c.seed + (u64)c.rank;
        const u32 uid = (u32)Wv::uni((int)gr.uid);
        for (int ply = 0; ply < plies; ++ply) {
            if (noise) {
                double* row = noise + ((size_t)g * plies + ply) * A;
                const double tot = Wv::sum_f64([&](int lane) -> double {
                    double acc = 0.0;
                    for (int a = lane; a < A; a += AZ_WAVE) {
                        const double gs = gamma_sample(c.alpha, key, (u32)g, uid, ((u32)ply << 12) | (u32)a);
                        row[a] = gs;
                        acc += gs;
                    }
                    return acc;
                });
                Wv::sync();
                Wv::lanes([&](int lane) {
                    for (int a = lane; a < A; a += AZ_WAVE) row[a] = row[a] / (tot > 0.0 ? tot : 1.0);
                });
            }
            if (unif) {
                Wv::lanes([&](int lane) {
                    for (int t = lane; t < tries; t += AZ_WAVE) {
                        u32 r[4];
                        Philox::gen(key, (u32)g, uid, ((u32)ply << 12) | 0xFFFu, 0x1000u + (u32)t, r);
                        unif[((size_t)g * plies + ply) * tries + t] = Philox::u01(r[0], r[1]);
                    }
                });
            }
        }
        Wv::sync();
    }

    // ---- end of a search: policy, move, sample, env step, re-root ---------------------------------
    static AZ_HD float pairwise_sum_f32(const float* a, int n) {
        // numpy's float32 add.reduce order (pairwise, 8 accumulators, blocks of 128), so that the
        // Gomoku policy (float32 in the reference) sums in the same order
        if (n < 8) {
            float r = 0.0f;
            for (int i = 0; i < n; ++i) r = r + a[i];
            return r;
        }
        if (n <= 128) {
            float r[8];
            for (int j = 0; j < 8; ++j) r[j] = a[j];
            int i = 8;
            for (; i < n - (n % 8); i += 8)
                for (int j = 0; j < 8; ++j) r[j] = r[j] + a[i + j];
            float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
            for (; i < n; ++i) res = res + a[i];
            return res;
        }
        int n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum_f32(a, n2) + pairwise_sum_f32(a + n2, n - n2);
    }

    // generate_search_policy (mcts_v2.py:265-298) into the wave's LDS row sc.pi[A] (and the move log when enabled).
    AZ_HD void search_policy(bool warm, double* log_pi) {
        const S& s = hdr(gr.root).st;
        const float* rn = rowN(gr.root);
        if (GAME == AZ_GO) {
            // legal(int64) * child_N(float32) -> float64; n**5 and the sum are exact integers in float64, so the
            // summation order is irrelevant and a wave reduction is bit-identical to np.sum
            const double sum = Wv::sum_f64([&](int lane) -> double {
                double part = 0.0;
                for (int a = lane; a < A; a += AZ_WAVE) {
                    double x = action_legal(s, a) ? (double)rn[a] : 0.0;
                    if (!warm) {
                        const double x2 = x * x;
                        x = x2 * x2 * x;
                    }
                    sc.pi[a] = x;
                    part += x;
                }
                return part;
            });
            Wv::sync();
            if (sum > 0.0) {
                Wv::lanes([&](int lane) {
                    for (int a = lane; a < A; a += AZ_WAVE) sc.pi[a] = sc.pi[a] / sum;
                });
            }
        } else {
            // legal(int8) * child_N(float32) stays float32 (Gomoku); np.sum order = pairwise_sum_f32
            Wv::lanes([&](int lane) {
                for (int a = lane; a < A; a += AZ_WAVE) {
                    float x = action_legal(s, a) ? rn[a] : 0.0f;
                    if (!warm) {
                        const double d = (double)x, d2 = d * d;
                        x = (float)(d2 * d2 * d);
                    }
                    sc.tmpf[a] = x;
                }
            });
            Wv::sync();
            const float sum = pairwise_sum_f32(sc.tmpf, A);
            Wv::lanes([&](int lane) {
                for (int a = lane; a < A; a += AZ_WAVE) {
                    const float x = sc.tmpf[a];
                    sc.pi[a] = (double)(sum > 0.0f ? x / sum : x);
                }
            });
        }
        Wv::sync();
        if (log_pi) {
            Wv::lanes([&](int lane) {
                for (int a = lane; a < A; a += AZ_WAVE) log_pi[a] = sc.pi[a];
            });
        }
    }

    // np.random.choice(p=pi): cdf = cumsum(p) (sequential, float64), cdf /= cdf[-1], searchsorted(cdf, u, 'right').
    // The cumulative sums are formed once, in order, in LDS; fl(cdf[a] / total) is monotone in a, so the first index
    // whose quotient exceeds u is found by bisection with the very same divisions NumPy performs.
    AZ_HD int sample_move(bool warm) {
        const S& s = hdr(gr.root).st;
        if (Wv::first()) {
            double acc = 0.0;
            for (int a = 0; a < A; ++a) {
                acc += sc.pi[a];
                sc.cdf[a] = acc;
            }
        }
        Wv::sync();
        const double tot = sc.cdf[A - 1];
        const double* inj = c.inject ? m.inj_unif + ((size_t)g * c.inj_moves + (gr.ply < c.inj_moves ? gr.ply : c.inj_moves - 1)) * AZ_INJ_K : nullptr;
        int mv = -1;
        for (int t = 0; t < 64; ++t) {
            double u;
            if (c.inject) u = inj[t < AZ_INJ_K ? t : AZ_INJ_K - 1];
            else {
                u32 r[4];
                Philox::gen(c.seed + (u64)c.rank, (u32)g, (u32)gr.uid, ((u32)gr.ply << 12) | 0xFFFu, 0x1000u + (u32)t, r);
                u = Philox::u01(r[0], r[1]);
            }
            int lo = 0, hi = A;  // first a in [0, A) with cdf[a]/tot > u, else A
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (sc.cdf[mid] / tot > u) hi = mid;
                else lo = mid + 1;
            }
            mv = lo < A ? lo : A - 1;
            // mcts_v2.py:433 / :640: redraw while pass during warm-up, or illegal
            const bool bad = (warm && GAME == AZ_GO && mv == NP) || !action_legal(s, mv);
            if (!bad) return mv;
        }
        // the reference would loop forever here
        fail(AZ_ERR_SAMPLE);
        return mv;
    }

    AZ_HD void record_sample() {
        int* sh = m.stg_hdr + ((size_t)g * 2 + gr.cur_buf) * SH_COUNT;
        const int k = sh[SH_LEN];
        if (k >= c.stage_cap) {
            fail(AZ_ERR_STAGE);
            return;
        }
        const size_t idx = ((size_t)g * 2 + gr.cur_buf) * c.stage_cap + k;
        u64* pl = m.stg_planes + idx * 16 * W;
        float* pi = m.stg_pi + idx * A;
        const int me = gr.env.to_play;
        // observation BEFORE the move, mover's perspective (pipeline.py:323): the history ring itself
        Wv::lanes([&](int lane) {
            for (int t = lane; t < 16 * W; t += AZ_WAVE) {
                const int w = t % W, pc = t / W, kk = pc >> 1, col = (pc & 1) ? 1 - me : me;
                pl[t] = gr.hist[kk][col][w];
            }
            for (int a = lane; a < A; a += AZ_WAVE) pi[a] = (float)sc.pi[a];
        });
        if (Wv::first()) {
            m.stg_meta[idx] = me == 0 ? 1 : 0;
            sh[SH_LEN] = k + 1;
        }
        Wv::sync();
    }

    AZ_HD void push_history(const S& s) {
        // hist[k] <- hist[k-1], hist[0] <- new board (base.py deque.appendleft); 16*W words moved lane-parallel via LDS
        u64* flat = &gr.hist[0][0][0];
        Wv::lanes([&](int lane) {
            for (int t = lane; t < 16 * W; t += AZ_WAVE) {
                const int w = t % W, kq = t / W, k = kq >> 1, q = kq & 1;
                sc.planes[kq][w] = k == 0 ? s.stones[q][w] : flat[t - 2 * W];
            }
        });
        Wv::sync();
        Wv::lanes([&](int lane) {
            for (int t = lane; t < 16 * W; t += AZ_WAVE) flat[t] = sc.planes[t / W][t % W];
        });
        Wv::sync();
    }

    // Keep the subtree below `child`, return every other node to the free stack (mcts_v2.py:436-446).
    AZ_HD void reroot(int child, int mv) {
        const int old_root = gr.root;
        const float cn = rowN(old_root)[mv], cw = rowW(old_root)[mv];
        int16_t* fs = m.free_stack + (size_t)g * c.max_nodes;
        const int mn = c.max_nodes;
        const bool lds = mn <= AZ_LDS_NODES;
        // A node is kept iff walking up its parents reaches `child` before the old root.  Unused nodes carry
        // parent == -2.  The parent links are first gathered into LDS (one strided load per node, all in flight
        // together); the walks then run at LDS latency instead of one dependent global load per hop.
        if (lds) {
            for (int base = 0; base < mn; base += AZ_WAVE)
                Wv::lanes([&](int lane) {
                    const int i = base + lane;
                    if (i < mn) sc.parent[i] = hdr(i).parent;
                });
            Wv::sync();
        }
        int nfree = gr.n_free;
        for (int base = 0; base < mn; base += AZ_WAVE) {
            const u64 drop = Wv::ballot([&](int lane) -> bool {
                const int i = base + lane;
                if (i >= mn) return false;
                int p = lds ? (int)sc.parent[i] : (int)hdr(i).parent;
                if (p == -2) return false;  // not in use
                int cur = i;
                for (int hops = 0; hops < mn; ++hops) {
                    if (cur == child) return false;  // kept
                    if (cur == old_root || p < 0) return true;
                    cur = p;
                    p = lds ? (int)sc.parent[cur] : (int)hdr(cur).parent;
                }
                return true;
            });
            Wv::sync();
            Wv::lanes([&](int lane) {
                if ((drop >> lane) & 1ull) {
                    const int pos = nfree + __builtin_popcountll(drop & ((1ull << lane) - 1ull));
                    fs[pos] = (int16_t)(base + lane);
                    hdr(base + lane).parent = -2;
                    if (lds) sc.parent[base + lane] = -2;  // later walks stop here (p < 0 => dropped)
                }
            });
            nfree += __builtin_popcountll(drop);
            Wv::sync();
        }
        if (Wv::first()) {
            gr.n_free = nfree;
            gr.root = child;
            hdr(child).parent = -1;
            gr.root_N = (int)cn;  // np.float32 copies of the child's slot (mcts_v2.py:439-443)
            gr.root_W = (double)cw;
            gr.root_fresh = 0;
            gr.root_noisy = 0;
            gr.noise_pending = c.root_noise;
        }
        Wv::sync();
    }

    AZ_HD void finalize_game(const S& fin, int last_player, int resigned) {
        int* sh = m.stg_hdr + ((size_t)g * 2 + gr.cur_buf) * SH_COUNT;
        if (Wv::first()) {
            sh[SH_WINNER] = fin.winner;
            sh[SH_AREA_B] = fin.area[0];
            sh[SH_AREA_W] = fin.area[1];
            sh[SH_PASSES] = gr.num_passes;
            sh[SH_RESIGNED] = resigned;
            sh[SH_RESIGN_DISABLED] = gr.resign_disabled;
            // pipeline.py:361-365
            const int marked = (c.has_resign && gr.resign_disabled && gr.marked_player >= 0) ? 1 : 0;
            sh[SH_MARKED] = marked;
            sh[SH_COULD_WON] = (marked && fin.winner == gr.marked_player) ? 1 : 0;
            sh[SH_MARKED_PLAYER] = gr.marked_player;
            sh[SH_UID] = gr.uid;
            sh[SH_TRAINING_STEPS] = gr.train_steps;  // the weights that STARTED the game (pipeline.py:237 -> :271)
            sh[SH_TS_END] = c.training_steps;        // != the above: the game straddled a weight hot-swap
            const u64 tb = az_bits_f64(gr.resign_thr);
            sh[SH_THR_LO] = (int)(unsigned)(tb & 0xffffffffull);
            sh[SH_THR_HI] = (int)(unsigned)(tb >> 32);
            sh[SH_REWARD] = fin.reward;  // reward of `last_player` (pipeline.py:349-354 turns it into z at harvest)
            sh[SH_LAST_PLAYER] = last_player;
            sh[SH_STATE] = AZB_COMPLETE;
            gr.games_done += 1;
        }
        cnt[AZC_GAMES]++;
        Wv::sync();
    }

    // Try to start the next game in the other staging buffer; stall (AZS_WAIT_BUF) until the host harvested it.
    AZ_HD void next_game() {
        if (c.stop_at_game_end) {
            if (Wv::first()) gr.status = AZS_IDLE;
            Wv::sync();
            return;
        }
        const int other = 1 - gr.cur_buf;
        const int* sh = m.stg_hdr + ((size_t)g * 2 + other) * SH_COUNT;
        if (sh[SH_STATE] != AZB_FREE) {
            if (Wv::first()) gr.status = AZS_WAIT_BUF;
            cnt[AZC_STALLS]++;
            Wv::sync();
            return;
        }
        if (Wv::first()) gr.cur_buf = other;
        Wv::sync();
        new_game();
    }

    // ---- end of a search (mcts_v2.py:421-450) ----------------------------------------------------
    AZ_HD double root_q() const {
        if (gr.root_N <= 0) return 0.0;
        return gr.root_fresh ? gr.root_W / (double)gr.root_N : (double)((float)gr.root_W / (float)gr.root_N);
    }
    AZ_HD int log_slot() const { return c.log_moves ? (gr.ply < c.log_cap ? gr.ply : c.log_cap - 1) : 0; }
    AZ_HD double* pi_slot() const { return m.log_pi + ((size_t)g * c.log_cap + log_slot()) * A; }

    // Policy + (batched mode) move choice.  In drop-in mode (stop_after_move) the caller samples the
    // move itself from the published pi, exactly like mcts_v2.py:433-434 does with np.random.choice.
    AZ_HD void search_done() {
        const bool warm = gr.warm_override >= 0 ? (gr.warm_override != 0) : !(gr.env.steps > c.warm_up_steps);  // pipeline.py:320
        search_policy(warm, (c.log_moves || c.stop_after_move) ? pi_slot() : nullptr);
        const float* rn = rowN(gr.root);
        if (c.log_moves || c.stop_after_move) {
            float* ln = m.log_childN + ((size_t)g * c.log_cap + log_slot()) * A;
            Wv::lanes([&](int lane) {
                for (int a = lane; a < A; a += AZ_WAVE) ln[a] = rn[a];
            });
        }
        if (c.stop_after_move) {
            if (Wv::first()) {
                gr.out_root_q = root_q();
                gr.status = AZS_MOVE_DONE;
            }
            Wv::sync();
            return;
        }
        int mv;
        if (c.deterministic) {
            mv = Wv::argmax_first([&](int lane, double& best, int& bi) {  // np.argmax(child_N), unmasked (:429)
                for (int a = lane; a < A; a += AZ_WAVE)
                    if (bi < 0 || (double)rn[a] > best) {
                        best = (double)rn[a];
                        bi = a;
                    }
            });
        } else {
            mv = sample_move(warm);
        }
        commit_actor(mv);
    }

    // best_child_Q = -Q(chosen child) in float32 (:446); 0.0 when the move has no child node (:425)
    AZ_HD double child_q_of(int mv, int& child) const {
        child = (mv >= 0 && mv < A) ? rowC(gr.root)[mv] : -1;
        if (child < 0) return 0.0;
        const float cn = rowN(gr.root)[mv], cw = rowW(gr.root)[mv];
        return cn > 0.0f ? (double)(-(cw / cn)) : -0.0;
    }
    AZ_HD void log_outputs(int mv, double rq, double cq) {
        if (Wv::first()) {
            gr.out_move = mv;
            gr.out_root_q = rq;
            gr.out_child_q = cq;
            if (c.log_moves && gr.ply < c.log_cap) {
                double* lq = m.log_q + ((size_t)g * c.log_cap + gr.ply) * 4;
                lq[0] = rq;
                lq[1] = cq;
                lq[2] = (double)gr.root_N;
                lq[3] = (double)mv;
            }
        }
        cnt[AZC_MOVES]++;
        Wv::sync();
    }

    // Drop-in uct_search: the caller steps its own env; only the tree advances (mcts_v2.py:436-446).
    AZ_HD void commit_host(int mv) {
        if (gr.status != AZS_MOVE_DONE) return;
        int child;
        const double cq = child_q_of(mv, child);
        log_outputs(mv, root_q(), cq);
        if (child >= 0 && c.reuse_tree && !(hdr(child).st.flags & AZF_TERMINAL)) {
            const S ns = hdr(child).st;
            if (Wv::first()) {
                gr.env = ns;
                gr.ply += 1;
            }
            Wv::sync();
            push_history(ns);
            reroot(child, mv);
            if (Wv::first()) gr.status = AZS_SEARCH;
        } else {
            free_all_nodes();
            if (Wv::first()) gr.status = AZS_IDLE;
        }
        if (Wv::first()) gr.noise_ready = 0;
        Wv::sync();
    }

    // Batched actor step (pipeline.py:323-346): sample, resignation rule, env step, re-root / game end.
    AZ_HD void commit_actor(int mv) {
        int child;
        const double rq = root_q();
        const double cq = child_q_of(mv, child);
        log_outputs(mv, rq, cq);
        if (child < 0) {  // cannot happen: a sampled / most-visited move always has a visited child
            fail(AZ_ERR_SAMPLE);
            if (Wv::first()) gr.status = AZS_IDLE;
            Wv::sync();
            return;
        }
        record_sample();
        const int mover = gr.env.to_play;
        bool resign = false;
        if (GAME == AZ_GO && c.has_resign && gr.resign_thr > -1.0 && gr.env.steps > c.check_resign_after) {
            // root_Q is a Python float on a fresh root, np.float32 otherwise; best_child_Q is np.float32
            const double thr = gr.resign_thr;  // this game's threshold (pipeline.py:241-242, :328-333)
            const bool lo_root = gr.root_fresh ? (rq < thr) : ((float)rq < (float)thr);
            const bool lo_child = (float)cq < (float)thr;
            if (lo_root && lo_child) {
                if (gr.marked_player < 0 && Wv::first()) gr.marked_player = mover;
                Wv::sync();
                if (!gr.resign_disabled) resign = true;
            }
        }
        S ns;
        if (resign) R::go_resign(gr.env, ns);
        else ns = hdr(child).st;
        if (Wv::first()) {
            // the move that leaves the recorded position: what env.history / to_sgf() would hold (resign is not a history move)
            const int* sh = m.stg_hdr + ((size_t)g * 2 + gr.cur_buf) * SH_COUNT;
            if (sh[SH_LEN] > 0) m.stg_move[((size_t)g * 2 + gr.cur_buf) * c.stage_cap + (sh[SH_LEN] - 1)] = (int16_t)(resign ? -1 : mv);
            if (GAME == AZ_GO && !resign && mv == NP) gr.num_passes += 1;
            gr.env = ns;
            gr.ply += 1;
        }
        Wv::sync();
        push_history(ns);
        if (ns.flags & AZF_TERMINAL) {
            finalize_game(ns, mover, resign ? 1 : 0);
            next_game();
            return;
        }
        if (c.max_plies > 0 && gr.ply >= c.max_plies) {
            if (Wv::first()) gr.status = AZS_IDLE;
            Wv::sync();
            return;
        }
        if (c.reuse_tree) {
            reroot(child, mv);
        } else {
            free_all_nodes();
            if (Wv::first()) gr.status = AZS_NEED_ROOT;
            Wv::sync();
        }
    }

    // ---- one engine round for this game ---------------------------------------------------------
    AZ_HD void flush_counters() {
        if (Wv::first()) {
            u64* row = m.counters + (size_t)g * AZC_COUNT;
            for (int i = 0; i < AZC_COUNT; ++i)
                if (cnt[i]) row[i] += cnt[i];
        }
    }
    // phase A: consume the evaluator's outputs
    AZ_HD void backup_phase(const float* priors, const float* values) {
        if (Wv::uni(gr.root_eval_pending) || Wv::uni(gr.n_leaves) > 0) apply_outputs(priors, values);
    }
    // phase B: finish as many moves as the budget allows (rare per round and per game)
    AZ_HD void endmove_phase() {
        if (Wv::uni(gr.status) == AZS_WAIT_BUF) next_game();
        for (int guard = 0; guard < 8; ++guard) {
            if (Wv::uni(gr.status) != AZS_SEARCH) break;
            if (Wv::uni(gr.noise_pending) && (!c.stop_after_move || Wv::uni(gr.noise_ready))) apply_noise();
            if (Wv::uni(gr.noise_pending)) break;  // drop-in mode: the noise vector arrives with begin_move
            if (Wv::uni(gr.root_N) < c.budget) break;
            search_done();
        }
    }
};
