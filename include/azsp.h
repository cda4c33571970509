/*
 * azsp.h -- C ABI of the MI355X batched self-play engine (libazsp.so).
 *
 * The upstream reference (michaelnny/alpha_zero) is pure Python and has no FFI; the "plugin API"
 * of its self-play hot path is three Python call surfaces.  This header is what a binding for
 * that path would bind (ctypes stub in INTEGRATION.md); each entry point names the reference
 * interface it replaces (paths relative to the reference repo):
 *
 *   azsp_create / azsp_destroy        engine for G concurrent games: replaces one actor process per game
 *                                     (alpha_zero/training_go.py:317-347, core/pipeline.py:166-218)
 *   azsp_reset_games                  env.reset() for every slot                (envs/base.py:93-112, envs/go.py:76-86)
 *   azsp_env_step                     BoardGameEnv.step / legal_actions / observation / score as a batch
 *                                     (envs/go.py:88-161, envs/gomoku.py:45-83, envs/go_engine.py:417-441,
 *                                      :123-152, envs/base.py:228-259)
 *   azsp_set_state                    the `env` argument of uct_search()        (core/mcts_v2.py:301-311)
 *   azsp_begin_move                   add_dirichlet_noise at the start of a search (core/mcts_v2.py:375-376, :565-566)
 *   azsp_select                       Phase 1 of (parallel_)uct_search: best_child descents, virtual loss,
 *                                     and env.observation() of the leaves       (core/mcts_v2.py:572-611)
 *   azsp_expand_backup                Phases 2-3: expand + backup, then (when the budget is met) policy,
 *                                     move, sample recording, env.step, re-root (core/mcts_v2.py:614-657,
 *                                      core/pipeline.py:323-346)
 *   azsp_round                        azsp_expand_backup + azsp_select in one launch
 *   azsp_get_status / azsp_get_search the tuple uct_search() returns            (core/mcts_v2.py:450)
 *   azsp_commit_move                  sub-tree reuse after the caller chose the move (core/mcts_v2.py:436-446)
 *   azsp_dropin_step                  one iteration of uct_search's simulation loop around the caller's eval_func: expand + backup of the
 *                                     evaluated leaves, selection of the next ones, and everything eval_func needs, in ONE host round trip
 *                                     (core/mcts_v2.py:378-421, :568-625)
 *   azsp_harvest                      data_queue.put((game_seq, stats))         (core/pipeline.py:283, :349-380)
 *   azsp_set_actor_state              per-game re-read of var_resign_threshold / checkpoint tag (core/pipeline.py:232-246)
 *   azsp_replay_gather                UniformReplay.sample + batch tensors + random transformation (core/replay.py:72-83,
 *                                      core/pipeline.py:636-643)
 *   azsp_dihedral                     apply_horizontal_flip / apply_vertical_flip / apply_rotation
 *                                     (utils/transformation.py:34-110)
 *   azsp_bias_act                     BatchNorm + residual add + ReLU after each convolution (core/network.py:42-82)
 *   azsp_conv3x3_tiled                a whole conv3x3 + BatchNorm (+ skip) + ReLU of a ResNetBlock (core/network.py:42-82)
 *   azsp_resblock_tiled               a whole ResNetBlock of a 64-filter tower in one launch (intermediate activation in LDS)
 *   azsp_tile_layout / azsp_tiled_bytes the tower's resident activation layout
 *   azsp_conv3x3_split                the same layer at the reference's fp32 precision class (core/pipeline.py:91-123 evaluates
 *                                     in fp32): values as hi + lo f16 pairs, three f16 MFMA products, fp32 accumulation
 *   azsp_split_layout / azsp_split_bytes the split-precision tower's activation layout
 *   azsp_split_features / azsp_stem_split / azsp_head_split   the stem and both heads of the same fp32-class evaluator
 *                                     (core/network.py:101-156)
 *   azsp_*_tiled_f16 / azsp_fc_heads_f16  the bf16 evaluator entries with f16 activations and weights
 *
 * Conventions: every function returns 0 on success or a negative AZSP_E* code; the message is
 * available from azsp_last_error().  No exceptions and no callbacks cross this boundary.  Pointers
 * named *_dev are device (HBM) addresses owned by the caller (e.g. torch tensors) and must stay
 * alive until `stream` (a hipStream_t, NULL = default stream) has drained; pointers named *_host
 * are host memory.  One host thread per engine.  Colours at this boundary use the reference's
 * ids: Go black = 1, white = -1; Gomoku black = 1, white = 2 (envs/go.py:63-64, envs/base.py:33-34).
 */
#ifndef AZSP_H
#define AZSP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AZSP_GAME_GO 0
#define AZSP_GAME_GOMOKU 1

#define AZSP_FEAT_I8 0
#define AZSP_FEAT_F32 1
#define AZSP_FEAT_BF16 2
#define AZSP_FEAT_F16 3
#define AZSP_FEAT_BF16_TILED 4 /* bf16 in the evaluator's tiled layout, 17 planes padded to 32 channels (azsp_stem_tiled) */
#define AZSP_FEAT_F16_TILED 5  /* the same layout with f16 elements (azsp_stem_tiled_f16) */
#define AZSP_FEAT_F16_SPLIT 6  /* the input of the fp32-class evaluator's stem (azsp_stem_split): split layout [row][plane: hi, lo][4 chunks]
                                  [N*N][8] f16 with 32 channels, azsp_split_bytes(rows, N, 32) bytes; the planes are 0 / 1 = exact f16 values,
                                  so only the hi plane is ever written (the caller zero-initialises the tensor once) */

#define AZSP_OK 0
#define AZSP_EINVAL (-1)   /* bad argument / unsupported configuration */
#define AZSP_ENOMEM (-2)
#define AZSP_EDEVICE (-3)  /* HIP runtime error */
#define AZSP_EENGINE (-4)  /* engine-side fault flag raised (node pool, depth, staging), see azsp_last_error */

/* Game status codes reported by azsp_get_status */
#define AZSP_ST_NEED_ROOT 0
#define AZSP_ST_SEARCH 1
#define AZSP_ST_MOVE_DONE 2
#define AZSP_ST_IDLE 3
#define AZSP_ST_WAIT_BUF 4

typedef struct AzspConfig {
    int32_t game;              /* AZSP_GAME_* */
    int32_t board_size;        /* Go: 5, 9, 13, 19   Gomoku: 7, 9, 13, 15 */
    int32_t num_games;         /* G: concurrent games (one wavefront each) */
    int32_t num_parallel;      /* P: leaves per game per round; 1 selects uct_search semantics (pipeline.py:132-156) */
    int32_t num_simulations;   /* budget: root.N < sims (+P when P > 1)      (mcts_v2.py:378, :568) */
    int32_t max_nodes;         /* node pool per game; 0 = sims + 3P + 8 */
    int32_t root_noise;        /* mcts_v2.py:375 */
    int32_t deterministic;     /* mcts_v2.py:427-429 */
    int32_t reuse_tree;        /* 0: root_node=None on every move (pipeline.py:836) */
    int32_t warm_up_steps;     /* pipeline.py:320 */
    int32_t has_resign;        /* Go only */
    int32_t check_resign_after_steps;
    int32_t force_resign_disabled; /* -1: draw per game with disable_resign_ratio (pipeline.py:244-246); 0/1: fixed */
    int32_t inject_random;     /* 1: noise / uniforms come from azsp_set_injection (parity runs) */
    int32_t inject_moves;      /* plies covered by the injection tables */
    int32_t stop_after_move;   /* 1: drop-in uct_search mode, the caller picks and commits the move */
    int32_t max_plies;         /* >0: a game idles after this many moves (tests) */
    int32_t stop_at_game_end;  /* 1: slots idle after their first game (tests) */
    int32_t feature_dtype;     /* AZSP_FEAT_* of the tensor written by azsp_select */
    int32_t log_moves;         /* keep per-move pi / child_N / Q logs (tests, drop-in mode) */
    int32_t log_capacity;      /* plies per game kept in the log */
    int32_t max_steps;         /* Go: 0 = 2*N*N (go.py:48) */
    int32_t num_to_win;        /* Gomoku (gomoku.py:26) */
    int32_t training_steps;    /* tag copied into every finished game (pipeline.py:271) */
    int32_t rank;              /* added to seed, mirrors set_seed(seed + rank) (pipeline.py:193) */
    int32_t device;            /* HIP device ordinal */
    float c_puct_base;         /* used by the caller to build the pb_c table (azsp_set_tables) */
    float c_puct_init;
    float disable_resign_ratio;
    float reserved0;
    double dirichlet_eps;      /* 0.25 (mcts_v2.py:235) */
    double dirichlet_alpha;    /* 0.03 */
    double resign_threshold;   /* <= -1 disables resignation (pipeline.py:216) */
    double komi;               /* 7.5 (go.py:46) */
    uint64_t seed;
} AzspConfig;

/* Sizes the caller needs to allocate its tensors. */
typedef struct AzspGeometry {
    int32_t num_actions;       /* A = N*N (+1 pass for Go) */
    int32_t num_points;        /* N*N */
    int32_t planes;            /* 17 */
    int32_t batch_rows;        /* G*P rows of features / priors / values */
    int32_t max_nodes;
    int32_t budget;
    int32_t table_len;         /* entries expected by azsp_set_tables */
    int32_t stage_capacity;    /* max samples per game */
    int64_t device_bytes;      /* HBM allocated by the engine */
    int32_t node_record_bytes;
    int32_t reserved;
} AzspGeometry;

int azsp_create(const AzspConfig* cfg, void** out_engine);
int azsp_destroy(void* engine);
const char* azsp_last_error(void* engine);
int azsp_geometry(void* engine, AzspGeometry* out);

/* pb_c(n) = log((1 + n + base)/base) + init for n = 0..len-1, as the reference evaluates it for a
 * node whose visit count is an np.float32 (pbc_np) or a Python float (fresh root, pbc_py), and
 * float32(sqrt(n)) (mcts_v2.py:99-102).  Host pointers; copied. */
int azsp_set_tables(void* engine, const double* pbc_np_host, const double* pbc_py_host, const float* sqrt32_host, int32_t len);

/* Injected randomness for parity runs: noise[G][moves][A] (np.random.dirichlet outputs) and
 * uniforms[G][moves][16] (the u of each np.random.choice draw).  Host pointers; copied. */
int azsp_set_injection(void* engine, const double* noise_host, const double* uniforms_host, int32_t moves);

/* Start a new game in every slot (env.reset()). */
int azsp_reset_games(void* engine, void* stream);

/* Standalone environment kernels.  actions_dev[G]: action, -1 = resign (Go), -2 = no-op (export only).
 * Outputs (each may be NULL): board int8[G][N*N] (reference colour ids), legal int8[G][A],
 * scalars int32[G][12] = {ko, caps_black, caps_white, steps, to_play, done, reward, winner, area_black,
 * area_white, illegal, last_was_pass}, obs int8[G][17][N][N]. */
int azsp_env_step(void* engine, const int32_t* actions_dev, int8_t* board_dev, int8_t* legal_dev, int32_t* scalars_dev,
                  int8_t* obs_dev, void* stream);

/* Load an arbitrary position into slot `slot` and make it the (fresh, unevaluated) search root.
 * board_host int8[N*N] and hist_host int8[8][N*N] (newest first, hist[0] == board) use reference colour ids. */
int azsp_set_state(void* engine, int32_t slot, const int8_t* board_host, const int8_t* hist_host, int32_t to_play,
                   int32_t steps, int32_t ko, int32_t last_was_pass, int32_t caps_black, int32_t caps_white, void* stream);

/* Drop-in mode: hand the Dirichlet draw of this search to the engine (noise_host double[G][A] or NULL) and the
 * `warm_up` flag of uct_search (1: temperature 1.0, 0: temperature 0.1, -1: derive from env.steps <= warm_up_steps). */
int azsp_begin_move(void* engine, const double* noise_host, int32_t warm_up, void* stream);

int azsp_select(void* engine, void* features_dev, uint8_t* valid_dev, void* stream);
int azsp_expand_backup(void* engine, const float* priors_dev, const float* values_dev, void* stream);
/* The same two phases for the games [g0, g1) only (g0 % 32 == 0; the tensors are the full-batch ones, indexed by game as above).
 * Games never interact during a search (mcts_v2.py:568-625 runs per game), so disjoint ranges in any order are the whole batch,
 * bit for bit (tested).  EXPERIMENTAL: no product path calls these two entries -- SelfPlayActor runs whole-batch rounds on ONE
 * stream.  They exist for the half-batch overlap experiment (tools/overlap_actor.py; DESIGN "Two streams": measured +0.3 %, not
 * adopted).  The ENGINE kernels of disjoint ranges are exact on two streams; two evaluator FORWARDS in flight at the same time
 * are a separate matter, see DESIGN before putting two forwards on one device. */
int azsp_select_range(void* engine, void* features_dev, uint8_t* valid_dev, int32_t g0, int32_t g1, void* stream);
int azsp_expand_backup_range(void* engine, const float* priors_dev, const float* values_dev, int32_t g0, int32_t g1, void* stream);
int azsp_round(void* engine, const float* priors_dev, const float* values_dev, void* features_dev, uint8_t* valid_dev,
               void* stream);

/* status_host int32[G][8] = {status, ply, root_N, n_leaves, last_move, games_done, root_eval_pending, noise_pending};
 * q_host double[G][2] = {root_Q, best_child_Q} of the last finished search.  Synchronises the stream. */
int azsp_get_status(void* engine, int32_t* status_host, double* q_host, void* stream);

/* Search outputs of `slot` at log index `ply` (log_moves or drop-in mode): pi double[A], child_N float[A],
 * q double[4] = {root_Q, best_child_Q, root_N, move}.  Synchronises the stream. */
int azsp_get_search(void* engine, int32_t slot, int32_t ply, double* pi_host, float* child_n_host, double* q_host, void* stream);

/* Drop-in mode: the caller's chosen moves (int32[G], host); re-roots each tree on the chosen child. */
int azsp_commit_move(void* engine, const int32_t* moves_host, void* stream);

/* Drop-in mode: one iteration of the simulation loop of uct_search / parallel_uct_search (core/mcts_v2.py:378-421, :568-625) around the
 * caller's eval_func in ONE kernel launch and ONE stream synchronisation (the separate entries above cost four launches and eight
 * synchronisations).  priors_host float[rows][A] / values_host float[rows] (rows = G * P) = eval_func's outputs for the leaves of the
 * previous call; both NULL on the first call of a search (nothing to back up yet).  The game's wave runs expand / backup (and the
 * end-of-search work when the budget is met), selects the next leaves into features_dev / valid_dev, and writes status_host int32[G][8],
 * q_host double[G][2] (as azsp_get_status; q_host may be NULL), valid_host uint8[rows] and the first features_bytes bytes of
 * features_dev (the observation planes eval_func receives; 0 = none; feature_dtype must be a plain [rows][17][N][N] tensor: I8 / F32 /
 * BF16 / F16) -- through a page-locked staging buffer of the engine that the kernel reads and writes directly, so no copy command is
 * issued.  priors_dev / values_dev are the caller's evaluator tensors (unused by this entry beyond validation; azsp_expand_backup
 * reads them).  Host pointers may be pageable. */
int azsp_dropin_step(void* engine, const float* priors_host, const float* values_host, float* priors_dev, float* values_dev,
                     void* features_dev, uint8_t* valid_dev, int32_t* status_host, double* q_host, uint8_t* valid_host, void* features_host,
                     int64_t features_bytes, void* stream);

/* Collect finished games.  states int8[cap][17][N][N], pi float[cap][A], z float[cap] receive the samples of
 * whole games back to back; games_host int32[max_games][16] = {start, length, winner(ref id, 0 none), area_black,
 * area_white, num_passes, resigned, resign_disabled, marked_for_resign, could_won, marked_player(ref id, 0 none),
 * uid, training_steps, reward, last_player(ref id), slot}.  Synchronises the stream. */
int azsp_harvest(void* engine, int8_t* states_dev, float* pi_dev, float* z_dev, int32_t sample_capacity,
                 int32_t* games_host, int32_t max_games, int32_t* n_samples_out, int32_t* n_games_out, void* stream);

/* Diagnostics: the production random streams (no injection) of every game slot for plies 0 .. plies-1 of its current game,
 * drawn exactly as the search draws them and without changing any state: noise_host double[G][plies][A] = the Dirichlet(alpha)
 * vectors add_dirichlet_noise uses (core/mcts_v2.py:259-260: one draw over all A actions), unif_host double[G][plies][tries] =
 * the uniforms behind np.random.choice (core/mcts_v2.py:434; try t is consumed only if try t-1 was rejected).  Counter-based
 * Philox4x32-10 keyed by (seed + rank, slot, game uid, ply, action / try): statistical tests in tests/ use this entry. */
int azsp_rng_probe(void* engine, int32_t plies, int32_t tries, double* noise_host, double* unif_host, void* stream);

/* Optional second output of azsp_harvest: moves_dev int16[sample_capacity] receives, for every harvested sample, the move
 * that was played from its position (a flat action index, board_size^2 = pass, -1 = the mover resigned) -- what the
 * reference's env.history / to_sgf() holds for a self-play game (envs/base.py:224-226, core/pipeline.py:276-281).
 * NULL (the default) disables it.  The buffer must stay valid for every later azsp_harvest call. */
int azsp_harvest_moves(void* engine, int16_t* moves_dev);

/* Third optional output of azsp_harvest: extra_host int32[max_games][4] (host memory, row i belongs to games_host row i) =
 * {training_steps of the weights in use when the game ENDED, the game's own resign threshold as the low / high word of its
 * double, 1 if the game straddled a weight hot-swap (end tag != start tag)}.  games_host column 12 is the tag of the weights that
 * STARTED the game, which is what the reference actor writes (core/pipeline.py:237 -> :271).  NULL (default) disables it. */
int azsp_harvest_extra(void* engine, int32_t* extra_host);

/* Per-game actor state: the reference actor re-reads the shared resign threshold (core/pipeline.py:241-242) and the weights'
 * training_steps (core/pipeline.py:232-239) before EVERY game.  Both values take effect for games that start after this call
 * (each game keeps the threshold / tag it started with; resign_disabled is drawn per game only while the threshold is > -1,
 * core/pipeline.py:244-246).  resign_threshold <= -1 disables resignation (the learner's warm-up value, core/pipeline.py:449-459). */
int azsp_set_actor_state(void* engine, double resign_threshold, int32_t training_steps);

/* counters_host uint64[16]: simulations, best_child calls, backup edges, leaves, duplicate leaves, terminal hits,
 * moves, games, root evaluations, nodes created, game-rounds, buffer stalls, speculative record prefetches of the select phase and
 * how many of them were used. */
int azsp_counters(void* engine, uint64_t* counters_host, int32_t reset, void* stream);

/* Dihedral-8 transform of a batch (no engine needed): op 0 identity, 1 h-flip, 2 v-flip, 3/4/5 = rot90/180/270
 * counter-clockwise (the reference's five), 6 transpose, 7 anti-transpose.  elem_size in bytes (1, 2, 4 or 8).
 * states [B][C][N][N], pi [B][A] with A == N*N or N*N+1 (pass column untouched). */
int azsp_dihedral(const void* states_in_dev, void* states_out_dev, int32_t state_elem_size, const void* pi_in_dev,
                  void* pi_out_dev, int32_t pi_elem_size, int32_t batch, int32_t channels, int32_t board_size,
                  int32_t num_actions, int32_t op, void* stream);

/* Fused convolution epilogue for the leaf evaluator (core/network.py ResNetBlock, network.py:42-82 in eval mode with
 * BatchNorm folded): y[r][c] = act(y[r][c] + bias[c] (+ residual[r][c])) in place, one pass over HBM instead of the
 * separate bias / residual-add / ReLU kernels.  y, residual: [rows][channels] (channels-last activations), dtype
 * AZSP_FEAT_F32 / _BF16 / _F16, channels % 8 == 0; accumulation in fp32, one rounding. */
int azsp_bias_act(void* y_dev, const void* bias_dev, const void* residual_dev, int64_t rows, int32_t channels, int32_t dtype,
                  int32_t relu, void* stream);

/* The residual tower's resident activation layout ("tiled"): [tile = T boards][C/8 channel chunks][T*S*S positions][8 ch]
 * bf16 with T = max(1, 256 / (S*S)) boards per tile (3 at 9x9, 1 from 13x13 up), azsp_tiled_bytes(boards, S, C) bytes.  A tile is what one workgroup of azsp_conv3x3_tiled multiplies at a time:
 * its LDS image equals its global image (flat LDS-DMA copy), fragment reads are conflict-free with immediate k offsets,
 * and the epilogue stores 512 contiguous bytes per instruction.  azsp_tile_layout converts channels-last rows
 * [boards][S][S][C] to (to_tiled = 1) / from (0) that layout at tower entry / exit.  Positions of boards past `boards`
 * in the last tile are never read into a valid output. */
int64_t azsp_tiled_bytes(int64_t boards, int32_t board_size, int32_t channels);
int azsp_tile_layout(const void* src_dev, void* dst_dev, int64_t boards, int32_t board_size, int32_t channels, int32_t to_tiled,
                     void* stream);
/* Fused 3x3 convolution of the residual tower (core/network.py:42-82, eval mode, BatchNorm folded) on the tiled layout:
 * y = act(conv3x3(x, w) + bias [+ residual]); x, residual, y tiled bf16, w_packed [9 taps (ky*3+kx)][C out][C in] bf16, bias
 * float[C].  Weight-stationary MFMA kernels, the filter bank stays in the registers of persistent workgroups.  On the device:
 * (S, C) = (9, 128) 9x9 Go, (17, 64) the 13x13 Gomoku tower, (9, 64) 9x9 Go with 64 filters, (19, 256) the jumbo Go tower (two launches per convolution, the
 * partial sum lives in y: x and residual must not alias y); AZSP_EINVAL for other shapes. */
int azsp_conv3x3_tiled(const void* x_dev, const void* w_packed_dev, const float* bias_dev, const void* residual_dev, void* y_dev,
                       int64_t boards, int32_t board_size, int32_t channels, int32_t relu, void* stream);

/* One whole ResNetBlock (core/network.py:42-82: conv3x3 + BN + ReLU + conv3x3 + BN, + skip, ReLU; eval mode, BatchNorm folded) of a
 * 64-filter tower in ONE launch on the tiled layout: y = relu(conv3x3(relu(conv3x3(x, w1) + b1), w2) + b2 + x).  The intermediate
 * activation stays in LDS and the skip is taken from the input tile already resident there, so a block moves two tensor passes
 * through HBM instead of five; results are bit-identical to two azsp_conv3x3_tiled calls (same MFMA order, same two bf16 roundings).
 * w1 / w2 packed [9 taps][64 out][64 in] bf16, b1 / b2 float[64]; y may alias x.  On the device: (S, C) = (17, 64) the 13x13 Gomoku
 * tower, (9, 64) the 9x9 Go tower with 64 filters (logs/go/9x9_12b64); AZSP_EINVAL for other shapes. */
int azsp_resblock_tiled(const void* x_dev, const void* w1_packed_dev, const float* bias1_dev, const void* w2_packed_dev, const float* bias2_dev,
                        void* y_dev, int64_t boards, int32_t board_size, int32_t channels, void* stream);

/* The residual tower at the reference's precision class (core/network.py:42-82 evaluated in fp32 by core/pipeline.py:91-123).
 * gfx950 multiplies fp32 matrices at 1/16 of its f16 rate and has no TF32, so here an fp32 value v travels as two f16 numbers,
 * hi = f16(v) and lo = f16((v - hi) * 2048), and a product is three f16 MFMAs (w_hi x_hi; w_hi x_lo + w_lo x_hi scaled by 1/2048)
 * accumulated in fp32: 22-bit significands, per-product error <= 3 * 2^-22, i.e. fp32 round-off class (bounded against fp64 next to
 * the library's fp32 convolution in tests/test_split_tower.py).  RANGE: a value beyond f16's finite range (|v| > 65504) cannot be
 * split; it is clamped to +-65504 where the reference's fp32 network would carry it on.  That never happens silently: every kernel
 * that splits values records such an event in a sticky device-side RANGE RECORD (below; the convolution epilogues look at the value
 * IN FRONT of the ReLU -- a pre-activation below -65504 counts although the ReLU zeroes it: deliberately conservative, it costs no
 * instruction and a tower whose negative pre-activations leave the range is about to lose its positive ones; BatchNorm-folded AlphaZero towers stay far
 * inside the range -- activations of the shipped networks peak at ~1e2 -- and alpha_zero_amd.core.network.InferenceNet rescales a
 * network's activations by an exact power of two when a calibration pass finds them near the limit).
 * "Split layout": [board][plane: hi, lo][C/8 channel chunks][S*S positions][8 ch] f16 = azsp_split_bytes(boards, S, C) bytes;
 * azsp_split_layout converts fp32 channels-last rows [boards][S][S][C] to (to_split = 1) / from (0) it.
 * azsp_conv3x3_split: y = act(conv3x3(x, w) + bias [+ residual]); x, residual, y in the split layout; x must not alias y; residual
 * may alias y at 9x9 and must NOT at 17x17 (AZSP_EINVAL: the half-board tiles repeat 15 positions per board in later column tiles);
 * w_split [2 planes: hi, lo][9 taps (ky*3+kx)][C out][C in] f16 with lo = (w - hi) * 2048, bias float[C].  On the device:
 * weight-stationary kernels for (S, C) = (9, 128), (9, 64) and (17, 64) (the 13x13 Gomoku tower behind its pad-3 stem,
 * network.py:101-105; half-board tiles, az_conv_sp17.h); since round 6 every other plane size 3 <= S <= 64 with C = 64, 128 or 256
 * on the wave-per-tile kernel (az_conv_spg.h: e.g. the 19x19 x 256 jumbo tower at the reference's own precision,
 * alpha_zero/training_go_jumbo.py:46-47) -- and the three shapes above too when the call is small (azsp_small_batch_waves below;
 * bit-identical results).  AZSP_EINVAL for other shapes.
 * range_rec_dev: the caller's range record -- two zero-initialised uint32 words in device memory, owned by the caller (one per
 * network: two evaluators in one process never see each other's events), read with azsp_split_range_read; NULL selects the library's
 * per-device default record (azsp_split_range_status). */
int64_t azsp_split_bytes(int64_t boards, int32_t board_size, int32_t channels);
int azsp_split_layout(const void* src_dev, void* dst_dev, int64_t boards, int32_t board_size, int32_t channels, int32_t to_split,
                      uint32_t* range_rec_dev, void* stream);
int azsp_conv3x3_split(const void* x_dev, const void* w_split_dev, const float* bias_dev, const void* residual_dev, void* y_dev,
                       int64_t boards, int32_t board_size, int32_t channels, int32_t relu, uint32_t* range_rec_dev, void* stream);

/* One whole ResNetBlock (core/network.py:42-82: conv3x3 + BN + ReLU + conv3x3 + BN, + skip, ReLU; eval mode, BatchNorm folded) at the
 * reference's precision class in ONE launch on the split layout: y = relu(conv3x3(relu(conv3x3(x, w1) + b1), w2) + b2 + x).  The
 * intermediate activation stays in LDS (half-board tiles, the halo row of each half recomputed), so a block moves two tensor passes
 * through HBM instead of five; results are bit-identical to two azsp_conv3x3_split calls (same MFMA order, same roundings).  w1 / w2 in
 * the packing of azsp_conv3x3_split, b1 / b2 float[64]; y must NOT alias x.  On the device: (S, C) = (17, 64), the 13x13 Gomoku tower
 * (half-board tiles, az_resblock_sp17.h), and (9, 64), the reference's 64-filter 9x9 Go towers (logs/go/9x9_12b64; two boards per tile, an odd
 * last board runs as the two unfused launches); AZSP_EINVAL for other shapes.  range_rec_dev as above. */
int azsp_resblock_split(const void* x_dev, const void* w1_split_dev, const float* bias1_dev, const void* w2_split_dev, const float* bias2_dev,
                        void* y_dev, int64_t boards, int32_t board_size, int32_t channels, uint32_t* range_rec_dev, void* stream);

/* The rest of the fp32-class evaluator on the split layout (core/network.py:101-156):
 * azsp_split_features: observation planes [boards][in_channels <= 32][S][S] fp32 (the engine's AZSP_FEAT_F32 features) -> split
 *   layout with 32 channels (the missing ones zero), azsp_split_bytes(boards, S, 32) bytes.  Not needed when the engine writes its
 *   features with feature_dtype = AZSP_FEAT_F16_SPLIT: that tensor IS the stem's input (what SelfPlayActor does).
 * azsp_stem_split: the stem convolution + BatchNorm + ReLU (network.py:101-110): x from azsp_split_features (board_size x
 *   board_size), w_split [2 planes][9 taps][C out][32 in] f16 (input channels >= the network's zero), y in the tower's split layout
 *   with planes of board_size + 2 (pad - 1): pad = 1 for Go, pad = 3 for Gomoku (network.py:101-105).  On the device: (board 9,
 *   C 128 or 64, pad 1) and (board 13, C 64, pad 3).
 * azsp_head_split: both heads in fp32 in one pass over the tower output (network.py:118-156): the two 1x1 convolutions + BatchNorm +
 *   ReLU (head_w [3][C], head_b [3]: npol policy planes first), policy Linear + softmax over all A actions (pol_fc_wt = the Linear's
 *   weight TRANSPOSED, [npol*S*S][A], inputs in nn.Flatten order), value Linear + ReLU + Linear + tanh (val_fc1_wt transposed
 *   [(3-npol)*S*S][F], val_fc2_w [F], val_fc2_b); priors float [boards][A], values float [boards].  Any (S, C) with C % 8 == 0. */
int azsp_split_features(const float* planes_dev, void* dst_dev, int64_t boards, int32_t board_size, int32_t in_channels,
                        uint32_t* range_rec_dev, void* stream);
int azsp_stem_split(const void* x_split32_dev, const void* w_split_dev, const float* bias_dev, void* y_dev, int64_t boards, int32_t board_size,
                    int32_t channels, int32_t pad, int32_t relu, uint32_t* range_rec_dev, void* stream);
/* azsp_stem_split for inputs whose lo plane is all zero -- values that are exact in f16, i.e. the engine's AZSP_FEAT_F16_SPLIT features (0 / 1
 * observation planes): the lo plane is neither loaded nor multiplied (its product is exactly zero), the result is identical to azsp_stem_split's. */
int azsp_stem_split_exact(const void* x_split32_dev, const void* w_split_dev, const float* bias_dev, void* y_dev, int64_t boards, int32_t board_size,
                          int32_t channels, int32_t pad, int32_t relu, uint32_t* range_rec_dev, void* stream);
int azsp_head_split(const void* x_dev, const float* head_w_dev, const float* head_b_dev, const float* pol_fc_wt_dev, const float* pol_fc_b_dev,
                    const float* val_fc1_wt_dev, const float* val_fc1_b_dev, const float* val_fc2_w_dev, float val_fc2_b, float* priors_dev,
                    float* values_dev, int64_t boards, int32_t board_size, int32_t channels, int32_t num_actions, int32_t fc_units, int32_t npol,
                    void* stream);

/* Range record of the split-precision evaluator (sticky): *events_host = how many kernel lanes have met a value with |v| > 65504 (or
 * a NaN among the fp32 inputs of azsp_split_layout / azsp_split_features, recorded as +inf) since the last reset -- each such value was
 * clamped; the reference would have carried it: core/pipeline.py:91-123 evaluates in fp32 -- and *max_abs_host = the largest such |v|
 * (0 if none).  Either pointer may be NULL; reset != 0 clears the record.  Synchronises `stream`.
 * azsp_split_range_read reads the record the caller passed to the kernels as range_rec_dev (NULL = the per-device default record);
 * azsp_split_range_status reads the default record.  alpha_zero_amd.core.network.InferenceNet owns one record per network;
 * SelfPlayActor polls it at every harvest and, on an event, rescales the network's activations by a power of two (exact) or falls
 * back to the library's fp32 convolutions -- see InferenceNet.calibrate_activation_scale. */
int azsp_split_range_read(const uint32_t* range_rec_dev, uint32_t* events_host, float* max_abs_host, int32_t reset, void* stream);
int azsp_split_range_status(uint32_t* events_host, float* max_abs_host, int32_t reset, void* stream);

/* Small batches (round 6).  The weight-stationary kernels behind azsp_conv3x3_split / azsp_resblock_split give every board to ONE
 * workgroup (built for tens of thousands of boards per launch: 10 - 33 us however few boards there are).  A call whose output is at most
 * `waves` tiles of 16 couts x 32 positions (one wave each; a 17x17 x 64 board is 40 of them) runs the wave-per-tile kernel k_conv3x3_spg
 * instead (csrc/az_conv_spg.h: a board is spread over the chip; BIT-IDENTICAL results, so an evaluation does not depend on the batch it
 * arrives in) -- the batch-1 forwards of a drop-in uct_search (alpha_zero/core/mcts_v2.py:301-450 evaluates one leaf at a time).
 * Sets the process-wide limit and returns the previous one; a negative argument only queries.  Default 1024 (one wave per SIMD of an
 * MI355X: the measured crossover, profiles/r06_spg_ab.txt), or the environment variable AZSP_SPG_MAX_WAVES; 0 = always the
 * weight-stationary kernels.  Shapes without a weight-stationary kernel (plane sizes 3 .. 64, 64 / 128 / 256 filters) always run
 * k_conv3x3_spg (beyond `waves`: k_conv3x3_spgw, 48-position tiles whose activations a workgroup shares through LDS). */
int64_t azsp_small_batch_waves(int64_t waves);

/* Replay sampling on the device (core/replay.py:72-83 UniformReplay.sample + core/pipeline.py:636-643: the batch
 * tensors and apply_random_transformation): out_states[b] = T_op(ring_states[idx[b]]) cast to state_dtype (AZSP_FEAT_I8 / F32 /
 * BF16 / F16), out_pi[b] = T_op(ring_pi[idx[b]]), out_z[b] = ring_z[idx[b]].  The ring is what azsp_harvest fills:
 * states int8 [capacity][channels][N][N], pi float [capacity][A], z float [capacity]; idx int64 [batch] (device); op as in
 * azsp_dihedral, one per batch like the reference.  One pass over HBM, no host round trip. */
int azsp_replay_gather(const int8_t* ring_states_dev, const float* ring_pi_dev, const float* ring_z_dev, const int64_t* idx_dev, int32_t batch,
                       int32_t channels, int32_t board_size, int32_t num_actions, int32_t op, int32_t state_dtype, void* out_states_dev,
                       float* out_pi_dev, float* out_z_dev, void* stream);

/* Stem of the evaluator (core/network.py:98-108 conv_block: conv3x3 17 -> C + BatchNorm + ReLU) on the tiled layout: the
 * input is the feature tensor azsp_select writes with feature_dtype = AZSP_FEAT_BF16_TILED (17 planes zero-padded to 32
 * channels: [tile][4][3*S*S][8] bf16, azsp_tiled_bytes(rows, S, 32) bytes); w_packed is [9 taps][C out][32 in] bf16 (input
 * channels 17..31 zero), the output is the tower's tiled layout with planes of board_size + 2 (pad - 1): pad = 1 for Go, pad = 3
 * for Gomoku (core/network.py:101-105: 13x13 boards become 17x17 planes).  Same kernels as azsp_conv3x3_tiled with 4 input chunks;
 * on the device: (board 9, 128 filters, pad 1), (board 9, 64 filters, pad 1), (board 13, 64 filters, pad 3) and (board 19, 256 filters, pad 1). */
int azsp_stem_tiled(const void* features_tiled_dev, const void* w_packed_dev, const float* bias_dev, void* y_tiled_dev, int64_t boards,
                    int32_t board_size, int32_t channels, int32_t pad, int32_t relu, void* stream);
/* Both 1x1 head convolutions (core/network.py:131-156: conv1x1 + BatchNorm + ReLU of the policy and the value head) in one
 * pass over the tiled tower output: w [policy_planes + value_planes][C] fp32 (BatchNorm folded), bias fp32;
 * pol_out [boards][policy_planes][S*S], val_out [boards][value_planes][S*S] bf16 (plane-major = nn.Flatten order); rows are
 * pol_stride / val_stride elements apart (0 = dense; azsp_fc_heads wants multiples of 16 with zero padding). */
int azsp_head_tiled(const void* x_tiled_dev, const float* w_dev, const float* bias_dev, void* pol_out_dev, void* val_out_dev, int64_t boards,
                    int32_t board_size, int32_t channels, int32_t policy_planes, int32_t value_planes, int32_t pol_stride, int32_t val_stride,
                    void* stream);
/* Fully connected layers of both heads with softmax / tanh (core/network.py:136-156 Linear layers, core/pipeline.py:105 softmax):
 * priors[b] = softmax(Wp pol[b] + bp) over A actions (fp32), values[b] = tanh(W2 relu(W1 val[b] + b1) + b2).  pol / val: bf16 rows
 * of k1_steps * 16 / k2_steps * 16 elements (zero padded), wp [ceil32(A)][k1_steps * 16] and w1 [ceil32(F)][k2_steps * 16] bf16
 * zero padded, bp / b1 / w2 fp32 padded to ceil32.  MFMA GEMMs, one wave per 32 boards; A <= 192, F <= 128 on the device. */
int azsp_fc_heads(const void* pol_dev, const void* val_dev, const void* wp_dev, const float* bp_dev, int32_t k1_steps, const void* w1_dev,
                  const float* b1_dev, int32_t k2_steps, const float* w2_dev, float b2, float* priors_dev, float* values_dev, int64_t boards,
                  int32_t num_actions, int32_t fc_width, void* stream);

/* f16 variants of the bf16 evaluator entries above: identical layouts, arguments and kernels, with f16 activations / weights / head
 * planes (the f16 MFMA runs at the bf16 rate and carries three more significand bits; outputs are clamped to f16's finite range).
 * Features: feature_dtype = AZSP_FEAT_F16_TILED.  On the device: the 9x9 x 128 evaluator ((S, C) = (9, 128), pad 1, 82 actions,
 * 128 fully connected units); AZSP_EINVAL for other shapes. */
int azsp_conv3x3_tiled_f16(const void* x_dev, const void* w_packed_dev, const float* bias_dev, const void* residual_dev, void* y_dev,
                           int64_t boards, int32_t board_size, int32_t channels, int32_t relu, void* stream);
int azsp_stem_tiled_f16(const void* features_tiled_dev, const void* w_packed_dev, const float* bias_dev, void* y_tiled_dev, int64_t boards,
                        int32_t board_size, int32_t channels, int32_t pad, int32_t relu, void* stream);
int azsp_head_tiled_f16(const void* x_tiled_dev, const float* w_dev, const float* bias_dev, void* pol_out_dev, void* val_out_dev, int64_t boards,
                        int32_t board_size, int32_t channels, int32_t policy_planes, int32_t value_planes, int32_t pol_stride,
                        int32_t val_stride, void* stream);
int azsp_fc_heads_f16(const void* pol_dev, const void* val_dev, const void* wp_dev, const float* bp_dev, int32_t k1_steps, const void* w1_dev,
                      const float* b1_dev, int32_t k2_steps, const float* w2_dev, float b2, float* priors_dev, float* values_dev, int64_t boards,
                      int32_t num_actions, int32_t fc_width, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AZSP_H */
