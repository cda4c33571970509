"""ctypes view of include/azsp.h (the C ABI of libazsp.so).  Library-agnostic: `Binding` wraps any
CDLL exporting the azsp_* symbols.  The product obtains its Binding from alpha_zero_amd._lib.load(),
which only ever loads the HIP build and refuses to run without a GPU."""
import ctypes as C

GAME_GO, GAME_GOMOKU = 0, 1
FEAT_I8, FEAT_F32, FEAT_BF16, FEAT_F16, FEAT_BF16_TILED, FEAT_F16_TILED, FEAT_F16_SPLIT = 0, 1, 2, 3, 4, 5, 6
ST_NEED_ROOT, ST_SEARCH, ST_MOVE_DONE, ST_IDLE, ST_WAIT_BUF = 0, 1, 2, 3, 4

SYMBOLS = [
    "azsp_create", "azsp_destroy", "azsp_last_error", "azsp_geometry", "azsp_set_tables", "azsp_set_injection",
    "azsp_reset_games", "azsp_env_step", "azsp_set_state", "azsp_begin_move", "azsp_select", "azsp_expand_backup",
    "azsp_round", "azsp_get_status", "azsp_get_search", "azsp_commit_move", "azsp_harvest", "azsp_counters", "azsp_dihedral", "azsp_bias_act",
    "azsp_conv3x3_tiled", "azsp_tile_layout", "azsp_tiled_bytes", "azsp_stem_tiled", "azsp_head_tiled", "azsp_replay_gather", "azsp_rng_probe", "azsp_harvest_moves", "azsp_fc_heads", "azsp_harvest_extra", "azsp_set_actor_state", "azsp_resblock_tiled", "azsp_select_range", "azsp_expand_backup_range", "azsp_split_bytes", "azsp_split_layout", "azsp_conv3x3_split", "azsp_split_features", "azsp_stem_split", "azsp_head_split", "azsp_conv3x3_tiled_f16", "azsp_stem_tiled_f16", "azsp_head_tiled_f16", "azsp_fc_heads_f16", "azsp_split_range_status", "azsp_stem_split_exact", "azsp_split_range_read", "azsp_resblock_split", "azsp_dropin_step", "azsp_small_batch_waves",
]

COUNTER_NAMES = ["sims", "node_visits", "backup_edges", "leaves", "dup_leaves", "terminal_hits", "moves", "games", "root_evals",
                 "nodes_created", "game_rounds", "stalls", "hint_prefetches", "hint_hits"]


class AzspConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "game", "board_size", "num_games", "num_parallel", "num_simulations", "max_nodes", "root_noise", "deterministic",
        "reuse_tree", "warm_up_steps", "has_resign", "check_resign_after_steps", "force_resign_disabled", "inject_random",
        "inject_moves", "stop_after_move", "max_plies", "stop_at_game_end", "feature_dtype", "log_moves", "log_capacity",
        "max_steps", "num_to_win", "training_steps", "rank", "device")] + [
        ("c_puct_base", C.c_float), ("c_puct_init", C.c_float), ("disable_resign_ratio", C.c_float), ("reserved0", C.c_float),
        ("dirichlet_eps", C.c_double), ("dirichlet_alpha", C.c_double), ("resign_threshold", C.c_double), ("komi", C.c_double),
        ("seed", C.c_uint64)]


class AzspGeometry(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_actions", "num_points", "planes", "batch_rows", "max_nodes", "budget", "table_len",
                                         "stage_capacity")] + [("device_bytes", C.c_int64), ("node_record_bytes", C.c_int32),
                                                               ("reserved", C.c_int32)]


class AzspError(RuntimeError):
    pass


class Binding:
    def __init__(self, cdll, name="libazsp"):
        self.dll, self.name = cdll, name
        missing = [s for s in SYMBOLS if not hasattr(cdll, s)]
        if missing:
            raise AzspError(f"{name} does not export {missing}")
        V, I, P = C.c_void_p, C.c_int32, C.POINTER
        sig = {
            "azsp_create": [P(AzspConfig), P(V)], "azsp_destroy": [V], "azsp_geometry": [V, P(AzspGeometry)],
            "azsp_set_tables": [V, V, V, V, I], "azsp_set_injection": [V, V, V, I], "azsp_reset_games": [V, V],
            "azsp_env_step": [V, V, V, V, V, V, V], "azsp_set_state": [V, I, V, V, I, I, I, I, I, I, V],
            "azsp_begin_move": [V, V, I, V], "azsp_select": [V, V, V, V], "azsp_expand_backup": [V, V, V, V],
            "azsp_round": [V, V, V, V, V, V], "azsp_select_range": [V, V, V, I, I, V], "azsp_expand_backup_range": [V, V, V, I, I, V], "azsp_get_status": [V, V, V, V], "azsp_get_search": [V, I, I, V, V, V, V],
            "azsp_commit_move": [V, V, V], "azsp_dropin_step": [V, V, V, V, V, V, V, V, V, V, V, C.c_int64, V], "azsp_harvest": [V, V, V, V, I, V, I, P(I), P(I), V],
            "azsp_counters": [V, V, I, V], "azsp_dihedral": [V, V, I, V, V, I, I, I, I, I, I, V],
            "azsp_bias_act": [V, V, V, C.c_int64, I, I, I, V],
            "azsp_conv3x3_tiled": [V, V, V, V, V, C.c_int64, I, I, I, V], "azsp_tile_layout": [V, V, C.c_int64, I, I, I, V],
            "azsp_resblock_tiled": [V, V, V, V, V, V, C.c_int64, I, I, V],
            "azsp_conv3x3_split": [V, V, V, V, V, C.c_int64, I, I, I, V, V], "azsp_split_layout": [V, V, C.c_int64, I, I, I, V, V],
            "azsp_split_features": [V, V, C.c_int64, I, I, V, V], "azsp_stem_split": [V, V, V, V, C.c_int64, I, I, I, I, V, V],
            "azsp_split_range_status": [V, V, I, V], "azsp_stem_split_exact": [V, V, V, V, C.c_int64, I, I, I, I, V, V],
            "azsp_split_range_read": [V, V, V, I, V], "azsp_resblock_split": [V, V, V, V, V, V, C.c_int64, I, I, V, V],
            "azsp_head_split": [V, V, V, V, V, V, V, V, C.c_float, V, V, C.c_int64, I, I, I, I, I, V],
            "azsp_stem_tiled": [V, V, V, V, C.c_int64, I, I, I, I, V], "azsp_head_tiled": [V, V, V, V, V, C.c_int64, I, I, I, I, I, I, V],
            "azsp_fc_heads": [V, V, V, V, I, V, V, I, V, C.c_float, V, V, C.c_int64, I, I, V],
            "azsp_replay_gather": [V, V, V, V, I, I, I, I, I, I, V, V, V, V], "azsp_rng_probe": [V, I, I, V, V, V], "azsp_harvest_moves": [V, V], "azsp_harvest_extra": [V, V], "azsp_set_actor_state": [V, C.c_double, I],
        }
        for k in ("azsp_conv3x3_tiled", "azsp_stem_tiled", "azsp_head_tiled", "azsp_fc_heads"):  # the f16 variants take the same arguments
            sig[k + "_f16"] = sig[k]
        for k, a in sig.items():
            f = getattr(cdll, k)
            f.argtypes, f.restype = a, C.c_int
        cdll.azsp_last_error.argtypes, cdll.azsp_last_error.restype = [V], C.c_char_p
        cdll.azsp_small_batch_waves.argtypes, cdll.azsp_small_batch_waves.restype = [C.c_int64], C.c_int64
        cdll.azsp_tiled_bytes.argtypes, cdll.azsp_tiled_bytes.restype = [C.c_int64, I, I], C.c_int64
        cdll.azsp_split_bytes.argtypes, cdll.azsp_split_bytes.restype = [C.c_int64, I, I], C.c_int64

    def check(self, rc, handle=None, what=""):
        if rc != 0:
            msg = self.dll.azsp_last_error(handle).decode() if handle else ""
            raise AzspError(f"{what} failed with code {rc} {msg}")
