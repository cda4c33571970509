"""Python host of the batched self-play engine: thin object wrapper over the C ABI (include/azsp.h).

All per-game state (trees, positions, samples) lives in HBM inside the engine; this class only owns
the evaluator-facing tensors (features / valid / priors / values) and passes raw device pointers.
Nothing here computes on the CPU except the tiny pb_c / sqrt lookup tables, which are evaluated with
the very expressions of the reference (mcts_v2.py:99-102) so their rounding is identical.
"""
import ctypes as C
import math
from dataclasses import dataclass

import numpy as np
import torch

from .. import _abi
from .._abi import AzspConfig, AzspGeometry, Binding

_FEAT_TORCH = {_abi.FEAT_I8: torch.int8, _abi.FEAT_F32: torch.float32, _abi.FEAT_BF16: torch.bfloat16, _abi.FEAT_F16: torch.float16}


@dataclass
class EngineConfig:
    game: str = "go"                 # "go" | "gomoku"
    board_size: int = 9
    num_games: int = 1
    num_parallel: int = 8
    num_simulations: int = 200
    c_puct_base: float = 19652.0
    c_puct_init: float = 1.25
    root_noise: bool = True
    deterministic: bool = False
    reuse_tree: bool = True
    warm_up_steps: int = 16
    komi: float = 7.5
    max_steps: int = 0
    num_to_win: int = 5
    resign_threshold: float = -1.0
    check_resign_after_steps: int = 40
    disable_resign_ratio: float = 0.1
    force_resign_disabled: int = -1
    dirichlet_eps: float = 0.25
    dirichlet_alpha: float = 0.03
    inject_random: bool = False
    inject_moves: int = 0
    stop_after_move: bool = False
    max_plies: int = 0
    stop_at_game_end: bool = False
    feature_dtype: int = _abi.FEAT_F32
    log_moves: bool = False
    log_capacity: int = 0
    max_nodes: int = 0
    training_steps: int = 0
    seed: int = 1
    rank: int = 0
    device_index: int = 0


def pbc_tables(c_puct_base, c_puct_init, n):
    """pb_c(N) as the reference evaluates it (mcts_v2.py:101): N is an np.float32 for interior nodes and
    re-used roots, a Python float for a freshly created root; sqrt(N) is a Python double that NumPy
    narrows to float32 when it divides the float32 row (mcts_v2.py:102)."""
    t_np = np.array([math.log((1 + np.float32(i) + c_puct_base) / c_puct_base) + c_puct_init for i in range(n)], dtype=np.float64)
    t_py = np.array([math.log((1 + float(i) + c_puct_base) / c_puct_base) + c_puct_init for i in range(n)], dtype=np.float64)
    sq = np.array([np.float32(math.sqrt(np.float32(i))) for i in range(n)], dtype=np.float32)
    return t_np, t_py, sq


class Engine:
    """One engine = G concurrent games on one device.  `binding` comes from alpha_zero_amd._lib.load()."""

    def __init__(self, binding: Binding, cfg: EngineConfig, device="cuda"):
        self.b, self.cfg, self.device = binding, cfg, torch.device(device)
        self.on_launch = None
        c = AzspConfig()
        c.game = _abi.GAME_GO if cfg.game == "go" else _abi.GAME_GOMOKU
        c.board_size, c.num_games, c.num_parallel, c.num_simulations = cfg.board_size, cfg.num_games, cfg.num_parallel, cfg.num_simulations
        c.max_nodes, c.root_noise, c.deterministic, c.reuse_tree = cfg.max_nodes, int(cfg.root_noise), int(cfg.deterministic), int(cfg.reuse_tree)
        c.warm_up_steps = cfg.warm_up_steps
        c.has_resign = 1 if cfg.game == "go" else 0
        c.check_resign_after_steps, c.force_resign_disabled = cfg.check_resign_after_steps, cfg.force_resign_disabled
        c.inject_random, c.inject_moves = int(cfg.inject_random), cfg.inject_moves
        c.stop_after_move, c.max_plies, c.stop_at_game_end = int(cfg.stop_after_move), cfg.max_plies, int(cfg.stop_at_game_end)
        c.feature_dtype, c.log_moves, c.log_capacity = cfg.feature_dtype, int(cfg.log_moves), cfg.log_capacity
        c.max_steps, c.num_to_win, c.training_steps, c.rank = cfg.max_steps, cfg.num_to_win, cfg.training_steps, cfg.rank
        c.device = cfg.device_index
        c.c_puct_base, c.c_puct_init, c.disable_resign_ratio = cfg.c_puct_base, cfg.c_puct_init, cfg.disable_resign_ratio
        c.dirichlet_eps, c.dirichlet_alpha, c.resign_threshold, c.komi = cfg.dirichlet_eps, cfg.dirichlet_alpha, cfg.resign_threshold, cfg.komi
        c.seed = cfg.seed
        self.h = C.c_void_p()
        self.b.check(self.b.dll.azsp_create(C.byref(c), C.byref(self.h)), None, "azsp_create")
        g = AzspGeometry()
        self._ck(self.b.dll.azsp_geometry(self.h, C.byref(g)), "azsp_geometry")
        self.geo = g
        self.A, self.NP, self.N, self.G, self.P = g.num_actions, g.num_points, cfg.board_size, cfg.num_games, cfg.num_parallel
        self.rows = g.batch_rows
        t_np, t_py, sq = pbc_tables(cfg.c_puct_base, cfg.c_puct_init, g.table_len)
        self._ck(self.b.dll.azsp_set_tables(self.h, t_np.ctypes.data, t_py.ctypes.data, sq.ctypes.data, g.table_len), "azsp_set_tables")
        # evaluator-facing tensors (caller-visible; the engine reads / writes them in place)
        self.features_tiled = cfg.feature_dtype in (_abi.FEAT_BF16_TILED, _abi.FEAT_F16_TILED)
        self.features_split = cfg.feature_dtype == _abi.FEAT_F16_SPLIT
        if self.features_split:  # the fp32-class stem's input (AZSP_FEAT_F16_SPLIT): [row][hi, lo][4][N*N][8] f16; only hi planes are ever written
            self.features = torch.zeros((self.b.dll.azsp_split_bytes(self.rows, self.N, 32) // 2,), dtype=torch.float16, device=self.device)
        elif self.features_tiled:  # the evaluator's tiled layout (include/azsp.h: AZSP_FEAT_BF16_TILED / _F16_TILED), flat and zero-initialised
            n = self.b.dll.azsp_tiled_bytes(self.rows, self.N, 32) // 2
            self.features = torch.zeros((n,), dtype=torch.float16 if cfg.feature_dtype == _abi.FEAT_F16_TILED else torch.bfloat16, device=self.device)
        else:
            self.features = torch.zeros((self.rows, 17, self.N, self.N), dtype=_FEAT_TORCH[cfg.feature_dtype], device=self.device)
        self.valid = torch.zeros((self.rows,), dtype=torch.uint8, device=self.device)
        self.priors = torch.zeros((self.rows, self.A), dtype=torch.float32, device=self.device)
        self.values = torch.zeros((self.rows,), dtype=torch.float32, device=self.device)
        self._actions = torch.zeros((self.G,), dtype=torch.int32, device=self.device)
        self._harvest_bufs = None

    # -- plumbing ---------------------------------------------------------------------------------
    def _ck(self, rc, what):
        self.b.check(rc, self.h, what)

    def _stream(self):
        if self.device.type == "cuda":
            cur = torch.cuda.current_stream(self.device)
            if self.on_launch is not None:  # experiment hook (tools/overlap_actor.py joins its two half-batch streams here); never set by the product
                self.on_launch(cur)
            return C.c_void_p(cur.cuda_stream)
        return None

    def close(self):
        if self.h:
            self.b.dll.azsp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration ------------------------------------------------------------------------------
    def set_injection(self, noise=None, uniforms=None):
        """noise float64[G, moves, A], uniforms float64[G, moves, 16] (recorded reference randomness)."""
        m = self.cfg.inject_moves
        n = np.ascontiguousarray(noise, dtype=np.float64) if noise is not None else None
        u = np.ascontiguousarray(uniforms, dtype=np.float64) if uniforms is not None else None
        assert n is None or n.shape == (self.G, m, self.A)
        assert u is None or u.shape == (self.G, m, 16)
        self._ck(self.b.dll.azsp_set_injection(self.h, n.ctypes.data if n is not None else None,
                                               u.ctypes.data if u is not None else None, m), "azsp_set_injection")

    # -- games / environment --------------------------------------------------------------------------
    def reset_games(self):
        self._ck(self.b.dll.azsp_reset_games(self.h, self._stream()), "azsp_reset_games")

    def env_step(self, actions=None, want_obs=False):
        """Standalone env kernels.  actions int[G] (None = export only; -2 no-op, -1 resign).
        Returns dict(board int8[G,N,N], legal int8[G,A], scalars int32[G,12], obs int8[G,17,N,N] | None) on the host."""
        dev = self.device
        if actions is None:
            ap = None
        else:
            self._actions.copy_(torch.as_tensor(np.asarray(actions, dtype=np.int32)))
            ap = self._actions.data_ptr()
        board = torch.empty((self.G, self.N, self.N), dtype=torch.int8, device=dev)
        legal = torch.empty((self.G, self.A), dtype=torch.int8, device=dev)
        scal = torch.empty((self.G, 12), dtype=torch.int32, device=dev)
        obs = torch.empty((self.G, 17, self.N, self.N), dtype=torch.int8, device=dev) if want_obs else None
        self._ck(self.b.dll.azsp_env_step(self.h, ap, board.data_ptr(), legal.data_ptr(), scal.data_ptr(),
                                          obs.data_ptr() if want_obs else None, self._stream()), "azsp_env_step")
        return dict(board=board.cpu().numpy(), legal=legal.cpu().numpy(), scalars=scal.cpu().numpy(),
                    obs=obs.cpu().numpy() if want_obs else None)

    def set_state(self, slot, board, hist, to_play, steps, ko=-1, last_was_pass=False, caps=(0, 0)):
        b = np.ascontiguousarray(board, dtype=np.int8).reshape(-1)
        h = np.ascontiguousarray(hist, dtype=np.int8).reshape(8, -1)
        self._ck(self.b.dll.azsp_set_state(self.h, slot, b.ctypes.data, h.ctypes.data, int(to_play), int(steps), int(ko),
                                           int(bool(last_was_pass)), int(caps[0]), int(caps[1]), self._stream()), "azsp_set_state")

    # -- search ---------------------------------------------------------------------------------------
    def begin_move(self, noise=None, warm_up=-1):
        n = np.ascontiguousarray(noise, dtype=np.float64).reshape(self.G, self.A) if noise is not None else None
        self._ck(self.b.dll.azsp_begin_move(self.h, n.ctypes.data if n is not None else None, int(warm_up), self._stream()), "azsp_begin_move")

    def select(self, g0=None, g1=None):
        """Select the next leaves of all games, or (g0, g1 given, g0 % 32 == 0) of the games [g0, g1) only."""
        if g0 is None:
            self._ck(self.b.dll.azsp_select(self.h, self.features.data_ptr(), self.valid.data_ptr(), self._stream()), "azsp_select")
        else:
            self._ck(self.b.dll.azsp_select_range(self.h, self.features.data_ptr(), self.valid.data_ptr(), g0, g1, self._stream()), "azsp_select_range")

    def expand_backup(self, g0=None, g1=None):
        if g0 is None:
            self._ck(self.b.dll.azsp_expand_backup(self.h, self.priors.data_ptr(), self.values.data_ptr(), self._stream()), "azsp_expand_backup")
        else:
            self._ck(self.b.dll.azsp_expand_backup_range(self.h, self.priors.data_ptr(), self.values.data_ptr(), g0, g1, self._stream()),
                     "azsp_expand_backup_range")

    def round(self):
        """expand/backup with the current priors/values, then select the next leaves into features/valid."""
        self._ck(self.b.dll.azsp_round(self.h, self.priors.data_ptr(), self.values.data_ptr(), self.features.data_ptr(),
                                       self.valid.data_ptr(), self._stream()), "azsp_round")

    def status(self):
        st = np.zeros((self.G, 8), dtype=np.int32)
        q = np.zeros((self.G, 2), dtype=np.float64)
        self._ck(self.b.dll.azsp_get_status(self.h, st.ctypes.data, q.ctypes.data, self._stream()), "azsp_get_status")
        return st, q

    def dropin_step(self, priors=None, values=None, feature_rows=None):
        """One iteration of uct_search's simulation loop in ONE host round trip (azsp_dropin_step): upload eval_func's `priors`
        float32[rows, A] / `values` float32[rows] for the previous leaves (None on the first call of a search), expand / backup, select
        the next leaves, and return (status int32[G, 8], q float64[G, 2], valid bool[rows], obs [feature_rows, 17, N, N] of the engine's
        feature dtype).  Feature dtypes with a plain [rows, 17, N, N] tensor only (AZSP_FEAT_I8 / F32)."""
        assert not (self.features_tiled or self.features_split)
        nrow = self.rows if feature_rows is None else int(feature_rows)
        if getattr(self, "_dropin_bufs", None) is None:
            np_dt = {torch.int8: np.int8, torch.float32: np.float32}[self.features.dtype]
            self._dropin_bufs = (np.zeros((self.G, 8), dtype=np.int32), np.zeros((self.G, 2), dtype=np.float64), np.zeros(self.rows, dtype=np.uint8),
                                 np.zeros((self.rows, 17, self.N, self.N), dtype=np_dt))
        st, q, valid, obs = self._dropin_bufs
        pp = vp = None
        if priors is not None:
            priors = np.ascontiguousarray(priors, dtype=np.float32).reshape(self.rows, self.A)
            values = np.ascontiguousarray(values, dtype=np.float32).reshape(self.rows)
            pp, vp = priors.ctypes.data, values.ctypes.data
        self._ck(self.b.dll.azsp_dropin_step(self.h, pp, vp, self.priors.data_ptr(), self.values.data_ptr(), self.features.data_ptr(),
                                             self.valid.data_ptr(), st.ctypes.data, q.ctypes.data, valid.ctypes.data, obs.ctypes.data,
                                             nrow * obs[0].nbytes, self._stream()), "azsp_dropin_step")
        return st.copy(), q.copy(), valid.astype(bool), obs[:nrow].copy()  # copies: eval_func may keep what it is handed (the reference passes fresh arrays)

    def get_search(self, slot, ply=0):
        pi = np.zeros(self.A, dtype=np.float64)
        cn = np.zeros(self.A, dtype=np.float32)
        q = np.zeros(4, dtype=np.float64)
        self._ck(self.b.dll.azsp_get_search(self.h, slot, ply, pi.ctypes.data, cn.ctypes.data, q.ctypes.data, self._stream()), "azsp_get_search")
        return pi, cn, q

    def commit_move(self, moves):
        m = np.ascontiguousarray(moves, dtype=np.int32).reshape(self.G)
        self._ck(self.b.dll.azsp_commit_move(self.h, m.ctypes.data, self._stream()), "azsp_commit_move")

    def set_actor_state(self, resign_threshold, training_steps):
        """Values every game that STARTS from now on reads (pipeline.py:232-246: the reference actor re-reads the shared resign
        threshold and the checkpoint's training_steps before each game); games in progress keep theirs."""
        self._ck(self.b.dll.azsp_set_actor_state(self.h, C.c_double(float(resign_threshold)), int(training_steps)), "azsp_set_actor_state")

    # -- samples ----------------------------------------------------------------------------------------
    def harvest(self, sample_capacity=None, max_games=None, with_moves=False):
        """Returns (states int8[n,17,N,N], pi float32[n,A], z float32[n], games int32[k,16]) -- device tensors + host meta;
        with_moves=True appends moves int16[n] (the move played from every sample's position, -1 = resigned).
        The per-game extras of the same call (azsp_harvest_extra) are left in `self.last_extra` int32[k,4].
        ALIASING: the device tensors are views of buffers this engine re-uses -- the NEXT harvest() overwrites them in place.
        Consume (or .clone()) them before harvesting again; SelfPlayActor.harvest_tensors(clone=True) does the latter."""
        cap = sample_capacity or max(4 * self.G, 2 * self.geo.stage_capacity)
        mg = max_games or 2 * self.G
        if self._harvest_bufs is None or self._harvest_bufs[0].shape[0] < cap:
            self._harvest_bufs = (torch.empty((cap, 17, self.N, self.N), dtype=torch.int8, device=self.device),
                                  torch.empty((cap, self.A), dtype=torch.float32, device=self.device),
                                  torch.empty((cap,), dtype=torch.float32, device=self.device),
                                  torch.empty((cap,), dtype=torch.int16, device=self.device))
        st, pi, z, mvbuf = self._harvest_bufs
        self._ck(self.b.dll.azsp_harvest_moves(self.h, mvbuf.data_ptr() if with_moves else None), "azsp_harvest_moves")
        games = np.zeros((mg, 16), dtype=np.int32)
        extra = np.zeros((mg, 4), dtype=np.int32)
        self._ck(self.b.dll.azsp_harvest_extra(self.h, extra.ctypes.data), "azsp_harvest_extra")
        ns, ng = C.c_int32(0), C.c_int32(0)
        self._ck(self.b.dll.azsp_harvest(self.h, st.data_ptr(), pi.data_ptr(), z.data_ptr(), st.shape[0], games.ctypes.data, mg,
                                         C.byref(ns), C.byref(ng), self._stream()), "azsp_harvest")
        n, k = ns.value, ng.value
        self._ck(self.b.dll.azsp_harvest_extra(self.h, None), "azsp_harvest_extra")  # the host array dies with this call
        self.last_extra = extra[:k]
        if with_moves:
            return st[:n], pi[:n], z[:n], games[:k], mvbuf[:n]
        return st[:n], pi[:n], z[:n], games[:k]

    def counters(self, reset=False):
        out = np.zeros(16, dtype=np.uint64)
        self._ck(self.b.dll.azsp_counters(self.h, out.ctypes.data, int(reset), self._stream()), "azsp_counters")
        return {k: int(out[i]) for i, k in enumerate(_abi.COUNTER_NAMES)}
