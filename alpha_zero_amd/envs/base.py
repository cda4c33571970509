"""Board-game environment surface of the reference (alpha_zero/envs/base.py:26-364) on top of the
engine's environment kernels.

Each env object owns a one-game engine (C ABI: azsp_env_step / azsp_set_state); `step()` launches the
rules kernel on the GPU and mirrors board / legal mask / scalars back into the NumPy attributes that the
reference search and actor read.  There is no host-side rules implementation.
"""
from collections import deque, namedtuple
from copy import copy

import numpy as np

from .coords import CoordsConvertor


class PlayerMove(namedtuple("PlayerMove", ["color", "move"])):
    """History record (base.py:19-22)."""


def _engine_for(game, board_size, komi, max_steps, num_to_win, binding, device):
    from ..core.engine import Engine, EngineConfig

    if binding is None:
        from .. import _lib

        binding, device = _lib.load(require_gpu=True), "cuda"
    cfg = EngineConfig(game=game, board_size=board_size, num_games=1, num_parallel=1, num_simulations=2, komi=komi,
                       max_steps=max_steps, num_to_win=num_to_win, stop_after_move=True)
    return Engine(binding, cfg, device=device), binding, device


class BoardGameEnv:
    """Attribute-compatible stand-in for the reference BoardGameEnv.  The plain base env (stones are placed,
    the game never ends: base.py:184-213) is served by the Gomoku kernel with an unreachable win length."""

    _game = "gomoku"
    metadata = {"render.modes": ["terminal"], "players": ["black", "white"]}

    def __init__(self, board_size=15, num_stack=8, black_player_id=1, white_player_id=2, has_pass_move=False,
                 has_resign_move=False, id="", *, komi=7.5, max_steps=0, num_to_win=99, _binding=None, _device=None):
        assert black_player_id != white_player_id != 0, "player ids can not be the same, and can not be zero"
        if num_stack != 8:
            raise ValueError("the engine stacks exactly 8 history boards (num_stack=8)")
        self.id, self.board_size, self.num_stack = id, board_size, num_stack
        self.black_player, self.white_player = black_player_id, white_player_id
        self.has_pass_move, self.has_resign_move = has_pass_move, has_resign_move
        self.action_dim = board_size ** 2 + 1 if has_pass_move else board_size ** 2
        self.pass_move = self.action_dim - 1 if has_pass_move else None
        self.resign_move = -1 if has_resign_move else None
        self.cc = CoordsConvertor(board_size)
        self.gtp_columns = "ABCDEFGHJKLMNOPQRSTUVWXYZ"
        self.gtp_rows = [str(i) for i in range(board_size, -1, -1)]
        self._komi, self._max_steps, self._num_to_win = komi, max_steps, num_to_win
        self._eng, self._binding, self._device = _engine_for(self._game, board_size, komi, max_steps, num_to_win, _binding, _device)
        self.history = []
        self.reset()

    # -- state mirror ---------------------------------------------------------------------------------
    def _pull(self, out):
        sc = out["scalars"][0]
        self.board = out["board"][0].copy()
        done = bool(sc[5])
        # dtype follows the reference: Go masks come from np.concatenate(..., [1]) -> int64 (go_engine.py:441);
        # Gomoku / base masks and the all-zero terminal mask are int8 (base.py:72, go.py:142)
        self.legal_actions = out["legal"][0].astype(np.int64 if (self._game == "go" and not done) else np.int8)
        self.to_play = int(sc[4])
        self.steps = int(sc[3])
        self.winner = int(sc[7]) or None
        self._done, self._reward = done, float(sc[6])
        self.ko, self._caps, self._last_pass = int(sc[0]), (int(sc[1]), int(sc[2])), bool(sc[11])
        self._areas = (int(sc[8]), int(sc[9]))
        self._obs = out["obs"][0].copy()

    def reset(self, **kwargs):
        self._eng.reset_games()
        self._pull(self._eng.env_step(None, want_obs=True))
        self.last_player = None
        self.last_move = None
        self.board_deltas = self.get_empty_queue()
        del self.history[:]
        return self.observation()

    def step(self, action):
        if self.is_game_over():
            raise RuntimeError("Game is over, call reset before using step method.")
        if action is not None and action != self.resign_move and not 0 <= int(action) <= self.action_dim - 1:
            raise ValueError(f"Invalid action. The action {action} is out of bound.")
        if action is not None and action != self.resign_move and self.legal_actions[int(action)] != 1:
            raise ValueError(f"Illegal action {action}.")
        mover = self.to_play
        out = self._eng.env_step([int(action)], want_obs=True)
        if out["scalars"][0][10]:
            raise ValueError(f"Illegal action {action}.")
        self._pull(out)
        self.last_move, self.last_player = copy(int(action)), mover
        self.add_to_history(mover, self.last_move)
        self.board_deltas.appendleft(np.copy(self.board))
        reward = self._reward if self._done else 0.0
        return self.observation(), reward, self._done, {}

    def observation(self):
        """[X_t, Y_t, ..., X_t-7, Y_t-7, C] int8 planes from the player to move's perspective (base.py:228-259)."""
        return self._obs.copy()

    # -- helpers with the reference's names -------------------------------------------------------------
    def close(self):
        self.board_deltas.clear()
        del self.history[:]

    def add_to_history(self, player_id, move):
        if move != self.resign_move:
            self.history.append(PlayerMove(color=self.get_player_name_by_id(player_id), move=move))

    def get_empty_queue(self):
        return deque([np.zeros((self.board_size, self.board_size))] * self.num_stack, maxlen=self.num_stack)

    def is_board_full(self):
        return bool(np.all(self.board != 0))

    def is_pass_move(self, move):
        return self.has_pass_move and move == self.pass_move

    def is_resign_move(self, move):
        return self.has_resign_move and move == self.resign_move

    def is_legal_move(self, move):
        if move is None or move < 0 or move > self.action_dim - 1:
            return False
        return self.legal_actions[move] == 1

    def is_coords_on_board(self, coords):
        x, y = coords
        return max(x, y) < self.board_size and min(x, y) >= 0

    def action_to_coords(self, action):
        return (-1, -1) if action is None else self.cc.from_flat(action)

    def action_to_gtp(self, action):
        try:
            return self.cc.to_gtp(self.cc.from_flat(action))
        except Exception:
            return None

    def coords_to_action(self, coords):
        try:
            return self.cc.to_flat(coords) if self.is_coords_on_board(coords) else None
        except Exception:
            return None

    def gtp_to_action(self, gtpc, check_illegal=True):
        try:
            action = self.cc.to_flat(self.cc.from_gtp(gtpc))
            if action < 0 or action >= self.action_dim:
                return None
            if check_illegal and self.legal_actions[action] != 1:
                return None
            return action
        except Exception:
            return None

    def get_player_name_by_id(self, id):
        return "B" if id == self.black_player else "W" if id == self.white_player else None

    def is_game_over(self):
        return bool(self._done)

    @property
    def opponent_player(self):
        return self.white_player if self.to_play == self.black_player else self.black_player

    def get_captures(self):
        return {self.black_player: 0, self.white_player: 0}

    def get_result_string(self):
        return ""

    def to_sgf(self):
        """Game record as SGF text (go.py:202-210 / gomoku.py:149-157): Go writes its ruleset and komi, Gomoku leaves both empty."""
        from ..utils.sgf import get_time_stamp, make_sgf

        is_go = self._game == "go"
        return make_sgf(board_size=self.board_size, move_history=self.history, result_string=self.get_result_string(),
                        ruleset="Chinese" if is_go else "", komi=self._komi if is_go else "", date=get_time_stamp())

    # -- copying: a copy is a new one-game engine loaded with this position (copy.deepcopy works, mcts_v2.py:382)
    def _hist_boards(self):
        return np.stack([np.asarray(b, dtype=np.int8) for b in self.board_deltas])

    def __deepcopy__(self, memo):
        new = object.__new__(type(self))
        for k, v in self.__dict__.items():
            if k in ("_eng",):
                continue
            new.__dict__[k] = v.copy() if isinstance(v, np.ndarray) else v
        new.history = list(self.history)
        new.board_deltas = deque([np.copy(b) for b in self.board_deltas], maxlen=self.num_stack)
        new._eng, _, _ = _engine_for(self._game, self.board_size, self._komi, self._max_steps, self._num_to_win, self._binding, self._device)
        if not self._done:  # a finished game needs no device state: step() raises before touching the engine
            new._eng.set_state(0, self.board, self._hist_boards(), self.to_play, self.steps, self.ko, self._last_pass, self._caps)
        return new
