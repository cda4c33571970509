"""Go environment (reference: alpha_zero/envs/go.py:19-210; rules alpha_zero/envs/go_engine.py) on the
engine's bitboard kernels: liberty / capture / suicide / simple ko / Tromp-Taylor area score."""
import numpy as np

from .base import BoardGameEnv

BLACK, WHITE = 1, -1


class GoEnv(BoardGameEnv):
    _game = "go"

    def __init__(self, komi=7.5, num_stack=8, max_steps=0, board_size=None, **kw):
        """The reference fixes the board size through the BOARD_SIZE environment variable read at import time
        (go_engine.py:31); here it is an argument, defaulting to that variable (19 if unset)."""
        import os

        n = board_size or int(os.environ.get("BOARD_SIZE", 19))
        self.komi = komi
        self.max_steps = max_steps or n * n * 2  # go.py:48
        super().__init__(id="Go", board_size=n, num_stack=num_stack, black_player_id=BLACK, white_player_id=WHITE,
                         has_pass_move=True, has_resign_move=True, komi=komi, max_steps=self.max_steps, **kw)

    def step(self, action):
        obs, reward, done, info = super().step(action)
        if action == self.resign_move:
            reward = -1  # go.py:119 returns the int -1
        return obs, reward, done, info

    def get_captures(self):
        return {self.black_player: self._caps[0], self.white_player: self._caps[1]}

    def area_score(self):
        """(black, white) Tromp-Taylor areas of the current board (go_engine.py:123-152)."""
        return self._areas

    def get_result_string(self):
        if self.last_move == self.resign_move:
            return "B+R" if self.winner == self.black_player else "W+R"  # go.py:194-198
        score = float(self._areas[0]) - (float(self._areas[1]) + self.komi)  # go_engine.py:509-534
        if score > 0:
            return "B+" + "%.1f" % score
        if score < 0:
            return "W+" + "%.1f" % abs(score)
        return "DRAW"
