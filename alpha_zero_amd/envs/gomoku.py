"""Free-style Gomoku environment (reference: alpha_zero/envs/gomoku.py:17-158) on the engine's kernels."""
from .base import BoardGameEnv


class GomokuEnv(BoardGameEnv):
    _game = "gomoku"

    def __init__(self, board_size=15, num_to_win=5, num_stack=8, **kw):
        self.num_to_win = num_to_win
        super().__init__(id="Freestyle Gomoku", board_size=board_size, num_stack=num_stack, has_pass_move=False,
                         has_resign_move=False, num_to_win=num_to_win, **kw)

    def get_result_string(self):
        if not self.is_game_over():
            return ""
        if self.winner == self.black_player:
            return "B+1.0"
        if self.winner == self.white_player:
            return "W+1.0"
        return "DRAW"
