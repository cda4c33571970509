"""Self-play actor on the batched engine (reference: alpha_zero/core/pipeline.py:83-382).

`game_stats_from_row` rebuilds the per-game `stats` dict of play_and_record_one_game
(pipeline.py:367-380) from the 16-int game record the engine emits at harvest time.
"""
from typing import Any, Dict

import numpy as np


def _result_string(row, game, komi):
    """go.py:194-200 + go_engine.py:527-534 / gomoku.py:138-147"""
    winner = int(row[2])
    if game == "go":
        if int(row[6]):  # resigned
            return "B+R" if winner == 1 else "W+R"
        score = float(int(row[3])) - (float(int(row[4])) + komi)
        if score > 0:
            return "B+" + "%.1f" % score
        if score < 0:
            return "W+" + "%.1f" % abs(score)
        return "DRAW"
    return "B+1.0" if winner == 1 else "W+1.0" if winner == 2 else "DRAW"


def game_stats_from_row(row, game: str, komi: float = 7.5, resign_threshold: float = -1.0) -> Dict[str, Any]:
    stats: Dict[str, Any] = {"game_length": int(row[1]), "game_result": _result_string(row, game, komi)}
    if game == "go":  # has_pass_move / has_resign_move (pipeline.py:372-380)
        stats["num_passes"] = int(row[5])
        stats["is_resign_disabled"] = bool(row[7])
        stats["is_marked_for_resign"] = bool(row[8])
        stats["is_could_won"] = bool(row[9])
        mp = int(row[10])
        stats["marked_resign_player"] = "B" if mp == 1 else "W" if mp == -1 else None
        stats["resign_threshold"] = resign_threshold
    return stats
