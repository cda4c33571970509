"""Golden SGF records from the reference's make_sgf / env.to_sgf (this container only): move lists, result strings
and the exact text the reference writes, with the date pinned.  Writes tests/golden/sgf_records.json."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

ref_harness.install(9)
import alpha_zero.envs.go as rgo  # noqa: E402
import alpha_zero.envs.gomoku as rgm  # noqa: E402
from alpha_zero.envs.base import PlayerMove  # noqa: E402
from alpha_zero.utils import sgf_wrapper  # noqa: E402

rgo.get_time_stamp = lambda *a, **k: "2024-01-02 03:04:05"
rgm.get_time_stamp = lambda *a, **k: "2024-01-02 03:04:05"
rng = np.random.Generator(np.random.PCG64(7))
records = []
for trial in range(3):
    env = rgo.GoEnv(komi=7.5, num_stack=8)
    env.reset()
    done, moves = False, []
    while not done:
        legal = np.flatnonzero(env.legal_actions)
        a = int(legal[rng.integers(len(legal))]) if rng.random() > 0.03 else env.pass_move
        if trial == 2 and env.steps == 30:
            a = env.resign_move
        moves.append(a)
        _, _, done, _ = env.step(a)
    records.append({"game": "go", "n": 9, "komi": 7.5, "moves": moves, "result": env.get_result_string(), "sgf": env.to_sgf()})
env = rgm.GomokuEnv(board_size=9, num_stack=8)
env.reset()
done, moves = False, []
while not done:
    legal = np.flatnonzero(env.legal_actions)
    a = int(legal[rng.integers(len(legal))])
    moves.append(a)
    _, _, done, _ = env.step(a)
records.append({"game": "gomoku", "n": 9, "moves": moves, "result": env.get_result_string(), "sgf": env.to_sgf()})
# make_sgf directly: comments (with a ']' to escape), names, ranks, more than 10 moves (line breaks)
hist = [PlayerMove("B" if i % 2 == 0 else "W", m) for i, m in enumerate([0, 80, 81, 40, 12, 13, 14, 15, 16, 17, 18, 19])]
comments = ["first", None, "pass [here]", None]
records.append({"game": "raw", "n": 9, "history": [[h.color, h.move] for h in hist], "comments": comments, "result": "W+0.5",
                "kwargs": {"ruleset": "Chinese", "komi": 5.5, "white_name": "w", "white_rank": "1d", "black_name": "b", "black_rank": "2k", "date": "d"},
                "sgf": sgf_wrapper.make_sgf(9, hist, "W+0.5", ruleset="Chinese", komi=5.5, white_name="w", white_rank="1d", black_name="b", black_rank="2k",
                                            date="d", comments=comments)})
dst = os.path.join(ROOT, "tests", "golden", "sgf_records.json")
json.dump(records, open(dst, "w"), indent=0)
print("wrote", dst, [len(r["sgf"]) for r in records]); print(records[-1]["sgf"])
