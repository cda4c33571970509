"""Restatement of the reference self-play actor (alpha_zero/core/pipeline.py:289-382).
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

`play_one_game` follows play_and_record_one_game: per move it searches, records
(observation before the move, search_pi, to_play), applies the resignation rule,
steps the env, and finally back-fills z.  `hooks` lets tests inject recorded
randomness per move and observe per-move search outputs.
"""
import copy
from typing import NamedTuple, Optional

import numpy as np

from . import mcts


class Transition(NamedTuple):  # core/replay.py:14-17
    state: Optional[np.ndarray]
    pi_prob: Optional[np.ndarray]
    value: Optional[float]


def play_one_game(env, eval_func, *, num_simulations, num_parallel, c_puct_base=19652.0, c_puct_init=1.25, warm_up_steps=16,
                  check_resign_after_steps=40, resign_threshold=-1.0, resign_disabled=True, root_noise=True,
                  deterministic=False, reuse_tree=True, rand_for_move=None, on_move=None, max_moves=None):
    obs = env.reset()
    done = False
    states, pis, values, to_plays = [], [], [], []
    root = None
    marked_resign_player = None
    is_marked_for_resign = False
    is_could_won = False
    num_passes = 0
    reward = 0.0
    k = 0
    while not done:
        if max_moves is not None and k >= max_moves:
            return None, None
        rand = rand_for_move(k) if rand_for_move is not None else None
        warm_up = False if env.steps > warm_up_steps else True  # pipeline.py:320
        if not reuse_tree:
            root = None
        kw = dict(env=env, eval_func=eval_func, root_node=root, c_puct_base=c_puct_base, c_puct_init=c_puct_init,
                  num_simulations=num_simulations, root_noise=root_noise, warm_up=warm_up, deterministic=deterministic, rand=rand)
        if num_parallel > 1:  # pipeline.py:132-156
            move, pi, root_q, child_q, root = mcts.parallel_uct_search(num_parallel=num_parallel, **kw)
        else:
            move, pi, root_q, child_q, root = mcts.uct_search(**kw)
        if on_move is not None:
            on_move(k, env, move, pi, root_q, child_q, root)
        states.append(obs)
        pis.append(pi)
        values.append(0.0)
        to_plays.append(env.to_play)
        # pipeline.py:328-341
        if env.has_resign_move and env.steps > check_resign_after_steps and root_q < resign_threshold and child_q < resign_threshold:
            if marked_resign_player is None:
                marked_resign_player = copy.copy(env.to_play)
            if not resign_disabled:
                move = env.resign_move
        obs, reward, done, _ = env.step(move)
        if env.has_pass_move and move == env.pass_move:
            num_passes += 1
        k += 1
    if reward != 0.0:  # pipeline.py:349-354
        for i, pid in enumerate(to_plays):
            values[i] = reward if pid == env.last_player else -reward
    game_seq = [Transition(state=s, pi_prob=p, value=v) for s, p, v in zip(states, pis, values)]
    if env.has_resign_move and resign_disabled and marked_resign_player is not None:  # pipeline.py:361-365
        is_marked_for_resign = True
        if env.winner == marked_resign_player:
            is_could_won = True
    stats = {"game_length": len(game_seq), "game_result": env.get_result_string()}
    if env.has_pass_move:
        stats["num_passes"] = num_passes
    if env.has_resign_move:
        stats["is_resign_disabled"] = resign_disabled
        stats["is_marked_for_resign"] = is_marked_for_resign
        stats["is_could_won"] = is_could_won
        stats["marked_resign_player"] = env.get_player_name_by_id(marked_resign_player)
        stats["resign_threshold"] = resign_threshold
    return game_seq, stats
