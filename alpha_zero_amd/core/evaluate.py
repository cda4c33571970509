"""Evaluator-side pieces of the pipeline on top of the engine.

  * EloRating / get_k_factor            reference alpha_zero/core/rating.py:11-70
  * create_mcts_player                  core/pipeline.py:83-163 (eval_position + act), searches = this package's drop-ins
  * eval_against_prev_ckpt              core/pipeline.py:815-867: latest network plays black, previous checkpoint white, no noise,
                                        argmax moves, a fresh tree every move; Elo update; returns the reference's stats dict
  * maybe_adjust_resign_threshold       core/pipeline.py:656-670
  * ResignController                    the learner's false-positive bookkeeping, core/pipeline.py:519-553
Every search runs in the HIP engine through `uct_search` / `parallel_uct_search` (core/mcts_v2.py of this package)."""
import math
from typing import Iterable

import numpy as np

from .mcts_v2 import parallel_uct_search, uct_search


def round_it(v, places=4):
    """core/pipeline.py:71-72"""
    return round(v, places)


def get_k_factor(player_ratings: Iterable[float]) -> int:
    """USCF K-factor (rating.py:11-30): 32 below 2100, 24 between 2100 and 2400, 16 above 2400."""
    r = list(player_ratings)
    k = 32
    if all(x < 2100 for x in r):
        k = 32
    elif all(x < 2400 for x in r) and any(x >= 2100 for x in r):
        k = 24
    elif all(x >= 2400 for x in r):
        k = 16
    return k


class EloRating:
    """rating.py:33-70"""

    def __init__(self, rating=0):
        self.rating = rating

    def expected_score(self, opponent_rating):
        return 1 / (1 + math.pow(10, (opponent_rating - self.rating) / 400))

    def update_rating(self, opponent_rating, actual_score):
        expected = self.expected_score(opponent_rating)
        self.rating += get_k_factor((self.rating, opponent_rating)) * (actual_score - expected)


def create_mcts_player(network=None, device=None, num_simulations=800, num_parallel=8, root_noise=False, deterministic=False, eval_func=None):
    """Same contract as the reference factory (pipeline.py:83-163): returns act(env, root_node, c_puct_base, c_puct_init,
    warm_up) -> (move, search_pi, root_Q, best_child_Q, next_root).  `eval_func` may replace the network wrapper (tests)."""
    if eval_func is None:
        import torch

        @torch.no_grad()
        def eval_func(state, batched=False):  # pipeline.py:91-123
            if not batched:
                state = state[None, ...]
            x = torch.from_numpy(np.ascontiguousarray(state)).to(dtype=torch.float32, device=device)
            pi_logits, v = network(x)
            pi = torch.softmax(pi_logits.float(), dim=-1).cpu().numpy()
            v = np.squeeze(v.float().cpu().numpy(), axis=1).tolist()
            pi = [pi[i] for i in range(x.shape[0])]
            return (pi, v) if batched else (pi[0], v[0])

    def act(env, root_node, c_puct_base, c_puct_init, warm_up=False):
        if num_parallel > 1:
            return parallel_uct_search(env=env, eval_func=eval_func, root_node=root_node, c_puct_base=c_puct_base, c_puct_init=c_puct_init,
                                       num_simulations=num_simulations, num_parallel=num_parallel, root_noise=root_noise, warm_up=warm_up,
                                       deterministic=deterministic)
        return uct_search(env=env, eval_func=eval_func, root_node=root_node, c_puct_base=c_puct_base, c_puct_init=c_puct_init,
                          num_simulations=num_simulations, root_noise=root_noise, warm_up=warm_up, deterministic=deterministic)

    return act


def eval_against_prev_ckpt(env, black_player, white_player, black_elo, white_elo, c_puct_base, c_puct_init):
    """One evaluation game and the Elo update (pipeline.py:815-867); the returned keys are the evaluation.csv columns."""
    env.reset()
    done, num_passes = False, 0
    while not done:
        player = black_player if env.to_play == env.black_player else white_player
        move, *_ = player(env=env, root_node=None, c_puct_base=c_puct_base, c_puct_init=c_puct_init, warm_up=False)
        _, _, done, _ = env.step(move)
        if env.has_pass_move and move == env.pass_move:
            num_passes += 1
    stats = {"game_length": env.steps, "game_result": env.get_result_string()}
    if env.has_pass_move:
        stats["num_passes"] = num_passes
    if env.winner is not None:
        winner, loser = (black_elo, white_elo) if env.winner == env.black_player else (white_elo, black_elo)
        winner.update_rating(loser.rating, 1)
        loser.update_rating(winner.rating, 0)
    stats["black_elo_rating"] = black_elo.rating
    stats["white_elo_rating"] = white_elo.rating
    return stats


def maybe_adjust_resign_threshold(current_v, current_rate, target_rate, min_v=-0.9999, smoothing_factor=0.5):
    """pipeline.py:656-670: raise |threshold| only while the measured false-positive rate exceeds the target."""
    rate_delta = current_rate - target_rate
    if rate_delta <= 0:
        return current_v
    new_v = current_v + current_v * rate_delta
    smoothed_v = smoothing_factor * new_v + (1 - smoothing_factor) * current_v
    return round_it(max(min_v, smoothed_v))


class ResignController:
    """The learner's resignation bookkeeping (pipeline.py:519-553) as an object: feed it the stats of every received game
    together with the replay's game count; `threshold` is what the reference keeps in `var_resign_threshold`."""

    def __init__(self, init_resign_threshold, no_resign_games, reset_fp_interval, games_per_ckpt, disable_resign_ratio, target_fp_rate=0.05):
        self.init, self.no_resign_games, self.reset_fp_interval = init_resign_threshold, no_resign_games, reset_fp_interval
        self.step = int(games_per_ckpt * 0.5 * disable_resign_ratio * 0.5)
        self.target_fp_rate = target_fp_rate
        # pipeline.py:449-459: -1 while resignation is permanently off (init <= -1) or during the no-resign warm-up games
        self.threshold = -1 if (init_resign_threshold <= -1 or no_resign_games > 0) else init_resign_threshold
        self.resign_count = self.last_resign_count = self.could_won_count = 0

    def on_game(self, stats, num_games_added):
        if not (self.init > -1.0 and num_games_added >= self.no_resign_games):
            return self.threshold
        if stats.get("is_resign_disabled") and stats.get("is_marked_for_resign") and "is_resign_disabled" in stats and "is_marked_for_resign" in stats:
            self.resign_count += 1
            if stats.get("is_could_won"):
                self.could_won_count += 1
        if num_games_added == self.no_resign_games or num_games_added % self.reset_fp_interval == 0:
            self.resign_count = self.last_resign_count = self.could_won_count = 0
            self.threshold = self.init
        elif self.resign_count > self.last_resign_count and self.resign_count % self.step == 0:
            self.last_resign_count = self.resign_count
            rate = 0 if self.resign_count == 0 else round_it(self.could_won_count / self.resign_count)
            self.threshold = maybe_adjust_resign_threshold(self.threshold, rate, self.target_fp_rate)
        return self.threshold
