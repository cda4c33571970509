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


class DeviceEvaluator:
    """An evaluator for play_eval_games_parallel whose leaf rows never leave the device: wraps an InferenceNet (core/network.py).
    `device_eval(x)` takes the engine's int8 observation rows [B,17,N,N] on the device and returns (priors f32[B,A], values f32[B])
    on the device; calling the object with (state, batched) numpy arrays honours the reference's eval_func contract
    (pipeline.py:91-123), so the same object also serves the drop-in searches."""

    POLL_EVERY = 32  # leaf batches between two polls of the fp32-class evaluator's range record (a poll synchronises the stream)
    MAX_GRAPHS = 8   # captured forwards kept (one per engine buffer set and evaluator state)

    def __init__(self, inference_net, use_graph=True):
        self.inf = inference_net
        self.use_graph = use_graph
        self._calls = 0
        self._graphs = {}

    def _forward(self, x, priors_out=None, values_out=None):
        if hasattr(self.inf, "forward_planes"):
            return self.inf.forward_planes(x, priors_out, values_out)
        return self.inf(x, priors_out, values_out) if priors_out is not None else self.inf(x)

    def _poll(self, x):
        self._calls += 1
        if self._calls % self.POLL_EVERY == 0 and hasattr(self.inf, "poll_range"):
            self.inf.poll_range(x)  # never silent, never left clamping: warn + rescale (InferenceNet.poll_range)

    def device_eval(self, x):
        import torch

        with torch.no_grad():
            pri, v = self._forward(x)
        self._poll(x)
        return pri.float(), v.float()

    def _state_key(self):
        i = self.inf  # everything a captured forward bakes in besides the (in-place updated) weight tensors
        return (id(i), getattr(i, "act_shift", 0), getattr(i, "act_calibrated", True), getattr(i, "split_fallback_reason", None),
                getattr(i, "stem_fallback_reason", None), getattr(i, "use_split_tower", None), getattr(i, "use_fused_block", None))

    def device_eval_into(self, x, priors_out, values_out):
        """The evaluator between an engine's own tensors: x = its feature rows, outputs written in place into its priors / values (fp32).
        On a HIP device the forward is replayed from a hipGraph captured per (buffers, evaluator state): a batch-1 forward of the drop-in
        searches is ~14 launches of a few microseconds each, i.e. launch-bound from Python (core/mcts_v2.py _simulate_on_device).  A change
        of the evaluator's state (activation scale, fallback to the library, kernel switches) drops the capture."""
        import torch

        if not (self.use_graph and x.is_cuda):
            with torch.no_grad():
                self._forward(x, priors_out, values_out)
            self._poll(x)
            return
        key = (x.data_ptr(), tuple(x.shape), x.dtype, priors_out.data_ptr(), values_out.data_ptr(), self._state_key())
        g = self._graphs.get(key)
        if g is None:
            with torch.no_grad():
                side = torch.cuda.Stream(x.device)
                side.wait_stream(torch.cuda.current_stream(x.device))
                with torch.cuda.stream(side):
                    for _ in range(2):  # eager first: calibration of the fp32-class activation scale, scratch buffers, library kernel choice
                        self._forward(x, priors_out, values_out)
                torch.cuda.current_stream(x.device).wait_stream(side)
                torch.cuda.synchronize(x.device)
                key = key[:-1] + (self._state_key(),)  # (the eager calls may have calibrated the scale or given the split kernels up)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._forward(x, priors_out, values_out)
            if len(self._graphs) >= self.MAX_GRAPHS:
                self._graphs.clear()
            self._graphs[key] = g
        g.replay()
        self._poll(x)

    def __call__(self, state, batched=False):
        import torch

        x = torch.from_numpy(np.ascontiguousarray(state if batched else state[None, ...])).to(next(self.inf.parameters()).device)
        pri, v = self.device_eval(x)
        pri, v = pri.cpu().numpy(), v.cpu().numpy().tolist()
        pi = [pri[i] for i in range(x.shape[0])]
        return (pi, v) if batched else (pi[0], v[0])


def play_eval_games_parallel(game, board_size, players, num_simulations, num_parallel, c_puct_base, c_puct_init, komi=7.5, num_to_win=5,
                             binding=None, device="cuda", max_rounds=1 << 20, openings=None):
    """SURVEY 8f-2 "many in parallel": G evaluation games of pipeline.py:815-867 advance in lock-step on ONE engine.

    players: one (black_eval, white_eval) pair per game; eval(states int8[B,17,N,N], True) -> (list of pi, list of v) with the
    reference's eval_func contract (pipeline.py:91-123).  Every game is what eval_against_prev_ckpt plays: no root noise, arg-max
    moves, a fresh tree for every move (`root_node=None`, pipeline.py:836), the searching player's own evaluator for all leaves of
    its search.  The whole game runs inside the engine (search, move, env step, termination, scoring).  Per round the leaf rows of all
    games whose side to move uses the same evaluator go to it in ONE batch; evaluators with a `device_eval` method (DeviceEvaluator)
    receive and return device tensors -- their rows are gathered / scattered by index on the device and never pass through the host
    (only the G side-to-move flags do).  openings: optional list of G move lists played (env.step) before the first search, e.g.
    random openings for a match between two evaluators.  Returns per game dict(moves, game_length, game_result, num_passes, winner)
    -- the same games, move for move, as G sequential calls (tests/arena_checks.py); `moves` includes the opening moves."""
    import torch

    from .. import _abi, _lib
    from .engine import Engine, EngineConfig
    from .pipeline import game_stats_from_row

    binding = binding or _lib.load(require_gpu=True)
    G = len(players)
    cfg = EngineConfig(game=game, board_size=board_size, num_games=G, num_parallel=num_parallel, num_simulations=num_simulations,
                       c_puct_base=c_puct_base, c_puct_init=c_puct_init, root_noise=False, deterministic=True, reuse_tree=False, warm_up_steps=-1,
                       komi=komi, num_to_win=num_to_win, resign_threshold=-1.0, stop_at_game_end=True, feature_dtype=_abi.FEAT_I8)
    eng = Engine(binding, cfg, device=device)
    try:
        eng.reset_games()
        A, P = eng.A, num_parallel
        if openings is not None:
            assert len(openings) == G
            for t in range(max(len(o) for o in openings)):
                acts = np.array([o[t] if t < len(o) else -2 for o in openings], dtype=np.int32)
                out = eng.env_step(acts)
                # the env kernel skips an illegal or post-terminal action and flags it (scalars[10]); an opening that was not
                # played as given would shift the colour / evaluator pairing by one ply and report moves that never happened
                bad = np.nonzero((acts != -2) & (out["scalars"][:, 10] != 0))[0]
                if len(bad):
                    raise ValueError(f"opening move {t} of game {int(bad[0])} (action {int(acts[bad[0]])}) was rejected by the rules")
        # distinct evaluator objects and, per game and colour, which one searches
        evs, who = [], np.zeros((G, 2), dtype=np.int64)
        for g, pair in enumerate(players):
            for c in (0, 1):
                k = next((i for i, e in enumerate(evs) if e is pair[c]), None)
                if k is None:
                    evs.append(pair[c])
                    k = len(evs) - 1
                who[g, c] = k
        on_device = [hasattr(e, "device_eval") for e in evs]
        for _ in range(max_rounds):
            eng.expand_backup()
            eng.select()
            st, _ = eng.status()
            if np.all(st[:, 0] == _abi.ST_IDLE):
                break
            side = (st[:, 1] & 1).astype(np.int64)                      # ply even: black is searching
            ev_of_game = who[np.arange(G), side]
            valid_dev = eng.valid.view(G, P).bool()
            host_needed = not all(on_device[k] for k in np.unique(ev_of_game))
            if host_needed:
                valid = valid_dev.cpu().numpy()
                feats = eng.features.cpu().numpy().reshape(G, P, 17, board_size, board_size)
                pri = np.zeros((G, P, A), dtype=np.float32)
                val = np.zeros((G, P), dtype=np.float32)
            for k, ev in enumerate(evs):
                games_k = np.flatnonzero(ev_of_game == k)
                if len(games_k) == 0:
                    continue
                if on_device[k]:  # gather this evaluator's valid leaf rows on the device, evaluate, scatter back
                    mask = torch.zeros(G, dtype=torch.bool, device=eng.device)
                    mask[torch.from_numpy(games_k).to(eng.device)] = True
                    idx = torch.nonzero((valid_dev & mask[:, None]).reshape(-1)).flatten()
                    if idx.numel() == 0:
                        continue
                    p_k, v_k = ev.device_eval(eng.features.index_select(0, idx))
                    eng.priors.index_copy_(0, idx, p_k.to(eng.priors.dtype))
                    eng.values.index_copy_(0, idx, v_k.to(eng.values.dtype))
                    continue
                gg, rr = np.nonzero(valid[games_k])
                if len(gg) == 0:
                    continue
                ps, vs = ev(feats[games_k[gg], rr], True)   # ONE batch per evaluator and round
                for g_, r_, p_, v_ in zip(games_k[gg], rr, ps, vs):
                    pri[g_, r_], val[g_, r_] = np.asarray(p_, dtype=np.float32), v_
            if host_needed:
                host_rows = torch.from_numpy(np.flatnonzero(np.repeat(~np.array([on_device[k] for k in ev_of_game]), P))).to(eng.device)
                if host_rows.numel():
                    eng.priors.index_copy_(0, host_rows, torch.from_numpy(pri.reshape(G * P, A)).to(eng.device).index_select(0, host_rows))
                    eng.values.index_copy_(0, host_rows, torch.from_numpy(val.reshape(G * P)).to(eng.device).index_select(0, host_rows))
        else:
            raise RuntimeError("evaluation games did not finish")
        got = eng.harvest(sample_capacity=G * eng.geo.stage_capacity, max_games=G, with_moves=True)  # room for every game at full length
        games, moves = got[3], got[4].cpu().numpy()
        out = [None] * G
        for row in games:
            g, s0, ln = int(row[15]), int(row[0]), int(row[1])
            stats = game_stats_from_row(row, game, komi)
            opening = [int(m) for m in openings[g]] if openings is not None else []
            out[g] = dict(moves=opening + [int(m) for m in moves[s0:s0 + ln]], game_length=len(opening) + stats["game_length"], game_result=stats["game_result"],
                          num_passes=stats.get("num_passes"), winner=int(row[2]))
        if not all(o is not None for o in out):
            raise RuntimeError("evaluation games did not all reach the harvest")
        return out
    finally:
        eng.close()  # the engine owns device memory and a native handle: released on every path (an evaluator may raise)


def eval_many_against_prev_ckpt(game, board_size, players, elos, num_simulations, num_parallel, c_puct_base, c_puct_init, **kw):
    """G evaluation games at once + the Elo bookkeeping of eval_against_prev_ckpt applied in game order: `elos` is one (black_elo,
    white_elo) pair of EloRating objects per game (the same objects may appear in several games, e.g. one checkpoint against its
    predecessor G times).  Returns the list of stats dicts (the evaluation.csv columns, pipeline.py:851-866)."""
    res = play_eval_games_parallel(game, board_size, players, num_simulations, num_parallel, c_puct_base, c_puct_init, **kw)
    black_id = 1
    out = []
    for r, (be, we) in zip(res, elos):
        stats = {"game_length": r["game_length"], "game_result": r["game_result"]}
        if game == "go":
            stats["num_passes"] = r["num_passes"]
        if r["winner"] != 0:
            winner, loser = (be, we) if r["winner"] == black_id else (we, be)
            winner.update_rating(loser.rating, 1)
            loser.update_rating(winner.rating, 0)
        stats["black_elo_rating"], stats["white_elo_rating"] = be.rating, we.rating
        out.append(stats)
    return out, res


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
