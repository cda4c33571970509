"""SURVEY 8f-2/3 checks shared by the CPU (host twin) and GPU tiers: evaluator game, Elo, resign-threshold controller."""
import json
import os

import numpy as np

import dropin_checks as dc
from synth_eval import make_eval_func


def check_arena(kind, golden_dir):
    from alpha_zero_amd.core.evaluate import EloRating, create_mcts_player, eval_against_prev_ckpt

    g = np.load(os.path.join(golden_dir, "eval_arena.npz"))
    for tag in ("p1", "p4"):
        cfg = json.loads(str(g[f"{tag}_cfg"]))
        want = json.loads(str(g[f"{tag}_stats"]))
        env = dc.make_env(kind, "go", 5, komi=cfg["komi"])
        black = create_mcts_player(num_simulations=cfg["sims"], num_parallel=cfg["P"], root_noise=False, deterministic=True,
                                   eval_func=make_eval_func(26, cfg["sharp_black"]))
        white = create_mcts_player(num_simulations=cfg["sims"], num_parallel=cfg["P"], root_noise=False, deterministic=True,
                                   eval_func=make_eval_func(26, cfg["sharp_white"]))
        be, we = EloRating(rating=0), EloRating(rating=0)
        for k in range(3):
            stats = eval_against_prev_ckpt(env, black, white, be, we, 19652, 1.25)
            assert stats == want[k], (tag, k, stats, want[k])  # bit-exact, floats included
            moves = [m if m is not None else -9 for m in [h.move for h in env.history]]
            assert moves == list(g[f"{tag}_moves_{k}"])


def check_parallel_arena(kind, golden_dir):
    """SURVEY 8f-2 "many in parallel": G evaluation games in lock-step on one engine reproduce the reference's games move for move
    and its Elo sequence (golden eval_arena.npz: three consecutive games per configuration), and a colour-swapped pairing equals
    the sequential drop-in path."""
    import engine_util as eu
    from alpha_zero_amd.core.evaluate import EloRating, create_mcts_player, eval_against_prev_ckpt, eval_many_against_prev_ckpt

    binding, dev = eu.backend(kind)
    g = np.load(os.path.join(golden_dir, "eval_arena.npz"))
    for tag in ("p1", "p4"):
        cfg = json.loads(str(g[f"{tag}_cfg"]))
        want = json.loads(str(g[f"{tag}_stats"]))
        strong, weak = make_eval_func(26, cfg["sharp_black"]), make_eval_func(26, cfg["sharp_white"])
        be, we = EloRating(rating=0), EloRating(rating=0)
        stats, res = eval_many_against_prev_ckpt("go", 5, [(strong, weak)] * 3, [(be, we)] * 3, cfg["sims"], cfg["P"], 19652, 1.25, komi=cfg["komi"],
                                                 binding=binding, device=dev)
        for k in range(3):
            assert stats[k] == want[k], (tag, k, stats[k], want[k])
            assert res[k]["moves"] == [int(m) for m in g[f"{tag}_moves_{k}"]]
        # mixed pairings in one batch (colours swapped in slot 1) vs the sequential drop-in evaluator game
        stats2, res2 = eval_many_against_prev_ckpt("go", 5, [(strong, weak), (weak, strong)], [(EloRating(0), EloRating(0)), (EloRating(0), EloRating(0))],
                                                   cfg["sims"], cfg["P"], 19652, 1.25, komi=cfg["komi"], binding=binding, device=dev)
        env = dc.make_env(kind, "go", 5, komi=cfg["komi"])
        mk = lambda sharp: create_mcts_player(num_simulations=cfg["sims"], num_parallel=cfg["P"], root_noise=False, deterministic=True,
                                               eval_func=make_eval_func(26, sharp))
        seq = eval_against_prev_ckpt(env, mk(cfg["sharp_white"]), mk(cfg["sharp_black"]), EloRating(0), EloRating(0), 19652, 1.25)
        assert stats2[1] == seq and res2[1]["moves"] == [h.move for h in env.history]
        assert res2[0]["moves"] == [int(m) for m in g[f"{tag}_moves_0"]]


class SynthDeviceEvaluator:
    """The synthetic evaluator behind the device route of play_eval_games_parallel (tensors in, tensors out)."""

    def __init__(self, A, sharp):
        import torch
        from synth_eval import eval_batch

        self.torch, self.eval_batch, self.A, self.sharp, self.calls = torch, eval_batch, A, sharp, 0

    def device_eval(self, x):
        self.calls += 1
        p, v = self.eval_batch(x.cpu().numpy().astype(np.int8), self.A, self.sharp)
        return self.torch.from_numpy(p).to(x.device), self.torch.from_numpy(v).to(x.device)


def check_device_route_arena(kind, golden_dir):
    """Evaluators with `device_eval` (leaf rows gathered / scattered by index on the engine's device, one batch per evaluator and
    round) play exactly the games of the host route -- the reference's golden games -- also mixed with a host evaluator in one batch."""
    import engine_util as eu
    from alpha_zero_amd.core.evaluate import play_eval_games_parallel

    binding, dev = eu.backend(kind)
    g = np.load(os.path.join(golden_dir, "eval_arena.npz"))
    for tag in ("p1", "p4"):
        cfg = json.loads(str(g[f"{tag}_cfg"]))
        strong_d, weak_d = SynthDeviceEvaluator(26, cfg["sharp_black"]), SynthDeviceEvaluator(26, cfg["sharp_white"])
        weak_h = make_eval_func(26, cfg["sharp_white"])
        res = play_eval_games_parallel("go", 5, [(strong_d, weak_d), (weak_d, strong_d), (strong_d, weak_h)], cfg["sims"], cfg["P"], 19652, 1.25,
                                       komi=cfg["komi"], binding=binding, device=dev)
        ref = play_eval_games_parallel("go", 5, [(make_eval_func(26, cfg["sharp_black"]), weak_h), (weak_h, make_eval_func(26, cfg["sharp_black"]))],
                                       cfg["sims"], cfg["P"], 19652, 1.25, komi=cfg["komi"], binding=binding, device=dev)
        assert res[0]["moves"] == [int(m) for m in g[f"{tag}_moves_0"]] == ref[0]["moves"]
        assert res[1] == ref[1] and res[2] == res[0]
        assert 0 < strong_d.calls  # one batch per evaluator and round, not one per game
        # random openings are played before the first search and reported with the moves
        op = [[12, 6], [12, 6]]
        r2 = play_eval_games_parallel("go", 5, [(strong_d, weak_d), (weak_d, strong_d)], cfg["sims"], cfg["P"], 19652, 1.25, komi=cfg["komi"],
                                      binding=binding, device=dev, openings=op)
        assert r2[0]["moves"][:2] == [12, 6] and r2[1]["moves"][:2] == [12, 6] and r2[0]["game_length"] == len(r2[0]["moves"])

        # an opening move the rules reject (the occupied point 12) raises instead of silently shifting the evaluator pairing by a ply
        try:
            play_eval_games_parallel("go", 5, [(strong_d, weak_d), (weak_d, strong_d)], cfg["sims"], cfg["P"], 19652, 1.25, komi=cfg["komi"],
                                     binding=binding, device=dev, openings=[[12, 6], [12, 12]])
        except ValueError as e:
            assert "opening move 1 of game 1" in str(e)
        else:
            raise AssertionError("an illegal opening move was accepted")

    class Boom:
        def __call__(self, *a, **k):
            raise KeyError("evaluator failed")

    try:  # an evaluator that raises must not leak the engine (try / finally around the game loop)
        play_eval_games_parallel("go", 5, [(Boom(), Boom())], 8, 1, 19652, 1.25, binding=binding, device=dev)
    except KeyError:
        pass
    else:
        raise AssertionError("the evaluator's exception was swallowed")


def check_elo_and_resign(golden_dir):
    from alpha_zero_amd.core.evaluate import EloRating, ResignController, get_k_factor, maybe_adjust_resign_threshold

    g = np.load(os.path.join(golden_dir, "eval_arena.npz"))
    for ra, rb, score, k, exp, new in g["elo_grid"]:
        e = EloRating(rating=ra)
        assert get_k_factor((ra, rb)) == k and e.expected_score(rb) == exp
        e.update_rating(rb, score)
        assert e.rating == new
    for cur, rate, target, want in g["resign_grid"]:
        assert maybe_adjust_resign_threshold(cur, rate, target) == want
    # bookkeeping walk-through of pipeline.py:519-553 with games_per_ckpt 400, disable ratio 0.1 -> adjust every 10 marked games
    rc = ResignController(-0.9, no_resign_games=100, reset_fp_interval=1000, games_per_ckpt=400, disable_resign_ratio=0.1, target_fp_rate=0.05)
    marked = {"is_resign_disabled": True, "is_marked_for_resign": True, "is_could_won": True}
    plain = {"game_length": 50}
    assert rc.threshold == -1 and rc.on_game(marked, 50) == -1 and rc.resign_count == 0  # warm-up: resignation off (pipeline.py:453-456)
    assert rc.on_game(marked, 100) == -0.9 and rc.resign_count == 0  # the hard reset at no_resign_games
    n = 100
    for i in range(9):
        n += 1
        assert rc.on_game(marked, n) == -0.9
    n += 1
    t = rc.on_game(marked, n)  # 10th marked game, all of them could have been won: fp rate 1.0
    assert t == maybe_adjust_resign_threshold(-0.9, 1.0, 0.05) and t < -0.9
    assert rc.on_game(plain, n + 1) == t
    assert rc.on_game(plain, 1000) == -0.9  # periodic reset
