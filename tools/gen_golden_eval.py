"""Golden vectors for SURVEY 8f-2/3 (evaluator game + Elo, resign-threshold controller) from the reference (this container only).

  * eval_against_prev_ckpt (core/pipeline.py:815-867) on 5x5 Go: black and white are the reference's own search entry points
    (mcts_v2.uct_search / parallel_uct_search, root_noise=False, deterministic=True, root_node=None every move) driven by two
    different synthetic evaluators (tests/synth_eval.py); three consecutive calls so that the Elo updates accumulate.
  * EloRating / get_k_factor (core/rating.py) on a grid of ratings and results.
  * maybe_adjust_resign_threshold (core/pipeline.py:656-670) on a grid.
Writes tests/golden/eval_arena.npz."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_harness  # noqa: E402
from synth_eval import make_eval_func  # noqa: E402

ref_harness.install(5)
from alpha_zero.core import pipeline as rp  # noqa: E402
from alpha_zero.core.mcts_v2 import parallel_uct_search, uct_search  # noqa: E402
from alpha_zero.core.rating import EloRating, get_k_factor  # noqa: E402
from alpha_zero.envs.go import GoEnv  # noqa: E402

out = {}
A = 26
for tag, P, sims in (("p1", 1, 40), ("p4", 4, 48)):
    def player(sharp, P=P, sims=sims):
        ev = make_eval_func(A, sharp)

        def act(env, root_node, c_puct_base, c_puct_init, warm_up=False):
            if P > 1:
                return parallel_uct_search(env=env, eval_func=ev, root_node=root_node, c_puct_base=c_puct_base, c_puct_init=c_puct_init,
                                           num_simulations=sims, num_parallel=P, root_noise=False, warm_up=warm_up, deterministic=True)
            return uct_search(env=env, eval_func=ev, root_node=root_node, c_puct_base=c_puct_base, c_puct_init=c_puct_init,
                              num_simulations=sims, root_noise=False, warm_up=warm_up, deterministic=True)

        return act

    env = GoEnv(komi=0.5, num_stack=8)
    black, white = player(2.0), player(1.3)
    be, we = EloRating(rating=0), EloRating(rating=0)
    rows = []
    for k in range(3):
        stats = rp.eval_against_prev_ckpt(env, black, white, be, we, 19652, 1.25)
        rows.append(stats)
        out[f"{tag}_moves_{k}"] = np.array(env.history, dtype=object if False else None) if False else np.array([m if m is not None else -9 for m in [h.move for h in env.history]], dtype=np.int64)
    out[f"{tag}_stats"] = json.dumps(rows)
    out[f"{tag}_cfg"] = json.dumps({"P": P, "sims": sims, "komi": 0.5, "sharp_black": 2.0, "sharp_white": 1.3})

grid = []
for ra in (-300.0, 0.0, 1500.0, 2099.0, 2100.0, 2250.0, 2399.9, 2400.0, 2600.0):
    for rb in (0.0, 2000.0, 2150.0, 2400.0, 2500.0):
        for score in (0, 1):
            e = EloRating(rating=ra)
            exp = e.expected_score(rb)
            e.update_rating(rb, score)
            grid.append((ra, rb, score, get_k_factor((ra, rb)), exp, e.rating))
out["elo_grid"] = np.array(grid, dtype=np.float64)
rg = []
for cur in (-0.95, -0.9, -0.85, -0.5, -0.9999):
    for rate in (0.0, 0.03, 0.05, 0.0501, 0.1, 0.25, 0.9):
        for target in (0.05, 0.1):
            rg.append((cur, rate, target, rp.maybe_adjust_resign_threshold(cur, rate, target)))
out["resign_grid"] = np.array(rg, dtype=np.float64)
dst = os.path.join(ROOT, "tests", "golden", "eval_arena.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, out["p1_stats"], out["p4_stats"], {k: v.shape for k, v in out.items() if hasattr(v, "shape")})
