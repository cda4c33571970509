"""Pins the oracle's search + actor (oracle/mcts.py, oracle/actor.py) to the upstream reference.

Golden files hold, per searched move of complete self-play games driven through the reference's
own play_and_record_one_game / (parallel_)uct_search, the injected Dirichlet noise and sampling
uniforms and the outputs (move, search_pi, root_Q, best_child_Q, root child_N, evaluation counts),
and per game the (state, pi, z) samples and the stats dict.  Everything must match BIT-EXACTLY.
"""
import json

import numpy as np
import pytest

import golden_mcts
from oracle import actor, mcts
from oracle.envs import OracleGoEnv, OracleGomokuEnv
from synth_eval import make_eval_func


@pytest.mark.parametrize("name", golden_mcts.names())
def test_oracle_search_and_actor_match_reference(name):
    G = golden_mcts.MctsGolden(name)
    g, cfg = G.g, G.cfg
    env = OracleGoEnv(cfg["n"]) if cfg["game"] == "go" else OracleGomokuEnv(cfg["n"])
    for gi in range(cfg["games"]):
        idx = G.moves_of_game(gi)
        evlog = []
        ef = make_eval_func(G.A, log=evlog)
        seen = []

        def rand_for_move(k):
            i = idx[k]
            return mcts.InjectedRand(g["noise"][i], g["uniforms"][i][: g["n_uniforms"][i]])

        def on_move(k, env_, move, pi, rq, cq, root):
            i = idx[k]
            assert move == g["move"][i], (name, gi, k)
            assert str(np.asarray(pi).dtype) == str(g["pi_dtype"][i])
            assert np.array_equal(np.asarray(pi, dtype=np.float64), g["pi"][i])
            assert float(rq) == g["root_q"][i] and float(cq) == g["child_q"][i]
            assert sum(evlog) == g["n_evals"][i] and len(evlog) == g["n_eval_calls"][i]
            if root is not None:  # re-rooted tree: the searched root is the new root's parent
                parent = root.parent[root.root]
                assert np.array_equal(root.N[parent], g["child_N"][i])
            assert (root is not None) == bool(g["has_next"][i])
            evlog.clear()
            seen.append(k)

        seq, stats = actor.play_one_game(
            env, ef, num_simulations=cfg["sims"], num_parallel=cfg["parallel"], warm_up_steps=cfg["warm_up_steps"],
            check_resign_after_steps=cfg.get("check_resign_after_steps", 40), resign_threshold=cfg.get("resign_threshold", -1.0),
            resign_disabled=cfg.get("resign_disabled", True), root_noise=cfg.get("root_noise", True),
            deterministic=cfg.get("deterministic", False), reuse_tree=cfg.get("reuse", True), rand_for_move=rand_for_move,
            on_move=on_move, max_moves=cfg.get("max_moves"))
        assert len(seen) == len(idx)
        if G.finished(gi):
            st, pis, zs, gstats = G.samples(gi)
            assert np.array_equal(np.stack([t.state for t in seq]), st)
            assert np.array_equal(np.stack([np.asarray(t.pi_prob, dtype=np.float64) for t in seq]), pis)
            assert np.array_equal(np.array([t.value for t in seq]), zs)
            assert gstats == json.loads(json.dumps(stats))
        else:
            assert seq is None


def test_search_argument_errors():
    """mcts_v2.py:356-361"""
    env = OracleGoEnv(5)
    env.reset()
    ef = make_eval_func(26)
    with pytest.raises(ValueError):
        mcts.uct_search(env, ef, None, 19652.0, 1.25, num_simulations=0)
    env.step(25)
    env.step(25)
    with pytest.raises(RuntimeError, match="Game is over"):
        mcts.uct_search(env, ef, None, 19652.0, 1.25, num_simulations=10)
