"""Batched actor + sample gather, CPU tier (host twin engine, fp32 torch-CPU network, gloo)."""
import os
import subprocess
import sys

import numpy as np
import torch

import engine_util as eu
from alpha_zero_amd.core.network import AlphaZeroNet
from alpha_zero_amd.core.pipeline import SelfPlayActor


def _actor(game="go", n=5, G=8, sims=16, P=4, **kw):
    torch.manual_seed(1)
    A = n * n + (1 if game == "go" else 0)
    net = AlphaZeroNet((17, n, n), A, 1, 8, 8, gomoku=(game != "go"))
    return SelfPlayActor(net, game=game, board_size=n, num_games=G, num_simulations=sims, num_parallel=P, warm_up_steps=4,
                         device="cpu", net_dtype=torch.float32, use_graph=False, binding=eu.hosttwin_binding(), **kw)


def test_actor_emits_reference_shaped_games():
    a = _actor()
    got = []
    for _ in range(200):
        a.run_rounds(10)
        got += a.harvest()
        if len(got) >= 12:
            break
    assert len(got) >= 12
    for seq, stats in got:
        assert stats["game_length"] == len(seq) and 0 < len(seq) <= 50
        assert set(stats) == {"game_length", "game_result", "num_passes", "is_resign_disabled", "is_marked_for_resign", "is_could_won",
                              "marked_resign_player", "resign_threshold", "training_steps"}
        z = np.array([t.value for t in seq])
        black = np.array([t.state[16, 0, 0] for t in seq])
        assert seq[0].state.shape == (17, 5, 5) and seq[0].state.dtype == np.int8 and seq[0].pi_prob.shape == (26,)
        assert seq[0].pi_prob.dtype == np.float64 and abs(seq[0].pi_prob.sum() - 1) < 1e-5
        res = stats["game_result"]
        # z is +1 for the winner's samples, -1 for the loser's (pipeline.py:349-354)
        if res.startswith("B+"):
            assert np.all(z[black == 1] == 1) and np.all(z[black == 0] == -1)
        elif res.startswith("W+"):
            assert np.all(z[black == 1] == -1) and np.all(z[black == 0] == 1)
        # colours alternate and the first position is the empty board with black to move
        assert black[0] == 1 and not seq[0].state[:16].any()
    c = a.counters()
    assert c["games"] >= 12 and c["moves"] > 0 and c["stalls"] == 0


def test_actor_gomoku_and_weight_swap():
    a = _actor(game="gomoku", n=7, G=4, sims=12, P=2)
    a.run_rounds(30)
    net2 = AlphaZeroNet((17, 7, 7), 49, 1, 8, 8, gomoku=True)
    a.set_network(net2, training_steps=1000)
    got = []
    for _ in range(300):
        a.run_rounds(10)
        got += a.harvest()
        if len(got) >= 4:
            break
    assert len(got) >= 4
    for seq, stats in got:
        assert stats["training_steps"] == 1000 and set(stats) == {"game_length", "game_result", "training_steps"}
        assert seq[0].pi_prob.dtype == np.float32
        assert stats["game_result"] in ("B+1.0", "W+1.0", "DRAW")


def test_sample_gather_two_ranks_gloo(tmp_path):
    """N > 1 data path: finished-game tensors of every rank arrive on rank 0 in rank order (gloo, world_size 2)."""
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gather_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    procs = [subprocess.Popen([sys.executable, script, str(r), "2", str(tmp_path)], env=env) for r in range(2)]
    assert all(p.wait(timeout=300) == 0 for p in procs)
    out = np.load(os.path.join(str(tmp_path), "rank0.npz"))
    games = out["games"]
    assert len(games) >= 2 and out["states"].shape[0] == out["z"].shape[0] == out["pi"].shape[0] == games[:, 1].sum()
    # starts are rebased: games tile the sample range without overlap
    order = np.argsort(games[:, 0])
    assert games[order[0], 0] == 0 and np.all(games[order][1:, 0] == np.cumsum(games[order][:-1, 1]))
    assert set(np.unique(games[:, 15] >> 20)) == {0, 1}
    # each rank's samples are byte-identical to what that rank harvested locally
    for r in range(2):
        loc = np.load(os.path.join(str(tmp_path), f"local{r}.npz"))
        mine = games[(games[:, 15] >> 20) == r]
        assert len(mine) == len(loc["games"])
        for row, lrow in zip(mine[np.argsort(mine[:, 0])], loc["games"][np.argsort(loc["games"][:, 0])]):
            assert np.array_equal(out["states"][row[0]:row[0] + row[1]], loc["states"][lrow[0]:lrow[0] + lrow[1]])
            assert np.array_equal(out["z"][row[0]:row[0] + row[1]], loc["z"][lrow[0]:lrow[0] + lrow[1]])


def test_actor_loop_writes_reference_csv(tmp_path):
    """run_selfplay_actor_loop with logs_dir: actor{rank}.csv carries the reference's columns in its order
    (header of the reference's logs/go/9x9/actor0.csv), and the queue receives (game_seq, stats) with the same key order."""
    import csv
    import queue
    import threading

    from alpha_zero_amd.core.pipeline import run_selfplay_actor_loop
    from alpha_zero_amd.envs.go import GoEnv

    torch.manual_seed(1)
    net = AlphaZeroNet((17, 5, 5), 26, 1, 8, 8)
    env = GoEnv(board_size=5, _binding=eu.hosttwin_binding(), _device="cpu")
    q, stop = queue.Queue(), threading.Event()
    th = threading.Thread(target=run_selfplay_actor_loop, args=(
        3, 0, net, "cpu", q, env, 12, 4, 19652, 1.25, 4, 10, 0.1), kwargs=dict(
        logs_dir=str(tmp_path), save_sgf_dir=str(tmp_path), save_sgf_interval=1, stop_event=stop, num_games=8, net_dtype=torch.float32,
        harvest_every=20, binding=eu.hosttwin_binding()))
    th.start()
    first = q.get(timeout=120)
    stop.set()
    th.join(timeout=120)
    assert not th.is_alive()
    want = "datetime,game_length,game_result,num_passes,is_resign_disabled,is_marked_for_resign,is_could_won,marked_resign_player,resign_threshold,time_per_game,training_steps"
    rows = list(csv.reader(open(os.path.join(str(tmp_path), "actor0.csv"))))
    assert ",".join(rows[0]) == want and len(rows) >= 2 and len(rows[1]) == len(rows[0])
    assert list(first[1].keys()) == want.split(",")[1:]
    assert int(rows[1][1]) > 0 and rows[1][2][0] in "BWD"
    sgfs = sorted(f for f in os.listdir(str(tmp_path)) if f.endswith(".sgf"))  # pipeline.py:276-281: periodic SGF dumps
    assert sgfs and sgfs[0].startswith("actor0_")
    text = open(os.path.join(str(tmp_path), sgfs[0])).read()
    assert text.startswith("(;\nCA[UTF-8]\nAP[AlphaZeroMini_sgfgenerator]\nRU[Chinese]") and "SZ[5]" in text and text.endswith(")")


def check_harvested_moves(a):
    """azsp_harvest_moves: the move list of every finished self-play game, replayed through the CPU oracle env, reproduces each
    recorded sample state, the game length, the pass count and the result string."""
    from oracle.envs import OracleGoEnv

    got = []
    for _ in range(300):
        a.run_rounds(10)
        got += a.harvest(with_moves=True)
        if len(got) >= 14:
            break
    assert len(got) >= 14
    resigned = 0
    for seq, stats, moves in got:
        env = OracleGoEnv(5)
        obs = env.reset()
        for i, t in enumerate(seq):
            assert np.array_equal(t.state, obs)  # the sample is the observation before move i
            if i < len(moves):
                obs, _, done, _ = env.step(moves[i])
        if len(moves) == len(seq) - 1:  # the last sample's mover resigned: not a history move (base.py:224-226)
            resigned += 1
            env.step(env.resign_move)
        else:
            assert len(moves) == len(seq)
        assert env.is_game_over() and env.get_result_string() == stats["game_result"] and env.steps == stats["game_length"]
        assert sum(1 for m in moves if m == 25) == stats["num_passes"]
        sgf = a.game_sgf(stats, moves, date="d")
        assert sgf.count(";B[") + sgf.count(";W[") == len(moves) and f"RE[{stats['game_result']}]" in sgf and "SZ[5]" in sgf
    assert resigned >= 1  # the resignation path was exercised


def test_harvested_moves_replay_to_the_samples_and_the_result():
    check_harvested_moves(_actor(G=6, sims=12, P=4, resign_threshold=-0.3, check_resign_after_steps=4, disable_resign_ratio=0.5))
