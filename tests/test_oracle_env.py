"""Pins the CPU oracle's environments (oracle/rules.c) to the upstream reference.

The golden files were produced by replaying the same move lists through the
reference's own GoEnv / GomokuEnv (tools/gen_golden.py, development container).
Each game carries a 16-byte digest over (board, legal mask, ko, captures, turn,
steps, done, reward) of every position and a second one over every observation
tensor, plus the final Tromp-Taylor areas and result string.
"""
import os

import numpy as np
import pytest

from oracle.envs import OracleGoEnv, OracleGomokuEnv, replay_digests


def _check_go_file(path, n, stride=1):
    g = np.load(path)
    env = OracleGoEnv(n)
    off, mv = g["offsets"], g["moves"]
    bad = []
    for i in range(0, len(off) - 1, stride):
        m = mv[off[i]:off[i + 1]]
        k, ds, do = replay_digests(env, m)
        b, w = env.area_score()
        ok = (k == len(m) and ds == g["state_digest"][i].tobytes() and do == g["obs_digest"][i].tobytes()
              and (b, w) == tuple(g["areas"][i]) and env.position_result() == str(g["results"][i]))
        if not ok:
            bad.append(i)
    return bad, (len(off) - 1 + stride - 1) // stride


def test_go9_all_shipped_sgf_games_bit_exact(golden_dir):
    # 10,288 games / 631,692 positions (games/pro_games/go/9x9 + games/9x9_matches/crazystone_vs_az)
    bad, n = _check_go_file(os.path.join(golden_dir, "go9_sgf.npz"), 9)
    assert n == 10288 and not bad, f"{len(bad)} of {n} games differ, first {bad[:5]}"


def test_go9_sgf_full_position_dumps(golden_dir):
    g = np.load(os.path.join(golden_dir, "go9_sgf.npz"))
    env = OracleGoEnv(9)
    for gi in g["dump_games"]:
        m = g["moves"][g["offsets"][gi]:g["offsets"][gi + 1]]
        env.reset()
        for t, a in enumerate(m):
            env.step(int(a))
            assert np.array_equal(env.board.ravel(), g[f"dump{gi}_board"][t])
            assert np.array_equal(env.legal_actions.astype(np.int8), g[f"dump{gi}_legal"][t])
            assert env.ko == g[f"dump{gi}_ko"][t]
            assert env.caps == tuple(g[f"dump{gi}_caps"][t])


@pytest.mark.parametrize("n", [5, 9, 13, 19])
def test_go_random_playouts_bit_exact(golden_dir, n):
    bad, cnt = _check_go_file(os.path.join(golden_dir, f"go{n}_random.npz"), n)
    assert cnt > 0 and not bad


def test_go_reference_unit_test_sequences(golden_dir):
    """unit_tests/envs/go_test.py (19x19): suicide :80-112, ko :114-127, score :175-209, planes :236-276."""
    g = np.load(os.path.join(golden_dir, "go19_known.npz"))
    for name in g["names"]:
        name = str(name)
        env = OracleGoEnv(19)
        env.reset()
        reward, done = 0.0, False
        for a in g[f"{name}__moves"]:
            _, reward, done, _ = env.step(int(a))
        assert bool(g[f"{name}__done"]) == done and float(g[f"{name}__reward"]) == reward
        assert int(g[f"{name}__winner"]) == (env.winner or 0)
        assert np.array_equal(g[f"{name}__legal"], env.legal_actions.astype(np.int8))
        assert np.array_equal(g[f"{name}__board"], env.board.ravel())
        assert np.array_equal(g[f"{name}__obs"], env.observation())
        if f"{name}__probe" in g:
            # the reference test asserts ValueError('Illegal action') for the probed move
            assert int(g[f"{name}__probe_legal"]) == 0
            with pytest.raises(ValueError, match="Illegal action"):
                env.step(int(g[f"{name}__probe"]))
    # asserted outcomes of go_test.py:175-209
    assert int(g["score_black__winner"]) == 1 and float(g["score_black__reward"]) == 1.0
    assert int(g["score_white__winner"]) == -1 and float(g["score_white__reward"]) == 1.0


def test_go_score_boards(golden_dir):
    """others/go_score_system.py:100-236 -- expected = what the reference's area_score computes."""
    import ctypes

    from oracle.envs import lib

    g = np.load(os.path.join(golden_dir, "go9_score_boards.npz"))
    assert len(g["boards"]) == 7
    for brd, (eb, ew) in zip(g["boards"], g["areas"]):
        b, w = ctypes.c_int(0), ctypes.c_int(0)
        brd = np.ascontiguousarray(brd, dtype=np.int8)
        lib().oracle_go_area_score(brd.ctypes.data, 9, ctypes.byref(b), ctypes.byref(w))
        assert (b.value, w.value) == (eb, ew)


def test_go_errors_and_endings():
    """go_test.py:63-78 (invalid / occupied), :129-173 (resign, double pass, single pass, max_steps), :211-220."""
    env = OracleGoEnv(19)
    env.reset()
    for a in (500, 19 * 19 + 2, 999):
        with pytest.raises(ValueError, match="Invalid action"):
            env.step(a)
    env.step(0)
    with pytest.raises(ValueError, match="Illegal action"):
        env.step(0)
    env.reset()
    for i in range(4):
        env.step(i)
    _, r, d, _ = env.step(env.resign_move)
    assert d and r == -1 and env.winner == -1  # black resigned after 4 moves -> white wins
    with pytest.raises(RuntimeError, match="Game is over"):
        env.step(6)
    for steps, winner in ((6, -1), (9, 1)):
        env.reset()
        for i in range(steps):
            env.step(i)
        env.step(env.resign_move)
        assert env.winner == winner
    env.reset()
    for i in range(4):
        env.step(i)
        env.step(env.pass_move)
    assert env.steps == 8 and not env.is_game_over()
    env.step(env.pass_move)  # second consecutive pass
    assert env.is_game_over() and not env.legal_actions.any()
    for max_steps in (31, 101):
        env = OracleGoEnv(19, max_steps=max_steps)
        env.reset()
        for i in range(max_steps):
            env.step(i)
        with pytest.raises(RuntimeError, match="Game is over"):
            env.step(max_steps + 1)
    # empty-board double pass: W+7.5, reward +1 for white who passed last
    env = OracleGoEnv(9)
    env.reset()
    env.step(81)
    _, r, d, _ = env.step(81)
    assert d and r == 1.0 and env.winner == -1 and env.get_result_string() == "W+7.5"


def test_gomoku_playouts_and_winning_lines_bit_exact(golden_dir):
    g = np.load(os.path.join(golden_dir, "gomoku.npz"))
    off, mv = g["offsets"], g["moves"]
    assert len(off) - 1 == 536
    for i in range(len(off) - 1):
        size, ntw, win, rew = (int(x) for x in g["meta"][i])
        env = OracleGomokuEnv(size, ntw)
        m = mv[off[i]:off[i + 1]]
        k, ds, do = replay_digests(env, m)
        assert k == len(m) and ds == g["state_digest"][i].tobytes() and do == g["obs_digest"][i].tobytes()
        assert (env.winner or 0) == win and env.get_result_string() == str(g["results"][i])


def test_observation_planes_literal():
    """base_test.py:125-174: empty stack and the 8-move stacked layout, restated as literals."""
    env = OracleGomokuEnv(15)
    obs = env.reset()
    exp = np.zeros((17, 15, 15), np.int8)
    exp[16] = 1
    assert np.array_equal(obs, exp)
    black, white = [0, 1, 2, 3], [5, 6, 7, 8]
    for b, w in zip(black, white):
        env.step(b)
        obs, _, _, _ = env.step(w)
    # plane 2k = black (to move) stones k plies ago, 2k+1 = white; newest first
    hist_b = [4, 4, 3, 3, 2, 2, 1, 1]
    hist_w = [4, 3, 3, 2, 2, 1, 1, 0]
    exp = np.zeros((17, 15, 15), np.int8)
    for k in range(8):
        for a in black[:hist_b[k]]:
            exp[2 * k].flat[a] = 1
        for a in white[:hist_w[k]]:
            exp[2 * k + 1].flat[a] = 1
    exp[16] = 1
    assert np.array_equal(obs, exp)
