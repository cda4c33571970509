"""Generate golden vectors by IMPORTING the upstream reference (this container only).

Usage:  python tools/gen_golden.py <task> [...]
Tasks:
  go_sgf            replay every shipped 9x9 SGF through the reference GoEnv      -> tests/golden/go9_sgf.npz
  go_random N       seeded random-legal playouts on an NxN board                  -> tests/golden/goN_random.npz
  go_known          the known-answer sequences of unit_tests/envs/go_test.py (19x19) -> tests/golden/go19_known.npz
  go_score_boards   the 7 hand-made boards of others/go_score_system.py (9x9)     -> tests/golden/go9_score_boards.npz
  gomoku            random playouts + gomoku_test.py winning lines                -> tests/golden/gomoku.npz
  mcts ...          see tools/gen_golden_mcts.py
Only derived DATA (moves, digests, arrays) is written; no reference source is copied.
go_engine.py reads BOARD_SIZE at import time, so every board size runs in its own process.
"""
import glob
import os
import re
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_harness  # noqa: E402
from statehash import TrajectoryHasher, env_record  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
SGF_COLS = "abcdefghijklmnopqrs"


def go_record(env, done, reward):
    pos = env.position
    ko = -1 if pos.ko is None else pos.ko[0] * env.board_size + pos.ko[1]
    return env_record(env.board, env.legal_actions, ko, pos.caps, env.to_play, env.steps, done, reward)


def replay_go(env, moves, dump=None):
    """Replay `moves` on a reference GoEnv; returns (n_played, state_digest, obs_digest, black_area, white_area, result)."""
    from alpha_zero.envs import go_engine

    h = TrajectoryHasher()
    obs = env.reset()
    h.add(go_record(env, False, 0), obs)
    n = 0
    done = False
    for a in moves:
        if done or env.legal_actions[a] != 1:
            break
        obs, reward, done, _ = env.step(int(a))
        h.add(go_record(env, done, reward), obs)
        if dump is not None:
            pos = env.position
            dump["board"].append(np.array(env.board, dtype=np.int8).ravel().copy())
            dump["legal"].append(np.array(env.legal_actions, dtype=np.int8).copy())
            dump["ko"].append(-1 if pos.ko is None else pos.ko[0] * env.board_size + pos.ko[1])
            dump["caps"].append(tuple(pos.caps))
        n += 1
    b, w = go_engine.area_score(env.board)
    ds, do = h.digests()
    return n, ds, do, b, w, env.position.result_string()


def task_go_sgf():
    ref_harness.install(9)
    from alpha_zero.envs.go import GoEnv

    files = sorted(glob.glob(os.path.join(ref_harness.REF_ROOT, "games/pro_games/go/9x9/*.sgf")))
    files += sorted(glob.glob(os.path.join(ref_harness.REF_ROOT, "games/9x9_matches/crazystone_vs_az/*.sgf")))
    env = GoEnv()
    all_moves, offsets, dig_s, dig_o, areas, results = [], [0], [], [], [], []
    dumps = {}
    n_caps_moves = n_ko = n_pass = 0
    for gi, f in enumerate(files):
        txt = open(f, errors="ignore").read()
        # same move regex the reference uses to scan SGFs (core/eval_dataset.py:124)
        seq = re.findall(r";[BW]\[[a-z]{0,2}\]", txt)
        moves = []
        for m in seq:
            c = m[3:-1]
            if c == "" or c == "tt":
                moves.append(81)
            else:
                moves.append(SGF_COLS.index(c[1]) * 9 + SGF_COLS.index(c[0]))
        dump = {"board": [], "legal": [], "ko": [], "caps": []} if gi % 320 == 0 else None
        n, ds, do, b, w, res = replay_go(env, moves, dump)
        moves = moves[:n]
        if dump is not None:
            dumps[gi] = dump
        all_moves.extend(moves)
        offsets.append(len(all_moves))
        dig_s.append(np.frombuffer(ds, dtype=np.uint8))
        dig_o.append(np.frombuffer(do, dtype=np.uint8))
        areas.append((b, w))
        results.append(res)
        n_pass += sum(1 for a in moves if a == 81)
    out = dict(
        moves=np.array(all_moves, dtype=np.uint8),
        offsets=np.array(offsets, dtype=np.int32),
        state_digest=np.stack(dig_s),
        obs_digest=np.stack(dig_o),
        areas=np.array(areas, dtype=np.int16),
        results=np.array(results),
        dump_games=np.array(sorted(dumps), dtype=np.int32),
    )
    for gi, d in dumps.items():
        out[f"dump{gi}_board"] = np.stack(d["board"]) if d["board"] else np.zeros((0, 81), np.int8)
        out[f"dump{gi}_legal"] = np.stack(d["legal"]) if d["legal"] else np.zeros((0, 82), np.int8)
        out[f"dump{gi}_ko"] = np.array(d["ko"], dtype=np.int16)
        out[f"dump{gi}_caps"] = np.array(d["caps"], dtype=np.int16).reshape(-1, 2)
    np.savez_compressed(os.path.join(GOLD, "go9_sgf.npz"), **out)
    print(f"go_sgf: {len(files)} games, {len(all_moves)} moves, {n_pass} passes")


def random_go_game(env, rng, pass_prob):
    """Uniform random legal playout on the reference env; returns the move list."""
    env.reset()
    moves = []
    done = False
    while not done:
        legal = np.flatnonzero(np.asarray(env.legal_actions)[:-1])
        if len(legal) == 0 or rng.random() < pass_prob:
            a = env.pass_move
        else:
            a = int(legal[rng.integers(len(legal))])
        _, _, done, _ = env.step(a)
        moves.append(a)
    return moves


def task_go_random(n, games, seed=1234):
    ref_harness.install(n)
    from alpha_zero.envs.go import GoEnv

    rng = np.random.Generator(np.random.PCG64(seed + n))
    env = GoEnv()
    all_moves, offsets, dig_s, dig_o, areas, results, rewards = [], [0], [], [], [], [], []
    for g in range(games):
        # a third of the games never pass voluntarily (run to max_steps / forced passes), the rest end by double pass
        moves = random_go_game(env, rng, 0.0 if g % 3 == 0 else 0.03)
        k, ds, do, b, w, res = replay_go(env, moves)
        assert k == len(moves)
        all_moves.extend(moves)
        offsets.append(len(all_moves))
        dig_s.append(np.frombuffer(ds, dtype=np.uint8))
        dig_o.append(np.frombuffer(do, dtype=np.uint8))
        areas.append((b, w))
        results.append(res)
    np.savez_compressed(
        os.path.join(GOLD, f"go{n}_random.npz"),
        moves=np.array(all_moves, dtype=np.uint16),
        offsets=np.array(offsets, dtype=np.int32),
        state_digest=np.stack(dig_s),
        obs_digest=np.stack(dig_o),
        areas=np.array(areas, dtype=np.int16),
        results=np.array(results),
    )
    print(f"go_random {n}x{n}: {games} games, {len(all_moves)} moves")


def task_go_known():
    """Known-answer sequences asserted by unit_tests/envs/go_test.py (board 19x19, go_test.py:15-17).
    Stored as DATA: flat action lists + the property the reference test asserts + what the reference computes."""
    ref_harness.install(19)
    from alpha_zero.envs.go import GoEnv

    env = GoEnv()
    g = lambda s: 361 if s == "PASS" else env.gtp_to_action(s, check_illegal=False)  # noqa: E731
    cases = []

    def run(name, gtp_moves, probe=None):
        env.reset()
        acts = [g(m) for m in gtp_moves]
        rec = dict(name=name, moves=acts)
        reward, done = 0.0, False
        for a in acts:
            _, reward, done, _ = env.step(a)
        rec["done"], rec["reward"] = bool(done), float(reward)
        rec["winner"] = 0 if env.winner is None else int(env.winner)
        rec["legal"] = np.array(env.legal_actions, dtype=np.int8)
        rec["board"] = np.array(env.board, dtype=np.int8).ravel()
        rec["obs"] = np.array(env.observation(), dtype=np.int8)
        if probe is not None:
            rec["probe"] = g(probe)
            rec["probe_legal"] = int(env.legal_actions[g(probe)])
        cases.append(rec)

    # go_test.py:80-112 suicide, :114-127 ko, :175-209 score, :236-276 stacked planes
    run("suicide_B1", ["A3", "A2", "B2", "A1", "C1"], "B1")
    run("suicide_F4", ["D3", "A1", "D4", "A2", "D5", "A3", "E3", "A4", "E5", "A5", "F3", "A6", "F5", "E4", "G4"], "F4")
    ko = []
    for b, w in zip(["A4", "B4", "C3", "C1", "D2"], ["A2", "A3", "B1", "B3", "C2"]):
        ko += [b, w]
    run("ko_C2", ko + ["B2"], "C2")
    run("score_black", ["C1", "A1", "B2", "A2", "A3", "PASS", "PASS"])
    run("score_white", ["A1", "D2", "A2", "C3", "A3", "C4", "B1", "D5", "D3", "E4", "D4", "E3", "PASS", "PASS"])
    st = []
    for b, w in zip(["B2", "C3", "C1", "B3"], ["A3", "A1", "C2", "B1"]):
        st += [b, w]
    run("stacked_planes", st)
    out = {"names": np.array([c["name"] for c in cases])}
    for c in cases:
        for k, v in c.items():
            if k != "name":
                out[f"{c['name']}__{k}"] = np.asarray(v)
    np.savez_compressed(os.path.join(GOLD, "go19_known.npz"), **out)
    print("go_known:", [(c["name"], c.get("probe_legal"), c["winner"], c["reward"]) for c in cases])


def task_go_score_boards():
    """The 7 hand-made final boards in others/go_score_system.py:100-236; expected = what area_score COMPUTES."""
    ref_harness.install(9)
    from alpha_zero.envs import go_engine

    path = os.path.join(ref_harness.REF_ROOT, "others", "go_score_system.py")
    src = open(path).read()
    # The script defines helpers, then calls run_test_on_board(board, ...) once per hand-made board.
    # Execute the board literals with a recording stand-in for that function (DATA capture only).
    boards = []
    ns = {"np": np, "B": go_engine.BLACK, "W": go_engine.WHITE}
    ns["run_test_on_board"] = lambda board, *a: boards.append((f"game{len(boards) + 1}", np.array(board, dtype=np.int8)))
    exec(compile(src[src.index("# Game 1"):], path, "exec"), ns)
    names, arrs, areas = [], [], []
    for k, b in boards:
        bb, ww = go_engine.area_score(b)
        names.append(k)
        arrs.append(b)
        areas.append((bb, ww))
    np.savez_compressed(
        os.path.join(GOLD, "go9_score_boards.npz"),
        names=np.array(names),
        boards=np.stack(arrs),
        areas=np.array(areas, dtype=np.int16),
    )
    print("go_score_boards:", list(zip(names, areas)))


def gomoku_record(env, done, reward):
    return env_record(env.board, env.legal_actions, -1, (0, 0), env.to_play, env.steps, done, reward)


def task_gomoku(seed=99):
    ref_harness.install(9)
    from alpha_zero.envs.gomoku import GomokuEnv

    rng = np.random.Generator(np.random.PCG64(seed))
    out = {}
    specs = []  # (board_size, num_to_win, moves)
    for size, ntw, games in [(13, 5, 300), (7, 5, 60), (9, 5, 60), (15, 5, 60), (7, 3, 20), (7, 4, 20)]:
        for _ in range(games):
            env = GomokuEnv(board_size=size, num_to_win=ntw)
            env.reset()
            moves, done = [], False
            while not done:
                legal = np.flatnonzero(env.legal_actions)
                a = int(legal[rng.integers(len(legal))])
                _, _, done, _ = env.step(a)
                moves.append(a)
            specs.append((size, ntw, moves))
    # the 16 winning lines of gomoku_test.py:17-34 (7x7), winner plays the line, opponent plays seeded filler
    lines = [[0, 8, 16, 24, 32], [9, 17, 25, 33, 41], [6, 12, 18, 24, 30], [20, 26, 32, 38, 44],
             [0, 7, 14, 21, 28], [8, 15, 22, 29, 36], [0, 1, 2, 3, 4], [37, 38, 39, 40, 41]]
    for line in lines:
        for winner in (1, 2):
            env = GomokuEnv(board_size=7)
            env.reset()
            moves, done, k = [], False, 0
            while not done:
                if env.to_play == winner:
                    a = line[k]
                    k += 1
                else:
                    cand = sorted(set(np.flatnonzero(env.legal_actions).tolist()) - set(line))
                    a = int(cand[rng.integers(len(cand))])
                _, reward, done, _ = env.step(a)
                moves.append(a)
            assert env.winner == winner and reward == 1.0 and k == 5
            specs.append((7, 5, moves))
    all_moves, offsets, meta, dig_s, dig_o, results = [], [0], [], [], [], []
    for size, ntw, moves in specs:
        env = GomokuEnv(board_size=size, num_to_win=ntw)
        h = TrajectoryHasher()
        obs = env.reset()
        h.add(gomoku_record(env, False, 0), obs)
        reward = 0.0
        for a in moves:
            obs, reward, done, _ = env.step(a)
            h.add(gomoku_record(env, done, reward), obs)
        ds, do = h.digests()
        all_moves.extend(moves)
        offsets.append(len(all_moves))
        meta.append((size, ntw, 0 if env.winner is None else env.winner, int(reward)))
        dig_s.append(np.frombuffer(ds, dtype=np.uint8))
        dig_o.append(np.frombuffer(do, dtype=np.uint8))
        results.append(env.get_result_string())
    np.savez_compressed(
        os.path.join(GOLD, "gomoku.npz"),
        moves=np.array(all_moves, dtype=np.uint16),
        offsets=np.array(offsets, dtype=np.int32),
        meta=np.array(meta, dtype=np.int16),
        state_digest=np.stack(dig_s),
        obs_digest=np.stack(dig_o),
        results=np.array(results),
    )
    print(f"gomoku: {len(specs)} games, {len(all_moves)} moves")


if __name__ == "__main__":
    task = sys.argv[1]
    if task == "go_sgf":
        task_go_sgf()
    elif task == "go_random":
        task_go_random(int(sys.argv[2]), int(sys.argv[3]))
    elif task == "go_known":
        task_go_known()
    elif task == "go_score_boards":
        task_go_score_boards()
    elif task == "gomoku":
        task_gomoku()
    else:
        raise SystemExit(f"unknown task {task}")
