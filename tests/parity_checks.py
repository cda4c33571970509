"""Assertions shared by the host-twin (CPU) and GPU parity tests."""
import json

import numpy as np

import engine_util as eu
import golden_mcts
from alpha_zero_amd.core.pipeline import game_stats_from_row
from synth_eval import eval_batch


def check_go_file(kind, path, n, chunk=2048):
    g = np.load(path)
    off, mv = g["offsets"], g["moves"]
    lists = [mv[off[i]:off[i + 1]].astype(np.int32) for i in range(len(off) - 1)]
    bad = []
    for c0 in range(0, len(lists), chunk):
        res = eu.replay_env_batch(kind, "go", n, lists[c0:c0 + chunk])
        for j, (k, ds, do, fin) in enumerate(res):
            i = c0 + j
            ok = (k == len(lists[i]) and ds == g["state_digest"][i].tobytes() and do == g["obs_digest"][i].tobytes()
                  and (fin[8], fin[9]) == tuple(g["areas"][i]))
            if not ok:
                bad.append(i)
    return bad, len(lists)


def check_gomoku_file(kind, path):
    g = np.load(path)
    off, mv = g["offsets"], g["moves"]
    groups = {}
    for i in range(len(off) - 1):
        groups.setdefault(tuple(int(x) for x in g["meta"][i][:2]), []).append(i)
    bad = []
    for (size, ntw), ids in groups.items():
        lists = [mv[off[i]:off[i + 1]].astype(np.int32) for i in ids]
        res = eu.replay_env_batch(kind, "gomoku", size, lists, num_to_win=ntw)
        for i, (k, ds, do, fin) in zip(ids, res):
            ok = (k == off[i + 1] - off[i] and ds == g["state_digest"][i].tobytes() and do == g["obs_digest"][i].tobytes()
                  and fin[7] == g["meta"][i][2] and fin[6] == g["meta"][i][3])
            if not ok:
                bad.append(i)
    return bad, len(off) - 1


def check_mcts_golden(kind, name, feature_dtype=None):
    """Batched actor with the recorded randomness injected vs the reference's outputs.
    Bit-exact: visit counts, chosen moves, root_Q, best_child_Q, evaluation counts, (Go) pi as float64,
    sample states / z / stats.  Gomoku pi (float32 in the reference, platform-dependent np.power) <= 1e-6."""
    G = golden_mcts.MctsGolden(name)
    g, cfg = G.g, G.cfg
    kw = {} if feature_dtype is None else {"feature_dtype": feature_dtype}
    logs, (states, pis, zs, games), n_evals, counters = eu.run_golden_selfplay(kind, G, eval_batch, **kw)
    for gi in range(cfg["games"]):
        ix = G.moves_of_game(gi)
        for k, i in enumerate(ix):
            L = logs[gi][k]
            where = (name, gi, k)
            assert np.array_equal(L["child_N"], g["child_N"][i]), where
            assert L["move"] == g["move"][i], where
            assert L["root_q"] == g["root_q"][i] and L["child_q"] == g["child_q"][i], where
            if cfg["game"] == "go":
                assert np.array_equal(L["pi"], g["pi"][i]), where
            else:
                assert np.abs(L["pi"] - g["pi"][i]).max() <= 1e-6, where
            assert n_evals[gi, k] == g["n_evals"][i], where
    by_slot = {int(row[15]): row for row in games}
    for gi in range(cfg["games"]):
        if not G.finished(gi):
            assert gi not in by_slot
            continue
        st, gp, gz, gstats = G.samples(gi)
        row = by_slot[gi]
        s0, ln = int(row[0]), int(row[1])
        assert ln == len(st)
        assert np.array_equal(states[s0:s0 + ln], st)
        assert np.array_equal(zs[s0:s0 + ln], gz.astype(np.float32))
        if cfg["game"] == "go":
            assert np.array_equal(pis[s0:s0 + ln], gp.astype(np.float32))
        else:
            assert np.abs(pis[s0:s0 + ln] - gp).max() <= 1e-6
        stats = game_stats_from_row(row, game=cfg["game"], komi=7.5, resign_threshold=cfg.get("resign_threshold", -1.0))
        assert json.loads(json.dumps(stats)) == gstats, (stats, gstats)
    return counters
