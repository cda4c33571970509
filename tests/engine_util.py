"""Shared drivers for the engine parity tests.  The same functions run against
  * the host twin (tests/hosttwin, CPU, `-m "not gpu"`): identical engine source, WaveHost policy
  * libazsp.so on a MI355X (`-m gpu`): the product
and compare with the golden vectors produced by the reference / with the CPU oracle."""
import ctypes
import hashlib
import os
import subprocess

import numpy as np
import torch

from alpha_zero_amd import _abi
from alpha_zero_amd.core.engine import Engine, EngineConfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
_twin = None


_twin_variants = {}


def hosttwin_variant(tag, defines):
    """A host twin compiled with extra -D flags (e.g. a tiny AZ_PATH_CAP to exercise the deep-path fallback)."""
    if tag not in _twin_variants:
        src = os.path.join(HERE, "hosttwin", "azsp_host.cpp")
        so = os.path.join(HERE, "hosttwin", f"libazsp_hosttwin_{tag}.so")
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-strict-aliasing"] + defines + ["-o", so, src])
        _twin_variants[tag] = _abi.Binding(ctypes.CDLL(so), f"hosttwin_{tag}")
    return _twin_variants[tag]


def hosttwin_binding():
    """Builds (once) and loads the host twin -- test infrastructure, never used by the product."""
    global _twin
    if _twin is None:
        src = os.path.join(HERE, "hosttwin", "azsp_host.cpp")
        so = os.path.join(HERE, "hosttwin", "libazsp_hosttwin.so")
        deps = [src] + [os.path.join(ROOT, "alpha_zero_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "alpha_zero_amd", "csrc"))
                        if f.endswith(".h")] + [os.path.join(ROOT, "include", "azsp.h")]
        if not os.path.exists(so) or any(os.path.getmtime(d) > os.path.getmtime(so) for d in deps):
            subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-strict-aliasing", "-o", so, src])
        _twin = _abi.Binding(ctypes.CDLL(so), "hosttwin")
    return _twin


def gpu_binding():
    from alpha_zero_amd import _lib

    return _lib.load(require_gpu=True)


def backend(kind):
    """kind: 'host' | 'gpu' | 'host:cap2' -> (binding, torch device string)"""
    if kind == "host:cap2":
        return hosttwin_variant("cap2", ["-DAZ_PATH_CAP=2"]), "cpu"
    return (hosttwin_binding(), "cpu") if kind == "host" else (gpu_binding(), "cuda")


# ---------------------------------------------------------------------------------------------------
# environment replay: G games advance in lock-step through azsp_env_step
# ---------------------------------------------------------------------------------------------------
def replay_env_batch(kind, game, n, move_lists, num_to_win=5, max_steps=0, komi=7.5, want_obs=True):
    """Returns per game (played, state_digest16, obs_digest16, final scalars row)."""
    binding, dev = backend(kind)
    G = len(move_lists)
    eng = Engine(binding, EngineConfig(game=game, board_size=n, num_games=G, num_parallel=1, num_simulations=2, num_to_win=num_to_win,
                                       max_steps=max_steps, komi=komi, stop_after_move=True), device=dev)
    hs = [hashlib.sha256() for _ in range(G)]
    ho = [hashlib.sha256() for _ in range(G)]
    lens = np.array([len(m) for m in move_lists])
    T = int(lens.max()) if G else 0
    played = np.zeros(G, dtype=np.int64)
    alive = np.ones(G, dtype=bool)
    out = eng.env_step(None, want_obs=want_obs)

    def absorb(out, mask):
        sc = out["scalars"]
        rec = np.concatenate([
            out["board"].reshape(G, -1).view(np.uint8), out["legal"].view(np.uint8),
            np.ascontiguousarray(sc[:, [0, 1, 2, 3]].astype("<i2")).view(np.uint8).reshape(G, 8),
            np.ascontiguousarray(sc[:, [4, 5, 6]].astype(np.int8)).view(np.uint8)], axis=1)
        obs = out["obs"].reshape(G, -1) if want_obs else None
        for g in np.flatnonzero(mask):
            hs[g].update(rec[g].tobytes())
            if want_obs:
                ho[g].update(obs[g].tobytes())

    absorb(out, alive)
    final = out["scalars"].copy()
    for t in range(T):
        acts = np.full(G, -2, dtype=np.int32)
        for g in range(G):
            if alive[g] and t < lens[g]:
                acts[g] = move_lists[g][t]
            else:
                alive[g] = False
        if not alive.any():
            break
        out = eng.env_step(acts, want_obs=want_obs)
        ill = out["scalars"][:, 10] != 0
        stepped = alive & ~ill
        alive &= ~ill
        absorb(out, stepped)
        played += stepped
        final[stepped] = out["scalars"][stepped]
        alive &= out["scalars"][:, 5] == 0  # stop after the game ended
    eng.close()
    return [(int(played[g]), hs[g].digest()[:16], ho[g].digest()[:16], final[g]) for g in range(G)]


# ---------------------------------------------------------------------------------------------------
# search / actor replay against an MCTS golden file
# ---------------------------------------------------------------------------------------------------
def untile_features(flat, rows, n):
    """AZSP_FEAT_BF16_TILED tensor ([tile][4][3 n^2][8] bf16) -> int8 planes [rows, 17, n, n]; checks the encoding on the way
    (only 0.0 / 1.0, padding channels zero)."""
    NP = n * n
    tb = max(1, 256 // NP)
    one = 0x3C00 if flat.dtype == torch.float16 else 0x3F80  # AZSP_FEAT_F16_TILED / _BF16_TILED
    t = flat.view(torch.int16).cpu().numpy().reshape(-1, 4, tb * NP, 8)
    x = np.ascontiguousarray(t.transpose(0, 2, 1, 3)).reshape(-1, 32)[: rows * NP].reshape(rows, NP, 32)
    assert np.all((x == 0) | (x == one)) and not x[:, :, 17:].any()
    return np.ascontiguousarray((x[:, :, :17] == one).astype(np.int8).transpose(0, 2, 1)).reshape(rows, 17, n, n)


def unsplit_features(flat, rows, n):
    """AZSP_FEAT_F16_SPLIT tensor ([row][plane: hi, lo][4][n^2][8] f16, the fp32-class stem's input) -> int8 planes [rows, 17, n, n]; checks
    the encoding on the way: hi halves only 0.0 / 1.0, padding channels zero, the lo plane never written (all zero)."""
    NP = n * n
    t = flat.view(torch.int16).cpu().numpy()[: rows * 2 * 4 * NP * 8].reshape(rows, 2, 4, NP, 8)
    assert not t[:, 1].any(), "lo plane of 0 / 1 observation planes must stay zero"
    x = np.ascontiguousarray(t[:, 0].transpose(0, 2, 1, 3)).reshape(rows, NP, 32)
    assert np.all((x == 0) | (x == 0x3C00)) and not x[:, :, 17:].any()
    return np.ascontiguousarray((x[:, :, :17] == 0x3C00).astype(np.int8).transpose(0, 2, 1)).reshape(rows, 17, n, n)


def split_features(x):
    """[rows, 17, n, n] 0/1 planes -> the AZSP_FEAT_F16_SPLIT tensor (hi plane written, lo plane zero), the inverse of unsplit_features."""
    rows, _, n, _ = x.shape
    NP = n * n
    full = torch.zeros(rows, 2, 4, NP, 8, dtype=torch.float16)
    hi = torch.zeros(rows, NP, 32, dtype=torch.float16)
    hi[:, :, :17] = x.reshape(rows, 17, NP).permute(0, 2, 1).to(torch.float16)
    full[:, 0] = hi.view(rows, NP, 4, 8).permute(0, 2, 1, 3)
    return full.reshape(-1)


def tile_features(x, dtype=torch.bfloat16):
    """[rows, 17, n, n] 0/1 planes -> the AZSP_FEAT_BF16_TILED (or, dtype = float16, _F16_TILED) tensor, the inverse of untile_features."""
    rows, _, n, _ = x.shape
    NP = n * n
    tb = max(1, 256 // NP)
    ntiles = (rows + tb - 1) // tb
    full = torch.zeros(ntiles * tb * NP, 32, dtype=dtype)
    full[: rows * NP, :17] = x.reshape(rows, 17, NP).permute(0, 2, 1).reshape(rows * NP, 17).to(dtype)
    return full.view(ntiles, tb * NP, 4, 8).permute(0, 2, 1, 3).contiguous().reshape(-1)


def run_golden_selfplay(kind, G_gold, eval_batch, feature_dtype=_abi.FEAT_I8):
    """Runs the batched actor on the games of one golden file with the recorded randomness injected.
    Returns (engine logs per game, harvest tuple)."""
    g, cfg = G_gold.g, G_gold.cfg
    binding, dev = backend(kind)
    ngames = cfg["games"]
    idxs = [G_gold.moves_of_game(i) for i in range(ngames)]
    M = max(len(ix) for ix in idxs) + 1
    A = G_gold.A
    noise = np.zeros((ngames, M, A))
    unif = np.zeros((ngames, M, 16))
    for gi, ix in enumerate(idxs):
        noise[gi, : len(ix)] = g["noise"][ix]
        unif[gi, : len(ix)] = g["uniforms"][ix]
    ec = EngineConfig(
        game=cfg["game"], board_size=cfg["n"], num_games=ngames, num_parallel=cfg["parallel"], num_simulations=cfg["sims"],
        c_puct_base=cfg["c_puct_base"], c_puct_init=cfg["c_puct_init"], root_noise=cfg.get("root_noise", True),
        deterministic=cfg.get("deterministic", False), reuse_tree=cfg.get("reuse", True), warm_up_steps=cfg["warm_up_steps"],
        resign_threshold=cfg.get("resign_threshold", -1.0), check_resign_after_steps=cfg.get("check_resign_after_steps", 40),
        force_resign_disabled=1 if cfg.get("resign_disabled", True) else 0, inject_random=True, inject_moves=M,
        max_plies=cfg.get("max_moves") or 0, stop_at_game_end=True, feature_dtype=feature_dtype, log_moves=True, log_capacity=M)
    eng = Engine(binding, ec, device=dev)
    eng.set_injection(noise, unif)
    eng.reset_games()
    n_evals = np.zeros((ngames, M), dtype=np.int64)
    rounds = 0
    eng.select()
    while True:
        valid = eng.valid.cpu().numpy().astype(bool)
        st, _ = eng.status()
        if not valid.any() and np.all(st[:, 0] == _abi.ST_IDLE):
            break
        feats = (untile_features(eng.features, eng.rows, eng.N) if eng.features_tiled else
                 unsplit_features(eng.features, eng.rows, eng.N) if eng.features_split else eng.features.cpu().numpy())
        pri = np.zeros((eng.rows, A), dtype=np.float32)
        val = np.zeros(eng.rows, dtype=np.float32)
        rows = np.flatnonzero(valid)
        if len(rows):
            p, v = eval_batch(feats[rows], A)
            pri[rows], val[rows] = p, v
            for r in rows:
                gi = r // eng.P
                n_evals[gi, min(st[gi, 1], M - 1)] += 1
        eng.priors.copy_(torch.from_numpy(pri))
        eng.values.copy_(torch.from_numpy(val))
        eng.round()
        rounds += 1
        assert rounds < 200000
    logs = []
    st, _ = eng.status()
    for gi, ix in enumerate(idxs):
        per = []
        for k in range(len(ix)):
            pi, cn, q = eng.get_search(gi, k)
            per.append(dict(pi=pi, child_N=cn, root_q=q[0], child_q=q[1], move=int(q[3])))
        logs.append(per)
    parts, base = [], 0
    while True:  # the output window may be smaller than all finished games: harvest until drained
        states, pi, z, games = eng.harvest()
        if len(games) == 0:
            break
        games = games.copy()
        games[:, 0] += base
        base += len(z)
        parts.append((states.cpu().numpy().copy(), pi.cpu().numpy().copy(), z.cpu().numpy().copy(), games))
    if parts:
        hv = tuple(np.concatenate([p[i] for p in parts]) for i in range(4))
    else:
        hv = (np.zeros((0, 17, eng.N, eng.N), np.int8), np.zeros((0, A), np.float32), np.zeros(0, np.float32), np.zeros((0, 16), np.int32))
    counters = eng.counters()
    eng.close()
    return logs, hv, n_evals, counters


# ---------------------------------------------------------------------------------------------------
# engine vs CPU oracle on fresh seeded games (no golden file needed; used by -m gpu tests and smoke())
# ---------------------------------------------------------------------------------------------------
def oracle_selfplay(game, n, sims, P, ngames, seed, max_moves, eval_func_factory, warm_up_steps=4, resign_threshold=-1.0,
                    resign_disabled=True, check_resign_after_steps=40, max_steps=0):
    """Plays `ngames` games with the CPU oracle; returns (noise, uniforms, per-move logs, game results)."""
    from oracle import actor, mcts
    from oracle.envs import OracleGoEnv, OracleGomokuEnv

    rng = np.random.Generator(np.random.PCG64(seed))
    A = n * n + (1 if game == "go" else 0)
    M = max_moves + 1
    noise = rng.dirichlet(np.full(A, 0.03), size=(ngames, M))
    unif = rng.random((ngames, M, 16))
    logs, results = [], []
    for gi in range(ngames):
        env = OracleGoEnv(n, max_steps=max_steps) if game == "go" else OracleGomokuEnv(n)
        per = []

        def on_move(k, env_, move, pi, rq, cq, root):
            cn = root.N[root.parent[root.root]].copy() if root is not None else None
            per.append(dict(move=int(move), pi=np.asarray(pi, dtype=np.float64), root_q=float(rq), child_q=float(cq), child_N=cn))

        seq, stats = actor.play_one_game(
            env, eval_func_factory(A), num_simulations=sims, num_parallel=P, warm_up_steps=warm_up_steps,
            check_resign_after_steps=check_resign_after_steps, resign_threshold=resign_threshold, resign_disabled=resign_disabled,
            rand_for_move=lambda k: mcts.InjectedRand(noise[gi, k], unif[gi, k]), on_move=on_move, max_moves=max_moves)
        logs.append(per)
        results.append((seq, stats))
    return noise, unif, logs, results


def engine_selfplay_injected(kind, game, n, sims, P, noise, unif, max_moves, eval_batch, warm_up_steps=4, resign_threshold=-1.0,
                             resign_disabled=True, check_resign_after_steps=40, max_steps=0, feature_dtype=_abi.FEAT_I8):
    binding, dev = backend(kind)
    ngames, M, A = noise.shape
    ec = EngineConfig(game=game, board_size=n, num_games=ngames, num_parallel=P, num_simulations=sims, warm_up_steps=warm_up_steps,
                      resign_threshold=resign_threshold, check_resign_after_steps=check_resign_after_steps,
                      force_resign_disabled=1 if resign_disabled else 0, inject_random=True, inject_moves=M, max_plies=max_moves,
                      stop_at_game_end=True, feature_dtype=feature_dtype, log_moves=True, log_capacity=M, max_steps=max_steps)
    eng = Engine(binding, ec, device=dev)
    eng.set_injection(noise, unif)
    eng.reset_games()
    eng.select()
    rounds = 0
    while True:
        valid = eng.valid.cpu().numpy().astype(bool)
        st, _ = eng.status()
        if not valid.any() and np.all(st[:, 0] == _abi.ST_IDLE):
            break
        rows = np.flatnonzero(valid)
        pri = np.zeros((eng.rows, A), dtype=np.float32)
        val = np.zeros(eng.rows, dtype=np.float32)
        if len(rows):
            feats = eng.features[torch.as_tensor(rows, device=eng.features.device)].to(torch.float32).cpu().numpy().astype(np.int8)
            pri[rows], val[rows] = eval_batch(feats, A)
        eng.priors.copy_(torch.from_numpy(pri))
        eng.values.copy_(torch.from_numpy(val))
        eng.round()
        rounds += 1
        assert rounds < 500000
    st, _ = eng.status()
    logs = []
    for gi in range(ngames):
        per = []
        for k in range(min(int(st[gi, 1]), M)):
            pi, cn, q = eng.get_search(gi, k)
            per.append(dict(pi=pi, child_N=cn, root_q=q[0], child_q=q[1], move=int(q[3])))
        logs.append(per)
    states, pi, z, games = eng.harvest(sample_capacity=max(64, ngames * eng.geo.stage_capacity), max_games=2 * ngames)
    hv = (states.cpu().numpy().copy(), pi.cpu().numpy().copy(), z.cpu().numpy().copy(), games.copy())
    cnt = eng.counters()
    eng.close()
    return logs, hv, cnt


def compare_engine_with_oracle(kind, game, n, sims, P, ngames, seed, max_moves, **kw):
    """Bit-exact comparison of the engine with the oracle on seeded games; returns the engine counters."""
    from alpha_zero_amd.core.pipeline import game_stats_from_row
    from synth_eval import eval_batch, make_eval_func

    noise, unif, ologs, ores = oracle_selfplay(game, n, sims, P, ngames, seed, max_moves, lambda A: make_eval_func(A), **kw)
    elogs, (states, pis, zs, games), cnt = engine_selfplay_injected(kind, game, n, sims, P, noise, unif, max_moves, eval_batch, **kw)
    by_slot = {int(r[15]): r for r in games}
    for gi in range(ngames):
        assert len(elogs[gi]) == len(ologs[gi]), (gi, len(elogs[gi]), len(ologs[gi]))
        for k, (e, o) in enumerate(zip(elogs[gi], ologs[gi])):
            where = (game, n, gi, k)
            assert e["move"] == o["move"], where
            if o["child_N"] is not None:
                assert np.array_equal(e["child_N"], o["child_N"]), where
            assert e["root_q"] == o["root_q"] and e["child_q"] == o["child_q"], where
            if game == "go":
                assert np.array_equal(e["pi"], o["pi"]), where
            else:
                assert np.abs(e["pi"] - o["pi"]).max() <= 1e-6, where
        seq, stats = ores[gi]
        if seq is None:
            assert gi not in by_slot
            continue
        row = by_slot[gi]
        s0, ln = int(row[0]), int(row[1])
        assert ln == len(seq)
        assert np.array_equal(states[s0:s0 + ln], np.stack([t.state for t in seq]))
        assert np.array_equal(zs[s0:s0 + ln], np.array([t.value for t in seq], dtype=np.float32))
        est = game_stats_from_row(row, game=game, komi=7.5, resign_threshold=kw.get("resign_threshold", -1.0))
        assert est == stats, (est, stats)
    return cnt
