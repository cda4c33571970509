"""GPU tier (MI355X): the HIP build (libazsp.so) through the C ABI vs the reference's golden vectors
and vs the CPU oracle.  Integer / board work is bit-exact; search statistics are bit-exact under
injected randomness (visit counts, moves, Q, Go pi); Gomoku pi within 1e-6 (float32 np.power in the
reference is platform dependent)."""
import os

import numpy as np
import pytest

import engine_util as eu
import golden_mcts
import parity_checks as pc

pytestmark = pytest.mark.gpu


def test_gpu_go9_all_shipped_sgf_games(golden_dir):
    bad, n = pc.check_go_file("gpu", os.path.join(golden_dir, "go9_sgf.npz"), 9, chunk=4096)
    assert n == 10288 and not bad, bad[:5]


@pytest.mark.parametrize("n", [5, 9, 13, 19])
def test_gpu_go_random_playouts(golden_dir, n):
    bad, cnt = pc.check_go_file("gpu", os.path.join(golden_dir, f"go{n}_random.npz"), n)
    assert cnt > 0 and not bad


def test_gpu_gomoku_playouts_and_lines(golden_dir):
    bad, cnt = pc.check_gomoku_file("gpu", os.path.join(golden_dir, "gomoku.npz"))
    assert cnt == 536 and not bad


@pytest.mark.parametrize("name", golden_mcts.names())
def test_gpu_search_and_actor_match_reference(name):
    pc.check_mcts_golden("gpu", name)


@pytest.mark.parametrize("game,n,sims,P,ngames,moves,kw", [
    ("go", 9, 200, 8, 8, 30, {}),
    ("go", 5, 48, 4, 16, 60, dict(resign_threshold=-0.3, resign_disabled=False, check_resign_after_steps=5)),
    ("go", 13, 64, 8, 2, 16, {}),
    ("go", 19, 64, 8, 2, 10, {}),
    ("go", 19, 800, 8, 1, 3, {}),   # BASELINE C5's budget: 800 sims/move, P = 8 (node pool 832 records of 5.6 KB)
    ("gomoku", 13, 200, 8, 4, 24, {}),
    ("gomoku", 7, 40, 1, 8, 49, {}),
    ("gomoku", 15, 64, 8, 2, 20, {}),
])
def test_gpu_engine_matches_oracle_on_fresh_games(game, n, sims, P, ngames, moves, kw):
    cnt = eu.compare_engine_with_oracle("gpu", game, n, sims, P, ngames, seed=100 + n + sims, max_moves=moves, **kw)
    assert cnt["sims"] > 0


def test_gpu_env_vs_oracle_random_playouts_4096_games():
    """BASELINE size G = 4096: random legal playouts chosen from the engine's own legal masks, then the
    whole trajectories are replayed through the C oracle and compared position by position."""
    from alpha_zero_amd.core.engine import Engine, EngineConfig
    from oracle.envs import OracleGoEnv

    binding, dev = eu.backend("gpu")
    G, n = 4096, 9
    eng = Engine(binding, EngineConfig(game="go", board_size=n, num_games=G, num_parallel=1, num_simulations=2, stop_after_move=True), device=dev)
    rng = np.random.Generator(np.random.PCG64(7))
    out = eng.env_step(None)
    traj = []
    alive = np.ones(G, dtype=bool)
    for t in range(60):
        legal = out["legal"].astype(bool)
        legal[:, -1] = False
        acts = np.full(G, -2, dtype=np.int32)
        r = rng.random((G, legal.shape[1])) * legal
        pick = r.argmax(axis=1)
        has = legal.any(axis=1)
        acts[alive & has] = pick[alive & has]
        acts[alive & ~has] = n * n
        out = eng.env_step(acts)
        traj.append((acts.copy(), out["board"].copy(), out["legal"].copy(), out["scalars"].copy()))
        assert not out["scalars"][alive, 10].any()
        alive &= out["scalars"][:, 5] == 0
    eng.close()
    for g in range(0, G, 37):  # the oracle is the slow side: check every 37th game completely
        env = OracleGoEnv(n)
        env.reset()
        for acts, board, legal, sc in traj:
            if acts[g] == -2:
                break
            env.step(int(acts[g]))
            assert np.array_equal(env.board, board[g])
            assert np.array_equal(env.legal_actions.astype(np.int8), legal[g])
            assert env.ko == sc[g, 0] and env.caps == (sc[g, 1], sc[g, 2]) and env.steps == sc[g, 3]


# ---- Python boundary on the GPU: env classes, uct_search drop-ins, Dihedral-8 -------------------------
import dropin_checks as dc  # noqa: E402


def test_gpu_env_classes():
    dc.check_env_surface("gpu")


@pytest.mark.parametrize("name", ["go9_p8_s200", "go5_p8_s64", "gomoku13_p1_s100", "gomoku7_p8_s64", "go5_p1_s40_det"])
def test_gpu_uct_search_dropin_matches_reference(name):
    dc.check_dropin_search("gpu", name, max_moves=8)


@pytest.mark.parametrize("name", ["go9_p8_s200", "gomoku13_p1_s100", "go5_p1_s40_det"])
def test_gpu_uct_search_dropin_with_a_device_resident_evaluator_matches_reference(name):
    """The device-resident simulation loop (no host round trip per simulation; core/mcts_v2.py _simulate_on_device) on the reference's goldens."""
    dc.check_dropin_search("gpu", name, max_moves=8, device_route=True)


def test_gpu_search_errors():
    dc.check_search_errors("gpu")


def test_gpu_dropin_step_equals_the_separate_entries():
    """azsp_dropin_step on the device: the kernel reads the evaluator outputs from, and writes status / valid flags / observation planes
    to, page-locked host memory directly -- identical to select / expand_backup / get_status + copies, for three games at once."""
    dc.check_dropin_step_equals_the_separate_entries("gpu")


def test_gpu_dihedral():
    dc.check_dihedral("gpu")


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", ["bf16_tiled", "f16_split"])
@pytest.mark.parametrize("name", ["go9_p8_s200", "gomoku13_p8_s200"])
def test_gpu_tiled_feature_layout_matches_reference(name, fmt):
    """AZSP_FEAT_BF16_TILED / AZSP_FEAT_F16_SPLIT observation planes on the device: golden games replay bit-exactly through the decoded tensor."""
    from alpha_zero_amd import _abi

    pc.check_mcts_golden("gpu", name, feature_dtype=_abi.FEAT_BF16_TILED if fmt == "bf16_tiled" else _abi.FEAT_F16_SPLIT)


@pytest.mark.gpu
def test_gpu_eval_against_prev_ckpt_matches_reference(golden_dir):
    """the evaluator's game (two players, deterministic, fresh tree per move) + Elo, through the device engine."""
    import arena_checks as ac

    ac.check_arena("gpu", golden_dir)


@pytest.mark.gpu
def test_gpu_production_randomness_statistics():
    import rng_checks as rc

    rc.check_production_rng("gpu")


@pytest.mark.gpu
def test_gpu_harvested_moves_replay_to_the_samples_and_the_result():
    """azsp_harvest_moves on the device: move lists of finished self-play games replay (CPU oracle env) to every recorded sample
    state, the game length, the pass count and the result string, resignations included."""
    import torch

    import test_actor_host as tah
    from alpha_zero_amd import _lib
    from alpha_zero_amd.core.network import AlphaZeroNet
    from alpha_zero_amd.core.pipeline import SelfPlayActor

    torch.manual_seed(1)
    net = AlphaZeroNet((17, 5, 5), 26, 1, 8, 8)
    a = SelfPlayActor(net, game="go", board_size=5, num_games=6, num_simulations=12, num_parallel=4, warm_up_steps=4, device="cuda",
                      net_dtype=torch.float32, use_graph=False, binding=_lib.load(), resign_threshold=-0.3, check_resign_after_steps=4,
                      disable_resign_ratio=0.5)
    tah.check_harvested_moves(a)


# ---- the reference's known-answer edge cases through the HIP env kernels ------------------------
def test_gpu_go19_known_sequences(golden_dir):
    """unit_tests/envs/go_test.py:80-209: suicide x2, ko, scoring sequences, stacked planes -- via azsp_env_step on the device."""
    import edge_checks as ec

    ec.check_go19_known_sequences("gpu", golden_dir)


def test_gpu_go9_score_boards(golden_dir):
    """others/go_score_system.py:100-236: the 7 boards via azsp_set_state + a scoring step on the device."""
    import edge_checks as ec

    ec.check_go9_score_boards("gpu", golden_dir)


# ---- actor loop semantics on the device ----------------------------------------------------------
def _gpu_actor(**kw):
    import torch

    from alpha_zero_amd import _lib
    from alpha_zero_amd.core.network import AlphaZeroNet
    from alpha_zero_amd.core.pipeline import SelfPlayActor

    torch.manual_seed(1)
    net = AlphaZeroNet((17, 5, 5), 26, 1, 8, 8)
    args = dict(game="go", board_size=5, num_games=6, num_simulations=12, num_parallel=4, warm_up_steps=4, device="cuda",
                net_dtype=torch.float32, use_graph=False, binding=_lib.load())
    args.update(kw)
    return SelfPlayActor(net, **args)


def test_gpu_per_game_resign_threshold_and_training_steps():
    """azsp_set_actor_state on the device: every game plays with, and reports, the threshold / weights tag it STARTED with
    (pipeline.py:232-246)."""
    import torch

    import test_actor_host as tah
    from alpha_zero_amd.core.network import AlphaZeroNet

    a = _gpu_actor(resign_threshold=-1.0, check_resign_after_steps=4, disable_resign_ratio=0.5)
    torch.manual_seed(2)
    a._net = AlphaZeroNet((17, 5, 5), 26, 1, 8, 8)
    got = tah.check_per_game_actor_state(a, n_games=20)
    old = [s for _, s in got if s["resign_threshold"] == -1.0]
    new = [s for _, s in got if s["resign_threshold"] == -0.3]
    assert all(s["training_steps"] == 0 for s in old) and all(s["training_steps"] == 77 for s in new)
    assert a.straddled_games == 6


def test_gpu_run_selfplay_actor_loop(tmp_path):
    """run_selfplay_actor_loop on the device (pipeline.py:166-286): see test_actor_host.check_actor_loop."""
    import test_actor_host as tah
    from alpha_zero_amd import _lib

    tah.check_actor_loop("cuda", _lib.load(), tmp_path)


@pytest.mark.gpu
def test_gpu_parallel_evaluation_games_match_reference(golden_dir):
    """several evaluation games (two evaluators, deterministic, fresh tree per move) in lock-step on one device engine."""
    import arena_checks as ac

    ac.check_parallel_arena("gpu", golden_dir)
    ac.check_device_route_arena("gpu", golden_dir)


def _selfplay_properties(n, G, sims, blocks, filters, open_lo, open_hi, rounds, min_games, dtype=None, want=None):
    """Size-independent properties of everything the actor emits: every harvested game alternates colours, z is +1 for the winner's
    samples and -1 for the loser's, every pi is a distribution that never puts mass on an occupied point of its own position,
    lengths are within max_steps, counters are consistent, no stall / fault."""
    import torch

    from alpha_zero_amd import _lib
    from alpha_zero_amd.core.network import AlphaZeroNet
    from alpha_zero_amd.core.pipeline import SelfPlayActor

    torch.manual_seed(1)
    NP = n * n
    net = AlphaZeroNet((17, n, n), NP + 1, blocks, filters, filters)
    dtype = torch.bfloat16 if dtype is None else dtype
    a = SelfPlayActor(net, game="go", board_size=n, num_games=G, num_simulations=sims, num_parallel=8, device="cuda", net_dtype=dtype,
                      binding=_lib.load())
    assert "hand-written" in a.evaluator_path and (want is None or want in a.evaluator_path), a.evaluator_path
    assert a.tiled_features == (dtype != torch.float32)
    digests = []
    # short games so that many finish: long random openings, then the search plays them out
    rng = np.random.Generator(np.random.PCG64(5))
    plies = rng.integers(open_lo, open_hi + 1, size=G)
    out = a.engine.env_step(None)
    for t in range(int(plies.max())):
        legal = out["legal"][:, :NP].astype(bool)
        r = rng.random(legal.shape) * legal
        acts = np.where((plies > t) & legal.any(axis=1) & (out["scalars"][:, 5] == 0), r.argmax(axis=1), -2).astype(np.int32)
        out = a.engine.env_step(acts)
    games = samples = 0
    for _ in range(rounds // 50):
        a.run_rounds(50)
        st, pi, z, rows = a.harvest_tensors()
        if not len(rows):
            continue
        stc, pic, zc = st.cpu().numpy(), pi.cpu().numpy(), z.cpu().numpy()
        import hashlib

        digests.append(hashlib.sha1(stc.tobytes() + pic.tobytes() + zc.tobytes() + np.ascontiguousarray(rows).tobytes()).hexdigest())
        assert np.all(np.isin(zc, (-1.0, 0.0, 1.0))) and np.allclose(pic.sum(axis=1), 1.0, atol=1e-4)
        occupied = (stc[:, 0] + stc[:, 1]).reshape(len(stc), NP) > 0   # planes 0 / 1: the current position's stones
        assert not np.any((pic[:, :NP] > 0) & occupied)               # no visit on an occupied point
        for row in rows:
            s0, ln = int(row[0]), int(row[1])
            assert 0 < ln <= 2 * NP
            black = stc[s0:s0 + ln, 16, 0, 0]
            assert np.all(black[1:] != black[:-1])
            if int(row[2]) != 0:
                wb = 1 if int(row[2]) == 1 else 0
                zz = zc[s0:s0 + ln]
                assert np.all(zz[black == wb] == 1) and np.all(zz[black != wb] == -1)
        games += len(rows)
        samples += len(zc)
    c = a.counters()
    assert games >= min_games and samples > 10 * min_games and c["stalls"] == 0
    assert c["sims"] == c["leaves"] + c["terminal_hits"] and c["games"] >= games and c["moves"] > 0
    assert a.range_events == 0
    return digests


@pytest.mark.gpu
def test_gpu_full_size_selfplay_properties():
    """BASELINE size: 9x9 Go, G = 4096, 200 sims, P = 8, 10x128 bf16 on the hand-written evaluator, 600 rounds."""
    _selfplay_properties(9, 4096, 200, 10, 128, 100, 150, 600, 1000)


@pytest.mark.gpu
def test_gpu_full_size_selfplay_properties_fp32_class_evaluator():
    """The HEADLINE configuration (bench.py default): 9x9 Go, G = 4096, 200 sims, P = 8, 10x128 at the reference's precision class on the
    hand-written split-precision evaluator (hi + lo f16 pairs, three MFMA products), 600 rounds: every harvested game / sample property
    of the bf16 run above, no out-of-range activation (range record stays 0), and DETERMINISM at full size: a second actor with the
    same seed reproduces the harvest stream of the first 300 rounds bit for bit (search, evaluator under hipGraph replay, production
    randomness, harvest row assignment)."""
    import torch

    d600 = _selfplay_properties(9, 4096, 200, 10, 128, 100, 150, 600, 1000, dtype=torch.float32, want="split-precision")
    d300 = _selfplay_properties(9, 4096, 200, 10, 128, 100, 150, 300, 300, dtype=torch.float32, want="split-precision")
    assert len(d300) >= 4 and d600[: len(d300)] == d300


@pytest.mark.gpu
def test_gpu_gomoku13_selfplay_fp32_class_evaluator_same_seed_same_stream():
    """BASELINE C2 shape at the reference's precision: 13x13 Gomoku, 6 x 64, the hand-written split-precision evaluator on 17x17 planes
    (az_conv_sp17.h), G = 1024, 64 simulations: two actors with the same seed give the same harvest stream bit for bit; games end with a
    winner or a draw, z follows the winner, pi is a distribution on empty points."""
    import torch
    from alpha_zero_amd.core.network import AlphaZeroNet
    from alpha_zero_amd.core.pipeline import SelfPlayActor

    torch.manual_seed(4)
    net = AlphaZeroNet((17, 13, 13), 169, 6, 64, 64, gomoku=True)
    streams = []
    for _ in range(2):
        act = SelfPlayActor(net, game="gomoku", board_size=13, num_games=1024, num_simulations=64, num_parallel=8, warm_up_steps=8, seed=5, device="cuda",
                            engine_kw={"max_steps": 40})
        assert "split-precision" in act.evaluator_path and "hand-written" in act.evaluator_path, act.evaluator_path
        out = []
        for _ in range(8):  # 64 simulations / P = 8: ~9 rounds per move, games of <= 40 plies: most slots finish a game within 480 rounds
            act.run_rounds(60)
            st, pi, z, games = act.harvest_tensors(clone=True)
            out.append((st.cpu(), pi.cpu(), z.cpu(), games.copy()))
        assert act.range_events == 0
        streams.append((out, act.counters()))
        del act
    (a, ca), (b, cb) = streams
    assert ca == cb and sum(len(g) for *_, g in a) > 150
    for (s0, p0, z0, g0), (s1, p1, z1, g1) in zip(a, b):
        assert torch.equal(s0, s1) and torch.equal(p0, p1) and torch.equal(z0, z1) and np.array_equal(g0, g1)
        assert torch.allclose(p0.sum(1), torch.ones(len(p0)), atol=1e-4)
        occupied = (s0[:, 0] + s0[:, 1]).reshape(len(s0), 169) > 0
        assert not bool(((p0 > 0) & occupied).any())


@pytest.mark.gpu
def test_gpu_gomoku13_full_size_c2_properties_and_same_seed_stream():
    """BASELINE C2 at FULL size (VERDICT r4 missing #5): 13x13 Gomoku, G = 4096 concurrent games, 200 sims/move, P = 8, 6 x 64 at the
    reference's precision class on the hand-written split-precision evaluator (pad-3 stem -> 17x17 planes, az_conv_sp17.h).  Slots start
    from random openings of 24-34 plies and games are capped at 44 plies so that most slots finish a game within 450 rounds.  Every
    harvested sample is checked (z in {-1, 0, 1} and following the winner, pi a distribution on empty points, alternating colour plane),
    the evaluator never leaves its range, and a second actor with the same seed reproduces the harvest stream bit for bit."""
    import hashlib

    import torch
    from alpha_zero_amd.core.network import AlphaZeroNet
    from alpha_zero_amd.core.pipeline import SelfPlayActor

    torch.manual_seed(4)
    G, NP = 4096, 169
    net = AlphaZeroNet((17, 13, 13), NP, 6, 64, 64, gomoku=True)
    streams = []
    for rep in range(2):
        act = SelfPlayActor(net, game="gomoku", board_size=13, num_games=G, num_simulations=200, num_parallel=8, warm_up_steps=8, seed=5, device="cuda",
                            engine_kw={"max_steps": 44})
        assert "split-precision" in act.evaluator_path and "hand-written" in act.evaluator_path, act.evaluator_path
        assert "azsp_resblock_split" in act.evaluator_path  # the tower runs on the one-launch-per-block kernel (az_resblock_sp17.h)
        rng = np.random.Generator(np.random.PCG64(9))
        plies = rng.integers(24, 35, size=G)
        out = act.engine.env_step(None)
        for t in range(int(plies.max())):
            legal = out["legal"][:, :NP].astype(bool)
            r = rng.random(legal.shape) * legal
            acts = np.where((plies > t) & legal.any(axis=1) & (out["scalars"][:, 5] == 0), r.argmax(axis=1), -2).astype(np.int32)
            out = act.engine.env_step(acts)
        digests, games, samples = [], 0, 0
        for _ in range(9):
            act.run_rounds(50)
            st, pi, z, rows = act.harvest_tensors()
            if not len(rows):
                continue
            stc, pic, zc = st.cpu().numpy(), pi.cpu().numpy(), z.cpu().numpy()
            digests.append(hashlib.sha1(stc.tobytes() + pic.tobytes() + zc.tobytes() + np.ascontiguousarray(rows).tobytes()).hexdigest())
            if rep == 0:
                assert np.all(np.isin(zc, (-1.0, 0.0, 1.0))) and np.allclose(pic.sum(axis=1), 1.0, atol=1e-4)
                occupied = (stc[:, 0] + stc[:, 1]).reshape(len(stc), NP) > 0
                assert not np.any((pic > 0) & occupied)
                for row in rows:
                    s0, ln = int(row[0]), int(row[1])
                    assert 0 < ln <= 44
                    black = stc[s0:s0 + ln, 16, 0, 0]
                    assert np.all(black[1:] != black[:-1])
                    if int(row[2]) != 0:
                        wb = 1 if int(row[2]) == 1 else 0
                        zz = zc[s0:s0 + ln]
                        assert np.all(zz[black == wb] == 1) and np.all(zz[black != wb] == -1)
                    else:
                        assert np.all(zc[s0:s0 + ln] == 0)
            games += len(rows)
            samples += len(zc)
        c = act.counters()
        assert games >= 1000 and samples > 5 * games and c["stalls"] == 0 and c["sims"] == c["leaves"] + c["terminal_hits"]
        assert act.range_events == 0 and act.infer.split_range_status()[0] == 0 and act.infer.act_shift == 0
        streams.append((digests, c))
        del act
    assert streams[0] == streams[1]


@pytest.mark.gpu
def test_gpu_go19_selfplay_properties():
    """19x19 boards on the C5-shaped evaluator kernels (256 filters, fewer blocks and simulations so that games finish)."""
    _selfplay_properties(19, 256, 32, 2, 256, 560, 700, 400, 100)


@pytest.mark.gpu
def test_gpu_sgf_text_matches_reference(golden_dir, monkeypatch):
    """SURVEY 8f-4 on the device: env.to_sgf() of games replayed through the HIP env kernels equals the reference's SGF text byte for
    byte (Go incl. a resigned game, Gomoku, comments)."""
    import sgf_checks as sc

    sc.check_sgf("gpu", golden_dir, monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("game,n,filters", [("go", 9, 128), ("gomoku", 13, 64)])
def test_gpu_game_range_rounds_and_half_batch_forwards_equal_whole_batch_rounds(game, n, filters):
    """azsp_select_range / azsp_expand_backup_range + the evaluator on tile-aligned sub-batches: rounds run as two disjoint game ranges
    one after the other (each with its own half-batch forward) produce exactly the games of whole-batch rounds: same production
    randomness (Philox keyed by seed, slot, game, ply), same evaluations (a row's evaluation does not depend on the batch), so every
    harvested sample, game record and counter is bit-identical.  Games never interact inside a search (mcts_v2.py:568-625 runs per game)."""
    import numpy as np
    import torch
    from alpha_zero_amd.core.network import AlphaZeroNet
    from alpha_zero_amd.core.pipeline import SelfPlayActor

    A = n * n + (1 if game == "go" else 0)
    torch.manual_seed(4)
    net = AlphaZeroNet((17, n, n), A, 2, filters, 64, gomoku=(game != "go"))
    G, P = 1184, 8
    tb = max(1, 256 // (n * n))
    g_split = 576  # a multiple of 32 games whose first leaf row (576 * 8) starts a feature tile (a multiple of 3 boards at 9x9)
    assert g_split % 32 == 0 and (g_split * P) % tb == 0
    out = []
    for split in (False, True):
        act = SelfPlayActor(net, game=game, board_size=n, num_games=G, num_simulations=24, num_parallel=P, warm_up_steps=4, resign_threshold=-1.0,
                            seed=7, device="cuda", use_graph=False, net_dtype=torch.bfloat16, engine_kw={"max_steps": 24})
        e = act.engine
        games_by_uid = {}
        for _ in range(6):
            for _ in range(40):
                if not split:
                    act.run_round()
                    continue
                for k, (g0, g1) in enumerate(((g_split, G), (0, g_split))):  # second half first: the order must not matter
                    e.expand_backup(g0, g1)
                    e.select(g0, g1)
                    r0, r1 = g0 * P, g1 * P
                    feat = e.features[(r0 // tb) * (32 * tb * n * n):]
                    act.infer.forward_tiled(feat, r1 - r0, n, e.priors[r0:r1], e.values[r0:r1], slot=1 + k)
            # room for EVERY finished game (all slots start in phase here and finish in bursts): a harvest that runs out of room keeps
            # the remaining buffers for the next call, and which ones it keeps depends on the order the waves arrive
            st, pi, z, games = e.harvest(sample_capacity=2 * G * 25, max_games=2 * G)
            st, pi, z = st.cpu(), pi.cpu(), z.cpu()
            for row in games:  # the harvest kernel hands out output rows first come first served: key the games by their uid
                a, ln = int(row[0]), int(row[1])
                assert int(row[11]) not in games_by_uid
                games_by_uid[int(row[11])] = (st[a:a + ln].clone(), pi[a:a + ln].clone(), z[a:a + ln].clone(), row[1:].copy())
        out.append((games_by_uid, act.counters()))
        del act
    (ga, c0), (gb, c1) = out
    assert len(ga) >= 250 and sum(v[0].shape[0] for v in ga.values()) > 5000 and ga.keys() == gb.keys()
    for uid, (s0, p0, z0, r0) in ga.items():
        s1, p1, z1, r1 = gb[uid]
        assert torch.equal(s0, s1) and torch.equal(p0, p1) and torch.equal(z0, z1) and np.array_equal(r0, r1), uid
    assert {k: v for k, v in c0.items() if not k.startswith("hint")} == {k: v for k, v in c1.items() if not k.startswith("hint")}


@pytest.mark.gpu
@pytest.mark.parametrize("dtype_name", ["bfloat16", "float32"])
def test_gpu_same_seed_gives_the_same_harvest_stream(dtype_name):
    """Two actors with the same seed produce the same harvest stream -- the same games in the same output rows, bit for bit: search,
    evaluator (hand-written bf16 kernels / the fp32-class split-precision kernels, hipGraph replay), production randomness and the
    harvest's row assignment are all deterministic.
    (Round 2 reserved harvest rows with a CAS race; a missing barrier behind the LDS zero fill of the 9x9 convolution kernel could,
    rarely, perturb a forward -- both would show here.)"""
    import numpy as np
    import torch
    from alpha_zero_amd.core.network import AlphaZeroNet
    from alpha_zero_amd.core.pipeline import SelfPlayActor

    torch.manual_seed(4)
    net = AlphaZeroNet((17, 9, 9), 82, 2, 128, 64)
    streams = []
    for _ in range(2):
        act = SelfPlayActor(net, game="go", board_size=9, num_games=1024, num_simulations=24, num_parallel=8, warm_up_steps=4, resign_threshold=-1.0,
                            seed=11, device="cuda", net_dtype=getattr(torch, dtype_name), engine_kw={"max_steps": 30})
        assert "hand-written" in act.evaluator_path
        out = []
        for _ in range(8):
            act.run_rounds(50)
            st, pi, z, games = act.harvest_tensors(clone=True)
            out.append((st.cpu(), pi.cpu(), z.cpu(), games.copy()))
        streams.append((out, act.counters()))
        del act
    (a, ca), (b, cb) = streams
    assert ca == cb and sum(len(g) for *_, g in a) > 500
    for (s0, p0, z0, g0), (s1, p1, z1, g1) in zip(a, b):
        assert torch.equal(s0, s1) and torch.equal(p0, p1) and torch.equal(z0, z1) and np.array_equal(g0, g1)
