"""Golden vectors for the search + actor path, produced by running the upstream
reference (mcts_v2.uct_search / parallel_uct_search driven by
pipeline.play_and_record_one_game) in this container with
  * the deterministic synthetic evaluator of tests/synth_eval.py,
  * np.random.dirichlet recorded, np.random.choice replaced by its own published
    algorithm (searchsorted(cumsum(p)/cumsum(p)[-1], u, 'right'), verified equal to
    numpy's legacy implementation on 20,000 draws) so the uniforms can be recorded.
Per searched move the file stores: injected noise, consumed uniforms, the root's
child_N / child_W after the search, search_pi, move, root_Q, best_child_Q, and the
NN-call batch sizes; per game: the (state, pi, z) samples and the stats dict.

Usage: python tools/gen_golden_mcts.py <name>     (one process per config; BOARD_SIZE is import-time)
"""
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

GOLD = os.path.join(ROOT, "tests", "golden")

# name -> config
CONFIGS = {
    # 9x9 Go, the BASELINE workload shape: parallel search, P=8, 200 sims, full self-play games
    "go9_p8_s200": dict(game="go", n=9, sims=200, parallel=8, games=2, seed=11, resign_threshold=-1.0, resign_disabled=True, max_moves=60),
    # BASELINE C4's budget (400 sims/move, P = 8)
    "go9_p8_s400": dict(game="go", n=9, sims=400, parallel=8, games=1, seed=31, resign_threshold=-1.0, resign_disabled=True, max_moves=36),
    "go9_p1_s50": dict(game="go", n=9, sims=50, parallel=1, games=2, seed=12, resign_threshold=-1.0, resign_disabled=True, max_moves=40),
    # tiny boards: games finish, terminal nodes are reached inside the tree, passes, max_steps
    "go5_p8_s64": dict(game="go", n=5, sims=64, parallel=8, games=4, seed=13, resign_threshold=-1.0, resign_disabled=True),
    "go5_p1_s40": dict(game="go", n=5, sims=40, parallel=1, games=3, seed=14, resign_threshold=-1.0, resign_disabled=True),
    # resignation rule exercised (enabled and disabled-but-marked)
    "go5_p4_s48_resign": dict(game="go", n=5, sims=48, parallel=4, games=4, seed=15, resign_threshold=-0.2, resign_disabled=False,
                              check_resign_after_steps=6),
    "go5_p4_s48_marked": dict(game="go", n=5, sims=48, parallel=4, games=3, seed=16, resign_threshold=-0.2, resign_disabled=True,
                              check_resign_after_steps=6),
    # Gomoku: BASELINE C1 (uct_search 100 sims) and C2 shape (P=8, 200 sims)
    "gomoku13_p1_s100": dict(game="gomoku", n=13, sims=100, parallel=1, games=2, seed=21, max_moves=40),
    "gomoku13_p8_s200": dict(game="gomoku", n=13, sims=200, parallel=8, games=2, seed=22, max_moves=40),
    "gomoku7_p8_s64": dict(game="gomoku", n=7, sims=64, parallel=8, games=3, seed=23),
    "gomoku7_p1_s40": dict(game="gomoku", n=7, sims=40, parallel=1, games=3, seed=24),
    # REAL network: the reference's own create_mcts_player (pipeline.py:83-163: fp32 eval_position on the CPU) with the shipped, trained
    # checkpoint checkpoints/gomoku/13x13/training_steps_200000.ckpt = BASELINE C1 (13x13 Gomoku, uct_search, 100 sims).  With sub-tree
    # reuse (the actor's behaviour) and without (every move an independent search of the recorded position: a later move stays
    # comparable even when an earlier search differed by an arg-max flip between two fp32 implementations of the network).
    "gomoku13_ckpt200000_p1_s100": dict(game="gomoku", n=13, sims=100, parallel=1, games=6, seed=41, max_moves=44, ckpt="training_steps_200000.ckpt"),
    "gomoku13_ckpt200000_p1_s100_fresh": dict(game="gomoku", n=13, sims=100, parallel=1, games=6, seed=42, max_moves=44, reuse=False,
                                              ckpt="training_steps_200000.ckpt"),
    "gomoku13_ckpt200000_p8_s200": dict(game="gomoku", n=13, sims=200, parallel=8, games=5, seed=43, max_moves=40, ckpt="training_steps_200000.ckpt"),
    # deterministic, no-noise evaluation mode (pipeline.py:834: root_noise=False, deterministic=True, no reuse)
    "go5_p1_s40_det": dict(game="go", n=5, sims=40, parallel=1, games=2, seed=17, resign_threshold=-1.0, resign_disabled=True,
                           root_noise=False, deterministic=True, reuse=False),
}


class _NullLogger:
    def debug(self, *a, **k):
        pass


def main(name):
    cfg = CONFIGS[name]
    ref_harness.install(cfg["n"])
    from alpha_zero.core import mcts_v2, pipeline

    if cfg["game"] == "go":
        from alpha_zero.envs.go import GoEnv

        env = GoEnv()
    else:
        from alpha_zero.envs.gomoku import GomokuEnv

        env = GomokuEnv(board_size=cfg["n"])
    A = env.action_dim
    K = 16  # max uniforms recorded per move
    moves_log = []  # per searched move
    boards_log = []  # the position every search started from
    cur = {}
    eval_log = []
    eval_func = make_eval_func(A, log=eval_log)
    root_evals = []  # real network only: (prior, value) of the root position of every searched move
    if cfg.get("ckpt"):
        # the reference's own evaluator closure: create_mcts_player builds eval_position (pipeline.py:91-123) around its AlphaZeroNet;
        # its `act` is never used here -- the closure is picked up from the keyword it passes to uct_search
        import torch
        from alpha_zero.core.network import AlphaZeroNet

        torch.set_num_threads(1)
        st = torch.load(os.path.join(ref_harness.REF_ROOT, "checkpoints", "gomoku", "13x13", cfg["ckpt"]), map_location="cpu", weights_only=False)
        net = AlphaZeroNet((17, 13, 13), 169, num_res_block=10, num_filters=40, num_fc_units=80, gomoku=True)
        net.load_state_dict(st["network"])
        net.eval()
        grabbed = {}

        class _Grab(Exception):
            pass

        def grab(**kw):
            grabbed["eval_func"] = kw["eval_func"]
            raise _Grab()

        keep = (pipeline.uct_search, pipeline.parallel_uct_search)
        pipeline.uct_search = pipeline.parallel_uct_search = grab
        try:
            pipeline.create_mcts_player(net, torch.device("cpu"), cfg["sims"], cfg["parallel"], True, False)(env, None, 19652.0, 1.25, False)
        except _Grab:
            pass
        pipeline.uct_search, pipeline.parallel_uct_search = keep
        ref_eval = grabbed["eval_func"]

        def eval_func(obs, batched=False):  # the reference closure + a log of its batch sizes
            eval_log.append(int(obs.shape[0]) if batched else 1)
            return ref_eval(obs, batched)

    real_dirichlet = np.random.dirichlet

    def rec_dirichlet(alphas):
        out = real_dirichlet(alphas)
        cur["noise"] = np.array(out, dtype=np.float64)
        return out

    def rec_choice(a, p=None):
        u = np.random.random_sample()
        cdf = np.asarray(p, dtype=np.float64).cumsum()
        cdf /= cdf[-1]
        cur.setdefault("uniforms", []).append(u)
        return a[cdf.searchsorted(u, side="right")]

    real_gsp = mcts_v2.generate_search_policy

    def rec_gsp(child_N, temperature, legal_actions):
        cur["child_N"] = np.array(child_N, dtype=np.float32).copy()
        cur["legal"] = np.array(legal_actions, dtype=np.int8).copy()
        return real_gsp(child_N, temperature, legal_actions)

    np.random.dirichlet = rec_dirichlet
    np.random.choice = rec_choice
    mcts_v2.generate_search_policy = rec_gsp

    root_noise = cfg.get("root_noise", True)
    deterministic = cfg.get("deterministic", False)
    reuse = cfg.get("reuse", True)
    P = cfg["parallel"]
    max_moves = cfg.get("max_moves")

    class StopGame(Exception):
        pass

    state = {"moves_in_game": 0}

    def player(env, root_node, c_puct_base, c_puct_init, warm_up=False):
        if max_moves is not None and state["moves_in_game"] >= max_moves:
            raise StopGame()
        cur.clear()
        eval_log.clear()
        if not reuse:
            root_node = None
        if cfg.get("ckpt"):
            p0, v0 = ref_eval(env.observation(), False)
            root_evals.append((np.asarray(p0, dtype=np.float32).copy(), float(v0)))
        root_in = root_node
        n0 = float(root_in.N) if root_in is not None else 0.0
        kw = dict(env=env, eval_func=eval_func, root_node=root_node, c_puct_base=c_puct_base, c_puct_init=c_puct_init,
                  num_simulations=cfg["sims"], root_noise=root_noise, warm_up=warm_up, deterministic=deterministic)
        if P > 1:
            out = mcts_v2.parallel_uct_search(num_parallel=P, **kw)
        else:
            out = mcts_v2.uct_search(**kw)
        move, pi, root_q, child_q, nxt = out
        boards_log.append(np.array(env.board, dtype=np.int8).copy())
        moves_log.append(dict(
            game=state["game"], ply=env.steps, warm_up=int(warm_up), root_reused=int(root_in is not None), root_n0=n0,
            noise=cur.get("noise", np.zeros(A)), uniforms=list(cur.get("uniforms", [])), child_N=cur["child_N"],
            legal=cur["legal"], pi=np.array(pi, dtype=np.float64), pi_dtype=str(np.asarray(pi).dtype), move=int(move),
            root_q=float(root_q), child_q=float(child_q), has_next=int(nxt is not None), evals=list(eval_log),
            to_play=int(env.to_play),
        ))
        state["moves_in_game"] += 1
        return out

    games = []
    for g in range(cfg["games"]):
        pipeline.set_seed(cfg["seed"] + g)
        state["game"] = g
        state["moves_in_game"] = 0
        try:
            game_seq, stats = pipeline.play_and_record_one_game(
                env=env, mcts_player=player, resign_disabled=cfg.get("resign_disabled", True), c_puct_base=19652.0,
                c_puct_init=1.25, warm_up_steps=cfg.get("warm_up_steps", 4 if cfg["n"] <= 7 else 16),
                check_resign_after_steps=cfg.get("check_resign_after_steps", 40),
                resign_threshold=cfg.get("resign_threshold", -1.0), logger=_NullLogger())
            games.append(dict(
                finished=1, states=np.stack([t.state for t in game_seq]).astype(np.int8),
                pis=np.stack([np.asarray(t.pi_prob, dtype=np.float64) for t in game_seq]),
                zs=np.array([t.value for t in game_seq], dtype=np.float64), stats=stats))
        except StopGame:
            games.append(dict(finished=0, stats={}))

    out = {"config": np.array(json.dumps({**cfg, "warm_up_steps": cfg.get("warm_up_steps", 4 if cfg["n"] <= 7 else 16),
                                           "c_puct_base": 19652.0, "c_puct_init": 1.25, "num_actions": A}))}
    M = len(moves_log)
    out["game"] = np.array([m["game"] for m in moves_log], dtype=np.int32)
    out["ply"] = np.array([m["ply"] for m in moves_log], dtype=np.int32)
    out["warm_up"] = np.array([m["warm_up"] for m in moves_log], dtype=np.int8)
    out["root_reused"] = np.array([m["root_reused"] for m in moves_log], dtype=np.int8)
    out["root_n0"] = np.array([m["root_n0"] for m in moves_log], dtype=np.float64)
    out["noise"] = np.stack([m["noise"] for m in moves_log])
    un = np.zeros((M, K), dtype=np.float64)
    uc = np.zeros(M, dtype=np.int32)
    for i, m in enumerate(moves_log):
        assert len(m["uniforms"]) <= K
        uc[i] = len(m["uniforms"])
        un[i, : uc[i]] = m["uniforms"]
    out["uniforms"], out["n_uniforms"] = un, uc
    out["child_N"] = np.stack([m["child_N"] for m in moves_log])
    out["legal"] = np.stack([m["legal"] for m in moves_log])
    out["pi"] = np.stack([m["pi"] for m in moves_log])
    out["pi_dtype"] = np.array([m["pi_dtype"] for m in moves_log])
    out["move"] = np.array([m["move"] for m in moves_log], dtype=np.int32)
    out["root_q"] = np.array([m["root_q"] for m in moves_log], dtype=np.float64)
    out["child_q"] = np.array([m["child_q"] for m in moves_log], dtype=np.float64)
    out["has_next"] = np.array([m["has_next"] for m in moves_log], dtype=np.int8)
    out["to_play"] = np.array([m["to_play"] for m in moves_log], dtype=np.int8)
    out["n_evals"] = np.array([sum(m["evals"]) for m in moves_log], dtype=np.int32)
    out["n_eval_calls"] = np.array([len(m["evals"]) for m in moves_log], dtype=np.int32)
    if root_evals:
        out["root_prior"] = np.stack([r[0] for r in root_evals])
        out["root_value"] = np.array([r[1] for r in root_evals], dtype=np.float64)
        out["board"] = np.stack(boards_log).astype(np.int8)
    for g, gm in enumerate(games):
        out[f"g{g}_finished"] = np.array(gm["finished"])
        if gm["finished"]:
            out[f"g{g}_states"] = np.packbits(gm["states"].astype(np.uint8).reshape(len(gm["states"]), -1), axis=1)
            out[f"g{g}_pis"] = gm["pis"]
            out[f"g{g}_zs"] = gm["zs"]
            out[f"g{g}_stats"] = np.array(json.dumps({k: (v if not isinstance(v, (np.floating, np.integer)) else v.item())
                                                     for k, v in gm["stats"].items()}))
    np.savez_compressed(os.path.join(GOLD, f"mcts_{name}.npz"), **out)
    fin = [g["finished"] for g in games]
    print(f"{name}: {M} searched moves, games finished={fin}, "
          f"results={[g['stats'].get('game_result') for g in games]}, lens={[g['stats'].get('game_length') for g in games]}")


if __name__ == "__main__":
    main(sys.argv[1])
