"""End-to-end search parity with a REAL network (VERDICT r2 Missing #3): the engine-backed `uct_search` / `parallel_uct_search`
drop-ins with the reference's shipped, trained 13x13 Gomoku checkpoint as evaluator, against goldens recorded from the reference's own
`create_mcts_player` + `mcts_v2` on that checkpoint (tools/gen_golden_mcts.py, configs gomoku13_ckpt200000_*; BASELINE C1 = uct_search,
100 simulations; mcts_v2.py:301-450, pipeline.py:91-123).  Shared by the CPU tier (host twin engine, fp32 torch-CPU evaluator: the very
kernels the golden run used, so everything must match exactly) and the GPU tier (fp32 and bf16 device evaluators: stated tolerances)."""
import os

import numpy as np
import torch

import dropin_checks as dc
import golden_mcts
from alpha_zero_amd.core.network import AlphaZeroNet

GOLDEN = golden_mcts.GOLDEN


def load_shipped():
    st = torch.load(os.path.join(GOLDEN, "gomoku13_ckpt200000_network.pt"), map_location="cpu", weights_only=True)
    net = AlphaZeroNet((17, 13, 13), 169, 10, 40, 80, gomoku=True)
    net.load_state_dict(st["network"], strict=True)
    return net.eval()


def module_eval_func(net, device="cpu"):
    """eval_func(obs, batched) around a torch module in fp32, with the reference wrapper's conventions (pipeline.py:91-123): float32
    softmax priors over all actions as a list of arrays (or one array), values as Python floats."""
    net = net.to(device).eval()

    @torch.no_grad()
    def eval_func(obs, batched=False):
        x = torch.from_numpy(obs if batched else obs[None, ...]).to(dtype=torch.float32, device=device)
        logits, v = net(x)
        pi = torch.softmax(logits, dim=-1).cpu().numpy()
        v = np.squeeze(v.cpu().numpy(), axis=1).tolist()
        pi = [pi[i] for i in range(x.shape[0])]
        return (pi, v) if batched else (pi[0], v[0])

    return eval_func


def inference_eval_func(inf, board_size, tiled):
    """eval_func around the product evaluator (InferenceNet): NCHW entry, or the engine's tiled feature layout -> hand-written kernels."""
    import engine_util as eu

    @torch.no_grad()
    def eval_func(obs, batched=False):
        x = torch.from_numpy(obs if batched else obs[None, ...]).float()
        if tiled:
            pri, v = inf.forward_tiled(eu.tile_features(x).cuda(), x.shape[0], board_size)
        else:
            pri, v = inf(x.cuda())
        pri, v = pri.float().cpu().numpy(), v.float().cpu().numpy().tolist()
        pi = [pri[i] for i in range(x.shape[0])]
        return (pi, v) if batched else (pi[0], v[0])

    return eval_func


def run_golden(kind, name, eval_func, teacher_forced):
    """Plays the golden's game through OUR uct_search with the golden's recorded Dirichlet draws and sampling uniforms.
    teacher_forced: the env follows the GOLDEN's moves and every search starts from a fresh root (use with a reuse=False golden): each
    move is then an independent comparison.  Otherwise the env follows our own moves with sub-tree reuse, and the comparison stops at
    the first move that differs.  Returns per-move records."""
    from alpha_zero_amd.core.mcts_v2 import parallel_uct_search, uct_search

    G = golden_mcts.MctsGolden(name)
    g, cfg = G.g, G.cfg
    env = dc.make_env(kind, cfg["game"], cfg["n"])
    real_dir, real_choice = np.random.dirichlet, np.random.choice
    recs = []
    total = 0
    try:
        for gi in range(cfg["games"]):
            env.reset()
            root = None
            idx = G.moves_of_game(gi)
            total += len(idx)
            for i in idx:
                assert np.array_equal(np.asarray(env.board, dtype=np.int8), g["board"][i]), "positions diverged"
                us = list(g["uniforms"][i][: g["n_uniforms"][i]])
                np.random.dirichlet = lambda alphas, _i=i: g["noise"][_i]

                def choice(a, p=None, _us=us):
                    cdf = np.asarray(p, dtype=np.float64).cumsum()
                    cdf /= cdf[-1]
                    return a[cdf.searchsorted(_us.pop(0) if _us else 0.5, side="right")]

                np.random.choice = choice
                reuse = cfg.get("reuse", True) and not teacher_forced
                kw = dict(env=env, eval_func=eval_func, root_node=root if reuse else None, c_puct_base=cfg["c_puct_base"], c_puct_init=cfg["c_puct_init"],
                          num_simulations=cfg["sims"], root_noise=True, warm_up=bool(g["warm_up"][i]), deterministic=False)
                if cfg["parallel"] > 1:
                    move, pi, rq, cq, root = parallel_uct_search(num_parallel=cfg["parallel"], **kw)
                else:
                    move, pi, rq, cq, root = uct_search(**kw)
                recs.append(dict(i=int(i), same_move=bool(move == g["move"][i]), dpi=float(np.abs(pi.astype(np.float64) - g["pi"][i]).max()),
                                 dq=float(abs(rq - g["root_q"][i])), dcq=float(abs(cq - g["child_q"][i])),
                                 top1=bool(int(np.argmax(pi)) == int(np.argmax(g["pi"][i]))), pi_dtype=str(pi.dtype) == str(g["pi_dtype"][i])))
                if not teacher_forced and move != g["move"][i]:
                    break  # the rest of this game is a different game; the next one starts from the empty board again
                env.step(int(g["move"][i]) if teacher_forced else int(move))
                if env.is_game_over():
                    break
    finally:
        np.random.dirichlet, np.random.choice = real_dir, real_choice
    return recs, total


def summarize(recs, total):
    n = len(recs)
    exact = sum(1 for r in recs if r["dpi"] <= 1e-6 and r["same_move"])
    return dict(moves_compared=n, moves_in_golden=int(total), exact_moves=exact, same_move=sum(r["same_move"] for r in recs) / n,
                top1=sum(r["top1"] for r in recs) / n, max_dpi=max(r["dpi"] for r in recs), mean_dpi=float(np.mean([r["dpi"] for r in recs])),
                max_dq=max(r["dq"] for r in recs), mean_dq=float(np.mean([r["dq"] for r in recs])), max_dcq=max(r["dcq"] for r in recs))


def check_exact(kind, name):
    """Same evaluator arithmetic as the golden run (fp32 torch on the CPU) -> everything equals the reference: moves, pi (float32 Gomoku
    pi: 1e-6), root_Q and best_child_Q exactly, through the whole game, with and without sub-tree reuse, uct_search and parallel."""
    ef = module_eval_func(load_shipped(), "cpu")
    recs, total = run_golden(kind, name, ef, teacher_forced=False)
    assert len(recs) == total and all(r["same_move"] and r["pi_dtype"] for r in recs), summarize(recs, total)
    # Q: exact where the evaluator runs on the CPU type that recorded the golden; another host CPU (the GPU box) may pick other fp32
    # convolution kernels inside torch -- last-bit differences in the values, the same visit counts
    assert max(r["dpi"] for r in recs) <= 1e-6 and max(r["dq"] for r in recs) <= 1e-6 and max(r["dcq"] for r in recs) <= 1e-6, summarize(recs, total)
    return summarize(recs, total)
