"""CPU-baseline calibration (SURVEY 8d last row): the reference-equivalent port that bench.py times on the GPU
box (oracle/baseline.py: oracle search + C rules) against the IMPORTED upstream reference, same process, one actor, one torch
thread, identical network, identical seeded openings and NumPy random state -- so both play the SAME moves (asserted) and only the
speed differs.  Writes tests/golden/cpu_baseline_calibration.json (ratio = port moves/s / reference moves/s), which bench.py reads
to report `cpu_baseline.reference_equivalent_value`.  Development container only: /root/reference does not exist on the GPU box.

Usage: python tools/calibrate_baseline.py [--moves 40] [--repeat 2]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
os.environ["OMP_NUM_THREADS"] = "1"
import ref_harness  # noqa: E402


def play(kind, n, sims, P, blocks, filters, moves, seed, opening):
    """One actor playing `moves` searched moves from the seeded opening; returns (move list, seconds)."""
    import torch

    torch.set_num_threads(1)
    from alpha_zero_amd.core.network import AlphaZeroNet  # same module tree / init as the reference network (state_dict compatible)

    torch.manual_seed(1)
    net = AlphaZeroNet((17, n, n), n * n + 1, blocks, filters, filters).eval()
    if kind == "reference":
        from alpha_zero.core import pipeline
        from alpha_zero.envs.go import GoEnv

        env = GoEnv()
        player = pipeline.create_mcts_player(network=net, device=torch.device("cpu"), num_simulations=sims, num_parallel=P,
                                             root_noise=True, deterministic=False)
    else:
        from oracle import mcts
        from oracle.envs import OracleGoEnv

        env = OracleGoEnv(n)

        @torch.no_grad()
        def eval_position(state, batched=False):  # oracle/baseline.py: the same evaluator as pipeline.py:91-123
            if not batched:
                state = state[None, ...]
            x = torch.from_numpy(state).to(dtype=torch.float32)
            logits, v = net(x)
            pi = torch.softmax(logits, dim=-1).cpu().numpy()
            v = np.squeeze(v.cpu().numpy(), axis=1).tolist()
            pi = [pi[i] for i in range(pi.shape[0])]
            return (pi, v) if batched else (pi[0], v[0])

        def player(env, root_node, c_puct_base, c_puct_init, warm_up):
            return mcts.parallel_uct_search(env=env, eval_func=eval_position, root_node=root_node, c_puct_base=c_puct_base,
                                            c_puct_init=c_puct_init, num_simulations=sims, num_parallel=P, root_noise=True, warm_up=warm_up)

    env.reset()
    for a in opening:
        env.step(int(a))
    np.random.seed(seed)
    out, root = [], None
    t0 = time.perf_counter()
    for _ in range(moves):
        if env.is_game_over():
            break
        mv, pi, rq, cq, root = player(env, root, 19652.0, 1.25, not (env.steps > 16))
        env.step(mv)
        out.append(int(mv))
    return out, time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--moves", type=int, default=40)
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--board", type=int, default=9)
    ap.add_argument("--sims", type=int, default=200)
    ap.add_argument("--parallel", type=int, default=8)
    ap.add_argument("--blocks", type=int, default=10)
    ap.add_argument("--filters", type=int, default=128)
    args = ap.parse_args()
    n = args.board
    ref_harness.install(n)
    rng = np.random.Generator(np.random.PCG64(99))
    runs = []
    for rep in range(args.repeat):
        # a random legal opening (no captures this early on an empty 9x9 board: distinct points)
        opening = rng.permutation(n * n)[: int(rng.integers(4, 20))]
        seed = 1000 + rep
        mr, tr = play("reference", n, args.sims, args.parallel, args.blocks, args.filters, args.moves, seed, opening)
        mp_, tp = play("port", n, args.sims, args.parallel, args.blocks, args.filters, args.moves, seed, opening)
        assert mr == mp_, f"port and reference diverged on rep {rep}: {mr} vs {mp_}"
        runs.append(dict(moves=len(mr), opening_plies=len(opening), reference_s=round(tr, 3), port_s=round(tp, 3),
                         reference_moves_per_s=round(len(mr) / tr, 4), port_moves_per_s=round(len(mp_) / tp, 4)))
        print(runs[-1], flush=True)
    ref_rate = sum(r["moves"] for r in runs) / sum(r["reference_s"] for r in runs)
    port_rate = sum(r["moves"] for r in runs) / sum(r["port_s"] for r in runs)
    key = f"go{n}_p{args.parallel}_s{args.sims}_{args.blocks}x{args.filters}"
    path = os.path.join(ROOT, "tests", "golden", "cpu_baseline_calibration.json")
    doc = json.load(open(path)) if os.path.exists(path) else {"ratios": {}}
    import platform

    import torch

    doc["what"] = ("port = oracle/baseline.py actor (oracle search + C rules), reference = /root/reference alpha_zero (create_mcts_player + GoEnv), "
                   "one actor, one torch thread, same network and NumPy random state; identical move lists asserted")
    doc["ratios"][key] = dict(ratio=round(port_rate / ref_rate, 4), reference_moves_per_s=round(ref_rate, 4), port_moves_per_s=round(port_rate, 4),
                              runs=runs, host=platform.processor() or platform.machine(), torch=torch.__version__, numpy=np.__version__)
    json.dump(doc, open(path, "w"), indent=1)
    print(json.dumps(doc["ratios"][key]))


if __name__ == "__main__":
    main()
