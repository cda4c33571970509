"""bench.py -- self-play moves/sec of the batched engine (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W         (N>1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 9x9 Go, G = 4096
concurrent games per GPU, 200 sims/move with the reference budget semantics (root.N >= sims + P,
inherited visits count: mcts_v2.py:568), num_parallel P = 8, 10-block x 128-filter AlphaZeroNet
(random Kaiming init, torch.manual_seed(1)), Dirichlet root noise, sub-tree reuse, resign disabled.
One "step" = one engine round over all games: expand/backup of the previous G*P leaf batch, end-of-move
work, selection of the next P leaves per game, observation planes, and the network forward on G*P rows.
All inputs live in HBM; nothing crosses PCIe inside the timed region except the periodic harvest counts.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = {"bf16": 2500.0, "fp16": 2500.0, "fp32": 157.3}


def net_flops_per_eval(n, A, blocks, filters, fc, gomoku):
    """2*MAC of the convolutions and linear layers."""
    s = (n + 4) if gomoku else n
    f = 2 * 17 * filters * 9 * s * s + blocks * 2 * (2 * filters * filters * 9 * s * s)
    f += 2 * filters * 3 * s * s + 2 * (2 * s * s) * A + 2 * (s * s) * fc + 2 * fc
    return float(f)


def usable_host_cores():
    """Host threads this job may actually use: min(os.cpu_count(), cgroup CPU quota).  The GPU boxes expose 256
    threads but cap the container at 16 CPUs (cpu.max = "1600000 100000"); oversubscribing them only slows the
    baseline down (measured: 16 procs 54 moves/s, 64 procs 44, 256 procs 38)."""
    n = os.cpu_count() or 1
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=60)
    ap.add_argument("--games", type=int, default=4096)
    ap.add_argument("--board", type=int, default=9)
    ap.add_argument("--game", default="go")
    ap.add_argument("--sims", type=int, default=200)
    ap.add_argument("--parallel", type=int, default=8)
    ap.add_argument("--blocks", type=int, default=10)
    ap.add_argument("--filters", type=int, default=128)
    ap.add_argument("--net-dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--stagger", type=int, default=60, help="random opening plies per slot so game phases are mixed from the start")
    ap.add_argument("--harvest-every", type=int, default=50)
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--cpu-cores", type=int, default=0, help="0 = all usable host cores (cgroup quota aware)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-miopen-find", action="store_true", help="disable torch.backends.cudnn.benchmark (MIOpen find) for the convs")
    ap.add_argument("--split-round", action="store_true", help="diagnostic: launch expand/backup and select as two kernels and time each")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (MI355X); there is no CPU fallback for the product path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=dev)

    from alpha_zero_amd.core.gather import gather_samples
    from alpha_zero_amd.core.network import AlphaZeroNet
    from alpha_zero_amd.core.pipeline import SelfPlayActor

    n, game = args.board, args.game
    A = n * n + (1 if game == "go" else 0)
    torch.manual_seed(1)
    net = AlphaZeroNet((17, n, n), A, args.blocks, args.filters, args.filters, gomoku=(game != "go"))
    dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[args.net_dtype]
    torch.backends.cudnn.benchmark = not args.no_miopen_find
    actor = SelfPlayActor(net, game=game, board_size=n, num_games=args.games, num_simulations=args.sims, num_parallel=args.parallel,
                          warm_up_steps=16 if n <= 13 else 30, resign_threshold=-1.0, seed=1, rank=rank, device=dev, net_dtype=dt,
                          use_graph=not args.no_graph)
    eng = actor.engine

    if args.stagger > 0:  # mixed game phases from the first timed round (documented in DESIGN.md "Measurement")
        rng = np.random.Generator(np.random.PCG64(1234 + rank))
        plies = rng.integers(0, args.stagger + 1, size=args.games)
        out = eng.env_step(None)
        for t in range(int(plies.max())):
            legal = out["legal"][:, : n * n].astype(bool)
            r = rng.random(legal.shape) * legal
            acts = np.where((plies > t) & legal.any(axis=1) & (out["scalars"][:, 5] == 0), r.argmax(axis=1), -2).astype(np.int32)
            out = eng.env_step(acts)

    def barrier():
        if world > 1:
            import torch.distributed as dist

            dist.barrier()
        torch.cuda.synchronize(dev)

    def harvest_and_gather():
        st, pi, z, games = actor.harvest_tensors()
        res = gather_samples(st, pi, z, games, dst=0)
        return 0 if res is None else int(res[0].shape[0])

    for i in range(args.warmup):
        actor.run_round()
        if (i + 1) % args.harvest_every == 0:
            harvest_and_gather()
    barrier()
    harvest_and_gather()
    actor.counters(reset=True)
    evs = [tuple(torch.cuda.Event(enable_timing=True) for _ in range(4)) for _ in range(args.steps)]
    samples_at_root = 0
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        actor.run_round(evs[i])
        if (i + 1) % args.harvest_every == 0:
            samples_at_root += harvest_and_gather()
    barrier()
    elapsed = time.perf_counter() - t0
    cnt = actor.counters()
    bk_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in evs]))  # expand/backup + end-of-move kernels
    k_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in evs]))   # select kernel (the dominant hand-written kernel)
    nn_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in evs]))

    moves = float(cnt["moves"])
    tot = torch.tensor([moves, float(cnt["sims"]), float(cnt["leaves"])], dtype=torch.float64, device=dev)
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        import torch.distributed as dist

        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed_max = float(tmax.item())
    total_moves, total_sims, total_evals = (float(x) for x in tot.tolist())

    # ---- dominant kernel (the tower convolution): average launch duration, HIP events on the launch stream -----------
    conv = None
    if rank == 0 and getattr(actor.infer, "_tiled", None) is not None and actor.tiled_features:
        import ctypes

        inf, dll = actor.infer, actor.binding.dll
        a, m, o = inf._tiled  # real activations of the last forward
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rows = eng.rows
        S_t = n + 2 * (inf.stem_pad - 1)  # tower planes (17x17 behind the Gomoku pad-3 stem)
        reps = 5
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2 * reps * inf.n_blocks + 1)]
        torch.cuda.synchronize(dev)
        k = 0
        ev[0].record()
        for _ in range(reps):
            for i in range(inf.n_blocks):  # the forward's own launch sequence, one event after every launch
                assert dll.azsp_conv3x3_tiled(a.data_ptr(), inf.wp[2 * i].data_ptr(), inf.b32[2 * i].data_ptr(), None, m.data_ptr(), rows, S_t, args.filters, 1, st) == 0
                k += 1
                ev[k].record()
                assert dll.azsp_conv3x3_tiled(m.data_ptr(), inf.wp[2 * i + 1].data_ptr(), inf.b32[2 * i + 1].data_ptr(), a.data_ptr(), o.data_ptr(), rows,
                                              S_t, args.filters, 1, st) == 0
                k += 1
                ev[k].record()
                a, o = o, a
        torch.cuda.synchronize(dev)
        d = [ev[j].elapsed_time(ev[j + 1]) for j in range(k)]
        conv = {"launches": k, "avg_ms": float(np.mean(d)), "avg_ms_plain": float(np.mean(d[0::2])), "avg_ms_residual": float(np.mean(d[1::2])),
                "planes": S_t}

    if args.split_round and rank == 0:
        ea = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ta = tb = 0.0
        for _ in range(20):
            ea[0].record()
            eng.expand_backup()
            ea[1].record()
            eng.select()
            ea[2].record()
            actor._graph.replay() if actor._graph is not None else actor._forward()
            torch.cuda.synchronize(dev)
            ta += ea[0].elapsed_time(ea[1]) / 20
            tb += ea[1].elapsed_time(ea[2]) / 20
        print(json.dumps({"split_round_ms": {"expand_backup_endmove": round(ta, 4), "select_features": round(tb, 4)}}), flush=True)
    if rank == 0:
        # ---- roofline of the dominant hand-written kernel: the fused round kernel (HBM bound) ----------
        e_bytes = {"bf16": 2, "fp16": 2, "fp32": 4}[args.net_dtype]
        W = (n * n + 63) // 64
        steps = max(1, args.steps)
        expanded = cnt["leaves"] - cnt["dup_leaves"] + cnt["root_evals"]
        created = cnt["nodes_created"]
        vloss_edges = cnt["backup_edges"] - (cnt["sims"] - cnt["leaves"] + expanded)  # informational
        # select kernel, reference-equivalent dense layout: N, W, P rows of every visited node, the new
        # node's position, virtual loss read-modify-write of W per path edge, the leaf's 17 planes + 8-board history
        # observation planes per leaf: 17 planes in the network dtype, or (tiled evaluator layout) 3 written 8-channel chunks
        feat_bytes = 3 * n * n * 16 if actor.tiled_features else 17 * n * n * e_bytes
        alg_bytes = (cnt["node_visits"] * 12 * A + created * 64 + cnt["leaves"] * (cnt["backup_edges"] / max(1, cnt["sims"])) * 8
                     + (cnt["leaves"] + cnt["root_evals"]) * (feat_bytes + 16 * W * 8)) / steps
        # expand/backup kernel: prior + value in, P/N/W rows out, N and W read-modify-write per path edge (+ vloss revert)
        bk_bytes = (expanded * (12 * A + 4 * A + 4) + cnt["backup_edges"] * 16 + cnt["leaves"] * 8 * 4.5) / steps
        achieved = alg_bytes / (k_ms * 1e-3) / 1e9
        traffic = None
        prof = os.path.join(ROOT, "profiles", "select_kernel_pmc.json")
        if os.path.exists(prof):
            try:
                pj = json.load(open(prof))
                if pj.get("games") == args.games and pj.get("board") == n:
                    traffic = pj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        engine_roof = {"kernel": "k_game<OpSelect> (PUCT descents + virtual loss + observation planes)", "bound": "hbm",
                       "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                       "traffic": traffic, "alg_bytes_per_launch": round(alg_bytes), "avg_launch_ms": round(k_ms, 4),
                       "share_of_step": round(k_ms / (elapsed_max / steps * 1e3), 4),
                       "backup_kernels": {"avg_ms": round(bk_ms, 4), "alg_bytes_per_launch": round(bk_bytes),
                                          "achieved_GBs": round(bk_bytes / (bk_ms * 1e-3) / 1e9, 2)}}
        flops_eval = net_flops_per_eval(n, A, args.blocks, args.filters, args.filters, game != "go")
        nn_tflops = flops_eval * args.games * args.parallel / (nn_ms * 1e-3) / 1e12
        peak = MFMA_PEAK_TFLOPS[args.net_dtype]
        nn_roof = {"kernel": "whole evaluator forward on G*P rows (stem + tower + heads)", "bound": "mfma",
                   "achieved": round(nn_tflops, 2), "peak": peak, "unit": "TFLOP/s",
                   "frac": round(nn_tflops / peak, 5), "avg_forward_ms": round(nn_ms, 3),
                   "share_of_step": round(nn_ms / (elapsed_max / steps * 1e3), 4),
                   "batch_fill": round((cnt["leaves"] + cnt["root_evals"]) / (steps * args.games * args.parallel), 4)}
        if conv is not None:
            # the step's dominant kernel: 2 * blocks launches per forward.  Algorithmic work per launch = the dense 3x3
            # convolution (padding taps counted, the usual convention): 2 * rows * N^2 * C * C * 9 flop.
            rows = args.games * args.parallel
            S_t = conv["planes"]
            conv_flops = 2.0 * rows * S_t * S_t * args.filters * args.filters * 9
            tf = conv_flops / (conv["avg_ms"] * 1e-3) / 1e12
            ctraffic = None
            cprof = os.path.join(ROOT, "profiles", "conv_kernel_pmc.json" if S_t == 9 else "conv64_kernel_pmc.json")
            if os.path.exists(cprof):
                try:
                    pj = json.load(open(cprof))
                    if pj.get("rows") == rows and pj.get("board") == n and pj.get("channels") == args.filters:
                        ctraffic = pj.get("hbm_bytes_per_launch")
                except Exception:
                    ctraffic = None
            roofline = {"kernel": ("k_conv3x3_tiled" if S_t == 9 else "k_conv3x3_t64") + " (weight-stationary MFMA 3x3 convolution of the residual tower, bf16)",
                        "bound": "mfma",
                        "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(tf / peak, 5), "traffic": ctraffic,
                        "alg_flops_per_launch": conv_flops, "alg_hbm_bytes_per_launch": round(rows * S_t * S_t * args.filters * 2 * 2.5),
                        "avg_launch_ms": round(conv["avg_ms"], 4), "avg_launch_ms_plain": round(conv["avg_ms_plain"], 4),
                        "avg_launch_ms_residual": round(conv["avg_ms_residual"], 4), "launches_per_step": 2 * args.blocks,
                        "share_of_step": round(2 * args.blocks * conv["avg_ms"] / (elapsed_max / steps * 1e3), 4)}
        else:
            roofline = engine_roof
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            from oracle import baseline

            cores = args.cpu_cores or usable_host_cores()
            cpu = baseline.run(cores, seconds=args.cpu_seconds, game=game, n=n, sims=args.sims, P=args.parallel, blocks=args.blocks,
                               filters=args.filters, stagger=args.stagger)
            cpu["value"] = round(cpu["value"], 3)
            cpu["per_core"] = round(cpu["per_core"], 4)
        line = {
            "metric": "self-play moves/sec (whole node), 9x9 Go @ 200 sims/move" if (game == "go" and n == 9 and args.sims == 200)
            else f"self-play moves/sec (whole node), {n}x{n} {game} @ {args.sims} sims/move",
            "value": round(total_moves / elapsed_max, 2), "unit": "moves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed_max / steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.net_dtype, "data": "synthetic",
            "config": {"workload": f"{n}x{n} {game}, {args.games} games/GPU, {args.sims} sims/move (reference budget semantics), P={args.parallel}, "
                                   f"{args.blocks}x{args.filters} net", "net_dtype": args.net_dtype, "tree_dtype": "f32 (f64 noisy root)",
                       "evaluator": "tiled layout, hand-written stem / tower / head kernels" if actor.tiled_features else "library convolutions + fused epilogue",
                       "games_per_gpu": args.games, "stagger_plies": args.stagger, "hip_graph_forward": not args.no_graph,
                       "parallelism": f"games sharded x{world}, sample gather to rank 0"},
            "sims_per_sec": round(total_sims / elapsed_max, 1), "evals_per_sec": round(total_evals / elapsed_max, 1),
            "sims_per_move": round(total_sims / max(1.0, total_moves), 2),
            "select_nodes_per_sim": round(cnt["node_visits"] / max(1, cnt["sims"]), 3),
            "backup_nodes_per_sim": round(cnt["backup_edges"] / max(1, cnt["sims"]), 3),
            "samples_gathered": samples_at_root, "roofline": roofline, "engine_roofline": engine_roof, "nn_roofline": nn_roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
