"""Worker of test_bench_measurement_flow_two_ranks_gloo: bench.py's own measurement functions (pre-roll with synchronised exit,
timed region with harvest + gather, totals over ranks) on 2 gloo ranks, each with a tiny host-twin actor of a DIFFERENT size so that
the ranks' pre-rolls would end at different rounds if they were not synchronised."""
import argparse
import json
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

rank, world, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dist.init_process_group("gloo", rank=rank, world_size=world)
import bench  # noqa: E402
import engine_util as eu  # noqa: E402
from alpha_zero_amd.core.network import AlphaZeroNet  # noqa: E402
from alpha_zero_amd.core.pipeline import SelfPlayActor  # noqa: E402

torch.manual_seed(1)
net = AlphaZeroNet((17, 5, 5), 26, 1, 8, 8)
sims = 12 if rank == 0 else (24 if world == 2 else 12 + 4 * (rank % 4))  # the ranks need different numbers of rounds per move
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 240
act = SelfPlayActor(net, game="go", board_size=5, num_games=4, num_simulations=sims, num_parallel=4, warm_up_steps=4, device="cpu",
                    net_dtype=torch.float32, use_graph=False, binding=eu.hosttwin_binding(), seed=1, rank=rank)
args = argparse.Namespace(preroll_rounds=10, preroll_moves=2, harvest_every=7, sims=sims, parallel=4)
dev = torch.device("cpu")
pre = bench.preroll(act, args, world, dev)
elapsed, cnt, evs, gathered = bench.timed(act, args, world, dev, warmup=3, steps=steps)
emax, moves, simsum, evals = bench.reduce_totals(cnt, elapsed, world, dev)
ranks = bench.per_rank_report(cnt, elapsed, world, dev)
json.dump(dict(preroll=pre, local_moves=cnt["moves"], total_moves=moves, elapsed=elapsed, elapsed_max=emax, gathered=gathered, per_rank=ranks),
          open(os.path.join(outdir, f"flow{rank}.json"), "w"))
dist.barrier()
dist.destroy_process_group()
