"""Worker of test_sample_gather_{two,eight}_ranks_gloo: one process per rank, host-twin engine, gloo backend.  With more than two ranks
the ranks play different budgets (uneven sample counts per harvest) and every third (rank, iteration) contributes NOTHING to the gather
(zero count: its finished games stay staged in the engine until its next turn)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

rank, world, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dist.init_process_group("gloo", rank=rank, world_size=world)
import engine_util as eu  # noqa: E402
from alpha_zero_amd.core.gather import gather_samples  # noqa: E402
from alpha_zero_amd.core.network import AlphaZeroNet  # noqa: E402
from alpha_zero_amd.core.pipeline import SelfPlayActor  # noqa: E402

torch.manual_seed(1)
net = AlphaZeroNet((17, 5, 5), 26, 1, 8, 8)
a = SelfPlayActor(net, game="go", board_size=5, num_games=4, num_simulations=12 if world <= 2 else 8 + 4 * (rank % 3), num_parallel=2, warm_up_steps=4, device="cpu",
                  net_dtype=torch.float32, use_graph=False, binding=eu.hosttwin_binding(), seed=1, rank=rank)
acc = None
for it in range(400):
    a.run_rounds(20)
    if world > 2 and (rank + it) % 3 == 0:  # this rank sits the exchange out: zero samples, zero games
        st, pi, z, games = (torch.empty((0, 17, 5, 5), dtype=torch.int8), torch.empty((0, 26), dtype=torch.float32), torch.empty((0,), dtype=torch.float32),
                            np.zeros((0, 16), dtype=np.int32))
    else:
        st, pi, z, games = a.harvest_tensors()
    np.savez(os.path.join(outdir, f"local{rank}_{it}.npz"), states=st.numpy(), pi=pi.numpy(), z=z.numpy(), games=games)
    res = gather_samples(st.clone(), pi.clone(), z.clone(), games, dst=0)
    flag = torch.tensor([0])
    if rank == 0:
        if acc is None:
            acc = [[], [], [], [], 0]
        if len(res[3]):
            g = res[3].copy()
            g[:, 0] += acc[4]
            acc[0].append(res[0].numpy().copy()), acc[1].append(res[1].numpy().copy()), acc[2].append(res[2].numpy().copy()), acc[3].append(g)
            acc[4] += res[0].shape[0]
        ranks_seen = set(int(x) >> 20 for gg in acc[3] for x in gg[:, 15])
        flag[0] = 1 if ranks_seen == set(range(world)) and it >= 5 else 0
    dist.broadcast(flag, 0)
    if flag.item():
        break
# merge this rank's local harvests for the comparison
parts, base = [[], [], [], []], 0
for j in range(it + 1):
    d = np.load(os.path.join(outdir, f"local{rank}_{j}.npz"))
    g = d["games"].copy()
    if len(g):
        g[:, 0] += base
    base += d["z"].shape[0]
    parts[0].append(d["states"]), parts[1].append(d["pi"]), parts[2].append(d["z"]), parts[3].append(g)
np.savez(os.path.join(outdir, f"local{rank}.npz"), states=np.concatenate(parts[0]), pi=np.concatenate(parts[1]), z=np.concatenate(parts[2]),
         games=np.concatenate(parts[3]))
if rank == 0:
    np.savez(os.path.join(outdir, "rank0.npz"), states=np.concatenate(acc[0]), pi=np.concatenate(acc[1]), z=np.concatenate(acc[2]),
             games=np.concatenate(acc[3]))
dist.barrier()
dist.destroy_process_group()
