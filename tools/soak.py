"""Long self-play soak on the GPU: many rounds with periodic harvests; checks engine faults, buffer stalls and the
sanity of every harvested game (lengths, z values, policy normalisation, colour alternation)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alpha_zero_amd.core.network import AlphaZeroNet
from alpha_zero_amd.core.pipeline import SelfPlayActor

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
game, n, G = (sys.argv[2] if len(sys.argv) > 2 else "go"), int(sys.argv[3]) if len(sys.argv) > 3 else 9, int(sys.argv[4]) if len(sys.argv) > 4 else 4096
DT = {"bf16": torch.bfloat16, "fp32": torch.float32}[sys.argv[5] if len(sys.argv) > 5 else "fp32"]  # fp32: the split-precision evaluator where the shape has it
A = n * n + (1 if game == "go" else 0)
blocks = int(sys.argv[6]) if len(sys.argv) > 6 else (10 if game == "go" else 6)    # e.g. `... 19 1024 bf16 20 256 800` = BASELINE C5's network
filters = int(sys.argv[7]) if len(sys.argv) > 7 else (128 if game == "go" else 64)
sims = int(sys.argv[8]) if len(sys.argv) > 8 else 200
torch.manual_seed(1)
net = AlphaZeroNet((17, n, n), A, blocks, filters, filters, gomoku=(game != "go"))
actor = SelfPlayActor(net, game=game, board_size=n, num_games=G, num_simulations=sims, num_parallel=8, device="cuda", net_dtype=DT)
t0 = time.time()
games = samples = 0
lens = []
results = {}
for r in range(0, rounds, 50):
    actor.run_rounds(50)
    st, pi, z, rows = actor.harvest_tensors()
    if len(rows):
        stc, pic, zc = st.cpu().numpy(), pi.cpu().numpy(), z.cpu().numpy()
        assert np.all(np.isin(zc, (-1.0, 0.0, 1.0)))
        assert np.allclose(pic.sum(axis=1), 1.0, atol=1e-4), float(np.abs(pic.sum(axis=1) - 1).max())
        for row in rows:
            s0, ln = int(row[0]), int(row[1])
            assert 0 < ln <= 2 * n * n
            black = stc[s0:s0 + ln, 16, 0, 0]
            assert black[0] == 1 and np.all(black[1:] != black[:-1])
            zz = zc[s0:s0 + ln]
            if int(row[2]) != 0:  # winner exists: its samples are +1, the other's -1
                wb = 1 if int(row[2]) == 1 else 0
                assert np.all(zz[black == wb] == 1) and np.all(zz[black != wb] == -1)
            lens.append(ln)
            results[int(row[2])] = results.get(int(row[2]), 0) + 1
        games += len(rows)
        samples += len(zc)
cnt = actor.counters()
dt = time.time() - t0
print(json.dumps(dict(evaluator=actor.evaluator_path, net=f"{blocks}x{filters}", sims=sims, clamped_games=actor.clamped_games, rounds=rounds, seconds=round(dt, 1), games=games, samples=samples, mean_len=round(float(np.mean(lens)), 1) if lens else None,
                      winners=results, moves=cnt["moves"], moves_per_s=round(cnt["moves"] / dt, 1), sims_per_move=round(cnt["sims"] / max(1, cnt["moves"]), 1),
                      stalls=cnt["stalls"], dup_leaves=cnt["dup_leaves"], terminal_hits=cnt["terminal_hits"], range_events=actor.range_events,
                      range_rescales=actor.range_rescales, act_shift=actor.infer.act_shift, calibrated_max_abs=round(actor.infer.act_max_abs, 3))))
