"""Debug aid: a whole-batch actor and a split-range actor (two disjoint game ranges per round, each with its half-batch forward) in
lock-step, same seed, harvest every 40 rounds as tests/test_engine_gpu.py does: the first round in which status / priors differ."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alpha_zero_amd.core.network import AlphaZeroNet  # noqa: E402
from alpha_zero_amd.core.pipeline import SelfPlayActor  # noqa: E402

n, filters, A, G, P, tb, g_split = 9, 128, 82, 1184, 8, 3, 576
torch.manual_seed(4)
net = AlphaZeroNet((17, n, n), A, 2, filters, 64)
mk = lambda: SelfPlayActor(net, game="go", board_size=n, num_games=G, num_simulations=24, num_parallel=P, warm_up_steps=4, resign_threshold=-1.0,
                           seed=7, device="cuda", use_graph=False, net_dtype=torch.bfloat16, engine_kw={"max_steps": 24})
W, S = mk(), mk()
harvest = os.environ.get("SPLIT_DEBUG_HARVEST", "1") == "1"
for r in range(240):
    W.run_round()
    e = S.engine
    for k, (g0, g1) in enumerate(((g_split, G), (0, g_split))):
        e.expand_backup(g0, g1)
        e.select(g0, g1)
        r0, r1 = g0 * P, g1 * P
        S.infer.forward_tiled(e.features[(r0 // tb) * (32 * tb * n * n):], r1 - r0, n, e.priors[r0:r1], e.values[r0:r1], slot=1 + k)
    sw, qw = W.engine.status()
    ss, qs = S.engine.status()
    dp = torch.nonzero((W.engine.priors != S.engine.priors).any(dim=1)).flatten()
    dv = torch.nonzero(W.engine.valid != S.engine.valid).flatten()
    df = int((W.engine.features != S.engine.features).sum())
    ds = np.flatnonzero((sw != ss).any(axis=1))
    if len(ds) or dp.numel() or dv.numel() or df:
        print("round", r, "status rows", len(ds), ds[:8].tolist(), "prior rows", dp.numel(), dp[:8].tolist(), "valid rows", dv.numel(), dv[:8].tolist(),
              "feature elements", df, "status W/S of first row", (sw[ds[0]].tolist(), ss[ds[0]].tolist()) if len(ds) else None, flush=True)
        break
    if harvest and (r + 1) % 40 == 0:
        gw = W.harvest_tensors()[3]
        gs = S.harvest_tensors()[3]
        print("round", r, "harvested", len(gw), len(gs), "same uid sets", set(gw[:, 11].tolist()) == set(gs[:, 11].tolist()), flush=True)
else:
    print("no difference in 240 rounds")
