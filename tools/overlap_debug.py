"""Debug aid: where do two actors with the same seed first differ?  Modes: serial, overlap (two streams), overlap1 (both halves on
ONE stream: half-batch evaluation without concurrency), nograph variants."""
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alpha_zero_amd.core.network import AlphaZeroNet  # noqa: E402
from alpha_zero_amd.core.pipeline import SelfPlayActor  # noqa: E402

game, n, filters = (sys.argv[1], int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ("go", 9, 128)
A = n * n + (1 if game == "go" else 0)
torch.manual_seed(4)
net = AlphaZeroNet((17, n, n), A, 2, filters, 64, gomoku=(game != "go"))


def make(mode):
    ov = mode.startswith("overlap")
    act = SelfPlayActor(net, game=game, board_size=n, num_games=1184, num_simulations=24, num_parallel=8, warm_up_steps=4, resign_threshold=-1.0,
                        seed=7, device="cuda", overlap_engine=ov, use_graph="nograph" not in mode, engine_kw={"max_steps": 24})
    if mode.startswith("overlap1"):
        act._streams = [act._streams[0], act._streams[0]]
    return act


def trace(mode, rounds=60):
    act = make(mode)
    out = []
    for r in range(rounds):
        act.run_round()
        st, q = act.engine.status()
        pri = act.engine.priors.clone().cpu()
        out.append((st.copy(), q.copy(), pri))
    return out, act._halves


ref, _ = trace("serial")
for mode in ("serial", "overlap1-nograph", "overlap1", "overlap-nograph", "overlap"):
    tr, halves = trace(mode)
    first = None
    for r, ((s0, q0, p0), (s1, q1, p1)) in enumerate(zip(ref, tr)):
        ds = np.flatnonzero((s0 != s1).any(axis=1))
        dq = np.flatnonzero((q0 != q1).any(axis=1))
        dp = torch.nonzero((p0 != p1).any(dim=1)).flatten().numpy()
        if len(ds) or len(dq) or len(dp):
            first = (r, len(ds), ds[:6].tolist(), len(dq), dq[:6].tolist(), len(dp), dp[:6].tolist(),
                     float((p0 - p1).abs().max()))
            break
    print(mode, "halves", halves, "first difference (round, #status rows, rows, #q rows, rows, #prior rows, rows, max |dprior|):", first, flush=True)
