"""Diagnostic: run one MCTS golden config on a backend and print per-game lengths / stats / first mismatch."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import engine_util as eu  # noqa: E402
import golden_mcts  # noqa: E402
from synth_eval import eval_batch  # noqa: E402

kind, name = sys.argv[1], sys.argv[2]
G = golden_mcts.MctsGolden(name)
logs, (states, pis, zs, games), nev, cnt = eu.run_golden_selfplay(kind, G, eval_batch)
print("counters", {k: v for k, v in cnt.items() if v})
print("games rows:\n", games)
for gi in range(G.cfg["games"]):
    ix = G.moves_of_game(gi)
    print("game", gi, "golden searched moves", len(ix), "finished", G.finished(gi))
    if G.finished(gi):
        st, gp, gz, gstats = G.samples(gi)
        print("  golden len", len(st), gstats)
        rows = [r for r in games if r[15] == gi]
        for r in rows:
            s0, ln = int(r[0]), int(r[1])
            n = min(ln, len(st))
            eq = [np.array_equal(states[s0 + k], st[k]) for k in range(n)]
            print("  engine len", ln, "first state mismatch", eq.index(False) if False in eq else None, "z eq", np.array_equal(zs[s0:s0 + n], gz[:n].astype(np.float32)))
    bad = [k for k, i in enumerate(ix) if k < len(logs[gi]) and (logs[gi][k]["move"] != G.g["move"][i] or not np.array_equal(logs[gi][k]["child_N"], G.g["child_N"][i]))]
    print("  log mismatches at plies", bad[:10], "moves engine", [l["move"] for l in logs[gi]][-6:], "golden", list(G.g["move"][ix][-6:]))
