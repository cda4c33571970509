"""Helpers shared by the oracle / host-twin / GPU search-parity tests: load an MCTS golden file
(produced by the reference, tools/gen_golden_mcts.py) and expose its per-move records."""
import glob
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names(real_network=False):
    """Goldens recorded with the synthetic hash evaluator (tests/synth_eval.py), or (real_network=True) the ones recorded with the
    reference's shipped checkpoint as evaluator (tests/realnet_checks.py)."""
    return sorted(n for n in (os.path.basename(f)[5:-4] for f in glob.glob(os.path.join(GOLDEN, "mcts_*.npz"))) if ("ckpt" in n) == real_network)


class MctsGolden:
    def __init__(self, name):
        self.g = np.load(os.path.join(GOLDEN, f"mcts_{name}.npz"))
        self.cfg = json.loads(str(self.g["config"]))
        self.A = self.cfg["num_actions"]

    def moves_of_game(self, gi):
        return np.flatnonzero(self.g["game"] == gi)

    def finished(self, gi):
        return bool(int(self.g[f"g{gi}_finished"]))

    def samples(self, gi):
        n = self.cfg["n"]
        st = np.unpackbits(self.g[f"g{gi}_states"], axis=1)[:, : 17 * n * n].reshape(-1, 17, n, n).astype(np.int8)
        return st, self.g[f"g{gi}_pis"], self.g[f"g{gi}_zs"], json.loads(str(self.g[f"g{gi}_stats"]))
