"""Golden vectors for the replay row from the reference's own UniformReplay (this container only).

Writes tests/golden/replay_uniform.npz: the transitions of a few synthetic games (5x5 Go shapes), the batches the reference
returns for a fixed RandomState, and its counters, with a capacity small enough that the ring wraps."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ref_harness  # noqa: E402

ref_harness.install(5)
from alpha_zero.core.replay import Transition, UniformReplay  # noqa: E402

N, A, CAP, BATCH = 5, 26, 37, 8
rng = np.random.Generator(np.random.PCG64(2024))
lengths = [9, 14, 3, 21, 7, 12]
games = []
for ln in lengths:
    st = (rng.random((ln, 17, N, N)) > 0.6).astype(np.int8)
    pi = rng.random((ln, A)).astype(np.float32)
    pi /= pi.sum(axis=1, keepdims=True)
    z = rng.choice(np.array([-1.0, 0.0, 1.0]), size=ln)
    games.append((st, pi.astype(np.float64), z))  # Go: float64 policy in the reference, float32-representable values

rp = UniformReplay(CAP, np.random.RandomState(5), compress_data=False)
out = {"capacity": CAP, "batch": BATCH, "lengths": np.array(lengths), "seed": 5}
batches, sizes, none_before = [], [], []
for gi, (st, pi, z) in enumerate(games):
    out[f"g{gi}_state"], out[f"g{gi}_pi"], out[f"g{gi}_z"] = st, pi, z
    early = rp.sample(BATCH) if gi == 0 else 1
    none_before.append(early is None)
    rp.add_game([Transition(state=st[i], pi_prob=pi[i], value=float(z[i])) for i in range(len(z))])
    sizes.append(rp.size)
    b = rp.sample(BATCH)
    if b is None:
        batches.append(None)
    else:
        batches.append(b)
for k, b in enumerate(batches):
    out[f"b{k}_none"] = b is None
    if b is not None:
        out[f"b{k}_state"], out[f"b{k}_pi"], out[f"b{k}_z"] = b.state, b.pi_prob, b.value
out["sizes"] = np.array(sizes)
out["none_before_first_game"] = np.array(none_before)
out["num_games_added"], out["num_samples_added"] = rp.num_games_added, rp.num_samples_added
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "replay_uniform.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, {k: (v.shape if hasattr(v, "shape") else v) for k, v in out.items() if k.startswith("b")})
