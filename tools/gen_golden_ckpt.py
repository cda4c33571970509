"""Fixture from the reference's SHIPPED checkpoint (checkpoints/gomoku/13x13/training_steps_200000.ckpt, 10 blocks x 40 filters,
80 fc units): the network weights (data; optimizer / scheduler state dropped) and the fp32 outputs of the REFERENCE AlphaZeroNet
module on positions from seeded random Gomoku games played in the reference's GomokuEnv.
 -> tests/golden/gomoku13_ckpt200000_network.pt   {"network": state_dict, "training_steps": 200000}
 -> tests/golden/gomoku13_ckpt200000_outputs.npz  states int8[n,17,13,13], logits f32[n,169], value f32[n]
Development container only (imports /root/reference)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

ref_harness.install(13)
from alpha_zero.core.network import AlphaZeroNet  # noqa: E402
from alpha_zero.envs.gomoku import GomokuEnv  # noqa: E402

CKPT = os.path.join(ref_harness.REF_ROOT, "checkpoints", "gomoku", "13x13", "training_steps_200000.ckpt")
st = torch.load(CKPT, map_location="cpu", weights_only=False)
net = AlphaZeroNet((17, 13, 13), 169, num_res_block=10, num_filters=40, num_fc_units=80, gomoku=True)
net.load_state_dict(st["network"])
net.eval()
rng = np.random.Generator(np.random.PCG64(5))
states = []
env = GomokuEnv(board_size=13)
for g in range(24):
    obs = env.reset()
    for t in range(int(rng.integers(2, 60))):
        legal = np.flatnonzero(env.legal_actions)
        if env.is_game_over() or len(legal) == 0:
            break
        # mostly near the centre, like real games
        c = legal[np.argsort(np.abs(legal // 13 - 6) + np.abs(legal % 13 - 6) + rng.random(len(legal)) * 6)[: 12]]
        obs, _, done, _ = env.step(int(c[rng.integers(len(c))]))
        if not done:
            states.append(obs.copy())
states = np.stack(states[:: max(1, len(states) // 96)][:96]).astype(np.int8)
with torch.no_grad():
    logits, value = net(torch.from_numpy(states).float())
torch.save({"network": {k: v.clone() for k, v in st["network"].items()}, "training_steps": int(st["training_steps"])},
           os.path.join(ROOT, "tests", "golden", "gomoku13_ckpt200000_network.pt"))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "gomoku13_ckpt200000_outputs.npz"), states=states, logits=logits.numpy(),
                    value=value.squeeze(1).numpy())
print(states.shape, logits.shape, float(torch.softmax(logits, -1).max(-1).values.mean()), value.abs().mean().item())
