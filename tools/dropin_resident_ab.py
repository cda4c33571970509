"""BASELINE C1 through uct_search with the product evaluator on the device, three ways on the same box: (a) behind a HOST callback
(upload + forward + read-back per simulation), (b) DeviceEvaluator, eager forward (device-resident loop), (c) DeviceEvaluator, hipGraph
forward.  usage: python tools/dropin_resident_ab.py [seconds]"""
import json
import os
import sys
import time

os.environ["OMP_NUM_THREADS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

torch.set_num_threads(1)
from alpha_zero_amd import _lib
from alpha_zero_amd.core.evaluate import DeviceEvaluator
from alpha_zero_amd.core.mcts_v2 import uct_search
from alpha_zero_amd.core.network import AlphaZeroNet, InferenceNet, widen_for_kernels
from alpha_zero_amd.envs.gomoku import GomokuEnv

n, sims = 13, 100
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
net = AlphaZeroNet((17, n, n), n * n, 10, 40, 80, gomoku=True)
ck = os.path.join(ROOT, "tests", "golden", "gomoku13_ckpt200000_network.pt")
if os.path.exists(ck):
    net.load_state_dict(torch.load(ck, map_location="cpu", weights_only=True)["network"], strict=True)
wnet, _ = widen_for_kernels(net.eval(), n, torch.float32)
inf = InferenceNet(wnet, dtype=torch.float32, binding=_lib.load()).cuda()


def play(ev, budget):
    np.random.seed(1)
    env = GomokuEnv(board_size=n)
    moves, t0, hist = 0, time.perf_counter(), []
    while time.perf_counter() - t0 < budget:
        env.reset()
        root, done = None, False
        while not done and time.perf_counter() - t0 < budget:
            mv, pi, rq, cq, root = uct_search(env=env, eval_func=ev, root_node=root, c_puct_base=19652.0, c_puct_init=1.25, num_simulations=sims,
                                              root_noise=True, warm_up=not (env.steps > 16))
            _, _, done, _ = env.step(mv)
            hist.append(int(mv))
            moves += 1
    return moves / (time.perf_counter() - t0), moves, hist


out = {}
for label, ev in (("host_callback", DeviceEvaluator(inf).__call__), ("resident_eager", DeviceEvaluator(inf, use_graph=False)),
                  ("resident_graph", DeviceEvaluator(inf, use_graph=True))):
    play(ev, 0.5)
    c0 = getattr(ev, "_calls", 0)
    v, m, hist = play(ev, seconds)
    out[label] = {"moves_per_s": round(v, 2), "moves": m, "first_moves": hist[:12]}
    if hasattr(ev, "_calls"):
        out[label]["forwards_per_move"] = round((ev._calls - c0) / max(m, 1), 1)
        out[label]["us_per_forward_all_in"] = round(1e6 * seconds / max(ev._calls - c0, 1), 1)
print(json.dumps(out, indent=1))
