"""Where a simulation of the drop-in uct_search goes on BASELINE C1 (13x13 Gomoku, 100 sims, the shipped 10 x 40 checkpoint, CPU eval_func):
wall time inside eval_func vs inside the engine step, per simulation.  usage: python tools/dropin_profile.py [spin]  (spin: hipDeviceScheduleSpin)"""
import ctypes
import os
import sys
import time

os.environ["OMP_NUM_THREADS"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "spin":
    hip = ctypes.CDLL("libamdhip64.so")
    print("hipSetDeviceFlags(spin) ->", hip.hipSetDeviceFlags(1))
import numpy as np
import torch

torch.set_num_threads(1)
from alpha_zero_amd.core import mcts_v2
from alpha_zero_amd.core.engine import Engine
from alpha_zero_amd.core.network import AlphaZeroNet
from alpha_zero_amd.envs.gomoku import GomokuEnv

n = 13
net = AlphaZeroNet((17, n, n), n * n, 10, 40, 80, gomoku=True)
ck = os.path.join(ROOT, "tests", "golden", "gomoku13_ckpt200000_network.pt")
if os.path.exists(ck):
    net.load_state_dict(torch.load(ck, map_location="cpu", weights_only=True)["network"], strict=True)
net = net.eval()
T = {"eval": 0.0, "step": 0.0, "n_eval": 0, "n_step": 0}


@torch.no_grad()
def eval_func(state, batched=False):
    t0 = time.perf_counter()
    x = torch.from_numpy(state if batched else state[None, ...]).to(dtype=torch.float32)
    logits, v = net(x)
    pi = torch.softmax(logits, dim=-1).cpu().numpy()
    v = np.squeeze(v.cpu().numpy(), axis=1).tolist()
    T["eval"] += time.perf_counter() - t0
    T["n_eval"] += 1
    return pi[0], v[0]


orig = Engine.dropin_step


def timed_step(self, *a, **k):
    t0 = time.perf_counter()
    r = orig(self, *a, **k)
    T["step"] += time.perf_counter() - t0
    T["n_step"] += 1
    return r


Engine.dropin_step = timed_step
def back_to_back(label, obs):
    for _ in range(20):
        eval_func(obs)
    t0 = time.perf_counter()
    for _ in range(300):
        eval_func(obs)
    print(f"eval_func back to back, {label}: {(time.perf_counter() - t0) / 300 * 1e3:.3f} ms per call", flush=True)


obs0 = (np.random.rand(17, n, n) > 0.7).astype(np.int8)
back_to_back("before the HIP runtime is initialised", obs0)
env = GomokuEnv(board_size=n)
obs = env.reset()
back_to_back("after (engine created, idle)", obs0)
for k in T:
    T[k] = 0
np.random.seed(1)
moves, t0, root = 0, time.perf_counter(), None
while time.perf_counter() - t0 < 8.0:
    mv, pi, rq, cq, root = mcts_v2.uct_search(env=env, eval_func=eval_func, root_node=root, c_puct_base=19652.0, c_puct_init=1.25, num_simulations=100,
                                              root_noise=True, warm_up=not (env.steps > 16))
    _, _, done, _ = env.step(mv)
    moves += 1
    if done:
        env.reset()
        root = None
dt = time.perf_counter() - t0
print(f"{moves / dt:.3f} moves/s; per move: {dt / moves * 1e3:.1f} ms = eval {T['eval'] / moves * 1e3:.1f} ms ({T['n_eval'] / moves:.1f} calls x {T['eval'] / max(1, T['n_eval']) * 1e3:.3f} ms)"
      f" + engine step {T['step'] / moves * 1e3:.1f} ms ({T['n_step'] / moves:.1f} calls x {T['step'] / max(1, T['n_step']) * 1e6:.1f} us) + rest {(dt - T['eval'] - T['step']) / moves * 1e3:.1f} ms")
