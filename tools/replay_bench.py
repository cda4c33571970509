"""SURVEY 8f-1 measurement: replay sampling on the device (azsp_replay_gather: gather + dihedral + cast in one pass) against
the reference-equivalent host path (oracle UniformReplay.sample + H2D + apply_random_transformation as core/pipeline.py:636-643).
One JSON line.  Run on the GPU box: python tools/replay_bench.py [capacity] [batch]."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alpha_zero_amd.core.replay import DeviceReplay, Transition  # noqa: E402
from oracle.replay import OracleUniformReplay  # noqa: E402

cap = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
N, A = 9, 82
g = torch.Generator().manual_seed(0)
dev = DeviceReplay(cap, np.random.RandomState(0), N, A, device="cuda")
for _ in range(cap // 50_000):  # fill device-to-device like harvests do
    st = (torch.rand(50_000, 17, N, N, generator=g) > 0.6).to(torch.int8).cuda()
    pi = torch.softmax(torch.randn(50_000, A, generator=g), -1).cuda()
    z = torch.randint(-1, 2, (50_000,), generator=g).float().cuda()
    dev.add_harvest(st, pi, z, games=[0] * 500)
for dt in (torch.float32, torch.bfloat16):
    for _ in range(5):
        dev.sample_device(batch, transform="random", state_dtype=dt)
    torch.cuda.synchronize()
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        dev.sample_device(batch, transform="random", state_dtype=dt)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / reps
    esz = 4 if dt == torch.float32 else 2
    bytes_moved = batch * (17 * N * N * (1 + esz) + A * 8 + 8)
    print(json.dumps({"metric": "replay samples/sec (sample + augment + cast, on device)", "value": round(batch / t), "unit": "samples/s", "batch": batch,
                      "capacity": cap, "state_dtype": str(dt), "ms_per_batch": round(t * 1e3, 4), "alg_GBs": round(bytes_moved / t / 1e9, 1)}))
# host path of the reference: list storage, per-sample stack, H2D, transform on the device
hcap = min(cap, 100_000)
orc = OracleUniformReplay(hcap, np.random.RandomState(0))
st = (np.random.rand(hcap, 17, N, N) > 0.6).astype(np.int8)
pi = np.random.rand(hcap, A)
orc.add_game([Transition(st[i], pi[i], 1.0) for i in range(hcap)])
from alpha_zero_amd.utils.transformation import apply_random_transformation  # noqa: E402

t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    tr = orc.sample(batch)
    s = torch.from_numpy(tr.state).to("cuda", torch.float32)
    p = torch.from_numpy(tr.pi_prob).to("cuda", torch.float32)
    v = torch.from_numpy(tr.value).to("cuda", torch.float32)
    apply_random_transformation(s, p, v)
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / reps
print(json.dumps({"metric": "replay samples/sec, reference-equivalent host path (oracle sample + H2D + transform)", "value": round(batch / t), "unit": "samples/s",
                  "batch": batch, "ms_per_batch": round(t * 1e3, 3), "cores": 1, "kind": "port"}))
