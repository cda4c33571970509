"""Whole-evaluator forward time of the fp32-class InferenceNet (10 x 128, 9x9 Go, 32768 rows): all hand-written split kernels /
split tower behind a library stem and heads / all library."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alpha_zero_amd import _lib
from alpha_zero_amd.core.network import AlphaZeroNet, InferenceNet

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
torch.manual_seed(1)
net = AlphaZeroNet((17, 9, 9), 82, 10, 128, 128)
inf = InferenceNet(net, dtype=torch.float32, binding=_lib.load()).cuda()
torch.backends.cudnn.benchmark = True
x = (torch.rand(B, 17, 9, 9, generator=torch.Generator().manual_seed(0)) > 0.6).float().cuda()
pri, v = torch.empty(B, 82, device="cuda"), torch.empty(B, device="cuda")
out = {"rows": B}
for name, heads, tower in (("split_evaluator", True, True), ("split_tower_library_heads", False, True), ("library", False, False)):
    inf.use_split_heads, inf.use_split_tower = heads, tower
    for _ in range(3):
        inf(x, pri, v)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10 if tower else 4
    e0.record()
    for _ in range(n):
        inf(x, pri, v)
    e1.record()
    torch.cuda.synchronize()
    out[name + "_ms"] = round(e0.elapsed_time(e1) / n, 3)
    print(name, out[name + "_ms"], "ms", flush=True)
print(json.dumps(out))
