"""Same-box, same-process A/B of the fused fp32-class 17x17 x 64 ResNetBlock (azsp_resblock_split, BASELINE C2's dominant kernel): the
library in the tree against another build (default tools/probes/libazsp_prev.so = the previous version of az_resblock_sp17.h), timed
alternately on the same post-ReLU-like activations; checks that both produce bit-identical outputs.
usage: python tools/resblock17_prev_ab.py [boards] [other_lib]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alpha_zero_amd import _abi, _lib
from alpha_zero_amd.core.network import split_weights_f16

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
other = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tools", "probes", "libazsp_prev.so")
libs = {"tree": _lib.load(), "other": _abi.Binding(ctypes.CDLL(other), "A/B build")}
S, C = 17, 64
g = torch.Generator().manual_seed(0)
t = torch.randn(B, C, S, S, generator=g)
x = torch.where(torch.rand(B, C, S, S, generator=g) < 0.5, torch.zeros(()), t.abs()).cuda().contiguous(memory_format=torch.channels_last)
ws = [split_weights_f16(torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda() for _ in range(2)]
bs = [(torch.randn(C, generator=g) * 0.1).cuda() for _ in range(2)]
n = libs["tree"].dll.azsp_split_bytes(B, S, C) // 2
xs = torch.zeros(n, dtype=torch.float16, device="cuda")
ys = {k: torch.zeros(n, dtype=torch.float16, device="cuda") for k in libs}
assert libs["tree"].dll.azsp_split_layout(x.data_ptr(), xs.data_ptr(), B, S, C, 1, None, None) == 0
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
flops = 2 * 2.0 * B * S * S * C * C * 9 * 3  # issued f16 MFMA products of the two convolutions


def run(k, reps):
    d = libs[k].dll
    for _ in range(reps):
        assert d.azsp_resblock_split(xs.data_ptr(), ws[0].data_ptr(), bs[0].data_ptr(), ws[1].data_ptr(), bs[1].data_ptr(), ys[k].data_ptr(), B, S, C, None, st) == 0


for k in libs:
    run(k, 3)
torch.cuda.synchronize()
print("bit-identical outputs:", torch.equal(ys["tree"], ys["other"]), flush=True)
for rnd in range(4):
    for k in ("tree", "other"):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(k, 30)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        print(f"round {rnd} {k:6s} {ms:7.3f} ms per block  {flops / ms / 1e9:7.1f} TFLOP/s of f16 products  frac {flops / ms / 1e9 / 2500:.4f}", flush=True)
