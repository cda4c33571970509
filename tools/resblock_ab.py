"""Same-box, same-process A/B of the fused fp32-class ResNetBlock (azsp_resblock_split) against the two unfused launches
(azsp_conv3x3_split x 2) it replaces, timed alternately on the same post-ReLU-like activations; checks bit-identical outputs.
usage: python tools/resblock_ab.py [S = 17 | 9] [boards]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alpha_zero_amd import _lib
from alpha_zero_amd.core.network import split_weights_f16

S = int(sys.argv[1]) if len(sys.argv) > 1 else 17
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
C = 64
d = _lib.load().dll
g = torch.Generator().manual_seed(0)
t = torch.randn(B, C, S, S, generator=g)
x = torch.where(torch.rand(B, C, S, S, generator=g) < 0.5, torch.zeros(()), t.abs()).cuda().contiguous(memory_format=torch.channels_last)
ws = [split_weights_f16(torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda() for _ in range(2)]
bs = [(torch.randn(C, generator=g) * 0.1).cuda() for _ in range(2)]
n = d.azsp_split_bytes(B, S, C) // 2
xs, ms, yf, y2 = (torch.zeros(n, dtype=torch.float16, device="cuda") for _ in range(4))
assert d.azsp_split_layout(x.data_ptr(), xs.data_ptr(), B, S, C, 1, None, None) == 0
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
flops = 2 * 2.0 * B * S * S * C * C * 9 * 3  # issued f16 MFMA products of the two convolutions (algorithmic positions)


def fused(reps):
    for _ in range(reps):
        assert d.azsp_resblock_split(xs.data_ptr(), ws[0].data_ptr(), bs[0].data_ptr(), ws[1].data_ptr(), bs[1].data_ptr(), yf.data_ptr(), B, S, C, None, st) == 0


def two(reps):
    for _ in range(reps):
        assert d.azsp_conv3x3_split(xs.data_ptr(), ws[0].data_ptr(), bs[0].data_ptr(), None, ms.data_ptr(), B, S, C, 1, None, st) == 0
        assert d.azsp_conv3x3_split(ms.data_ptr(), ws[1].data_ptr(), bs[1].data_ptr(), xs.data_ptr(), y2.data_ptr(), B, S, C, 1, None, st) == 0


fused(3), two(3)
torch.cuda.synchronize()
print(f"{S}x{S} x {C}, {B} boards; bit-identical outputs:", torch.equal(yf, y2), flush=True)
for rnd in range(4):
    for name, f in (("fused", fused), ("two launches", two)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        f(30)
        e1.record()
        torch.cuda.synchronize()
        ms_ = e0.elapsed_time(e1) / 30
        print(f"round {rnd} {name:13s} {ms_:7.3f} ms per block  {flops / ms_ / 1e9:7.1f} TFLOP/s of f16 products  frac {flops / ms_ / 1e9 / 2500:.4f}", flush=True)
