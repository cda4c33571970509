"""Times azsp_resblock_split (the fused split-precision ResNetBlock at 17x17 x 64) and, beside it, the two azsp_conv3x3_split launches it
replaces, at the C2 bench shape (32768 rows), HIP events on the launch stream, post-ReLU-like activations.  AZ_BENCH_LIB = an
alternative build of the library (same-box A/B; the ablation builds of tools/probes/make_rb_abl.py)."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alpha_zero_amd import _abi, _lib
from alpha_zero_amd.core.network import split_weights_f16

b = _lib.load()
label = "product"
if os.environ.get("AZ_BENCH_LIB"):
    b = _abi.Binding(ctypes.CDLL(os.environ["AZ_BENCH_LIB"]), "A/B build")
    label = os.path.basename(os.environ["AZ_BENCH_LIB"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
N = int(os.environ.get("RB_BENCH_LAUNCHES", "12"))
S, C = 17, 64
g = torch.Generator().manual_seed(0)
t = torch.randn(B, C, S, S, generator=g)
x = torch.where(torch.rand(B, C, S, S, generator=g) < 0.5, torch.zeros(()), t.abs()).cuda().contiguous(memory_format=torch.channels_last)
del t
ws = [split_weights_f16(torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda() for _ in range(2)]
bs = [(torch.randn(C, generator=g) * 0.1).cuda() for _ in range(2)]
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
n = b.dll.azsp_split_bytes(B, S, C) // 2
xs, ms, ys = (torch.zeros(n, dtype=torch.float16, device="cuda") for _ in range(3))
assert b.dll.azsp_split_layout(x.data_ptr(), xs.data_ptr(), B, S, C, 1, None, st) == 0


def fused():
    assert b.dll.azsp_resblock_split(xs.data_ptr(), ws[0].data_ptr(), bs[0].data_ptr(), ws[1].data_ptr(), bs[1].data_ptr(), ys.data_ptr(), B, S, C, None, st) == 0


def two():
    assert b.dll.azsp_conv3x3_split(xs.data_ptr(), ws[0].data_ptr(), bs[0].data_ptr(), None, ms.data_ptr(), B, S, C, 1, None, st) == 0
    assert b.dll.azsp_conv3x3_split(ms.data_ptr(), ws[1].data_ptr(), bs[1].data_ptr(), xs.data_ptr(), ys.data_ptr(), B, S, C, 1, None, st) == 0


flops = 2 * 2.0 * B * S * S * C * C * 9
out = {"lib": label, "rows": B}
for name, f in (("fused_block", fused), ("two_launches", two)):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms_ = e0.elapsed_time(e1) / N
    out[name] = {"ms_per_block": round(ms_, 4), "frac_of_f16_mfma_peak": round(3 * flops / ms_ / 1e9 / 2500.0, 4)}
print(json.dumps(out))
