"""Times azsp_conv3x3_split (the fp32-class split-precision tower convolution) against the library's fp32 convolution + fused epilogue at
the bench shape (9x9, 128 filters, 32768 rows).  Post-ReLU-like activations (half zeros), He-scaled weights."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alpha_zero_amd import _lib
from alpha_zero_amd.core.network import split_weights_f16

b = _lib.load()
if os.environ.get("AZ_BENCH_LIB"):  # an alternative build of the library (same-box A/B of two kernel versions)
    from alpha_zero_amd import _abi

    b = _abi.Binding(ctypes.CDLL(os.environ["AZ_BENCH_LIB"]), "A/B build")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
S, C = (int(v) for v in os.environ.get("CONV_BENCH_SHAPE", "9,128").split(","))
g = torch.Generator().manual_seed(0)


def act():
    t = torch.randn(B, C, S, S, generator=g)
    return torch.where(torch.rand(B, C, S, S, generator=g) < 0.5, torch.zeros(()), t.abs()).cuda().contiguous(memory_format=torch.channels_last)


x, res = act(), act()
w = (torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5)
bias = (torch.randn(C, generator=g) * 0.1).cuda()
wsp = split_weights_f16(w).cuda()
wl = w.cuda().contiguous(memory_format=torch.channels_last)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.backends.cudnn.benchmark = True
n = b.dll.azsp_split_bytes(B, S, C) // 2
xs, rs, ys = (torch.zeros(n, dtype=torch.float16, device="cuda") for _ in range(3))
assert b.dll.azsp_split_layout(x.data_ptr(), xs.data_ptr(), B, S, C, 1, None, st) == 0
assert b.dll.azsp_split_layout(res.data_ptr(), rs.data_ptr(), B, S, C, 1, None, st) == 0


def split(r):
    assert b.dll.azsp_conv3x3_split(xs.data_ptr(), wsp.data_ptr(), bias.data_ptr(), rs.data_ptr() if r is not None else None, ys.data_ptr(), B, S, C, 1, None, st) == 0


def lib(r):
    t = torch.nn.functional.conv2d(x, wl, None, padding=1)
    b.dll.azsp_bias_act(t.data_ptr(), bias.data_ptr(), r.data_ptr() if r is not None else None, B * S * S, C, 1, 1, st)
    return t


flops = 2.0 * B * S * S * C * C * 9
out = {"rows": B, "shape": [S, C]}
for name, f in (("split", split), ("library_fp32", lib)):
    for r in (None, res):
        for _ in range(3):
            f(r)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            f(r)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        out[f"{name}_{'residual' if r is not None else 'plain'}"] = {"ms": round(ms, 4), "fp32_equivalent_tflops": round(flops / ms / 1e9, 1)}
        print(f"{name:14s} residual={r is not None!s:5s} {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s (fp32-equivalent)", flush=True)
xk = x.clone()
for r in (None, res):  # correctness on the full batch against the library's fp32 result
    ref = lib(r).clone()
    split(r)
    y2 = torch.empty_like(x)
    assert b.dll.azsp_split_layout(ys.data_ptr(), y2.data_ptr(), B, S, C, 0, None, st) == 0
    torch.cuda.synchronize()
    d = (y2 - ref).abs().max().item() / ref.abs().max().item()
    out[f"max_rel_diff_vs_library_{'residual' if r is not None else 'plain'}"] = d
    print(f"split residual={r is not None}: max |split - library| / max|library| = {d:.3e}")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    assert b.dll.azsp_split_layout(xk.data_ptr(), xs.data_ptr(), B, S, C, 1, None, st) == 0
e1.record()
torch.cuda.synchronize()
out["split_layout_ms"] = round(e0.elapsed_time(e1) / 10, 4)
print("split_layout", out["split_layout_ms"], "ms")
print(json.dumps(out))
