"""Times the fused MFMA conv3x3 kernel against the library convolution + fused epilogue at the bench shape."""
import ctypes
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alpha_zero_amd import _lib

b = _lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
S, C = (int(v) for v in os.environ.get('CONV_BENCH_SHAPE', '9,128').split(','))  # '17,64' = the 13x13 Gomoku tower
g = torch.Generator().manual_seed(0)
SCALE = float(os.environ.get('CONV_BENCH_SCALE', '1'))  # 0: all-zero activations (data-dependent power check)
x = SCALE * torch.randn(B, C, S, S, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
res = SCALE * torch.randn(B, C, S, S, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
bias = torch.randn(C, generator=g).cuda()
bias16 = bias.to(torch.bfloat16)
wp = w.permute(2, 3, 0, 1).reshape(9, C, C).contiguous()
y = torch.empty_like(x)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.backends.cudnn.benchmark = True


def nhwc_rows(t):  # [B, C, S, S] channels-last tensor -> its [B*S*S, C] storage
    return t.permute(0, 2, 3, 1).reshape(-1, C)


nbytes = b.dll.azsp_tiled_bytes(B, S, C)
xt, rt, yt = (torch.zeros(nbytes // 2, dtype=torch.bfloat16, device="cuda") for _ in range(3))
b.dll.azsp_tile_layout(x.data_ptr(), xt.data_ptr(), B, S, C, 1, st)
b.dll.azsp_tile_layout(res.data_ptr(), rt.data_ptr(), B, S, C, 1, st)


def tiled(r):
    b.dll.azsp_conv3x3_tiled(xt.data_ptr(), wp.data_ptr(), bias.data_ptr(), rt.data_ptr() if r is not None else None, yt.data_ptr(), B, S, C, 1, st)


def layout(r):
    b.dll.azsp_tile_layout(x.data_ptr(), xt.data_ptr(), B, S, C, 1, st)


def lib(r):
    t = torch.nn.functional.conv2d(x, w, None, padding=1)
    b.dll.azsp_bias_act(t.data_ptr(), bias16.data_ptr(), r.data_ptr() if r is not None else None, B * S * S, C, 2, 1, st)
    return t


w2 = (torch.randn(C, C, 3, 3, generator=g) * 0.05).to(torch.bfloat16).cuda()
wp2 = w2.permute(2, 3, 0, 1).reshape(9, C, C).contiguous()


def block_fused(r):  # one whole ResNetBlock per launch (64-filter towers only)
    assert b.dll.azsp_resblock_tiled(xt.data_ptr(), wp.data_ptr(), bias.data_ptr(), wp2.data_ptr(), bias.data_ptr(), yt.data_ptr(), B, S, C, st) == 0


def block_two_launches(r):
    b.dll.azsp_conv3x3_tiled(xt.data_ptr(), wp.data_ptr(), bias.data_ptr(), None, rt.data_ptr(), B, S, C, 1, st)
    b.dll.azsp_conv3x3_tiled(rt.data_ptr(), wp2.data_ptr(), bias.data_ptr(), xt.data_ptr(), yt.data_ptr(), B, S, C, 1, st)


flops = 2.0 * B * S * S * C * C * 9
if C == 64:
    for name, f in (("block_fused", block_fused), ("block_2_launches", block_two_launches)):
        for _ in range(5):
            f(None)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f(None)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name:16s} (2 convolutions)  {ms:8.3f} ms  {2 * flops / ms / 1e9:8.1f} TFLOP/s")
    b.dll.azsp_tile_layout(res.data_ptr(), rt.data_ptr(), B, S, C, 1, st)  # block_two_launches used rt as its intermediate
for name, f in (("tiled_ws", tiled), ("miopen+epilogue", lib), ("tile_layout", layout)):
    for r in (None, res):
        for _ in range(5):
            f(r)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f(r)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        print(f"{name:16s} residual={r is not None!s:5s} {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s")
for r in (None, res):
    ref = lib(r).float()
    tiled(r)
    y2 = torch.empty_like(x)
    b.dll.azsp_tile_layout(yt.data_ptr(), y2.data_ptr(), B, S, C, 0, st)
    torch.cuda.synchronize()
    d = (y2.float() - ref).abs()
    print(f"tiled residual={r is not None}: max |tiled - library| = {d.max().item()}, mismatches > 0.13: {(d > 0.13).sum().item()}")
x2 = torch.empty_like(x)
b.dll.azsp_tile_layout(xt.data_ptr(), x2.data_ptr(), B, S, C, 0, st)
torch.cuda.synchronize()
print("layout round trip exact:", torch.equal(x2, x))
