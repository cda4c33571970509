"""Same-box A/B of the wave-per-tile fp32-class convolution k_conv3x3_spg (csrc/az_conv_spg.h) against the weight-stationary kernels of
the tailored shapes over the boards per launch (where is the crossover: azsp_small_batch_waves), of the fused block against two
wave-per-tile launches, and of the 19x19 x 256 tower convolution against the library's fp32 convolution.
usage: python tools/spg_ab.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F

from alpha_zero_amd import _lib
from alpha_zero_amd.core.network import split_weights_f16

bnd = _lib.load()
dll = bnd.dll
dev = "cuda"


def bufs(B, S, C, seed=1):
    g = torch.Generator().manual_seed(seed)
    n = dll.azsp_split_bytes(B, S, C) // 2
    x = torch.relu(torch.randn(B, C, S, S, generator=g)).to(dev).contiguous(memory_format=torch.channels_last)
    xs, ms, ys = (torch.zeros(n, dtype=torch.float16, device=dev) for _ in range(3))
    assert dll.azsp_split_layout(x.data_ptr(), xs.data_ptr(), B, S, C, 1, None, None) == 0
    w = [split_weights_f16(torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).to(dev) for _ in range(2)]
    b = [(torch.randn(C, generator=g) * 0.1).to(dev) for _ in range(2)]
    return x, xs, ms, ys, w, b


def timed(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


old = dll.azsp_small_batch_waves(-1)
out = {"conv_us": {}, "block_us": {}}
for S, C in ((9, 128), (9, 64), (17, 64)):
    for B in (1, 2, 4, 8, 16, 32, 48, 64, 96, 128, 192, 256, 512):
        x, xs, ms, ys, w, b = bufs(B, S, C)
        conv = lambda: dll.azsp_conv3x3_split(xs.data_ptr(), w[0].data_ptr(), b[0].data_ptr(), xs.data_ptr(), ys.data_ptr(), B, S, C, 1, None, None)
        dll.azsp_small_batch_waves(0)
        t_ws = timed(conv, 200)
        dll.azsp_small_batch_waves(1 << 20)
        t_pt = timed(conv, 200)
        out["conv_us"][f"{S}x{S}x{C} boards={B}"] = {"weight_stationary": round(t_ws, 2), "wave_per_tile": round(t_pt, 2)}
        if C == 64:
            blk = lambda: dll.azsp_resblock_split(xs.data_ptr(), w[0].data_ptr(), b[0].data_ptr(), w[1].data_ptr(), b[1].data_ptr(), ys.data_ptr(), B, S, C, None, None)
            dll.azsp_small_batch_waves(0)
            t_f = timed(blk, 200)
            dll.azsp_small_batch_waves(1 << 20)
            t_2 = timed(blk, 200) if B <= 256 else None
            out["block_us"][f"{S}x{S}x{C} boards={B}"] = {"fused_block": round(t_f, 2), "two_wave_per_tile_launches": round(t_2, 2) if t_2 else None}
        print(S, C, B, out["conv_us"][f"{S}x{S}x{C} boards={B}"], out["block_us"].get(f"{S}x{S}x{C} boards={B}"), flush=True)
# shapes without a tailored kernel: throughput tiles against the library's fp32 convolution (+ bias + residual + ReLU as separate torch ops)
out["untailored"] = {}
for S, C, B in ((19, 256, 1024), (19, 128, 2048), (13, 64, 8192), (19, 256, 8)):
    x, xs, ms, ys, w, b = bufs(B, S, C)
    conv = lambda: dll.azsp_conv3x3_split(xs.data_ptr(), w[0].data_ptr(), b[0].data_ptr(), xs.data_ptr(), ys.data_ptr(), B, S, C, 1, None, None)
    dll.azsp_small_batch_waves(0)
    t_tp = timed(conv, 20)
    dll.azsp_small_batch_waves(1 << 20)
    t_lat = timed(conv, 20)
    g = torch.Generator().manual_seed(3)
    wl = (torch.randn(C, C, 3, 3, generator=g) * 0.05).to(dev).contiguous(memory_format=torch.channels_last)
    bl = torch.randn(C, generator=g).to(dev)
    lib = lambda: torch.relu_(F.conv2d(x, wl, bl, padding=1).add_(x))
    t_lib = timed(lib, 20)
    flop = 2.0 * B * S * S * C * C * 9
    out["untailored"][f"{S}x{S}x{C} boards={B}"] = {"throughput_tiles_us": round(t_tp, 1), "latency_tiles_us": round(t_lat, 1), "library_fp32_us": round(t_lib, 1),
                                                   "fp32_equiv_tflops": round(flop / t_tp / 1e6, 1), "frac_of_f16_peak_x3": round(3 * flop / t_tp / 1e6 / 2500.0, 4),
                                                   "library_fp32_tflops": round(flop / t_lib / 1e6, 1)}
    print(S, C, B, out["untailored"][f"{S}x{S}x{C} boards={B}"], flush=True)
dll.azsp_small_batch_waves(old)
print(json.dumps(out))
