"""A handful of launches of ONE tower-convolution kernel family on post-ReLU-like data, for `rocprofv3 --pmc ...` / `--kernel-trace` passes.
usage: python tools/pmc_launches.py <family> [boards]    family: split9 (k_conv3x3_sp, 9x9 x 128), split9_64, split17 (k_conv3x3_sp17,
17x17 x 64), splitblock17 / splitblock9_64 (k_resblock_sp: one launch per ResNetBlock, 17x17 x 64 / 9x9 x 64), tiled9 (k_conv3x3_tiled bf16, 9x9 x 128), hb19
(k_conv3x3_hb19 bf16, 19x19 x 256, two launches per convolution), spg19 (k_conv3x3_spgw: the wave-per-tile fp32-class convolution on 19x19 x 256, 1024 boards).
Launch i uses a residual when i is odd (the forward alternates plain / residual layers)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alpha_zero_amd import _lib
from alpha_zero_amd.core.network import split_weights_f16

fam = sys.argv[1]
S, C, split = {"split9": (9, 128, True), "split9_64": (9, 64, True), "splitblock9_64": (9, 64, True), "split17": (17, 64, True), "splitblock17": (17, 64, True), "tiled9": (9, 128, False),
               "hb19": (19, 256, False), "spg19": (19, 256, True)}[fam]
B = int(sys.argv[2]) if len(sys.argv) > 2 else (4096 if fam == "hb19" else 1024 if fam == "spg19" else 32768)
N = int(os.environ.get("PMC_LAUNCHES", "6"))
b = _lib.load()
g = torch.Generator().manual_seed(0)


def acts():
    t = torch.randn(B, C, S, S, generator=g)
    return torch.where(torch.rand(B, C, S, S, generator=g) < 0.5, torch.zeros(()), t.abs())


w = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
bias = (torch.randn(C, generator=g) * 0.1).cuda()
if split:
    n = b.dll.azsp_split_bytes(B, S, C) // 2
    xs, rs, ys = (torch.zeros(n, dtype=torch.float16, device="cuda") for _ in range(3))
    for dst in (xs, rs):
        t = acts().cuda().contiguous(memory_format=torch.channels_last)
        assert b.dll.azsp_split_layout(t.data_ptr(), dst.data_ptr(), B, S, C, 1, None, None) == 0
        del t
    wp = split_weights_f16(w).cuda()
    for i in range(N if fam.startswith("splitblock") else 0):
        assert b.dll.azsp_resblock_split(xs.data_ptr(), wp.data_ptr(), bias.data_ptr(), wp.data_ptr(), bias.data_ptr(), ys.data_ptr(), B, S, C, None, None) == 0
    for i in range(0 if fam.startswith("splitblock") else N):
        assert b.dll.azsp_conv3x3_split(xs.data_ptr(), wp.data_ptr(), bias.data_ptr(), rs.data_ptr() if i % 2 else None, ys.data_ptr(), B, S, C, 1, None, None) == 0
else:
    n = b.dll.azsp_tiled_bytes(B, S, C) // 2
    xs, rs, ys = (torch.zeros(n, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    for dst in (xs, rs):
        t = acts().to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
        assert b.dll.azsp_tile_layout(t.data_ptr(), dst.data_ptr(), B, S, C, 1, None) == 0
        del t
    wp = w.to(torch.bfloat16).permute(2, 3, 0, 1).reshape(9, C, C).contiguous().cuda()
    for i in range(N):
        assert b.dll.azsp_conv3x3_tiled(xs.data_ptr(), wp.data_ptr(), bias.data_ptr(), rs.data_ptr() if i % 2 else None, ys.data_ptr(), B, S, C, 1, None) == 0
torch.cuda.synchronize()
ev, mx = (0, 0.0)
print("launched", N, fam, "boards", B)
