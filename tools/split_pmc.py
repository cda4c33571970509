"""A handful of azsp_conv3x3_split / azsp_split_layout launches at the bench shape (9x9, 128 filters, 32768 rows) for rocprofv3 --pmc passes."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alpha_zero_amd import _lib
from alpha_zero_amd.core.network import split_weights_f16

b = _lib.load()
B, S, C = int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 9, 128
g = torch.Generator().manual_seed(0)
n = b.dll.azsp_split_bytes(B, S, C) // 2
xs, rs, ys = (torch.zeros(n, dtype=torch.float16, device="cuda") for _ in range(3))
for dst in (xs, rs):
    t = torch.randn(B, C, S, S, generator=g)
    t = torch.where(torch.rand(B, C, S, S, generator=g) < 0.5, torch.zeros(()), t.abs()).cuda().contiguous(memory_format=torch.channels_last)
    assert b.dll.azsp_split_layout(t.data_ptr(), dst.data_ptr(), B, S, C, 1, None, None) == 0
    del t
wsp = split_weights_f16(torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
bias = (torch.randn(C, generator=g) * 0.1).cuda()
for i in range(6):
    assert b.dll.azsp_conv3x3_split(xs.data_ptr(), wsp.data_ptr(), bias.data_ptr(), rs.data_ptr() if i % 2 else None, ys.data_ptr(), B, S, C, 1, None, None) == 0
torch.cuda.synchronize()
