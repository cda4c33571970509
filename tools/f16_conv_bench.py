"""k_conv3x3_tiled on f16 vs bf16 elements at the bench shape (9x9, 128 filters, 32768 rows), post-ReLU-like activations: ms per launch."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from alpha_zero_amd import _lib

b = _lib.load()
B, S, C = int(sys.argv[1]) if len(sys.argv) > 1 else 32768, 9, 128
g = torch.Generator().manual_seed(0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
out = {}
for rep in range(2):
    for name, dt, fn in (("bf16", torch.bfloat16, b.dll.azsp_conv3x3_tiled), ("f16", torch.float16, b.dll.azsp_conv3x3_tiled_f16)):
        def act():
            t = torch.randn(B, C, S, S, generator=g)
            return torch.where(torch.rand(B, C, S, S, generator=g) < 0.5, torch.zeros(()), t.abs()).to(dt).cuda().contiguous(memory_format=torch.channels_last)
        n = b.dll.azsp_tiled_bytes(B, S, C) // 2
        xt, rt, yt = (torch.zeros(n, dtype=dt, device="cuda") for _ in range(3))
        for src, dst in ((act(), xt), (act(), rt)):
            assert b.dll.azsp_tile_layout(src.data_ptr(), dst.data_ptr(), B, S, C, 1, st) == 0
        wp = (torch.randn(C, C, 3, 3, generator=g) * 0.03).to(dt).cuda().permute(2, 3, 0, 1).reshape(9, C, C).contiguous()
        bias = (torch.randn(C, generator=g) * 0.1).cuda()
        for r in (None, rt):
            for _ in range(5):
                assert fn(xt.data_ptr(), wp.data_ptr(), bias.data_ptr(), r.data_ptr() if r is not None else None, yt.data_ptr(), B, S, C, 1, st) == 0
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn(xt.data_ptr(), wp.data_ptr(), bias.data_ptr(), r.data_ptr() if r is not None else None, yt.data_ptr(), B, S, C, 1, st)
            e1.record()
            torch.cuda.synchronize()
            out[f"{name}_{'residual' if r is not None else 'plain'}_ms_pass{rep}"] = round(e0.elapsed_time(e1) / 20, 4)
        del xt, rt, yt
print(json.dumps(out))
