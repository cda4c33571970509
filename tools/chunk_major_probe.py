"""Does the ORDER of the tower's launches matter?  The product runs the residual tower layer-major: every layer over all G*P boards (at the
bench shape 32768 boards x 41.5 KB = 1.36 GB per activation tensor: every layer streams its input from HBM and its output to HBM).  Chunk-major
(all layers over a chunk of boards, then the next chunk) keeps a chunk's three activation tensors inside the 256 MB memory-side cache (and, for
very small chunks, the 8 x 4 MB L2s) at the price of more, shorter launches (each reloads its weight registers and refills its pipeline).
Times both orders of the same 20 launches-per-board-chunk through azsp_conv3x3_split on post-ReLU-like activations; the outputs must be
bit-identical (the kernels see the same boards).  usage: python tools/chunk_major_probe.py [blocks = 10] [boards = 32768]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alpha_zero_amd import _lib
from alpha_zero_amd.core.network import split_weights_f16

NB = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
S, C = 9, 128
d = _lib.load().dll
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator().manual_seed(0)
t = torch.randn(B, C, S, S, generator=g)
x0 = torch.where(torch.rand(B, C, S, S, generator=g) < 0.5, torch.zeros(()), t.abs()).cuda().contiguous(memory_format=torch.channels_last)
# a residual tower that keeps its activations O(1): He-scaled first convolution, small second one
ws = [split_weights_f16(torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5 * (1.0 if i % 2 == 0 else 0.3)).cuda() for i in range(2 * NB)]
bs = [(torch.randn(C, generator=g) * 0.05).cuda() for _ in range(2 * NB)]
per_board = d.azsp_split_bytes(1, S, C)
n = d.azsp_split_bytes(B, S, C) // 2
xin, a, m, o = (torch.zeros(n, dtype=torch.float16, device="cuda") for _ in range(4))
assert d.azsp_split_layout(x0.data_ptr(), xin.data_ptr(), B, S, C, 1, None, st) == 0
del x0, t
flops = 2 * NB * 2.0 * B * S * S * C * C * 9 * 3


def tower(chunk):
    """the tower over boards [c0, c0 + chunk) for every chunk; buffers a (in / out), m, o rotate as the product's do"""
    for c0 in range(0, B, chunk):
        nb = min(chunk, B - c0)
        off = c0 * per_board
        pa, pm, po = a.data_ptr() + off, m.data_ptr() + off, o.data_ptr() + off
        for i in range(NB):
            assert d.azsp_conv3x3_split(pa, ws[2 * i].data_ptr(), bs[2 * i].data_ptr(), None, pm, nb, S, C, 1, None, st) == 0
            assert d.azsp_conv3x3_split(pm, ws[2 * i + 1].data_ptr(), bs[2 * i + 1].data_ptr(), pa, po, nb, S, C, 1, None, st) == 0
            pa, po = po, pa
    return NB % 2  # 1: the result is in o


def run(chunk, reps):
    for _ in range(reps):
        a.copy_(xin)
        where = tower(chunk)
    return o if where else a


ref = run(B, 1).clone()
torch.cuda.synchronize()
print(f"9x9 x 128, {NB} blocks, {B} boards; copy of the input included in every timing", flush=True)
chunks = [B, 16384, 8192, 4096, 2048, 1024, 512]
res = {c: [] for c in chunks}
for c in chunks:
    out = run(c, 1)
    torch.cuda.synchronize()
    print(f"chunk {c:6d}: bit-identical to layer-major: {torch.equal(out, ref)}", flush=True)
for rnd in range(4):
    for c in (chunks if rnd % 2 == 0 else chunks[::-1]):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(c, 3)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        if rnd > 0:
            res[c].append(ms)
        print(f"  round {rnd} chunk {c:6d} {ms:8.3f} ms per tower  frac {flops / ms / 1e9 / 2500:.4f}", flush=True)
base = sum(res[B]) / len(res[B])
for c in chunks:
    v = sum(res[c]) / len(res[c])
    print(f"chunk {c:6d} ({c * per_board * 3 / 2**20:7.1f} MB in three tensors): {v:8.3f} ms  {v / base:.4f} x layer-major")
