"""Same-box, same-process A/B of the fp32-class tower kernels of two builds of the library: the one in the tree against another build
(default tools/probes/libazsp_prev.so), timed alternately on the same post-ReLU-like activations, with a bitwise comparison of the outputs.
Cases: the 9x9 x 128 convolution (k_conv3x3_sp2, plain and with a residual), the fused 17x17 x 64 and 9x9 x 64 blocks (k_resblock_sp<Sb17 | Sb9>).
(+ the bf16 19x19 x 256 convolution, k_conv3x3_op19q).  PREV_AB_CASES=<substring> selects cases.
usage: python tools/split_prev_ab.py [other_lib] [rounds]"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alpha_zero_amd import _abi, _lib
from alpha_zero_amd.core.network import split_weights_f16

other = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "tools", "probes", "libazsp_prev.so")
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 5
libs = {"tree": _lib.load(), "other": _abi.Binding(ctypes.CDLL(other), "A/B build")}
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 32768


def make(S, C, seed):
    g = torch.Generator().manual_seed(seed)

    def act():
        t = torch.randn(B, C, S, S, generator=g)
        return torch.where(torch.rand(B, C, S, S, generator=g) < 0.5, torch.zeros(()), t.abs()).cuda().contiguous(memory_format=torch.channels_last)

    d = libs["tree"].dll
    n = d.azsp_split_bytes(B, S, C) // 2
    xs, rs = (torch.zeros(n, dtype=torch.float16, device="cuda") for _ in range(2))
    for src, dst in ((act(), xs), (act(), rs)):
        assert d.azsp_split_layout(src.data_ptr(), dst.data_ptr(), B, S, C, 1, None, st) == 0
    ws = [split_weights_f16(torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda() for _ in range(2)]
    bs = [(torch.randn(C, generator=g) * 0.1).cuda() for _ in range(2)]
    ys = {k: torch.zeros(n, dtype=torch.float16, device="cuda") for k in libs}
    torch.cuda.synchronize()
    return xs, rs, ws, bs, ys


def case_conv(S, C, residual):
    xs, rs, ws, bs, ys = make(S, C, 1)

    def run(k, reps):
        for _ in range(reps):
            assert libs[k].dll.azsp_conv3x3_split(xs.data_ptr(), ws[0].data_ptr(), bs[0].data_ptr(), rs.data_ptr() if residual else None, ys[k].data_ptr(), B, S, C, 1, None, st) == 0

    return run, ys, 2.0 * B * S * S * C * C * 9 * 3


def case_block(S, C):
    xs, rs, ws, bs, ys = make(S, C, 2)

    def run(k, reps):
        for _ in range(reps):
            assert libs[k].dll.azsp_resblock_split(xs.data_ptr(), ws[0].data_ptr(), bs[0].data_ptr(), ws[1].data_ptr(), bs[1].data_ptr(), ys[k].data_ptr(), B, S, C, None, st) == 0

    return run, ys, 2 * 2.0 * B * S * S * C * C * 9 * 3


def case_conv19(residual, boards=4096):
    """the bf16 19x19 x 256 tower convolution (BASELINE C5's dominant kernel, k_conv3x3_op19q) through azsp_conv3x3_tiled"""
    S, C = 19, 256
    d = libs["tree"].dll
    g = torch.Generator().manual_seed(3)
    n = d.azsp_tiled_bytes(boards, S, C) // 2
    xt, rt = (torch.zeros(n, dtype=torch.bfloat16, device="cuda") for _ in range(2))
    for dst in (xt, rt):
        t = torch.randn(boards, C, S, S, generator=g)
        t = torch.where(torch.rand(boards, C, S, S, generator=g) < 0.5, torch.zeros(()), t.abs()).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
        assert d.azsp_tile_layout(t.data_ptr(), dst.data_ptr(), boards, S, C, 1, None) == 0
        del t
    w = (torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).to(torch.bfloat16).cuda()
    wp = w.permute(2, 3, 0, 1).reshape(9, C, C).contiguous()
    bias = (torch.randn(C, generator=g) * 0.1).cuda()
    ys = {k: torch.zeros(n, dtype=torch.bfloat16, device="cuda") for k in libs}
    torch.cuda.synchronize()

    def run(k, reps):
        for _ in range(reps):
            assert libs[k].dll.azsp_conv3x3_tiled(xt.data_ptr(), wp.data_ptr(), bias.data_ptr(), rt.data_ptr() if residual else None, ys[k].data_ptr(), boards, S, C, 1, st) == 0

    return run, ys, 2.0 * boards * S * S * C * C * 9


CASES = (("conv 9x9 x 128 plain", lambda: case_conv(9, 128, False)), ("conv 9x9 x 128 residual", lambda: case_conv(9, 128, True)),
         ("block 17x17 x 64", lambda: case_block(17, 64)), ("block 9x9 x 64", lambda: case_block(9, 64)),
         ("conv 19x19 x 256 bf16 plain", lambda: case_conv19(False)), ("conv 19x19 x 256 bf16 residual", lambda: case_conv19(True)))
ONLY = os.environ.get("PREV_AB_CASES", "")
for name, mk in [c for c in CASES if ONLY in c[0]]:
    run, ys, flops = mk()
    for k in libs:
        run(k, 3)
    torch.cuda.synchronize()
    print(f"{name}: bit-identical outputs: {torch.equal(ys['tree'], ys['other'])}", flush=True)
    tot = {k: 0.0 for k in libs}
    for rnd in range(ROUNDS):
        for k in (("tree", "other") if rnd % 2 == 0 else ("other", "tree")):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(k, 30)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 30
            tot[k] += ms if rnd > 0 else 0.0  # (round 0 warms the clocks up)
            print(f"  round {rnd} {k:6s} {ms:7.4f} ms  frac {flops / ms / 1e9 / 2500:.4f}", flush=True)
    print(f"  mean of rounds 1.. : tree {tot['tree'] / (ROUNDS - 1):.4f} ms, other {tot['other'] / (ROUNDS - 1):.4f} ms: tree / other = {tot['tree'] / tot['other']:.4f}", flush=True)
    del run, ys
    torch.cuda.empty_cache()
