"""A/B of the 19x19 x 256 tower convolution (BASELINE C5's dominant kernel) on the same box: the one-pass kernel (round 6,
k_conv3x3_op19) against rounds 2-5's two-launch scheme (k_conv3x3_hb19, selected with AZSP_CONV19_TWO_LAUNCH=1 -- the library reads
the switch once per process, so each variant runs in its own process).  Post-ReLU-like activations (half zeros), alternating plain /
residual layers like the forward.  usage: python tools/conv19_ab.py [boards]   (no argument: both variants, as sub-processes)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(B):
    import torch

    sys.path.insert(0, ROOT)
    from alpha_zero_amd import _lib

    b = _lib.load()
    S, C = 19, 256
    g = torch.Generator().manual_seed(0)

    def acts():
        t = torch.randn(B, C, S, S, generator=g)
        return torch.where(torch.rand(B, C, S, S, generator=g) < 0.5, torch.zeros(()), t.abs()).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)

    n = b.dll.azsp_tiled_bytes(B, S, C) // 2
    xt, rt, yt = (torch.zeros(n, dtype=torch.bfloat16, device="cuda") for _ in range(3))
    for dst in (xt, rt):
        t = acts()
        assert b.dll.azsp_tile_layout(t.data_ptr(), dst.data_ptr(), B, S, C, 1, None) == 0
        del t
    w = (torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).to(torch.bfloat16).cuda()
    wp = w.permute(2, 3, 0, 1).reshape(9, C, C).contiguous()
    bias = (torch.randn(C, generator=g) * 0.1).cuda()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    flops = 2.0 * B * S * S * C * C * 9
    out = {}
    for name, r in (("plain", None), ("residual", rt)):
        for _ in range(4):
            assert b.dll.azsp_conv3x3_tiled(xt.data_ptr(), wp.data_ptr(), bias.data_ptr(), r.data_ptr() if r is not None else None, yt.data_ptr(), B, S, C, 1, st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            b.dll.azsp_conv3x3_tiled(xt.data_ptr(), wp.data_ptr(), bias.data_ptr(), r.data_ptr() if r is not None else None, yt.data_ptr(), B, S, C, 1, st)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        out[name] = ms
        print(f"  {name:9s} {ms:7.3f} ms per convolution  {flops / ms / 1e9:7.1f} TFLOP/s  frac of the 2.5 PF/s bf16 peak {flops / ms / 1e9 / 2500:.4f}", flush=True)
    mean = (out["plain"] + out["residual"]) / 2
    print(f"  mean      {mean:7.3f} ms  frac {flops / mean / 1e9 / 2500:.4f}", flush=True)


if __name__ == "__main__":
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    if os.environ.get("CONV19_AB_CHILD"):
        one(B)
    else:
        for rep in range(2):
            variants = [("one pass (k_conv3x3_op19)" + (f", variant {v}" if v else ""), ({"AZSP_OP19_VARIANT": v} if v else {})) for v in os.environ.get("CONV19_AB_VARIANTS", "").split(",")]
            for label, extra in variants + [("two launches (k_conv3x3_hb19)", {"AZSP_CONV19_TWO_LAUNCH": "1"})]:
                print(f"{label}, {B} boards, run {rep}:", flush=True)
                env = {k: v for k, v in os.environ.items() if k not in ("AZSP_CONV19_TWO_LAUNCH", "AZSP_OP19_VARIANT")}
                env.update(CONV19_AB_CHILD="1", **extra)
                subprocess.run([sys.executable, os.path.abspath(__file__), str(B)], env=env, check=False)
