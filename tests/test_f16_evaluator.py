"""The f16 variants of the bf16 evaluator kernels (include/azsp.h: azsp_conv3x3_tiled_f16, azsp_stem_tiled_f16, azsp_head_tiled_f16,
azsp_fc_heads_f16; AZSP_FEAT_F16_TILED): same layouts and kernels (alpha_zero_amd/csrc/az_conv.h CvFmt), f16 elements -- three more
significand bits than bf16 at the same MFMA rate.  Reference = the fp32 torch network / convolution (core/network.py:85-173)."""
import json
import os

import numpy as np
import pytest
import torch

from alpha_zero_amd.core.network import AlphaZeroNet, InferenceNet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _conv_check(bnd, boards, device, relu=1):
    """azsp_conv3x3_tiled_f16 on the tiled layout vs an fp32 torch convolution of the same f16 operands (9x9, 128 channels)."""
    C, S = 128, 9
    g = torch.Generator().manual_seed(200 + boards)
    x = torch.randn(boards, C, S, S, generator=g).to(torch.float16).to(device).contiguous(memory_format=torch.channels_last)
    res = torch.randn(boards, C, S, S, generator=g).to(torch.float16).to(device).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 3, 3, generator=g) * 0.05).to(torch.float16).to(device)
    bias = torch.randn(C, generator=g).to(device)
    wp = w.permute(2, 3, 0, 1).reshape(9, C, C).contiguous()
    n = bnd.dll.azsp_tiled_bytes(boards, S, C) // 2
    xt, rt, yt = (torch.zeros(n, dtype=torch.float16, device=device) for _ in range(3))
    assert bnd.dll.azsp_tile_layout(x.data_ptr(), xt.data_ptr(), boards, S, C, 1, None) == 0
    assert bnd.dll.azsp_tile_layout(res.data_ptr(), rt.data_ptr(), boards, S, C, 1, None) == 0
    worst = 0.0
    for r, rtile in ((None, None), (res, rt)):
        assert bnd.dll.azsp_conv3x3_tiled_f16(xt.data_ptr(), wp.data_ptr(), bias.data_ptr(), rtile.data_ptr() if rtile is not None else None,
                                              yt.data_ptr(), boards, S, C, relu, None) == 0
        y = torch.empty_like(x)
        assert bnd.dll.azsp_tile_layout(yt.data_ptr(), y.data_ptr(), boards, S, C, 0, None) == 0
        if device != "cpu":
            torch.cuda.synchronize()
        ref = torch.nn.functional.conv2d(x.float(), w.float(), bias.float(), padding=1)
        if r is not None:
            ref = ref + r.float()
        if relu:
            ref = torch.relu(ref)
        err = (y.float() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
        assert err <= 1.0 / 1024, err  # one f16 rounding of the result (2^-11 relative) + fp32 accumulation order
        worst = max(worst, err)
    return worst


def test_f16_conv_abi_host_twin():
    """azsp_conv3x3_tiled_f16 through the ABI on the host twin (plain loops with f16 rounding); other shapes are refused."""
    import engine_util as eu

    b = eu.hosttwin_binding()
    _conv_check(b, 1, "cpu")
    z = torch.zeros(4096, dtype=torch.float16)
    assert b.dll.azsp_conv3x3_tiled_f16(z.data_ptr(), z.data_ptr(), z.float().data_ptr(), None, z.clone().data_ptr(), 1, 17, 64, 1, None) != 0


def _net(blocks):
    torch.manual_seed(6)
    net = AlphaZeroNet((17, 9, 9), 82, blocks, 128, 128).eval()
    with torch.no_grad():  # a random-init net has logits of +-19: shrink the last layers so that softmax is well conditioned
        net.policy_head[4].weight.mul_(0.2)
        net.value_head[6].weight.mul_(0.3)
    return net


def test_f16_evaluator_host_twin():
    """Stem -> block -> heads -> fully connected layers of the f16 variants through InferenceNet.forward_tiled on the host twin,
    fed the AZSP_FEAT_F16_TILED encoding, vs the fp32 module."""
    import engine_util as eu

    net = _net(1)
    inf = InferenceNet(net, dtype=torch.float16, binding=eu.hosttwin_binding())
    x = (torch.rand(2, 17, 9, 9, generator=torch.Generator().manual_seed(3)) > 0.6).float()
    pri, v = inf.forward_tiled(eu.tile_features(x, torch.float16), 2, 9)
    with torch.no_grad():
        lg, vr = net(x)
    dp, dv = (pri - torch.softmax(lg, -1)).abs().max().item(), (v - vr.squeeze(1)).abs().max().item()
    assert dp <= 1e-3 and dv <= 3e-3, (dp, dv)


@pytest.mark.gpu
@pytest.mark.parametrize("boards", [1, 3, 7, 768, 770, 1539])
def test_gpu_f16_conv3x3_matches_torch(boards):
    """k_conv3x3_tiled<RES, 16, F16 = true> vs an fp32 torch convolution of the same f16 operands; 1..many tiles per workgroup, partial
    last tiles, with and without ReLU."""
    from alpha_zero_amd import _lib

    bnd = _lib.load()
    _conv_check(bnd, boards, "cuda")
    if boards in (7, 770):
        _conv_check(bnd, boards, "cuda", relu=0)


@pytest.mark.gpu
def test_gpu_f16_evaluator_vs_fp32_and_bf16():
    """The whole 10 x 128 evaluator on the f16 variants against the fp32 module, with the bf16 kernels beside it on the same
    positions (a sharp random-init network: every path's error is amplified through the 10 blocks).  Measured on MI355X (r03): f16
    max |dprior| 0.0080, |dvalue| 0.0074, arg-max agreement 100 %; bf16 0.0625 / 0.093 / 98.3 %.  The hand-written kernels pick no
    algorithm at run time and the checker is the fp32 module on the CPU, so these numbers are the same on every box (two boxes:
    profiles/r04_bounds_two_boxes.txt).  Bound = measured + 25 %: f16 within 1.0e-2, and at most 0.3 x the bf16 distance."""
    import engine_util as eu
    from alpha_zero_amd import _lib

    net = _net(10)
    x = (torch.rand(300, 17, 9, 9, generator=torch.Generator().manual_seed(4)) > 0.6).float()
    with torch.no_grad():
        lg, vr = net(x)
    pr, vr = torch.softmax(lg, -1), vr.squeeze(1)
    out = {}
    for name, dt in (("f16", torch.float16), ("bf16", torch.bfloat16)):
        inf = InferenceNet(net, dtype=dt, binding=_lib.load()).cuda()
        assert inf.supports_tiled_features(9, "cuda")
        p, v = inf.forward_tiled(eu.tile_features(x, dt).cuda(), 300, 9)
        out[name] = ((p.cpu() - pr).abs().max().item(), (v.cpu() - vr).abs().max().item(), float((p.cpu().argmax(1) == pr.argmax(1)).float().mean()))
    print(json.dumps(out))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "f16_evaluator_error.json"), "w"))
    assert out["f16"][0] <= 1.0e-2 and out["f16"][1] <= 1.0e-2, out
    assert out["f16"][0] <= 0.3 * out["bf16"][0] and out["f16"][1] <= 0.3 * out["bf16"][1], out


@pytest.mark.gpu
def test_gpu_f16_search_close_to_fp32_search_go9():
    """The search-level statement of tests/test_precision_parity.py for the f16 evaluator: same engine, positions, noise and uniforms,
    one full search per position (200 simulations, P = 8), evaluator = f16 kernels vs the library's fp32 network, random-init 10 x 128
    network (nearly flat priors: the hardest case).  Measured on MI355X (r03): top-1 0.958, moves 0.997, TV 0.020, |dQ| 0.0083; the
    bf16 kernels measure 0.893 / 0.992 / 0.084 / 0.025 on this set."""
    import test_precision_parity as tp

    torch.manual_seed(1)
    net = AlphaZeroNet((17, 9, 9), 82, 10, 128, 128)
    G, sims, P, stagger = 384, 200, 8, 40
    pa, va, ma, qa, live, tiled = tp._searched_policies(net, "go", 9, torch.float16, G, sims, P, stagger)
    assert tiled, "the f16 run must go through the hand-written tiled evaluator"
    pb, vb, mb, qb, live_b, _ = tp._searched_policies(net, "go", 9, torch.float32, G, sims, P, stagger, split_tower=False)
    assert np.array_equal(live, live_b)
    va, vb, ma, mb, qa, qb = va[live], vb[live], ma[live], mb[live], qa[live], qb[live]
    r = dict(name="go9_10x128_random_init_f16_vs_library_fp32", positions=int(live.sum()), top1_agreement=float((va.argmax(1) == vb.argmax(1)).mean()),
             move_agreement=float((ma == mb).mean()), mean_tv=float(0.5 * np.abs(va - vb).sum(1).mean()), mean_abs_root_q_diff=float(np.abs(qa - qb).mean()))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(r, open(os.path.join(ROOT, "gpurun_out", "precision_parity_f16_vs_library_fp32.json"), "w"), indent=1)
    print(json.dumps(r))
    assert r["top1_agreement"] >= 0.93 and r["move_agreement"] >= 0.99 and r["mean_tv"] <= 0.03 and r["mean_abs_root_q_diff"] <= 0.012, r
