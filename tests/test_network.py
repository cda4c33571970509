"""AlphaZeroNet must be state_dict-compatible with the reference module and reproduce its outputs;
InferenceNet (BN folded, channels-last) must agree with it (fp32 1e-4; bf16 on the GPU tier)."""
import os

import pytest
import torch

from alpha_zero_amd.core.network import AlphaZeroNet, InferenceNet

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["go", "gomoku"])
def test_network_matches_reference_outputs(golden_dir, name):
    d = torch.load(os.path.join(golden_dir, f"net_{name}.pt"), weights_only=False)
    net = AlphaZeroNet(**d["args"])
    assert list(net.state_dict().keys()) == list(d["state_dict"].keys())
    net.load_state_dict(d["state_dict"])
    net.eval()
    with torch.no_grad():
        logits, value = net(d["x"].float())
    assert torch.allclose(logits, d["logits"], atol=1e-5) and torch.allclose(value, d["value"], atol=1e-6)
    inf = InferenceNet(net, dtype=torch.float32)
    pri, v = inf(d["x"])
    assert torch.allclose(pri, torch.softmax(d["logits"], -1), atol=1e-5)
    assert torch.allclose(v, d["value"].squeeze(1), atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["go", "gomoku"])
@pytest.mark.parametrize("dtype,tol_p,tol_v", [(torch.float32, 1e-4, 1e-4), (torch.bfloat16, 2e-2, 3e-2), (torch.float16, 5e-3, 5e-3)])
def test_gpu_inference_net(golden_dir, name, dtype, tol_p, tol_v):
    d = torch.load(os.path.join(golden_dir, f"net_{name}.pt"), weights_only=False)
    net = AlphaZeroNet(**d["args"])
    net.load_state_dict(d["state_dict"])
    inf = InferenceNet(net, dtype=dtype).cuda()
    pri, v = inf(d["x"].cuda())
    assert (pri.cpu() - torch.softmax(d["logits"], -1)).abs().max() <= tol_p
    assert (v.cpu() - d["value"].squeeze(1)).abs().max() <= tol_v


def test_fused_epilogue_matches_torch_host_twin():
    """azsp_bias_act (host twin build of the same source): relu(y + b [+ res]) for fp32 / bf16 / fp16."""
    import engine_util as eu

    b = eu.hosttwin_binding()
    g = torch.Generator().manual_seed(0)
    for dt, code, tol in ((torch.float32, 1, 0.0), (torch.bfloat16, 2, 1e-2), (torch.float16, 3, 2e-3)):
        y = torch.randn(37 * 81, 128, generator=g).to(dt)
        bias = torch.randn(128, generator=g).to(dt)
        res = torch.randn(37 * 81, 128, generator=g).to(dt)
        for r in (None, res):
            out = y.clone()
            rc = b.dll.azsp_bias_act(out.data_ptr(), bias.data_ptr(), r.data_ptr() if r is not None else None, out.shape[0], 128, code, 1, None)
            assert rc == 0
            ref = torch.relu(y.float() + bias.float() + (r.float() if r is not None else 0.0))
            assert (out.float() - ref).abs().max() <= tol * max(1.0, ref.abs().max().item())


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol_p,tol_v", [(torch.float32, 1e-4, 1e-4), (torch.bfloat16, 2e-2, 3e-2)])
def test_gpu_inference_net_fused_epilogue(golden_dir, dtype, tol_p, tol_v):
    from alpha_zero_amd import _lib

    d = torch.load(os.path.join(golden_dir, "net_go.pt"), weights_only=False)
    net = AlphaZeroNet(**d["args"])
    net.load_state_dict(d["state_dict"])
    inf = InferenceNet(net, dtype=dtype, binding=_lib.load()).cuda()
    pri, v = inf(d["x"].cuda())
    assert (pri.cpu() - torch.softmax(d["logits"], -1)).abs().max() <= tol_p
    assert (v.cpu() - d["value"].squeeze(1)).abs().max() <= tol_v


def _conv_ref(x, w, b, res):
    y = torch.nn.functional.conv2d(x.float(), w.float(), b.float(), padding=1)
    if res is not None:
        y = y + res.float()
    return torch.relu(y)


def _tiled_roundtrip_and_conv(bnd, boards, C, S, device, relu=1, tol=1.0 / 128):
    """Shared by the host-twin and GPU tiers: layout round trip exact; tiled conv == fp32 torch conv within bf16 rounding."""
    g = torch.Generator().manual_seed(100 + boards)
    x = torch.randn(boards, C, S, S, generator=g).to(torch.bfloat16).to(device).contiguous(memory_format=torch.channels_last)
    res = torch.randn(boards, C, S, S, generator=g).to(torch.bfloat16).to(device).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(C, C, 3, 3, generator=g) * (0.05 if C >= 64 else 0.2)).to(torch.bfloat16).to(device)
    bias = torch.randn(C, generator=g).to(device)
    wp = w.permute(2, 3, 0, 1).reshape(9, C, C).contiguous()
    n = bnd.dll.azsp_tiled_bytes(boards, S, C) // 2
    tb = max(1, 256 // (S * S))  # boards per tile
    assert n == (boards + tb - 1) // tb * tb * S * S * C
    xt, rt, yt = (torch.zeros(n, dtype=torch.bfloat16, device=device) for _ in range(3))
    assert bnd.dll.azsp_tile_layout(x.data_ptr(), xt.data_ptr(), boards, S, C, 1, None) == 0
    assert bnd.dll.azsp_tile_layout(res.data_ptr(), rt.data_ptr(), boards, S, C, 1, None) == 0
    back = torch.empty_like(x)
    assert bnd.dll.azsp_tile_layout(xt.data_ptr(), back.data_ptr(), boards, S, C, 0, None) == 0
    if device != "cpu":
        torch.cuda.synchronize()
    assert torch.equal(back, x)
    # the layout itself: [tile][C/8][tb*S*S][8]
    rows = x.permute(0, 2, 3, 1).reshape(boards * S * S, C // 8, 8)
    full = torch.zeros((boards + tb - 1) // tb * tb * S * S, C // 8, 8, dtype=torch.bfloat16, device=device)
    full[: rows.shape[0]] = rows
    assert torch.equal(xt.view(-1, C // 8, tb * S * S, 8), full.view(-1, tb * S * S, C // 8, 8).permute(0, 2, 1, 3))
    for r, rtile in ((None, None), (res, rt)):
        rc = bnd.dll.azsp_conv3x3_tiled(xt.data_ptr(), wp.data_ptr(), bias.data_ptr(), rtile.data_ptr() if rtile is not None else None,
                                        yt.data_ptr(), boards, S, C, relu, None)
        assert rc == 0
        y = torch.empty_like(x)
        assert bnd.dll.azsp_tile_layout(yt.data_ptr(), y.data_ptr(), boards, S, C, 0, None) == 0
        if device != "cpu":
            torch.cuda.synchronize()
        ref = torch.nn.functional.conv2d(x.float(), w.float(), bias.float(), padding=1)
        if r is not None:
            ref = ref + r.float()
        if relu:
            ref = torch.relu(ref)
        err = (y.float() - ref).abs().max().item()
        assert err <= tol * max(1.0, ref.abs().max().item()), (err, tol)


def test_tiled_conv_abi_host_twin():
    """azsp_tile_layout / azsp_conv3x3_tiled through the ABI on the host twin (plain-loop restatement), tiny shapes."""
    import engine_util as eu

    b = eu.hosttwin_binding()
    for boards in (1, 3, 4):
        _tiled_roundtrip_and_conv(b, boards, 16, 5, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("boards", [1, 2, 3, 7, 768, 770, 1000, 1539])
def test_gpu_tiled_conv3x3_matches_torch(boards):
    """The weight-stationary MFMA kernel on the tiled layout vs an fp32 torch convolution of the same bf16 operands
    (asymmetric random weights; 1..many tiles per workgroup; partial last tiles; with and without ReLU)."""
    from alpha_zero_amd import _lib

    bnd = _lib.load()
    _tiled_roundtrip_and_conv(bnd, boards, 128, 9, "cuda")
    if boards in (7, 770):  # no ReLU (identity epilogue), also with a second tile per workgroup
        _tiled_roundtrip_and_conv(bnd, boards, 128, 9, "cuda", relu=0)


@pytest.mark.gpu
@pytest.mark.parametrize("boards", [1, 2, 5, 31, 32, 33, 70, 130, 257])
def test_gpu_tiled_conv3x3_go19_256_matches_torch(boards):
    """BASELINE C5 shape (19x19 planes x 256 filters, training_go_jumbo.py:46): the ONE-PASS half-board weight-stationary kernel
    (round 6: 64 couts x all 256 input channels per CU, the cin halves meet through LDS, fp32 accumulation end to end, the image refilled
    band by band) vs an fp32 torch convolution.  1..9 tiles per tile stream (32 streams on a full device), with / without residual and
    ReLU.  Bound: ONE bf16 rounding of the result (half an ulp <= 2^-8 of the value: 8 significand bits) + fp32 accumulation slack --
    rounds 2-5's two-launch scheme rounded the partial sum to bf16 as well and needed 2^-7."""
    from alpha_zero_amd import _lib

    bnd = _lib.load()
    tol = 2.0 ** -8 * 1.05
    _tiled_roundtrip_and_conv(bnd, boards, 256, 19, "cuda", tol=tol)
    if boards in (5, 70):
        _tiled_roundtrip_and_conv(bnd, boards, 256, 19, "cuda", relu=0, tol=tol)


@pytest.mark.gpu
def test_gpu_tiled_conv3x3_go19_256_one_pass_vs_the_two_launch_scheme():
    """The one-pass kernel against rounds 2-5's two-launch scheme (still in the library behind AZSP_CONV19_TWO_LAUNCH, read once per
    process: a sub-process) on the same inputs: both within their bounds of fp32 torch, the one-pass result at least as close."""
    import json
    import subprocess
    import sys

    src = """
import json, os, sys, torch
sys.path.insert(0, %r)
from alpha_zero_amd import _lib
b = _lib.load()
B, S, C = 37, 19, 256
g = torch.Generator().manual_seed(5)
x = torch.randn(B, C, S, S, generator=g).abs().to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
r = torch.randn(B, C, S, S, generator=g).to(torch.bfloat16).cuda().contiguous(memory_format=torch.channels_last)
w = (torch.randn(C, C, 3, 3, generator=g) * 0.03).to(torch.bfloat16).cuda()
bias = torch.randn(C, generator=g).cuda()
wp = w.permute(2, 3, 0, 1).reshape(9, C, C).contiguous()
n = b.dll.azsp_tiled_bytes(B, S, C) // 2
xt, rt, yt = (torch.zeros(n, dtype=torch.bfloat16, device="cuda") for _ in range(3))
assert b.dll.azsp_tile_layout(x.data_ptr(), xt.data_ptr(), B, S, C, 1, None) == 0
assert b.dll.azsp_tile_layout(r.data_ptr(), rt.data_ptr(), B, S, C, 1, None) == 0
assert b.dll.azsp_conv3x3_tiled(xt.data_ptr(), wp.data_ptr(), bias.data_ptr(), rt.data_ptr(), yt.data_ptr(), B, S, C, 1, None) == 0
y = torch.empty_like(x)
assert b.dll.azsp_tile_layout(yt.data_ptr(), y.data_ptr(), B, S, C, 0, None) == 0
torch.cuda.synchronize()
ref = torch.relu(torch.nn.functional.conv2d(x.float(), w.float(), bias, padding=1) + r.float())
print(json.dumps({"err": (y.float() - ref).abs().max().item(), "mean_err": (y.float() - ref).abs().mean().item(), "scale": ref.abs().max().item()}))
""" % ROOT
    res = {}
    for name, extra in (("one_pass", {}), ("two_launch", {"AZSP_CONV19_TWO_LAUNCH": "1"})):
        env = {k: v for k, v in os.environ.items() if k != "AZSP_CONV19_TWO_LAUNCH"}
        env.update(extra)
        out = subprocess.run([sys.executable, "-c", src], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        res[name] = json.loads(out.stdout.strip().splitlines()[-1])
    s1, s2 = res["one_pass"], res["two_launch"]
    assert s1["err"] <= 2.0 ** -8 * 1.05 * max(1.0, s1["scale"]) and s2["err"] <= 1.0 / 128 * max(1.0, s2["scale"]), res
    assert s1["mean_err"] <= s2["mean_err"] * 1.001, res  # one rounding instead of two


@pytest.mark.gpu
@pytest.mark.parametrize("boards", [1, 3, 66])
def test_gpu_stem_head_tiled_go19_256_match_torch(boards):
    from alpha_zero_amd import _lib

    _stem_head_checks(_lib.load(), boards, 256, 19, "cuda")


@pytest.mark.gpu
def test_gpu_go19_256_network_tiled_forward_matches_fp32():
    """The whole C5-shaped evaluator (stem, tower, heads, FC layers; fewer blocks) on the hand-written kernels vs the fp32 module."""
    import engine_util as eu
    from alpha_zero_amd import _lib

    torch.manual_seed(3)
    net = AlphaZeroNet((17, 19, 19), 362, 3, 256, 256)
    with torch.no_grad():
        net.policy_head[4].weight.mul_(0.2)
    inf = InferenceNet(net, dtype=torch.bfloat16, binding=_lib.load()).cuda()
    assert inf.supports_tiled_features(19, "cuda")
    x = (torch.rand(9, 17, 19, 19) > 0.6).float()
    pri, v = inf.forward_tiled(eu.tile_features(x).cuda(), 9, 19)
    logits, vr = net.eval()(x)
    dp, dv = (pri.cpu() - torch.softmax(logits, -1)).abs().max().item(), (v.cpu() - vr.squeeze(1)).abs().max().item()
    assert dp <= 2e-2 and dv <= 3e-2, (dp, dv)
    p2, v2 = inf(x.cuda())  # NCHW entry: library stem, tiled tower
    assert (p2.cpu() - torch.softmax(logits, -1)).abs().max().item() <= 2e-2 and (v2.cpu() - vr.squeeze(1)).abs().max().item() <= 3e-2


def _stem_head_checks(bnd, boards, C, S, device, pad=1):
    """azsp_stem_tiled / azsp_head_tiled vs torch on the same bf16 operands (pad = 3: the Gomoku stem, planes grow to S + 4)."""
    import engine_util as eu

    g = torch.Generator().manual_seed(7 + boards)
    x = (torch.rand(boards, 17, S, S, generator=g) > 0.6).float()
    w = (torch.randn(C, 17, 3, 3, generator=g) * 0.2).to(torch.bfloat16)
    bias = torch.randn(C, generator=g)
    wp = torch.zeros(9, C, 32)
    wp[:, :, :17] = w.float().permute(2, 3, 0, 1).reshape(9, C, 17)
    wp = wp.to(torch.bfloat16).contiguous().to(device)
    feat = eu.tile_features(x).to(device)
    So = S + 2 * (pad - 1)
    n = bnd.dll.azsp_tiled_bytes(boards, So, C) // 2
    yt = torch.zeros(n, dtype=torch.bfloat16, device=device)
    assert bnd.dll.azsp_stem_tiled(feat.data_ptr(), wp.data_ptr(), bias.to(device).data_ptr(), yt.data_ptr(), boards, S, C, pad, 1, None) == 0
    y = torch.empty(boards, C, So, So, dtype=torch.bfloat16, device=device).contiguous(memory_format=torch.channels_last)
    assert bnd.dll.azsp_tile_layout(yt.data_ptr(), y.data_ptr(), boards, So, C, 0, None) == 0
    ref = torch.relu(torch.nn.functional.conv2d(x, w.float(), bias, padding=pad))
    if device != "cpu":
        torch.cuda.synchronize()
    assert (y.float().cpu() - ref).abs().max().item() <= 1.0 / 128 * max(1.0, ref.abs().max().item())
    # heads on the stem output
    hw = torch.randn(3, C, generator=g) * 0.2
    hb = torch.randn(3, generator=g)
    pol = torch.empty(boards, 2 * So * So, dtype=torch.bfloat16, device=device)
    val = torch.empty(boards, So * So, dtype=torch.bfloat16, device=device)
    assert bnd.dll.azsp_head_tiled(yt.data_ptr(), hw.to(device).data_ptr(), hb.to(device).data_ptr(), pol.data_ptr(), val.data_ptr(), boards, So, C,
                                   2, 1, 0, 0, None) == 0
    if device != "cpu":
        torch.cuda.synchronize()
    h = torch.relu(torch.nn.functional.conv2d(y.float().cpu(), hw.view(3, C, 1, 1), hb))
    tol = 1.0 / 128 * max(1.0, h.abs().max().item())
    assert (pol.float().cpu() - h[:, :2].reshape(boards, -1)).abs().max().item() <= tol
    assert (val.float().cpu() - h[:, 2:].reshape(boards, -1)).abs().max().item() <= tol


def test_stem_head_abi_host_twin():
    import engine_util as eu

    for boards in (1, 4):
        _stem_head_checks(eu.hosttwin_binding(), boards, 16, 5, "cpu")
    _stem_head_checks(eu.hosttwin_binding(), 3, 16, 5, "cpu", pad=3)


@pytest.mark.gpu
@pytest.mark.parametrize("boards", [1, 5, 770])
def test_gpu_stem_head_tiled_match_torch(boards):
    from alpha_zero_amd import _lib

    _stem_head_checks(_lib.load(), boards, 128, 9, "cuda")
    _stem_head_checks(_lib.load(), boards, 64, 13, "cuda", pad=3)  # the Gomoku stem / heads (17x17 planes)
    _stem_head_checks(_lib.load(), boards, 64, 9, "cuda")  # 9x9 Go with 64 filters (three boards per tile)


@pytest.mark.gpu
def test_gpu_forward_tiled_equals_forward():
    """Whole evaluator on tiled features (stem, tower and head kernels) vs the same network fed NCHW planes."""
    import engine_util as eu
    from alpha_zero_amd import _lib

    torch.manual_seed(6)
    net = AlphaZeroNet((17, 9, 9), 82, 3, 128, 64)
    with torch.no_grad():  # a random-init net has logits of +-19: shrink the last layers so that softmax is well conditioned
        net.policy_head[4].weight.mul_(0.2)
        net.value_head[6].weight.mul_(0.3)
    inf = InferenceNet(net, dtype=torch.bfloat16, binding=_lib.load()).cuda()
    assert inf.supports_tiled_features(9, "cuda")
    x = (torch.rand(100, 17, 9, 9) > 0.6).float()
    p1, v1 = inf.forward_tiled(eu.tile_features(x).cuda(), 100, 9)
    p2, v2 = inf(x.cuda())
    assert (p1 - p2).abs().max().item() <= 1e-2 and (v1 - v2).abs().max().item() <= 2e-2
    ref_logits, ref_v = net.eval()(x)
    assert (p1.cpu() - torch.softmax(ref_logits, -1)).abs().max().item() <= 2e-2 and (v1.cpu() - ref_v.squeeze(1)).abs().max().item() <= 3e-2


@pytest.mark.gpu
def test_gpu_forward_tiled_full_batch_is_batch_independent():
    """BASELINE size (G*P = 32 768 rows): a position's evaluation must not depend on what else is in the batch, which tile or
    which workgroup it lands in.  Rows evaluated inside the full batch equal the same rows evaluated alone, bit for bit, and
    duplicated positions get identical outputs wherever they sit."""
    import engine_util as eu
    from alpha_zero_amd import _lib

    torch.manual_seed(8)
    net = AlphaZeroNet((17, 9, 9), 82, 10, 128, 128)
    inf = InferenceNet(net, dtype=torch.bfloat16, binding=_lib.load()).cuda()
    rows = 32768
    x = (torch.rand(rows, 17, 9, 9) > 0.6).float()
    x[20000:20100] = x[:100]          # duplicates far away (other tiles, other workgroups, other tile phase: 20000 % 3 == 2)
    x[rows - 50:] = x[100:150]        # ... and in the ragged last tile
    feat = eu.tile_features(x).cuda()
    p_all, v_all = (t.clone() for t in inf.forward_tiled(feat, rows, 9))
    assert torch.isfinite(p_all).all() and torch.isfinite(v_all).all() and torch.allclose(p_all.sum(1), torch.ones(rows, device="cuda"), atol=1e-3)
    assert torch.equal(p_all[20000:20100], p_all[:100]) and torch.equal(v_all[20000:20100], v_all[:100])
    assert torch.equal(p_all[rows - 50:], p_all[100:150]) and torch.equal(v_all[rows - 50:], v_all[100:150])
    p_sub, v_sub = inf.forward_tiled(eu.tile_features(x[:301]).cuda(), 301, 9)
    assert torch.equal(p_sub, p_all[:301]) and torch.equal(v_sub, v_all[:301])


@pytest.mark.gpu
@pytest.mark.parametrize("boards", [1, 2, 5, 256, 300, 1000])
def test_gpu_tiled_conv3x3_gomoku_shape_matches_torch(boards):
    """k_conv3x3_t64 (17x17 planes, 64 filters: the 13x13 Gomoku tower) vs an fp32 torch convolution of the same bf16 operands."""
    from alpha_zero_amd import _lib

    bnd = _lib.load()
    _tiled_roundtrip_and_conv(bnd, boards, 64, 17, "cuda")
    if boards == 5:
        _tiled_roundtrip_and_conv(bnd, boards, 64, 17, "cuda", relu=0)


@pytest.mark.gpu
def test_gpu_gomoku_network_tiled_tower():
    """The 13x13 Gomoku network (pad-3 stem -> 17x17 planes, 64 filters): tower on k_conv3x3_t64 vs the library-convolution path
    of the same InferenceNet and vs the fp32 module."""
    from alpha_zero_amd import _lib

    torch.manual_seed(9)
    net = AlphaZeroNet((17, 13, 13), 169, 3, 64, 64, gomoku=True)
    with torch.no_grad():
        net.policy_head[4].weight.mul_(0.2)
        net.value_head[6].weight.mul_(0.3)
    inf = InferenceNet(net, dtype=torch.bfloat16, binding=_lib.load()).cuda()
    x = (torch.rand(70, 17, 13, 13) > 0.6).float()
    p1, v1 = inf(x.cuda())
    inf.use_tiled_tower = False
    p2, v2 = inf(x.cuda())
    assert (p1 - p2).abs().max().item() <= 1e-2 and (v1 - v2).abs().max().item() <= 2e-2
    logits, vr = net.eval()(x)
    assert (p1.cpu() - torch.softmax(logits, -1)).abs().max().item() <= 2e-2 and (v1.cpu() - vr.squeeze(1)).abs().max().item() <= 3e-2
    # ... and the whole evaluator on the engine's tiled 13x13 features: pad-3 stem, tower and heads in hand-written kernels
    import engine_util as eu

    inf.use_tiled_tower = True
    assert inf.supports_tiled_features(13, "cuda")
    p3, v3 = inf.forward_tiled(eu.tile_features(x).cuda(), 70, 13)
    assert (p3 - p2).abs().max().item() <= 1e-2 and (v3 - v2).abs().max().item() <= 2e-2
    assert (p3.cpu() - torch.softmax(logits, -1)).abs().max().item() <= 2e-2 and (v3.cpu() - vr.squeeze(1)).abs().max().item() <= 3e-2
    p3, v3 = p3.clone(), v3.clone()
    inf.use_fused_block = False  # two launches per block: the fused block kernel must reproduce them bit for bit
    p4, v4 = inf.forward_tiled(eu.tile_features(x).cuda(), 70, 13)
    assert torch.equal(p4, p3) and torch.equal(v4, v3)


def _resblock_checks(bnd, boards, C, S, device, bit_exact_vs_two_launches):
    """azsp_resblock_tiled == relu(conv(relu(conv(x, w1) + b1), w2) + b2 + x): against fp32 torch on the same bf16 operands (the
    intermediate rounded to bf16, as both kernel paths do) and, bit for bit, against two azsp_conv3x3_tiled launches."""
    g = torch.Generator().manual_seed(300 + boards)
    x = torch.randn(boards, C, S, S, generator=g).to(torch.bfloat16).to(device).contiguous(memory_format=torch.channels_last)
    ws = [(torch.randn(C, C, 3, 3, generator=g) * (0.05 if C >= 64 else 0.2)).to(torch.bfloat16).to(device) for _ in range(2)]
    bs = [torch.randn(C, generator=g).to(device) for _ in range(2)]
    wps = [w.permute(2, 3, 0, 1).reshape(9, C, C).contiguous() for w in ws]
    n = bnd.dll.azsp_tiled_bytes(boards, S, C) // 2
    xt, mt, y2, yf = (torch.zeros(n, dtype=torch.bfloat16, device=device) for _ in range(4))
    assert bnd.dll.azsp_tile_layout(x.data_ptr(), xt.data_ptr(), boards, S, C, 1, None) == 0
    x_before = xt.clone()
    assert bnd.dll.azsp_resblock_tiled(xt.data_ptr(), wps[0].data_ptr(), bs[0].data_ptr(), wps[1].data_ptr(), bs[1].data_ptr(), yf.data_ptr(),
                                       boards, S, C, None) == 0
    assert bnd.dll.azsp_conv3x3_tiled(xt.data_ptr(), wps[0].data_ptr(), bs[0].data_ptr(), None, mt.data_ptr(), boards, S, C, 1, None) == 0
    assert bnd.dll.azsp_conv3x3_tiled(mt.data_ptr(), wps[1].data_ptr(), bs[1].data_ptr(), xt.data_ptr(), y2.data_ptr(), boards, S, C, 1, None) == 0
    y = torch.empty_like(x)
    assert bnd.dll.azsp_tile_layout(yf.data_ptr(), y.data_ptr(), boards, S, C, 0, None) == 0
    if device != "cpu":
        torch.cuda.synchronize()
    assert torch.equal(xt, x_before)  # the input tile is read-only
    mid = torch.relu(torch.nn.functional.conv2d(x.float(), ws[0].float(), bs[0].float(), padding=1)).to(torch.bfloat16).float()
    ref = torch.relu(torch.nn.functional.conv2d(mid, ws[1].float(), bs[1].float(), padding=1) + x.float())
    err = (y.float() - ref).abs().max().item()
    assert err <= 1.0 / 64 * max(1.0, ref.abs().max().item()), err  # two bf16 roundings (a flipped rounding of the intermediate moves the output)
    if bit_exact_vs_two_launches:
        rows = boards * S * S  # valid positions; the rest of a ragged last tile is unspecified
        tb = max(1, 256 // (S * S))
        a = yf.view(-1, C // 8, tb * S * S, 8).permute(0, 2, 1, 3).reshape(-1, C)[:rows]
        b = y2.view(-1, C // 8, tb * S * S, 8).permute(0, 2, 1, 3).reshape(-1, C)[:rows]
        assert torch.equal(a, b)
    # in place (y aliases x): a tile is wholly in LDS before its first output is stored
    assert bnd.dll.azsp_resblock_tiled(xt.data_ptr(), wps[0].data_ptr(), bs[0].data_ptr(), wps[1].data_ptr(), bs[1].data_ptr(), xt.data_ptr(),
                                       boards, S, C, None) == 0
    if device != "cpu":
        torch.cuda.synchronize()
    rows = boards * S * S
    tb = max(1, 256 // (S * S))
    assert torch.equal(xt.view(-1, C // 8, tb * S * S, 8).permute(0, 2, 1, 3).reshape(-1, C)[:rows],
                       yf.view(-1, C // 8, tb * S * S, 8).permute(0, 2, 1, 3).reshape(-1, C)[:rows])
    assert bnd.dll.azsp_resblock_tiled(xt.data_ptr(), wps[0].data_ptr(), bs[0].data_ptr(), wps[1].data_ptr(), bs[1].data_ptr(), yf.data_ptr(),
                                       boards, 11, C, None) != 0  # unsupported geometry is refused, never silently computed


def test_resblock_abi_host_twin():
    """azsp_resblock_tiled through the ABI on the host twin (plain loops), tiny shapes."""
    import engine_util as eu

    b = eu.hosttwin_binding()
    for boards in (1, 4):
        g = torch.Generator().manual_seed(boards)
        C, S = 16, 5
        x = torch.randn(boards, C, S, S, generator=g).to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        ws = [(torch.randn(C, C, 3, 3, generator=g) * 0.2).to(torch.bfloat16) for _ in range(2)]
        bs = [torch.randn(C, generator=g) for _ in range(2)]
        wps = [w.permute(2, 3, 0, 1).reshape(9, C, C).contiguous() for w in ws]
        n = b.dll.azsp_tiled_bytes(boards, S, C) // 2
        xt, yt = torch.zeros(n, dtype=torch.bfloat16), torch.zeros(n, dtype=torch.bfloat16)
        assert b.dll.azsp_tile_layout(x.data_ptr(), xt.data_ptr(), boards, S, C, 1, None) == 0
        assert b.dll.azsp_resblock_tiled(xt.data_ptr(), wps[0].data_ptr(), bs[0].data_ptr(), wps[1].data_ptr(), bs[1].data_ptr(), yt.data_ptr(),
                                         boards, S, C, None) == 0
        y = torch.empty_like(x)
        assert b.dll.azsp_tile_layout(yt.data_ptr(), y.data_ptr(), boards, S, C, 0, None) == 0
        mid = torch.relu(torch.nn.functional.conv2d(x.float(), ws[0].float(), bs[0].float(), padding=1)).to(torch.bfloat16).float()
        ref = torch.relu(torch.nn.functional.conv2d(mid, ws[1].float(), bs[1].float(), padding=1) + x.float())
        assert (y.float() - ref).abs().max().item() <= 1.0 / 64 * max(1.0, ref.abs().max().item())
    assert b.dll.azsp_resblock_tiled(None, None, None, None, None, None, 1, 17, 64, None) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("boards", [1, 2, 5, 255, 256, 257, 513, 1000, 3000])
def test_gpu_resblock_gomoku_shape(boards):
    """k_resblock64<17> (one launch per ResNetBlock, 17x17 planes x 64 filters): vs torch, and bit-identical to the two-launch path.
    Board counts around one / two tiles per CU exercise the first-tile, has-next and last-tile paths of the persistent loop."""
    from alpha_zero_amd import _lib

    _resblock_checks(_lib.load(), boards, 64, 17, "cuda", True)


@pytest.mark.gpu
@pytest.mark.parametrize("boards", [1, 3, 4, 7, 767, 768, 770, 1539, 5000])
def test_gpu_resblock_go9_64_shape(boards):
    """k_resblock64<9> / k_conv3x3_t64<9> (9x9 planes x 64 filters, three boards per tile): the reference's 9x9_12b64 network shape."""
    from alpha_zero_amd import _lib

    bnd = _lib.load()
    _tiled_roundtrip_and_conv(bnd, boards, 64, 9, "cuda")
    _resblock_checks(bnd, boards, 64, 9, "cuda", True)


@pytest.mark.gpu
def test_gpu_go9_64_network_tiled_forward():
    """9x9 Go with 64 filters (logs/go/9x9_12b64/run.log:1: 11 blocks x 64): the whole evaluator on the engine's tiled features (stem,
    fused residual blocks, heads, FC) vs the same InferenceNet on the library path and vs the fp32 module."""
    import engine_util as eu
    from alpha_zero_amd import _lib

    torch.manual_seed(12)
    net = AlphaZeroNet((17, 9, 9), 82, 4, 64, 64)
    with torch.no_grad():
        net.policy_head[4].weight.mul_(0.2)
        net.value_head[6].weight.mul_(0.3)
    inf = InferenceNet(net, dtype=torch.bfloat16, binding=_lib.load()).cuda()
    assert inf.supports_tiled_features(9, "cuda") and "hand-written" in inf.evaluator_path(9, "cuda")
    x = (torch.rand(100, 17, 9, 9) > 0.6).float()
    p1, v1 = inf.forward_tiled(eu.tile_features(x).cuda(), 100, 9)
    p2, v2 = inf(x.cuda())  # NCHW entry: library stem, tiled fused tower
    inf.use_tiled_tower = False
    p3, v3 = inf(x.cuda())  # library convolutions throughout
    inf.use_tiled_tower = True
    # bf16 path vs bf16 path (the library FC layers also round their outputs to bf16): both sit within the fp32 bound below
    assert (p1 - p2).abs().max().item() <= 1e-2 and (v1 - v2).abs().max().item() <= 4e-2
    assert (p1 - p3).abs().max().item() <= 1e-2 and (v1 - v3).abs().max().item() <= 4e-2
    logits, vr = net.eval()(x)
    assert (p1.cpu() - torch.softmax(logits, -1)).abs().max().item() <= 2e-2 and (v1.cpu() - vr.squeeze(1)).abs().max().item() <= 3e-2
    # fused blocks == two launches per block, bit for bit, through the whole forward
    inf.use_fused_block = False
    p4, v4 = inf.forward_tiled(eu.tile_features(x).cuda(), 100, 9)
    assert torch.equal(p4, p1) and torch.equal(v4, v1)


def test_tiled_bytes_rule_host_twin():
    """azsp_tiled_bytes: T = max(1, 256 // S^2) boards per tile, whole tiles allocated."""
    import engine_util as eu

    b = eu.hosttwin_binding()
    for S, C, boards in ((5, 16, 7), (9, 128, 32768), (9, 32, 10), (13, 32, 5), (17, 64, 3), (19, 256, 2)):
        tb = max(1, 256 // (S * S))
        assert b.dll.azsp_tiled_bytes(boards, S, C) == (boards + tb - 1) // tb * tb * S * S * C * 2
    assert b.dll.azsp_tiled_bytes(-1, 9, 128) == -1 and b.dll.azsp_tiled_bytes(3, 9, 12) == -1


@pytest.mark.gpu
def test_gpu_gomoku_forward_tiled_full_batch_is_batch_independent():
    """BASELINE C2 size (32 768 rows of 13x13 boards, 6 x 64 network): evaluations do not depend on the batch composition."""
    import engine_util as eu
    from alpha_zero_amd import _lib

    torch.manual_seed(10)
    net = AlphaZeroNet((17, 13, 13), 169, 6, 64, 64, gomoku=True)
    inf = InferenceNet(net, dtype=torch.bfloat16, binding=_lib.load()).cuda()
    rows = 32768
    x = (torch.rand(rows, 17, 13, 13) > 0.7).float()
    x[30000:30064] = x[:64]
    p_all, v_all = (t.clone() for t in inf.forward_tiled(eu.tile_features(x).cuda(), rows, 13))
    assert torch.isfinite(p_all).all() and torch.isfinite(v_all).all()
    assert torch.equal(p_all[30000:30064], p_all[:64]) and torch.equal(v_all[30000:30064], v_all[:64])
    p_sub, v_sub = inf.forward_tiled(eu.tile_features(x[:257]).cuda(), 257, 13)
    assert torch.equal(p_sub, p_all[:257]) and torch.equal(v_sub, v_all[:257])


def _fc_heads_check(bnd, boards, P2, A, Fw, device):
    """azsp_fc_heads (MFMA GEMMs + softmax / tanh) vs torch on the same bf16 operands."""
    import ctypes

    g = torch.Generator().manual_seed(boards + A)
    k1, k2 = (2 * P2 + 15) // 16 * 16, (P2 + 15) // 16 * 16
    pol = torch.zeros(boards + 1, k1)
    val = torch.zeros(boards + 1, k2)
    pol[:boards, : 2 * P2] = torch.relu(torch.randn(boards, 2 * P2, generator=g))
    val[:boards, :P2] = torch.relu(torch.randn(boards, P2, generator=g))
    pol, val = pol.to(torch.bfloat16), val.to(torch.bfloat16)
    wp, bp = (torch.randn(A, 2 * P2, generator=g) * 0.1).to(torch.bfloat16), torch.randn(A, generator=g) * 0.1
    w1, b1 = (torch.randn(Fw, P2, generator=g) * 0.1).to(torch.bfloat16), torch.randn(Fw, generator=g) * 0.1
    w2, b2 = torch.randn(Fw, generator=g) * 0.2, 0.05

    def padw(w):
        o = torch.zeros((w.shape[0] + 31) // 32 * 32, (w.shape[1] + 15) // 16 * 16, dtype=torch.bfloat16)
        o[: w.shape[0], : w.shape[1]] = w
        return o.to(device)

    def padv(v):
        o = torch.zeros((v.numel() + 31) // 32 * 32)
        o[: v.numel()] = v
        return o.to(device)

    pri = torch.empty(boards, A, device=device)
    v = torch.empty(boards, device=device)
    keep = (pol.to(device), val.to(device), padw(wp), padv(bp), padw(w1), padv(b1), padv(w2))
    rc = bnd.dll.azsp_fc_heads(keep[0].data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(), k1 // 16, keep[4].data_ptr(),
                               keep[5].data_ptr(), k2 // 16, keep[6].data_ptr(), ctypes.c_float(b2), pri.data_ptr(), v.data_ptr(), boards, A, Fw, None)
    assert rc == 0
    if device != "cpu":
        torch.cuda.synchronize()
    logits = pol[:boards, : 2 * P2].float() @ wp.float().T + bp
    want_p = torch.softmax(logits, -1)
    want_v = torch.tanh(torch.relu(val[:boards, :P2].float() @ w1.float().T + b1) @ w2 + b2)
    assert (pri.cpu() - want_p).abs().max().item() <= 2e-5 and (v.cpu() - want_v).abs().max().item() <= 2e-5
    assert torch.allclose(pri.sum(1).cpu(), torch.ones(boards), atol=1e-5)


def test_fc_heads_abi_host_twin():
    import engine_util as eu

    _fc_heads_check(eu.hosttwin_binding(), 5, 25, 26, 16, "cpu")


@pytest.mark.gpu
@pytest.mark.parametrize("boards,P2,A,Fw", [(1, 81, 82, 128), (33, 81, 82, 64), (1000, 81, 82, 128), (70, 289, 169, 64)])
def test_gpu_fc_heads_match_torch(boards, P2, A, Fw):
    from alpha_zero_amd import _lib

    _fc_heads_check(_lib.load(), boards, P2, A, Fw, "cuda")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype_name", ["bfloat16", "float32"])
def test_gpu_two_forwards_in_flight_on_two_streams_equal_the_serial_results(dtype_name):
    """Two evaluator forwards of one InferenceNet in flight at the same time on two HIP streams (different scratch slots) give, bit for
    bit, the results of the same forwards run one after the other -- 30 rounds, batches large enough that the two forwards really
    overlap on the device (~4.6 k rows each: every tower launch fills all CUs).  Round 3 saw the head planes of the forward launched
    first differ in 36 of 40 such rounds; round 4 traced it to the head kernel's LDS weight table (tools/probes/make_head_probe.py,
    profiles/r04_concurrency_probe2_*.txt) and removed the table (az_conv.h k_head_tiled).  Round 5 removed the same table from the
    fp32-class head kernel (az_conv_sp.h k_head_split: 1x1 weights through scalar loads from global memory; LDS holds only the planes the
    workgroup itself produced) -- no product kernel keeps the pattern; the fp32-class path runs 200 rounds of this check."""
    import json
    import os

    import engine_util as eu
    from alpha_zero_amd import _lib

    dt = getattr(torch, dtype_name)
    torch.manual_seed(2)
    net = AlphaZeroNet((17, 9, 9), 82, 3, 128, 128).eval()
    inf = InferenceNet(net, dtype=dt, binding=_lib.load()).cuda()
    rows = (4608, 4863)
    g = torch.Generator().manual_seed(7)
    planes = [(torch.rand(r, 17, 9, 9, generator=g) > 0.6).float() for r in rows]
    if dt == torch.float32:
        assert inf.supports_split_features(9, "cuda")
        feats = [p.cuda().contiguous() for p in planes]
        fwd = lambda k: inf.forward_split(feats[k], slot=1 + k)  # noqa: E731
    else:
        assert inf.supports_tiled_features(9, "cuda")
        feats = [eu.tile_features(p).cuda() for p in planes]

        def fwd(k):
            pri, v = inf.forward_tiled(feats[k], rows[k], 9, slot=1 + k)
            return pri.clone(), v.clone()

    serial = []
    for k in range(2):
        serial.append(fwd(k))
        torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    main = torch.cuda.current_stream()
    bad, n_rounds = [], (200 if dt == torch.float32 else 30)
    for r in range(n_rounds):
        out = [None, None]
        for k in range(2):
            streams[k].wait_stream(main)
            with torch.cuda.stream(streams[k]):
                out[k] = fwd(k)
        for s in streams:
            main.wait_stream(s)
        torch.cuda.synchronize()
        for k in range(2):
            if not (torch.equal(out[k][0], serial[k][0]) and torch.equal(out[k][1], serial[k][1])):
                bad.append((r, k, float((out[k][0] - serial[k][0]).abs().max()), float((out[k][1] - serial[k][1]).abs().max())))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    json.dump(dict(dtype=dtype_name, rounds=n_rounds, rows=rows, differing=bad), open(os.path.join(root, "gpurun_out", f"two_stream_forwards_{dtype_name}.json"), "w"))
    assert not bad, bad[:5]


def test_widen_for_kernels_only_when_it_pays():
    """widen_for_kernels: the reference's shipped 10 x 40 Gomoku width goes to 64 (2.56x the FLOPs, on kernels 4.5 - 5x the library's
    speed); widths that have kernels stay; a widening beyond 4x the tower's FLOPs (64 -> 256 at 19x19 = 16x) is left to the library."""
    from alpha_zero_amd.core.network import widen_for_kernels

    g40 = AlphaZeroNet((17, 13, 13), 169, 2, 40, 64, gomoku=True)
    w, note = widen_for_kernels(g40, 13, torch.float32)
    assert w.conv_block[0].out_channels == 64 and "40 -> 64" in note
    x = (torch.rand(3, 17, 13, 13) > 0.6).float()
    with torch.no_grad():
        (l0, v0), (l1, v1) = g40.eval()(x), w.eval()(x)
    assert torch.allclose(l0, l1, atol=1e-6) and torch.allclose(v0, v1, atol=1e-6)  # the same function
    g64 = AlphaZeroNet((17, 13, 13), 169, 1, 64, 64, gomoku=True)
    assert widen_for_kernels(g64, 13, torch.float32)[0] is g64
    go64_19 = AlphaZeroNet((17, 19, 19), 362, 1, 64, 64)
    assert widen_for_kernels(go64_19, 19, torch.bfloat16) == (go64_19, "")          # 16x the FLOPs: no
    go128_19 = AlphaZeroNet((17, 19, 19), 362, 1, 128, 64)
    w2, note2 = widen_for_kernels(go128_19, 19, torch.bfloat16)
    assert w2.conv_block[0].out_channels == 256 and "128 -> 256" in note2           # 4x: still worth it
    assert widen_for_kernels(go128_19, 19, torch.float32) == (go128_19, "")          # no fp32-class kernels at 19x19
