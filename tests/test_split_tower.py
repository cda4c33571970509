"""The residual tower at the reference's precision class (core/pipeline.py:91-123 evaluates the network in fp32) on the f16-rate
matrix cores: azsp_split_layout / azsp_conv3x3_split (include/azsp.h, alpha_zero_amd/csrc/az_conv_sp.h).

Checker = torch in fp64 on the CPU (the same convolution, exact to ~1e-16), with the library's own fp32 convolution measured beside
the kernel: the statement is "the kernel's error against fp64 is of the size of fp32 round-off", tolerance written in each test."""
import json
import os

import pytest
import torch
import torch.nn.functional as F

from alpha_zero_amd.core.network import AlphaZeroNet, InferenceNet, split_weights_f16

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _inputs(boards, C, S, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(boards, C, S, S, generator=g) * scale
    x = torch.where(torch.rand(boards, C, S, S, generator=g) < 0.5, torch.zeros(()), x.abs())  # post-ReLU-like: half zeros
    x[:, : C // 8] *= 37.0    # a few loud channels ...
    x[:, -C // 8 :] *= 3e-3   # ... and a few quiet ones
    r = torch.randn(boards, C, S, S, generator=g).abs() * scale
    w = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(C, generator=g) * 0.1
    return x, r, w, b


def _run_split_conv(bnd, x, r, w, b, relu, device):
    """x, r: [B,C,S,S] fp32 -> y [B,C,S,S] fp32 through split_layout -> conv3x3_split -> split_layout."""
    B, C, S, _ = x.shape
    dll = bnd.dll
    xc = x.to(device).contiguous(memory_format=torch.channels_last)
    n = dll.azsp_split_bytes(B, S, C) // 2
    assert n == B * 2 * S * S * C
    xs, ys = torch.zeros(n, dtype=torch.float16, device=device), torch.zeros(n, dtype=torch.float16, device=device)
    assert dll.azsp_split_layout(xc.data_ptr(), xs.data_ptr(), B, S, C, 1, None, None) == 0
    rs = None
    if r is not None:
        rc = r.to(device).contiguous(memory_format=torch.channels_last)
        rs = torch.zeros(n, dtype=torch.float16, device=device)
        assert dll.azsp_split_layout(rc.data_ptr(), rs.data_ptr(), B, S, C, 1, None, None) == 0
    wsp, bb = split_weights_f16(w).to(device), b.float().to(device)
    assert dll.azsp_conv3x3_split(xs.data_ptr(), wsp.data_ptr(), bb.data_ptr(), rs.data_ptr() if rs is not None else None, ys.data_ptr(),
                                  B, S, C, relu, None, None) == 0
    y = torch.empty_like(xc)
    assert dll.azsp_split_layout(ys.data_ptr(), y.data_ptr(), B, S, C, 0, None, None) == 0
    # the layout round trip of the input itself: 22-bit significands
    back = torch.empty_like(xc)
    assert dll.azsp_split_layout(xs.data_ptr(), back.data_ptr(), B, S, C, 0, None, None) == 0
    if device != "cpu":
        torch.cuda.synchronize()
    # |v - hi - lo / 2048| <= 2^-22 |v| (+ the fp32 rounding of the join); below f16's normal range the lo half is a multiple of
    # 2^-24 / 2048 = 2^-35: the excess over that absolute floor, relative to |v|
    rt = (((back.cpu() - x).abs() - 2.0 ** -35).clamp_min(0) / x.abs().clamp_min(1e-30)).max().item()
    return y.cpu().contiguous(), rt


def _ref64(x, r, w, b, relu):
    y = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if r is not None:
        y = y + r.double()
    return torch.relu(y) if relu else y


def test_split_abi_host_twin():
    """azsp_split_layout / azsp_conv3x3_split through the ABI on the host twin (plain loops on the same hi / lo f16 arithmetic)."""
    import engine_util as eu

    bnd = eu.hosttwin_binding()
    for boards, res, relu in ((1, False, 1), (2, True, 1), (1, True, 0)):
        x, r, w, b = _inputs(boards, 64, 9, 10 + boards)
        if not relu:
            x = x - 0.3
        y, rt = _run_split_conv(bnd, x, r if res else None, w, b, relu, "cpu")
        ref = _ref64(x, r if res else None, w, b, relu)
        err = (y.double() - ref).abs().max().item() / ref.abs().max().item()
        assert rt <= 2.0 ** -21, rt
        assert err <= 2e-6, (boards, res, relu, err)
    assert bnd.dll.azsp_split_bytes(3, 9, 128) == 3 * 2 * 81 * 128 * 2 and bnd.dll.azsp_split_bytes(1, 9, 12) == -1
    assert bnd.dll.azsp_conv3x3_split(None, None, None, None, None, 1, 9, 128, 1, None, None) != 0
    z = torch.zeros(2 * 2 * 121 * 64, dtype=torch.float16)
    zb = torch.zeros(64)
    assert bnd.dll.azsp_conv3x3_split(z.data_ptr(), z.data_ptr(), zb.data_ptr(), None, z.clone().data_ptr(), 1, 11, 64, 1, None, None) == 0  # (round 6: every plane size)
    assert bnd.dll.azsp_conv3x3_split(z.data_ptr(), z.data_ptr(), zb.data_ptr(), None, z.clone().data_ptr(), 1, 11, 32, 1, None, None) != 0  # unsupported filter count
    assert bnd.dll.azsp_conv3x3_split(z.data_ptr(), z.data_ptr(), zb.data_ptr(), None, z.clone().data_ptr(), 1, 2, 64, 1, None, None) != 0   # unsupported plane size


def test_split_abi_host_twin_17x17_and_pad3_stem():
    """The 13x13 Gomoku shapes through the ABI on the host twin: azsp_conv3x3_split at (S, C) = (17, 64), and the pad-3 stem (13x13
    feature board -> 17x17 planes, network.py:101-105) + tower + heads through InferenceNet.forward_split vs the fp64 module."""
    import engine_util as eu

    bnd = eu.hosttwin_binding()
    x, r, w, b = _inputs(1, 64, 17, 5)
    y, rt = _run_split_conv(bnd, x, r, w, b, 1, "cpu")
    ref = _ref64(x, r, w, b, 1)
    assert (y.double() - ref).abs().max().item() / ref.abs().max().item() <= 2e-6 and rt <= 2.0 ** -21
    net = _trained_like_gomoku_net(64, 1)
    inf = InferenceNet(net, dtype=torch.float32, binding=bnd)
    xs = (torch.rand(2, 17, 13, 13, generator=torch.Generator().manual_seed(2)) > 0.6).float()
    pri, v = inf.forward_split(xs)
    with torch.no_grad():
        lg, v64 = net.double()(xs.double())
    dp, dv = (pri.double() - torch.softmax(lg, -1)).abs().max().item(), (v.double() - v64.squeeze(1)).abs().max().item()
    assert dp <= 2e-6 and dv <= 2e-6, (dp, dv)
    # shapes without a kernel are refused, not guessed
    z = torch.zeros(2 * 2 * 169 * 64, dtype=torch.float16)
    assert bnd.dll.azsp_stem_split(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.clone().data_ptr(), 1, 13, 64, 1, 1, None, None) != 0  # 13x13 with pad 1
    assert bnd.dll.azsp_stem_split(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.clone().data_ptr(), 1, 9, 64, 3, 1, None, None) != 0


def test_split_range_record_host_twin_and_weight_check():
    """Out-of-range values are clamped AND recorded (azsp_split_range_status: events, largest |v|, reset); folded weights beyond f16's
    range are refused when they are packed (ValueError) instead of turning into inf / a silent clamp."""
    import ctypes

    import engine_util as eu

    bnd = eu.hosttwin_binding()
    ev, mx = ctypes.c_uint32(7), ctypes.c_float(7.0)
    assert bnd.dll.azsp_split_range_status(ctypes.byref(ev), ctypes.byref(mx), 1, None) == 0  # reset whatever earlier tests left
    assert bnd.dll.azsp_split_range_status(ctypes.byref(ev), ctypes.byref(mx), 0, None) == 0 and ev.value == 0 and mx.value == 0.0
    x = torch.tensor([1.0, -3.0, 65504.0, 1e-7, 0.0, 3.0, -2.5e-5, 60000.0]).reshape(1, 8, 1, 1).contiguous(memory_format=torch.channels_last)
    s = torch.zeros(2 * 8, dtype=torch.float16)
    assert bnd.dll.azsp_split_layout(x.data_ptr(), s.data_ptr(), 1, 1, 8, 1, None, None) == 0
    assert bnd.dll.azsp_split_range_status(ctypes.byref(ev), ctypes.byref(mx), 0, None) == 0 and ev.value == 0  # +-65504 itself is in range
    x[0, 1, 0, 0], x[0, 5, 0, 0] = -1e5, 70000.0
    assert bnd.dll.azsp_split_layout(x.data_ptr(), s.data_ptr(), 1, 1, 8, 1, None, None) == 0
    assert bnd.dll.azsp_split_range_status(ctypes.byref(ev), None, 0, None) == 0 and ev.value == 2
    assert bnd.dll.azsp_split_range_status(None, ctypes.byref(mx), 1, None) == 0 and mx.value == 1e5
    assert bnd.dll.azsp_split_range_status(ctypes.byref(ev), ctypes.byref(mx), 0, None) == 0 and ev.value == 0 and mx.value == 0.0
    w = torch.randn(8, 8, 3, 3)
    assert split_weights_f16(w).shape == (2, 9, 8, 8)
    for bad in (7e4, float("inf"), float("nan")):
        w2 = w.clone()
        w2[3, 2, 1, 1] = bad
        with pytest.raises(ValueError):
            split_weights_f16(w2)


def test_split_tower_clamps_at_f16_range_host_twin():
    """Values beyond f16's largest finite number are clamped when they are split (documented in include/azsp.h), never inf / NaN."""
    import engine_util as eu

    bnd = eu.hosttwin_binding()
    x = torch.tensor([1e5, -1e5, 65504.0, 1e-7, 0.0, 3.0, -2.5e-5, 70000.0]).reshape(1, 8, 1, 1).contiguous(memory_format=torch.channels_last)
    s = torch.zeros(2 * 8, dtype=torch.float16)
    assert bnd.dll.azsp_split_layout(x.data_ptr(), s.data_ptr(), 1, 1, 8, 1, None, None) == 0
    back = torch.empty_like(x)
    assert bnd.dll.azsp_split_layout(s.data_ptr(), back.data_ptr(), 1, 1, 8, 0, None, None) == 0
    want = x.clamp(-65504.0, 65504.0)
    assert torch.isfinite(back).all() and ((back - want).abs() <= want.abs() * 2.0 ** -21 + 1e-11).all(), back.flatten()


@pytest.mark.gpu
@pytest.mark.parametrize("boards,C", [(b, c) for c in (128, 64) for b in (1, 2, 127, 128, 129, 300, 1000)] + [(2181, 128), (4357, 64)])
def test_gpu_split_conv_error_vs_fp64(boards, C):
    """k_conv3x3_sp vs fp64, next to the library's fp32 convolution on the same inputs.  Board counts around one / two boards per
    workgroup slot (128 slots at 128 filters, 256 at 64) exercise the first-board, has-next and last-board paths of the persistent loop;
    the two large counts give every workgroup 17 boards + some an 18th: a full 16-board corner group followed by a partial one.
    Bound: the kernel picks no algorithm at run time, so its error is the same on every box -- measured max 6.49e-7 of max|y64| on the boxes
    of rounds 3 and 4 (profiles/r03_split_conv_error.jsonl, profiles/r04_bounds_two_boxes.txt) -- and is asserted ABSOLUTELY: <= 8e-7.
    The library's own fp32 error on the same inputs is recorded beside it (2.7e-7 .. 8.8e-7 depending on the case and on the algorithm
    MIOpen picks on that box); round 3's bound "<= 2 x library + 5e-7" made the test depend on that choice and is gone."""
    from alpha_zero_amd import _lib

    bnd = _lib.load()
    out = []
    for res, relu in ((False, 1), (True, 1), (True, 0)):
        x, r, w, b = _inputs(boards, C, 9, 100 + boards)
        if not relu:
            x = x - 0.3
        y, rt = _run_split_conv(bnd, x, r if res else None, w, b, relu, "cuda")
        ref = _ref64(x, r if res else None, w, b, relu)
        lib = F.conv2d(x.cuda(), w.cuda(), b.cuda(), padding=1)
        if res:
            lib = lib + r.cuda()
        lib = (torch.relu(lib) if relu else lib).cpu()
        scale = ref.abs().max().item()
        err, lib_err = (y.double() - ref).abs().max().item() / scale, (lib.double() - ref).abs().max().item() / scale
        out.append(dict(boards=boards, C=C, residual=res, relu=relu, err=err, library_fp32_err=lib_err, layout_roundtrip_rel=rt))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "split_conv_error.jsonl"), "a") as f:
        for o in out:
            f.write(json.dumps(o) + "\n")
    for o in out:
        assert o["layout_roundtrip_rel"] <= 2.0 ** -21, o
        assert o["err"] <= 8e-7, o


@pytest.mark.gpu
def test_gpu_split_conv_small_and_large_magnitudes():
    """The lo halves are scaled by 2^11, so precision must hold for activations far from 1: tiny (below f16's normal range the hi half
    turns subnormal and the scaled lo half carries the value) and large (up to the clamp)."""
    from alpha_zero_amd import _lib

    bnd = _lib.load()
    res = {}
    for name, scale in (("1e-4", 1e-4), ("1e-2", 1e-2), ("1e2", 1e2)):
        x, r, w, b = _inputs(40, 128, 9, 7, scale=scale)
        b = b * scale
        y, _ = _run_split_conv(bnd, x, r, w, b, 1, "cuda")
        ref = _ref64(x, r, w, b, 1)
        res[name] = (y.double() - ref).abs().max().item() / ref.abs().max().item()
    print(json.dumps(res))
    # measured: 6.7e-7 / 5.7e-7 / 5.1e-7 -- the f16 MFMA takes subnormal hi halves as they are, the scaled lo halves carry the rest
    assert max(res.values()) <= 2e-6, res


def test_column_tile_map_in_the_header_is_the_generated_one():
    """az_conv_sp.h carries the lane -> position table of tools/gen_sp_map.py (conflict-free ds_read_b128 lane groups; the header
    static_asserts the property, this test keeps the table and its generator in step)."""
    import re
    import subprocess
    import sys

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_sp_map.py")], capture_output=True, text=True, check=True).stdout
    gen = [int(v) for v in re.findall(r"\d+", out.split("\n", 1)[1])]
    hdr = open(os.path.join(ROOT, "alpha_zero_amd", "csrc", "az_conv_sp.h")).read()
    body = hdr[hdr.index("constexpr SpMap sp_map_table = {{") :]
    tab = [int(v) for v in re.findall(r"\d+", body[: body.index("}};")].split("{{", 1)[1])]
    assert len(gen) == 80 and gen == tab
    assert sorted(gen) == [p for p in range(81) if p != 72]  # every position but the corner (8, 0), once


def _trained_like_gomoku_net(filters, blocks, seed=3, fc=64):
    torch.manual_seed(seed)
    net = AlphaZeroNet((17, 13, 13), 169, blocks, filters, fc, gomoku=True).eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2), m.running_var.uniform_(0.5, 1.5), m.weight.uniform_(0.7, 1.3), m.bias.normal_(0, 0.2)
    return net


def _trained_like_net(filters, blocks, seed=3):
    torch.manual_seed(seed)
    net = AlphaZeroNet((17, 9, 9), 82, blocks, filters, 128).eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):  # non-trivial running statistics, as after training
                m.running_mean.normal_(0, 0.2), m.running_var.uniform_(0.5, 1.5), m.weight.uniform_(0.7, 1.3), m.bias.normal_(0, 0.2)
    return net


def test_split_evaluator_host_twin():
    """azsp_split_features -> azsp_stem_split -> azsp_conv3x3_split x 2 -> azsp_head_split through InferenceNet.forward_split on the host
    twin (plain loops on the same arithmetic): a 1-block 64-filter network on 3 positions vs the fp64 module."""
    import engine_util as eu

    net = _trained_like_net(64, 1)
    inf = InferenceNet(net, dtype=torch.float32, binding=eu.hosttwin_binding())
    x = (torch.rand(3, 17, 9, 9, generator=torch.Generator().manual_seed(2)) > 0.6).float()
    pri, v = inf.forward_split(x)
    with torch.no_grad():
        lg, v64 = net.double()(x.double())
    dp, dv = (pri.double() - torch.softmax(lg, -1)).abs().max().item(), (v.double() - v64.squeeze(1)).abs().max().item()
    assert dp <= 2e-6 and dv <= 2e-6, (dp, dv)
    assert abs(pri.sum(1) - 1).max().item() <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("filters", [128, 64])
def test_gpu_fp32_network_on_the_split_kernels(filters):
    """The whole fp32-class evaluator (azsp_split_features / azsp_stem_split / azsp_conv3x3_split / azsp_head_split), the same tower
    behind a library stem and heads, and the all-library fp32 InferenceNet, each against the fp64 module (10 blocks, non-trivial
    BatchNorm statistics: round-off is amplified through the depth for every fp32 path, and the library's own distance to fp64 moves
    with the convolution algorithm it picks: 1.9e-5 .. 9.3e-5 on priors between two boxes).  Bound: within 2e-4 of fp64 absolutely and
    at most 4x + 2e-5 the library path's distance (measured on MI355X, r03, priors / values at 128 filters: split evaluator
    5.6e-5 / 2.4e-5, library 1.9e-5 .. 9.3e-5 / 2.9e-5 .. 4.0e-5)."""
    from alpha_zero_amd import _lib

    net = _trained_like_net(filters, 10)
    inf = InferenceNet(net, dtype=torch.float32, binding=_lib.load()).cuda()
    assert inf.supports_split_features(9, "cuda") and "hand-written" in inf.evaluator_path(9, "cuda")
    x = (torch.rand(203, 17, 9, 9, generator=torch.Generator().manual_seed(1)) > 0.6).float()
    with torch.no_grad():
        lg, v64 = net.double()(x.double())
    p64, v64 = torch.softmax(lg, -1), v64.squeeze(1)

    def dist(pv):
        return (pv[0].cpu().double() - p64).abs().max().item(), (pv[1].cpu().double() - v64).abs().max().item()

    d = dict(filters=filters)
    inf._split = None
    d["split_evaluator_vs_fp64"] = dist(inf(x.cuda()))
    assert inf._split is not None, "the split kernels did not run"
    inf.use_split_heads, inf._split = False, None
    d["split_tower_library_heads_vs_fp64"] = dist(inf(x.cuda()))
    assert inf._split is not None and "azsp_conv3x3_split" in inf.evaluator_path(9, "cuda")
    inf.use_split_tower = False
    d["library_vs_fp64"] = dist(inf(x.cuda()))
    print(json.dumps(d))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(d, open(os.path.join(ROOT, "gpurun_out", f"split_network_error_{filters}.json"), "w"))
    for name in ("split_evaluator_vs_fp64", "split_tower_library_heads_vs_fp64"):
        for k in (0, 1):
            assert d[name][k] <= 2e-4 and d[name][k] <= 4 * d["library_vs_fp64"][k] + 2e-5, d


@pytest.mark.gpu
@pytest.mark.parametrize("boards", [1, 3, 4, 5, 203])
def test_gpu_split_stem_and_heads_vs_fp64(boards):
    """azsp_split_features + azsp_stem_split and azsp_head_split on their own (a 0-block network = stem -> heads) vs the fp64 module,
    board counts around the head kernel's 4 boards per workgroup."""
    from alpha_zero_amd import _lib

    for filters in (128, 64):
        net = _trained_like_net(filters, 0, seed=boards)
        inf = InferenceNet(net, dtype=torch.float32, binding=_lib.load()).cuda()
        x = (torch.rand(boards, 17, 9, 9, generator=torch.Generator().manual_seed(boards)) > 0.6).float()
        pri, v = inf(x.cuda())
        assert inf._split is not None
        with torch.no_grad():
            lg, v64 = net.double()(x.double())
        dp, dv = (pri.cpu().double() - torch.softmax(lg, -1)).abs().max().item(), (v.cpu().double() - v64.squeeze(1)).abs().max().item()
        assert dp <= 2e-6 and dv <= 4e-6, (filters, boards, dp, dv)


@pytest.mark.gpu
@pytest.mark.parametrize("boards", [1, 2, 255, 256, 257, 300, 600, 1100])
def test_gpu_split_conv17_error_vs_fp64(boards):
    """k_conv3x3_sp17 (17x17 planes x 64 filters: the 13x13 Gomoku tower, half-board tiles) vs fp64, next to the library's fp32
    convolution.  Board counts around one / two / four boards per workgroup (256 persistent workgroups) exercise the first-board,
    has-next and last-board paths of the two-tiles-per-board loop.  Bound: absolute, like the 9x9 kernel's (the kernel's error does
    not depend on the box): measured max 4.45e-7 of max|y64| (profiles/r04_split_conv17_error.jsonl), asserted <= 6e-7; the library's own
    fp32 error (up to 7.1e-7 here) is recorded beside it."""
    from alpha_zero_amd import _lib

    bnd = _lib.load()
    out = []
    for res, relu in ((False, 1), (True, 1), (True, 0)):
        x, r, w, b = _inputs(boards, 64, 17, 300 + boards)
        if not relu:
            x = x - 0.3
        y, rt = _run_split_conv(bnd, x, r if res else None, w, b, relu, "cuda")
        ref = _ref64(x, r if res else None, w, b, relu)
        lib = F.conv2d(x.cuda(), w.cuda(), b.cuda(), padding=1)
        if res:
            lib = lib + r.cuda()
        lib = (torch.relu(lib) if relu else lib).cpu()
        scale = ref.abs().max().item()
        d = (y.double() - ref).abs()
        err, lib_err = d.max().item() / scale, (lib.double() - ref).abs().max().item() / scale
        worst = [int(v) for v in torch.nonzero(d == d.max())[0]]  # (board, channel, row, column) of the worst element: a layout bug shows here
        out.append(dict(boards=boards, S=17, C=64, residual=res, relu=relu, err=err, library_fp32_err=lib_err, layout_roundtrip_rel=rt, worst=worst))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "split_conv17_error.jsonl"), "a") as f:
        for o in out:
            f.write(json.dumps(o) + "\n")
    for o in out:
        assert o["layout_roundtrip_rel"] <= 2.0 ** -21, o
        assert o["err"] <= 6e-7, o


@pytest.mark.gpu
def test_gpu_fp32_gomoku_network_on_the_split_kernels():
    """The whole fp32-class evaluator of the 13x13 Gomoku network (BASELINE C2: 6 x 64; azsp_split_features -> azsp_stem_split with the
    pad-3 stem -> azsp_conv3x3_split at 17x17 -> azsp_head_split) and the all-library fp32 InferenceNet, each against the fp64 module.
    Same bound as the 9x9 networks: within 2e-4 of fp64 and at most 4x + 2e-5 the library path's distance."""
    from alpha_zero_amd import _lib

    net = _trained_like_gomoku_net(64, 6)
    inf = InferenceNet(net, dtype=torch.float32, binding=_lib.load()).cuda()
    assert inf.supports_split_features(13, "cuda") and "hand-written" in inf.evaluator_path(13, "cuda")
    x = (torch.rand(131, 17, 13, 13, generator=torch.Generator().manual_seed(1)) > 0.6).float()
    with torch.no_grad():
        lg, v64 = net.double()(x.double())
    p64, v64 = torch.softmax(lg, -1), v64.squeeze(1)

    def dist(pv):
        return (pv[0].cpu().double() - p64).abs().max().item(), (pv[1].cpu().double() - v64).abs().max().item()

    d = dict(filters=64, board=13)
    inf._split = None
    d["split_evaluator_vs_fp64"] = dist(inf(x.cuda()))
    assert inf._split is not None, "the split kernels did not run"
    inf.use_split_heads, inf._split = False, None
    d["split_tower_library_heads_vs_fp64"] = dist(inf(x.cuda()))
    assert inf._split is not None and "azsp_conv3x3_split" in inf.evaluator_path(13, "cuda")
    inf.use_split_tower = False
    d["library_vs_fp64"] = dist(inf(x.cuda()))
    assert inf.split_range_status(reset=True)[0] == 0
    print(json.dumps(d))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(d, open(os.path.join(ROOT, "gpurun_out", "split_network_error_gomoku13_64.json"), "w"))
    for name in ("split_evaluator_vs_fp64", "split_tower_library_heads_vs_fp64"):
        for k in (0, 1):
            assert d[name][k] <= 2e-4 and d[name][k] <= 4 * d["library_vs_fp64"][k] + 2e-5, d


@pytest.mark.gpu
@pytest.mark.parametrize("boards", [1, 3, 4, 5, 259])
def test_gpu_split_gomoku_stem_and_heads_vs_fp64(boards):
    """azsp_split_features + azsp_stem_split (13x13 board, pad 3 -> 17x17 planes) and azsp_head_split (289 positions, 169 actions) on
    their own (a 0-block network = stem -> heads) vs the fp64 module."""
    from alpha_zero_amd import _lib

    net = _trained_like_gomoku_net(64, 0, seed=boards, fc=80)
    inf = InferenceNet(net, dtype=torch.float32, binding=_lib.load()).cuda()
    x = (torch.rand(boards, 17, 13, 13, generator=torch.Generator().manual_seed(boards)) > 0.6).float()
    inf._split = None
    pri, v = inf(x.cuda())
    assert inf._split is not None
    with torch.no_grad():
        lg, v64 = net.double()(x.double())
    dp, dv = (pri.cpu().double() - torch.softmax(lg, -1)).abs().max().item(), (v.cpu().double() - v64.squeeze(1)).abs().max().item()
    assert dp <= 2e-6 and dv <= 4e-6, (boards, dp, dv)


def _check_pre_relu_record(bnd, device, S, C):
    """The convolution epilogues record |v| IN FRONT of the ReLU (include/azsp.h, RANGE: deliberately conservative): a strongly negative
    bias drives one channel's pre-activation below -65504; the ReLU zeroes the output, and the event is still counted with its
    magnitude -- by the kernels and by the host twin alike (they disagreed)."""
    import ctypes

    dll = bnd.dll
    ev, mx = ctypes.c_uint32(0), ctypes.c_float(0.0)
    assert dll.azsp_split_range_status(None, None, 1, None) == 0
    x, r, w, b = _inputs(3, C, S, 21)
    b = b.clone()
    b[5] = -1.0e5
    y, _ = _run_split_conv(bnd, x, None, w, b, 1, device)
    assert dll.azsp_split_range_status(ctypes.byref(ev), ctypes.byref(mx), 1, None) == 0
    ref = _ref64(x, None, w, b, 0)  # pre-activations
    assert ev.value >= 1 and abs(mx.value - ref[:, 5].abs().max().item()) <= 1e-2 * 1e5 and mx.value > 65504.0, (ev.value, mx.value)
    assert float(y[:, 5].abs().max()) == 0.0 and torch.isfinite(y).all()
    others = [c for c in range(C) if c != 5]
    err = (y[:, others].double() - torch.relu(ref[:, others])).abs().max().item() / ref.abs().max().item()
    assert err <= 2e-6, err
    # without the ReLU the same value is clamped to -65504 and recorded once more
    y0, _ = _run_split_conv(bnd, x, None, w, b, 0, device)
    assert dll.azsp_split_range_status(ctypes.byref(ev), ctypes.byref(mx), 1, None) == 0
    assert ev.value >= 1 and float(y0[:, 5].min()) == -65504.0


@pytest.mark.parametrize("S,C", [(9, 64), (17, 64)])
def test_split_range_record_is_taken_in_front_of_the_relu_host_twin(S, C):
    import engine_util as eu

    _check_pre_relu_record(eu.hosttwin_binding(), "cpu", S, C)


@pytest.mark.gpu
@pytest.mark.parametrize("S,C", [(9, 128), (9, 64), (17, 64)])
def test_gpu_split_range_record_is_taken_in_front_of_the_relu(S, C):
    from alpha_zero_amd import _lib

    _check_pre_relu_record(_lib.load(), "cuda", S, C)


@pytest.mark.gpu
@pytest.mark.parametrize("S,C", [(9, 128), (9, 64), (17, 64)])
def test_gpu_split_range_record_trips_and_resets(S, C):
    """The +-65504 clamp of the split kernels is OBSERVABLE: in-range launches leave the sticky record at zero; an
    activation that leaves f16's range inside the convolution's epilogue (here: a 9-tap sum of 3e4-sized inputs, and a residual that
    pushes one output over) is clamped, counted and its magnitude reported; the layout conversion reports out-of-range inputs; reset
    clears the record.  The clamped result itself is finite (65504), never inf / NaN."""
    import ctypes

    from alpha_zero_amd import _lib

    bnd = _lib.load()
    dll = bnd.dll

    def status(reset=0):
        ev, mx = ctypes.c_uint32(0), ctypes.c_float(0.0)
        assert dll.azsp_split_range_status(ctypes.byref(ev), ctypes.byref(mx), reset, None) == 0
        return ev.value, mx.value

    status(1)
    x, r, w, b = _inputs(5, C, S, 11)
    y, _ = _run_split_conv(bnd, x, r, w, b, 1, "cuda")
    assert status() == (0, 0.0)
    # identity-like filter bank (centre tap of channel c -> c, weight 4): y = 4 x + b (+ r)
    w2 = torch.zeros(C, C, 3, 3)
    w2[torch.arange(C), torch.arange(C), 1, 1] = 4.0
    x2 = torch.full((5, C, S, S), 100.0)
    x2[3, 5, S - 1, 0] = 3e4          # 4 * 3e4 = 1.2e5 > 65504 in the epilogue of the convolution (the 9x9 kernel's corner position)
    x2[1, 7, 2, 3] = 2e4              # 8e4: an ordinary position
    b2 = torch.zeros(C)
    y2, _ = _run_split_conv(bnd, x2, None, w2, b2, 1, "cuda")
    ev, mx = status()
    assert ev >= 2 and mx == 120000.0, (ev, mx)
    assert torch.isfinite(y2).all() and y2[3, 5, S - 1, 0] == 65504.0 and y2[1, 7, 2, 3] == 65504.0 and abs(y2[0, 0, 0, 0].item() - 400.0) < 1e-3
    assert status(1)[0] == ev and status() == (0, 0.0)
    # the conversion into the split layout reports what it clamps, too
    x3 = x.clone()
    x3[2, 1, 0, 0] = -1e6
    xs = torch.zeros(dll.azsp_split_bytes(5, S, C) // 2, dtype=torch.float16, device="cuda")
    assert dll.azsp_split_layout(x3.cuda().contiguous(memory_format=torch.channels_last).data_ptr(), xs.data_ptr(), 5, S, C, 1, None, None) == 0
    assert status(1) == (1, 1e6)


def test_split_feature_tensor_round_trip_host():
    """engine_util.split_features builds the AZSP_FEAT_F16_SPLIT tensor the engine writes (checked by unsplit_features' own assertions), and
    the host twin's forward on it equals its forward on the fp32 planes (azsp_stem_split_exact == azsp_split_features + azsp_stem_split)."""
    import engine_util as eu

    x = (torch.rand(3, 17, 9, 9, generator=torch.Generator().manual_seed(5)) > 0.6).float()
    t = eu.split_features(x)
    assert (torch.from_numpy(eu.unsplit_features(t, 3, 9)).float() == x).all()
    net = _trained_like_net(64, 1)
    inf = InferenceNet(net, dtype=torch.float32, binding=eu.hosttwin_binding())
    p0, v0 = inf.forward_split(x)
    p1, v1 = inf.forward_split(t, split_features=(3, 9))
    assert torch.equal(p0, p1) and torch.equal(v0, v1)


@pytest.mark.gpu
@pytest.mark.parametrize("game,n,filters", [("go", 9, 128), ("go", 9, 64), ("gomoku", 13, 64)])
def test_gpu_stem_on_engine_written_features_equals_the_general_stem(game, n, filters):
    """azsp_stem_split_exact on the engine's AZSP_FEAT_F16_SPLIT features (hi plane only: the lo plane of 0 / 1 planes is zero, so its
    loads and its product are skipped) gives bit for bit the network outputs of azsp_split_features + azsp_stem_split on the same planes."""
    import engine_util as eu
    from alpha_zero_amd import _lib

    net = _trained_like_net(filters, 2) if game == "go" else _trained_like_gomoku_net(filters, 2)
    inf = InferenceNet(net, dtype=torch.float32, binding=_lib.load()).cuda()
    for rows in (1, 5, 300, 1031):
        x = (torch.rand(rows, 17, n, n, generator=torch.Generator().manual_seed(rows)) > 0.6).float()
        p0, v0 = inf.forward_split(x.cuda().contiguous())
        p1, v1 = inf.forward_split(eu.split_features(x).cuda(), split_features=(rows, n))
        assert torch.equal(p0, p1) and torch.equal(v0, v1), (game, n, filters, rows, float((p0 - p1).abs().max()))


# ---- fused split-precision ResNetBlock at 17x17 x 64 (az_resblock_sp17.h, azsp_resblock_split) ------------------------------------
def _resblock_inputs(boards, seed, S=17):
    g = torch.Generator().manual_seed(seed)
    C = 64
    x = torch.randn(boards, C, S, S, generator=g)
    x = torch.where(torch.rand(boards, C, S, S, generator=g) < 0.5, torch.zeros(()), x.abs())
    x[:, :8] *= 37.0
    x[:, -8:] *= 3e-3
    ws = [torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5 for _ in range(2)]
    bs = [torch.randn(C, generator=g) * 0.1 for _ in range(2)]
    return x, ws, bs


def _resblock_both_ways(bnd, x, ws, bs, device):
    """(fused y, two-launch y) in the split layout, as raw f16 tensors, + the fused result as fp32 NCHW."""
    B, C, S, _ = x.shape
    dll = bnd.dll
    n = dll.azsp_split_bytes(B, S, C) // 2
    xs, ms, y2, yf = (torch.zeros(n, dtype=torch.float16, device=device) for _ in range(4))
    xc = x.to(device).contiguous(memory_format=torch.channels_last)
    assert dll.azsp_split_layout(xc.data_ptr(), xs.data_ptr(), B, S, C, 1, None, None) == 0
    wsp = [split_weights_f16(w).to(device) for w in ws]
    bb = [b.float().to(device) for b in bs]
    assert dll.azsp_conv3x3_split(xs.data_ptr(), wsp[0].data_ptr(), bb[0].data_ptr(), None, ms.data_ptr(), B, S, C, 1, None, None) == 0
    assert dll.azsp_conv3x3_split(ms.data_ptr(), wsp[1].data_ptr(), bb[1].data_ptr(), xs.data_ptr(), y2.data_ptr(), B, S, C, 1, None, None) == 0
    assert dll.azsp_resblock_split(xs.data_ptr(), wsp[0].data_ptr(), bb[0].data_ptr(), wsp[1].data_ptr(), bb[1].data_ptr(), yf.data_ptr(), B, S, C, None, None) == 0
    y = torch.empty_like(xc)
    assert dll.azsp_split_layout(yf.data_ptr(), y.data_ptr(), B, S, C, 0, None, None) == 0
    if device != "cpu":
        torch.cuda.synchronize()
    return yf, y2, y.cpu().contiguous()


def test_split_resblock_abi_host_twin():
    """azsp_resblock_split through the ABI on the host twin: equals two azsp_conv3x3_split calls; argument checks (y == x refused,
    shapes without a fused kernel refused)."""
    import engine_util as eu

    bnd = eu.hosttwin_binding()
    x, ws, bs = _resblock_inputs(2, 3)
    yf, y2, y = _resblock_both_ways(bnd, x, ws, bs, "cpu")
    assert torch.equal(yf, y2)
    mid = torch.relu(F.conv2d(x.double(), ws[0].double(), bs[0].double(), padding=1))
    ref = torch.relu(F.conv2d(mid, ws[1].double(), bs[1].double(), padding=1) + x.double())
    assert ((y.double() - ref).abs().max() / ref.abs().max()).item() <= 2e-6
    z = torch.zeros(2 * 2 * 17 * 17 * 64, dtype=torch.float16)
    assert bnd.dll.azsp_resblock_split(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), 1, 17, 64, None, None) != 0  # y == x
    zo = z.clone()
    assert bnd.dll.azsp_resblock_split(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), zo.data_ptr(), 1, 13, 64, None, None) != 0
    assert bnd.dll.azsp_resblock_split(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), zo.data_ptr(), 1, 17, 128, None, None) != 0
    assert bnd.dll.azsp_resblock_split(z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), zo.data_ptr(), 1, 9, 128, None, None) != 0
    # the 9x9 x 64 shape (round 6) through the ABI on the twin: equals the two launches, 1 - 3 boards
    for nb in (1, 2, 3):
        x9, ws9, bs9 = _resblock_inputs(nb, 40 + nb, S=9)
        yf9, y29, _ = _resblock_both_ways(bnd, x9, ws9, bs9, "cpu")
        assert torch.equal(yf9, y29)
    assert bnd.dll.azsp_resblock_split(None, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(), zo.data_ptr(), 1, 17, 64, None, None) != 0


@pytest.mark.gpu
@pytest.mark.parametrize("boards", [1, 2, 3, 255, 256, 257, 513, 1100])
def test_gpu_split_resblock17_bit_identical_to_two_launches(boards):
    """k_resblock_sp17 (one launch per ResNetBlock, intermediate activation in LDS, half-board tiles with a recomputed halo row) gives
    bit for bit the split-layout output of two k_conv3x3_sp17 launches -- every position's MFMA order and every rounding are the same.
    Board counts around one / two / four boards per workgroup slot (256 slots) exercise the first-board, has-next and last-board paths
    of the persistent loop.  Also against fp64: the block's error is of the size of two fp32 convolutions' round-off."""
    from alpha_zero_amd import _lib

    x, ws, bs = _resblock_inputs(boards, 100 + boards)
    yf, y2, y = _resblock_both_ways(_lib.load(), x, ws, bs, "cuda")
    if not torch.equal(yf, y2):
        B, C, S = boards, 64, 17
        d = (yf.view(B, 2, C // 8, S * S, 8) != y2.view(B, 2, C // 8, S * S, 8)).any(dim=4).any(dim=1)  # [B, chunk, pos]
        bad = d.nonzero()
        raise AssertionError(f"{len(bad)} (board, chunk, position) cells differ; first {bad[:12].tolist()}; rows {sorted(set((bad[:, 2] // 17).tolist()))[:20]}")
    mid = torch.relu(F.conv2d(x.double(), ws[0].double(), bs[0].double(), padding=1))
    ref = torch.relu(F.conv2d(mid, ws[1].double(), bs[1].double(), padding=1) + x.double())
    assert ((y.double() - ref).abs().max() / ref.abs().max()).item() <= 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("boards", [1, 2, 3, 4, 5, 511, 512, 513, 514, 1025, 2181])
def test_gpu_split_resblock9_bit_identical_to_two_launches(boards):
    """k_resblock_sp<Sb9> (round 6: the fused fp32-class block for 9x9 x 64, the reference's logs/go/9x9_12b64 shape; TWO boards per tile,
    stacked with a separator row, the intermediate activation in LDS, the skip from the x image in LDS) gives bit for bit the output of
    two k_conv3x3_sp launches.  Odd board counts exercise the unfused last board; counts around one / two / four pairs per workgroup
    slot (256 slots) the first-tile, has-next and last-tile paths.  Also against fp64."""
    from alpha_zero_amd import _lib

    x, ws, bs = _resblock_inputs(boards, 300 + boards, S=9)
    yf, y2, y = _resblock_both_ways(_lib.load(), x, ws, bs, "cuda")
    if not torch.equal(yf, y2):
        B, C, S = boards, 64, 9
        d = (yf.view(B, 2, C // 8, S * S, 8) != y2.view(B, 2, C // 8, S * S, 8)).any(dim=4).any(dim=1)  # [B, chunk, pos]
        bad = d.nonzero()
        raise AssertionError(f"{len(bad)} (board, chunk, position) cells differ; first {bad[:12].tolist()}; boards {sorted(set(bad[:, 0].tolist()))[:20]}")
    mid = torch.relu(F.conv2d(x.double(), ws[0].double(), bs[0].double(), padding=1))
    ref = torch.relu(F.conv2d(mid, ws[1].double(), bs[1].double(), padding=1) + x.double())
    assert ((y.double() - ref).abs().max() / ref.abs().max()).item() <= 2e-6


@pytest.mark.gpu
def test_gpu_split_resblock9_records_intermediate_overflow():
    """The intermediate activation of the fused 9x9 block is split inside the kernel, too: a value beyond f16's range in m is recorded."""
    import ctypes

    from alpha_zero_amd import _lib

    bnd = _lib.load()
    dll = bnd.dll
    assert dll.azsp_split_range_status(None, None, 1, None) == 0
    x, ws, bs = _resblock_inputs(6, 77, S=9)
    bs[0] = bs[0].clone()
    bs[0][3] = 3.0e5  # conv1's bias pushes one channel of m beyond 65504
    yf, y2, _ = _resblock_both_ways(bnd, x, ws, bs, "cuda")
    ev, mx = ctypes.c_uint32(0), ctypes.c_float(0.0)
    assert dll.azsp_split_range_status(ctypes.byref(ev), ctypes.byref(mx), 1, None) == 0
    assert ev.value > 0 and mx.value > 65504.0 and torch.equal(yf, y2)


@pytest.mark.gpu
def test_gpu_split_resblock17_records_intermediate_overflow():
    """The intermediate activation of the fused block is clamped and recorded exactly like the output of a first launch would be: a
    block whose FIRST convolution leaves f16's range reports events in the caller's record although that tensor never reaches HBM."""
    from alpha_zero_amd import _lib

    dll = _lib.load().dll
    x, ws, bs = _resblock_inputs(5, 9)
    ws[0] = ws[0] * 4000.0   # m ~ 1e5 .. 1e6
    ws[1] = ws[1] / 4000.0
    B, C, S = 5, 64, 17
    n = dll.azsp_split_bytes(B, S, C) // 2
    xs, yf = torch.zeros(n, dtype=torch.float16, device="cuda"), torch.zeros(n, dtype=torch.float16, device="cuda")
    xc = x.cuda().contiguous(memory_format=torch.channels_last)
    rec = torch.zeros(2, dtype=torch.int32, device="cuda")
    assert dll.azsp_split_layout(xc.data_ptr(), xs.data_ptr(), B, S, C, 1, rec.data_ptr(), None) == 0
    wsp = [split_weights_f16(w).cuda() for w in ws]
    bb = [b.float().cuda() for b in bs]
    assert dll.azsp_resblock_split(xs.data_ptr(), wsp[0].data_ptr(), bb[0].data_ptr(), wsp[1].data_ptr(), bb[1].data_ptr(), yf.data_ptr(), B, S, C,
                                   rec.data_ptr(), None) == 0
    torch.cuda.synchronize()
    r = rec.cpu()
    assert int(r[0]) > 0 and float(r[1:].view(torch.float32)[0]) > 65504.0


# ---- round 6: the wave-per-tile kernel k_conv3x3_spg (small batches of the tailored shapes, every other shape) -------------------------
def _raw_split_conv(bnd, x, r, w, b, relu, device="cuda"):
    """azsp_conv3x3_split on fp32 NCHW inputs; returns the raw split-layout output (f16 words) and the joined fp32 NCHW output."""
    B, C, S, _ = x.shape
    dll = bnd.dll
    n = dll.azsp_split_bytes(B, S, C) // 2
    xs, ys = torch.zeros(n, dtype=torch.float16, device=device), torch.zeros(n, dtype=torch.float16, device=device)
    xc = x.to(device).contiguous(memory_format=torch.channels_last)
    assert dll.azsp_split_layout(xc.data_ptr(), xs.data_ptr(), B, S, C, 1, None, None) == 0
    rs = None
    if r is not None:
        rs = torch.zeros(n, dtype=torch.float16, device=device)
        assert dll.azsp_split_layout(r.to(device).contiguous(memory_format=torch.channels_last).data_ptr(), rs.data_ptr(), B, S, C, 1, None, None) == 0
    wsp, bb = split_weights_f16(w).to(device), b.float().to(device)
    assert dll.azsp_conv3x3_split(xs.data_ptr(), wsp.data_ptr(), bb.data_ptr(), rs.data_ptr() if rs is not None else None, ys.data_ptr(),
                                  B, S, C, relu, None, None) == 0
    y = torch.empty_like(xc)
    assert dll.azsp_split_layout(ys.data_ptr(), y.data_ptr(), B, S, C, 0, None, None) == 0
    if device != "cpu":
        torch.cuda.synchronize()
    return ys.cpu(), y.cpu().contiguous()


def test_small_batch_knob_host_twin():
    """azsp_small_batch_waves: sets, returns the previous value, a negative argument only queries (the twin's loops ignore it)."""
    import engine_util as eu

    b, _ = eu.backend("host")
    old = b.dll.azsp_small_batch_waves(-1)
    assert b.dll.azsp_small_batch_waves(7) == old and b.dll.azsp_small_batch_waves(-1) == 7
    assert b.dll.azsp_small_batch_waves(old) == 7 and b.dll.azsp_small_batch_waves(-5) == old
    # shapes beyond the tailored ones go through the ABI since round 6 (19x19 x 64 here: the twin's plain loops vs fp64)
    x, r, w, bb = _inputs(2, 64, 19, 5)
    y, rt = _run_split_conv(b, x, r, w, bb, 1, "cpu")
    ref = _ref64(x, r, w, bb, 1)
    assert ((y.double() - ref).abs().max() / ref.abs().max()).item() <= 8e-7 and rt <= 2.0 ** -21


@pytest.mark.gpu
@pytest.mark.parametrize("S,C", [(9, 128), (9, 64), (17, 64)])
def test_gpu_wave_per_tile_conv_is_bit_identical_to_the_weight_stationary_kernels(S, C):
    """k_conv3x3_spg (one wave per 16-cout x 32-position tile, fragments straight from global memory) against the tailored kernel of the
    same shape -- k_conv3x3_sp2 (two accumulation chains), k_conv3x3_sp<.., 8, 1>, k_conv3x3_sp17 -- on the same inputs: the raw hi / lo
    f16 words of the output are equal, with and without residual and ReLU.  The evaluator's result for a position therefore does not
    depend on whether it arrives in a batch of 1 (drop-in uct_search) or of 32 768 (the self-play actor)."""
    from alpha_zero_amd import _lib

    bnd = _lib.load()
    old = bnd.dll.azsp_small_batch_waves(-1)
    try:
        for boards in (1, 3, 19):
            for res, relu in ((False, 1), (True, 1), (True, 0)):
                x, r, w, b = _inputs(boards, C, S, 300 + boards)
                if not relu:
                    x = x - 0.3
                bnd.dll.azsp_small_batch_waves(0)
                ya, fa = _raw_split_conv(bnd, x, r if res else None, w, b, relu)
                bnd.dll.azsp_small_batch_waves(1 << 20)
                yb, fb = _raw_split_conv(bnd, x, r if res else None, w, b, relu)
                if not torch.equal(ya, yb):
                    d = (ya.view(boards, 2, C // 8, S * S, 8) != yb.view(boards, 2, C // 8, S * S, 8)).any(dim=4).any(dim=1).nonzero()
                    raise AssertionError(f"S={S} C={C} boards={boards} res={res} relu={relu}: {len(d)} cells differ; first {d[:10].tolist()}; "
                                         f"max |d| = {(fa - fb).abs().max().item():.3g}")
    finally:
        bnd.dll.azsp_small_batch_waves(old)


@pytest.mark.gpu
@pytest.mark.parametrize("S", [17, 9])
def test_gpu_small_batch_resblock_is_bit_identical_to_the_fused_block(S):
    """azsp_resblock_split on a handful of boards = two wave-per-tile convolutions through a scratch buffer: the same bits as the fused
    one-launch block (which is itself bit-identical to two weight-stationary launches)."""
    from alpha_zero_amd import _lib

    bnd = _lib.load()
    old = bnd.dll.azsp_small_batch_waves(-1)
    try:
        for boards in (1, 2, 5, 25):
            x, ws, bs = _resblock_inputs(boards, 400 + boards, S)
            bnd.dll.azsp_small_batch_waves(0)
            yf0, y20, _ = _resblock_both_ways(bnd, x, ws, bs, "cuda")
            bnd.dll.azsp_small_batch_waves(1 << 20)
            yf1, y21, _ = _resblock_both_ways(bnd, x, ws, bs, "cuda")
            assert torch.equal(yf0, y20) and torch.equal(yf1, y21) and torch.equal(yf0, yf1), (S, boards)
    finally:
        bnd.dll.azsp_small_batch_waves(old)


@pytest.mark.gpu
@pytest.mark.parametrize("S,C,boards", [(19, 256, 2), (19, 256, 70), (19, 128, 5), (19, 64, 3), (13, 64, 4), (13, 128, 33), (5, 64, 7), (25, 64, 2)])
def test_gpu_wave_per_tile_conv_on_shapes_without_a_tailored_kernel(S, C, boards):
    """The fp32-class convolution for EVERY plane size (the reference's 19x19 x 256 jumbo tower at its own precision,
    alpha_zero/training_go_jumbo.py:46-47; 13x13 Go; 5x5 test boards): against fp64 with the bound of the tailored kernels (8e-7 of
    max |y|; 1.2e-6 at 256 filters, twice the products per output), the library's fp32 error beside it; the two tile shapes of the kernel (latency / throughput) give the same bits."""
    from alpha_zero_amd import _lib

    bnd = _lib.load()
    old = bnd.dll.azsp_small_batch_waves(-1)
    try:
        for res, relu in ((False, 1), (True, 1), (True, 0)):
            x, r, w, b = _inputs(boards, C, S, 500 + boards)
            if not relu:
                x = x - 0.3
            bnd.dll.azsp_small_batch_waves(0)        # k_conv3x3_spgw: 48 positions per wave, B fragments shared by the workgroup through LDS
            ya, y = _raw_split_conv(bnd, x, r if res else None, w, b, relu)
            bnd.dll.azsp_small_batch_waves(1 << 20)  # k_conv3x3_spg: 16 couts x 32 positions per wave, no LDS
            yb, _ = _raw_split_conv(bnd, x, r if res else None, w, b, relu)
            assert torch.equal(ya, yb), (S, C, boards, res, relu)
            ref = _ref64(x, r if res else None, w, b, relu)
            lib = F.conv2d(x.cuda(), w.cuda(), b.cuda(), padding=1)
            if res:
                lib = lib + r.cuda()
            lib = (torch.relu(lib) if relu else lib).cpu()
            scale = ref.abs().max().item()
            err, lib_err = (y.double() - ref).abs().max().item() / scale, (lib.double() - ref).abs().max().item() / scale
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "split_conv_generic_error.jsonl"), "a") as f:
                f.write(json.dumps(dict(S=S, C=C, boards=boards, residual=res, relu=relu, err=err, library_fp32_err=lib_err)) + "\n")
            assert err <= (8e-7 if C <= 128 else 1.2e-6), (S, C, boards, res, relu, err, lib_err)  # (2304 products per output at 256 filters)
    finally:
        bnd.dll.azsp_small_batch_waves(old)
