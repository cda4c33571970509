"""Policy/value ResNet of the self-play path on PyTorch-ROCm (reference: alpha_zero/core/network.py:85-173).

`AlphaZeroNet` keeps the reference's module tree, so its state_dict keys line up one-to-one
(`conv_block.{0,1}`, `res_blocks.{i}.conv_block{1,2}.{0,1}`, `policy_head.{0,1,4}`,
`value_head.{0,1,4,6}`) and shipped / learner checkpoints load unchanged.

`InferenceNet` is the leaf evaluator the engine calls every round: eval-mode BatchNorm folded into
the preceding convolution, softmax over all A actions in fp32 (priors are never masked or
renormalised: pipeline.py:102-117), outputs written straight into the engine's priors/values
tensors.  The matrix cores are used here and only here -- by libazsp's hand-written MFMA kernels:
the fp32-class split-precision stem / tower / heads (default: the reference's precision class;
9x9 x {128, 64}, 13x13 Gomoku x 64), the tiled bf16 / f16 families (opt-in; also 19x19 x 256);
other shapes fall back to the library's convolutions (MIOpen) + the fused azsp_bias_act epilogue,
announced by `evaluator_path` and a RuntimeWarning.
"""
from typing import Tuple

from contextlib import nullcontext as _nullcontext

import torch
import torch.nn.functional as F
from torch import nn


def _conv_bn(cin, cout, k, pad):
    return [nn.Conv2d(cin, cout, kernel_size=k, stride=1, padding=pad, bias=False), nn.BatchNorm2d(cout)]


class ResNetBlock(nn.Module):
    """conv-BN-ReLU-conv-BN, + skip, ReLU (network.py:42-82)."""

    def __init__(self, num_filters: int) -> None:
        super().__init__()
        self.conv_block1 = nn.Sequential(*_conv_bn(num_filters, num_filters, 3, 1), nn.ReLU())
        self.conv_block2 = nn.Sequential(*_conv_bn(num_filters, num_filters, 3, 1))

    def forward(self, x):
        return F.relu(self.conv_block2(self.conv_block1(x)) + x)


class AlphaZeroNet(nn.Module):
    def __init__(self, input_shape: Tuple, num_actions: int, num_res_block: int = 19, num_filters: int = 256,
                 num_fc_units: int = 256, gomoku: bool = False) -> None:
        super().__init__()
        c, h, w = input_shape
        pad = 3 if gomoku else 1  # network.py:101-105: Gomoku pads the stem by 3 -> spatial size grows by 4
        oh, ow = h + 2 * pad - 2, w + 2 * pad - 2
        self.conv_block = nn.Sequential(*_conv_bn(c, num_filters, 3, pad), nn.ReLU())
        self.res_blocks = nn.Sequential(*[ResNetBlock(num_filters) for _ in range(num_res_block)])
        self.policy_head = nn.Sequential(*_conv_bn(num_filters, 2, 1, 0), nn.ReLU(), nn.Flatten(), nn.Linear(2 * oh * ow, num_actions))
        self.value_head = nn.Sequential(*_conv_bn(num_filters, 1, 1, 0), nn.ReLU(), nn.Flatten(), nn.Linear(oh * ow, num_fc_units),
                                        nn.ReLU(), nn.Linear(num_fc_units, 1), nn.Tanh())
        for m in self.modules():  # network.py:30-39
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.kaiming_uniform_(m.weight, nonlinearity="relu")
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x):
        f = self.res_blocks(self.conv_block(x))
        return self.policy_head(f), self.value_head(f)


@torch.no_grad()
def widen_network(net: "AlphaZeroNet", num_filters: int) -> "AlphaZeroNet":
    """Function-preserving copy of `net` with `num_filters` >= its own filter count: the extra channels have zero convolution
    weights and identity BatchNorm (mean 0, var 1, gamma 1, beta 0), so they carry exact zeros through every ReLU and contribute
    nothing to the heads.  Lets a trained network of an unsupported width (e.g. the reference's shipped 10 x 40 Gomoku checkpoint)
    run on the hand-written evaluator kernels of the next supported width (64)."""
    c0 = net.conv_block[0].out_channels
    if num_filters < c0:
        raise ValueError(f"cannot narrow {c0} filters to {num_filters}")
    cin, h, w = net.conv_block[0].in_channels, None, None
    A, fc = net.policy_head[4].out_features, net.value_head[4].out_features
    oh_ow = net.value_head[4].in_features
    side = int(round(oh_ow ** 0.5))
    pad = net.conv_block[0].padding[0]
    n = side - 2 * pad + 2
    out = AlphaZeroNet((cin, n, n), A, len(net.res_blocks), num_filters, fc, gomoku=(pad == 3)).eval()

    def conv(dst, src, pad_in=True):
        dst.weight.zero_()
        dst.weight[: src.weight.shape[0], : src.weight.shape[1]] = src.weight

    def bn(dst, src):
        k = src.num_features
        dst.weight.fill_(1.0), dst.bias.zero_(), dst.running_mean.zero_(), dst.running_var.fill_(1.0)
        dst.weight[:k], dst.bias[:k], dst.running_mean[:k], dst.running_var[:k] = src.weight, src.bias, src.running_mean, src.running_var
        dst.eps, dst.num_batches_tracked = src.eps, src.num_batches_tracked.clone()

    conv(out.conv_block[0], net.conv_block[0]), bn(out.conv_block[1], net.conv_block[1])
    for bo, bi in zip(out.res_blocks, net.res_blocks):
        conv(bo.conv_block1[0], bi.conv_block1[0]), bn(bo.conv_block1[1], bi.conv_block1[1])
        conv(bo.conv_block2[0], bi.conv_block2[0]), bn(bo.conv_block2[1], bi.conv_block2[1])
    conv(out.policy_head[0], net.policy_head[0]), bn(out.policy_head[1], net.policy_head[1])
    conv(out.value_head[0], net.value_head[0]), bn(out.value_head[1], net.value_head[1])
    for i, head in ((4, "policy_head"), (4, "value_head"), (6, "value_head")):
        getattr(out, head)[i].load_state_dict(getattr(net, head)[i].state_dict())
    return out


F16_MAX = 65504.0

# tower widths with hand-written evaluator kernels per (board size, stem padding); dtype class "fp32" = the split-precision kernels,
# "lowp" = the bf16 / f16 tiled kernels (see InferenceNet.supports_split_features / supports_tiled_features)
KERNEL_WIDTHS = {"fp32": {(9, 1): (64, 128), (13, 3): (64,)}, "lowp": {(9, 1): (64, 128), (13, 3): (64,), (19, 1): (256,)}}


# widening multiplies the tower's FLOPs by (w / f)^2; the hand-written kernels are 4.5 - 5x the library's fp32 convolutions (measured: 40 -> 64 =
# 2.56x the FLOPs is still 1.7x faster than the library at 40), so anything beyond 4x the FLOPs is left to the library path
MAX_WIDEN_FLOPS_RATIO = 4.0


def widen_for_kernels(net: "AlphaZeroNet", board_size: int, dtype):
    """The network to hand to InferenceNet so that its evaluation runs on hand-written kernels: `net` itself when its filter count has
    kernels for this board (or none could help), else a function-preserving widened copy (widen_network: zero-weight extra channels,
    identity BatchNorm -- exactly the same function) at the next supported width.  This is how the reference's shipped networks of other
    widths (13x13 Gomoku: 10 x 40, training_gomoku.py:37-38) reach the kernels by default instead of the library fallback; the price is
    the extra FLOPs of the wider tower (40 -> 64: 2.56x), still 1.7x faster than the library's fp32 convolutions at the true width.
    Returns (network, note) with note = '' or a description for `evaluator_path`."""
    cls = "fp32" if dtype == torch.float32 else "lowp"
    pad = net.conv_block[0].padding[0]
    widths = KERNEL_WIDTHS[cls].get((board_size, pad), ())
    f = net.conv_block[0].out_channels
    if dtype == torch.float16:  # the f16 variants exist for the 9x9 x 128 evaluator with 82 actions and 128 fully connected units only
        return net, ""
    if f in widths or not widths or f > max(widths):
        return net, ""
    w = min(v for v in widths if v >= f)
    if (w / f) ** 2 > MAX_WIDEN_FLOPS_RATIO:  # e.g. 64 -> 256 at 19x19: 16x the FLOPs of the tower -- the library at the true width is faster
        return net, ""
    return widen_network(net, w), f" (network widened {f} -> {w} filters, function-preserving)"


def split_weights_f16(w):
    """Convolution weights [Cout,Cin,3,3] fp32 -> the packing of the split-precision kernels (include/azsp.h, azsp_conv3x3_split):
    [plane: hi, lo][tap = ky*3+kx][Cout][Cin] f16 with hi = f16(w) and lo = f16((w - hi) * 2048).  A (BatchNorm-folded) weight that is
    not finite or lies beyond f16's finite range cannot be carried by the split format: ValueError, never a silent clamp / inf."""
    if not bool(torch.isfinite(w).all()) or float(w.abs().max()) > F16_MAX:
        raise ValueError(f"split-precision packing: folded convolution weights must be finite and within +-{F16_MAX} "
                         f"(max |w| = {float(w.abs().max())}); evaluate this network with use_split_tower = False (library fp32)")
    w9 = w.permute(2, 3, 0, 1).reshape(9, w.shape[0], w.shape[1]).float()
    hi = w9.to(torch.float16)
    lo = ((w9 - hi.float()) * 2048.0).to(torch.float16)
    return torch.stack([hi, lo]).contiguous()


def _fold(conv: nn.Conv2d, bn: nn.BatchNorm2d):
    """Eval-mode BN(conv(x)) == conv'(x) + b'."""
    s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return conv.weight * s.view(-1, 1, 1, 1), bn.bias - bn.running_mean * s


class InferenceNet(nn.Module):
    """Frozen, BN-folded, channels-last copy of an AlphaZeroNet for the engine's leaf batches."""

    def __init__(self, net: AlphaZeroNet, dtype=torch.bfloat16, channels_last=True, binding=None):
        """binding: the libazsp Binding; when given (and the activations are channels-last on the GPU) every 3x3
        convolution is followed by ONE fused kernel (azsp_bias_act: bias + residual + ReLU) instead of the separate
        bias-add / add / clamp passes PyTorch would launch."""
        super().__init__()
        net = net.eval()
        self.dtype = dtype
        self.binding = binding if channels_last else None
        self.use_fused_conv = True
        self.use_tiled_tower = True
        self.use_fused_block = True  # 64-filter towers: azsp_resblock_tiled instead of two azsp_conv3x3_tiled launches per block
        self.use_split_tower = True  # fp32 networks: azsp_conv3x3_split (hi + lo f16 pairs, three MFMA products) instead of the library
        self.use_split_heads = True  # ... and azsp_split_features / azsp_stem_split / azsp_head_split around it (whole evaluator hand-written)
        self.mf = torch.channels_last if channels_last else torch.contiguous_format
        self.stem_pad = net.conv_block[0].padding[0]
        with torch.no_grad():
            convs = [_fold(net.conv_block[0], net.conv_block[1])]
            for blk in net.res_blocks:
                convs.append(_fold(blk.conv_block1[0], blk.conv_block1[1]))
                convs.append(_fold(blk.conv_block2[0], blk.conv_block2[1]))
            self.n_blocks = len(net.res_blocks)
            # hand-written MFMA convolutions (azsp_conv3x3_tiled): weights as [tap = ky*3+kx][cout][cin] bf16, bias fp32
            self.filters = net.conv_block[0].out_channels
            # element format of the tiled kernels: bf16 (default) or f16 (dtype = torch.float16: azsp_*_f16, same MFMA rate, 3 more significand bits)
            pk = self.pack_dtype = torch.float16 if dtype == torch.float16 else torch.bfloat16
            self.wp = nn.ParameterList([nn.Parameter(w.permute(2, 3, 0, 1).reshape(9, w.shape[0], w.shape[1]).to(pk).contiguous(),
                                                     requires_grad=False) for w, _ in convs[1:]])
            self.b32 = nn.ParameterList([nn.Parameter(b.float().contiguous(), requires_grad=False) for _, b in convs[1:]])
            # fp32-class path: every activation is carried as v * 2^-act_shift (an exact rescaling of a ReLU + skip tower, see set_act_shift)
            self.act_shift, self.act_calibrated, self.act_max_abs = 0, False, 0.0
            self.split_fallback_reason = ""  # set when the fp32-class kernels are given up for this network (library fp32 instead)
            self.stem_fallback_reason = ""   # set when only the split STEM cannot carry this network (library stem + heads around the split tower)
            if dtype == torch.float32:  # split-precision tower (azsp_conv3x3_split)
                try:
                    packed = [split_weights_f16(w) for w, _ in convs[1:]]
                except ValueError as e:  # a folded tower weight the f16-pair format cannot carry: this network runs on the library
                    packed = [torch.zeros(2, 9, w.shape[0], w.shape[1], dtype=torch.float16) for w, _ in convs[1:]]
                    self.split_fallback_reason = str(e).split(";")[0]
                self.wsp = nn.ParameterList([nn.Parameter(p, requires_grad=False) for p in packed])
                # tower biases as the split kernels see them: b * 2^-act_shift (set_act_shift); b32 keeps the unscaled values
                self.b_sp = nn.ParameterList([nn.Parameter(b.float().clone().contiguous(), requires_grad=False) for _, b in convs[1:]])
                # this network's own range record (include/azsp.h: range_rec_dev): [events, bits of the largest |v|]
                self.register_buffer("range_rec", torch.zeros(2, dtype=torch.int32), persistent=False)
            self.w = nn.ParameterList([nn.Parameter(w.to(dtype).contiguous(memory_format=self.mf), requires_grad=False) for w, _ in convs])
            self.b = nn.ParameterList([nn.Parameter(b.to(dtype), requires_grad=False) for _, b in convs])
            # stem for the tiled path (azsp_stem_tiled): [tap][cout][32 in], input channels 17.. zero
            sw = convs[0][0]
            self.stem_ok = sw.shape[1] <= 32 and sw.shape[2] == 3 and sw.shape[3] == 3 and self.stem_pad in (1, 3)
            if self.stem_ok:
                swp = torch.zeros(9, sw.shape[0], 32)
                swp[:, :, : sw.shape[1]] = sw.permute(2, 3, 0, 1).reshape(9, sw.shape[0], sw.shape[1])
                self.stem_wp = nn.Parameter(swp.to(pk).contiguous(), requires_grad=False)
                self.stem_b32 = nn.Parameter(convs[0][1].float().contiguous(), requires_grad=False)
                if dtype == torch.float32:  # azsp_stem_split: [plane][tap][cout][32 in] f16, input channels 17.. zero
                    sw32 = torch.zeros(sw.shape[0], 32, 3, 3)
                    sw32[:, : sw.shape[1]] = sw
                    self.register_buffer("stem_w32", sw32.float().contiguous(), persistent=False)  # unscaled source of stem_wsp
                    # loud stem weights (|w| > 65504) are carried by an initial activation shift; beyond MAX_ACT_SHIFT: library fp32
                    wmax = float(sw32.abs().max()) if bool(torch.isfinite(sw32).all()) else float("inf")
                    k0 = 0
                    while wmax * 2.0 ** -k0 > F16_MAX and k0 <= self.MAX_ACT_SHIFT:
                        k0 += 1
                    if k0 > self.MAX_ACT_SHIFT:
                        # The split STEM cannot carry these weights (recorded separately: stem_fallback_reason).  The tower is given up WITH
                        # it although its own weights may be ordinary (ADVICE r5 asked to keep it): folded stem weights beyond
                        # 65504 * 2^MAX_ACT_SHIFT = 3.4e7 on 0 / 1 input planes mean stem outputs of that size, which the tower's f16 pairs
                        # cannot carry at any allowed scale -- and the tower-only path has no calibration pass that would find out before
                        # it clamps (test_gpu_network_beyond_the_format_falls_back_to_library_fp32 is exactly this network).
                        self.stem_fallback_reason = f"folded stem weights reach {wmax:.3g}: beyond the f16-pair format"
                        self.split_fallback_reason = self.split_fallback_reason or self.stem_fallback_reason
                        k0 = 0
                    self.stem_wsp = nn.Parameter(split_weights_f16(sw32 * 2.0 ** -k0) if not (self.split_fallback_reason or self.stem_fallback_reason)
                                                 else torch.zeros(2, 9, sw.shape[0], 32, dtype=torch.float16), requires_grad=False)
                    self.stem_b_sp = nn.Parameter(convs[0][1].float().clone().contiguous(), requires_grad=False)
                    self._initial_act_shift = k0
            pw, pb = _fold(net.policy_head[0], net.policy_head[1])
            vw, vb = _fold(net.value_head[0], net.value_head[1])
            self.npol, self.nval = pw.shape[0], vw.shape[0]
            self.head_w32 = nn.Parameter(torch.cat([pw, vw], 0).reshape(pw.shape[0] + vw.shape[0], -1).float().contiguous(), requires_grad=False)
            if dtype == torch.float32:  # azsp_head_split reads head_w32 * 2^act_shift (undoes the activation scale exactly)
                self.head_w_sp = nn.Parameter(self.head_w32.detach().clone(), requires_grad=False)
                if getattr(self, "_initial_act_shift", 0) and not (self.split_fallback_reason or self.stem_fallback_reason):
                    self.set_act_shift(self._initial_act_shift)
            self.head_b32 = nn.Parameter(torch.cat([pb, vb], 0).float().contiguous(), requires_grad=False)
            # both 1x1 heads share one convolution (2 policy planes + 1 value plane)
            self.head_w = nn.Parameter(torch.cat([pw, vw], 0).to(dtype).contiguous(memory_format=self.mf), requires_grad=False)
            self.head_b = nn.Parameter(torch.cat([pb, vb], 0).to(dtype), requires_grad=False)
            self.pol_fc_w = nn.Parameter(net.policy_head[4].weight.to(dtype), requires_grad=False)
            self.pol_fc_b = nn.Parameter(net.policy_head[4].bias.to(dtype), requires_grad=False)
            self.val_fc1_w = nn.Parameter(net.value_head[4].weight.to(dtype), requires_grad=False)
            self.val_fc1_b = nn.Parameter(net.value_head[4].bias.to(dtype), requires_grad=False)
            self.val_fc2_w = nn.Parameter(net.value_head[6].weight.to(dtype), requires_grad=False)
            self.val_fc2_b = nn.Parameter(net.value_head[6].bias.to(dtype), requires_grad=False)
            # zero-padded copies for azsp_fc_heads: weights [ceil32(out)][ceil16(in)] bf16, biases / last layer fp32
            def _pad_w(wt):
                out = torch.zeros((wt.shape[0] + 31) // 32 * 32, (wt.shape[1] + 15) // 16 * 16)
                out[: wt.shape[0], : wt.shape[1]] = wt
                return nn.Parameter(out.to(pk).contiguous(), requires_grad=False)

            def _pad_v(v):
                out = torch.zeros((v.numel() + 31) // 32 * 32)
                out[: v.numel()] = v.reshape(-1)
                return nn.Parameter(out.float().contiguous(), requires_grad=False)

            self.fc_wp, self.fc_bp = _pad_w(net.policy_head[4].weight), _pad_v(net.policy_head[4].bias)
            self.fc_w1, self.fc_b1 = _pad_w(net.value_head[4].weight), _pad_v(net.value_head[4].bias)
            self.fc_w2, self.fc_b2 = _pad_v(net.value_head[6].weight), float(net.value_head[6].bias.item())
            self.num_actions, self.fc_width = net.policy_head[4].weight.shape[0], net.value_head[4].weight.shape[0]
            if dtype == torch.float32:  # azsp_head_split: fp32 Linear weights transposed [inputs][outputs]
                self.pol_fc_wt = nn.Parameter(net.policy_head[4].weight.float().t().contiguous(), requires_grad=False)
                self.val_fc1_wt = nn.Parameter(net.value_head[4].weight.float().t().contiguous(), requires_grad=False)
                self.pol_fc_b32 = nn.Parameter(net.policy_head[4].bias.float().contiguous(), requires_grad=False)
                self.val_fc1_b32 = nn.Parameter(net.value_head[4].bias.float().contiguous(), requires_grad=False)
                self.val_fc2_w32 = nn.Parameter(net.value_head[6].weight.float().reshape(-1).contiguous(), requires_grad=False)
            self.use_fused_fc = True

    def _tiled_tower_ok(self, x):
        """Shapes with a weight-stationary tower kernel (azsp_conv3x3_tiled / azsp_resblock_tiled): 9x9 planes x 128 filters (Go 9x9),
        17x17 planes x 64 filters (the 13x13 Gomoku network after its pad-3 stem), 9x9 planes x 64 filters (the reference's 9x9_12b64
        run) and 19x19 planes x 256 filters (the jumbo Go network)."""
        return (self.binding is not None and self.use_fused_conv and self.use_tiled_tower and x.is_cuda and x.dtype == torch.bfloat16
                and x.shape[2] == x.shape[3] and (x.shape[1], x.shape[2]) in ((128, 9), (64, 17), (64, 9), (256, 19))
                and x.is_contiguous(memory_format=torch.channels_last))

    SPLIT_TOWER_SHAPES = ((128, 9), (64, 9), (64, 17))          # (filters, tower planes) with a weight-stationary azsp_conv3x3_split kernel
    SPLIT_TOWER_ANY_PLANES = (64, 128, 256)                     # filters of the wave-per-tile kernel (csrc/az_conv_spg.h): planes 3 .. 64
    SPLIT_FUSED_SHAPES = ((64, 17), (64, 9))                    # ... with a one-launch-per-block kernel (azsp_resblock_split)
    SPLIT_EVAL_SHAPES = ((128, 9, 1), (64, 9, 1), (64, 13, 3))  # (filters, board, stem pad) whose whole evaluator runs on the split kernels

    @classmethod
    def split_tower_shape(cls, filters, planes):
        """(filters, tower planes) that azsp_conv3x3_split takes: the weight-stationary shapes, and since round 6 every plane size with
        64 / 128 / 256 filters (k_conv3x3_spg: the reference's 19x19 x 256 jumbo tower at its own precision, training_go_jumbo.py:46-47)."""
        return (filters, planes) in cls.SPLIT_TOWER_SHAPES or (filters in cls.SPLIT_TOWER_ANY_PLANES and 3 <= planes <= 64)

    def _split_tower_ok(self, x):
        """fp32 networks with 64 / 128 / 256 filters: the tower runs on azsp_conv3x3_split (include/azsp.h) -- the reference's precision
        class (pipeline.py:91-123 evaluates in fp32) on the f16 matrix cores: the weight-stationary kernels on 9x9 planes with 128 or
        64 filters and on 17x17 planes with 64 filters (the 13x13 Gomoku tower), the wave-per-tile kernel on every other plane size."""
        return (self.binding is not None and self.use_fused_conv and self.use_split_tower and x.is_cuda and x.dtype == torch.float32
                and self.dtype == torch.float32 and x.shape[2] == x.shape[3] and self.split_tower_shape(x.shape[1], x.shape[2])
                and not self.split_fallback_reason and x.is_contiguous(memory_format=torch.channels_last))

    def supports_split_features(self, board_size, device):
        """True when the WHOLE fp32 evaluator runs on the split-precision kernels (azsp_split_features -> azsp_stem_split ->
        azsp_conv3x3_split tower -> azsp_head_split): fp32 networks, 9x9 Go with 128 or 64 filters (pad-1 stem), 13x13 Gomoku with 64
        filters (pad-3 stem, 17x17 planes)."""
        return (self.binding is not None and torch.device(device).type == "cuda" and self.dtype == torch.float32 and self.use_fused_conv
                and self.use_split_tower and self.use_split_heads and self.stem_ok and self.npol + self.nval == 3
                and not self.split_fallback_reason and not self.stem_fallback_reason
                and (self.filters, board_size, self.stem_pad) in self.SPLIT_EVAL_SHAPES)

    def _split_buffers(self, B, S, C, device, slot=0, board_size=None):
        """Scratch of the split-precision evaluator per `slot` (the scheme of _tiled_buffers): three rotating tower buffers, the output
        rows and -- only for callers that hand over fp32 planes (board_size given) -- the stem's feature buffer, sized with the BOARD
        (13 at Gomoku, not the 17 of the tower planes); the engine-facing forward passes the engine's own AZSP_FEAT_F16_SPLIT tensor and
        never allocates it.  A slot holds one batch size at a time; slot 0 is the engine-facing forward (the one SelfPlayActor captures
        in a hipGraph -- its buffers must never be freed by another caller of the same InferenceNet), slot 3 every other call
        (evaluation games, drop-in eval_func wrappers, tests)."""
        dll = self.binding.dll
        nb = dll.azsp_split_bytes(B, S, C) // 2
        cache = self.__dict__.setdefault("_split_cache", {})
        key = (slot, B, S, str(device))
        if key not in cache:
            for k in [k for k in cache if k[0] == slot]:
                del cache[k]
            cache[key] = [[torch.zeros(nb, dtype=torch.float16, device=device) for _ in range(3)], None,
                          torch.empty((B, self.num_actions), dtype=torch.float32, device=device),
                          torch.empty((B,), dtype=torch.float32, device=device)]
        ent = cache[key]
        if board_size is not None and ent[1] is None:
            ent[1] = torch.zeros(dll.azsp_split_bytes(B, board_size, 32) // 2, dtype=torch.float16, device=device)
        return ent

    @torch.no_grad()
    def forward_split(self, planes, priors_out=None, values_out=None, slot=None, split_features=None, probe=None):
        """planes: observation planes [B,17,N,N] fp32, contiguous NCHW (the engine's AZSP_FEAT_F32 features) -- or, with
        split_features = (rows, board_size), the engine's AZSP_FEAT_F16_SPLIT tensor itself (the stem's input layout: no conversion
        launch).  The whole evaluator at the reference's precision class (pipeline.py:91-123 evaluates in fp32) on hand-written kernels.
        slot: scratch buffers to use (see _split_buffers); None = 0 when the outputs go to caller tensors (the engine's forward), 3 otherwise.
        probe: optional callback(buffer, B) after the stem and after every tower convolution (calibrate_activation_scale)."""
        import ctypes

        if not self.act_calibrated and probe is None and planes.is_cuda and not torch.cuda.is_current_stream_capturing():
            self.calibrate_activation_scale(planes, split_features=split_features, slot=slot if slot is not None else (0 if priors_out is not None else 3))
            if self.split_fallback_reason:  # the calibration gave the fp32-class kernels up for this network: library fp32 convolutions
                return self._forward_after_split_fallback(planes, priors_out, values_out, split_features)
        dll, ck = self.binding.dll, self._ck
        st = ctypes.c_void_p(torch.cuda.current_stream(planes.device).cuda_stream) if planes.is_cuda else None  # (host twin: CPU tensors)
        if split_features is not None:
            B, n = split_features
        else:
            B, cin, n, _ = planes.shape
        C = self.filters
        S = n + 2 * (self.stem_pad - 1)  # planes of the tower (network.py:101-105: the Gomoku stem pads by 3)
        if slot is None:
            slot = 0 if priors_out is not None else 3
        (a, m, o), feat, pri_buf, v_buf = self._split_buffers(B, S, C, planes.device, slot, board_size=None if split_features is not None else n)
        self._split = (a, m, o, B)  # marks that the split kernels ran (tests); bench.py replays the tower on slot 0's buffers
        rr = self._range_ptr(planes.device)
        if split_features is not None:
            assert planes.dtype == torch.float16 and planes.numel() >= dll.azsp_split_bytes(B, n, 32) // 2
            feat = planes
        else:
            ck(dll.azsp_split_features(planes.data_ptr(), feat.data_ptr(), B, n, cin, rr, st), "azsp_split_features")
        # engine-written features are 0 / 1 planes (exact f16 values, lo plane never written): the stem skips the lo plane (identical result)
        stem = dll.azsp_stem_split_exact if split_features is not None else dll.azsp_stem_split
        ck(stem(feat.data_ptr(), self.stem_wsp.data_ptr(), self.stem_b_sp.data_ptr(), a.data_ptr(), B, n, C, self.stem_pad, 1, rr, st), "azsp_stem_split")
        if probe is not None:
            probe(a, B)
        a = self._blocks_split(a, m, o, B, S, C, st, rr, probe)
        if probe is not None:
            return None
        pri = priors_out if priors_out is not None else pri_buf
        v = values_out if values_out is not None else v_buf
        ck(dll.azsp_head_split(a.data_ptr(), self.head_w_sp.data_ptr(), self.head_b32.data_ptr(), self.pol_fc_wt.data_ptr(), self.pol_fc_b32.data_ptr(),
                               self.val_fc1_wt.data_ptr(), self.val_fc1_b32.data_ptr(), self.val_fc2_w32.data_ptr(), ctypes.c_float(self.fc_b2),
                               pri.data_ptr(), v.data_ptr(), B, S, C, self.num_actions, self.fc_width, self.npol, st), "azsp_head_split")
        return (pri, v) if priors_out is not None else (pri.clone(), v.clone())  # the cached output buffers are reused by the next call

    def _range_ptr(self, device):
        """Device pointer of this network's range record (None on the host twin's CPU tensors: the twin's default record)."""
        if torch.device(device).type != "cuda":
            return None
        if self.range_rec.device != torch.device(device):
            raise RuntimeError(f"InferenceNet lives on {self.range_rec.device}, its input on {device}")
        return self.range_rec.data_ptr()

    def split_range_status(self, reset=False, stream=None):
        """(events, max_abs) of THIS network's sticky range record (include/azsp.h azsp_split_range_read): how many kernel lanes met a
        value beyond f16's finite range (clamped to +-65504 where the reference's fp32 network would carry it) since the last reset,
        and the largest such |v| in the kernels' own (scaled) units: multiply by 2^act_shift for the network's units.  Synchronises
        the stream.  Another InferenceNet in the same process has its own record."""
        import ctypes

        ev, mx = ctypes.c_uint32(0), ctypes.c_float(0.0)
        rec = self.range_rec.data_ptr() if self.range_rec.is_cuda else None
        if rec is not None and stream is None:
            stream = ctypes.c_void_p(torch.cuda.current_stream(self.range_rec.device).cuda_stream)
        with torch.cuda.device(self.range_rec.device) if rec is not None else _nullcontext():
            self._ck(self.binding.dll.azsp_split_range_read(rec, ctypes.byref(ev), ctypes.byref(mx), int(bool(reset)), stream), "azsp_split_range_read")
        return int(ev.value), float(mx.value)

    def poll_range(self, planes=None):
        """For callers that drive the InferenceNet themselves (evaluation games, drop-in eval_func wrappers: no SelfPlayActor polls for
        them): reads and resets this network's range record; on an event it warns, raises the activation scale and -- given the batch
        that was just evaluated -- re-calibrates on it.  Returns the number of events.  Synchronises the stream."""
        if self.dtype != torch.float32 or not hasattr(self, "range_rec") or not self.range_rec.is_cuda or self.split_fallback_reason:
            return 0
        ev, mx = self.split_range_status(reset=True)
        if ev:
            import warnings

            old = self.act_shift
            if planes is not None and self.supports_split_features(planes.shape[2], planes.device):
                self.set_act_shift(min(self.MAX_ACT_SHIFT, old + 2))
                self.calibrate_activation_scale(planes.float().contiguous())
            else:
                # the split tower behind a library stem / heads (or no batch given): no layer-by-layer calibration pass -- raise the scale
                # by what the record shows (a lower bound: clamped values hide the true maximum) + 16x, as SelfPlayActor does
                import math

                k = min(self.MAX_ACT_SHIFT, old + max(2, math.ceil(math.log2(max(mx, F16_MAX) / F16_MAX)) + 4))
                if k > old:
                    self.set_act_shift(k)
            what = (f"library fp32 convolutions from now on ({self.split_fallback_reason})" if self.split_fallback_reason
                    else f"activation scale 2^-{old} -> 2^-{self.act_shift}" if self.act_shift > old
                    else f"the activation scale is at its limit (2^-{old}): evaluate this network with use_split_tower = False")
            warnings.warn(f"alpha_zero_amd: the fp32-class evaluator clamped {ev} activation lanes beyond f16's range (largest |v| = "
                          f"{mx * 2.0 ** old:.6g}); {what}", RuntimeWarning, stacklevel=2)
        return ev

    # -- range safety: exact power-of-two activation scale -------------------------------------------------------------------
    MAX_ACT_SHIFT = 9   # beyond 2^-9 the scaled stem weights lose fp32-class accuracy (their hi halves become f16 subnormals)
    ACT_HEADROOM = 16.0  # calibration leaves this factor between the largest activation it saw and f16's limit

    @torch.no_grad()
    def set_act_shift(self, k):
        """Carry every activation of the fp32-class path as v * 2^-k: the stem's weights and bias and every tower bias are multiplied
        by 2^-k, the 1x1 head weights by 2^k.  ReLU and the skip addition are positively homogeneous, a power of two multiplies fp32
        and f16 values exactly: the evaluator computes the same function (network.py:42-82, :118-156) while its activations stay
        2^k further inside f16's range.  All updates are in place (a captured hipGraph keeps pointing at the right tensors)."""
        if self.dtype != torch.float32 or not hasattr(self, "b_sp"):
            raise RuntimeError("set_act_shift: fp32 networks only")
        k = int(k)
        if k < 0 or k > 24:
            raise ValueError(f"act_shift {k} out of range")
        dn, up = 2.0 ** -k, 2.0 ** k
        if self.stem_ok:
            if self.split_fallback_reason or self.stem_fallback_reason or float(self.stem_w32.abs().max()) * dn > F16_MAX:
                # the split stem is out of use (library fallback), or the scaled stem weights do not fit the f16 pairs (a stem that
                # needed an initial shift, being reset to 0 on the way to the library): nothing may read stem_wsp -- zero it, never raise
                self.stem_wsp.zero_()
            else:
                self.stem_wsp.copy_(split_weights_f16(self.stem_w32 * dn))
            self.stem_b_sp.copy_(self.stem_b32 * dn)
        for d, b in zip(self.b_sp, self.b32):
            d.copy_(b * dn)
        self.head_w_sp.copy_(self.head_w32 * up)
        self.act_shift = k

    @torch.no_grad()
    def calibrate_activation_scale(self, planes, split_features=None, slot=3):
        """One calibration pass of the fp32-class evaluator on a real batch: runs the stem and the tower launch by launch, takes the
        largest |activation| of every layer's output and raises act_shift until that maximum sits ACT_HEADROOM below f16's limit
        (never lowers it).  If that needs more than MAX_ACT_SHIFT the fp32-class kernels are given up for this network
        (split_fallback_reason; the caller's forward then runs the library's fp32 convolutions and `evaluator_path` says so).
        Returns (act_shift, largest |activation| in the network's own units).  Leaves the range record clean.  Not capturable."""
        import math

        if planes.is_cuda and torch.cuda.is_current_stream_capturing():
            raise RuntimeError("calibrate_activation_scale cannot run inside a hipGraph capture")
        self.act_calibrated = True  # (set first: forward_split below must not recurse)
        shift, worst = self.act_shift, 0.0
        for _ in range(8):
            if shift != self.act_shift:
                self.set_act_shift(shift)
            peak = torch.zeros((), dtype=torch.float32, device=planes.device)

            def probe(buf, B, peak=peak):
                peak.copy_(torch.maximum(peak, buf.view(B, 2, -1)[:, 0].abs().max().float()))

            self.forward_split(planes, slot=slot, split_features=split_features, probe=probe)
            mx = float(peak.item())
            if not math.isfinite(mx):
                self.split_fallback_reason = "non-finite activations in the calibration pass"
                break
            if mx >= F16_MAX:  # saturated somewhere: the true maximum is unknown -- take a big step and look again
                worst = max(worst, mx * 2.0 ** shift)
                shift += 6
            else:
                worst = mx * 2.0 ** shift
                need = math.ceil(math.log2(mx * self.ACT_HEADROOM / F16_MAX)) if mx > 0.0 else 0
                if need <= 0:
                    break
                shift += need
            if shift > self.MAX_ACT_SHIFT:
                self.split_fallback_reason = (f"activations reach {worst:.3g}: beyond what the f16-pair format carries even scaled by "
                                              f"2^-{self.MAX_ACT_SHIFT}")
                break
        if self.split_fallback_reason and self.act_shift != 0:
            self.set_act_shift(0)
        self.act_max_abs = max(self.act_max_abs, worst)
        if self.range_rec.is_cuda:
            self.split_range_status(reset=True)  # the passes above may have clamped: that is what they were looking for
        return self.act_shift, worst

    def _forward_after_split_fallback(self, planes, priors_out, values_out, split_features):
        """The forward of a network whose fp32-class kernels were given up (split_fallback_reason): library fp32 convolutions.  The
        engine's AZSP_FEAT_F16_SPLIT tensor is unpacked to NCHW fp32 planes first (its hi plane holds the 0 / 1 observation planes)."""
        if split_features is not None:
            B, n = split_features
            cin = self.w[0].shape[1]
            planes = planes[: B * 2 * 32 * n * n].view(B, 2, 4, n * n, 8)[:, 0].permute(0, 1, 3, 2).reshape(B, 32, n, n)[:, :cin].float()
        return self.forward(planes, priors_out, values_out)

    def _blocks_split(self, a, m, o, B, S, C, st, rr=None, probe=None):
        """All residual blocks on split-layout buffers; returns the buffer holding the tower output."""
        dll, ck = self.binding.dll, self._ck
        if self.use_fused_block and (C, S) in self.SPLIT_FUSED_SHAPES and probe is None:
            # 64 filters on 17x17 planes (13x13 Gomoku) or 9x9 planes (9x9 Go, two boards per tile): one launch per ResNetBlock, the intermediate
            # activation stays in LDS (azsp_resblock_split: two tensor passes through HBM per block instead of five; bit-identical to the
            # two launches below)
            for i in range(self.n_blocks):
                ck(dll.azsp_resblock_split(a.data_ptr(), self.wsp[2 * i].data_ptr(), self.b_sp[2 * i].data_ptr(), self.wsp[2 * i + 1].data_ptr(),
                                           self.b_sp[2 * i + 1].data_ptr(), o.data_ptr(), B, S, C, rr, st), "azsp_resblock_split")
                a, o = o, a
            return a
        for i in range(self.n_blocks):
            ck(dll.azsp_conv3x3_split(a.data_ptr(), self.wsp[2 * i].data_ptr(), self.b_sp[2 * i].data_ptr(), None, m.data_ptr(), B, S, C, 1, rr, st),
               "azsp_conv3x3_split")
            if probe is not None:
                probe(m, B)
            ck(dll.azsp_conv3x3_split(m.data_ptr(), self.wsp[2 * i + 1].data_ptr(), self.b_sp[2 * i + 1].data_ptr(), a.data_ptr(), o.data_ptr(),
                                      B, S, C, 1, rr, st), "azsp_conv3x3_split")
            if probe is not None:
                probe(o, B)
            a, o = o, a
        return a

    def _conv(self, x, i, res=None):
        """relu(conv3x3(x) + bias [+ res]) of tower convolution i (0-based) for shapes WITHOUT a hand-written tower kernel: the library
        convolution followed by the fused epilogue kernel (see `evaluator_path`)."""
        return self._epilogue(F.conv2d(x, self.w[1 + i], None, padding=1), self.b[1 + i], res)

    def _epilogue(self, y, bias, res=None):
        """relu(y + bias [+ res]) in place on a channels-last activation."""
        if self.binding is not None and y.is_cuda and y.is_contiguous(memory_format=torch.channels_last) and y.shape[1] % 8 == 0:
            import ctypes

            B, C, H, W = y.shape
            dt = {torch.float32: 1, torch.bfloat16: 2, torch.float16: 3}[y.dtype]
            rc = self.binding.dll.azsp_bias_act(y.data_ptr(), bias.data_ptr(), res.data_ptr() if res is not None else None, B * H * W, C,
                                                dt, 1, ctypes.c_void_p(torch.cuda.current_stream(y.device).cuda_stream))
            if rc != 0:
                raise RuntimeError(f"azsp_bias_act failed with code {rc}")
            return y
        y.add_(bias.view(1, -1, 1, 1))
        if res is not None:
            y.add_(res)
        return F.relu_(y)

    def _tiled_buffers(self, B, S, C, device, slot=0):
        """Three rotating tower buffers (block input, middle, block output) per `slot`: callers whose forwards must not share scratch
        memory use different slots (slot 0: the actor's graph-captured forward; slot 3: forward_planes, i.e. evaluation games / drop-in
        eval_func calls on the same InferenceNet; slots 1-2: the half-batch experiment of tools/overlap_actor.py)."""
        n = self.binding.dll.azsp_tiled_bytes(B, S, C) // 2
        cache = self.__dict__.setdefault("_tiled_cache", {})
        key = (slot, n, str(device))
        if key not in cache:
            for k in [k for k in cache if k[0] == slot]:  # a slot holds one size at a time
                del cache[k]
            cache[key] = [torch.zeros(n, dtype=self.pack_dtype, device=device) for _ in range(3)]
        if slot == 0:
            self._tiled = cache[key]  # (bench.py replays the tower on the activations of the last full-batch forward)
        return cache[key]

    def _head_buffers(self, B, k1, k2, device, slot=0):
        cache = self.__dict__.setdefault("_head_cache", {})
        key = (slot, B, str(device))
        if key not in cache:
            for k in [k for k in cache if k[0] == slot]:
                del cache[k]
            cache[key] = (torch.zeros((B + 1, k1), dtype=self.pack_dtype, device=device), torch.zeros((B + 1, k2), dtype=self.pack_dtype, device=device),
                          torch.empty((B, self.num_actions), dtype=torch.float32, device=device), torch.empty((B,), dtype=torch.float32, device=device))
        return cache[key]

    @staticmethod
    def _ck(rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed with code {rc}")

    def _blocks_tiled(self, a, m, o, B, S, C, st):
        """All residual blocks on tiled buffers; returns the buffer holding the tower output."""
        dll, ck = self.binding.dll, self._ck
        conv = dll.azsp_conv3x3_tiled_f16 if self.pack_dtype == torch.float16 else dll.azsp_conv3x3_tiled
        if self.use_fused_block and (C, S) in ((64, 17), (64, 9)) and self.pack_dtype != torch.float16:
            # 64 filters: both filter banks of a block fit in a CU's registers -> one launch per ResNetBlock, intermediate in LDS
            for i in range(self.n_blocks):
                ck(dll.azsp_resblock_tiled(a.data_ptr(), self.wp[2 * i].data_ptr(), self.b32[2 * i].data_ptr(), self.wp[2 * i + 1].data_ptr(),
                                           self.b32[2 * i + 1].data_ptr(), o.data_ptr(), B, S, C, st), "azsp_resblock_tiled")
                a, o = o, a
            return a
        for i in range(self.n_blocks):
            ck(conv(a.data_ptr(), self.wp[2 * i].data_ptr(), self.b32[2 * i].data_ptr(), None, m.data_ptr(), B, S, C, 1, st), "azsp_conv3x3_tiled")
            ck(conv(m.data_ptr(), self.wp[2 * i + 1].data_ptr(), self.b32[2 * i + 1].data_ptr(), a.data_ptr(), o.data_ptr(), B, S, C, 1, st),
               "azsp_conv3x3_tiled")
            a, o = o, a
        return a

    def evaluator_path(self, board_size, device):
        """Which kernels the forward pass of this network runs on `device` -- reported by bench.py (`config.evaluator`) and logged
        once by SelfPlayActor, so that an unsupported shape never degrades silently to the library path."""
        if self.supports_tiled_features(board_size, device):
            return "hand-written: tiled stem / tower / head / FC kernels (libazsp)" + (", f16 activations and weights" if self.dtype == torch.float16 else "")
        if torch.device(device).type == "cuda" and self.dtype == torch.bfloat16 and self.binding is not None:
            s = board_size + 2 * (self.stem_pad - 1)
            if (self.filters, s) in ((128, 9), (64, 17), (64, 9), (256, 19)):
                return "hand-written tower (azsp_conv3x3_tiled) behind a library stem and heads"
        if self.supports_split_features(board_size, device):
            fused = self.use_fused_block and (self.filters, board_size + 2 * (self.stem_pad - 1)) in self.SPLIT_FUSED_SHAPES
            tower = ("azsp_resblock_split: one launch per ResNetBlock, intermediate activation in LDS; " if fused else "azsp_conv3x3_split: ")
            return (f"fp32 class, hand-written: split-precision stem / tower ({tower}hi + lo f16 pairs, three MFMA products, "
                    "fp32 accumulation) / fp32 heads (libazsp)")
        if torch.device(device).type == "cuda" and self.dtype == torch.float32 and self.binding is not None and self.use_split_tower and not self.split_fallback_reason:
            if self.split_tower_shape(self.filters, board_size + 2 * (self.stem_pad - 1)):
                kern = "" if (self.filters, board_size + 2 * (self.stem_pad - 1)) in self.SPLIT_TOWER_SHAPES else ", wave-per-tile kernel k_conv3x3_spg"
                return (f"fp32 class: hand-written split-precision tower (azsp_conv3x3_split{kern}: hi + lo f16 pairs, three MFMA products, fp32 "
                        "accumulation) behind a library fp32 stem and heads" + (f" ({self.stem_fallback_reason})" if self.stem_fallback_reason else ""))
        if self.split_fallback_reason or self.stem_fallback_reason:
            return ("library fp32 convolutions + azsp_bias_act epilogue (fp32-class kernels given up for this network: "
                    f"{self.split_fallback_reason or self.stem_fallback_reason})")
        return f"library convolutions + azsp_bias_act epilogue (no hand-written kernel for {self.filters} filters on {board_size}x{board_size}, {self.dtype})"

    def supports_tiled_features(self, board_size, device):
        """True when the whole evaluator can run on the tiled layout (azsp_stem_tiled -> tower -> azsp_head_tiled): 9x9 Go with 128 or
        64 filters (pad-1 stem), 13x13 Gomoku with 64 filters (pad-3 stem, 17x17 planes) and 19x19 Go with 256 filters."""
        shape_ok = (self.filters, board_size, self.stem_pad) in ((128, 9, 1), (64, 9, 1), (64, 13, 3), (256, 19, 1))
        if self.dtype == torch.float16:  # the f16 variants exist for the 9x9 x 128 evaluator (82 actions, 128 fully connected units)
            shape_ok = (self.filters, board_size, self.stem_pad, self.num_actions, self.fc_width) == (128, 9, 1, 82, 128)
        return (self.binding is not None and torch.device(device).type == "cuda" and self.dtype in (torch.bfloat16, torch.float16) and shape_ok
                and self.stem_ok and self.npol + self.nval == 3 and self.use_fused_conv and self.use_tiled_tower)

    @torch.no_grad()
    def forward_tiled(self, feat, rows, board_size, priors_out=None, values_out=None, slot=0):
        """feat: the engine's AZSP_FEAT_BF16_TILED feature tensor for `rows` leaf positions (or a tile-aligned slice of it).  Stem,
        tower, the 1x1 head convolutions and the fully connected layers run on the tiled layout in hand-written kernels.  `slot`
        selects the scratch buffers (see _tiled_buffers)."""
        import ctypes

        dll, ck = self.binding.dll, self._ck
        f16 = self.pack_dtype == torch.float16  # azsp_*_f16: the same kernels on f16 elements (feat is then the engine's AZSP_FEAT_F16_TILED tensor)
        stem, head, fc = ((dll.azsp_stem_tiled_f16, dll.azsp_head_tiled_f16, dll.azsp_fc_heads_f16) if f16 else
                          (dll.azsp_stem_tiled, dll.azsp_head_tiled, dll.azsp_fc_heads))
        st = ctypes.c_void_p(torch.cuda.current_stream(feat.device).cuda_stream) if feat.is_cuda else None  # (host twin: CPU tensors)
        B, C = rows, self.filters
        S = board_size + 2 * (self.stem_pad - 1)  # planes of the tower (network.py:101-105: the Gomoku stem pads by 3)
        a, m, o = self._tiled_buffers(B, S, C, feat.device, slot)
        ck(stem(feat.data_ptr(), self.stem_wp.data_ptr(), self.stem_b32.data_ptr(), a.data_ptr(), B, board_size, C, self.stem_pad, 1, st), "azsp_stem_tiled")
        a = self._blocks_tiled(a, m, o, B, S, C, st)
        k1, k2 = self.fc_wp.shape[1], self.fc_w1.shape[1]  # head-plane rows padded to the k-steps of azsp_fc_heads (zero padding)
        pol, val, pri_buf, v_buf = self._head_buffers(B, k1, k2, feat.device, slot)
        ck(head(a.data_ptr(), self.head_w32.data_ptr(), self.head_b32.data_ptr(), pol.data_ptr(), val.data_ptr(), B, S, C, self.npol, self.nval, k1, k2, st),
           "azsp_head_tiled")
        nt = ((self.num_actions + 31) // 32, (self.fc_width + 31) // 32)
        fused_fc = self.use_fused_fc and nt in ((3, 2), (3, 4), (6, 2), (6, 4), (12, 8)) and (not f16 or nt == (3, 4))
        if not fused_fc:
            return self._fc_heads(pol[:B, : self.npol * S * S], val[:B, : self.nval * S * S], priors_out, values_out)
        pri = priors_out if priors_out is not None else pri_buf
        v = values_out if values_out is not None else v_buf
        ck(fc(pol.data_ptr(), val.data_ptr(), self.fc_wp.data_ptr(), self.fc_bp.data_ptr(), k1 // 16, self.fc_w1.data_ptr(), self.fc_b1.data_ptr(),
              k2 // 16, self.fc_w2.data_ptr(), ctypes.c_float(self.fc_b2), pri.data_ptr(), v.data_ptr(), B, self.num_actions, self.fc_width, st),
           "azsp_fc_heads")
        return pri, v

    @torch.no_grad()
    def forward_planes(self, x, priors_out=None, values_out=None):
        """x: observation planes [B,17,N,N] (any dtype, on the evaluator's device).  Runs the whole evaluator on the hand-written kernels
        when this network / board has them (the planes are padded to 32 channels and converted to the tiled feature layout on the
        device: azsp_tile_layout), otherwise `forward`.  Used where leaf rows arrive as NCHW planes: evaluation games
        (core/evaluate.py DeviceEvaluator), drop-in eval_func wrappers."""
        B, _, n, _ = x.shape
        if not (x.is_cuda and self.supports_tiled_features(n, x.device)):
            return self.forward(x, priors_out, values_out)
        import ctypes

        xb = torch.zeros((B, 32, n, n), dtype=self.pack_dtype, device=x.device).contiguous(memory_format=torch.channels_last)
        xb[:, : x.shape[1]] = x.to(self.pack_dtype)
        feat = torch.zeros(self.binding.dll.azsp_tiled_bytes(B, n, 32) // 2, dtype=self.pack_dtype, device=x.device)
        st = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        self._ck(self.binding.dll.azsp_tile_layout(xb.data_ptr(), feat.data_ptr(), B, n, 32, 1, st), "azsp_tile_layout")
        pri, v = self.forward_tiled(feat, B, n, priors_out, values_out, slot=3)
        return (pri, v) if priors_out is not None else (pri.clone(), v.clone())  # the slot's output buffers are reused by the next call

    def _fc_heads(self, pol, val, priors_out, values_out):
        """Fully connected layers of both heads (core/network.py:136-156) on the flattened head planes."""
        logits = F.linear(pol, self.pol_fc_w, self.pol_fc_b)
        v = torch.tanh(F.linear(F.relu_(F.linear(val, self.val_fc1_w, self.val_fc1_b)), self.val_fc2_w, self.val_fc2_b))
        pri = torch.softmax(logits.float(), dim=-1)
        v = v.float().squeeze(1)
        if priors_out is not None:
            priors_out.copy_(pri)
            values_out.copy_(v)
            return priors_out, values_out
        return pri, v

    def _tower_tiled(self, x):
        """The whole residual tower on the tiled activation layout (include/azsp.h: azsp_tile_layout /
        azsp_conv3x3_tiled): the weight-stationary MFMA kernel, activations converted once on entry and once on exit."""
        import ctypes

        dll, ck = self.binding.dll, self._ck
        st = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        B, C, S = x.shape[0], x.shape[1], x.shape[2]
        a, m, o = self._tiled_buffers(B, S, C, x.device)
        ck(dll.azsp_tile_layout(x.data_ptr(), a.data_ptr(), B, S, C, 1, st), "azsp_tile_layout")
        a = self._blocks_tiled(a, m, o, B, S, C, st)
        ck(dll.azsp_tile_layout(a.data_ptr(), x.data_ptr(), B, S, C, 0, st), "azsp_tile_layout")
        return x

    def _tower_split(self, x, slot=3):
        """The whole residual tower of an fp32 network on the split layout (azsp_split_layout / azsp_conv3x3_split): activations are
        converted once on entry and once on exit; x is channels-last fp32 [B,C,S,S] and is overwritten with the tower's output.
        (The evaluator of shapes whose stem or heads have no split kernel: a library stem and heads around the hand-written tower.)"""
        import ctypes

        dll, ck = self.binding.dll, self._ck
        st = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        rr = self._range_ptr(x.device)
        B, C, S = x.shape[0], x.shape[1], x.shape[2]
        (a, m, o), _, _, _ = self._split_buffers(B, S, C, x.device, slot)
        self._split = (a, m, o, B)
        if self.act_shift:
            x.mul_(2.0 ** -self.act_shift)  # exact; the tower biases b_sp carry the same factor
        ck(dll.azsp_split_layout(x.data_ptr(), a.data_ptr(), B, S, C, 1, rr, st), "azsp_split_layout")
        a = self._blocks_split(a, m, o, B, S, C, st, rr)
        ck(dll.azsp_split_layout(a.data_ptr(), x.data_ptr(), B, S, C, 0, rr, st), "azsp_split_layout")
        if self.act_shift:
            x.mul_(2.0 ** self.act_shift)
        return x

    @torch.no_grad()
    def forward(self, x, priors_out=None, values_out=None):
        """x: [B,17,N,N] any dtype -> (priors fp32 [B,A], values fp32 [B])."""
        if x.is_cuda and self.supports_split_features(x.shape[2], x.device):
            return self.forward_split(x.float().contiguous(), priors_out, values_out)
        x = x.to(self.dtype).contiguous(memory_format=self.mf)
        x = self._epilogue(F.conv2d(x, self.w[0], None, padding=self.stem_pad), self.b[0])
        if self._tiled_tower_ok(x):
            x = self._tower_tiled(x)
        elif self._split_tower_ok(x):
            x = self._tower_split(x, slot=0 if priors_out is not None else 3)
        else:
            for i in range(self.n_blocks):
                y = self._conv(x, 2 * i)
                x = self._conv(y, 2 * i + 1, x)
        h = F.relu_(F.conv2d(x, self.head_w, self.head_b))
        B = h.shape[0]
        pol = h[:, : self.npol].contiguous(memory_format=torch.contiguous_format).reshape(B, -1)  # NCHW flatten order (nn.Flatten)
        val = h[:, self.npol :].contiguous(memory_format=torch.contiguous_format).reshape(B, -1)
        return self._fc_heads(pol, val, priors_out, values_out)
