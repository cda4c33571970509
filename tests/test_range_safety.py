"""Range safety of the fp32-class (split-precision) evaluator.

The reference evaluates its network in fp32 with no range limit (alpha_zero/core/pipeline.py:91-123, core/network.py:42-82); the
split-precision kernels carry every activation as an f16 pair and clamp |v| > 65504.  This file pins what round 5 built around that:
  * the range record belongs to ONE network (`range_rec_dev` of the azsp_*_split entries): two evaluators in one process never see
    each other's events;
  * an exact power-of-two activation scale (InferenceNet.set_act_shift / calibrate_activation_scale) keeps a network whose
    activations reach 1e6 inside the format: priors / values within the whole-network bound of the fp64 module, zero range events;
  * a network the format cannot carry falls back to the library's fp32 convolutions and says so;
  * azsp_conv3x3_split refuses residual == y at 17x17 and computes it correctly at 9x9.
"""
import ctypes
import os
import warnings

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from alpha_zero_amd.core.network import AlphaZeroNet, InferenceNet, split_weights_f16

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _loud_net(board, filters, blocks, gain, gomoku=False, seed=0):
    """A random-init network whose tower activations grow to `gain` x the usual size: the stem's BatchNorm scale is multiplied by
    `gain` (everything behind it is positively homogeneous up to the biases), the 1x1 head convolutions are divided by it so that
    the heads stay in their usual range."""
    torch.manual_seed(seed)
    A = board * board + (0 if gomoku else 1)
    net = AlphaZeroNet((17, board, board), A, blocks, filters, 64, gomoku=gomoku).eval()
    with torch.no_grad():
        net.conv_block[1].weight.mul_(gain)
        net.conv_block[1].bias.add_(0.05 * gain)
        net.policy_head[0].weight.div_(gain)
        net.value_head[0].weight.div_(gain)
        net.policy_head[4].weight.mul_(0.2)
    return net


def _ref64(net, x):
    net64 = AlphaZeroNet((17, x.shape[2], x.shape[2]), net.policy_head[4].out_features, len(net.res_blocks), net.conv_block[0].out_channels,
                         net.value_head[4].out_features, gomoku=net.conv_block[0].padding[0] == 3).double().eval()
    net64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in net.state_dict().items()})
    with torch.no_grad():
        lg, v = net64(x.double())
    return torch.softmax(lg, -1), v.squeeze(1)


def _tower_peak(net, x):
    """Largest |activation| anywhere in the tower of the fp32 module (what the f16-pair format has to carry)."""
    peak = 0.0
    with torch.no_grad():
        f = net.conv_block(x)
        peak = max(peak, float(f.abs().max()))
        for blk in net.res_blocks:
            mid = blk.conv_block1(f)
            peak = max(peak, float(mid.abs().max()))
            f = blk(f)
            peak = max(peak, float(f.abs().max()))
    return peak


# ---- host twin (CPU tier) ---------------------------------------------------------------------------------------------------------
def test_range_record_belongs_to_the_caller_host_twin():
    """range_rec_dev: events land in the record the call names; NULL = the library's default record (azsp_split_range_status)."""
    import engine_util as eu

    bnd = eu.hosttwin_binding()
    ev, mx = ctypes.c_uint32(0), ctypes.c_float(0.0)
    assert bnd.dll.azsp_split_range_status(None, None, 1, None) == 0
    x = torch.tensor([1.0, -1e5, 65504.0, 1e-7, 0.0, 7e4, -2.5e-5, 60000.0]).reshape(1, 8, 1, 1).contiguous(memory_format=torch.channels_last)
    s = torch.zeros(2 * 8, dtype=torch.float16)
    rec_a, rec_b = np.zeros(2, dtype=np.uint32), np.zeros(2, dtype=np.uint32)
    assert bnd.dll.azsp_split_layout(x.data_ptr(), s.data_ptr(), 1, 1, 8, 1, rec_a.ctypes.data, None) == 0
    assert rec_a[0] == 2 and rec_a[1:].view(np.float32)[0] == 1e5 and rec_b[0] == 0
    assert bnd.dll.azsp_split_range_status(ctypes.byref(ev), ctypes.byref(mx), 0, None) == 0 and ev.value == 0  # the default record saw nothing
    assert bnd.dll.azsp_split_range_read(rec_a.ctypes.data, ctypes.byref(ev), ctypes.byref(mx), 1, None) == 0 and ev.value == 2 and mx.value == 1e5
    assert rec_a[0] == 0 and rec_a[1] == 0  # reset
    x[0, 3, 0, 0] = float("nan")  # a NaN among the fp32 inputs is an event too (recorded as +inf)
    assert bnd.dll.azsp_split_layout(x.data_ptr(), s.data_ptr(), 1, 1, 8, 1, rec_b.ctypes.data, None) == 0
    assert rec_b[0] == 3 and np.isinf(rec_b[1:].view(np.float32)[0])


def test_activation_shift_is_function_preserving_host_twin():
    """set_act_shift(k) on the host twin's split arithmetic: the evaluator computes the same priors / values (a power of two scales
    fp32 and normal f16 values exactly; ReLU and the skip are positively homogeneous; only halves that fall into f16's subnormal
    range round differently: fp32 round-off class, bound 1e-6) while every tower activation is 2^k smaller."""
    import engine_util as eu

    torch.manual_seed(3)
    net = AlphaZeroNet((17, 9, 9), 82, 2, 64, 64).eval()
    with torch.no_grad():
        net.policy_head[4].weight.mul_(0.2)
    inf = InferenceNet(net, dtype=torch.float32, binding=eu.hosttwin_binding())
    x = (torch.rand(3, 17, 9, 9) > 0.6).float()
    p0, v0 = inf.forward_split(x)
    peaks = []
    for k in (0, 3, 7):
        inf.set_act_shift(k)
        pk = [0.0]
        inf.forward_split(x, probe=lambda buf, B, pk=pk: pk.__setitem__(0, max(pk[0], float(buf.view(B, 2, -1)[:, 0].float().abs().max()))))
        peaks.append(pk[0])
        p, v = inf.forward_split(x)
        assert (p - p0).abs().max().item() <= 1e-6 and (v - v0).abs().max().item() <= 1e-6, k
    assert abs(peaks[1] * 8 / peaks[0] - 1) < 1e-3 and abs(peaks[2] * 128 / peaks[0] - 1) < 1e-3, peaks
    with pytest.raises(ValueError):
        inf.set_act_shift(-1)


# ---- GPU tier ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("board,filters,gomoku", [(9, 128, False), (13, 64, True)])
def test_gpu_network_with_activations_of_1e6_runs_in_range_after_calibration(board, filters, gomoku):
    """a network whose tower activations reach ~1e6 (15x beyond f16's largest number) -- the 10-block x 128 Go network of
    the headline shape with its block-10 output at 1e6, and the 6 x 64 Gomoku network of C2.  Uncalibrated the kernels clamp (and record
    it); the calibration pass picks an exact power-of-two scale and the evaluator then matches the fp64 module within the whole-network
    bound (2e-4) with range_events == 0."""
    from alpha_zero_amd import _lib

    net = _loud_net(board, filters, 6 if gomoku else 10, gain=3e4 if gomoku else 6.7e3, gomoku=gomoku)
    g = torch.Generator().manual_seed(11)
    x = (torch.rand(24, 17, board, board, generator=g) > 0.6).float()
    peak = _tower_peak(net, x)
    assert 5e5 < peak < 1.9e6, peak  # the premise: ~1e6, far outside f16 (65504), inside what MAX_ACT_SHIFT carries with its headroom
    p64, v64 = _ref64(net, x)
    inf = InferenceNet(net, dtype=torch.float32, binding=_lib.load()).cuda()
    # (1) uncalibrated, shift 0: clamped and recorded in this network's record
    inf.act_calibrated = True
    pc, vc = inf(x.cuda())
    ev, mx = inf.split_range_status(reset=True)
    assert ev > 0 and mx > 65504.0
    # (2) calibrated: exact rescaling, no event, fp32-class accuracy
    inf.act_calibrated = False
    p, v = inf(x.cuda())
    assert inf.act_calibrated and 4 <= inf.act_shift <= inf.MAX_ACT_SHIFT and not inf.split_fallback_reason, (inf.act_shift, inf.split_fallback_reason)
    assert abs(inf.act_max_abs / peak - 1) < 1e-3, (inf.act_max_abs, peak)
    ev, mx = inf.split_range_status(reset=True)
    dp, dv = (p.cpu().double() - p64).abs().max().item(), (v.cpu().double() - v64).abs().max().item()
    assert ev == 0 and dp <= 2e-4 and dv <= 2e-4, (ev, dp, dv)
    assert getattr(inf, "_split", None) is not None and "split" in inf.evaluator_path(board, "cuda")


@pytest.mark.gpu
def test_gpu_network_beyond_the_format_falls_back_to_library_fp32():
    """Activations of ~1e9 cannot be carried even scaled by 2^-MAX_ACT_SHIFT: the calibration gives the fp32-class kernels up for this
    network, the forward runs the library's fp32 convolutions (still within the bound of the fp64 module) and evaluator_path says so."""
    from alpha_zero_amd import _lib

    net = _loud_net(9, 64, 4, gain=3e8)
    x = (torch.rand(8, 17, 9, 9, generator=torch.Generator().manual_seed(5)) > 0.6).float()
    p64, v64 = _ref64(net, x)
    inf = InferenceNet(net, dtype=torch.float32, binding=_lib.load()).cuda()
    p, v = inf(x.cuda())
    assert inf.split_fallback_reason and inf.act_shift == 0
    assert "library fp32" in inf.evaluator_path(9, "cuda") and not inf.supports_split_features(9, "cuda")
    assert inf.split_range_status(reset=True)[0] == 0
    assert (p.cpu().double() - p64).abs().max().item() <= 2e-4 and (v.cpu().double() - v64).abs().max().item() <= 2e-4


@pytest.mark.gpu
def test_gpu_two_actors_in_one_process_each_see_only_their_own_range_events():
    """VERDICT r4 #1b: the record is per network.  Actor A plays a network that leaves f16's range once its calibration is defeated,
    actor B a tame one, both in this process on the same device: A's harvest reports (and repairs) A's events, B never sees any."""
    from alpha_zero_amd.core.pipeline import SelfPlayActor

    loud, tame = _loud_net(9, 64, 2, gain=2e5, seed=1), _loud_net(9, 64, 2, gain=1.0, seed=2)
    kw = dict(game="go", board_size=9, num_games=64, num_simulations=16, num_parallel=4, seed=3, use_graph=False)
    a, b = SelfPlayActor(loud, **kw), SelfPlayActor(tame, **kw)
    a.infer.act_calibrated = True  # defeat A's calibration: it plays at shift 0 and clamps
    for _ in range(6):
        a.run_round()
        b.run_round()
    torch.cuda.synchronize()
    with warnings.catch_warnings(record=True) as wb:
        warnings.simplefilter("always")
        b.harvest()
    assert b.range_events == 0 and not [w for w in wb if "clamped" in str(w.message)]
    assert b.infer.split_range_status()[0] == 0
    with warnings.catch_warnings(record=True) as wa:
        warnings.simplefilter("always")
        a.harvest()
    assert a.range_events > 0 and a.range_rescales == 1 and [w for w in wa if "clamped" in str(w.message)]
    assert a.infer.act_shift >= 2 and a.range_max_abs > 65504.0 and "activations carried" in a.evaluator_path
    # after the repair A plays in range: no further event
    for _ in range(4):
        a.run_round()
    a.harvest()
    assert a.range_rescales == 1 and a.infer.split_range_status()[0] == 0
    assert b.range_events == 0 and b.infer.act_shift == 0


@pytest.mark.gpu
def test_gpu_no_unmarked_game_from_a_clamp_window():
    """VERDICT r5 Weak #1: a clamp event is FORCED mid-run (one tower bias of the live evaluator is overwritten with 1e6 for three rounds: the
    kernels really clamp) and found by the actor's poll.  Every game that ran a round between the last clean poll and the detecting one
    -- finished in the window or still in progress -- must come out of the harvest marked; games harvested before, and games that
    started after the repair, must not.  The repair restores the evaluator (bias back from the unscaled copy, no further events)."""
    from alpha_zero_amd.core.pipeline import SelfPlayActor

    G = 64
    a = SelfPlayActor(_loud_net(9, 64, 2, gain=1.0, seed=5), game="go", board_size=9, num_games=G, num_simulations=16, num_parallel=4, seed=7, use_graph=True)
    rows_seen = []

    def take():
        st, pi, z, games = a.harvest_tensors()
        rows_seen.extend((int(r[11]) % G, int(r[11]) // G, bool(m)) for r, m in zip(games, a.last_harvest_clamped))

    for _ in range(400):  # (a 9x9 game takes a few hundred rounds at this budget)
        a.run_rounds(10)
        take()
        if len(rows_seen) >= 8:
            break
    assert a.poll_evaluator_range() == 0 and rows_seen and not any(m for _, _, m in rows_seen)
    n_clean = len(rows_seen)
    done0 = a.engine.status()[0][:, 5].copy()            # the last clean poll: games with index < done0[slot] finished before the window
    front = a.clamp_window.next_unharvested.copy()
    good_bias = a.infer.b_sp[1].clone()
    a.infer.b_sp[1].fill_(1.0e6)                          # in place: the captured graph reads it -- the next forwards clamp
    a.run_rounds(3)
    done1 = a.engine.status()[0][:, 5].copy()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert a.poll_evaluator_range() > 0               # the detecting poll: repair + the window is remembered
    assert [m for m in w if "clamped" in str(m.message)] and a.range_rescales == 1 and a.clamp_window.events == 1
    assert torch.equal(a.infer.b_sp[1], a.infer.b32[1] * 2.0 ** -a.infer.act_shift)  # the poke is gone: biases rebuilt from the unscaled copy
    assert a.infer.act_shift > 0 or torch.equal(a.infer.b_sp[1], good_bias)
    for _ in range(800):
        a.run_rounds(10)
        take()
        if (a.clamp_window.hi == -1).all() and any(idx > done1[slot] for slot, idx, _ in rows_seen[n_clean:]):
            break
    a.run_rounds(30)
    take()
    assert (a.clamp_window.hi == -1).all() and a.poll_evaluator_range() == a.range_events  # no further event after the repair
    later = rows_seen[n_clean:]
    for slot, idx, marked in later:
        in_window = done0[slot] <= idx <= done1[slot]       # ran a round inside (last clean poll, detecting poll]
        if in_window:
            assert marked, (slot, idx)
        if idx > done1[slot]:
            assert not marked, (slot, idx)                  # started after the repair
        if marked:
            assert front[slot] <= idx <= done1[slot]
    assert sum(m for _, _, m in later) >= G and any(not m for _, _, m in later)  # all 64 games in progress were suspect; fresh ones are clean
    assert a.clamped_games == sum(m for _, _, m in later)


@pytest.mark.gpu
@pytest.mark.parametrize("S,C", [(9, 128), (9, 64), (17, 64)])
def test_gpu_split_conv_residual_aliasing_contract(S, C):
    """include/azsp.h: residual may alias y at 9x9 (each position is read before it is written, once) and must not at 17x17, where the
    half-board tiles repeat 15 positions per board in later column tiles: AZSP_EINVAL instead of a silently doubled skip."""
    from alpha_zero_amd import _lib

    dll = _lib.load().dll
    B = 37
    g = torch.Generator().manual_seed(S * C)
    x = torch.randn(B, C, S, S, generator=g).abs()
    r = torch.randn(B, C, S, S, generator=g).abs()
    w = torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
    b = torch.randn(C, generator=g) * 0.1
    n = dll.azsp_split_bytes(B, S, C) // 2
    xs, rs, ys = (torch.zeros(n, dtype=torch.float16, device="cuda") for _ in range(3))
    for src, dst in ((x, xs), (r, rs)):
        sc = src.cuda().contiguous(memory_format=torch.channels_last)
        assert dll.azsp_split_layout(sc.data_ptr(), dst.data_ptr(), B, S, C, 1, None, None) == 0
    wsp, bb = split_weights_f16(w).cuda(), b.cuda()
    assert dll.azsp_conv3x3_split(xs.data_ptr(), wsp.data_ptr(), bb.data_ptr(), rs.data_ptr(), ys.data_ptr(), B, S, C, 1, None, None) == 0
    torch.cuda.synchronize()
    rc = dll.azsp_conv3x3_split(xs.data_ptr(), wsp.data_ptr(), bb.data_ptr(), rs.data_ptr(), rs.data_ptr(), B, S, C, 1, None, None)
    torch.cuda.synchronize()
    if S == 17:
        assert rc != 0
    else:
        assert rc == 0 and torch.equal(rs, ys)  # in place == out of place, bit for bit
    assert dll.azsp_conv3x3_split(xs.data_ptr(), wsp.data_ptr(), bb.data_ptr(), None, xs.data_ptr(), B, S, C, 1, None, None) != 0  # x == y: never


@pytest.mark.gpu
def test_gpu_split_conv_keeps_f16_subnormal_operands():
    """The scaled stem weights of a rescaled network have f16-subnormal hi halves: the matrix cores must multiply them, not flush them.
    A convolution whose weights are all below f16's smallest normal number (6.1e-5) against fp64."""
    from alpha_zero_amd import _lib

    dll = _lib.load().dll
    B, C, S = 5, 64, 9
    g = torch.Generator().manual_seed(7)
    x = torch.randn(B, C, S, S, generator=g).abs() * 100.0
    w = torch.randn(C, C, 3, 3, generator=g) * 1e-5  # |w| mostly < 6.1e-5: hi = f16(w) is subnormal
    b = torch.zeros(C)
    n = dll.azsp_split_bytes(B, S, C) // 2
    xs, ys = torch.zeros(n, dtype=torch.float16, device="cuda"), torch.zeros(n, dtype=torch.float16, device="cuda")
    xc = x.cuda().contiguous(memory_format=torch.channels_last)
    assert dll.azsp_split_layout(xc.data_ptr(), xs.data_ptr(), B, S, C, 1, None, None) == 0
    wsp, bb = split_weights_f16(w).cuda(), b.cuda()
    assert float((wsp[0].float().abs() < 6.1e-5).float().mean()) > 0.9
    assert dll.azsp_conv3x3_split(xs.data_ptr(), wsp.data_ptr(), bb.data_ptr(), None, ys.data_ptr(), B, S, C, 0, None, None) == 0
    y = torch.empty_like(xc)
    assert dll.azsp_split_layout(ys.data_ptr(), y.data_ptr(), B, S, C, 0, None, None) == 0
    torch.cuda.synchronize()
    y64 = F.conv2d(x.double(), w.double(), None, padding=1)
    err = (y.cpu().double() - y64).abs().max().item() / y64.abs().max().item()
    assert err <= 5e-6, err  # flushed subnormals would leave y = 0: an error of 1


@pytest.mark.gpu
def test_gpu_poll_range_repairs_a_directly_driven_network():
    """Callers that drive an InferenceNet themselves (evaluation games through DeviceEvaluator, drop-in eval_func wrappers) have no actor
    polling for them: InferenceNet.poll_range reads the network's own record, warns, and re-calibrates on the batch it is given."""
    from alpha_zero_amd import _lib
    from alpha_zero_amd.core.evaluate import DeviceEvaluator

    net = _loud_net(9, 64, 2, gain=2e5, seed=4)
    inf = InferenceNet(net, dtype=torch.float32, binding=_lib.load()).cuda()
    inf.act_calibrated = True  # defeat the first-call calibration: the network clamps at shift 0
    ev = DeviceEvaluator(inf)
    ev.POLL_EVERY = 2
    x = (torch.rand(16, 17, 9, 9, generator=torch.Generator().manual_seed(2)) > 0.6).to(torch.int8).cuda()
    p64, v64 = _ref64(net, x.cpu().float())
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ev.device_eval(x)
        assert not w and inf.act_shift == 0
        ev.device_eval(x)  # the second call polls: events -> warning + re-calibration on this batch
    assert [m for m in w if "clamped" in str(m.message)] and inf.act_shift >= 2 and not inf.split_fallback_reason
    p, v = ev.device_eval(x)
    assert inf.poll_range(x) == 0
    assert (p.cpu().double() - p64).abs().max().item() <= 2e-4 and (v.cpu().double() - v64).abs().max().item() <= 2e-4


def test_calibration_pass_on_the_host_twin():
    """InferenceNet.calibrate_activation_scale driven by hand on the host twin (CPU tensors; the twin has the kernels' arithmetic and their
    range record): a network whose activations reach ~1e6 clamps at shift 0, the calibration pass finds the exact power-of-two scale, and the
    evaluator then matches the fp64 module within the whole-network bound.  Also: a network beyond the format falls back at construction."""
    import engine_util as eu

    bnd = eu.hosttwin_binding()
    net = _loud_net(9, 64, 2, gain=2e5, seed=1)
    x = (torch.rand(2, 17, 9, 9, generator=torch.Generator().manual_seed(3)) > 0.6).float()
    peak = _tower_peak(net, x)
    assert 3e5 < peak < 1.9e6, peak
    p64, v64 = _ref64(net, x)
    inf = InferenceNet(net, dtype=torch.float32, binding=bnd)
    assert bnd.dll.azsp_split_range_status(None, None, 1, None) == 0
    pc, vc = inf.forward_split(x)  # (CPU tensors: no automatic calibration) -- clamps, and the twin's default record says so
    ev = ctypes.c_uint32(0)
    assert bnd.dll.azsp_split_range_status(ctypes.byref(ev), None, 1, None) == 0 and ev.value > 0
    shift, worst = inf.calibrate_activation_scale(x)
    assert inf.act_calibrated and 3 <= shift <= inf.MAX_ACT_SHIFT and abs(worst / peak - 1) < 1e-3 and not inf.split_fallback_reason
    assert bnd.dll.azsp_split_range_status(None, None, 1, None) == 0  # (the probing passes clamped by design)
    p, v = inf.forward_split(x)
    assert bnd.dll.azsp_split_range_status(ctypes.byref(ev), None, 1, None) == 0 and ev.value == 0
    assert (p.double() - p64).abs().max().item() <= 2e-4 and (v.double() - v64).abs().max().item() <= 2e-4
    far = InferenceNet(_loud_net(9, 64, 1, gain=3e8), dtype=torch.float32, binding=bnd)
    # the STEM's folded weights are beyond the format (recorded separately); the tower goes with it: stem outputs of that size
    # are beyond what its f16 pairs carry at any allowed scale
    assert far.stem_fallback_reason and far.split_fallback_reason and "library fp32" in far.evaluator_path(9, "cpu") and far.act_shift == 0
    assert not far.supports_split_features(9, "cuda") and float(far.stem_wsp.abs().max()) == 0.0


def test_loud_stem_weights_fall_back_to_the_library_without_raising_on_the_host_twin():
    """folded stem weights beyond 65504 start the network at a non-zero activation shift; when the calibration pass
    then gives the fp32-class kernels up it resets the shift to 0 -- which used to re-pack the UNSCALED stem weights and raise ValueError
    inside run_round / harvest / poll_range.  The fallback must be silent about the stem (nothing reads stem_wsp any more) and the
    library forward must be the fp32 module's."""
    import engine_util as eu

    bnd = eu.hosttwin_binding()
    torch.manual_seed(2)
    net = AlphaZeroNet((17, 9, 9), 82, 2, 64, 64).eval()
    with torch.no_grad():
        net.conv_block[0].weight.mul_(3e6)       # folded stem weights ~ 1e6: initial shift 4; stem outputs ~ 1e7: beyond shift 9
        net.policy_head[0].weight.div_(3e6)
        net.value_head[0].weight.div_(3e6)
    inf = InferenceNet(net, dtype=torch.float32, binding=bnd)
    assert inf.act_shift >= 3 and not inf.split_fallback_reason and not inf.stem_fallback_reason, (inf.act_shift, inf.stem_fallback_reason)
    x = (torch.rand(2, 17, 9, 9, generator=torch.Generator().manual_seed(4)) > 0.6).float()
    shift, worst = inf.calibrate_activation_scale(x)  # raised ValueError before the fix
    assert inf.split_fallback_reason and shift == 0 and inf.act_shift == 0 and worst > 65504.0, (shift, worst)  # (a lower bound: saturated passes hide the true maximum)
    assert float(inf.stem_wsp.abs().max()) == 0.0 and "library fp32" in inf.evaluator_path(9, "cpu")
    inf.set_act_shift(0)  # idempotent, still no ValueError
    p, v = inf._forward_after_split_fallback(x, None, None, None)
    with torch.no_grad():
        lg, vr = net(x)
    assert (p - torch.softmax(lg, -1)).abs().max().item() <= 1e-4 and (v - vr.squeeze(1)).abs().max().item() <= 1e-4
    assert bnd.dll.azsp_split_range_status(None, None, 1, None) == 0
