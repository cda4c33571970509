"""The reference's shipped checkpoint (checkpoints/gomoku/13x13/training_steps_200000.ckpt: 10 x 40, 80 fc units) as a fixture
(tools/gen_golden_ckpt.py): it loads into our AlphaZeroNet unchanged (state_dict keys / shapes, core/network.py:85-173), the
outputs equal the reference module's, and the function-preserving widening to 64 filters -- which puts a TRAINED network on the
hand-written 17x17x64 evaluator kernels -- changes nothing."""
import os

import numpy as np
import pytest
import torch

from alpha_zero_amd.core.network import AlphaZeroNet, InferenceNet, widen_network


def load_shipped(golden_dir):
    st = torch.load(os.path.join(golden_dir, "gomoku13_ckpt200000_network.pt"), map_location="cpu", weights_only=True)  # tensors + an int only
    net = AlphaZeroNet((17, 13, 13), 169, 10, 40, 80, gomoku=True)
    missing = net.load_state_dict(st["network"], strict=True)  # the reference's own keys, no renaming
    assert not missing.missing_keys and not missing.unexpected_keys and st["training_steps"] == 200000
    return net.eval()


def test_shipped_checkpoint_loads_and_matches_reference_outputs(golden_dir):
    net = load_shipped(golden_dir)
    g = np.load(os.path.join(golden_dir, "gomoku13_ckpt200000_outputs.npz"))
    x = torch.from_numpy(g["states"]).float()
    with torch.no_grad():
        logits, v = net(x)
        wl, wv = widen_network(net, 64)(x)
        pri, vi = InferenceNet(net, dtype=torch.float32)(x)
    assert np.abs(logits.numpy() - g["logits"]).max() <= 2e-5 and np.abs(v.squeeze(1).numpy() - g["value"]).max() <= 2e-6
    assert (wl - logits).abs().max().item() <= 1e-5 and (wv - v).abs().max().item() <= 1e-6  # zero channels: same function
    ref_p = torch.softmax(torch.from_numpy(g["logits"]), -1)
    assert (pri - ref_p).abs().max().item() <= 1e-5 and (vi - torch.from_numpy(g["value"])).abs().max().item() <= 1e-5
    assert float(ref_p.max(-1).values.mean()) > 0.5  # a trained, sharp policy -- not a random-init network


@pytest.mark.gpu
def test_gpu_shipped_checkpoint_on_the_hand_written_kernels(golden_dir):
    """Trained weights, widened 40 -> 64 filters, through azsp_stem_tiled / azsp_conv3x3_tiled (k_conv3x3_t64) / azsp_head_tiled /
    azsp_fc_heads in bf16 vs the reference module's fp32 outputs.  Tolerances as for the random-init networks."""
    import engine_util as eu
    from alpha_zero_amd import _lib

    net = widen_network(load_shipped(golden_dir), 64)
    g = np.load(os.path.join(golden_dir, "gomoku13_ckpt200000_outputs.npz"))
    inf = InferenceNet(net, dtype=torch.bfloat16, binding=_lib.load()).cuda()
    assert inf.supports_tiled_features(13, "cuda")
    x = torch.from_numpy(g["states"]).float()
    pri, v = inf.forward_tiled(eu.tile_features(x).cuda(), x.shape[0], 13)
    ref_p = torch.softmax(torch.from_numpy(g["logits"]), -1)
    dp, dv = (pri.cpu() - ref_p).abs().max().item(), (v.cpu() - torch.from_numpy(g["value"])).abs().max().item()
    agree = (pri.cpu().argmax(-1) == ref_p.argmax(-1)).float().mean().item()
    assert dp <= 6e-2 and dv <= 6e-2 and agree >= 0.95, (dp, dv, agree)


def test_checkpoint_loader_never_unpickles_code_by_default(tmp_path, monkeypatch):
    """load_checkpoint_state (the actor's start-up load and its hot-swap, pipeline.py:208-212 / :232-239 in the reference): the
    learner's dictionary (network, optimizer state, MultiStepLR state with its collections.Counter, training_steps) loads with the
    safe loader; a file that needs the full unpickler is REFUSED unless the caller opts in, and a missing / corrupt file raises its
    own error instead of being retried unsafely."""
    import collections
    import pickle

    import pytest

    from alpha_zero_amd.core.pipeline import load_checkpoint_state

    good = os.path.join(str(tmp_path), "good.ckpt")
    torch.save({"network": {"w": torch.ones(3)}, "optimizer": {"state": {}, "param_groups": [{"lr": 0.1}]},
                "lr_scheduler": {"milestones": collections.Counter({100: 1, 200: 1}), "gamma": 0.1, "last_epoch": 5}, "training_steps": 7}, good)
    st = load_checkpoint_state(good)
    assert st["training_steps"] == 7 and st["lr_scheduler"]["milestones"][100] == 1 and torch.equal(st["network"]["w"], torch.ones(3))

    marker = os.path.join(str(tmp_path), "code_ran")

    class Code:  # unpickling this object calls a function (here: os.mkdir): exactly what weights_only=True exists to stop
        def __reduce__(self):
            return (os.mkdir, (marker,))

    bad = os.path.join(str(tmp_path), "bad.ckpt")
    torch.save({"network": {}, "training_steps": 1, "extra": Code()}, bad)
    monkeypatch.delenv("AZSP_ALLOW_PICKLE_CKPT", raising=False)
    with pytest.raises(pickle.UnpicklingError):
        load_checkpoint_state(bad)
    assert not os.path.exists(marker)
    assert load_checkpoint_state(bad, allow_pickle=True)["training_steps"] == 1 and os.path.isdir(marker)  # explicit opt-in only
    with pytest.raises(FileNotFoundError):
        load_checkpoint_state(os.path.join(str(tmp_path), "missing.ckpt"))
    trunc = os.path.join(str(tmp_path), "trunc.ckpt")
    open(trunc, "wb").write(open(good, "rb").read()[:100])
    with pytest.raises(Exception):
        load_checkpoint_state(trunc)


def test_widen_for_kernels_picks_the_next_supported_width_and_keeps_the_function():
    """Networks whose filter count has no hand-written evaluator kernel are widened (function-preserving) to the next width that has:
    the shipped 10 x 40 Gomoku shape -> 64, a 96-filter 9x9 Go network -> 128; supported widths, widths beyond the largest kernel and
    boards without kernels are left alone.  The widened copy computes the SAME function (zero-weight channels: exact in fp32)."""
    from alpha_zero_amd.core.network import widen_for_kernels

    torch.manual_seed(3)
    net = AlphaZeroNet((17, 13, 13), 169, 2, 40, 80, gomoku=True).eval()
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0, 0.2), m.running_var.uniform_(0.5, 1.5), m.weight.uniform_(0.7, 1.3), m.bias.normal_(0, 0.2)
    w, note = widen_for_kernels(net, 13, torch.float32)
    assert w.conv_block[0].out_channels == 64 and "40 -> 64" in note
    x = (torch.rand(5, 17, 13, 13, generator=torch.Generator().manual_seed(1)) > 0.6).float()
    with torch.no_grad():
        (l0, v0), (l1, v1) = net(x), w(x)
    assert (l0 - l1).abs().max().item() <= 1e-5 and (v0 - v1).abs().max().item() <= 1e-5
    go96 = AlphaZeroNet((17, 9, 9), 82, 1, 96, 32)
    assert widen_for_kernels(go96, 9, torch.float32)[0].conv_block[0].out_channels == 128
    assert widen_for_kernels(go96, 9, torch.bfloat16)[0].conv_block[0].out_channels == 128
    for f, n, gomoku in ((128, 9, False), (64, 9, False), (64, 13, True), (300, 9, False), (32, 7, False)):
        same = AlphaZeroNet((17, n, n), n * n + (0 if gomoku else 1), 1, f, 16, gomoku=gomoku)
        got, note = widen_for_kernels(same, n, torch.float32)
        assert got is same and note == ""
    go19 = AlphaZeroNet((17, 19, 19), 362, 1, 192, 16)
    assert widen_for_kernels(go19, 19, torch.float32)[0] is go19                                   # no fp32-class kernels at 19x19
    assert widen_for_kernels(go19, 19, torch.bfloat16)[0].conv_block[0].out_channels == 256       # the bf16 kernels exist at 256


@pytest.mark.gpu
def test_gpu_actor_runs_the_shipped_network_width_on_hand_written_kernels_by_default():
    """SelfPlayActor's defaults with a network of the shipped Gomoku shape (10 x 40, training_gomoku.py:37-38): the reference's precision
    (fp32 class) on the hand-written split-precision kernels, reached by widening 40 -> 64 filters -- no library fallback, no warning."""
    import warnings

    from alpha_zero_amd.core.pipeline import SelfPlayActor

    torch.manual_seed(2)
    net = AlphaZeroNet((17, 13, 13), 169, 2, 40, 80, gomoku=True)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)
        act = SelfPlayActor(net, game="gomoku", board_size=13, num_games=64, num_simulations=16, num_parallel=4, warm_up_steps=4, device="cuda",
                            seed=3)
        assert "hand-written" in act.evaluator_path and "split-precision" in act.evaluator_path and "widened 40 -> 64" in act.evaluator_path
        act.run_rounds(260)  # 16 simulations / P = 4: ~5 rounds per move
        got = act.harvest()
    assert len(got) >= 4 and act.range_events == 0
    # the evaluator the actor built equals the original 40-filter network (fp32 module on the CPU) on real positions
    x = torch.stack([torch.from_numpy(t.state.astype("float32")) for seq, _ in got[:8] for t in seq[:4]])
    pri, v = act.infer(x.cuda())
    with torch.no_grad():
        lg, vr = net.eval()(x)
    assert (pri.cpu() - torch.softmax(lg, -1)).abs().max().item() <= 2e-5 and (v.cpu() - vr.squeeze(1)).abs().max().item() <= 2e-5
