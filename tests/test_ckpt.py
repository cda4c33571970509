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
