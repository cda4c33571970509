"""SGF text and checkpoint dictionary vs the reference (shared by the CPU and GPU tiers)."""
import json
import os

import torch

import dropin_checks as dc


def check_sgf(kind, golden_dir, monkeypatch):
    from alpha_zero_amd.envs.base import PlayerMove
    from alpha_zero_amd.utils import sgf

    monkeypatch.setattr(sgf, "get_time_stamp", lambda *a, **k: "2024-01-02 03:04:05")
    for rec in json.load(open(os.path.join(golden_dir, "sgf_records.json"))):
        if rec["game"] == "raw":
            hist = [PlayerMove(c, m) for c, m in rec["history"]]
            assert sgf.make_sgf(rec["n"], hist, rec["result"], comments=rec["comments"], **rec["kwargs"]) == rec["sgf"]
            continue
        env = dc.make_env(kind, rec["game"], rec["n"], **({"komi": rec["komi"]} if rec["game"] == "go" else {}))
        env.reset()
        for a in rec["moves"]:
            env.step(a)
        assert env.get_result_string() == rec["result"]
        assert env.to_sgf() == rec["sgf"]


def check_checkpoint(tmp_path):
    from alpha_zero_amd.core.network import AlphaZeroNet
    from alpha_zero_amd.utils.sgf import load_checkpoint, save_checkpoint

    torch.manual_seed(0)
    net = AlphaZeroNet((17, 5, 5), 26, 2, 16, 16)
    opt = torch.optim.SGD(net.parameters(), lr=0.1, momentum=0.9)
    sch = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[10, 20], gamma=0.1)
    path = os.path.join(tmp_path, "training_steps_123.ckpt")
    save_checkpoint(path, net, opt, sch, 123)
    raw = torch.load(path, map_location="cpu")
    assert set(raw) == {"network", "optimizer", "lr_scheduler", "training_steps"} and raw["training_steps"] == 123
    assert list(raw["network"])[0] == "conv_block.0.weight"  # the reference's module names
    net2 = AlphaZeroNet((17, 5, 5), 26, 2, 16, 16)
    assert load_checkpoint(path, net2) == 123
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), net2.state_dict().values()))
