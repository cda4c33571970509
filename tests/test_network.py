"""AlphaZeroNet must be state_dict-compatible with the reference module and reproduce its outputs;
InferenceNet (BN folded, channels-last) must agree with it (fp32 1e-4; bf16 on the GPU tier)."""
import os

import pytest
import torch

from alpha_zero_amd.core.network import AlphaZeroNet, InferenceNet


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
