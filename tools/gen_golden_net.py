"""Golden vectors for the network: small seeded instances of the REFERENCE AlphaZeroNet (Go and Gomoku
stems), their state_dicts, inputs and fp32 outputs -> tests/golden/net_{go,gomoku}.pt (data only)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

ref_harness.install(9)
from alpha_zero.core.network import AlphaZeroNet  # noqa: E402

for name, shape, A, gomoku in (("go", (17, 9, 9), 82, False), ("gomoku", (17, 13, 13), 169, True)):
    torch.manual_seed(1)
    net = AlphaZeroNet(shape, A, num_res_block=2, num_filters=16, num_fc_units=32, gomoku=gomoku)
    # non-trivial BatchNorm statistics, as after training
    g = torch.Generator().manual_seed(2)
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.2)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.weight.data.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=g) * 0.1)
    net.eval()
    x = (torch.rand((32,) + shape, generator=g) < 0.2).to(torch.int8)
    with torch.no_grad():
        logits, value = net(x.float())
    torch.save({"state_dict": net.state_dict(), "x": x, "logits": logits, "value": value,
                "args": dict(input_shape=shape, num_actions=A, num_res_block=2, num_filters=16, num_fc_units=32, gomoku=gomoku)},
               os.path.join(ROOT, "tests", "golden", f"net_{name}.pt"))
    print(name, logits.shape, value.shape)
