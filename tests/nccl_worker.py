"""Worker of tests/test_nccl_single_rank.py (`-m gpu`): a SINGLE-rank "nccl" process group on cuda:0, so that RCCL initialisation,
barrier, the count all_gather, the packed sample `dist.gather` on device tensors and the weight broadcast execute on hardware once
(the 8-GPU run itself is the driver's; reference fan-out being replaced: training_go.py:317-347, pipeline.py:283)."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
assert dist.get_backend() == "nccl" and dist.get_world_size() == 1

from alpha_zero_amd.core import gather as G  # noqa: E402
from alpha_zero_amd.core.network import AlphaZeroNet  # noqa: E402
from alpha_zero_amd.core.pipeline import SelfPlayActor  # noqa: E402

calls = {"all_gather": 0, "gather": 0, "broadcast": 0}
for name in calls:  # count the collectives that really run (no early return for a single-rank group)
    orig = getattr(dist, name)

    def wrap(*a, _o=orig, _n=name, **k):
        calls[_n] += 1
        return _o(*a, **k)

    setattr(dist, name, wrap)
    setattr(G.dist, name, wrap)

torch.manual_seed(1)
net = AlphaZeroNet((17, 5, 5), 26, 1, 8, 8)
act = SelfPlayActor(net, game="go", board_size=5, num_games=64, num_simulations=16, num_parallel=4, warm_up_steps=4, device=dev,
                    net_dtype=torch.float32, use_graph=False, seed=1, rank=0)
got = 0
for _ in range(40):
    act.run_rounds(25)
    st, pi, z, games = act.harvest_tensors(clone=True)
    res = G.gather_samples(st, pi, z, games, dst=0)
    assert res is not None
    assert torch.equal(res[0], st) and torch.equal(res[1], pi) and torch.equal(res[2], z)  # bit-packed planes / float bytes round trip on the device
    assert res[0].is_cuda and np.array_equal(res[3][:, :15], games[:, :15]) and np.array_equal(res[3][:, 15], games[:, 15])  # rank 0: slot + 0 << 20
    got += int(st.shape[0])
    if got > 200:
        break
assert got > 200 and calls["gather"] >= 1 and calls["all_gather"] >= calls["gather"], (got, calls)
net2 = AlphaZeroNet((17, 5, 5), 26, 1, 8, 8).to(dev)
before = [p.detach().clone() for p in net2.parameters()]
G.broadcast_weights(net2, src=0)
assert calls["broadcast"] == 1 and all(torch.equal(a, b) for a, b in zip(before, net2.parameters()))
dist.barrier()
torch.cuda.synchronize()
print(json.dumps({"ok": True, "samples": got, "collectives": calls, "backend": dist.get_backend()}), flush=True)
dist.destroy_process_group()
