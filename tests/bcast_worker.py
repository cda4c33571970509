"""Worker of test_broadcast_weights_two_ranks_gloo: every rank starts from different weights; after broadcast_weights(src=0)
all ranks hold rank 0's parameters AND BatchNorm statistics bit for bit."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

rank, world, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
dist.init_process_group("gloo", rank=rank, world_size=world)
from alpha_zero_amd.core.gather import broadcast_weights  # noqa: E402
from alpha_zero_amd.core.network import AlphaZeroNet  # noqa: E402

torch.manual_seed(100 + rank)
net = AlphaZeroNet((17, 5, 5), 26, 2, 8, 8)
with torch.no_grad():
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):  # running statistics differ per rank too
            m.running_mean.normal_()
            m.running_var.uniform_(0.5, 2.0)
            m.num_batches_tracked.fill_(7 + rank)
before = {k: v.clone() for k, v in net.state_dict().items()}
broadcast_weights(net, src=0)
sd = net.state_dict()
np.savez(os.path.join(outdir, f"w{rank}.npz"), **{k: v.numpy() for k, v in sd.items()})
if rank != 0:  # something must actually have changed on the receiving rank
    assert any(not torch.equal(before[k], sd[k]) for k in sd if sd[k].is_floating_point())
dist.barrier()
dist.destroy_process_group()
