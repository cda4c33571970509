"""Sample gather to the replay rank: replaces `data_queue.put` across actor processes
(reference: alpha_zero/core/pipeline.py:283 -> learner :485).  One process per GPU; games never interact
during search, so this is the only exchange on the data path: per harvest, one all_gather of counts and
point-to-point sends of the finished-game tensors to `dst` (RCCL over xGMI when the backend is "nccl":
every sender uses its own direct link into the root).  Works unchanged on gloo/CPU tensors (tests)."""
import numpy as np
import torch
import torch.distributed as dist


def gather_samples(states, pi, z, games, dst=0, group=None):
    """states int8[n,17,N,N], pi f32[n,A], z f32[n] (device tensors of this rank), games int32[k,16] (numpy).
    Returns on `dst` the concatenation over ranks (rank order) with game `start` offsets rebased and column 15
    (slot) made global as rank*2^20 + slot; on other ranks returns None."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return states, pi, z, games
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = states.device
    g_t = torch.as_tensor(np.ascontiguousarray(games, dtype=np.int32)).to(dev)
    counts = torch.tensor([states.shape[0], g_t.shape[0]], dtype=torch.int64, device=dev)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    all_counts = torch.stack(all_counts).cpu().numpy()
    if rank != dst:
        ops = []
        if counts[0] > 0:
            ops += [dist.P2POp(dist.isend, t.contiguous(), dst, group) for t in (states, pi, z)]
        if counts[1] > 0:
            ops.append(dist.P2POp(dist.isend, g_t, dst, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return None
    parts_s, parts_p, parts_z, parts_g = [], [], [], []
    ops, bufs = [], {}
    for r in range(world):
        n, k = int(all_counts[r, 0]), int(all_counts[r, 1])
        if r == dst:
            bufs[r] = (states, pi, z, g_t)
            continue
        bs = torch.empty((n,) + tuple(states.shape[1:]), dtype=states.dtype, device=dev)
        bp = torch.empty((n,) + tuple(pi.shape[1:]), dtype=pi.dtype, device=dev)
        bz = torch.empty((n,), dtype=z.dtype, device=dev)
        bg = torch.empty((k, 16), dtype=torch.int32, device=dev)
        bufs[r] = (bs, bp, bz, bg)
        if n > 0:
            ops += [dist.P2POp(dist.irecv, t, r, group) for t in (bs, bp, bz)]
        if k > 0:
            ops.append(dist.P2POp(dist.irecv, bg, r, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    base = 0
    for r in range(world):
        bs, bp, bz, bg = bufs[r]
        gg = bg.cpu().numpy().copy()
        if len(gg):
            gg[:, 0] += base
            gg[:, 15] += r << 20
        base += bs.shape[0]
        parts_s.append(bs), parts_p.append(bp), parts_z.append(bz), parts_g.append(gg)
    return torch.cat(parts_s), torch.cat(parts_p), torch.cat(parts_z), np.concatenate(parts_g) if parts_g else games


def broadcast_weights(network: torch.nn.Module, src=0, group=None):
    """New-checkpoint hand-over to every actor rank: replaces the checkpoint FILE + mp.Value path signalling of the
    reference (pipeline.py:232-239, :597-610) by one broadcast of the flattened parameters and BatchNorm buffers
    (RCCL ncclBroadcast over xGMI with the "nccl" backend; 3.0 M values for the 10x128 net)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return network
    tensors = [t for t in list(network.parameters()) + list(network.buffers()) if t.is_floating_point()]
    flat = torch.cat([t.detach().reshape(-1).to(torch.float32) for t in tensors])
    dist.broadcast(flat, src, group=group)
    o = 0
    with torch.no_grad():
        for t in tensors:
            n = t.numel()
            t.copy_(flat[o:o + n].reshape(t.shape).to(t.dtype))
            o += n
    return network
