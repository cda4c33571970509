"""Sample gather to the replay rank: replaces `data_queue.put` across actor processes
(reference: alpha_zero/core/pipeline.py:283 -> learner :485).  One process per GPU; games never interact
during search, so this is the only exchange on the data path: per harvest, one all_gather of counts and one
`gather` per tensor of the finished games to `dst` (RCCL over xGMI when the backend is "nccl": every sender uses
its own direct link into the root).  Works unchanged on gloo/CPU tensors (tests)."""
import numpy as np
import torch
import torch.distributed as dist


def gather_samples(states, pi, z, games, dst=0, group=None):
    """states int8[n,17,N,N], pi f32[n,A], z f32[n] (device tensors of this rank), games int32[k,16] (numpy).
    Returns on `dst` the concatenation over ranks (rank order) with game `start` offsets rebased and column 15
    (slot) made global as rank*2^20 + slot; on other ranks returns None.

    Collectives only (every rank takes part in every call, no point-to-point pairing to get wrong): one all_gather of the
    (samples, games) counts, then one `gather` to `dst` per tensor, padded to the largest count of this harvest.  The volume
    is ~1.7 KB per sample, a few MB per harvest: padding costs nothing against one xGMI link."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return states, pi, z, games
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = states.device
    g_t = torch.as_tensor(np.ascontiguousarray(games, dtype=np.int32)).reshape(-1, 16).to(dev)
    counts = torch.tensor([states.shape[0], g_t.shape[0]], dtype=torch.int64, device=dev)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    all_counts = torch.stack(all_counts).cpu().numpy()
    maxn, maxk = int(all_counts[:, 0].max()), int(all_counts[:, 1].max())

    def gather_padded(t, m):
        """t [c, ...] -> list over ranks of [m, ...] on dst (None elsewhere); every rank calls with the same m."""
        pad = torch.zeros((m,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
        pad[: t.shape[0]] = t
        out = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, out, dst=dst, group=group)
        return out

    gs = gather_padded(states, maxn) if maxn else None
    gp = gather_padded(pi, maxn) if maxn else None
    gz = gather_padded(z, maxn) if maxn else None
    gg = gather_padded(g_t, maxk) if maxk else None
    if rank != dst:
        return None
    parts_s, parts_p, parts_z, parts_g = [], [], [], []
    base = 0
    for r in range(world):
        n, k = int(all_counts[r, 0]), int(all_counts[r, 1])
        if n:
            parts_s.append(gs[r][:n]), parts_p.append(gp[r][:n]), parts_z.append(gz[r][:n])
        if k:
            rows = gg[r][:k].cpu().numpy().copy()
            rows[:, 0] += base
            rows[:, 15] += r << 20
            parts_g.append(rows)
        base += n
    if not parts_s:
        return states[:0], pi[:0], z[:0], np.zeros((0, 16), dtype=np.int32)
    return torch.cat(parts_s), torch.cat(parts_p), torch.cat(parts_z), (np.concatenate(parts_g) if parts_g else np.zeros((0, 16), dtype=np.int32))


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
