"""Sample gather to the replay rank: replaces `data_queue.put` across actor processes
(reference: alpha_zero/core/pipeline.py:283 -> learner :485).  One process per GPU; games never interact
during search, so this is the only exchange on the data path: per harvest, one all_gather of counts and ONE
`gather` of a packed byte buffer to `dst` (RCCL over xGMI when the backend is "nccl": every sender uses
its own direct link into the root).  Works unchanged on gloo/CPU tensors (tests).

Wire format per rank and harvest: [maxn rows][row_bytes] + [maxk][64] bytes, a row =
ceil(17 N^2 / 8) bytes of bit-packed 0/1 observation planes | A float32 of pi | 1 float32 of z  (173 + 328 + 4 = 505 B per
sample at 9x9 instead of 1709 B unpacked), the tail = the 16-int game records.  Padded to the largest rank of this harvest."""
import numpy as np
import torch
import torch.distributed as dist

def _bit_weights(dev):
    return torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.uint8, device=dev)


def pack_samples(states, pi, z):
    """states int8[n,C,N,N] of 0/1 planes, pi f32[n,A], z f32[n] -> uint8[n, row_bytes] (bit-packed planes | pi bytes | z bytes)."""
    n = states.shape[0]
    if n == 0:  # nothing to pack (a rank without a finished game): empty tensors have no byte views
        return torch.zeros((0, (int(np.prod(states.shape[1:])) + 7) // 8 + 4 * pi.shape[1] + 4), dtype=torch.uint8, device=states.device)
    flat = states.reshape(n, int(np.prod(states.shape[1:]))).to(torch.uint8)
    nb = (flat.shape[1] + 7) // 8
    if flat.shape[1] != nb * 8:
        flat = torch.cat([flat, torch.zeros((n, nb * 8 - flat.shape[1]), dtype=torch.uint8, device=flat.device)], 1)
    bits = (flat.view(n, nb, 8) * _bit_weights(flat.device)).sum(-1, dtype=torch.int32).to(torch.uint8)
    return torch.cat([bits, pi.contiguous().view(torch.uint8).reshape(n, 4 * pi.shape[1]), z.contiguous().reshape(n, 1).view(torch.uint8).reshape(n, 4)], 1)


def unpack_samples(rows, state_shape, A):
    """Inverse of pack_samples for rows uint8[n, row_bytes]; state_shape = (C, N, N)."""
    n = rows.shape[0]
    nel = int(np.prod(state_shape))
    nb = (nel + 7) // 8
    bits = rows[:, :nb]
    planes = ((bits.unsqueeze(-1) & _bit_weights(rows.device)) != 0).to(torch.int8).reshape(n, nb * 8)[:, :nel].reshape((n,) + tuple(state_shape))
    # a fresh, DENSE, 4-byte aligned storage: a [1, k] slice of a row whose length is not a multiple of 4 (9x9 Go: 505 B) counts as
    # dense under preserve_format and would keep the row stride, which the float32 view rejects
    pi = rows[:, nb:nb + 4 * A].clone(memory_format=torch.contiguous_format).view(torch.float32).reshape(n, A)
    z = rows[:, nb + 4 * A:nb + 4 * A + 4].clone(memory_format=torch.contiguous_format).view(torch.float32).reshape(n)
    return planes, pi, z


def gather_samples(states, pi, z, games, dst=0, group=None):
    """states int8[n,17,N,N], pi f32[n,A], z f32[n] (device tensors of this rank), games int32[k,16] (numpy).
    Returns on `dst` the concatenation over ranks (rank order) with game `start` offsets rebased and column 15
    (slot) made global as rank*2^20 + slot; on other ranks returns None.

    Collectives only (every rank takes part in every call, no point-to-point pairing to get wrong): one all_gather of the
    (samples, games) counts, then ONE `gather` of the packed byte buffer to `dst`, padded to the largest count of this harvest.
    The volume is ~0.5 KB per sample, a few MB per harvest.  Without an initialised process group (a plain single-GPU run) the
    inputs are returned as they are; with one -- even of a single rank -- the collectives run, so the RCCL path can be exercised
    on one GPU (tests/test_nccl_single_rank.py)."""
    if not dist.is_available() or not dist.is_initialized():
        return states, pi, z, games
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = states.device
    state_shape, A = tuple(states.shape[1:]), pi.shape[1]
    g_t = torch.as_tensor(np.ascontiguousarray(games, dtype=np.int32)).reshape(-1, 16).to(dev)
    counts = torch.tensor([states.shape[0], g_t.shape[0]], dtype=torch.int64, device=dev)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts, group=group)
    all_counts = torch.stack(all_counts).cpu().numpy()  # (the harvest that produced the inputs already synchronised this stream)
    maxn, maxk = int(all_counts[:, 0].max()), int(all_counts[:, 1].max())
    if maxn == 0 and maxk == 0:
        return (states[:0], pi[:0], z[:0], np.zeros((0, 16), dtype=np.int32)) if rank == dst else None
    rows = pack_samples(states, pi, z)
    rb = rows.shape[1]
    buf = torch.zeros((maxn * rb + maxk * 64,), dtype=torch.uint8, device=dev)
    if rows.numel():
        buf[: rows.numel()] = rows.reshape(-1)
    if g_t.numel():  # (a rank with no finished game in this harvest: an empty [0, 16] tensor has no byte view -- found by the 8-rank gloo test)
        buf[maxn * rb: maxn * rb + g_t.numel() * 4] = g_t.contiguous().view(torch.uint8).reshape(-1)
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst, group=group)
    if rank != dst:
        return None
    parts_s, parts_p, parts_z, parts_g = [], [], [], []
    base = 0
    for r in range(world):
        n, k = int(all_counts[r, 0]), int(all_counts[r, 1])
        if n:
            s_r, p_r, z_r = unpack_samples(out[r][: n * rb].reshape(n, rb), state_shape, A)
            parts_s.append(s_r), parts_p.append(p_r), parts_z.append(z_r)
        if k:
            grows = out[r][maxn * rb: maxn * rb + k * 64].clone().view(torch.int32).reshape(k, 16).cpu().numpy().copy()
            grows[:, 0] += base
            grows[:, 15] += r << 20
            parts_g.append(grows)
        base += n
    if not parts_s:
        return states[:0], pi[:0], z[:0], (np.concatenate(parts_g) if parts_g else np.zeros((0, 16), dtype=np.int32))
    return torch.cat(parts_s), torch.cat(parts_p), torch.cat(parts_z), (np.concatenate(parts_g) if parts_g else np.zeros((0, 16), dtype=np.int32))


def broadcast_weights(network: torch.nn.Module, src=0, group=None):
    """New-checkpoint hand-over to every actor rank: replaces the checkpoint FILE + mp.Value path signalling of the
    reference (pipeline.py:232-239, :597-610) by one broadcast of the flattened parameters and BatchNorm buffers
    (RCCL ncclBroadcast over xGMI with the "nccl" backend; 3.0 M values for the 10x128 net).  Runs whenever a process group is
    initialised (a single-rank group included)."""
    if not dist.is_available() or not dist.is_initialized():
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
