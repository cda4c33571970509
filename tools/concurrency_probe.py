"""Which evaluator kernel changes its results when two forwards run CONCURRENTLY on two streams?  (The two half-batch streams of
SelfPlayActor(overlap_engine=True) produced priors that differ from the serial run in a few rows per round, although the same
half-batches on ONE stream are bit-identical.)  Runs the 9x9 x 128 evaluator on two different batches, serially (reference) and
concurrently, with a checkpoint (clone) after the stem, after every tower convolution, after the head planes and after the FC layers,
and reports the first stage whose output differs."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import engine_util as eu  # noqa: E402
from alpha_zero_amd import _lib  # noqa: E402
from alpha_zero_amd.core.network import AlphaZeroNet, InferenceNet  # noqa: E402

game = sys.argv[1] if len(sys.argv) > 1 else "go"
n, filters, A = (9, 128, 82) if game == "go" else (13, 64, 169)
torch.manual_seed(4)
net = AlphaZeroNet((17, n, n), A, 2, filters, 64, gomoku=(game != "go"))
bnd = _lib.load()
inf = InferenceNet(net, dtype=torch.bfloat16, binding=bnd).cuda()
dll = bnd.dll
rows = [4608, 4864]
feats = [eu.tile_features((torch.rand(r, 17, n, n) > 0.6).float()).cuda() for r in rows]
S = n + 2 * (inf.stem_pad - 1)
C = filters


def staged(k, feat, B):
    """forward_tiled's launch sequence with a clone after every stage (on the current stream)."""
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    a, m, o = inf._tiled_buffers(B, S, C, feat.device, 1 + k)
    out = []
    assert dll.azsp_stem_tiled(feat.data_ptr(), inf.stem_wp.data_ptr(), inf.stem_b32.data_ptr(), a.data_ptr(), B, n, C, inf.stem_pad, 1, st) == 0
    out.append(("stem", a.clone()))
    for i in range(inf.n_blocks):
        if inf.use_fused_block and (C, S) in ((64, 17), (64, 9)):
            assert dll.azsp_resblock_tiled(a.data_ptr(), inf.wp[2 * i].data_ptr(), inf.b32[2 * i].data_ptr(), inf.wp[2 * i + 1].data_ptr(),
                                           inf.b32[2 * i + 1].data_ptr(), o.data_ptr(), B, S, C, st) == 0
            out.append((f"block{i}", o.clone()))
        else:
            assert dll.azsp_conv3x3_tiled(a.data_ptr(), inf.wp[2 * i].data_ptr(), inf.b32[2 * i].data_ptr(), None, m.data_ptr(), B, S, C, 1, st) == 0
            out.append((f"conv{2 * i}", m.clone()))
            assert dll.azsp_conv3x3_tiled(m.data_ptr(), inf.wp[2 * i + 1].data_ptr(), inf.b32[2 * i + 1].data_ptr(), a.data_ptr(), o.data_ptr(), B, S, C, 1, st) == 0
            out.append((f"conv{2 * i + 1}", o.clone()))
        a, o = o, a
    k1, k2 = inf.fc_wp.shape[1], inf.fc_w1.shape[1]
    pol, val, pri, v = inf._head_buffers(B, k1, k2, feat.device, 1 + k)
    assert dll.azsp_head_tiled(a.data_ptr(), inf.head_w32.data_ptr(), inf.head_b32.data_ptr(), pol.data_ptr(), val.data_ptr(), B, S, C, inf.npol, inf.nval, k1, k2, st) == 0
    out.append(("head", torch.cat([pol.flatten(), val.flatten()]).clone()))
    assert dll.azsp_fc_heads(pol.data_ptr(), val.data_ptr(), inf.fc_wp.data_ptr(), inf.fc_bp.data_ptr(), k1 // 16, inf.fc_w1.data_ptr(), inf.fc_b1.data_ptr(),
                             k2 // 16, inf.fc_w2.data_ptr(), ctypes.c_float(inf.fc_b2), pri.data_ptr(), v.data_ptr(), B, inf.num_actions, inf.fc_width, st) == 0
    out.append(("fc", torch.cat([pri.flatten(), v]).clone()))
    return out


ref = [staged(k, feats[k], rows[k]) for k in range(2)]
torch.cuda.synchronize()
again = [staged(k, feats[k], rows[k]) for k in range(2)]
torch.cuda.synchronize()
print("serial repeat identical:", all(torch.equal(x[1], y[1]) for k in range(2) for x, y in zip(ref[k], again[k])))
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
bad = {}
for it in range(40):
    outs = [None, None]
    for k in range(2):
        with torch.cuda.stream(streams[k]):
            outs[k] = staged(k, feats[k], rows[k])
    torch.cuda.synchronize()
    for k in range(2):
        for (name, x), (_, y) in zip(ref[k], outs[k]):
            if not torch.equal(x, y):
                d = (x.float() - y.float()).abs()
                bad.setdefault((k, name), []).append((it, int((d > 0).sum()), float(d.max())))
                break  # only the first differing stage of this forward
print("concurrent runs: first differing stage per (batch, stage): ", {k: (len(v), v[:3]) for k, v in bad.items()} or "none")
