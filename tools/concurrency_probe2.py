"""Follow-up of tools/overlap_debug.py: inside the actor flow, two half-batch forwards launched together on two streams give priors
that differ in a few rows from the same forwards run one after the other (tools/concurrency_probe.py, which runs two forwards on
separately allocated inputs, sees no difference).  Here: rounds of the real actor; after the engine kernels of a round the two half
forwards run (a) concurrently and then (b) serially on the SAME features, with the stem output, the tower output, the head planes
and the priors of both captured; reports the first stage that differs."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from alpha_zero_amd.core.network import AlphaZeroNet  # noqa: E402
from overlap_actor import OverlapActor  # noqa: E402

if os.environ.get("AZ_PROBE_LIB"):  # an alternative build of the library (e.g. the head kernel with 96 KB of extra dynamic LDS, so that
    import ctypes                    # none of its workgroups can share a CU with a convolution workgroup)

    from alpha_zero_amd import _abi, _lib

    _lib._binding = _abi.Binding(ctypes.CDLL(os.environ["AZ_PROBE_LIB"]), "probe build")
torch.manual_seed(4)
net = AlphaZeroNet((17, 9, 9), 82, 2, 128, 64)
act = OverlapActor(net, game="go", board_size=9, num_games=1184, num_simulations=24, num_parallel=8, warm_up_steps=4, resign_threshold=-1.0, seed=7,
                   device="cuda", overlap=True, use_graph=False, engine_kw={"max_steps": 24})
act.engine.on_launch = None
e, inf = act.engine, act.infer
captured = {}
orig_blocks = inf._blocks_tiled


def blocks(a, m, o, B, S, C, st):
    captured[("stem", B)] = a.clone()
    out = orig_blocks(a, m, o, B, S, C, st)
    captured[("tower", B)] = out.clone()
    return out


inf._blocks_tiled = blocks


def snapshot(k):
    g0, g1 = act._halves[k]
    B = (g1 - g0) * e.P
    pol, val, _, _ = inf._head_buffers(B, inf.fc_wp.shape[1], inf.fc_w1.shape[1], e.features.device, 1 + k)
    return {"stem": captured[("stem", B)], "tower": captured[("tower", B)], "pol": pol.clone(), "val": val.clone(),
            "pri": e.priors[g0 * e.P:g1 * e.P].clone(), "v": e.values[g0 * e.P:g1 * e.P].clone()}


# PROBE_B_KERNELS = stem | tower | head: the second stream does not run a whole forward but only that kernel family, again and again
# (bisects which neighbour it takes); its results are then not compared (half 1 is skipped below)
B_KIND = os.environ.get("PROBE_B_KERNELS", "all")
if B_KIND != "all":
    import ctypes

    _orig_half = act._forward_half

    def _forward_half(k):
        if k == 0:
            return _orig_half(0)
        dll, st = act.binding.dll, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        g0, g1 = act._halves[1]
        r0, rows = g0 * e.P, (g1 - g0) * e.P
        tb = 3
        feat = e.features[(r0 // tb) * (32 * tb * 81):]
        a, m, o = inf._tiled_buffers(rows, 9, 128, feat.device, 2)
        k1, k2 = inf.fc_wp.shape[1], inf.fc_w1.shape[1]
        pol, val, _, _ = inf._head_buffers(rows, k1, k2, feat.device, 2)
        pri_b, v_b = e.priors[r0:r0 + rows], e.values[r0:r0 + rows]
        seq = B_KIND.split("+")  # e.g. "tower+head": the listed kernels one after the other, the whole list PROBE_B_REPS times
        for _ in range(int(os.environ.get("PROBE_B_REPS", "8"))):
            for kname in seq:
                if kname == "stem":
                    assert dll.azsp_stem_tiled(feat.data_ptr(), inf.stem_wp.data_ptr(), inf.stem_b32.data_ptr(), a.data_ptr(), rows, 9, 128, 1, 1, st) == 0
                elif kname == "tower":
                    assert dll.azsp_conv3x3_tiled(a.data_ptr(), inf.wp[0].data_ptr(), inf.b32[0].data_ptr(), None, m.data_ptr(), rows, 9, 128, 1, st) == 0
                elif kname == "towerres":
                    assert dll.azsp_conv3x3_tiled(m.data_ptr(), inf.wp[1].data_ptr(), inf.b32[1].data_ptr(), a.data_ptr(), o.data_ptr(), rows, 9, 128, 1, st) == 0
                elif kname == "head":
                    assert dll.azsp_head_tiled(a.data_ptr(), inf.head_w32.data_ptr(), inf.head_b32.data_ptr(), pol.data_ptr(), val.data_ptr(), rows, 9, 128,
                                               inf.npol, inf.nval, k1, k2, st) == 0
                elif kname == "fc":
                    assert dll.azsp_fc_heads(pol.data_ptr(), val.data_ptr(), inf.fc_wp.data_ptr(), inf.fc_bp.data_ptr(), k1 // 16, inf.fc_w1.data_ptr(),
                                             inf.fc_b1.data_ptr(), k2 // 16, inf.fc_w2.data_ptr(), ctypes.c_float(inf.fc_b2), pri_b.data_ptr(), v_b.data_ptr(),
                                             rows, inf.num_actions, inf.fc_width, st) == 0
                else:
                    raise SystemExit("unknown kernel " + kname)

    act._forward_half = _forward_half
main = torch.cuda.current_stream()
found = {}
for r in range(40):
    for g0, g1 in act._halves:
        e.expand_backup(g0, g1)
        e.select(g0, g1)
    for k in range(2):
        act._streams[k].wait_stream(main)
        with torch.cuda.stream(act._streams[k]):
            act._forward_half(k)
    for s in act._streams:
        main.wait_stream(s)
    torch.cuda.synchronize()
    halves = (0, 1) if B_KIND == "all" else (0,)
    conc = [snapshot(k) for k in halves]
    ser = []
    for k in halves:
        (act._forward_half if B_KIND == "all" else _orig_half)(k)
        torch.cuda.synchronize()
        ser.append(snapshot(k))
    for k in halves:
        for name in ("stem", "tower", "pol", "val", "pri", "v"):
            if not torch.equal(conc[k][name], ser[k][name]):
                d = (conc[k][name].float() - ser[k][name].float()).abs()
                nz = torch.nonzero(d.flatten() > 0).flatten()
                found.setdefault((k, name), []).append((r, int(nz.numel()), nz[:8].tolist(), float(d.max())))
                break
print("second stream runs:", B_KIND)
print("first differing stage per (half, stage) over 40 rounds:", {k: (len(v), v[:3]) for k, v in found.items()} or "none", flush=True)
# where the differing head-plane elements sit: (row of the head buffer = board, column = plane * P2 + position) -> head workgroup = flat position // 256
for (k, name), v in found.items():
    if name in ("pol", "val"):
        pol, val, _, _ = inf._head_buffers((act._halves[k][1] - act._halves[k][0]) * e.P, inf.fc_wp.shape[1], inf.fc_w1.shape[1], e.features.device, 1 + k)
        width = (pol if name == "pol" else val).shape[1]
        for r, cnt, idx, dmax in v[:6]:
            print(f"  half {k} {name} round {r}: {cnt} elements, first (board, column): {[(i // width, i % width) for i in idx]}, max |d| {dmax}")
dll = act.binding.dll
if hasattr(dll, "azsp_probe_head_dbg"):
    import ctypes

    out = (ctypes.c_uint32 * 4)()
    assert dll.azsp_probe_head_dbg(out, 0) == 0
    print("head probe record: LDS weight mismatches", out[0], " threads whose second read gave other planes", out[1], " threads checked", out[2], flush=True)
