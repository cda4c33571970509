"""A/B of the evaluator forward with azsp_fc_heads vs the library GEMMs for the head FC layers (same process, same box)."""
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
import engine_util as eu
from alpha_zero_amd import _lib
from alpha_zero_amd.core.network import AlphaZeroNet, InferenceNet
torch.manual_seed(0)
net = AlphaZeroNet((17, 9, 9), 82, 10, 128, 128)
inf = InferenceNet(net, dtype=torch.bfloat16, binding=_lib.load()).cuda()
rows = 32768
feat = eu.tile_features((torch.rand(rows, 17, 9, 9) > 0.6).float()).cuda()
for fused in (True, False, True, False):
    inf.use_fused_fc = fused
    for _ in range(3):
        inf.forward_tiled(feat, rows, 9)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        inf.forward_tiled(feat, rows, 9)
    e1.record()
    torch.cuda.synchronize()
    print("fused_fc" if fused else "library ", round(e0.elapsed_time(e1) / 20, 4), "ms per forward")
