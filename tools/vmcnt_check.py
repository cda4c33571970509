"""the split kernels wait for the next tile's LDS-DMA pieces with a COUNTED `s_waitcnt vmcnt(N)` (N derived at compile time from
the schedule).  This compares, bit for bit, the product library with a build whose waits are all vmcnt(0)
(tools/probes/make_sp_abl.py VMCNT0 -> tools/probes/libazsp_abl_VMCNT0.so): a piece still in flight at a barrier would show up as a
mismatch.  Prints one line per case and a verdict; run on the GPU box."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from alpha_zero_amd import _abi, _lib
from alpha_zero_amd.core.network import split_weights_f16

prod = _lib.load()
ref = _abi.Binding(ctypes.CDLL(os.path.join(ROOT, "tools", "probes", "libazsp_abl_VMCNT0.so")), "vmcnt(0) build")
bad = 0
for S, C in ((9, 128), (9, 64), (17, 64)):
    for boards in (1, 2, 129, 257, 1000, 4357):
        g = torch.Generator().manual_seed(S * 1000 + C + boards)
        x = torch.randn(boards, C, S, S, generator=g).abs().cuda().contiguous(memory_format=torch.channels_last)
        r = torch.randn(boards, C, S, S, generator=g).abs().cuda().contiguous(memory_format=torch.channels_last)
        w = split_weights_f16(torch.randn(C, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5).cuda()
        b = (torch.randn(C, generator=g) * 0.1).cuda()
        n = prod.dll.azsp_split_bytes(boards, S, C) // 2
        xs, rs = torch.zeros(n, dtype=torch.float16, device="cuda"), torch.zeros(n, dtype=torch.float16, device="cuda")
        assert prod.dll.azsp_split_layout(x.data_ptr(), xs.data_ptr(), boards, S, C, 1, None, None) == 0
        assert prod.dll.azsp_split_layout(r.data_ptr(), rs.data_ptr(), boards, S, C, 1, None, None) == 0
        for res in (None, rs):
            outs = []
            for lib in (prod, ref):
                for rep in range(3):  # the race, if any, is timing dependent: a few repetitions each
                    y = torch.zeros(n, dtype=torch.float16, device="cuda")
                    assert lib.dll.azsp_conv3x3_split(xs.data_ptr(), w.data_ptr(), b.data_ptr(), None if res is None else res.data_ptr(), y.data_ptr(),
                                                      boards, S, C, 1, None, None) == 0
                    outs.append(y)
            torch.cuda.synchronize()
            same = all(torch.equal(outs[0], o) for o in outs[1:])
            bad += not same
            print(f"S={S} C={C} boards={boards} residual={res is not None}: {'identical' if same else 'MISMATCH'}")
print("VERDICT:", "all bit-identical (counted vmcnt == vmcnt(0))" if bad == 0 else f"{bad} cases differ")
sys.exit(1 if bad else 0)
