# Samples rocm-smi clocks / power while the conv kernel runs back to back (documents the power-limited clock of DESIGN.md).
R=$GRAFT_REPO_ROOT
python - <<'PY' &
import ctypes, os, sys, time, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
from alpha_zero_amd import _lib
b = _lib.load()
B, C, S = 32768, 128, 9
n = b.dll.azsp_tiled_bytes(B, S, C) // 2
scale = float(os.environ.get("SCALE", "1"))
x = (scale * torch.randn(n)).to(torch.bfloat16).cuda(); y = torch.zeros(n, dtype=torch.bfloat16, device="cuda")
w = (torch.randn(9, C, C) * 0.05).to(torch.bfloat16).cuda(); bias = torch.randn(C).cuda()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
t0 = time.time()
while time.time() - t0 < 12:
    for _ in range(200):
        b.dll.azsp_conv3x3_tiled(x.data_ptr(), w.data_ptr(), bias.data_ptr(), None, y.data_ptr(), B, S, C, 1, st)
    torch.cuda.synchronize()
PY
sleep 6
for i in 1 2 3; do /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "sclk\|power\|mclk\|junction" | head -8; sleep 1.5; done
wait
