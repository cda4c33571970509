"""Instruction mix of one device kernel of libazsp.so: opcode counts and the histogram of non-MFMA instructions issued between two
consecutive MFMAs (how well epilogue micro-ops / loads are spread over the matrix stream).  usage: python tools/kernel_gaps.py <mangled-name-substring>"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_resources import ROOT, code_object  # noqa: E402

pat = sys.argv[1]
with tempfile.NamedTemporaryFile(suffix=".co") as f:
    f.write(code_object(os.path.join(ROOT, "alpha_zero_amd", "libazsp.so")))
    f.flush()
    txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True).stdout
blocks = re.split(r"\n(?=[0-9a-f]{16} <)", txt)
for blk in blocks:
    head = blk.split("\n", 1)[0]
    if pat not in head or ">:" not in head:
        continue
    ops = []
    for ln in blk.split("\n")[1:]:
        m = re.match(r"\s+(\S+)", ln)
        if m:
            ops.append(m.group(1))
    c = Counter(ops)
    print(head[:100])
    print(" instructions:", len(ops), " mfma:", sum(v for k, v in c.items() if k.startswith("v_mfma")))
    print(" top:", ", ".join(f"{k} {v}" for k, v in c.most_common(22)))
    gaps, g = [], 0
    for o in ops:
        if o.startswith("v_mfma"):
            gaps.append(g)
            g = 0
        elif not o.startswith("s_nop"):
            g += 1
    print(" non-MFMA instructions per MFMA gap (count: gaps):", sorted(Counter(gaps).items()))
