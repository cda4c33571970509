"""The fp32-class tower kernels request their B fragments one ds_read_b128 per MFMA gap (az_conv_sp.h `load_frag`): with one wave per
SIMD a burst of reads in front of a k-step leaves the matrix pipe without an instruction to issue (measured: 3 - 6 % of a launch).  This
test disassembles the built library (llvm-objdump, no GPU) and holds the property: between two consecutive MFMAs of a tower kernel there
are at most two fragment reads, except the one exposed re-start of the fused block (the m image's first fragments behind its barrier)."""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def _reads_per_mfma_gap(pattern):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_resources import code_object

    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(code_object(os.path.join(ROOT, "alpha_zero_amd", "libazsp.so")))
        f.flush()
        txt = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", f.name], capture_output=True, text=True, check=True).stdout
    out = {}
    for blk in re.split(r"\n(?=[0-9a-f]{16} <)", txt):
        head = blk.split("\n", 1)[0]
        if pattern not in head or ">:" not in head:
            continue
        reads, g, first = [], 0, True
        for ln in blk.split("\n")[1:]:
            m = re.match(r"\s+(\S+)", ln)
            if not m:
                continue
            if m.group(1).startswith("v_mfma"):
                if not first:  # (the gap in front of the first MFMA is the kernel's prologue)
                    reads.append(g)
                first, g = False, 0
            elif m.group(1) == "ds_read_b128":
                g += 1
        out[head] = reads
    return out


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="llvm-objdump of the ROCm image")
@pytest.mark.parametrize("pattern,exposed", [("k_conv3x3_sp2I", 0), ("k_resblock_spI", 2), ("k_conv3x3_sp17ILb1ELi8", 0), ("k_conv3x3_sp17ILb0ELi8", 0)])
def test_fragment_reads_sit_in_mfma_gaps_of_their_own(pattern, exposed):
    kernels = _reads_per_mfma_gap(pattern)
    assert kernels, pattern
    for name, reads in kernels.items():
        hist = Counter(reads)
        bursts = sum(n for k, n in hist.items() if k > 2)
        assert len(reads) > 250 and bursts <= exposed, (name, sorted(hist.items()))
        assert hist[1] >= 0.25 * len(reads), (name, sorted(hist.items()))  # the reads are there, one per gap
