"""Digest of the HIP sources a tower-kernel family is compiled from.  A PMC summary under profiles/ records the digest of the
sources it was measured on; bench.py compares it with the digest of the tree it runs from and labels a `traffic` figure measured on
an older version of the kernel as STALE instead of presenting it as evidence for the current binary (Weak #3)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "alpha_zero_amd", "csrc")

# family (tools/pmc_launches.py) -> the headers whose text decides the kernel's code (the shared MFMA / epilogue helpers included)
FAMILY_SOURCES = {
    "split9": ["az_conv_sp2.h", "az_conv_sp.h", "az_conv.h"],
    "split9_64": ["az_conv_sp.h", "az_conv.h"],
    "splitblock9_64": ["az_resblock_sp17.h", "az_conv_sp17.h", "az_conv_sp.h", "az_conv.h"],
    "split17": ["az_conv_sp17.h", "az_conv_sp.h", "az_conv.h"],
    "splitblock17": ["az_resblock_sp17.h", "az_conv_sp17.h", "az_conv_sp.h", "az_conv.h"],
    "tiled9": ["az_conv.h"],
    "hb19": ["az_conv19.h", "az_conv.h"],
    "spg19": ["az_conv_spg.h", "az_conv_sp.h", "az_conv.h"],
}
# profiles/<file> -> family
PMC_FILE_FAMILY = {
    "split_kernel_pmc.json": "split9",
    "split17_kernel_pmc.json": "split17",
    "splitblock17_kernel_pmc.json": "splitblock17",
    "splitblock9_64_kernel_pmc.json": "splitblock9_64",
    "conv_kernel_pmc.json": "tiled9",
    "conv19_kernel_pmc.json": "hb19",
    "spg19_kernel_pmc.json": "spg19",
}


def kernel_source_digest(family):
    """sha256 over the family's header files (name + content), hex; headers that do not exist (yet) are skipped."""
    h = hashlib.sha256()
    for name in FAMILY_SOURCES[family]:
        p = os.path.join(CSRC, name)
        if os.path.exists(p):
            h.update(name.encode() + b"\0" + open(p, "rb").read() + b"\0")
    return h.hexdigest()


if __name__ == "__main__":
    import sys

    for fam in sys.argv[1:] or sorted(FAMILY_SOURCES):
        print(fam, kernel_source_digest(fam))
