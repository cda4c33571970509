"""Ablation builds of k_conv3x3_op19q (az_conv19.h: the bf16 19x19 x 256 tower convolution, BASELINE C5's dominant kernel): copies of the
product sources with ONE part of the kernel removed per build (results are wrong, timings tell what the part costs), compiled to
tools/probes/libazsp_abl19_<VARIANT>.so; tools/split_prev_ab.py times them against the product (PREV_AB_CASES=19x19).
Variants: NO_XW (no hand-over writes), NO_XR (no hand-over reads / adds), NO_X (neither), NO_DMA (the next bands' LDS-DMA pieces are not
issued), NO_B (barrier B removed, its counted vmcnt wait kept), NO_F (barrier F removed: the product's own -DC1Q_ABLATE_NO_F switch)."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-strict-aliasing", "-Wno-unused-value"]


def patch(text, variant):
    head, tail = text[:text.index("template <bool ADD> struct C1QSched {")], text[text.index("template <bool ADD> struct C1QSched {"):]

    def rep(old, new):
        nonlocal tail
        assert tail.count(old) == 1, old[:60]
        tail = tail.replace(old, new)

    if variant in ("NO_XW", "NO_X"):
        rep("    auto xwrite = [&](int set, int o) {\n", "    auto xwrite = [&](int set, int o) {\n        return;\n")
    if variant in ("NO_XR", "NO_X"):
        rep("        if (o < 30) {\n", "        if (o < 30) return;\n        if (false) {\n")
    if variant == "NO_DMA":
        rep("                if constexpr (SC::dma_unit(u) && t >= SC::dma_slot(0)", "                if constexpr (false && SC::dma_unit(u) && t >= SC::dma_slot(0)")
    if variant == "NO_B":
        rep("                    CV_BARRIER();  // B:", "                    // B:")
    return head + tail


def build(variant):
    bd = os.path.join("/tmp", "op19q_abl_" + variant)
    shutil.rmtree(bd, ignore_errors=True)
    shutil.copytree(os.path.join(ROOT, "alpha_zero_amd", "csrc"), os.path.join(bd, "alpha_zero_amd", "csrc"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(bd, "include"))
    extra = []
    if variant == "NO_F":
        extra = ["-DC1Q_ABLATE_NO_F"]
    else:
        p = os.path.join(bd, "alpha_zero_amd", "csrc", "az_conv19.h")
        src = open(p).read()
        open(p, "w").write(patch(src, variant))
    out = os.path.join(HERE, f"libazsp_abl19_{variant}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-o", out, os.path.join(bd, "alpha_zero_amd", "csrc", "azsp_hip.hip")])
    return out


if __name__ == "__main__":
    variants = sys.argv[1:] or ["NO_X", "NO_XR", "NO_DMA", "NO_B"]
    with ThreadPoolExecutor(4) as ex:
        for o in ex.map(build, variants):
            print("built", o)
