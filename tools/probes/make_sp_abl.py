"""Ablation builds of the split-precision tower kernels (az_conv_sp.h, az_conv_sp17.h): copies of the product sources with ONE part of
the kernels removed per build (results are wrong, timings tell what each part costs).  The product headers carry no switches: this
script patches copies under a build directory and compiles them to tools/probes/libazsp_abl_<VARIANT>.so, which tools/split_bench.py
times through AZ_BENCH_LIB.  Variants: FULL (unpatched), HALF_FRAG (B fragments of every second k-step are not read from LDS: the ring
slot keeps its old contents), NO_FRAG (no fragment reads after the first k-steps), NO_DMA (the next tile's LDS-DMA pieces are not
issued), VMCNT0 (NOT an ablation: every counted s_waitcnt vmcnt(N) before a tile barrier replaced by vmcnt(0) -- tools/vmcnt_check.py compares
its outputs bit for bit with the product's), NO_CORNER (9x9: no corner phase), NO_STORE (no output stores and no residual loads; the compiler then drops the whole epilogue arithmetic as dead code)."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-strict-aliasing", "-Wno-unused-value"]


def patch(text, variant, name):
    def rep(old, new, count=1):
        nonlocal text
        assert text.count(old) >= 1, (name, old[:60])
        text = text.replace(old, new) if count == 0 else text.replace(old, new, count)

    if variant in ("HALF_FRAG", "NO_FRAG"):
        cond = "(s & 1) == 0" if variant == "HALF_FRAG" else "false"
        # (every occurrence: load_step of the prologue AND load_frag, the per-gap requests inside the units since the end of round 6)
        rep("        const int tap = s / KSUB;\n        const int off =", "        if (!(%s)) return;\n        const int tap = s / KSUB;\n        const int off =" % cond, 0)
    elif variant == "NO_DMA":
        rep("        const unsigned long long mask = live ? dmask", "        live = false;\n        const unsigned long long mask = live ? dmask")
    elif variant.startswith("SPREAD"):  # the next tile's DMA pieces every STR-th k-step across the units instead of one per k-step in unit 0
        STR = int(variant[6:])
        if name == "az_conv_sp.h":
            rep("                if constexpr (i == 0 && t >= 1 && t - 1 < NPIECE) dma_piece(nsrc, ndst, has_next, t - 1);",
                "                if constexpr (g >= 1 && (g - 1) %% %d == 0 && (g - 1) / %d < NPIECE) dma_piece(nsrc, ndst, has_next, (g - 1) / %d);" % (STR, STR, STR))
            rep('if (have_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_YOUNGER) : "memory");', 'if (false) asm volatile("s_nop 0");')
        else:
            rep("                if constexpr (U == 0 && t >= 1 && t - 1 < NPIECE) dma_piece(nsrc, ndst, nlive, H ^ 1, t - 1);",
                "                if constexpr (g >= 1 && (g - 1) %% %d == 0 && (g - 1) / %d < NPIECE) dma_piece(nsrc, ndst, nlive, H ^ 1, (g - 1) / %d);" % (STR, STR, STR))
            rep('if (H == 0 && !have_prev) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");', 'if (true) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");')
    elif variant == "NO_CORNER":  # 9x9 kernel only: the corner phase (position (8, 0) of the last <= 16 boards) is skipped
        if name == "az_conv_sp.h":
            rep("        if ((it & 15) == 15 || !has_next) {", "        if (false) {")
    elif variant == "VMCNT0":  # every counted s_waitcnt vmcnt(N) in front of a tile barrier becomes vmcnt(0): results must stay bit-identical
        if name == "az_conv_sp.h":
            rep('if (have_prev) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_YOUNGER) : "memory");', 'if (false) asm volatile("s_nop 0");')
            rep('else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // first board', 'asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // first board')
        else:
            rep('else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(VM_YOUNGER) : "memory");', 'else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");')
    elif variant == "NO_STORE":
        rep("if (store_ok) *(cv_u32x2*)", "if (false) *(cv_u32x2*)", 0)
        rep("} else if (store_ok) *(__attribute__((address_space(1))) cv_u32x2*)", "} else if (false) *(__attribute__((address_space(1))) cv_u32x2*)", 0)
        rep("                        if constexpr (rp == 0) rr[set][rj][rp] = *(const cv_u32x2*)", "                        if constexpr (false) rr[set][rj][rp] = *(const cv_u32x2*)")
        rep("                        else rr[set][rj][rp] = *(const __attribute__((address_space(1))) cv_u32x2*)",
            "                        else if constexpr (false) rr[set][rj][rp] = *(const __attribute__((address_space(1))) cv_u32x2*)")
    return text


def build(variant):
    bd = os.path.join("/tmp", "sp_abl_" + variant)
    shutil.rmtree(bd, ignore_errors=True)
    shutil.copytree(os.path.join(ROOT, "alpha_zero_amd", "csrc"), os.path.join(bd, "alpha_zero_amd", "csrc"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(bd, "include"))
    if variant != "FULL":
        for name in ("az_conv_sp.h", "az_conv_sp17.h"):
            p = os.path.join(bd, "alpha_zero_amd", "csrc", name)
            src = open(p).read()
            open(p, "w").write(patch(src, variant, name))
    out = os.path.join(HERE, f"libazsp_abl_{variant}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, os.path.join(bd, "alpha_zero_amd", "csrc", "azsp_hip.hip")])
    return out


if __name__ == "__main__":
    variants = sys.argv[1:] or ["HALF_FRAG", "NO_FRAG", "NO_DMA", "NO_STORE"]
    with ThreadPoolExecutor(4) as ex:
        for o in ex.map(build, variants):
            print("built", o)
