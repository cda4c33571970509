"""Ablation builds of the fused split-precision ResNetBlock kernel (az_resblock_sp17.h): copies of the product sources with ONE part of
the kernel removed per build (results are wrong, timings tell what each part costs).  The product header carries no switches: this
script patches copies under /tmp and compiles them to tools/probes/libazsp_abl_RB_<VARIANT>.so, which tools/rb_bench.py times through
AZ_BENCH_LIB.  Variants: FULL (unpatched), NO_DMA (the next tile's LDS-DMA pieces are not issued), NO_VMWAIT (no s_waitcnt vmcnt(0)
in front of a tile's second barrier), NO_BAR (the two barriers of a tile are removed: waves run free), NO_STORE (no y stores, no skip
loads: the compiler drops phase B's epilogue arithmetic as dead code), NO_MWRITE (phase A's epilogue does not write the m image),
NO_FRAG (no B-fragment reads after the first k-steps), NO_EXPOSED (phase A's last unit skips its exposed epilogue)."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-strict-aliasing", "-Wno-unused-value"]


def patch(text, variant):
    def rep(old, new, count=1):
        nonlocal text
        assert text.count(old) >= 1, (variant, old[:70])
        text = text.replace(old, new) if count == 0 else text.replace(old, new, count)

    if variant == "NO_DMA":
        rep("        const unsigned long long mask = live ? dmask[h][pc] : 0ull;", "        const unsigned long long mask = 0ull;")
    elif variant == "NO_VMWAIT":
        rep('                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n                    CV_BARRIER();', "                    CV_BARRIER();")
    elif variant == "NO_BAR":
        rep('                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n                    CV_BARRIER();', "")
        rep("                CV_BARRIER();\n#pragma unroll\n                for (int s = 0; s < R - 1; ++s) load_step(CpInt<1>{}, Ms, NROT, nnj, s, s);",
            "#pragma unroll\n                for (int s = 0; s < R - 1; ++s) load_step(CpInt<1>{}, Ms, NROT, nnj, s, s);")
    elif variant == "NO_STORE":
        rep("if (store_ok) *(cv_u32x2*)(out + ooff)", "if (false) *(cv_u32x2*)(out + ooff)")
        rep("} else if (store_ok) *(__attribute__((address_space(1))) cv_u32x2*)(out_lo + ooff)", "} else if (false) *(__attribute__((address_space(1))) cv_u32x2*)(out_lo + ooff)")
        rep("                        else rr[set][rj][rp] = *(const cv_u32x2*)(Xs + (lm[ROT][rj].x + skc)", "                        else if (false) rr[set][rj][rp] = *(const cv_u32x2*)(Xs + (lm[ROT][rj].x + skc)")
        rep("                        rrx[rj][rp] = *(const cv_u32x2*)", "                        if (false) rrx[rj][rp] = *(const cv_u32x2*)")
    elif variant == "NO_MWRITE":
        rep("if (o == 2 * PAIR) *(cv_u32x2*)(Mw + ooff) = (cv_u32x2){hpk[0], hpk[1]};", "if (false) *(cv_u32x2*)(Mw + ooff) = (cv_u32x2){hpk[0], hpk[1]};")
        rep("else *(cv_u32x2*)(Mw + MPLANE + ooff) = (cv_u32x2){lpk[0], lpk[1]};", "else if (false) *(cv_u32x2*)(Mw + MPLANE + ooff) = (cv_u32x2){lpk[0], lpk[1]};")
    elif variant == "NO_FRAG":
        rep("        const int tap = s / KSUB;\n        const int off =", "        if (it > 0) return;\n        const int tap = s / KSUB;\n        const int off =", 0)  # load_step AND load_frag
    elif variant == "NO_EXPOSED":
        rep("                    for (int o = 0; o < CTM; ++o) epi_op(CpInt<0>{}, set, j, lm[ROT][j].y, nullptr, (sb17_gptr)0, o, true);",
            "                    for (int o = 0; o < 0; ++o) epi_op(CpInt<0>{}, set, j, lm[ROT][j].y, nullptr, (sb17_gptr)0, o, true);")
    elif variant.startswith("DMA_Q"):  # NOT an ablation: the next tile's DMA piece of a k-step is issued behind MFMA slot Q instead of behind the last
        q = int(variant[5:])
        rep("                if constexpr (PH == 1 && FIRST && t >= 1 && t - 1 < NPIECE) {\n", "                if constexpr (false) {\n")
        rep("                    __builtin_amdgcn_sched_barrier(0);\n                }, typename CpMakeSeq<NQ>::type{});",
            "                    if constexpr (PH == 1 && FIRST && t >= 1 && t - 1 < NPIECE && q == %d) {\n"
            "                        if constexpr (H == 0) dma_piece(xb, true, 1, t - 1);\n"
            "                        else dma_piece(xnb, has_next, 0, t - 1);\n"
            "                    }\n"
            "                    __builtin_amdgcn_sched_barrier(0);\n                }, typename CpMakeSeq<NQ>::type{});" % q)
    elif variant == "TWO_PER_STEP":  # two DMA pieces per k-step in k-steps 1 .. 8: the x image is complete half a unit earlier
        rep("                if constexpr (PH == 1 && FIRST && t >= 1 && t - 1 < NPIECE) {\n                    // the tile after this one: the lower half of this board, or the upper half of this workgroup's next board\n"
            "                    if constexpr (H == 0) dma_piece(xb, true, 1, t - 1);\n                    else dma_piece(xnb, has_next, 0, t - 1);\n",
            "                if constexpr (PH == 1 && FIRST && t >= 1 && 2 * (t - 1) < NPIECE) {\n"
            "                    if constexpr (H == 0) dma_piece(xb, true, 1, 2 * (t - 1)), dma_piece(xb, true, 1, 2 * (t - 1) + 1);\n"
            "                    else dma_piece(xnb, has_next, 0, 2 * (t - 1)), dma_piece(xnb, has_next, 0, 2 * (t - 1) + 1);\n")
    elif variant != "FULL":
        raise SystemExit("unknown variant " + variant)
    return text


def build(variant):
    bd = os.path.join("/tmp", "rb_abl_" + variant)
    shutil.rmtree(bd, ignore_errors=True)
    shutil.copytree(os.path.join(ROOT, "alpha_zero_amd", "csrc"), os.path.join(bd, "alpha_zero_amd", "csrc"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(bd, "include"))
    p = os.path.join(bd, "alpha_zero_amd", "csrc", "az_resblock_sp17.h")
    src = open(p).read()
    if variant == "NO_FRAG":  # `it` must be visible to load_step: declare the board counter before the lambdas
        src = src.replace("    int it = 0;\n    unsigned char* yprev = y;", "    unsigned char* yprev = y;").replace(
            "    sp_f16x8 bb[R][2][NJM];", "    int it = 0;\n    sp_f16x8 bb[R][2][NJM];")
    open(p, "w").write(patch(src, variant))
    out = os.path.join(HERE, f"libazsp_abl_RB_{variant}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, os.path.join(bd, "alpha_zero_amd", "csrc", "azsp_hip.hip")])
    return out


if __name__ == "__main__":
    variants = sys.argv[1:] or ["FULL", "NO_DMA", "NO_VMWAIT", "NO_BAR", "NO_STORE", "NO_MWRITE", "NO_FRAG", "NO_EXPOSED"]
    with ThreadPoolExecutor(8) as ex:
        for o in ex.map(build, variants):
            print("built", o)
