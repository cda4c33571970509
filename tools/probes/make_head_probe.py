"""Probe builds for the concurrency question of DESIGN "Two forwards in flight": with two evaluator forwards on two streams, the head
planes k_head_tiled wrote for the forward launched first differed from a serial run in a few neighbouring positions, although its input
(the tower output) was identical.  Rounds 1-3 staged the head's 1x1 weights in an LDS table; round 4 found that this table is what it
takes (weights from global memory: exact) and removed it from the product.  This script rebuilds the OLD behaviour from patched COPIES of
the product sources (the product carries no switches) -> tools/probes/libazsp_head_<V>.so, loaded by tools/concurrency_probe2.py through
AZ_PROBE_LIB:
  LDSTABLE the round-3 kernel: weights staged in a dynamic-LDS table, read with wave-uniform ds_read2_b32
  CHECK    LDSTABLE + at its end the kernel verifies that (a) the table still equals the global weights and (b) recomputing the planes from
           a second read of the input with weights from global memory gives the same result; counted in a device record read by
           azsp_probe_head_dbg (probe-only export)
  PAD512   LDSTABLE + 512 B of static LDS (2048 B per head workgroup: two of them no longer fit beside a tower workgroup)
  PAD      LDSTABLE with 96 KB of extra static LDS (one head workgroup per CU): the round-3 observation that this hides the effect
The product build itself is the fourth variant (no AZ_PROBE_LIB)."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-strict-aliasing", "-Wno-unused-value"]

DBG_DECL = """__device__ unsigned g_head_dbg[4];
"""
CHECK_TAIL = """    {   // ---- probe: (a) LDS weight copy intact?  (b) does a second read of the input give the same planes?
        for (int k = threadIdx.x; k < NPL * C; k += 256)
            if (ws[k] != w[k]) atomicAdd(&g_head_dbg[0], 1u);
        float acc2[NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) acc2[pl] = bias[pl];
        for (int c = 0; c < nch; ++c) {
            const cv_u32x4 v = *(const volatile cv_u32x4*)(src + (size_t)c * tile_rows * 16);
            typedef CvFmt<F16> FM;
            const float f[8] = {FM::lo(v.x), FM::hi(v.x), FM::lo(v.y), FM::hi(v.y), FM::lo(v.z), FM::hi(v.z), FM::lo(v.w), FM::hi(v.w)};
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc2[pl] += f[e] * w[pl * C + c * 8 + e];
        }
        bool same = true;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) same = same && (acc2[pl] == acc[pl]);
        if (!same) atomicAdd(&g_head_dbg[1], 1u);
        atomicAdd(&g_head_dbg[2], 1u);
    }
"""
EXPORT = """
extern "C" int azsp_probe_head_dbg(unsigned* out4, int reset) {
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_head_dbg), 4 * sizeof(unsigned)) != hipSuccess) return -1;
    if (reset) {
        const unsigned z[4] = {0u, 0u, 0u, 0u};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_head_dbg), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
"""


def build(variant):
    bd = os.path.join("/tmp", "head_probe_" + variant)
    shutil.rmtree(bd, ignore_errors=True)
    shutil.copytree(os.path.join(ROOT, "alpha_zero_amd", "csrc"), os.path.join(bd, "alpha_zero_amd", "csrc"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(bd, "include"))
    p = os.path.join(bd, "alpha_zero_amd", "csrc", "az_conv.h")
    t = open(p).read()
    head0 = t.index("template <int NPL, bool F16 = false> __global__ void __launch_bounds__(256)\nk_head_tiled(")
    head1 = t.index("// Fully connected layers of both heads + softmax / tanh")
    body = t[head0:head1]
    stage = "    const float* __restrict__ ws = w;  // [NPL][C], wave-uniform indices below -> s_load\n"
    assert body.count(stage) == 1
    body = body.replace(stage, "    extern __shared__ float ws[];  // [NPL][C]\n    for (int i = threadIdx.x; i < NPL * C; i += 256) ws[i] = w[i];\n    __syncthreads();\n")
    if variant == "CHECK":
        marker = "#pragma unroll\n    for (int pl = 0; pl < NPL; ++pl) {\n        const float v = fmaxf(acc[pl], 0.0f);"
        assert body.count(marker) == 1
        body = body.replace(marker, CHECK_TAIL + marker)
        body = DBG_DECL + body
    elif variant == "PAD":
        marker = "    extern __shared__ float ws[];  // [NPL][C]\n"
        assert body.count(marker) == 1
        body = body.replace(marker, marker + "    __shared__ float pad_lds[24 * 1024];\n    if (C < 0) pad_lds[threadIdx.x] = 0.0f, ws[0] = pad_lds[threadIdx.x + 1];\n")
    elif variant == "PAD512":  # table + 512 B: two head workgroups no longer fit beside a 160 768-byte tower workgroup (LDS exactly full with 1536 B each)
        marker = "    extern __shared__ float ws[];  // [NPL][C]\n"
        assert body.count(marker) == 1
        body = body.replace(marker, marker + "    __shared__ float pad_lds[128];\n    if (C < 0) pad_lds[threadIdx.x & 127] = 0.0f, ws[0] = pad_lds[(threadIdx.x + 1) & 127];\n")
    else:
        assert variant == "LDSTABLE"
    open(p, "w").write(t[:head0] + body + t[head1:])
    hp = os.path.join(bd, "alpha_zero_amd", "csrc", "azsp_hip.hip")
    h = open(hp).read()
    assert h.count("(k_head_tiled<3, true>), dim3((unsigned)((npos + 255) / 256)), dim3(256), 0,") == 1 and h.count("(k_head_tiled<3>), dim3((unsigned)((npos + 255) / 256)), dim3(256), 0,") == 1
    h = h.replace("(k_head_tiled<3, true>), dim3((unsigned)((npos + 255) / 256)), dim3(256), 0,", "(k_head_tiled<3, true>), dim3((unsigned)((npos + 255) / 256)), dim3(256), 3 * C * sizeof(float),")
    h = h.replace("(k_head_tiled<3>), dim3((unsigned)((npos + 255) / 256)), dim3(256), 0,", "(k_head_tiled<3>), dim3((unsigned)((npos + 255) / 256)), dim3(256), 3 * C * sizeof(float),")
    if variant == "CHECK":
        h += EXPORT
    open(hp, "w").write(h)
    out = os.path.join(HERE, f"libazsp_head_{variant}.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-o", out, hp])
    return out


if __name__ == "__main__":
    variants = sys.argv[1:] or ["LDSTABLE", "CHECK", "PAD"]
    with ThreadPoolExecutor(3) as ex:
        for o in ex.map(build, variants):
            print("built", o)
