"""Turn the text a tools/profile_r06.sh PMC run wrote for one kernel family (sections `== kernel-trace`, `== <counters>` with per-kernel
means) into the summary bench.py reads from profiles/: HBM bytes per launch (FETCH_SIZE x 2: gfx950 tallies a 128-B request as 64 B,
profiles/r01_pmc_calibration.txt; WRITE_SIZE x 1), matrix-pipe busy, cycles per MFMA, effective clock, wave-cycle breakdown -- and the
digest of the kernel sources the pass ran on (tools/kernel_digest.py), so that a later edit of the kernel makes the figures show as stale.
usage: python tools/pmc_to_json.py <family> <pmc_txt> <out_json> [rows]"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_digest import kernel_source_digest  # noqa: E402

GEOM = {"split9": (9, 9, 128, 4, 2.5), "split9_64": (9, 9, 64, 4, 2.5), "splitblock9_64": (9, 9, 64, 4, 2.0), "split17": (13, 17, 64, 4, 2.5),
        "splitblock17": (13, 17, 64, 4, 2.0), "tiled9": (9, 9, 128, 2, 2.5), "hb19": (19, 19, 256, 2, 2.5), "spg19": (19, 19, 256, 4, 2.5)}  # board, planes, channels, bytes/elem, passes


def parse(txt):
    sec, out, trace = None, {}, []
    for ln in txt.splitlines():
        if ln.startswith("== "):
            sec = ln[3:].strip()
            continue
        if sec == "kernel-trace":
            m = re.match(r"\s*(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)\s+(\S.*)$", ln)
            if m:
                trace.append({"calls": int(m.group(1)), "avg_us": float(m.group(3)), "name": m.group(5).strip()})
                continue
        m = re.match(r"\s+(\S.*?)\s+([A-Z][A-Za-z_0-9]+)\s+n=(\d+)\s+mean=([\d.e+-]+)", ln)  # (demangled kernel names contain blanks)
        if m:
            out.setdefault(m.group(1).strip(), {})[m.group(2)] = float(m.group(4))
    return trace, out


def short_name(n):
    """`k_conv3x3_sp2`, `k_resblock_sp<Sb17>` ... from a mangled or demangled kernel name: the label names what the pass MEASURED."""
    m = re.match(r"_Z(\d+)", n)
    base = n[m.end():m.end() + int(m.group(1))] if m else re.split(r"[<(]", n.replace("void ", ""))[0].strip()
    g = re.search(r"Sb\d+", n)
    return base + (f"<{g.group(0)}>" if g else "")


def main():
    fam, src, dst = sys.argv[1], sys.argv[2], sys.argv[3]
    rows = int(sys.argv[4]) if len(sys.argv) > 4 else (4096 if fam == "hb19" else 1024 if fam == "spg19" else 32768)
    board, planes, ch, elem, passes = GEOM[fam]
    trace, ctr = parse(open(src).read())
    kernels = {}
    for kname, c in ctr.items():
        t = [r for r in trace if r["name"].startswith(kname[:60])]
        us = t[0]["avg_us"] if t else None
        k = {"launch_us_in_the_trace_pass": us}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            k["FETCH_SIZE_KiB"], k["WRITE_SIZE_KiB"] = c["FETCH_SIZE"], c["WRITE_SIZE"]
            k["hbm_bytes_per_launch"] = round((2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)
        if "SQ_INSTS_MFMA" in c and "GRBM_GUI_ACTIVE" in c:
            cyc = c["GRBM_GUI_ACTIVE"] / 8.0  # the counter sums the 8 XCDs
            k["SQ_INSTS_MFMA"], k["gpu_cycles_per_launch"] = c["SQ_INSTS_MFMA"], cyc
            k["cycles_per_mfma"] = round(cyc / (c["SQ_INSTS_MFMA"] / 1024.0), 3)  # 1024 SIMDs, one wave each
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
                k["mfma_busy_fraction"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0), 4)
            if us:
                k["effective_clock_GHz"] = round(cyc / (us * 1e3), 4)
        for n in ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_INSTS_VALU", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "TCC_HIT_sum",
                  "TCC_MISS_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"):
            if n in c:
                k[n] = c[n]
        if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            k["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        if "SQ_LDS_BANK_CONFLICT" in c and c.get("SQ_LDS_IDX_ACTIVE"):
            k["lds_bank_conflict_fraction"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
        kernels[kname] = k
    js = json.load(open(dst)) if os.path.exists(dst) else {}
    if any("op19" in n for n in kernels):
        js["scheme"] = "one_pass (k_conv3x3_op19, round 6)"
        for k in [k for k in js if k.startswith(("hbm_bytes_per_convolution", "algorithmic_bytes_two", "ratio_to_two", "ratio_to_one_pass", "hbm_bytes_per_half", "mfma_busy_fraction_round4", "lds_bank_conflict_fraction_round4"))]:
            js.setdefault("earlier_rounds", {})[k] = js.pop(k)
    hb = [k["hbm_bytes_per_launch"] for k in kernels.values() if "hbm_bytes_per_launch" in k]
    alg = rows * planes * planes * ch * elem * passes
    js.update({"rows": rows, "board": board, "planes": planes, "channels": ch, "family": fam,
               "kernel_source_sha256": kernel_source_digest(fam), "measured_in_round": 6, "per_kernel": kernels})
    if hb:
        per_call = sum(hb) if fam == "hb19" and len(hb) > 1 and "one_pass" not in js.get("scheme", "") else sum(hb) / len(hb)
        js["hbm_bytes_per_launch"] = round(per_call)
        js["algorithmic_bytes_mean_layer"] = round(alg)
        js["ratio_to_algorithmic"] = round(per_call / alg, 4)
    cm = [k["cycles_per_mfma"] for k in kernels.values() if "cycles_per_mfma" in k]
    if cm:
        js["cycles_per_mfma"] = round(sum(cm) / len(cm), 3)
        js["effective_clock_GHz_mean"] = round(sum(k["effective_clock_GHz"] for k in kernels.values() if "effective_clock_GHz" in k) / max(1, len(cm)), 4)
        js["mfma_busy_fraction_mean"] = round(sum(k.get("mfma_busy_fraction", 0) for k in kernels.values()) / len(cm), 4)
        js["gpu_cycles_per_launch_mean"] = round(sum(k["gpu_cycles_per_launch"] for k in kernels.values() if "gpu_cycles_per_launch" in k) / len(cm), 1)
    # kernels templated on <RES, ...>: the plain (ILb0) and the residual (ILb1) instantiation, in the summary's long-standing keys
    pl = [k for n, k in kernels.items() if "ILb0E" in n or "<false" in n]
    rs = [k for n, k in kernels.items() if "ILb1E" in n or "<true" in n]
    if len(pl) == 1 and len(rs) == 1 and "hbm_bytes_per_launch" in pl[0] and "hbm_bytes_per_launch" in rs[0]:
        js["hbm_bytes_per_launch_plain"], js["hbm_bytes_per_launch_residual"] = pl[0]["hbm_bytes_per_launch"], rs[0]["hbm_bytes_per_launch"]
        js["FETCH_SIZE_KiB"] = {"plain": pl[0]["FETCH_SIZE_KiB"], "residual": rs[0]["FETCH_SIZE_KiB"]}
        js["WRITE_SIZE_KiB"] = {"plain": pl[0]["WRITE_SIZE_KiB"], "residual": rs[0]["WRITE_SIZE_KiB"]}
        if "mfma_busy_fraction" in pl[0]:
            js["mfma_busy_fraction"] = {"plain": pl[0]["mfma_busy_fraction"], "residual": rs[0]["mfma_busy_fraction"]}
            js["SQ_INSTS_MFMA"] = pl[0]["SQ_INSTS_MFMA"]
        if "lds_bank_conflict_fraction" in pl[0]:
            js["lds_bank_conflict_fraction"] = pl[0]["lds_bank_conflict_fraction"]
    if len(kernels) == 1:  # a one-kernel family (the fused blocks): the summary's long-standing scalar keys
        k1 = list(kernels.values())[0]
        if "hbm_bytes_per_launch" in k1:
            js["FETCH_SIZE_KiB"], js["WRITE_SIZE_KiB"] = k1["FETCH_SIZE_KiB"], k1["WRITE_SIZE_KiB"]
            tensor = rows * planes * planes * ch * elem
            js["read_ratio_to_x"], js["write_ratio_to_y"] = round(2.0 * k1["FETCH_SIZE_KiB"] * 1024 / tensor, 4), round(k1["WRITE_SIZE_KiB"] * 1024 / tensor, 4)
        for key in ("mfma_busy_fraction", "effective_clock_GHz", "lds_bank_conflict_fraction", "SQ_INSTS_MFMA"):
            if key in k1:
                if isinstance(js.get(key), dict):
                    js.setdefault("earlier_rounds", {})[key] = js[key]
                js[key] = k1[key]
        for key in [k for k in js if k.startswith("wave_cycle_breakdown_")]:
            js.setdefault("earlier_rounds", {})[key] = js.pop(key)
    for stale in ("round5_counters",):  # superseded by this round's pass (the files they came from stay under profiles/)
        if stale in js:
            js.setdefault("earlier_rounds", {})[stale] = js.pop(stale)
    if "source" in js:
        js.setdefault("earlier_rounds", {})["source"] = js.pop("source")
    if "note" in js:
        js.setdefault("earlier_rounds", {})["note"] = js.pop("note")
    js["kernel"] = " / ".join(sorted({short_name(n) for n in kernels})) + " (round 6 pass on this round's sources; both instantiations where templated on the residual add)"
    js["note"] = ("separate rocprofv3 runs per counter group (tools/profile_r06.sh over tools/pmc_launches.py, post-ReLU-like data); FETCH_SIZE counts 64 B per "
                  "128-B request (x2, profiles/r01_pmc_calibration.txt); cycles = GRBM_GUI_ACTIVE / 8 XCDs; effective clock = cycles / the launch time of the "
                  "kernel-trace pass of the same script (6 cold launches: lower than inside a long run -- bench.py divides the cycles by ITS launch time)")
    js["source_round6"] = ["profiles/" + os.path.basename(src), "profiles/r01_pmc_calibration.txt (FETCH_SIZE x 2)"]
    json.dump(js, open(dst, "w"), indent=1)
    print(json.dumps({k: js[k] for k in ("family", "hbm_bytes_per_launch", "ratio_to_algorithmic", "cycles_per_mfma", "effective_clock_GHz_mean", "kernel_source_sha256") if k in js}))


if __name__ == "__main__":
    main()
