"""bench.py's roofline arithmetic without a GPU: algorithmic flops per launch, issued MFMA products of the fp32-class kernels,
the traffic and power-ceiling annotations read from profiles/ (nothing hard-coded), for the unfused 9x9 x 128 tower convolution and the
fused 17x17 x 64 block."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _args(**kw):
    base = dict(board=9, game="go", games=4096, parallel=8, blocks=10, filters=128, net_dtype="fp32")
    base.update(kw)
    return argparse.Namespace(**base)


def test_tower_roofline_of_the_headline_kernel():
    import bench

    conv = {"launches": 100, "avg_ms": 1.74, "planes": 9, "fused_block": False, "split": True, "avg_ms_plain": 1.735, "avg_ms_residual": 1.745, "data": "real"}
    r = bench.tower_roofline(conv, _args(), step_ms=36.0)
    flops = 2.0 * 32768 * 81 * 128 * 128 * 9
    assert r["alg_flops_per_launch"] == flops and r["issued_f16_mfma_flops_per_launch"] == 3 * flops and r["mfma_products_per_multiply"] == 3
    assert abs(r["achieved"] - 3 * flops / 1.74e-3 / 1e12) < 0.01 and r["peak"] == 2500.0 and abs(r["frac"] - r["achieved"] / 2500.0) < 1e-4
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["launches_per_step"] == 20 and "k_conv3x3_sp" in r["kernel"]
    assert r["alg_hbm_bytes_per_launch"] == round(32768 * 81 * 128 * 4 * 2.5)
    pj = json.load(open(os.path.join(ROOT, "profiles", "split_kernel_pmc.json")))
    assert r["traffic"] == round(pj["hbm_bytes_per_launch"] * 32768 / pj["rows"]) and "from_profiles" in r["traffic_source"]
    pc = json.load(open(os.path.join(ROOT, "profiles", "mfma_power_probe.json")))
    # the ceilings: the 16x16x32 form's ISSUE ceiling (all-zero operands: power cannot bind) -- what rounds 4-5 mislabelled "power-limited"
    c = r["mfma_ceilings"]
    zeros16 = [x["tflops"] for x in pc["cases"] if x["mfma"] == "v_mfma_f32_16x16x32_bf16" and x["operands"] == "zeros"][0]
    assert c["instruction_form"] == "v_mfma_f32_16x16x32_f16" and c["issue_ceiling_tflops"] == pc.get("issue_ceiling_16x16x32_tflops", zeros16)
    # on real operands the split mix sustains 1.85-1.98 PF/s depending on the box: at, or a few percent under, the issue ceiling
    assert c["sustained_tflops"] == pc["split_mix_mfma_only_tflops"] <= 1.02 * c["issue_ceiling_tflops"] and c["limit"] in ("issue", "power")
    assert "power_limited_mfma_only_tflops" not in r and "frac_of_power_limited_ceiling" not in r
    assert abs(r["frac_of_issue_ceiling"] - r["achieved"] / c["issue_ceiling_tflops"]) < 1e-3
    assert abs(r["instruction_form_ceiling"] - c["issue_ceiling_tflops"] / 2500.0) < 1e-3 and 0.75 < r["instruction_form_ceiling"] < 0.82
    assert abs(r["frac_algorithmic"] - flops / 1.74e-3 / 1e12 / 2500.0) < 1e-4  # algorithmic flops / dense f16 peak (0.18)
    assert r["traffic_stale"] in (True, False, None) and ("STALE" in r["traffic_source"]) == (r["traffic_stale"] is True)
    if pj.get("cycles_per_mfma"):  # a round-6 PMC pass: the decomposition of frac
        assert abs(r["issue_efficiency"] - (16.0 / r["instruction_form_ceiling"]) / pj["cycles_per_mfma"]) < 2e-3
        assert abs(r["clock_fraction"] - pj["gpu_cycles_per_launch_mean"] / 1.74e-3 / 2.4e9) < 1e-3 and 0.5 < r["clock_fraction"] < 1.0
        assert 0.99 < r["issued_over_algorithmic_mfma"] < 1.03  # 9x9 x 128: 81 positions in 5 column tiles + 1 / 16
        assert abs(r["decomposition_product"] - r["instruction_form_ceiling"] * r["issue_efficiency"] * r["clock_fraction"] / r["issued_over_algorithmic_mfma"]) < 1e-3
        assert abs(r["decomposition_product"] - r["frac"]) < 0.03  # the three factors explain the fraction


def test_tower_roofline_of_the_fused_gomoku_block():
    import bench

    conv = {"launches": 30, "avg_ms": 3.45, "planes": 17, "fused_block": True, "split": True, "avg_ms_plain": None, "avg_ms_residual": None, "data": "real"}
    r = bench.tower_roofline(conv, _args(board=13, game="gomoku", blocks=6, filters=64), step_ms=22.5)
    flops = 2 * 2.0 * 32768 * 289 * 64 * 64 * 9  # two convolutions per launch
    assert r["alg_flops_per_launch"] == flops and r["launches_per_step"] == 6 and "k_resblock_sp<Sb17>" in r["kernel"]
    assert abs(r["frac"] - 3 * flops / 3.45e-3 / 1e12 / 2500.0) < 1e-4
    assert r["alg_hbm_bytes_per_launch"] == round(32768 * 289 * 64 * 4 * 2.0)  # x in, y out
    pj = json.load(open(os.path.join(ROOT, "profiles", "splitblock17_kernel_pmc.json")))
    assert r["traffic"] == pj["hbm_bytes_per_launch"] and pj["hbm_bytes_per_launch"] < 0.55 * pj["two_launch_hbm_bytes_per_block"]


def test_a_pmc_summary_of_edited_kernel_sources_is_labelled_stale(tmp_path, monkeypatch):
    import bench

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_digest

    pj, stale = bench.kernel_pmc("split_kernel_pmc.json")
    assert pj is not None
    if pj.get("kernel_source_sha256"):
        assert stale == (pj["kernel_source_sha256"] != kernel_digest.kernel_source_digest("split9"))
    monkeypatch.setattr(kernel_digest, "kernel_source_digest", lambda fam: "0" * 64)  # "the header was edited"
    pj2, stale2 = bench.kernel_pmc("split_kernel_pmc.json")
    assert stale2 is (True if pj2.get("kernel_source_sha256") else None)
    assert bench.kernel_pmc("no_such_file.json") == (None, None)


def test_every_digest_carrying_pmc_summary_is_fresh_and_labelled_with_the_kernel_it_measured():
    """a summary whose label names another kernel than the one in its counter rows is a stale label. For each
    summary that carries a source digest: the digest matches the tree (bench.py would otherwise print `traffic_stale`), the label is the
    measured kernels' names, the cited counter file exists under profiles/ and names the same kernels, and bench.py's roofline label for
    that shape names the same kernel."""
    import bench

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_digest
    from pmc_to_json import short_name

    seen = 0
    for fn in sorted(os.listdir(os.path.join(ROOT, "profiles"))):
        if not fn.endswith("_kernel_pmc.json"):
            continue
        pj = json.load(open(os.path.join(ROOT, "profiles", fn)))
        if not pj.get("kernel_source_sha256"):
            continue
        seen += 1
        assert pj["kernel_source_sha256"] == kernel_digest.kernel_source_digest(pj["family"]), fn
        names = sorted({short_name(n) for n in pj["per_kernel"]})
        assert pj["kernel"].split(" (round")[0] == " / ".join(names), (fn, pj["kernel"])
        src = os.path.join(ROOT, pj["source_round6"][0])
        assert os.path.exists(src), src
        text = open(src).read()
        assert all(n.split("<")[0] in text for n in names), (fn, names)
    assert seen >= 4
    conv = {"launches": 100, "avg_ms": 1.64, "planes": 9, "fused_block": False, "split": True, "avg_ms_plain": 1.637, "avg_ms_residual": 1.652, "data": "real"}
    r = bench.tower_roofline(conv, _args(), step_ms=36.0)
    assert "k_conv3x3_sp2" in r["kernel"] and r.get("traffic_stale") in (False, None)


def test_tower_roofline_decomposition_of_the_other_shapes():
    """The decomposition closes on every shape with a round-6 PMC pass: form x issue x clock / (issued over algorithmic MFMAs) ~ frac.
    19x19 x 256 runs on v_mfma_f32_32x32x16 (32 768 flop per instruction, twice the 16x16x32 form's): 361 positions in 12 column tiles of
    32 = x1.064 the algorithm's; the fused 17x17 block recomputes halo rows of its half-board tiles: x1.107."""
    import bench

    conv = {"launches": 80, "avg_ms": 2.4956, "planes": 19, "fused_block": False, "split": False, "avg_ms_plain": 2.47, "avg_ms_residual": 2.52, "data": "real"}
    r = bench.tower_roofline(conv, _args(board=19, games=512, blocks=20, filters=256, net_dtype="bf16"), step_ms=101.0)
    assert "k_conv3x3_op19q" in r["kernel"] and r["mfma_ceilings"]["instruction_form"] == "v_mfma_f32_32x32x16_bf16"
    assert abs(r["issued_over_algorithmic_mfma"] - 12 * 32 / 361.0) < 0.01 and abs(r["decomposition_product"] - r["frac"]) < 0.03
    conv = {"launches": 30, "avg_ms": 3.2863, "planes": 17, "fused_block": True, "split": True, "avg_ms_plain": None, "avg_ms_residual": None, "data": "real"}
    r = bench.tower_roofline(conv, _args(board=13, game="gomoku", blocks=6, filters=64), step_ms=21.4)
    assert 1.08 < r["issued_over_algorithmic_mfma"] < 1.13 and abs(r["decomposition_product"] - r["frac"]) < 0.03
    conv = {"launches": 30, "avg_ms": 0.9206, "planes": 9, "fused_block": True, "split": True, "avg_ms_plain": None, "avg_ms_residual": None, "data": "real"}
    r = bench.tower_roofline(conv, _args(blocks=12, filters=64), step_ms=11.95)
    assert "k_resblock_sp<Sb9>" in r["kernel"] and 0.99 < r["issued_over_algorithmic_mfma"] < 1.12 and abs(r["decomposition_product"] - r["frac"]) < 0.03


def test_net_flops_and_power_ceiling_helpers():
    import bench

    f = bench.net_flops_per_eval(9, 82, 10, 128, 128, False)
    assert f == 2 * 17 * 128 * 9 * 81 + 10 * 2 * (2 * 128 * 128 * 9 * 81) + 2 * 128 * 3 * 81 + 2 * (2 * 81) * 82 + 2 * 81 * 128 + 2 * 128
    cs, cb = bench.mfma_ceilings(True), bench.mfma_ceilings(False)
    assert cs["issue_ceiling_tflops"] > 1500 and cb["issue_ceiling_tflops"] > 2200 and cb["sustained_tflops"] > 1500
    assert cb["limit"] == "power" and cs["limit"] in ("issue", "power")  # 32x32x16 on dense data is power-limited; 16x16x32 sits where both limits meet
    if "other_form" in cs:  # round 6: the split mix on the 32x32x16 form issues at ~2.49 PF/s with zero operands and sustains LESS than 16x16x32 on real ones
        assert cs["other_form"]["issue_ceiling_tflops"] > 2300 and cs["other_form"]["sustained_tflops"] < 2200
    assert "profiles/mfma_power_probe.json" in cs["source"]
