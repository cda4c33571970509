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
    assert r["power_limited_mfma_only_tflops"]["value"] == pc["split_mix_mfma_only_tflops"]
    assert abs(r["frac_of_power_limited_ceiling"] - r["achieved"] / pc["split_mix_mfma_only_tflops"]) < 1e-3


def test_tower_roofline_of_the_fused_gomoku_block():
    import bench

    conv = {"launches": 30, "avg_ms": 3.45, "planes": 17, "fused_block": True, "split": True, "avg_ms_plain": None, "avg_ms_residual": None, "data": "real"}
    r = bench.tower_roofline(conv, _args(board=13, game="gomoku", blocks=6, filters=64), step_ms=22.5)
    flops = 2 * 2.0 * 32768 * 289 * 64 * 64 * 9  # two convolutions per launch
    assert r["alg_flops_per_launch"] == flops and r["launches_per_step"] == 6 and "k_resblock_sp17" in r["kernel"]
    assert abs(r["frac"] - 3 * flops / 3.45e-3 / 1e12 / 2500.0) < 1e-4
    assert r["alg_hbm_bytes_per_launch"] == round(32768 * 289 * 64 * 4 * 2.0)  # x in, y out
    pj = json.load(open(os.path.join(ROOT, "profiles", "splitblock17_kernel_pmc.json")))
    assert r["traffic"] == pj["hbm_bytes_per_launch"] and pj["hbm_bytes_per_launch"] < 0.55 * pj["two_launch_hbm_bytes_per_block"]


def test_net_flops_and_power_ceiling_helpers():
    import bench

    f = bench.net_flops_per_eval(9, 82, 10, 128, 128, False)
    assert f == 2 * 17 * 128 * 9 * 81 + 10 * 2 * (2 * 128 * 128 * 9 * 81) + 2 * 128 * 3 * 81 + 2 * (2 * 81) * 82 + 2 * 81 * 128 + 2 * 128
    assert bench.power_ceiling(True)["value"] > 1500 and bench.power_ceiling(False)["value"] > 1500
    assert "profiles/mfma_power_probe.json" in bench.power_ceiling(True)["source"]
