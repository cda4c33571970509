"""CPU tier: the engine source (alpha_zero_amd/csrc/az_*.h + azsp_impl.h) compiled as the host twin
(tests/hosttwin) is driven through the same C ABI and compared with the golden vectors produced by
the reference.  This validates the shared logic; the HIP build is exercised by the `-m gpu` tests."""
import ctypes
import os

import numpy as np
import pytest

import golden_mcts
import parity_checks as pc


def test_go9_all_shipped_sgf_games(golden_dir):
    bad, n = pc.check_go_file("host", os.path.join(golden_dir, "go9_sgf.npz"), 9)
    assert n == 10288 and not bad, bad[:5]


@pytest.mark.parametrize("n", [5, 9, 13, 19])
def test_go_random_playouts(golden_dir, n):
    bad, cnt = pc.check_go_file("host", os.path.join(golden_dir, f"go{n}_random.npz"), n)
    assert cnt > 0 and not bad


def test_gomoku_playouts_and_lines(golden_dir):
    bad, cnt = pc.check_gomoku_file("host", os.path.join(golden_dir, "gomoku.npz"))
    assert cnt == 536 and not bad


@pytest.mark.parametrize("name", golden_mcts.names())
def test_search_and_actor_match_reference(name):
    pc.check_mcts_golden("host", name)


@pytest.mark.parametrize("fmt", ["bf16", "f16", "f16_split"])
@pytest.mark.parametrize("name", ["go5_p4_s48_resign", "go9_p1_s50", "gomoku7_p8_s64"])
def test_tiled_feature_layout_matches_reference(name, fmt):
    """Observation planes written in the evaluators' own input layouts (AZSP_FEAT_BF16_TILED / AZSP_FEAT_F16_TILED; AZSP_FEAT_F16_SPLIT =
    the fp32-class stem's split layout, hi plane only) carry exactly the reference's planes: the whole golden game replays bit-exactly
    when the evaluator sees the decoded tensor (the decoders also check the encoding: only 0 / 1, padding zero, lo plane untouched)."""
    from alpha_zero_amd import _abi

    pc.check_mcts_golden("host", name, feature_dtype={"bf16": _abi.FEAT_BF16_TILED, "f16": _abi.FEAT_F16_TILED, "f16_split": _abi.FEAT_F16_SPLIT}[fmt])


@pytest.mark.parametrize("name", ["go5_p8_s64", "go9_p8_s200", "gomoku7_p1_s40"])
def test_deep_path_fallback_matches_reference(name):
    """Paths deeper than AZ_PATH_CAP walk the parent links instead of the lane-parallel path store; with the cap
    compiled down to 2 almost every backup / virtual loss takes that route and must still match the reference."""
    pc.check_mcts_golden("host:cap2", name)


def test_product_library_exports_every_abi_symbol():
    """libazsp.so (HIP build) must load on a CPU-only box and export all of include/azsp.h; no compute calls here."""
    from alpha_zero_amd import _abi, _lib

    if not os.path.exists(_lib.library_path()):
        import __graft_entry__

        __graft_entry__.build()
    b = _lib.load(require_gpu=False)
    for s in _abi.SYMBOLS:
        assert hasattr(b.dll, s)
    import re

    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "azsp.h")).read()
    declared = set(re.findall(r"\b(azsp_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_abi.SYMBOLS)


def test_product_refuses_to_run_without_gpu():
    import torch

    from alpha_zero_amd import _abi, _lib

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_abi.AzspError, match="no CPU fallback"):
        _lib.load(require_gpu=True)


def test_production_randomness_statistics():
    import rng_checks as rc

    rc.check_production_rng("host")


def test_host_go19_known_sequences_and_score_boards(golden_dir):
    import edge_checks as ec

    ec.check_go19_known_sequences("host", golden_dir)
    ec.check_go9_score_boards("host", golden_dir)
