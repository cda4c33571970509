"""CPU tier of the Python boundary: env classes, uct_search / parallel_uct_search drop-ins, Dihedral-8 (host twin)."""
import pytest

import dropin_checks as dc


def test_env_classes():
    dc.check_env_surface("host")


@pytest.mark.parametrize("name", ["go9_p8_s200", "go9_p1_s50", "go5_p8_s64", "go5_p1_s40", "gomoku13_p1_s100", "gomoku7_p8_s64",
                                  "go5_p1_s40_det"])
def test_uct_search_dropin_matches_reference(name):
    dc.check_dropin_search("host", name)


@pytest.mark.parametrize("name", ["go9_p8_s200", "go5_p1_s40", "gomoku13_p1_s100", "gomoku7_p8_s64", "go5_p1_s40_det"])
def test_uct_search_dropin_with_a_device_resident_evaluator_matches_reference(name):
    dc.check_dropin_search("host", name, device_route=True)


def test_search_errors():
    dc.check_search_errors("host")


def test_dihedral():
    dc.check_dihedral("host")


def test_dropin_step_equals_the_separate_entries():
    dc.check_dropin_step_equals_the_separate_entries("host")
