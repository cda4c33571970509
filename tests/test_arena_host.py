"""SURVEY 8f-2/3 on the CPU tier (host twin of the engine)."""
import arena_checks as ac


def test_elo_and_resign_controller_match_reference(golden_dir):
    ac.check_elo_and_resign(golden_dir)


def test_eval_against_prev_ckpt_matches_reference(golden_dir):
    ac.check_arena("host", golden_dir)


def test_sgf_text_matches_reference(golden_dir, monkeypatch):
    import sgf_checks as sc

    sc.check_sgf("host", golden_dir, monkeypatch)


def test_checkpoint_dictionary(tmp_path):
    import sgf_checks as sc

    sc.check_checkpoint(str(tmp_path))


def test_parallel_evaluation_games_match_reference(golden_dir):
    ac.check_parallel_arena("host", golden_dir)


def test_device_route_evaluation_games_match_reference(golden_dir):
    ac.check_device_route_arena("host", golden_dir)
