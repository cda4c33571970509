"""The minimal ctypes binding INTEGRATION.md section 3 shows a maintainer (struct layout, call order, per-game actor state, harvest
with extras) executed for real -- on the host twin in the CPU tier and on libazsp.so in the GPU tier -- so the document cannot rot."""
import ctypes as C

import numpy as np
import pytest
import torch

import engine_util as eu


class AzspConfig(C.Structure):  # copied from INTEGRATION.md: 26 int32, 4 float, 4 double, 1 uint64
    _fields_ = [(n, C.c_int32) for n in (
        "game", "board_size", "num_games", "num_parallel", "num_simulations", "max_nodes", "root_noise",
        "deterministic", "reuse_tree", "warm_up_steps", "has_resign", "check_resign_after_steps",
        "force_resign_disabled", "inject_random", "inject_moves", "stop_after_move", "max_plies",
        "stop_at_game_end", "feature_dtype", "log_moves", "log_capacity", "max_steps", "num_to_win",
        "training_steps", "rank", "device")] + [
        ("c_puct_base", C.c_float), ("c_puct_init", C.c_float), ("disable_resign_ratio", C.c_float),
        ("reserved0", C.c_float), ("dirichlet_eps", C.c_double), ("dirichlet_alpha", C.c_double),
        ("resign_threshold", C.c_double), ("komi", C.c_double), ("seed", C.c_uint64)]


def _run_stub(kind):
    from alpha_zero_amd.core.engine import pbc_tables

    binding, dev = eu.backend(kind)
    lib = C.CDLL(binding.dll._name)  # a fresh handle without the package's argtypes: exactly what INTEGRATION.md's reader has
    V = C.c_void_p
    G, P, n, A, sims = 4, 4, 5, 26, 16
    eng = C.c_void_p()
    cfg = AzspConfig(game=0, board_size=n, num_games=G, num_parallel=P, num_simulations=sims, root_noise=1, reuse_tree=1, warm_up_steps=4,
                     has_resign=1, check_resign_after_steps=4, force_resign_disabled=-1, feature_dtype=1, dirichlet_eps=0.25, dirichlet_alpha=0.03,
                     resign_threshold=-1.0, komi=7.5, c_puct_base=19652.0, c_puct_init=1.25, disable_resign_ratio=0.5, seed=1)
    assert lib.azsp_create(C.byref(cfg), C.byref(eng)) == 0
    table_len = sims + P + 3 * P + 16
    pbc_np, pbc_py, sqrt32 = pbc_tables(19652.0, 1.25, table_len)
    assert lib.azsp_set_tables(eng, V(pbc_np.ctypes.data), V(pbc_py.ctypes.data), V(sqrt32.ctypes.data), table_len) == 0
    assert lib.azsp_reset_games(eng, None) == 0
    feats = torch.zeros((G * P, 17, n, n), dtype=torch.float32, device=dev)
    valid = torch.zeros((G * P,), dtype=torch.uint8, device=dev)
    priors = torch.full((G * P, A), 1.0 / A, dtype=torch.float32, device=dev)
    values = torch.zeros((G * P,), dtype=torch.float32, device=dev)
    states = torch.empty((4 * G * 50, 17, n, n), dtype=torch.int8, device=dev)
    pi = torch.empty((4 * G * 50, A), dtype=torch.float32, device=dev)
    z = torch.empty((4 * G * 50,), dtype=torch.float32, device=dev)
    games = np.zeros((2 * G, 16), dtype=np.int32)
    extra = np.zeros((2 * G, 4), dtype=np.int32)
    got, thr_seen = 0, set()
    ns, ng = C.c_int32(0), C.c_int32(0)
    for r in range(3000):
        assert lib.azsp_round(eng, C.c_void_p(priors.data_ptr()), C.c_void_p(values.data_ptr()), C.c_void_p(feats.data_ptr()),
                              C.c_void_p(valid.data_ptr()), None) == 0
        # "network": uniform priors, value from a cheap function of the planes (enough to drive complete games)
        values.copy_((feats[:, 0].sum(dim=(1, 2)) - feats[:, 1].sum(dim=(1, 2))) / 25.0)
        if r == 20:  # pipeline.py:232-246: new threshold / checkpoint tag for games that start from now on
            assert lib.azsp_set_actor_state(eng, C.c_double(-0.5), 123) == 0
        if r % 25 == 24:
            assert lib.azsp_harvest_extra(eng, V(extra.ctypes.data)) == 0
            assert lib.azsp_harvest(eng, C.c_void_p(states.data_ptr()), C.c_void_p(pi.data_ptr()), C.c_void_p(z.data_ptr()), states.shape[0],
                                    V(games.ctypes.data), 2 * G, C.byref(ns), C.byref(ng), None) == 0
            for k in range(ng.value):
                thr = float(np.array([extra[k, 1], extra[k, 2]], dtype=np.int32).view(np.float64)[0])
                thr_seen.add((int(games[k, 12]), thr))
                assert games[k, 1] > 0 and abs(float(pi[games[k, 0]].sum()) - 1.0) < 1e-4
            got += ng.value
            if (123, -0.5) in thr_seen and got >= 2 * G:
                break
    assert lib.azsp_harvest_extra(eng, None) == 0
    assert (0, -1.0) in thr_seen and (123, -0.5) in thr_seen and thr_seen <= {(0, -1.0), (123, -0.5)}
    assert lib.azsp_destroy(eng) == 0


def test_integration_stub_host_twin():
    _run_stub("host")


@pytest.mark.gpu
def test_gpu_integration_stub():
    _run_stub("gpu")
