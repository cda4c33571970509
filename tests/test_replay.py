"""replay ingest / sampling on the device vs the reference's UniformReplay (golden) and the oracle restatement."""
import os
import random
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import engine_util as eu  # noqa: E402
from alpha_zero_amd.core.replay import DeviceReplay, Transition  # noqa: E402
from oracle.replay import OracleUniformReplay  # noqa: E402


def _golden(golden_dir):
    return np.load(os.path.join(golden_dir, "replay_uniform.npz"))


def _games(g):
    return [(g[f"g{i}_state"], g[f"g{i}_pi"], g[f"g{i}_z"]) for i in range(len(g["lengths"]))]


def test_oracle_replay_matches_reference(golden_dir):
    g = _golden(golden_dir)
    rp = OracleUniformReplay(int(g["capacity"]), np.random.RandomState(int(g["seed"])))
    assert rp.sample(int(g["batch"])) is None
    for gi, (st, pi, z) in enumerate(_games(g)):
        rp.add_game([Transition(st[i], pi[i], float(z[i])) for i in range(len(z))])
        assert rp.size == g["sizes"][gi]
        b = rp.sample(int(g["batch"]))
        assert np.array_equal(b.state, g[f"b{gi}_state"]) and np.array_equal(b.pi_prob, g[f"b{gi}_pi"]) and np.array_equal(b.value, g[f"b{gi}_z"])
    assert rp.num_games_added == g["num_games_added"] and rp.num_samples_added == g["num_samples_added"]
    with pytest.raises(ValueError):
        OracleUniformReplay(0, np.random.RandomState(0))


def _check_device_replay(binding, device, golden_dir):
    g = _golden(golden_dir)
    cap, batch = int(g["capacity"]), int(g["batch"])
    rp = DeviceReplay(cap, np.random.RandomState(int(g["seed"])), 5, 26, device=device, binding=binding)
    assert rp.sample(batch) is None and rp.sample_device(batch) is None
    for gi, (st, pi, z) in enumerate(_games(g)):
        if gi % 2 == 0:  # host-side reference signature
            rp.add_game([Transition(st[i], pi[i], float(z[i])) for i in range(len(z))])
        else:            # what the actor's harvest hands over: concatenated tensors
            rp.add_harvest(torch.from_numpy(st).to(device), torch.from_numpy(pi.astype(np.float32)).to(device), torch.from_numpy(z.astype(np.float32)).to(device),
                           games=[0])
        assert rp.size == g["sizes"][gi]
        b = rp.sample(batch)
        assert b.state.dtype == np.int8 and b.pi_prob.dtype == np.float64
        assert np.array_equal(b.state, g[f"b{gi}_state"]) and np.array_equal(b.pi_prob, g[f"b{gi}_pi"]) and np.array_equal(b.value, g[f"b{gi}_z"])
    assert rp.num_games_added == g["num_games_added"] and rp.num_samples_added == g["num_samples_added"]
    # device batches: same indices as the oracle with the same RandomState; every dihedral op against the reference's definition
    orc = OracleUniformReplay(cap, np.random.RandomState(77))
    orc.set_state({k: (v if k != "storage" else list(v)) for k, v in rp.get_state().items()})
    rp.random_state = np.random.RandomState(77)
    for op in range(8):
        want = orc.sample(batch)
        st, pi, z = rp.sample_device(batch, transform=op, state_dtype=torch.float32)
        ws, wp = torch.from_numpy(want.state.astype(np.float32)), torch.from_numpy(want.pi_prob.astype(np.float32))
        board, pas = wp[:, :25].reshape(batch, 1, 5, 5), wp[:, 25:]
        def T(x):
            return {0: x, 1: torch.flip(x, dims=[-1]), 2: torch.flip(x, dims=[-2]), 3: torch.rot90(x, 1, [-2, -1]), 4: torch.rot90(x, 2, [-2, -1]),
                    5: torch.rot90(x, 3, [-2, -1]), 6: x.transpose(-1, -2), 7: torch.flip(x.transpose(-1, -2), dims=[-1, -2])}[op]
        assert torch.equal(st.cpu(), T(ws)) and torch.equal(pi.cpu(), torch.cat([T(board).reshape(batch, 25), pas], 1))
        assert np.array_equal(z.cpu().numpy(), want.value.astype(np.float32))
    # "random" follows apply_random_transformation's use of Python's `random`
    random.seed(3)
    draws = [(random.random() > 0.5) and random.choice(["h_flip", "v_flip", "rotate90", "rotate180", "rotate270"]) for _ in range(6)]
    random.seed(3)
    rp.random_state, orc.random_state = np.random.RandomState(9), np.random.RandomState(9)
    for d in draws:
        want = orc.sample(batch)
        st, pi, z = rp.sample_device(batch, transform="random", state_dtype=torch.bfloat16)
        ws = torch.from_numpy(want.state.astype(np.float32))
        exp = {False: ws, "h_flip": torch.flip(ws, dims=[-1]), "v_flip": torch.flip(ws, dims=[-2]), "rotate90": torch.rot90(ws, 1, [-2, -1]),
               "rotate180": torch.rot90(ws, 2, [-2, -1]), "rotate270": torch.rot90(ws, 3, [-2, -1])}[d]
        assert st.dtype == torch.bfloat16 and torch.equal(st.float().cpu(), exp)
    # persistence round trip in the reference's dictionary format
    rp2 = DeviceReplay(cap, np.random.RandomState(1), 5, 26, device=device, binding=binding)
    rp2.set_state(rp.get_state())
    assert torch.equal(rp2.states, rp.states) and torch.equal(rp2.pi, rp.pi) and torch.equal(rp2.z, rp.z) and rp2.size == rp.size
    # more samples than the capacity in one call: last `capacity` win, like the reference's sequential overwrite
    big = DeviceReplay(10, np.random.RandomState(1), 5, 26, device=device, binding=binding)
    ob = OracleUniformReplay(10, np.random.RandomState(1))
    st, pi, z = _games(g)[3]
    big.add_samples(torch.from_numpy(st), torch.from_numpy(pi.astype(np.float32)), torch.from_numpy(z.astype(np.float32)))
    ob.add_game([Transition(st[i], pi[i], float(z[i])) for i in range(len(z))])
    a, b2 = big.sample(6), ob.sample(6)
    assert np.array_equal(a.state, b2.state) and np.array_equal(a.pi_prob, b2.pi_prob) and np.array_equal(a.value, b2.value)
    with pytest.raises(ValueError):
        DeviceReplay(0, np.random.RandomState(0), 5, 26, device=device, binding=binding)


def test_device_replay_host_twin(golden_dir):
    _check_device_replay(eu.hosttwin_binding(), "cpu", golden_dir)


@pytest.mark.gpu
def test_gpu_device_replay(golden_dir):
    from alpha_zero_amd import _lib

    _check_device_replay(_lib.load(), "cuda", golden_dir)
