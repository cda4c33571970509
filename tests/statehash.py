"""Canonical byte records used to fingerprint environment trajectories.

The SAME record layout is produced from (a) the upstream reference envs when the
golden vectors are generated (tools/gen_golden.py, this container only), (b) the
CPU oracle and (c) the HIP engine in the tests, so a 16-byte digest per game
pins board / legal mask / ko / captures / turn / step count / termination /
reward / observation planes bit-exactly for every position of the game.
"""
import hashlib
import numpy as np


def env_record(board, legal, ko, caps, to_play, steps, done, reward) -> bytes:
    """One position -> bytes.  `ko` is a flat point index or -1; caps = (black, white)."""
    head = np.array([ko, caps[0], caps[1], steps], dtype="<i2").tobytes()
    tail = np.array([to_play, 1 if done else 0, int(reward)], dtype=np.int8).tobytes()
    return (
        np.ascontiguousarray(board, dtype=np.int8).tobytes()
        + np.ascontiguousarray(legal, dtype=np.int8).tobytes()
        + head
        + tail
    )


class TrajectoryHasher:
    """Two running digests per game: environment state stream and observation stream."""

    def __init__(self):
        self.h_state = hashlib.sha256()
        self.h_obs = hashlib.sha256()

    def add(self, record: bytes, obs) -> None:
        self.h_state.update(record)
        self.h_obs.update(np.ascontiguousarray(obs, dtype=np.int8).tobytes())

    def digests(self):
        return self.h_state.digest()[:16], self.h_obs.digest()[:16]
