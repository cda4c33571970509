"""Deterministic synthetic evaluator used to pin MCTS parity.

It stands in for `eval_position` (pipeline.py:91-123): priors are a float32
softmax over ALL actions (illegal ones included, never renormalised) and the
value is a Python float that is float32-representable, exactly the shapes and
dtypes the reference search receives from the real network wrapper.  Outputs
depend only on the observation bytes, so the reference, the oracle and the HIP
engine can all be driven by the very same function.
"""
import hashlib

import numpy as np


def _one(obs: np.ndarray, num_actions: int, sharp: float):
    seed = int.from_bytes(hashlib.blake2b(np.ascontiguousarray(obs, dtype=np.int8).tobytes(), digest_size=8).digest(), "little")
    rng = np.random.Generator(np.random.PCG64(seed))
    logits = rng.normal(size=num_actions) * sharp
    logits -= logits.max()
    p = np.exp(logits)
    p = (p / p.sum()).astype(np.float32)
    v = float(np.float32(np.tanh(rng.normal() * 0.6)))
    return p, v


def make_eval_func(num_actions: int, sharp: float = 2.0, log=None):
    """Returns eval_func(obs, batched) with the reference signature (mcts_v2.py:303)."""

    def eval_func(obs, batched=False):
        if not batched:
            p, v = _one(obs, num_actions, sharp)
            if log is not None:
                log.append(1)
            return p, v
        outs = [_one(o, num_actions, sharp) for o in obs]
        if log is not None:
            log.append(len(outs))
        return [o[0] for o in outs], [o[1] for o in outs]

    return eval_func


def eval_batch(obs_batch: np.ndarray, num_actions: int, sharp: float = 2.0):
    """Array form: int8[B,17,N,N] -> (float32[B,A], float32[B])."""
    outs = [_one(o, num_actions, sharp) for o in obs_batch]
    return np.stack([o[0] for o in outs]), np.array([o[1] for o in outs], dtype=np.float32)
