"""Sample format of the self-play path (reference: alpha_zero/core/replay.py:14-20)."""
from typing import NamedTuple, Optional

import numpy as np


class Transition(NamedTuple):
    state: Optional[np.ndarray]    # int8[17, N, N] observation before the move, mover's perspective
    pi_prob: Optional[np.ndarray]  # search policy over all A actions
    value: Optional[float]         # z: +reward for the eventual last player's samples, -reward for the other's


TransitionStructure = Transition(state=None, pi_prob=None, value=None)
