"""Sample format of the self-play path (reference: alpha_zero/core/replay.py:14-20)."""
from typing import NamedTuple, Optional

import numpy as np


class Transition(NamedTuple):
    state: Optional[np.ndarray]    # int8[17, N, N] observation before the move, mover's perspective
    pi_prob: Optional[np.ndarray]  # search policy over all A actions
    value: Optional[float]         # z: +reward for the eventual last player's samples, -reward for the other's


TransitionStructure = Transition(state=None, pi_prob=None, value=None)


class DeviceReplay:
    """UniformReplay (reference alpha_zero/core/replay.py:35-118) with its storage as an HBM ring.

    Same semantics: ring write at `num_samples_added % capacity` (:64-68), `sample` returns None while
    `size < batch_size` and otherwise draws `random_state.randint(0, size, batch_size)` with replacement (:72-83), so with
    the same RandomState and the same games the batches equal the reference's.  What is different is where the data
    lives: harvested games go from the engine's output tensors into the ring device-to-device (`add_harvest`), and
    `sample_device` returns the batch as device tensors, already transformed by one dihedral op and cast for the network
    (core/pipeline.py:636-643), from one gather kernel (`azsp_replay_gather`) -- no host round trip per batch.
    `add_game` / `sample` / `get_state` / `set_state` keep the reference's host-side signatures."""

    def __init__(self, capacity, random_state, board_size, num_actions, channels=17, device="cuda", binding=None, pi_dtype=np.float64):
        import torch

        if capacity <= 0:
            raise ValueError(f"Expect capacity to be a positive integer, got {capacity}")
        if binding is None:
            from .. import _lib

            binding = _lib.load(require_gpu=True)
        self.b = binding
        self.structure = TransitionStructure
        self.capacity, self.random_state = capacity, random_state
        self.N, self.A, self.C = board_size, num_actions, channels
        self.device = torch.device(device)
        self.pi_dtype = pi_dtype
        self.states = torch.zeros((capacity, channels, board_size, board_size), dtype=torch.int8, device=self.device)
        self.pi = torch.zeros((capacity, num_actions), dtype=torch.float32, device=self.device)
        self.z = torch.zeros((capacity,), dtype=torch.float32, device=self.device)
        self.num_games_added = 0
        self.num_samples_added = 0

    # -- writes ------------------------------------------------------------------------------------------------------
    def add_samples(self, states, pi, z, num_games=1):
        """Appends samples in order (device or host tensors): the ring positions the reference's per-transition loop
        would have written (:55-68).  More samples than the capacity keep only the last `capacity` (the earlier ones
        would have been overwritten by the same call)."""
        import torch

        n = int(z.shape[0])
        if n:
            states = torch.as_tensor(states).to(self.device, torch.int8)
            pi = torch.as_tensor(pi).to(self.device, torch.float32)
            z = torch.as_tensor(z).to(self.device, torch.float32)
            skip = max(0, n - self.capacity)
            pos = (torch.arange(self.num_samples_added + skip, self.num_samples_added + n, device=self.device, dtype=torch.int64)
                   % self.capacity)
            self.states.index_copy_(0, pos, states[skip:])
            self.pi.index_copy_(0, pos, pi[skip:])
            self.z.index_copy_(0, pos, z[skip:])
            self.num_samples_added += n
        self.num_games_added += num_games

    def add_harvest(self, states, pi, z, games):
        """Everything `SelfPlayActor.harvest_tensors()` returned: finished games, concatenated in harvest order."""
        self.add_samples(states, pi, z, num_games=len(games))

    def add_game(self, game_seq):
        import torch

        if len(game_seq) == 0:
            self.num_games_added += 1
            return
        self.add_samples(torch.from_numpy(np.stack([t.state for t in game_seq])), torch.from_numpy(np.stack([t.pi_prob for t in game_seq]).astype(np.float32)),
                         torch.tensor([float(t.value) for t in game_seq], dtype=torch.float32))

    @property
    def size(self):
        return min(self.num_samples_added, self.capacity)

    # -- reads -------------------------------------------------------------------------------------------------------
    def _gather(self, indices, op, state_dtype):
        import ctypes

        import torch

        from .. import _abi

        code = {torch.int8: _abi.FEAT_I8, torch.float32: _abi.FEAT_F32, torch.bfloat16: _abi.FEAT_BF16, torch.float16: _abi.FEAT_F16}[state_dtype]
        B = len(indices)
        idx = torch.as_tensor(np.asarray(indices, dtype=np.int64)).to(self.device)
        st = torch.empty((B, self.C, self.N, self.N), dtype=state_dtype, device=self.device)
        pi = torch.empty((B, self.A), dtype=torch.float32, device=self.device)
        z = torch.empty((B,), dtype=torch.float32, device=self.device)
        stream = ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream) if self.device.type == "cuda" else None
        rc = self.b.dll.azsp_replay_gather(self.states.data_ptr(), self.pi.data_ptr(), self.z.data_ptr(), idx.data_ptr(), B, self.C, self.N, self.A,
                                           int(op), code, st.data_ptr(), pi.data_ptr(), z.data_ptr(), stream)
        if rc != 0:
            raise RuntimeError(f"azsp_replay_gather failed with code {rc}")
        return st, pi, z

    def sample(self, batch_size):
        """Reference signature (:72-83): Transition of stacked numpy arrays, or None while the replay is too small."""
        if self.size < batch_size:
            return None
        indices = self.random_state.randint(low=0, high=self.size, size=batch_size)
        st, pi, z = self._gather(indices, 0, __import__("torch").int8)
        return Transition(st.cpu().numpy(), pi.cpu().numpy().astype(self.pi_dtype), z.cpu().numpy().astype(np.float64))

    def sample_device(self, batch_size, transform=None, state_dtype=None):
        """The batch as device tensors (states cast to `state_dtype`, default float32 like core/pipeline.py:636).
        transform: None, an op code 0..7 (azsp_dihedral numbering) or "random" = apply_random_transformation
        (utils/transformation.py:160-167: with probability 0.5 one of the five reference transforms for the whole batch,
        chosen with Python's `random`)."""
        import random

        import torch

        if self.size < batch_size:
            return None
        indices = self.random_state.randint(low=0, high=self.size, size=batch_size)
        op = 0
        if transform == "random":
            if random.random() > 0.5:
                op = {"h_flip": 1, "v_flip": 2, "rotate90": 3, "rotate180": 4, "rotate270": 5}[random.choice(["h_flip", "v_flip", "rotate90", "rotate180", "rotate270"])]
        elif transform is not None:
            op = int(transform)
        return self._gather(indices, op, state_dtype or torch.float32)

    # -- persistence (reference dictionary, :99-111; uncompressed transitions) ----------------------------------------
    def get_state(self):
        n = self.size
        st, pi, z = self.states[:n].cpu().numpy(), self.pi[:n].cpu().numpy().astype(self.pi_dtype), self.z[:n].cpu().numpy()
        storage = [Transition(st[i].copy(), pi[i].copy(), float(z[i])) for i in range(n)] + [None] * (self.capacity - n)
        return {"num_games_added": self.num_games_added, "num_samples_added": self.num_samples_added, "storage": storage}

    def set_state(self, state):
        import torch

        storage = state["storage"]
        for i, t in enumerate(storage[: self.capacity]):
            if t is None:
                continue
            s = t.state
            if isinstance(s, tuple):  # the reference's snappy-compressed form (compress_array, :24-26)
                import snappy

                s = np.frombuffer(snappy.uncompress(s[0]), dtype=s[2]).reshape(s[1])
            self.states[i] = torch.from_numpy(np.ascontiguousarray(s)).to(self.device)
            self.pi[i] = torch.from_numpy(np.asarray(t.pi_prob, dtype=np.float32)).to(self.device)
            self.z[i] = float(t.value)
        self.num_games_added = state["num_games_added"]
        self.num_samples_added = state["num_samples_added"]
