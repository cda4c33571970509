"""Import harness for the upstream reference (this container only).

Installs tiny stand-ins for third-party modules that are absent from the image
(gym, sgf, snappy, torchvision) into ``sys.modules`` and puts /root/reference on
``sys.path`` so that ``alpha_zero.*`` can be imported to GENERATE golden vectors.

Nothing here ships to the GPU box; /root/reference does not exist there.  The
stand-ins only cover third-party *packages*, never reference code:
  * gym.Env / gym.spaces.{Box,Discrete}: attribute holders (base.py:13-15 only reads .shape/.n)
  * torchvision.transforms.functional.{rotate,hflip,vflip}: expressed with
    torch.rot90/torch.flip, which is exactly what unit_tests/transformation_test.py
    asserts them to equal.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("AZ_REFERENCE_ROOT", "/root/reference")


def install(board_size: int = 9):
    """Must be called before any alpha_zero.envs.go* import (go_engine.py:31 reads BOARD_SIZE at import)."""
    os.environ["BOARD_SIZE"] = str(board_size)
    if "gym" not in sys.modules:
        gym = types.ModuleType("gym")

        class Env:  # minimal gym.Env
            def reset(self, **kwargs):
                return None

            def close(self):
                return None

        spaces = types.ModuleType("gym.spaces")

        class Box:
            def __init__(self, low=None, high=None, shape=None, dtype=None):
                self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

        class Discrete:
            def __init__(self, n):
                self.n = n

        spaces.Box, spaces.Discrete = Box, Discrete
        gym.Env, gym.spaces = Env, spaces
        sys.modules["gym"] = gym
        sys.modules["gym.spaces"] = spaces
    for name in ("sgf", "snappy"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    if "torchvision" not in sys.modules:
        import torch

        tv = types.ModuleType("torchvision")
        tr = types.ModuleType("torchvision.transforms")
        fn = types.ModuleType("torchvision.transforms.functional")
        fn.rotate = lambda x, angle: torch.rot90(x, k=int(angle) // 90, dims=[-2, -1])
        fn.hflip = lambda x: torch.flip(x, dims=[-1])
        fn.vflip = lambda x: torch.flip(x, dims=[-2])
        tv.transforms, tr.functional = tr, fn
        sys.modules["torchvision"] = tv
        sys.modules["torchvision.transforms"] = tr
        sys.modules["torchvision.transforms.functional"] = fn
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
