"""Dihedral augmentation of (states, pi) batches (reference: alpha_zero/utils/transformation.py:34-167) as a
gather kernel (azsp_dihedral): pure index permutation, any dtype, pass column untouched.  On top of the
reference's five transforms (h-flip, v-flip, rot90/180/270 counter-clockwise) the kernel also offers the
two remaining D8 elements (ops 6, 7)."""
import random

import torch

OPS = {"identity": 0, "h_flip": 1, "v_flip": 2, "rotate90": 3, "rotate180": 4, "rotate270": 5, "transpose": 6, "anti_transpose": 7}


def _binding_for(t):
    from .. import _lib

    return _lib.load(require_gpu=True)


def dihedral(states: torch.Tensor, pi_probs: torch.Tensor, op: int, binding=None):
    if not isinstance(states, torch.Tensor) or len(states.shape) != 4:
        raise ValueError(f"Expect states to be a 4D torch.Tensor, got {states}")
    if not isinstance(pi_probs, torch.Tensor) or len(pi_probs.shape) != 2:
        raise ValueError(f"Expect pi_probs to be a 2D torch.Tensor, got {pi_probs}")
    b = binding or _binding_for(states)
    states, pi_probs = states.contiguous(), pi_probs.contiguous()
    B, C, N, _ = states.shape
    so, po = torch.empty_like(states), torch.empty_like(pi_probs)
    stream = None
    if states.is_cuda:
        import ctypes

        stream = ctypes.c_void_p(torch.cuda.current_stream(states.device).cuda_stream)
    rc = b.dll.azsp_dihedral(states.data_ptr(), so.data_ptr(), states.element_size(), pi_probs.data_ptr(), po.data_ptr(),
                             pi_probs.element_size(), B, C, N, pi_probs.shape[1], op, stream)
    if rc != 0:
        raise ValueError(f"Expect pi_probs with N*N or N*N+1 columns matching the {N}x{N} states (azsp_dihedral code {rc})")
    return so, po


def apply_horizontal_flip(states, pi_probs, binding=None):
    return dihedral(states, pi_probs, OPS["h_flip"], binding)


def apply_vertical_flip(states, pi_probs, binding=None):
    return dihedral(states, pi_probs, OPS["v_flip"], binding)


def apply_rotation(states, pi_probs, angle, binding=None):
    if not isinstance(states, torch.Tensor) or len(states.shape) != 4:
        raise ValueError(f"Expect states to be a 4D torch.Tensor, got {states}")
    if angle not in [90, 180, 270]:
        raise ValueError(f"Expect angle to be one of [90, 180, 270], got {angle}")
    return dihedral(states, pi_probs, {90: 3, 180: 4, 270: 5}[angle], binding)


TRANSFORMATIONS = ["h_flip", "v_flip", "rotate90", "rotate180", "rotate270"]


def apply_random_transformation(states, pi_probs, values, binding=None):
    """One transform for the whole batch with probability 0.5, chosen with Python's `random` (transformation.py:160-167)."""
    if random.random() > 0.5:
        t = random.choice(TRANSFORMATIONS)
        states, pi_probs = dihedral(states, pi_probs, OPS[t], binding)
    return states, pi_probs, values
