"""Loader of the HIP engine library.  There is NO CPU fallback: importing the product on a machine
without the built libazsp.so or without a HIP device raises."""
import ctypes
import os

from ._abi import AzspError, Binding

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libazsp.so")
_binding = None


def library_path():
    return LIB_PATH


def load(require_gpu=True):
    """Returns the Binding over libazsp.so.  require_gpu=False only skips the device check so that the
    CPU-side build test can verify the library loads and exports every symbol of include/azsp.h."""
    global _binding
    if _binding is None:
        if not os.path.exists(LIB_PATH):
            raise AzspError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        _binding = Binding(ctypes.CDLL(LIB_PATH), "libazsp.so")
    if require_gpu:
        import torch

        if not torch.cuda.is_available():
            raise AzspError("alpha_zero_amd needs a HIP device (MI355X / gfx950); none is visible and there is no CPU fallback.")
    return _binding
