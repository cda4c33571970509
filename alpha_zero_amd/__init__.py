"""alpha_zero_amd -- MI355X-native batched self-play engine behind the michaelnny/alpha_zero
self-play API (uct_search / parallel_uct_search, GoEnv / GomokuEnv, (state, pi, z) samples).

Python host -> C ABI (include/azsp.h, libazsp.so) -> hand-written HIP kernels for gfx950.
The policy/value ResNet runs on PyTorch-ROCm.  See DESIGN.md and INTEGRATION.md."""
__version__ = "0.1.0"
