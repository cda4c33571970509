// Host twin of libazsp.so -- TEST INFRASTRUCTURE ONLY.
// Compiles the *same* engine source (alpha_zero_amd/csrc/az_*.h, azsp_impl.h) with the WaveHost
// policy: every "kernel" is a loop over games, every wave section a loop over 64 lanes.  It lets the
// CPU-only test tier exercise tree search, rules and the C ABI logic without a GPU.  The product
// package never loads this library (alpha_zero_amd/_lib.py only accepts libazsp.so + a HIP device).
#include <stdlib.h>

#include "../../alpha_zero_amd/csrc/azsp_impl.h"

namespace azb {
void* alloc(size_t n) { return calloc(1, n); }
void release(void* p) { free(p); }
int h2d(void* d, const void* s, size_t n, void*) { memcpy(d, s, n); return 0; }
int d2h(void* d, const void* s, size_t n, void*) { memcpy(d, s, n); return 0; }
int zero(void* d, size_t n, void*) { memset(d, 0, n); return 0; }
int sync(void*) { return 0; }
int set_device(int) { return 0; }
const char* backend_error() { return "host twin"; }
template <int N, int GAME, class Op> int launch(const AzCfg& c, const AzMem& m, const Op& op, void*) {
    for (int g = 0; g < c.G; ++g) {
        typename Engine<WaveHost, N, GAME>::SC sc;
        Engine<WaveHost, N, GAME> e(c, m, g, sc);
        op(e);
    }
    return 0;
}
int launch_dihedral(const DihedralArgs& a, long long total, void*) {
    for (long long t = 0; t < total; ++t) az_dihedral_elem(a, t);
    return 0;
}
int launch_bias_act(const BiasActArgs& a, void*) {
    for (long long i = 0; i < a.nvec; ++i) az_bias_act_vec(a, i);
    return 0;
}
int launch_conv3x3(const void* x, const void* w, const float* bias, const void* res, void* y, long long boards, int S, int C, int relu, void*) {
    cv_host_conv3x3((const unsigned short*)x, (const unsigned short*)w, bias, (const unsigned short*)res, (unsigned short*)y, (int)boards, S, C, relu);
    return 0;
}
}  // namespace azb
