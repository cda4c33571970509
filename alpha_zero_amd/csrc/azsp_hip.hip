// azsp_hip.hip -- HIP / gfx950 backend of the engine: builds libazsp.so (the product).
// One workgroup = 256 threads = 4 wavefronts = 4 independent games (no inter-wave traffic, no
// __syncthreads); game g runs on block g/4, which the dispatcher places on XCD (g/4) % 8 in every
// launch, so a game's tree stays affine to one XCD's L2 across rounds.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "azsp_impl.h"
#include "az_conv64.h"
#include "az_conv19.h"
#include "az_conv_sp.h"
#include "az_conv_sp17.h"
#include "az_resblock_sp17.h"
#include "az_conv_sp2.h"
#include "az_conv_spg.h"

static hipError_t g_last = hipSuccess;
#define AZ_HIP(x) ((g_last = (x)) == hipSuccess ? 0 : -1)

template <int N, int GAME, class Op>
__global__ void __launch_bounds__(256) k_game(const AzCfg c, const AzMem m, const Op op, const int g0, const int g1) {
    __shared__ typename Engine<WaveDev, N, GAME>::SC sc[4];
    const int wave = (int)(threadIdx.x >> 6);
    const int g = g0 + (int)blockIdx.x * 4 + wave;  // games [g0, g1): a sub-range launch (g0 % 32 == 0) keeps every game on its XCD
    if (g >= g1) return;
    Engine<WaveDev, N, GAME> e(c, m, g, sc[wave]);
    op(e);
}

__global__ void __launch_bounds__(256) k_dihedral(const DihedralArgs a, long long total) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x)
        az_dihedral_elem(a, t);
}

__global__ void __launch_bounds__(256) k_replay_gather(const ReplayGatherArgs a, long long total) {
    for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x)
        az_replay_gather_elem(a, t);
}

__global__ void __launch_bounds__(256) k_bias_act(const BiasActArgs a) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.nvec; i += (long long)gridDim.x * blockDim.x)
        az_bias_act_vec(a, i);
}

// One workgroup: thread t owns a contiguous run of the rotated buffer order, the runs' totals are scanned in LDS.
__global__ void __launch_bounds__(1024) k_harvest_scan(const int* __restrict__ len, int* __restrict__ ofs, int* __restrict__ counts, int n2, int rot,
                                                        int cap, int max_games) {
    __shared__ int ss[1024], sg[1024], best[2];
    const int t = (int)threadIdx.x, per = (n2 + 1023) / 1024, lo = t * per < n2 ? t * per : n2, hi = lo + per < n2 ? lo + per : n2;
    int s = 0, g = 0;
    for (int i = lo; i < hi; ++i) {
        const int l = len[(i + rot) % n2];
        s += l;
        g += l > 0;
    }
    ss[t] = s;
    sg[t] = g;
    if (t < 2) best[t] = 0;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {  // inclusive Hillis-Steele scan of the run totals
        const int a = t >= d ? ss[t - d] : 0, b = t >= d ? sg[t - d] : 0;
        __syncthreads();
        ss[t] += a;
        sg[t] += b;
        __syncthreads();
    }
    int mine[2] = {0, 0};
    az_harvest_scan_range(len, ofs, n2, rot, cap, max_games, lo, hi, ss[t] - s, sg[t] - g, mine);
    if (mine[1] > 0) {  // the buffers that fit form a prefix of the order: the furthest fitting end is the harvest's total
        atomicMax(&best[0], mine[0]);
        atomicMax(&best[1], mine[1]);
    }
    __syncthreads();
    if (t < 2) counts[t] = best[t];
}

namespace azb {
void* alloc(size_t n) {
    void* p = nullptr;
    if (AZ_HIP(hipMalloc(&p, n))) return nullptr;
    if (AZ_HIP(hipMemset(p, 0, n))) {
        hipFree(p);
        return nullptr;
    }
    return p;
}
void release(void* p) { (void)hipFree(p); }
int h2d(void* d, const void* s, size_t n, void* st) {
    if (AZ_HIP(hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, (hipStream_t)st))) return -1;
    return AZ_HIP(hipStreamSynchronize((hipStream_t)st));  // the host buffer may be reused by the caller
}
int d2h(void* d, const void* s, size_t n, void* st) {
    if (AZ_HIP(hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, (hipStream_t)st))) return -1;
    return AZ_HIP(hipStreamSynchronize((hipStream_t)st));
}
int zero(void* d, size_t n, void* st) { return AZ_HIP(hipMemsetAsync(d, 0, n, (hipStream_t)st)); }
int sync(void* st) { return AZ_HIP(hipStreamSynchronize((hipStream_t)st)); }
void* host_alloc(size_t n) {
    void* p = nullptr;
    return AZ_HIP(hipHostMalloc(&p, n, hipHostMallocDefault)) ? nullptr : p;
}
void host_release(void* p) { (void)hipHostFree(p); }
int set_device(int dev) {
    int n = 0;
    if (AZ_HIP(hipGetDeviceCount(&n)) || n < 1) return -1;
    return AZ_HIP(hipSetDevice(dev));
}
const char* backend_error() { return hipGetErrorString(g_last); }
template <int N, int GAME, class Op> int launch(const AzCfg& c, const AzMem& m, const Op& op, void* st, int g0, int g1) {
    const dim3 grid((unsigned)((g1 - g0 + 3) / 4)), block(256);
    hipLaunchKernelGGL((k_game<N, GAME, Op>), grid, block, 0, (hipStream_t)st, c, m, op, g0, g1);
    return AZ_HIP(hipGetLastError());
}
int launch_dihedral(const DihedralArgs& a, long long total, void* st) {
    long long blocks = (total + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_dihedral, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)st, a, total);
    return AZ_HIP(hipGetLastError());
}
int launch_replay_gather(const ReplayGatherArgs& a, long long total, void* st) {
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_replay_gather, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)st, a, total);
    return AZ_HIP(hipGetLastError());
}
int launch_harvest_scan(const int* len, int* ofs, int* counts, int n2, int rot, int cap, int max_games, void* st) {
    hipLaunchKernelGGL(k_harvest_scan, dim3(1), dim3(1024), 0, (hipStream_t)st, len, ofs, counts, n2, rot, cap, max_games);
    return AZ_HIP(hipGetLastError());
}
int launch_bias_act(const BiasActArgs& a, void* st) {
    long long blocks = (a.nvec + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;  // grid-stride beyond 32 blocks per CU
    hipLaunchKernelGGL(k_bias_act, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)st, a);
    return AZ_HIP(hipGetLastError());
}
static int cu_count() {
    static int n_cu = 0;
    if (n_cu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return -1;
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return n_cu;
}
int launch_conv3x3_tiled(const void* x, const void* w, const float* bias, const void* res, void* y, long long boards, int S, int C,
                         int relu, void* st, int f16) {
    if (f16 && (S != CV_S || C != CV_C)) return 1;  // f16 activations: the 9x9 x 128 kernels only
    if (C == C6_C && (S == 17 || S == 9)) {  // 64 filters: 17x17 planes (13x13 Gomoku network, one board per tile) or 9x9 Go (three boards per tile)
        const int n_cu = cu_count();
        if (n_cu < 0) return -1;
        const long long ntiles = S == 17 ? boards : (boards + 2) / 3;
        const dim3 grid((unsigned)(ntiles < n_cu ? ntiles : n_cu)), block(CW_THREADS);
#define AZ_T64(GEO, RES)                                                                                                                   \
    hipLaunchKernelGGL((k_conv3x3_t64<C6Geo<GEO>, RES, 8>), grid, block, 0, (hipStream_t)st, (const unsigned char*)x, (const unsigned short*)w, \
                       bias, (const unsigned char*)res, (unsigned char*)y, (int)ntiles, relu)
        if (S == 17) {
            if (res) AZ_T64(17, true);
            else AZ_T64(17, false);
        } else {
            if (res) AZ_T64(9, true);
            else AZ_T64(9, false);
        }
#undef AZ_T64
        return AZ_HIP(hipGetLastError());
    }
    if (S == C9_S && C == 256) {  // 19x19 boards, 256 filters (jumbo Go network)
        const int n_cu = cu_count();
        if (n_cu < 0) return -1;
        if (x == y || res == y) return 1;  // other CUs read x's halo rows while y is written; the contract keeps the residual apart too
        static const bool two_launch = getenv("AZSP_CONV19_TWO_LAUNCH") != nullptr;  // A/B switch for measurements: round 2-5's two-launch scheme
        if (!two_launch) {
            // ONE launch (round 6, k_conv3x3_op19): a CU = 64 couts x all 256 cin of a half board, cin halves meet through LDS, fp32 end to end;
            // stripes of 8 workgroups (4 cout groups x 2 board halves) per tile stream
            const long long groups = n_cu / 8 > 0 ? n_cu / 8 : 1;
            const long long nst = boards < groups ? boards : groups;
            const dim3 grid((unsigned)(8 * nst)), block(CW_THREADS);
            static const int variant = getenv("AZSP_OP19_VARIANT") ? atoi(getenv("AZSP_OP19_VARIANT")) : C1_VARIANT;  // A/B switch, see C1Sched
#define AZ_OP19(VV)                                                                                                                          \
    case VV:                                                                                                                                 \
        if (res)                                                                                                                             \
            hipLaunchKernelGGL((k_conv3x3_op19<true, VV>), grid, block, 0, (hipStream_t)st, (const unsigned char*)x, (const unsigned short*)w, \
                               bias, (const unsigned char*)res, (unsigned char*)y, (int)boards, relu);                                     \
        else                                                                                                                                 \
            hipLaunchKernelGGL((k_conv3x3_op19<false, VV>), grid, block, 0, (hipStream_t)st, (const unsigned char*)x, (const unsigned short*)w, \
                               bias, (const unsigned char*)nullptr, (unsigned char*)y, (int)boards, relu);                                 \
        break;
            if (variant == 8) {  // the four-way cin split (k_conv3x3_op19q): half the LDS fragment reads per flop
                if (res)
                    hipLaunchKernelGGL((k_conv3x3_op19q<true>), grid, block, 0, (hipStream_t)st, (const unsigned char*)x, (const unsigned short*)w, bias,
                                       (const unsigned char*)res, (unsigned char*)y, (int)boards, relu);
                else
                    hipLaunchKernelGGL((k_conv3x3_op19q<false>), grid, block, 0, (hipStream_t)st, (const unsigned char*)x, (const unsigned short*)w, bias,
                                       (const unsigned char*)nullptr, (unsigned char*)y, (int)boards, relu);
                return AZ_HIP(hipGetLastError());
            }
            switch (variant) {
                AZ_OP19(0) AZ_OP19(6)  // 6: counted lgkmcnt + later barrier, +0.5 % in profiles/r06_conv19_ab.txt (kept for A/B runs)
                default: return 1;
            }
#undef AZ_OP19
            return AZ_HIP(hipGetLastError());
        }
        // two launches, one per 128-channel half of the input; y holds the bf16 partial sum between them
        const long long groups = n_cu / 4 > 0 ? n_cu / 4 : 1;  // stripes of 4 workgroups; a device (partition) with < 4 CUs still gets one stripe
        const long long nst = boards < groups ? boards : groups;
        const dim3 grid((unsigned)(4 * nst)), block(CW_THREADS);
        if (res)
            hipLaunchKernelGGL((k_conv3x3_hb19<true, 16>), grid, block, 0, (hipStream_t)st, (const unsigned char*)x, (const unsigned short*)w, bias,
                               (const unsigned char*)res, (unsigned char*)y, (int)boards, 0, 1, 256, 0, 32, 0);
        else
            hipLaunchKernelGGL((k_conv3x3_hb19<false, 16>), grid, block, 0, (hipStream_t)st, (const unsigned char*)x, (const unsigned short*)w, bias,
                               (const unsigned char*)nullptr, (unsigned char*)y, (int)boards, 0, 1, 256, 0, 32, 0);
        if (AZ_HIP(hipGetLastError())) return -1;  // every launch is checked, not only the last one
        hipLaunchKernelGGL((k_conv3x3_hb19<true, 16>), grid, block, 0, (hipStream_t)st, (const unsigned char*)x, (const unsigned short*)w, bias,
                           (const unsigned char*)y, (unsigned char*)y, (int)boards, relu, 0, 256, 128, 32, 16);
        return AZ_HIP(hipGetLastError());
    }
    if (S != CV_S || C != CV_C) return 1;
    const int n_cu = cu_count();
    if (n_cu < 0) return -1;
    const long long ntiles = (boards + CV_TB - 1) / CV_TB;
    const unsigned grid = (unsigned)(ntiles < n_cu ? ntiles : n_cu);  // one persistent workgroup per CU
#define AZ_CT(RES, F16)                                                                                                              \
    hipLaunchKernelGGL((k_conv3x3_tiled<RES, 16, F16>), dim3(grid), dim3(CW_THREADS), 0, (hipStream_t)st, (const unsigned char*)x, \
                       (const unsigned short*)w, bias, (const unsigned char*)res, (unsigned char*)y, (int)ntiles, relu)
    if (f16) {
        if (res) AZ_CT(true, true);
        else AZ_CT(false, true);
    } else {
        if (res) AZ_CT(true, false);
        else AZ_CT(false, false);
    }
#undef AZ_CT
    return AZ_HIP(hipGetLastError());
}
int launch_resblock_tiled(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, long long boards, int S, int C,
                          void* st) {
    if (C != C6_C || (S != 17 && S != 9)) return 1;
    const int n_cu = cu_count();
    if (n_cu < 0) return -1;
    const long long ntiles = S == 17 ? boards : (boards + 2) / 3;
    const dim3 grid((unsigned)(ntiles < n_cu ? ntiles : n_cu)), block(CW_THREADS);  // one persistent workgroup per CU
    if (S == 17)
        hipLaunchKernelGGL((k_resblock64<C6Geo<17>>), grid, block, 0, (hipStream_t)st, (const unsigned char*)x, (const unsigned short*)w1, b1,
                           (const unsigned short*)w2, b2, (unsigned char*)y, (int)ntiles);
    else
        hipLaunchKernelGGL((k_resblock64<C6Geo<9>>), grid, block, 0, (hipStream_t)st, (const unsigned char*)x, (const unsigned short*)w1, b1,
                           (const unsigned short*)w2, b2, (unsigned char*)y, (int)ntiles);
    return AZ_HIP(hipGetLastError());
}
int launch_stem_tiled(const void* x, const void* w, const float* bias, void* y, long long boards, int S, int C, int pad, int relu, void* st, int f16) {
    const int n_cu = cu_count();
    if (n_cu < 0) return -1;
    if (f16 && (S != CV_S || C != CV_C || pad != 1)) return 1;  // f16 activations: the 9x9 x 128 kernels only
    if (C == C6_C && ((S == 13 && pad == 3) || (S == 9 && pad == 1))) {  // Gomoku: 13x13 boards -> 17x17 planes; Go 9x9 x 64: three boards per tile
        const long long ntiles = S == 13 ? boards : (boards + 2) / 3;
        const dim3 grid((unsigned)(ntiles < n_cu ? ntiles : n_cu)), block(CW_THREADS);
        if (S == 13)
            hipLaunchKernelGGL((k_conv3x3_t64<C6Geo<17>, false, 4>), grid, block, 0, (hipStream_t)st, (const unsigned char*)x, (const unsigned short*)w,
                               bias, (const unsigned char*)nullptr, (unsigned char*)y, (int)ntiles, relu);
        else
            hipLaunchKernelGGL((k_conv3x3_t64<C6Geo<9>, false, 4>), grid, block, 0, (hipStream_t)st, (const unsigned char*)x, (const unsigned short*)w,
                               bias, (const unsigned char*)nullptr, (unsigned char*)y, (int)ntiles, relu);
        return AZ_HIP(hipGetLastError());
    }
    if (S == C9_S && C == 256 && pad == 1) {  // 19x19 Go: 17 planes (padded to 32) -> 256 filters, one launch
        const long long groups = n_cu / 4 > 0 ? n_cu / 4 : 1;
        const long long nst = boards < groups ? boards : groups;
        hipLaunchKernelGGL((k_conv3x3_hb19<false, 4>), dim3((unsigned)(4 * nst)), dim3(CW_THREADS), 0, (hipStream_t)st, (const unsigned char*)x,
                           (const unsigned short*)w, bias, (const unsigned char*)nullptr, (unsigned char*)y, (int)boards, relu, 1, 32, 0, 4, 0);
        return AZ_HIP(hipGetLastError());
    }
    if (S != CV_S || C != CV_C || pad != 1) return 1;
    const long long ntiles = (boards + CV_TB - 1) / CV_TB;
    const unsigned grid = (unsigned)(ntiles < n_cu ? ntiles : n_cu);
    if (f16)
        hipLaunchKernelGGL((k_conv3x3_tiled<false, 4, true>), dim3(grid), dim3(CW_THREADS), 0, (hipStream_t)st, (const unsigned char*)x,
                           (const unsigned short*)w, bias, (const unsigned char*)nullptr, (unsigned char*)y, (int)ntiles, relu);
    else
        hipLaunchKernelGGL((k_conv3x3_tiled<false, 4>), dim3(grid), dim3(CW_THREADS), 0, (hipStream_t)st, (const unsigned char*)x,
                           (const unsigned short*)w, bias, (const unsigned char*)nullptr, (unsigned char*)y, (int)ntiles, relu);
    return AZ_HIP(hipGetLastError());
}
int launch_head_tiled(const void* x, const float* w, const float* bias, void* pol, void* val, long long boards, int S, int C, int npol, int nval,
                      int pol_stride, int val_stride, void* st, int f16) {
    if (C % 8 || C > 1024 || npol + nval != 3) return 1;
    const long long npos = boards * S * S;
    if (f16)
        hipLaunchKernelGGL((k_head_tiled<3, true>), dim3((unsigned)((npos + 255) / 256)), dim3(256), 0, (hipStream_t)st,
                           (const unsigned char*)x, w, bias, (unsigned short*)pol, (unsigned short*)val, npos, npol, C, S * S,
                           cv_tile_boards(S) * S * S, pol_stride, val_stride);
    else
        hipLaunchKernelGGL((k_head_tiled<3>), dim3((unsigned)((npos + 255) / 256)), dim3(256), 0, (hipStream_t)st,
                           (const unsigned char*)x, w, bias, (unsigned short*)pol, (unsigned short*)val, npos, npol, C, S * S,
                           cv_tile_boards(S) * S * S, pol_stride, val_stride);
    return AZ_HIP(hipGetLastError());
}
int launch_fc_heads(const FcHeadsArgs& a, void* st) {
    const int nt1 = (a.A + 31) / 32, nt2 = (a.F + 31) / 32;
    const unsigned grid = (unsigned)((a.boards + 127) / 128);
#define AZ_FC_CASE(T1, T2, F16)                                                                                                         \
    if (nt1 == T1 && nt2 == T2 && (a.f16 != 0) == F16) {                                                                                \
        hipLaunchKernelGGL((k_fc_heads<T1, T2, F16>), dim3(grid), dim3(256), 0, (hipStream_t)st, (const unsigned short*)a.pol,            \
                           (const unsigned short*)a.val, (const unsigned short*)a.wp, a.bp, a.ks1, (const unsigned short*)a.w1, a.b1, a.ks2, \
                           a.w2, a.b2, a.priors, a.values, a.boards, a.A);                                                               \
        return AZ_HIP(hipGetLastError());                                                                                               \
    }
    AZ_FC_CASE(3, 2, false) AZ_FC_CASE(3, 4, false) AZ_FC_CASE(6, 2, false) AZ_FC_CASE(6, 4, false) AZ_FC_CASE(12, 8, false)
    AZ_FC_CASE(3, 4, true)  // f16: the 9x9 x 128 evaluator (82 actions, 128 units)
#undef AZ_FC_CASE
    return 1;
}
int launch_tile_layout(const void* src, void* dst, long long boards, int S, int C, int to_tiled, void* st) {
    if (C % 8 || S < 1) return 1;
    const int nch = C / 8, tile_rows = cv_tile_boards(S) * S * S;
    const long long nchunks = boards * S * S * nch;
    hipLaunchKernelGGL(k_tile_layout, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, (hipStream_t)st, (const unsigned char*)src,
                       (unsigned char*)dst, nchunks, to_tiled, nch, tile_rows);
    return AZ_HIP(hipGetLastError());
}
int launch_split_layout(const void* src, void* dst, long long boards, int S, int C, int to_split, void* st, unsigned* range) {
    if (C % 8 || S < 1) return 1;
    const long long nchunks = boards * S * S * (C / 8);
    hipLaunchKernelGGL(k_split_layout, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, (hipStream_t)st, (const unsigned char*)src,
                       (unsigned char*)dst, nchunks, to_split, C / 8, S * S, range);
    return AZ_HIP(hipGetLastError());
}
template <bool RES, int NCH, int NCG, bool XLO0 = false>
static int launch_sp(const void* x, const void* w, const float* bias, const void* res, void* y, long long boards, int relu, void* st, unsigned* range) {
    const int n_cu = cu_count();
    if (n_cu < 0) return -1;
    long long nslot = n_cu / NCG > 0 ? n_cu / NCG : 1;  // one persistent workgroup per CU; the cout groups of a board run side by side
    if (boards < nslot) nslot = boards;
    hipLaunchKernelGGL((k_conv3x3_sp<RES, NCH, NCG, XLO0>), dim3((unsigned)(nslot * NCG)), dim3(CW_THREADS), 0, (hipStream_t)st, (const unsigned char*)x,
                       (const _Float16*)w, bias, (const unsigned char*)res, (unsigned char*)y, (int)boards, relu, range);
    return AZ_HIP(hipGetLastError());
}
template <bool RES, int NCH, bool XLO0 = false>
static int launch_sp17(const void* x, const void* w, const float* bias, const void* res, void* y, long long boards, int relu, void* st, unsigned* range) {
    const int n_cu = cu_count();
    if (n_cu < 0) return -1;
    const long long nslot = boards < n_cu ? boards : n_cu;  // one persistent workgroup per CU; a board = two half-board tiles
    hipLaunchKernelGGL((k_conv3x3_sp17<RES, NCH, XLO0>), dim3((unsigned)nslot), dim3(CW_THREADS), 0, (hipStream_t)st, (const unsigned char*)x,
                       (const _Float16*)w, bias, (const unsigned char*)res, (unsigned char*)y, (int)boards, relu, range);
    return AZ_HIP(hipGetLastError());
}
// k_conv3x3_spg (az_conv_spg.h): the fp32-class convolution with one WAVE per output tile -- any plane size, 64 / 128 / 256 filters.
// `latency`: small tiles (16 couts x 32 positions per wave) so that a handful of boards fill the chip; otherwise k_conv3x3_spgw:
// 16 NT couts x 48 positions per wave, the B fragments shared by the four waves of a workgroup through LDS.
// `halves` = 2: k_conv3x3_sp2's two accumulation chains (bit-identical to it at 9x9 x 128).
template <bool RES, int KSUB, int NT, int NJ, int HALVES>
static int launch_spg_t(const void* x, const void* w, const float* bias, const void* res, void* y, long long boards, int S, int C, int relu, void* st,
                        unsigned* range) {
    const long long nct = ((long long)S * S + 15) / 16, items = boards * ((nct + NJ - 1) / NJ) * (C / (16 * NT)), grid = (items + 3) / 4;
    if (grid > 0x7fffffffLL) return 1;
    hipLaunchKernelGGL((k_conv3x3_spg<RES, KSUB, NT, NJ, HALVES>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)st, (const unsigned char*)x, (const _Float16*)w,
                       bias, (const unsigned char*)(RES ? res : nullptr), (unsigned char*)y, (int)boards, S, C, relu, range);
    return AZ_HIP(hipGetLastError());
}
// large calls: the four waves of a workgroup share their B fragments through LDS (k_conv3x3_spgw; NT cout tiles per wave, C = 64 NT x groups)
template <bool RES, int KSUB, int NT, int HALVES>
static int launch_spgw_t(const void* x, const void* w, const float* bias, const void* res, void* y, long long boards, int S, int C, int relu, void* st,
                         unsigned* range) {
    const long long nct = ((long long)S * S + 15) / 16, grid = (boards * ((nct + 2) / 3) * (C / (64 * NT)) + 7) / 8 * 8;  // (a multiple of 8: XCD-aware order)
    if (grid > 0x7fffffffLL || C % (64 * NT)) return 1;
    hipLaunchKernelGGL((k_conv3x3_spgw<RES, KSUB, NT, HALVES>), dim3((unsigned)grid), dim3(256), 0, (hipStream_t)st, (const unsigned char*)x, (const _Float16*)w, bias,
                       (const unsigned char*)(RES ? res : nullptr), (unsigned char*)y, (int)boards, S, C, relu, range);
    return AZ_HIP(hipGetLastError());
}
template <int KSUB, int HALVES>
static int launch_spg_k(const void* x, const void* w, const float* bias, const void* res, void* y, long long boards, int S, int C, int relu, void* st,
                        unsigned* range, bool latency) {
    if (latency)
        return res ? launch_spg_t<true, KSUB, 1, 2, HALVES>(x, w, bias, res, y, boards, S, C, relu, st, range)
                   : launch_spg_t<false, KSUB, 1, 2, HALVES>(x, w, bias, res, y, boards, S, C, relu, st, range);
    constexpr int NT = KSUB == 2 ? 1 : 2;
    return res ? launch_spgw_t<true, KSUB, NT, HALVES>(x, w, bias, res, y, boards, S, C, relu, st, range)
               : launch_spgw_t<false, KSUB, NT, HALVES>(x, w, bias, res, y, boards, S, C, relu, st, range);
}
static int launch_spg(const void* x, const void* w, const float* bias, const void* res, void* y, long long boards, int S, int C, int relu, void* st,
                      unsigned* range, bool latency, int halves) {
    if (S < 3 || S > 64 || boards < 1 || boards > 0x7fffffffLL) return 1;
    if (C == 64) return launch_spg_k<2, 1>(x, w, bias, res, y, boards, S, C, relu, st, range, latency);
    if (C == 128)
        return halves == 2 ? launch_spg_k<4, 2>(x, w, bias, res, y, boards, S, C, relu, st, range, latency)
                           : launch_spg_k<4, 1>(x, w, bias, res, y, boards, S, C, relu, st, range, latency);
    if (C == 256) return launch_spg_k<8, 1>(x, w, bias, res, y, boards, S, C, relu, st, range, latency);
    return 1;
}
// Latency tiles (16 couts x 32 positions = one wave) per launch up to which the wave-per-tile kernel replaces a weight-stationary one of
// the same shape (bit-identical results; the weight-stationary kernels give a board to ONE workgroup: 10 - 33 us however few boards
// there are).  Default 1024 = one wave per SIMD of the chip: the measured crossover (tools/spg_ab.py, profiles/r06_spg_ab.txt: the kernel
// reads its fragments through L1, a second wave per SIMD doubles its time).  AZSP_SPG_MAX_WAVES sets the initial value (0 = never),
// azsp_small_batch_waves changes it at run time.
static long long& spg_max_waves_ref() {
    static long long n = [] {
        const char* e = getenv("AZSP_SPG_MAX_WAVES");
        const long long v = e ? atoll(e) : 1024LL;
        return v < 0 ? 0LL : v;
    }();
    return n;
}
static long long spg_max_waves() { return spg_max_waves_ref(); }
static long long spg_latency_waves(long long boards, int S, int C) { return boards * ((((long long)S * S + 15) / 16 + 1) / 2) * (C / 16); }
long long small_batch_waves(long long n) {
    const long long old = spg_max_waves_ref();
    if (n >= 0) spg_max_waves_ref() = n;
    return old;
}
int launch_conv3x3_split(const void* x, const void* w, const float* bias, const void* res, void* y, long long boards, int S, int C, int relu,
                         void* st, unsigned* range) {
    static const bool sp1_env = getenv("AZSP_SP1") != nullptr;
    const bool tailored = (S == Sp17Geo::S && C == 64) || (S == SpGeo9::S && (C == 128 || C == 64));
    if (!tailored) return launch_spg(x, w, bias, res, y, boards, S, C, relu, st, range, spg_latency_waves(boards, S, C) <= spg_max_waves(), 1);
    if (spg_latency_waves(boards, S, C) <= spg_max_waves()) return launch_spg(x, w, bias, res, y, boards, S, C, relu, st, range, true, (C == 128 && !sp1_env) ? 2 : 1);
    if (S == Sp17Geo::S && C == 64)  // 17x17 planes x 64 filters: the 13x13 Gomoku tower (half-board tiles)
        return res ? launch_sp17<true, 8>(x, w, bias, res, y, boards, relu, st, range) : launch_sp17<false, 8>(x, w, bias, res, y, boards, relu, st, range);
    if (S != SpGeo9::S || (C != 128 && C != 64)) return 1;
    if (C == 128) {
        // 9x9 x 128: k_conv3x3_sp2, the 2 x 2 split of a CU's work between its waves (round 6, az_conv_sp2.h: half the LDS fragment reads per MFMA);
        // AZSP_SP1 (read once per process) selects rounds 3-5's k_conv3x3_sp for same-box A/B runs
        static const bool sp1 = getenv("AZSP_SP1") != nullptr;
        if (!sp1) {
            const int n_cu = cu_count();
            if (n_cu < 0) return -1;
            long long nslot = n_cu / 2 > 0 ? n_cu / 2 : 1;
            if (boards < nslot) nslot = boards;
            if (res)
                hipLaunchKernelGGL((k_conv3x3_sp2<true>), dim3((unsigned)(nslot * 2)), dim3(CW_THREADS), 0, (hipStream_t)st, (const unsigned char*)x, (const _Float16*)w, bias,
                                   (const unsigned char*)res, (unsigned char*)y, (int)boards, relu, range);
            else
                hipLaunchKernelGGL((k_conv3x3_sp2<false>), dim3((unsigned)(nslot * 2)), dim3(CW_THREADS), 0, (hipStream_t)st, (const unsigned char*)x, (const _Float16*)w, bias,
                                   (const unsigned char*)nullptr, (unsigned char*)y, (int)boards, relu, range);
            return AZ_HIP(hipGetLastError());
        }
        return res ? launch_sp<true, 16, 2>(x, w, bias, res, y, boards, relu, st, range) : launch_sp<false, 16, 2>(x, w, bias, res, y, boards, relu, st, range);
    }
    return res ? launch_sp<true, 8, 1>(x, w, bias, res, y, boards, relu, st, range) : launch_sp<false, 8, 1>(x, w, bias, res, y, boards, relu, st, range);
}
// scratch for the intermediate activation of ONE 9x9 x 64 board: the odd last board of azsp_resblock_split at 9x9 runs as two unfused
// launches (the fused kernel takes pairs of boards).  One lazily allocated buffer per device, never freed; calls on different streams of one
// device that both end in an odd board would share it -- the evaluator runs one stream per network.
// scratch for the intermediate activations of azsp_resblock_split on a handful of boards (two wave-per-tile convolutions): one lazily
// allocated buffer per device for SPG_SCRATCH_BOARDS boards of 17x17 x 64, never freed; same sharing rule as sp9_tail_scratch.
static constexpr long long SPG_SCRATCH_BOARDS = 256;
static void* spg_block_scratch() {
    static void* buf[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (!buf[dev] && AZ_HIP(hipMalloc(&buf[dev], (size_t)SPG_SCRATCH_BOARDS * 2 * 8 * Sb17::S * Sb17::S * 16))) return nullptr;
    return buf[dev];
}
static void* sp9_tail_scratch() {
    static void* buf[16] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    if (!buf[dev] && AZ_HIP(hipMalloc(&buf[dev], (size_t)Sb9::GTILEB))) return nullptr;
    return buf[dev];
}
int launch_resblock_split(const void* x, const void* w1, const float* b1, const void* w2, const float* b2, void* y, long long boards, int S, int C,
                          void* st, unsigned* range) {
    if (C != 64 || (S != Sb17::S && S != Sb9::S)) return 1;  // 17x17 planes (the 13x13 Gomoku tower) or 9x9 planes (9x9 Go, logs/go/9x9_12b64) x 64 filters
    const int n_cu = cu_count();
    if (n_cu < 0) return -1;
    if (spg_latency_waves(boards, S, C) <= spg_max_waves() && boards <= SPG_SCRATCH_BOARDS) {
        // a handful of boards: two wave-per-tile convolutions (az_conv_spg.h) instead of one workgroup per board -- bit-identical results
        void* mid = spg_block_scratch();
        if (!mid) return -1;
        int rc = launch_spg(x, w1, b1, nullptr, mid, boards, S, C, 1, st, range, true, 1);
        if (rc) return rc;
        return launch_spg(mid, w2, b2, x, y, boards, S, C, 1, st, range, true, 1);
    }
    if (S == Sb9::S) {  // blocks of TWO boards; an odd last board: the two unfused convolutions (bit-identical results)
        const long long pairs = boards / 2;
        if (pairs > 0) {
            const long long nslot = pairs < n_cu ? pairs : n_cu;
            hipLaunchKernelGGL((k_resblock_sp<Sb9, 6>), dim3((unsigned)nslot), dim3(CW_THREADS), 0, (hipStream_t)st, (const unsigned char*)x, (const _Float16*)w1, b1,
                               (const _Float16*)w2, b2, (unsigned char*)y, (int)pairs, range);
            if (AZ_HIP(hipGetLastError())) return -1;
        }
        if (boards & 1) {
            void* mid = sp9_tail_scratch();
            if (!mid) return -1;
            const size_t off = (size_t)(boards - 1) * Sb9::GTILEB;
            int rc = launch_sp<false, 8, 1>((const unsigned char*)x + off, w1, b1, nullptr, mid, 1, 1, st, range);
            if (rc) return rc;
            return launch_sp<true, 8, 1>(mid, w2, b2, (const unsigned char*)x + off, (unsigned char*)y + off, 1, 1, st, range);
        }
        return 0;
    }
    const long long nslot = boards < n_cu ? boards : n_cu;  // one persistent workgroup per CU; a board = two half-board tiles x two phases
    // (the 3-slot ring of the A/B in profiles/r05_pmc_splitblock17_ring3*.txt: build with -DAZSP_EXPERIMENT_RING3 and set AZSP_RB_RING=3)
#ifdef AZSP_EXPERIMENT_RING3
    static const int ring3 = [] {
        const char* e = getenv("AZSP_RB_RING");
        return e && atoi(e) == 3;
    }();
    if (ring3) {
        hipLaunchKernelGGL((k_resblock_sp<Sb17, 3>), dim3((unsigned)nslot), dim3(CW_THREADS), 0, (hipStream_t)st, (const unsigned char*)x, (const _Float16*)w1, b1,
                           (const _Float16*)w2, b2, (unsigned char*)y, (int)boards, range);
        return AZ_HIP(hipGetLastError());
    }
#endif
    hipLaunchKernelGGL((k_resblock_sp<Sb17, 6>), dim3((unsigned)nslot), dim3(CW_THREADS), 0, (hipStream_t)st, (const unsigned char*)x, (const _Float16*)w1, b1,
                       (const _Float16*)w2, b2, (unsigned char*)y, (int)boards, range);
    return AZ_HIP(hipGetLastError());
}
int launch_split_features(const float* src, void* dst, long long boards, int S, int cin, void* st, unsigned* range) {
    if (cin < 1 || cin > 32 || S < 1) return 1;
    const long long nitems = boards * 4 * S * S;
    hipLaunchKernelGGL(k_split_features, dim3((unsigned)((nitems + 255) / 256)), dim3(256), 0, (hipStream_t)st, src, (unsigned char*)dst, nitems, cin,
                       S * S, range);
    return AZ_HIP(hipGetLastError());
}
int launch_stem_split(const void* x, const void* w, const float* bias, void* y, long long boards, int S, int C, int pad, int relu, void* st,
                      int x_lo_zero, unsigned* range) {
    if (S == 13 && C == 64 && pad == 3)  // 13x13 boards -> 17x17 planes
        return x_lo_zero ? launch_sp17<false, 4, true>(x, w, bias, nullptr, y, boards, relu, st, range) : launch_sp17<false, 4>(x, w, bias, nullptr, y, boards, relu, st, range);
    if (S != SpGeo9::S || (C != 128 && C != 64) || pad != 1) return 1;
    if (x_lo_zero)
        return C == 128 ? launch_sp<false, 4, 2, true>(x, w, bias, nullptr, y, boards, relu, st, range) : launch_sp<false, 4, 1, true>(x, w, bias, nullptr, y, boards, relu, st, range);
    return C == 128 ? launch_sp<false, 4, 2>(x, w, bias, nullptr, y, boards, relu, st, range) : launch_sp<false, 4, 1>(x, w, bias, nullptr, y, boards, relu, st, range);
}
int split_range_read(const unsigned* rec, unsigned out[2], int reset, void* st) {
    void* p = (void*)rec;  // null: the per-device default record
    if (!p && AZ_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(g_sp_range)))) return -1;
    if (AZ_HIP(hipMemcpyAsync(out, p, 2 * sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)st))) return -1;
    if (reset && AZ_HIP(hipMemsetAsync(p, 0, 2 * sizeof(unsigned), (hipStream_t)st))) return -1;
    return AZ_HIP(hipStreamSynchronize((hipStream_t)st));
}
template <int BPB> static int launch_head_split_bpb(const HeadSplitArgs& a, void* st) {
    const int P2 = a.S * a.S;
    const size_t lds = (size_t)(BPB * (3 * ((P2 + 3) & ~3) + a.A + a.F)) * sizeof(float);
    if (lds > 64 * 1024) return 1;
    hipLaunchKernelGGL((k_head_split<BPB>), dim3((unsigned)((a.boards + BPB - 1) / BPB)), dim3(256), lds, (hipStream_t)st, (const unsigned char*)a.x, a.hw,
                       a.hb, a.wp_t, a.bp, a.w1_t, a.b1, a.w2, a.b2, a.priors, a.values, a.boards, a.C, P2, a.A, a.F, a.npol);
    return AZ_HIP(hipGetLastError());
}
int launch_head_split(const HeadSplitArgs& a, void* st) {
    if (a.C % 8 || a.npol < 1 || a.npol > 2) return 1;
    return launch_head_split_bpb<4>(a, st);  // (8 boards per workgroup: measured slower, see k_head_split)
}
}  // namespace azb
