// Common device helpers for the HydraNet gfx950 kernels (CDNA4 only: wave64, MFMA, 160 KB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define HN_OK 0
#define HN_ERR_ARG 1
#define HN_ERR_LAUNCH 2
#define HN_ERR_UNSUPPORTED 3

#define HN_CHECK_ARG(cond) do { if (!(cond)) return HN_ERR_ARG; } while (0)
#define HN_LAUNCH_CHECK() do { if (hipGetLastError() != hipSuccess) return HN_ERR_LAUNCH; return HN_OK; } while (0)

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }
__device__ __forceinline__ float bfround(float v) { return (float)((bf16)v); }

__device__ __forceinline__ bf16x8 ld8(const bf16* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ void st8(bf16* p, bf16x8 v) { *reinterpret_cast<bf16x8*>(p) = v; }
__device__ __forceinline__ bf16x8 zero8() {
    bf16x8 z;
#pragma unroll
    for (int i = 0; i < 8; ++i) z[i] = (bf16)0.0f;
    return z;
}

// activation codes shared by the C-ABI
// tuning knobs for tools/ sweeps (hn_debug_knob; defaults = the shipped heuristics): 0 TN split target (workgroups), 1 TN minimum rows per
// split, 2 fused-BatchNorm apply-pass target workgroups, 3 fused reduce-pass row-block divisor, 4 / 5 pixel thresholds of the 64x64 GEMM tile,
// 6 = 1: no two-K-group GEMM variant (> 1: its K threshold, default 512), 7 depthwise-backward block target, 8 = 0: fused passes without the
// XCD row-order placement, 9 grouped-TN debug bits (skip stores / MFMAs / loads), 10 grouped-TN tile variant, 11 = 1: direct 3x3 kernel walks
// the patches of one cout tile first (measured: no gain), 12 / 13 workgroup targets of the 3x3 patch weight-gradient / grouped-conv group plans,
// 16 / 17 channels and rows per workgroup of hn_se_gate_apply (0 = the heuristic), 18 = 1: no narrow-K form of the direct 3x3 kernel
#ifdef HN_TUNING
extern long g_hn_knob[20];
#else
extern const long g_hn_knob[20];      // product build: the shipped heuristics as constants, no setter (the library has no mutable state)
#endif
#define HN_ACT_NONE 0
#define HN_ACT_RELU 1
#define HN_ACT_SWISH 2
#define HN_ACT_ELU 3
#define HN_ACT_SIGMOID 4

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

__device__ __forceinline__ float act_fwd(float x, int act) {
    switch (act) {
        case HN_ACT_RELU: return x > 0.f ? x : 0.f;
        case HN_ACT_SWISH: return x * sigmoidf_(x);
        case HN_ACT_ELU: return x > 0.f ? x : (__expf(x) - 1.0f);
        case HN_ACT_SIGMOID: return sigmoidf_(x);
        default: return x;
    }
}
// derivative w.r.t. the pre-activation x
__device__ __forceinline__ float act_bwd(float x, int act) {
    switch (act) {
        case HN_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case HN_ACT_SWISH: { float s = sigmoidf_(x); return s * (1.f + x * (1.f - s)); }
        case HN_ACT_ELU: return x > 0.f ? 1.f : __expf(x);
        case HN_ACT_SIGMOID: { float s = sigmoidf_(x); return s * (1.f - s); }
        default: return 1.f;
    }
}

// Pyramid levels packed into one [sum_l N*H_l*W_l rows][C] tensor (det-head towers: the five levels share the conv weights, so one
// launch serves all of them).  Level l occupies rows [row_off[l], row_off[l+1]).
#define HN_MAX_LEVELS 5
struct Levels {
    int n;
    int H[HN_MAX_LEVELS], W[HN_MAX_LEVELS];
    long row_off[HN_MAX_LEVELS + 1];
    long work_off[HN_MAX_LEVELS + 1];      // kernel-specific cumulative work items per level
};
__device__ __forceinline__ int level_of_row(const Levels& L, long row) {
    int lv = 0;
    while (lv + 1 < L.n && row >= L.row_off[lv + 1]) ++lv;
    return lv;
}

// Array forms: ONE uniform branch on `act` around an unrolled loop.  A per-element switch costs a scalar branch chain per value, which
// dominated the GEMM epilogues (~1000 branches per wave for a 64x64 wave tile); hoisting it is worth 2-3x on short-K launches.
template <int N>
__device__ __forceinline__ void act_fwd_n(float (&v)[N], int act) {
    if (act == HN_ACT_NONE) return;
    if (act == HN_ACT_RELU) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
    } else if (act == HN_ACT_SWISH) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = v[k] * sigmoidf_(v[k]);
    } else if (act == HN_ACT_ELU) {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = v[k] > 0.f ? v[k] : (__expf(v[k]) - 1.0f);
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = sigmoidf_(v[k]);
    }
}
// g[k] *= act'(x[k])
template <int N>
__device__ __forceinline__ void act_bwd_n(const float (&x)[N], float (&g)[N], int act) {
    if (act == HN_ACT_NONE) return;
    if (act == HN_ACT_RELU) {
#pragma unroll
        for (int k = 0; k < N; ++k) g[k] = x[k] > 0.f ? g[k] : 0.f;
    } else if (act == HN_ACT_SWISH) {
#pragma unroll
        for (int k = 0; k < N; ++k) { const float s = sigmoidf_(x[k]); g[k] *= s * (1.f + x[k] * (1.f - s)); }
    } else if (act == HN_ACT_ELU) {
#pragma unroll
        for (int k = 0; k < N; ++k) g[k] *= x[k] > 0.f ? 1.f : __expf(x[k]);
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) { const float s = sigmoidf_(x[k]); g[k] *= s * (1.f - s); }
    }
}

// XCD-aware block order (8 XCDs, private L2s, workgroups dealt round-robin): hardware id -> logical id such that consecutive LOGICAL
// ids run on one XCD.  Inside a launch, tiles that share an operand panel hit that XCD's L2; ACROSS launches it is a placement convention:
// every kernel walks its tensor in row order with this mapping, so XCD k owns rows [k M / 8, (k + 1) M / 8) (images 2k, 2k+1 of a batch of
// 16, at every pyramid level) in the producer and in the consumer, and what a launch wrote with plain stores is still in the L2 the next
// launch reads it through (measured: +1 % on the whole step from the BatchNorm passes alone).  Bijective for any grid size.  Speed only.
__device__ __forceinline__ int xcd_remap(int hw, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, x = hw & 7;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (hw >> 3);
}

// sum over each row of 16 lanes (all 16 get the total): DPP row rotations are plain VALU ops, ~10x cheaper than ds_bpermute shuffles
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0x128>(v);   // row_ror:8
    v += dpp_mov<0x124>(v);   // row_ror:4
    v += dpp_mov<0x122>(v);   // row_ror:2
    v += dpp_mov<0x121>(v);   // row_ror:1
    return v;
}

// sum over the 64 lanes of a wave (every lane gets the total; call with all lanes active): DPP row sums, then the four rows through
// v_readlane.  (Six __shfl_xor steps are six DEPENDENT ds_bpermute round trips through the LDS pipe, ~0.4 us per sum: the eight sums of
// hn_se_gate_apply's prologue were 3 us of its 3.8.)
__device__ __forceinline__ float wave_sum(float v) {
    const int b = __builtin_bit_cast(int, row16_sum(v));
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Opt-in for more than 64 KiB of dynamic LDS.  The attribute is per (kernel, device): `done` is one bit per device id, set after every
// kernel of the list accepted the attribute on the calling thread's current device (atomic: backward runs on autograd worker threads).
// Returns false when the runtime refuses (the caller reports HN_ERR_LAUNCH instead of launching a kernel that cannot get its LDS).
#include <atomic>
#include <initializer_list>
static inline bool lds_optin(std::atomic<unsigned long long>& done, std::initializer_list<const void*> kernels) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    const unsigned long long bit = 1ull << dev;
    if (done.load(std::memory_order_acquire) & bit) return true;
    for (const void* k : kernels)
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return false;
    done.fetch_or(bit, std::memory_order_release);
    return true;
}
