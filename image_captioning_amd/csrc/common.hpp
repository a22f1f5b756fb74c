// Shared device helpers for the gfx950 captioning kernels.
// Everything here is wave64 / CDNA4 specific; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icap {

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_SWISH = 2, ACT_SIGMOID = 3, ACT_TANH = 4 };

// Error codes returned across the C ABI (0 = ok). Values >= 1000 are
// 1000 + hipError_t of a failed launch.
enum Err : int {
    ICAP_OK = 0,
    ICAP_ERR_ARG = 1,        // null pointer / non-positive size / unsupported shape
    ICAP_ERR_ALIGN = 2,      // pointer or channel count not aligned as the kernel needs
    ICAP_ERR_LIMIT = 3,      // a size exceeds a compiled-in maximum
    ICAP_ERR_HIP_BASE = 1000
};

void set_error(const char* fmt, ...);

inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return ICAP_ERR_HIP_BASE + (int)e;
    }
    return ICAP_OK;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// Sigmoid for the activation epilogues of the CNN (hundreds of millions of elements per batch):
// hardware exp2 and reciprocal (about 1 ulp each) instead of the ~40-instruction libm expf + division,
// which made the swish epilogue of the 1x1 convolutions VALU-bound.
__device__ __forceinline__ float sigmoid_fast(float v) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
}

// tanh for the additive-attention scores (batch x batch x width evaluations per decode step, three times that in a training
// step): 1 - 2 / (exp(2v) + 1) on the hardware exp2 and reciprocal, absolute error about 2e-7 (the libm tanhf is ~30
// instructions with two paths); saturates to +-1 through exp2 -> inf / 0.
__device__ __forceinline__ float tanh_fast(float v) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * v));
}

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_SWISH: return v * sigmoid_fast(v);
        case ACT_SIGMOID: return sigmoid_fast(v);
        case ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// Compile-time activation: kernels dispatch on the runtime code ONCE per tile / thread and run a
// straight-line epilogue; a per-element runtime switch costs a scalar branch tree per value and keeps
// the compiler from forming 16-byte stores.
template <int ACT>
__device__ __forceinline__ float act_c(float v) {
    if constexpr (ACT == ACT_RELU) return fmaxf(v, 0.f);
    else if constexpr (ACT == ACT_SWISH) return v * sigmoid_fast(v);
    else if constexpr (ACT == ACT_SIGMOID) return sigmoid_fast(v);
    else if constexpr (ACT == ACT_TANH) return tanhf(v);
    else return v;
}

template <int ACT>
struct ActTag { static constexpr int value = ACT; };

// Calls f(ActTag<act>{}) for the runtime activation code.
template <typename F>
__device__ __forceinline__ void dispatch_act(int act, F&& f) {
    switch (act) {
        case ACT_RELU: f(ActTag<ACT_RELU>{}); break;
        case ACT_SWISH: f(ActTag<ACT_SWISH>{}); break;
        case ACT_SIGMOID: f(ActTag<ACT_SIGMOID>{}); break;
        case ACT_TANH: f(ActTag<ACT_TANH>{}); break;
        default: f(ActTag<ACT_NONE>{}); break;
    }
}

// Wave-wide reductions.  All 64 lanes of the wave must be active.  Lanes exchange
// values with DPP modifiers inside each row of 16 (a few cycles each, against
// >100 cycles for every LDS-crossbar ds_bpermute a shuffle compiles to) and the four
// row results meet through v_readlane; every lane ends with the result.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) {
    return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true);
}
constexpr int DPP_QUAD_SWAP1 = 0xB1;   // quad_perm [1,0,3,2]
constexpr int DPP_QUAD_SWAP2 = 0x4E;   // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // reverse within 8 lanes
constexpr int DPP_ROW_MIRROR = 0x140;  // reverse within 16 lanes

__device__ __forceinline__ float lane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<DPP_QUAD_SWAP1>(v);
    v += dpp_f<DPP_QUAD_SWAP2>(v);
    v += dpp_f<DPP_HALF_MIRROR>(v);
    v += dpp_f<DPP_ROW_MIRROR>(v);
    return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}

__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<DPP_QUAD_SWAP1>(v));
    v = fmaxf(v, dpp_f<DPP_QUAD_SWAP2>(v));
    v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<DPP_ROW_MIRROR>(v));
    return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}

// (value, index) max with first-index tie-break, the order torch.argmax uses.
__device__ __forceinline__ void argmax_combine(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}

template <int CTRL>
__device__ __forceinline__ void argmax_dpp(float& v, int& i) {
    const float ov = dpp_f<CTRL>(v);
    const int oi = dpp_i<CTRL>(i);
    argmax_combine(v, i, ov, oi);
}

__device__ __forceinline__ void wave_argmax(float& v, int& i) {
    argmax_dpp<DPP_QUAD_SWAP1>(v, i);
    argmax_dpp<DPP_QUAD_SWAP2>(v, i);
    argmax_dpp<DPP_HALF_MIRROR>(v, i);
    argmax_dpp<DPP_ROW_MIRROR>(v, i);
    float bv = lane_f(v, 0);
    int bi = __builtin_amdgcn_readlane(i, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16) argmax_combine(bv, bi, lane_f(v, r), __builtin_amdgcn_readlane(i, r));
    v = bv;
    i = bi;
}

__device__ __forceinline__ float dot4(float4 a, float4 b, float acc) {
    acc = fmaf(a.x, b.x, acc);
    acc = fmaf(a.y, b.y, acc);
    acc = fmaf(a.z, b.z, acc);
    acc = fmaf(a.w, b.w, acc);
    return acc;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Running statistics of a set of values: count, sum, and squared deviations about the set's own mean.  stat_merge adds a
// second set (Chan's update), so no squares of un-centred values are ever formed.
struct ColStat { float n, s, q; };

__device__ __forceinline__ void stat_merge(ColStat& a, float nb, float sb, float qb) {
    if (nb <= 0.f) return;
    if (a.n <= 0.f) { a.n = nb; a.s = sb; a.q = qb; return; }
    const float n = a.n + nb;
    const float d = sb / nb - a.s / a.n;
    a.q = a.q + qb + d * d * (a.n * nb / n);
    a.s += sb;
    a.n = n;
}

// Column statistics in a GEMM epilogue (for the train-mode BatchNorm that follows the GEMM).  While a pass of the tile
// sits in LDS every thread already walks its rows with four columns in registers; it accumulates the moments of those values
// about `shift` - the pass's first row, a sample of every column, so the later subtraction stays benign - and
// stat_from_shifted turns a column's totals over nrows rows into (sum, squared deviations about the mean).
__device__ __forceinline__ void stat_acc4(float4& s1, float4& s2, float4 v, float4 shift) {
    const float dx = v.x - shift.x, dy = v.y - shift.y, dz = v.z - shift.z, dw = v.w - shift.w;
    s1.x += dx; s1.y += dy; s1.z += dz; s1.w += dw;
    s2.x = fmaf(dx, dx, s2.x); s2.y = fmaf(dy, dy, s2.y); s2.z = fmaf(dz, dz, s2.z); s2.w = fmaf(dw, dw, s2.w);
}

__device__ __forceinline__ void stat_from_shifted(float nrows, float shift, float s1, float s2, float* sum, float* m2) {
    *sum = fmaf(nrows, shift, s1);
    *m2 = nrows > 0.f ? fmaxf(s2 - s1 * s1 / nrows, 0.f) : 0.f;
}

// Counter-based uniform in [0, 1): element idx of stream `seed` (a splitmix64 finaliser of seed + idx * golden ratio).
__device__ __forceinline__ float uniform01(unsigned long long seed, unsigned long long idx) {
    unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    z = z ^ (z >> 31);
    return (float)(z >> 40) * (1.0f / 16777216.0f);
}

}  // namespace icap
